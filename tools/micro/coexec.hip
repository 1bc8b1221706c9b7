// Micro-benchmark: do MFMA from one wave and VALU (fma + v_sin + cvt) from another wave of the same
// SIMD overlap?  8 waves per workgroup, one workgroup per CU: waves 0-3 (one per SIMD) run MFMAs,
// waves 4-7 run vector ALU work.  Modes: 0 = MFMA only, 1 = VALU only, 2 = both concurrently,
// 3 = every wave alternates MFMA block / VALU block in lockstep (what the fused kernel does today),
// 4 = as 3 but waves 4-7 run the opposite phase (anti-phase).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ void mfma_block(f32x16 (&acc)[4], bf16x8 a, bf16x8 b, int n) {
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[q], 0, 0, 0);
  }
}
__device__ __forceinline__ void valu_block(float (&v)[16], int n) {
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float s = __builtin_amdgcn_sinf(v[q] * 0.15915494f);
      v[q] = fmaf(s, s, v[q]);
    }
  }
}

__global__ __launch_bounds__(512, 1) void k(int mode, int iters, int nm, int nv, float* out) {
  const int wave = threadIdx.x >> 6;
  f32x16 acc[4];
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (threadIdx.x + j)); b[j] = (__bf16)(0.002f * j); }
  float v[16];
  for (int q = 0; q < 16; ++q) v[q] = 0.01f * (threadIdx.x + q);
  const bool first = wave < 4;
  for (int it = 0; it < iters; ++it) {
    if (mode == 0) { if (first) mfma_block(acc, a, b, nm); }
    else if (mode == 1) { if (!first) valu_block(v, nv); }
    else if (mode == 2) { if (first) mfma_block(acc, a, b, nm); else valu_block(v, nv); }
    else if (mode == 3) { mfma_block(acc, a, b, nm / 2); __builtin_amdgcn_s_barrier(); valu_block(v, nv / 2); __builtin_amdgcn_s_barrier(); }
    else {
      if (first) mfma_block(acc, a, b, nm / 2); else valu_block(v, nv / 2);
      __builtin_amdgcn_s_barrier();
      if (first) valu_block(v, nv / 2); else mfma_block(acc, a, b, nm / 2);
      __builtin_amdgcn_s_barrier();
    }
  }
  float s = 0.f;
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
  for (int q = 0; q < 16; ++q) s += v[q];
  if (s == 123.456f) out[threadIdx.x] = s;
}

int main() {
  float* out; hipMalloc(&out, 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 200, nm = 16, nv = 10;     // per iteration: 64 MFMAs (2048 pipe cycles) vs 160 sin+fma+mul
  for (int nvv : {5, 10, 20}) {
    for (int mode = 0; mode < 5; ++mode) {
      hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, 10, nm, nvv, out);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, iters, nm, nvv, out);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("nv=%d mode %d: %.1f us  (%.0f ns / iteration)\n", nvv, mode, ms * 1e3, ms * 1e6 / iters);
    }
  }
  return 0;
}
