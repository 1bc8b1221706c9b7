// Micro-benchmark: same-wave interleave of snake-epilogue VALU work into the gaps of an MFMA stream.
// One wave per SIMD (256 threads, 1 workgroup per CU).  Per "layer": 16 k-steps x 4 MFMAs (x frags
// from LDS) and an epilogue of 64 values (mul, v_sin, fma, cvt_pk) belonging to ANOTHER accumulator set.
//   mode 0: MFMAs only   mode 1: epilogue only   mode 2: MFMA block, then epilogue block (serial)
//   mode 3: epilogue slices placed between the k-steps (4 values per k-step)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(int layers, const bf16x8* __restrict__ w, float* out) {
  __shared__ __attribute__((aligned(16))) char lds[65536];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 65536 / 16; i += 256) ((bf16x8*)lds)[i] = w[i & 1023];
  __syncthreads();
  f32x16 accA[2][2], accB[2][2];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) { accA[a][b][r] = 0.f; accB[a][b][r] = 0.01f * (r + lane); }
  bf16x8 wf[2];
  wf[0] = w[lane]; wf[1] = w[64 + lane];
  for (int l = 0; l < layers; ++l) {
    uint32_t packed[2][2][8];
    bf16x8 xn0, xn1;
    float sn[2][2];
    if (MODE != 1) {
      xn0 = *(const bf16x8*)(lds + ((0 * 2 + 0) * 64 + lane) * 16 + (l & 1) * 32768);
      xn1 = *(const bf16x8*)(lds + ((0 * 2 + 1) * 64 + lane) * 16 + (l & 1) * 32768);
    }
    if (MODE == 4 || MODE == 5) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int bt = 0; bt < 2; ++bt) sn[nt][bt] = __builtin_amdgcn_sinf(accB[nt][bt][0] * 0.15915494f);
    }
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      if (MODE != 1) {
        const bf16x8 x0 = xn0, x1 = xn1;
        if (ks < 15) {
          xn0 = *(const bf16x8*)(lds + (((ks + 1) * 2 + 0) * 64 + lane) * 16 + (l & 1) * 32768);
          xn1 = *(const bf16x8*)(lds + (((ks + 1) * 2 + 1) * 64 + lane) * 16 + (l & 1) * 32768);
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          accA[nt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nt], x0, accA[nt][0], 0, 0, 0);
          accA[nt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nt], x1, accA[nt][1], 0, 0, 0);
        }
      }
      if (MODE == 5) {          // as 4, with an enforced 1 MFMA : 4 VALU issue pattern (sched_group_barrier)
        const int r = ks;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int bt = 0; bt < 2; ++bt) {
            const float s0 = sn[nt][bt];
            if (r < 15) sn[nt][bt] = __builtin_amdgcn_sinf(accB[nt][bt][r + 1] * 0.15915494f);
            accB[nt][bt][r] = fmaf(s0, s0, accB[nt][bt][r]);
            if (r & 1) {
              const bf16x2 p = {(__bf16)accB[nt][bt][r - 1], (__bf16)accB[nt][bt][r]};
              packed[nt][bt][r >> 1] = __builtin_bit_cast(uint32_t, p);
            }
          }
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // the two x reads of the next k-step
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);  // four vector ALU fillers in its shadow
        }
      }
      if (MODE == 4) {          // software-pipelined: sin of r+1 issued before the fma of r consumes sin of r
        const int r = ks;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int bt = 0; bt < 2; ++bt) {
            const float s0 = sn[nt][bt];
            if (r < 15) sn[nt][bt] = __builtin_amdgcn_sinf(accB[nt][bt][r + 1] * 0.15915494f);
            accB[nt][bt][r] = fmaf(s0, s0, accB[nt][bt][r]);
            if (r & 1) {
              const bf16x2 p = {(__bf16)accB[nt][bt][r - 1], (__bf16)accB[nt][bt][r]};
              packed[nt][bt][r >> 1] = __builtin_bit_cast(uint32_t, p);
            }
          }
      }
      if (MODE == 3 || MODE == 1) {
        const int r = ks;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int bt = 0; bt < 2; ++bt) {
            const float z = accB[nt][bt][r];
            const float s = __builtin_amdgcn_sinf(z * 0.15915494f);
            accB[nt][bt][r] = fmaf(s, s, z);
            if (r & 1) {
              const bf16x2 p = {(__bf16)accB[nt][bt][r - 1], (__bf16)accB[nt][bt][r]};
              packed[nt][bt][r >> 1] = __builtin_bit_cast(uint32_t, p);
            }
          }
      }
    }
    if (MODE == 2) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int bt = 0; bt < 2; ++bt) {
            const float z = accB[nt][bt][r];
            const float s = __builtin_amdgcn_sinf(z * 0.15915494f);
            accB[nt][bt][r] = fmaf(s, s, z);
            if (r & 1) {
              const bf16x2 p = {(__bf16)accB[nt][bt][r - 1], (__bf16)accB[nt][bt][r]};
              packed[nt][bt][r >> 1] = __builtin_bit_cast(uint32_t, p);
            }
          }
    }
    if (MODE != 0) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int bt = 0; bt < 2; ++bt)
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            uint4 v = {packed[nt][bt][4 * s], packed[nt][bt][4 * s + 1], packed[nt][bt][4 * s + 2], packed[nt][bt][4 * s + 3]};
            *(uint4*)(lds + ((((wave * 2 + nt) * 2 + s) * 2 + bt) * 64 + lane) * 16 + ((l + 1) & 1) * 32768) = v;
          }
    }
    __syncthreads();
  }
  float s = 0.f;
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += accA[a][b][r] + accB[a][b][r];
  if (s == 123.456f) out[threadIdx.x] = s;
}

template <int MODE> float run(const bf16x8* w, float* out, int layers) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, 4, w, out);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, layers, w, out);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6f / layers;
}
int main() {
  bf16x8* w; float* out;
  (void)hipMalloc(&w, 1 << 20); (void)hipMemset(w, 0, 1 << 20); (void)hipMalloc(&out, 4096);
  const int layers = 2000;
  printf("per layer (64 MFMAs = 2048 pipe cycles; 64-value snake epilogue)\n");
  printf("mode 0 mfma only      %.0f ns\n", run<0>(w, out, layers));
  printf("mode 1 epilogue only  %.0f ns\n", run<1>(w, out, layers));
  printf("mode 2 serial         %.0f ns\n", run<2>(w, out, layers));
  printf("mode 3 interleaved    %.0f ns\n", run<3>(w, out, layers));
  printf("mode 4 interleaved+sw pipelined %.0f ns\n", run<4>(w, out, layers));
  printf("mode 5 + sched_group_barrier    %.0f ns\n", run<5>(w, out, layers));
  return 0;
}
