// tools/micro/fp8_probe.hip -- round 6: what the 8-bit weight-gradient path relies on, read from the hardware.
//   (1) ds_read_b64_tr_b8: which LDS bytes each lane of a 16-lane group receives
//   (2) v_mfma_scale_f32_32x32x64_f8f6f4: the K position of every (lane half, operand byte) of A and of B (one-hot probes),
//       which output rows one lane's scale byte touches
//   (3) v_cvt_pk_fp8_f32 / v_cvt_pk_bf8_f32 on out-of-range inputs
//   (4) cycles per MFMA: scaled 32x32x64 (bf8 x fp8) against 32x32x16 bf16
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/fp8_probe.hip -o build_ab/fp8_probe ; run on the GPU box, prints text.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) v2i lds_v2i;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void tr8_kernel(int* out, int pass) {
  __shared__ __attribute__((aligned(16))) unsigned char sm[512];
  for (int i = threadIdx.x; i < 512; i += 64) sm[i] = pass ? (unsigned char)(i >> 8) : (unsigned char)(i & 255);
  __syncthreads();
  const int lane = threadIdx.x;
  v2i t = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2i*)(sm + (lane >> 4) * 128 + (lane & 15) * 8));
  out[lane * 2] = t[0];
  out[lane * 2 + 1] = t[1];
}

// one-hot probe: A has 1.0 at (lane half ha, byte ea) of every row lane; B has 1.0 at (hb, eb) of every column lane.
// D[i][j] = 1 for all i, j iff k_A(ha, ea) == k_B(hb, eb).
__global__ void onehot_kernel(float* out) {
  const int pa = blockIdx.x, pb = blockIdx.y;           // 0..63: (half, byte 0..31)
  const int lane = threadIdx.x, h = lane >> 5;
  v8i a = {}, b = {};
  if (h == (pa >> 5)) { const int e = pa & 31; a[e >> 2] = 0x38 << (8 * (e & 3)); }
  if (h == (pb >> 5)) { const int e = pb & 31; b[e >> 2] = 0x38 << (8 * (e & 3)); }
  v16f c = {};
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += c[r];
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (lane == 0) out[pa * 64 + pb] = s;
}

// scale probe: all-ones operands (D = 64 at unit scale); lane `sl` of A (or B) carries scale byte 128 (= 2.0) in byte 0
__global__ void scale_kernel(float* out, int sl, int which) {
  const int lane = threadIdx.x;
  v8i a, b;
  for (int i = 0; i < 8; ++i) { a[i] = 0x38383838; b[i] = 0x38383838; }
  const int sa = (which == 0 && lane == sl) ? 0x7f7f7f80 : 0x7f7f7f7f;
  const int sb = (which == 1 && lane == sl) ? 0x7f7f7f80 : 0x7f7f7f7f;
  v16f c = {};
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
  for (int r = 0; r < 16; ++r) out[(r * 64 + lane)] = c[r];
}

// mixed formats: A bf8 (cbsz 1) value 1.0 = 0x3c, B fp8 1.0 = 0x38, uniform scale byte sA -> expect 64 * 2^(sA - 127)
__global__ void mixed_kernel(float* out, int sa_byte) {
  v8i a, b;
  for (int i = 0; i < 8; ++i) { a[i] = 0x3c3c3c3c; b[i] = 0x38383838; }
  v16f c = {};
  const int sa = sa_byte * 0x01010101;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 1, 0, 0, sa, 0, 0x7f7f7f7f);
  if (threadIdx.x == 0) out[0] = c[0];
}

__global__ void cvt_kernel(const float* in, int n, int* out) {
  const int i = threadIdx.x;
  if (i < n) {
    out[2 * i] = __builtin_amdgcn_cvt_pk_fp8_f32(in[i], in[i], 0, false) & 0xffff;
    out[2 * i + 1] = __builtin_amdgcn_cvt_pk_bf8_f32(in[i], in[i], 0, false) & 0xffff;
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(long long* out, int iters, const int* seed) {
  v8i a, b;
  for (int i = 0; i < 8; ++i) { a[i] = seed[threadIdx.x & 63] * (i + 3); b[i] = seed[(threadIdx.x + 7) & 63] * (i + 5); }
  v16f c[4] = {};
  bf16x8 a16 = __builtin_bit_cast(bf16x8, (int __attribute__((ext_vector_type(4)))){a[0], a[1], a[2], a[3]});
  bf16x8 b16 = __builtin_bit_cast(bf16x8, (int __attribute__((ext_vector_type(4)))){b[0], b[1], b[2], b[3]});
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (MODE == 0) c[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a16, b16, c[q], 0, 0, 0);
      else c[q] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c[q], 1, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += c[q][r];
  if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = (long long)s; }
}

int main() {
  int* d_i; float* d_f; long long* d_l;
  CK(hipMalloc(&d_i, 1 << 20)); CK(hipMalloc(&d_f, 1 << 20)); CK(hipMalloc(&d_l, 1 << 16));
  // (1)
  std::vector<int> pa(128), pb(128);
  hipLaunchKernelGGL(tr8_kernel, dim3(1), dim3(64), 0, 0, d_i, 0); CK(hipMemcpy(pa.data(), d_i, 512, hipMemcpyDeviceToHost));
  hipLaunchKernelGGL(tr8_kernel, dim3(1), dim3(64), 0, 0, d_i, 1); CK(hipMemcpy(pb.data(), d_i, 512, hipMemcpyDeviceToHost));
  printf("== (1) ds_read_b64_tr_b8: lane l of 16-lane group g reads address g*128 + 8*(l&15); bytes received = LDS offsets\n");
  int hyp_ok = 1;
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int e = 0; e < 8; ++e) {
      const int lo = (pa[l * 2 + (e >> 2)] >> (8 * (e & 3))) & 255, hi = (pb[l * 2 + (e >> 2)] >> (8 * (e & 3))) & 255;
      const int off = lo + 256 * hi;
      printf(" %3d", off);
      // hypothesis: lane i of the group gets byte (i & 7) of chunk 2 e + (i >> 3)
      const int i = l & 15, want = (l >> 4) * 128 + (2 * e + (i >> 3)) * 8 + (i & 7);
      if (off != want) hyp_ok = 0;
    }
    printf("\n");
  }
  printf("hypothesis 'lane i <- byte (i&7) of chunk 2e+(i>>3), element e' %s\n", hyp_ok ? "HOLDS" : "FAILS");
  // (2)
  std::vector<float> oh(4096);
  hipLaunchKernelGGL(onehot_kernel, dim3(64, 64), dim3(64), 0, 0, d_f); CK(hipMemcpy(oh.data(), d_f, 4096 * 4, hipMemcpyDeviceToHost));
  int sym = 1, nmatch = 0;
  for (int a = 0; a < 64; ++a) for (int b = 0; b < 64; ++b) { const bool m = oh[a * 64 + b] != 0.f; nmatch += m; if (m != (a == b)) sym = 0; }
  printf("== (2) one-hot K probe: %d matching (A pos, B pos) pairs of 4096; 'k_A(h,e) == k_B(h,e) and nothing else' %s\n", nmatch, sym ? "HOLDS" : "FAILS");
  if (!sym) for (int a = 0; a < 64; ++a) { printf("A pos %2d matches B pos:", a); for (int b = 0; b < 64; ++b) if (oh[a * 64 + b] != 0.f) printf(" %d(%g)", b, oh[a * 64 + b]); printf("\n"); }
  else printf("sum at a match = %g (expect 1024 = 32x32 ones)\n", oh[0]);
  std::vector<float> sc(1024);
  for (int which = 0; which < 2; ++which)
    for (int sl : {0, 5, 37}) {
      hipLaunchKernelGGL(scale_kernel, dim3(1), dim3(64), 0, 0, d_f, sl, which); CK(hipMemcpy(sc.data(), d_f, 4096, hipMemcpyDeviceToHost));
      printf("scale %c lane %2d = 2.0:", which ? 'B' : 'A', sl);
      // D element (row i, col j): lane = j + 32 * ((i >> 2) & 1), reg = (i & 3) + 4 * (i >> 3)
      int shown = 0;
      for (int i = 0; i < 32 && shown < 6; ++i) for (int j = 0; j < 32 && shown < 6; ++j) {
        const int lane = j + 32 * ((i >> 2) & 1), reg = (i & 3) + 4 * (i >> 3);
        const float v = sc[reg * 64 + lane];
        if (v != 64.f) { printf(" D[%d][%d]=%g", i, j, v); ++shown; }
      }
      int cnt = 0; for (int q = 0; q < 1024; ++q) cnt += sc[q] != 64.f;
      printf("  (%d of 1024 elements differ from 64)\n", cnt);
    }
  for (int sb : {127, 120, 133}) {
    hipLaunchKernelGGL(mixed_kernel, dim3(1), dim3(64), 0, 0, d_f, sb); float v; CK(hipMemcpy(&v, d_f, 4, hipMemcpyDeviceToHost));
    printf("bf8 x fp8 ones, scale A byte %d: D = %g (expect %g)\n", sb, v, 64.0 * __builtin_exp2(sb - 127.0));
  }
  // (3)
  const float vals[] = {1e-10f, 7.6e-6f, 1.53e-5f, 6.1e-5f, 0.0019f, 0.002f, 0.0156f, 1.0f, -1.0f, 447.f, 448.f, 464.f, 465.f, 500.f, 6e4f, 57344.f, 61440.f, 61441.f, 1e5f, 1e9f, __builtin_inff(), -__builtin_inff(), __builtin_nanf("")};
  const int nv = sizeof(vals) / sizeof(float);
  CK(hipMemcpy(d_f, vals, sizeof(vals), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(cvt_kernel, dim3(1), dim3(64), 0, 0, d_f, nv, d_i);
  std::vector<int> cv(2 * nv); CK(hipMemcpy(cv.data(), d_i, 8 * nv, hipMemcpyDeviceToHost));
  printf("== (3) v_cvt_pk_fp8_f32 / v_cvt_pk_bf8_f32 (low byte)\n");
  for (int i = 0; i < nv; ++i) printf("  %12g -> fp8 0x%02x  bf8 0x%02x\n", vals[i], cv[2 * i] & 255, cv[2 * i + 1] & 255);
  // (4)
  std::vector<int> seed(64); for (int i = 0; i < 64; ++i) seed[i] = 0x3a3b3c3d + i * 0x01010101;
  CK(hipMemcpy(d_i, seed.data(), 256, hipMemcpyHostToDevice));
  for (int mode = 0; mode < 2; ++mode) {
    const int iters = 2000, blocks = 256;
    for (int rep = 0; rep < 2; ++rep) {
      if (mode == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(blocks), dim3(256), 0, 0, d_l, iters, d_i);
      else hipLaunchKernelGGL(rate_kernel<1>, dim3(blocks), dim3(256), 0, 0, d_l, iters, d_i);
    }
    std::vector<long long> t(2 * blocks); CK(hipMemcpy(t.data(), d_l, 16 * blocks, hipMemcpyDeviceToHost));
    long long mn = t[0]; for (int b = 0; b < blocks; ++b) mn = t[2 * b] < mn ? t[2 * b] : mn;
    printf("== (4) %s: %.1f s_memtime ticks per MFMA per wave (one wave per SIMD, 4 accumulators, 256 workgroups)\n",
           mode ? "scaled 32x32x64 bf8 x fp8" : "32x32x16 bf16", (double)mn / (iters * 4));
  }
  CK(hipDeviceSynchronize());
  printf("done\n");
  return 0;
}
