// npp_search.hip -- SURVEY.md 8 f4: the brute-force displacement search of the periodicity proposal,
// compute_loss of NPP_proposal/feature_searching.py:208-264.  For every candidate displacement (dx, dy):
//   loss = sum_{y,x} mask[y,x] mask[y+dy,x+dx] sum_{c < C-1} f(act_c[y+dy,x+dx], act_c[y,x]),
//   f = -a b (edge_searching) or (a - b)^2, zero outside the map (the reference indexes a zero-padded canvas).
// The reference materialises, per batch of displacements, a gathered (bs, C, h, w) copy of the map and three more
// temporaries of that size; here one workgroup owns kSearchSH displacements, walks the map once (the unshifted values are
// loaded once for all of them), and reduces in registers -> shuffles -> LDS.  The map (C h w floats, ~1 MB) stays in L2.
#include "npp_common.h"

namespace npp {

constexpr int kSearchSH = 4;      // displacements per workgroup

__global__ __launch_bounds__(256) void shift_search_kernel(const float* __restrict__ act, const float* __restrict__ mask, int C, int h,
                                                           int w, const int32_t* __restrict__ shifts, int n, int edge,
                                                           float* __restrict__ losses) {
  __shared__ float red[4][kSearchSH];
  const int s0 = blockIdx.x * kSearchSH;
  int dx[kSearchSH], dy[kSearchSH];
#pragma unroll
  for (int q = 0; q < kSearchSH; ++q) {
    const int s = min(s0 + q, n - 1);
    dx[q] = shifts[2 * s];
    dy[q] = shifts[2 * s + 1];
  }
  const int hw = h * w;
  float acc[kSearchSH];
#pragma unroll
  for (int q = 0; q < kSearchSH; ++q) acc[q] = 0.0f;
  for (int p = threadIdx.x; p < hw; p += 256) {
    const float m0 = mask[p];
    if (m0 == 0.0f) continue;
    const int y = p / w, x = p - y * w;
    int ps[kSearchSH];
    float mm[kSearchSH];
    bool any = false;
#pragma unroll
    for (int q = 0; q < kSearchSH; ++q) {
      const int ys = y + dy[q], xs = x + dx[q];
      const bool in = ys >= 0 && ys < h && xs >= 0 && xs < w;
      ps[q] = in ? ys * w + xs : p;
      mm[q] = in ? m0 * mask[ps[q]] : 0.0f;
      any |= mm[q] != 0.0f;
    }
    if (!any) continue;
    float t[kSearchSH];
#pragma unroll
    for (int q = 0; q < kSearchSH; ++q) t[q] = 0.0f;
    for (int c = 0; c < C - 1; ++c) {
      const float* a = act + (int64_t)c * hw;
      const float b = a[p];
#pragma unroll
      for (int q = 0; q < kSearchSH; ++q) {
        const float v = a[ps[q]];
        if (edge) t[q] = fmaf(-v, b, t[q]);
        else { const float d = v - b; t[q] = fmaf(d, d, t[q]); }
      }
    }
#pragma unroll
    for (int q = 0; q < kSearchSH; ++q) acc[q] = fmaf(t[q], mm[q], acc[q]);
  }
#pragma unroll
  for (int q = 0; q < kSearchSH; ++q) {
    float v = acc[q];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][q] = v;
  }
  __syncthreads();
  if (threadIdx.x < kSearchSH && s0 + (int)threadIdx.x < n)
    losses[s0 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

}  // namespace npp

using namespace npp;

extern "C" int npp_shift_search(const float* d_act_chw, const float* d_mask_hw, int C, int h, int w, const int32_t* d_shifts_xy, int n,
                                int edge_searching, float* d_losses, void* stream) {
  if (!d_act_chw || !d_mask_hw || !d_shifts_xy || !d_losses || C < 2 || h < 1 || w < 1 || n < 1) {
    set_error("npp_shift_search: bad argument (C=%d h=%d w=%d n=%d; C counts the extra last channel)", C, h, w, n);
    return NPP_ERR_ARG;
  }
  hipLaunchKernelGGL(shift_search_kernel, dim3((unsigned)((n + kSearchSH - 1) / kSearchSH)), dim3(256), 0, (hipStream_t)stream, d_act_chw,
                     d_mask_hw, C, h, w, d_shifts_xy, n, edge_searching ? 1 : 0, d_losses);
  return check_launch("npp_shift_search");
}
