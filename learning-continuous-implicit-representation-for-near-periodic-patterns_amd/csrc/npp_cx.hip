// npp_cx.hip -- K5 integer patch gather and K6 contextual-loss core (forward + d/dx).
//
// K5 replaces extract_glimpse(mode='nearest', padding 'zeros') as models/sampler.py:171-178,
// 284-291 calls it (utils/extract_glimpse.py:53-79): an integer crop [c - P/2, c + P/2) with
// zeros outside the image; the reference first tiles the whole image once per crop.
//
// K6 replaces contextual_loss(x, y, band_width, weight, 'cosine')
// (externel_lib/contextual_loss/functional.py:9-63, :127-163) and its autograd backward w.r.t.
// x, starting from the feature tensors (N, C, h, w) fp32 (NCHW, as the VGG trunk returns them):
//   mu_c = mean_{n,h,w} y ; xh = normalize_C(x - mu) ; yh = normalize_C(y - mu)
//   raw = xh^T yh (N, I, J) ; D = 1 - clamp(raw, 0, 1) ; Dt = D / (min_j D + 1e-5)
//   w = exp((1 - Dt)/h) ; cx = w / sum_j w ; cxn = mean_j max_i cx ; loss = mean_n -log(cxn + 1e-5)
// The similarity needs fp32 (D/(min D + 1e-5) amplifies operand rounding by up to 1e5), so the two
// matrix products run on v_mfma_f32_32x32x2_f32 (exact fp32, 157 TFLOP/s peak).  The (I x J) matrices
// are written once (D) and once more (cx, overwritten in place by d raw in the backward); the
// reference materialises 4-5 of them plus autograd copies.
//
// Backward (closed form, derivation in DESIGN.md section 7 / oracle cx_backward): the loss sees cx
// only at the per-column arg-max rows i*(j):  G_ij = g_n / J [i == i*(j)],  A_i = sum_j G_ij cx_ij,
//   dw = (G - A_i)/s_i ; dDt = -dw w / h ; dD = dDt/(m_i + 1e-5) - [j == argmin_j D_ij] sum_j dDt D/(m_i+1e-5)^2
//   draw = -dD on 0 < raw < 1 ; dxh = draw yh^T ; dx = (dxh - xh (xh . dxh)) / |x - mu|.
#include "npp_common.h"

namespace npp {

// ---------------------------------------------------------------- K5 patch gather
__global__ void patch_gather_kernel(const float* __restrict__ img, const float* __restrict__ mask, int H, int W,
                                    const int32_t* __restrict__ centres, int M, int P, float* __restrict__ out_rgb,
                                    float* __restrict__ out_mask) {
  const int m = blockIdx.y;
  const int cy = centres[2 * m], cx = centres[2 * m + 1];
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < P * P; idx += gridDim.x * blockDim.x) {
    const int py = idx / P, px = idx - py * P;
    const int y = cy - P / 2 + py, x = cx - P / 2 + px;
    const bool in = y >= 0 && y < H && x >= 0 && x < W;
    const int64_t src = (int64_t)y * W + x;
#pragma unroll
    for (int c = 0; c < 3; ++c) out_rgb[((int64_t)m * 3 + c) * P * P + idx] = in ? img[src * 3 + c] : 0.0f;
    if (out_mask) out_mask[(int64_t)m * P * P + idx] = in ? mask[src] : 0.0f;
  }
}

// ---------------------------------------------------------------- K6 contextual loss
struct CxWs {           // workspace carve (floats unless noted)
  float* mu;            // [C]
  float* inx;           // [N*hw]  1 / max(|x - mu|, 1e-12)
  float* iny;           // [N*hw]
  unsigned* dmin;       // [N*hw]  row min of D as float bits (D >= 0: unsigned order == float order)
  float* s;             // [N*hw]  row sum of w
  unsigned* cmax;       // [N*hw]  column max of cx as float bits
  float* g;             // [N]     dL/dcxn / J
  float* D;             // [N*hw*hw]
  float* cx;            // [N*hw*hw]  cx, then d raw
};

__host__ __device__ inline int64_t cx_ws_floats(int N, int C, int hw) {
  return (int64_t)C + 5LL * N * hw + N + 2LL * N * hw * hw + 64;
}

__host__ inline CxWs carve(float* base, int N, int C, int hw) {
  CxWs w;
  float* p = base;
  w.mu = p; p += (C + 15) / 16 * 16;
  const int64_t nh = (int64_t)N * hw;
  w.inx = p; p += nh;
  w.iny = p; p += nh;
  w.dmin = (unsigned*)p; p += nh;
  w.s = p; p += nh;
  w.cmax = (unsigned*)p; p += nh;
  w.g = p; p += (N + 15) / 16 * 16;
  w.D = p; p += nh * hw;
  w.cx = p;
  return w;
}

// mu_c = mean over (n, pos) of y   (functional.py:141) ; one workgroup per channel
__global__ void cx_mean_kernel(const float* __restrict__ y, int N, int C, int hw, float* __restrict__ mu) {
  __shared__ float red[4];
  const int c = blockIdx.x;
  float acc = 0.0f;
  for (int n = 0; n < N; ++n)
    for (int p = threadIdx.x; p < hw; p += blockDim.x) acc += y[((int64_t)n * C + c) * hw + p];
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) mu[c] = (red[0] + red[1] + red[2] + red[3]) / (float)((int64_t)N * hw);
}

// inverse L2 norms over channels of (x - mu), (y - mu) per position (F.normalize eps 1e-12);
// also initialises the row-min / column-max cells.
__global__ void cx_norm_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ mu,
                               int N, int C, int hw, CxWs w) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)N * hw) return;
  const int n = (int)(t / hw), p = (int)(t - (int64_t)n * hw);
  float sx = 0.0f, sy = 0.0f;
  for (int c = 0; c < C; ++c) {
    const float m = mu[c];
    const float a = x[((int64_t)n * C + c) * hw + p] - m, b = y[((int64_t)n * C + c) * hw + p] - m;
    sx = fmaf(a, a, sx);
    sy = fmaf(b, b, sy);
  }
  w.inx[t] = 1.0f / fmaxf(sqrtf(sx), 1e-12f);
  w.iny[t] = 1.0f / fmaxf(sqrtf(sy), 1e-12f);
  w.dmin[t] = 0x7f800000u;   // +inf
  w.cmax[t] = 0u;
}

// D = 1 - clamp(inx_i iny_j sum_c (x_ci - mu_c)(y_cj - mu_c), 0, 1), row minima by atomicMin.
// One wave per 64x64 output tile (2x2 MFMA tiles); operands read straight from the NCHW
// tensors: for a fixed channel the 32 positions of a fragment are contiguous (128-B segments).
__global__ __launch_bounds__(256) void cx_sim_kernel(const float* __restrict__ x, const float* __restrict__ y, int N,
                                                     int C, int hw, CxWs w) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tiles = (hw + 63) / 64;
  const int64_t wid = (int64_t)blockIdx.x * 4 + wave;
  if (wid >= (int64_t)N * tiles * tiles) return;
  const int n = (int)(wid / (tiles * tiles));
  const int tij = (int)(wid - (int64_t)n * tiles * tiles);
  const int i0 = (tij / tiles) * 64, j0 = (tij % tiles) * 64;
  const int l31 = lane & 31, kh = lane >> 5;
  const float* xn = x + (int64_t)n * C * hw;
  const float* yn = y + (int64_t)n * C * hw;
  int ia[2], jb[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    ia[t] = min(i0 + 32 * t + l31, hw - 1);     // clamp: out-of-range rows/cols are discarded at the store
    jb[t] = min(j0 + 32 * t + l31, hw - 1);
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
#pragma unroll 4
  for (int c0 = 0; c0 < C; c0 += 2) {
    const int c = c0 + kh;
    const float m = w.mu[c];
    float av[2], bv[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      av[t] = xn[(int64_t)c * hw + ia[t]] - m;
      bv[t] = yn[(int64_t)c * hw + jb[t]] - m;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
  }
  // accumulator: column (lane & 31) = j, register r = row acc_row(r, lane >> 5) = i
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int j = j0 + 32 * b + l31;
    const float sj = j < hw ? w.iny[(int64_t)n * hw + j] : 0.0f;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = i0 + 32 * a + acc_row(r, kh);
        if (i < hw && j < hw) {
          const float raw = acc[a][b][r] * w.inx[(int64_t)n * hw + i] * sj;
          const float d = 1.0f - fminf(fmaxf(raw, 0.0f), 1.0f);
          w.D[((int64_t)n * hw + i) * hw + j] = d;
          atomicMin(&w.dmin[(int64_t)n * hw + i], __float_as_uint(d));
        }
      }
    }
  }
}

// one wave per row (n, i): w = exp((1 - D/(dmin + 1e-5))/h), s = sum_j w, cx = w/s, column max.
__global__ __launch_bounds__(256) void cx_rows_fwd_kernel(int N, int hw, float inv_h, CxWs w) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;
  if (row >= (int64_t)N * hw) return;
  const int n = (int)(row / hw);
  const float dm = __uint_as_float(w.dmin[row]) + 1e-5f;
  const float* Dr = w.D + row * hw;
  float* cr = w.cx + row * hw;
  float s = 0.0f;
  for (int j = lane; j < hw; j += 64) {
    const float wv = __expf((1.0f - Dr[j] / dm) * inv_h);
    cr[j] = wv;
    s += wv;
  }
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (lane == 0) w.s[row] = s;
  const float inv = 1.0f / s;
  for (int j = lane; j < hw; j += 64) {
    const float c = cr[j] * inv;
    cr[j] = c;
    atomicMax(&w.cmax[(int64_t)n * hw + j], __float_as_uint(c));
  }
}

// per sample: cxn = mean_j cmax ; loss += scale * (-log(cxn [* weight] + 1e-5)) [/ N] ; g = dL/dcxn / J
__global__ void cx_loss_kernel(int N, int hw, const float* __restrict__ weight, float scale, float* __restrict__ loss,
                               CxWs w) {
  __shared__ float red[4];
  const int n = blockIdx.x;
  float acc = 0.0f;
  for (int j = threadIdx.x; j < hw; j += blockDim.x) acc += __uint_as_float(w.cmax[(int64_t)n * hw + j]);
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float cxn = (red[0] + red[1] + red[2] + red[3]) / (float)hw;
    float l, dcxn;
    if (weight) {                       // functional.py:55-57: sum(-log(cx * w + 1e-5))
      const float wt = weight[n];
      l = -logf(cxn * wt + 1e-5f);
      dcxn = -wt / (cxn * wt + 1e-5f);
    } else {                            // mean over samples
      l = -logf(cxn + 1e-5f) / (float)N;
      dcxn = -1.0f / ((float)N * (cxn + 1e-5f));
    }
    atomicAdd(loss, scale * l);
    w.g[n] = scale * dcxn / (float)hw;
  }
}

// one wave per row: cx row -> d raw row (in place).
__global__ __launch_bounds__(256) void cx_rows_bwd_kernel(int N, int hw, float inv_h, CxWs w) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;
  if (row >= (int64_t)N * hw) return;
  const int n = (int)(row / hw);
  const float dmv = __uint_as_float(w.dmin[row]);
  const float dm = dmv + 1e-5f;
  const float s = w.s[row], g = w.g[n];
  const float* Dr = w.D + row * hw;
  float* cr = w.cx + row * hw;
  const unsigned* cm = w.cmax + (int64_t)n * hw;
  // A_i = sum_j G_ij cx_ij over the columns whose maximum sits in this row
  float A = 0.0f;
  for (int j = lane; j < hw; j += 64) {
    const float c = cr[j];
    if (__float_as_uint(c) == cm[j]) A += g * c;
  }
  for (int off = 32; off > 0; off >>= 1) A += __shfl_xor(A, off, 64);
  float corr = 0.0f;
  int jmin = 0x7fffffff;
  for (int j = lane; j < hw; j += 64) {
    const float c = cr[j], d = Dr[j];
    const float G = (__float_as_uint(c) == cm[j]) ? g : 0.0f;
    const float wv = c * s;
    const float dDt = -((G - A) / s) * wv * inv_h;
    corr += dDt * d;
    const float dD = dDt / dm;
    cr[j] = (d > 0.0f && d < 1.0f) ? -dD : 0.0f;       // clamp(raw, 0, 1) passes gradient inside (0,1)
    if (d == dmv) jmin = min(jmin, j);
  }
  for (int off = 32; off > 0; off >>= 1) {
    corr += __shfl_xor(corr, off, 64);
    jmin = min(jmin, __shfl_xor(jmin, off, 64));
  }
  if (jmin < hw && lane == (jmin & 63)) {     // the lane that stored cr[jmin]: same-lane program order
    const float d = Dr[jmin];
    if (d > 0.0f && d < 1.0f) cr[jmin] += corr / (dm * dm);   // -(-corr/dm^2): the min's own gradient path
  }
}

// dxh[c][i] = sum_j draw[i][j] * (y[c][j] - mu_c) * iny[j]  -> written as [n][c][i] scratch (= out, then
// finished in place by cx_dx_finish).  Output tile per wave: 32 channels x 64 positions; both
// operands are contiguous along j, so each lane reads 8 consecutive j (two 16-byte loads) for its
// row and the 8 products go through 8 MFMAs whose two k-slots are the lane halves.
__global__ __launch_bounds__(256) void cx_dx_kernel(const float* __restrict__ y, int N, int C, int hw, CxWs w,
                                                    float* __restrict__ dxh) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int itiles = (hw + 63) / 64, ctiles = C / 32;
  const int64_t wid = (int64_t)blockIdx.x * 4 + wave;
  if (wid >= (int64_t)N * ctiles * itiles) return;
  const int n = (int)(wid / (ctiles * itiles));
  const int rem = (int)(wid - (int64_t)n * ctiles * itiles);
  const int c0 = (rem / itiles) * 32, i0 = (rem % itiles) * 64;
  const int l31 = lane & 31, kh = lane >> 5;
  const int c = c0 + l31;
  const float m = w.mu[c];
  const float* yr = y + ((int64_t)n * C + c) * hw;
  const float* iny = w.iny + (int64_t)n * hw;
  const float* dr[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) dr[t] = w.cx + ((int64_t)n * hw + min(i0 + 32 * t + l31, hw - 1)) * hw;
  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
  for (int j0 = 0; j0 < hw; j0 += 16) {
    const int jb = j0 + 8 * kh;                      // this lane half's 8 columns
    float a[8], b0[8], b1[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int j = jb + q;
      const bool ok = j < hw;
      a[q] = ok ? (yr[j] - m) * iny[j] : 0.0f;       // A[m = channel][k = j]
      b0[q] = ok ? dr[0][j] : 0.0f;                  // B[k = j][n = position]
      b1[q] = ok ? dr[1][j] : 0.0f;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b0[q], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b1[q], acc[1], 0, 0, 0);
    }
  }
  // accumulator: column = position (lane & 31), rows = channels
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int i = i0 + 32 * t + l31;
    if (i < hw) {
#pragma unroll
      for (int r = 0; r < 16; ++r) dxh[((int64_t)n * C + c0 + acc_row(r, kh)) * hw + i] = acc[t][r];
    }
  }
}

// dx = (dxh - xh (xh . dxh)) * inx, xh = (x - mu) inx ; thread per (n, position), in place on dxh
__global__ void cx_dx_finish_kernel(const float* __restrict__ x, int N, int C, int hw, CxWs w, float* __restrict__ dx) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)N * hw) return;
  const int n = (int)(t / hw), p = (int)(t - (int64_t)n * hw);
  const float inx = w.inx[t];
  float dot = 0.0f;
  for (int c = 0; c < C; ++c) {
    const int64_t o = ((int64_t)n * C + c) * hw + p;
    dot = fmaf((x[o] - w.mu[c]) * inx, dx[o], dot);
  }
  for (int c = 0; c < C; ++c) {
    const int64_t o = ((int64_t)n * C + c) * hw + p;
    dx[o] = (dx[o] - (x[o] - w.mu[c]) * inx * dot) * inx;
  }
}

}  // namespace npp

using namespace npp;

extern "C" int npp_patch_gather(const float* d_img_hwc, const float* d_mask_hw, int H, int W, const int32_t* d_centres_yx,
                                int M, int P, float* d_out_rgb, float* d_out_mask, void* stream) {
  if (!d_img_hwc || !d_centres_yx || !d_out_rgb || H < 1 || W < 1 || M < 0 || P < 2 || (P & 1)) {
    set_error("npp_patch_gather: bad arguments (M=%d, P=%d must be even)", M, P);
    return NPP_ERR_ARG;
  }
  if (d_out_mask && !d_mask_hw) { set_error("npp_patch_gather: out_mask without mask"); return NPP_ERR_ARG; }
  if (M == 0) return NPP_OK;
  const int bx = (P * P + 255) / 256;
  hipLaunchKernelGGL(patch_gather_kernel, dim3(bx, M), dim3(256), 0, (hipStream_t)stream, d_img_hwc, d_mask_hw, H, W,
                     d_centres_yx, M, P, d_out_rgb, d_out_mask);
  return check_launch("npp_patch_gather");
}

extern "C" int64_t npp_cx_workspace_bytes(int N, int C, int hw) {
  if (N < 1 || C < 32 || (C % 32) || hw < 1) { set_error("npp_cx_workspace_bytes: need N>=1, C multiple of 32, hw>=1"); return -1; }
  return 4 * cx_ws_floats(N, C, hw);
}

extern "C" int npp_cx_fwd_bwd(const float* d_fx, const float* d_fy, int N, int C, int hw, float band_width,
                              const float* d_weight, float scale, float* d_loss, float* d_dfx, void* d_workspace,
                              int64_t workspace_bytes, void* stream) {
  if (!d_fx || !d_fy || !d_loss || !d_workspace || N < 1 || C < 32 || (C % 32) || hw < 1 || !(band_width > 0.0f)) {
    set_error("npp_cx_fwd_bwd: bad arguments (N=%d C=%d hw=%d)", N, C, hw);
    return NPP_ERR_ARG;
  }
  if (workspace_bytes < 4 * cx_ws_floats(N, C, hw)) { set_error("npp_cx_fwd_bwd: workspace too small"); return NPP_ERR_ARG; }
  hipStream_t s = (hipStream_t)stream;
  const CxWs w = carve((float*)d_workspace, N, C, hw);
  const int64_t nh = (int64_t)N * hw;
  const float inv_h = 1.0f / band_width;
  hipLaunchKernelGGL(cx_mean_kernel, dim3(C), dim3(256), 0, s, d_fy, N, C, hw, w.mu);
  hipLaunchKernelGGL(cx_norm_kernel, dim3((unsigned)((nh + 255) / 256)), dim3(256), 0, s, d_fx, d_fy, w.mu, N, C, hw, w);
  const int tiles = (hw + 63) / 64;
  hipLaunchKernelGGL(cx_sim_kernel, dim3((unsigned)(((int64_t)N * tiles * tiles + 3) / 4)), dim3(256), 0, s, d_fx, d_fy, N,
                     C, hw, w);
  hipLaunchKernelGGL(cx_rows_fwd_kernel, dim3((unsigned)((nh + 3) / 4)), dim3(256), 0, s, N, hw, inv_h, w);
  hipLaunchKernelGGL(cx_loss_kernel, dim3(N), dim3(256), 0, s, N, hw, d_weight, scale, d_loss, w);
  if (d_dfx) {
    hipLaunchKernelGGL(cx_rows_bwd_kernel, dim3((unsigned)((nh + 3) / 4)), dim3(256), 0, s, N, hw, inv_h, w);
    const int64_t waves = (int64_t)N * (C / 32) * tiles;
    hipLaunchKernelGGL(cx_dx_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, d_fy, N, C, hw, w, d_dfx);
    hipLaunchKernelGGL(cx_dx_finish_kernel, dim3((unsigned)((nh + 255) / 256)), dim3(256), 0, s, d_fx, N, C, hw, w, d_dfx);
  }
  return check_launch("npp_cx_fwd_bwd");
}
