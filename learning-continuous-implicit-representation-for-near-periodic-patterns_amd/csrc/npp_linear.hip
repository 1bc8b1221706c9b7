// npp_linear.hip -- generic dense layers in exact fp32 (v_mfma_f32_32x32x2_f32): what F.linear + SnakeActivation and
// their autograd do for topologies the fused chain kernels are not specialised for.  First user: the proposal-ranking
// fits of NPP_proposal/search.py:85-205 with NPP_Net_light (models/networks.py:176-263): 2048 rows x ~0.3 M parameters x
// 300 iterations per candidate -- a launch-bound problem, so the layers stay separate launches and exact fp32.
//
// One GEMM kernel, C[m][n] = sum_k A(m,k) B(k,n) with arbitrary element strides, serves
//   forward      y  = act(x W^T + b)    A = x  (k contiguous), B = W read as (k,n) (k contiguous)
//   data grad    dx = dz W              A = dz (k contiguous), B = W read as (k=n_out, col) (col contiguous)
//   weight grad  dW = dz^T x            A = dz read as (m=n_out, k=row) (m contiguous), B = x (col contiguous)
// 64 x 64 output tile per workgroup (4 waves of 32 x 32), 32-wide k chunks staged through LDS as [k][m]: the fp32 MFMA
// takes ONE float per lane and operand, so the operands cannot come straight from global memory (measured on the
// contextual-loss kernels: texture-addresser-bound).  The staging thread map follows each operand's contiguous index.
#include "npp_common.h"

namespace npp {

struct GemmArgs {
  const float* A; int64_t sam, sak;
  const float* B; int64_t sbk, sbn;
  float* C; int64_t ldc;
  float* Z; int64_t ldz;          // optional copy of the pre-activation (forward)
  const float* bias;              // optional, per column n
  float* rowsum;                  // optional: rowsum[m] += sum_k A(m,k) (the bias gradient of the weight-gradient form)
  int M, N, K, act, accumulate;   // act: 0 none, 1 snake (x + sin^2 x)
  int kchunk;                     // k range per blockIdx.z (split-K: partial sums meet in C by atomicAdd, C pre-zeroed)
  int nbatch;                     // > 1: blockIdx.z is a batch index (no split-K); element strides between batches:
  int64_t sab, sbb, scb;
};

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm32_kernel(GemmArgs g) {
  __shared__ float sA[32][65];
  __shared__ float sB[32][65];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64, wm = wave >> 1, wn = wave & 1;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  const bool batched = g.nbatch > 1;
  if (batched) { g.A += (int64_t)blockIdx.z * g.sab; g.B += (int64_t)blockIdx.z * g.sbb; g.C += (int64_t)blockIdx.z * g.scb; }
  const int kbeg = batched ? 0 : blockIdx.z * g.kchunk, kend = batched ? g.K : min(g.K, kbeg + g.kchunk);
  const bool split = !batched && gridDim.z > 1;
  // operand chunk k0 -> registers (the loads of chunk k0 + 32 fly under the MFMAs of chunk k0)
  float ra[8], rb[8];
  auto gload = [&](int k0) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      int m, k;
      if (A_KC) { k = tid & 31; m = (tid >> 5) + 8 * r; } else { m = tid & 63; k = (tid >> 6) + 4 * r; }
      ra[r] = (m0 + m < g.M && k0 + k < kend) ? g.A[(int64_t)(m0 + m) * g.sam + (int64_t)(k0 + k) * g.sak] : 0.0f;
      int n, kb;
      if (B_KC) { kb = tid & 31; n = (tid >> 5) + 8 * r; } else { n = tid & 63; kb = (tid >> 6) + 4 * r; }
      rb[r] = (n0 + n < g.N && k0 + kb < kend) ? g.B[(int64_t)(k0 + kb) * g.sbk + (int64_t)(n0 + n) * g.sbn] : 0.0f;
    }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      int m, k;
      if (A_KC) { k = tid & 31; m = (tid >> 5) + 8 * r; } else { m = tid & 63; k = (tid >> 6) + 4 * r; }
      sA[k][m] = ra[r];
      int n, kb;
      if (B_KC) { kb = tid & 31; n = (tid >> 5) + 8 * r; } else { n = tid & 63; kb = (tid >> 6) + 4 * r; }
      sB[kb][n] = rb[r];
    }
  };
  const bool do_rowsum = g.rowsum && blockIdx.x == 0 && tid < 64;
  float rs = 0.0f;
  if (kbeg < kend) gload(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += 32) {
    sstore();
    __syncthreads();
    if (k0 + 32 < kend) gload(k0 + 32);
    if (do_rowsum) {
#pragma unroll
      for (int k = 0; k < 32; ++k) rs += sA[k][tid];
    }
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sA[2 * ks + kh][wm * 32 + l31], sB[2 * ks + kh][wn * 32 + l31], acc, 0, 0, 0);
    __syncthreads();
  }
  if (do_rowsum && m0 + tid < g.M) atomicAdd(g.rowsum + m0 + tid, rs);
  const int n = n0 + wn * 32 + l31;
  if (n >= g.N) return;
  const float bv = (g.bias && (batched || blockIdx.z == 0)) ? g.bias[n] : 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm * 32 + acc_row(r, kh);
    if (m >= g.M) continue;
    float v = acc[r] + bv;
    if (split) { atomicAdd(g.C + (int64_t)m * g.ldc + n, v); continue; }      // linear outputs only (launcher guarantees)
    if (g.Z) g.Z[(int64_t)m * g.ldz + n] = v;
    if (g.act == 1) { const float s = sinf(v); v = fmaf(s, s, v); }      // activations.py:29-35, a = 1
    else if (g.act == 2) v = fmaxf(v, 0.0f);                             // F.relu (networks.py:66-67, activation='relu')
    float* c = g.C + (int64_t)m * g.ldc + n;
    *c = g.accumulate ? *c + v : v;
  }
}

// dz = dy * act'(.) : act 1 snake from the stashed pre-activation z (1 + sin 2z); 2 sigmoid from its output y (y (1 - y));
// 3 tanh from its output (1 - y^2); 4 relu from z ([z > 0])
__global__ void act_bwd_kernel(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ zy, int64_t ldzy, int64_t B, int N,
                               int act, float* __restrict__ dz, int64_t lddz) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * N) return;
  const int64_t r = t / N;
  const int n = (int)(t - r * N);
  const float v = zy[r * ldzy + n];
  const float d = act == 1 ? 1.0f + sinf(2.0f * v) : (act == 2 ? v * (1.0f - v) : (act == 3 ? 1.0f - v * v : (act == 4 ? (v > 0.0f ? 1.0f : 0.0f) : 1.0f)));
  dz[r * lddz + n] = dy[r * lddy + n] * d;
}

// y = sigmoid(x) / tanh(x) elementwise (render's output squash, helpers.py:55-58)
__global__ void act_fwd_kernel(const float* __restrict__ x, int64_t n, int act, float* __restrict__ y) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const float v = x[t];
  y[t] = act == 2 ? 1.0f / (1.0f + expf(-v)) : (act == 3 ? tanhf(v) : v);
}

// LPIPS.forward(use_robust=False), one tap (lpips.py:99-101,110,117,130): sum over (n, pos) of
// sum_c lin_c (f0_c / (|f0| + eps) - f1_c / (|f1| + eps))^2, scaled by coef.  16 positions x 16 channel lanes per block.
__global__ __launch_bounds__(256) void lpips_plain_kernel(const float* __restrict__ f0, const float* __restrict__ f1, int N, int C, int hw,
                                                          const float* __restrict__ lin, float coef, float* __restrict__ out) {
  __shared__ float red[2][16][17];
  __shared__ float tot[4];
  const int pl = threadIdx.x & 15, cl = threadIdx.x >> 4;
  const int64_t npos = (int64_t)N * hw, ngroups = (npos + 15) / 16;
  float val = 0.0f;
  for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int64_t t = grp * 16 + pl;
    const bool live = t < npos;
    const int n = live ? (int)(t / hw) : 0, p = live ? (int)(t - (int64_t)n * hw) : 0;
    const float* a0 = f0 + (int64_t)n * C * hw + p;
    const float* a1 = f1 + (int64_t)n * C * hw + p;
    float s0 = 0.0f, s1 = 0.0f;
    if (live)
      for (int c = cl; c < C; c += 16) {
        const float u = a0[(int64_t)c * hw], v = a1[(int64_t)c * hw];
        s0 = fmaf(u, u, s0);
        s1 = fmaf(v, v, s1);
      }
    __syncthreads();
    red[0][cl][pl] = s0;
    red[1][cl][pl] = s1;
    __syncthreads();
    s0 = 0.0f; s1 = 0.0f;
#pragma unroll
    for (int q = 0; q < 16; ++q) { s0 += red[0][q][pl]; s1 += red[1][q][pl]; }
    const float i0 = 1.0f / (sqrtf(s0) + 1e-10f), i1 = 1.0f / (sqrtf(s1) + 1e-10f);
    if (live)
      for (int c = cl; c < C; c += 16) {
        const float d = a0[(int64_t)c * hw] * i0 - a1[(int64_t)c * hw] * i1;
        val = fmaf(lin[c] * d, d, val);
      }
  }
  for (int off = 32; off > 0; off >>= 1) val += __shfl_xor(val, off, 64);
  if ((threadIdx.x & 63) == 0) tot[threadIdx.x >> 6] = val;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, coef * (tot[0] + tot[1] + tot[2] + tot[3]));
}

// Per-element adaptive robust NLL (robust_loss_pytorch/adaptive.py:183-204 with num_dims = D latents): one thread per
// element index j walks the N samples -- models/style_loss.py:60-69 applies it to the N x C^2 differences of two Gram
// matrices.  loss += sum_n coef_n sum_j nll(d[n][j]); dd[n][j] = coef_n dnll/dx; dlatent[j] / [D + j] += the latent gradients.
__global__ __launch_bounds__(256) void robust_elem_kernel(const float* __restrict__ d, int N, int D, const ChanParams* __restrict__ cp,
                                                          const float* __restrict__ coef_n, float* __restrict__ loss,
                                                          float* __restrict__ dd, float* __restrict__ dlatent) {
  __shared__ float tot[4];
  const int j = blockIdx.x * 256 + threadIdx.x;
  float val = 0.0f;
  if (j < D) {
    const ChanParams P = cp[j];
    float ga = 0.0f, gc = 0.0f;
    for (int n = 0; n < N; ++n) {
      const float x = d[(int64_t)n * D + j], cf = coef_n[n];
      const float xs = x / P.c, ssx = xs * xs;
      const float uu = ssx / P.beta + 1.0f, e = 0.5f * P.alpha, lnu = logf(uu);
      const float ue = expf(e * lnu), ue1 = ue / uu;
      val = fmaf(cf, (P.beta / P.alpha) * (ue - 1.0f) + P.logc_plus_logz, val);
      if (dd) {
        dd[(int64_t)n * D + j] = cf * (x / (P.c * P.c)) * ue1;
        ga = fmaf(cf, -(2.0f / (P.alpha * P.alpha)) * (ue - 1.0f) + (P.beta / P.alpha) * ue * (0.5f * lnu + e * ssx / (P.beta * P.beta * uu)) + P.dlogz, ga);
        gc = fmaf(cf, -(x * x) / (P.c * P.c * P.c) * ue1 + 1.0f / P.c, gc);
      }
    }
    if (dlatent) {
      dlatent[j] += ga * P.dalpha_dl;
      dlatent[D + j] += gc * P.dc_dl;
    }
  }
  for (int off = 32; off > 0; off >>= 1) val += __shfl_xor(val, off, 64);
  if ((threadIdx.x & 63) == 0) tot[threadIdx.x >> 6] = val;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss, tot[0] + tot[1] + tot[2] + tot[3]);
}

__global__ void elem_chan_kernel(const float* __restrict__ latents, int D, const float* __restrict__ spline, int n_knots, float x_scale,
                                 ChanParams* __restrict__ cp) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < D) cp[c] = chan_params(latents[c], latents[D + c], spline, n_knots, x_scale);
}

// c = a - b elementwise
__global__ void sub_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n, float* __restrict__ c) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) c[t] = a[t] - b[t];
}

static int gemm_launch(GemmArgs g, bool a_kc, bool b_kc, hipStream_t s) {
  // Split the contraction when the output is small and K long (weight gradients: 256 x 256 outputs over 2048 rows would
  // be 16 workgroups looping 64 chunks each): partial sums by atomicAdd into a zeroed C.  Only for plain linear outputs
  // written densely (ldc == N), which is what the weight-gradient form produces.
  const int tiles = ((g.N + 63) / 64) * ((g.M + 63) / 64);
  if (g.nbatch > 1) {
    g.kchunk = g.K;
    const dim3 gridb((unsigned)((g.N + 63) / 64), (unsigned)((g.M + 63) / 64), (unsigned)g.nbatch);
    if (a_kc && b_kc) hipLaunchKernelGGL((gemm32_kernel<true, true>), gridb, dim3(256), 0, s, g);
    else if (a_kc) hipLaunchKernelGGL((gemm32_kernel<true, false>), gridb, dim3(256), 0, s, g);
    else if (b_kc) hipLaunchKernelGGL((gemm32_kernel<false, true>), gridb, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((gemm32_kernel<false, false>), gridb, dim3(256), 0, s, g);
    return NPP_OK;
  }
  int splits = 1;
  if (g.act == 0 && !g.Z && g.ldc == g.N && tiles < 128 && g.K >= 256) {
    splits = min(32, min((g.K + 63) / 64, (512 + tiles - 1) / tiles));
  }
  g.kchunk = ((g.K + splits - 1) / splits + 31) / 32 * 32;
  splits = (g.K + g.kchunk - 1) / g.kchunk;
  if (splits > 1 && !g.accumulate) (void)hipMemsetAsync(g.C, 0, (size_t)g.M * g.N * sizeof(float), s);
  if (g.rowsum && !g.accumulate) (void)hipMemsetAsync(g.rowsum, 0, (size_t)g.M * sizeof(float), s);
  const dim3 grid((unsigned)((g.N + 63) / 64), (unsigned)((g.M + 63) / 64), (unsigned)splits);
  if (a_kc && b_kc) hipLaunchKernelGGL((gemm32_kernel<true, true>), grid, dim3(256), 0, s, g);
  else if (a_kc) hipLaunchKernelGGL((gemm32_kernel<true, false>), grid, dim3(256), 0, s, g);
  else if (b_kc) hipLaunchKernelGGL((gemm32_kernel<false, true>), grid, dim3(256), 0, s, g);
  else hipLaunchKernelGGL((gemm32_kernel<false, false>), grid, dim3(256), 0, s, g);
  return NPP_OK;
}

}  // namespace npp

using namespace npp;

static bool lin_dims_ok(int64_t B, int in, int out) { return B >= 1 && B < (1LL << 31) && in >= 1 && out >= 1; }

extern "C" int npp_linear_fwd(const float* d_x, int64_t ldx, const float* d_w, const float* d_b, int64_t B, int in, int out, int act,
                              float* d_y, int64_t ldy, float* d_z, int64_t ldz, void* stream) {
  if (!d_x || !d_w || !d_y || !lin_dims_ok(B, in, out) || ldx < in || ldy < out || (d_z && ldz < out) || act < 0 || act > 2) {
    set_error("npp_linear_fwd: bad argument (B=%lld in=%d out=%d act=%d)", (long long)B, in, out, act);
    return NPP_ERR_ARG;
  }
  GemmArgs g{};
  g.A = d_x; g.sam = ldx; g.sak = 1;
  g.B = d_w; g.sbk = 1; g.sbn = in;                  // B(k, n) = W[n][k]
  g.C = d_y; g.ldc = ldy; g.Z = d_z; g.ldz = ldz; g.bias = d_b;
  g.M = (int)B; g.N = out; g.K = in; g.act = act; g.accumulate = 0;
  gemm_launch(g, true, true, (hipStream_t)stream);
  return check_launch("npp_linear_fwd");
}

extern "C" int npp_linear_bwd_data(const float* d_dz, int64_t lddz, const float* d_w, int64_t B, int in, int out, float* d_dx,
                                   int64_t lddx, int in_used, int accumulate, void* stream) {
  if (!d_dz || !d_w || !d_dx || !lin_dims_ok(B, in, out) || lddz < out || in_used < 1 || in_used > in || lddx < in_used) {
    set_error("npp_linear_bwd_data: bad argument (B=%lld in=%d out=%d in_used=%d)", (long long)B, in, out, in_used);
    return NPP_ERR_ARG;
  }
  GemmArgs g{};
  g.A = d_dz; g.sam = lddz; g.sak = 1;               // A(m = row, k = n_out)
  g.B = d_w; g.sbk = in; g.sbn = 1;                  // B(k = n_out, col) = W[n_out][col]
  g.C = d_dx; g.ldc = lddx;
  g.M = (int)B; g.N = in_used; g.K = out; g.act = 0; g.accumulate = accumulate;
  gemm_launch(g, true, false, (hipStream_t)stream);
  return check_launch("npp_linear_bwd_data");
}

extern "C" int npp_linear_bwd_weight(const float* d_dz, int64_t lddz, const float* d_x, int64_t ldx, int64_t B, int in, int out,
                                     float* d_dw, float* d_db, int accumulate, void* stream) {
  if (!d_dz || !d_x || !d_dw || !lin_dims_ok(B, in, out) || lddz < out || ldx < in) {
    set_error("npp_linear_bwd_weight: bad argument (B=%lld in=%d out=%d)", (long long)B, in, out);
    return NPP_ERR_ARG;
  }
  GemmArgs g{};
  g.A = d_dz; g.sam = 1; g.sak = lddz;               // A(m = n_out, k = row) = dz[row][n_out]
  g.B = d_x; g.sbk = ldx; g.sbn = 1;                 // B(k = row, col) = x[row][col]
  g.C = d_dw; g.ldc = in;
  g.M = out; g.N = in; g.K = (int)B; g.act = 0; g.accumulate = accumulate;
  g.rowsum = d_db;                                   // db[n_out] = sum_rows dz[row][n_out] = row sums of A
  gemm_launch(g, false, false, (hipStream_t)stream);
  return check_launch("npp_linear_bwd_weight");
}

extern "C" int npp_act_bwd(const float* d_dy, int64_t lddy, const float* d_zy, int64_t ldzy, int64_t B, int n, int act, float* d_dz,
                           int64_t lddz, void* stream) {
  if (!d_dy || !d_zy || !d_dz || B < 1 || n < 1 || lddy < n || ldzy < n || lddz < n || act < 0 || act > 4) {
    set_error("npp_act_bwd: bad argument");
    return NPP_ERR_ARG;
  }
  const int64_t t = B * n;
  hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_dy, lddy, d_zy, ldzy, B, n, act,
                     d_dz, lddz);
  return check_launch("npp_act_bwd");
}

extern "C" int npp_act_fwd(const float* d_x, int64_t n, int act, float* d_y, void* stream) {
  if (!d_x || !d_y || n < 1 || act < 0 || act > 3) { set_error("npp_act_fwd: bad argument"); return NPP_ERR_ARG; }
  hipLaunchKernelGGL(act_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_x, n, act, d_y);
  return check_launch("npp_act_fwd");
}

extern "C" int npp_lpips_plain_layer(const float* d_f0, const float* d_f1, int N, int C, int hw, const float* d_lin, float scale,
                                     float* d_out, void* stream) {
  if (!d_f0 || !d_f1 || !d_lin || !d_out || N < 1 || C < 16 || (C % 16) || hw < 1) {
    set_error("npp_lpips_plain_layer: bad arguments (N=%d C=%d hw=%d)", N, C, hw);
    return NPP_ERR_ARG;
  }
  const int64_t groups = ((int64_t)N * hw + 15) / 16;
  hipLaunchKernelGGL(lpips_plain_kernel, dim3((unsigned)(groups < 256 ? groups : 256)), dim3(256), 0, (hipStream_t)stream, d_f0, d_f1, N, C,
                     hw, d_lin, scale / (float)hw, d_out);
  return check_launch("npp_lpips_plain_layer");
}

/* Gram matrices G[n] = F[n] F[n]^T of (N, C, hw) features (models/style_loss.py:55-58), and the backward
 * dF[n] = (dG[n] + dG[n]^T) F[n]. */
extern "C" int npp_gram_fwd(const float* d_f, int N, int C, int hw, float* d_g, void* stream) {
  if (!d_f || !d_g || N < 1 || C < 1 || hw < 1) { set_error("npp_gram_fwd: bad argument"); return NPP_ERR_ARG; }
  GemmArgs g{};
  g.A = d_f; g.sam = hw; g.sak = 1;                  // A(m = c, k = pos)
  g.B = d_f; g.sbk = 1; g.sbn = hw;                  // B(k = pos, n = c') = F[c'][pos]
  g.C = d_g; g.ldc = C;
  g.M = C; g.N = C; g.K = hw; g.nbatch = N; g.sab = g.sbb = (int64_t)C * hw; g.scb = (int64_t)C * C;
  gemm_launch(g, true, true, (hipStream_t)stream);
  return check_launch("npp_gram_fwd");
}

extern "C" int npp_gram_bwd(const float* d_dg, const float* d_f, int N, int C, int hw, float* d_df, void* stream) {
  if (!d_dg || !d_f || !d_df || N < 1 || C < 1 || hw < 1) { set_error("npp_gram_bwd: bad argument"); return NPP_ERR_ARG; }
  GemmArgs g{};
  g.B = d_f; g.sbk = hw; g.sbn = 1;                  // B(k = c', n = pos)
  g.C = d_df; g.ldc = hw;
  g.M = C; g.N = hw; g.K = C; g.nbatch = N; g.sab = (int64_t)C * C; g.sbb = g.scb = (int64_t)C * hw;
  g.A = d_dg; g.sam = C; g.sak = 1;                  // dG F
  gemm_launch(g, true, false, (hipStream_t)stream);
  g.sam = 1; g.sak = C; g.accumulate = 1;            // + dG^T F
  if (N == 1) g.nbatch = 1;
  gemm_launch(g, false, false, (hipStream_t)stream);
  return check_launch("npp_gram_bwd");
}

/* diff = a - b, then the per-element adaptive robust NLL over (N, D) with D latent pairs [alpha(D) | scale(D)]:
 * d_loss[0] += sum_n coef_n sum_j nll ; d_ddiff (N, D) = coef_n dnll/dx ; d_dlatent [2 D] += latent gradients (both nullable
 * together).  d_coef_n: N per-sample factors (host array).  d_workspace: npp_lpips_workspace_bytes(D) bytes + N floats. */
extern "C" int npp_robust_elem(const float* d_a, const float* d_b, int N, int D, const float* d_latents, const float* d_spline,
                               int n_knots, float x_scale, const float* coef_n, float* d_loss, float* d_diff, float* d_ddiff,
                               float* d_dlatent, void* d_workspace, void* stream) {
  if (!d_a || !d_b || !d_latents || !d_spline || !coef_n || !d_loss || !d_diff || !d_workspace || N < 1 || N > 64 || D < 1 || n_knots < 2 ||
      ((d_ddiff == nullptr) != (d_dlatent == nullptr))) {
    set_error("npp_robust_elem: bad argument (N=%d D=%d; N <= 64)", N, D);
    return NPP_ERR_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  ChanParams* cp = (ChanParams*)d_workspace;
  float* d_coef = (float*)((char*)d_workspace + (size_t)D * sizeof(ChanParams));
  (void)hipMemcpyAsync(d_coef, coef_n, sizeof(float) * N, hipMemcpyHostToDevice, s);
  const int64_t n = (int64_t)N * D;
  hipLaunchKernelGGL(sub_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_a, d_b, n, d_diff);
  hipLaunchKernelGGL(elem_chan_kernel, dim3((unsigned)((D + 255) / 256)), dim3(256), 0, s, d_latents, D, d_spline, n_knots, x_scale, cp);
  hipLaunchKernelGGL(robust_elem_kernel, dim3((unsigned)((D + 255) / 256)), dim3(256), 0, s, d_diff, N, D, cp, d_coef, d_loss, d_ddiff, d_dlatent);
  return check_launch("npp_robust_elem");
}

extern "C" int64_t npp_robust_elem_workspace_bytes(int D) { return D < 1 ? NPP_ERR_ARG : (int64_t)D * sizeof(ChanParams) + 64 * sizeof(float); }
