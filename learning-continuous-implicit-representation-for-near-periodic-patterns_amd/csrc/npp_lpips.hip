// npp_lpips.hip -- K7: LPIPS head with the reference's adaptive-robust modification, forward
// and backward, one call per VGG16 tap.
//
// Replaces, from the feature tensors onward, LPIPS.forward(in0, in1, use_robust=True,
// normalize=True) (externel_lib/lpips/lpips.py:92-133): normalize_tensor (lpips/__init__.py:
// 42-44, eps 1e-10) -> per-channel AdaptiveLossFunction(num_dims=C).lossfun on the difference
// (:57-61, :103-107) -> NetLinLayer 1x1 conv with the vendored non-negative weights (:146-154,
// dropout inactive in eval) -> spatial_average (:16-17) -> sum over taps; the caller's
// torch.mean over the batch (NPP_completion/train.py:249) is folded into `scale`.
// Backward: d/d feats0 (through the channel normalisation) and d/d latents (per channel).
#include "npp_common.h"
#include "npp_trunk_layout.h"

namespace npp {

// Block = 16 positions x 16 channel lanes (thread = position pl, channels cl, cl+16, ...: Q = C/16 per thread): the deep
// taps have few positions (6x6) but 512 channels, so the channel axis must be spread over threads (one thread per
// position took 0.5 ms per call on them).  Blocks are persistent over groups of 16 positions.
// Prologue: the per-channel quantities of the adaptive loss (ChanParams: sigmoid / softplus / spline of the latents) and
// the lin weights into LDS -- no separate launch, no per-element global reads of them.
// Per group: the thread's 2 Q feature values are loaded ONCE and stay in registers through the three phases
// (channel norms -> robust NLL, its derivative and the latent gradients -> normalisation backward); the gradient is
// written once.  (First version: three global passes over the features, gradient staged in memory, parameters from a
// separate kernel: 39 us per tap on average.)
// Latent gradients: reduced over the 16 positions by shuffles into per-block LDS accumulators, one global atomic per
// channel and block at the very end (a first version issued two per channel and 16-position group: 1152 same-address
// atomics per latent on the first tap = 60 of its 70 us).
// PL positions x CL = 256 / PL channel lanes per block: 16 x 16 for the wide taps; 4 x 64 for the deep ones (512 channels
// on 12 x 12 / 6 x 6 maps: with 16 positions per block only 18 / 5 blocks exist and every thread walks 32 channels of
// ~350 VALU instructions each -- 50 us of one serial chain; 4 positions per block quarter both).
// Per-channel divisions (1/c, 1/c^2, 1/c^3, 1/beta, beta/alpha, 2/alpha^2, alpha/2 / beta^2 ...) are hoisted into the prologue.
constexpr int kLpipsMaxC = 512;
struct LpChan {
  float e, inv_c, inv_c2, x2_inv_c3, inv_beta, boa, toa2, e_inv_b2, logc_plus_logz, dlogz, dalpha_dl, dc_dl;
};
struct LpTap {
  const float* f0; const float* f1; const float* lin; const float* latents;
  float* df0; float* dlatent; unsigned long long* fix;
  int32_t hw, C, few, nb;            // few: the 4-position block shape; nb: blocks of this tap (blockIdx.x >= nb exit)
  float coef;
  // gradient straight into the trunk's flat bf16 layout (npp_trunk_layout.h) instead of fp32 (N, C, H, W): what
  // npp_trunk_grad_in(df0, NULL, ...) would make of df0 in a launch of its own (the tap gradient the backward pass adds in)
  __bf16* dflat; int64_t flat_nposp; int32_t fW, fS;      // fW = W, fS = (H + 2) (W + 2)
  const _Float16* yact;                                    // optional ReLU gate of the tapped layer itself: dflat *= [yact > 0]
};
constexpr int kLpMaxTaps = 5;
struct LpMulti {
  LpTap t[kLpMaxTaps];
  const float* spline; float* loss;
  int32_t n_taps, N, n_knots; float x_scale;
};
// LDS of one tap's block (dynamic: the taps of a grouped launch differ): red[3][CL][PL + 1] | tot[4] | sdl[2 C] | slin[C] | scp[C]
constexpr int lp_smem_bytes(int C, int PL) { return (3 * (256 / PL) * (PL + 1) + 4 + 3 * C) * 4 + C * (int)sizeof(LpChan); }

template <int Q, int PL>
__device__ __forceinline__ void lpips_layer_body(const LpTap& T, int N, const float* __restrict__ spline, int n_knots, float x_scale,
                                                 float* __restrict__ loss, char* smem, int bid, int nb) {
  constexpr int CL = 256 / PL, C = CL * Q;
  const float* __restrict__ f0 = T.f0;
  const float* __restrict__ f1 = T.f1;
  const float* __restrict__ lin = T.lin;
  const float* __restrict__ latents = T.latents;
  float* __restrict__ df0 = T.df0;
  const bool grad = T.df0 != nullptr || T.dflat != nullptr;
  float* __restrict__ dlatent = T.dlatent;
  unsigned long long* __restrict__ fix = T.fix;
  const int hw = T.hw;
  const float coef = T.coef;
  float (*red)[CL][PL + 1] = (float (*)[CL][PL + 1])smem;
  float* tot = (float*)smem + 3 * CL * (PL + 1);
  float* sdl = tot + 4;
  float* slin = sdl + 2 * C;
  LpChan* scp = (LpChan*)(slin + C);
  const int pl = threadIdx.x % PL, cl = threadIdx.x / PL;
  // latents == nullptr: LPIPS.forward(use_robust=False) (lpips.py:108-109: diffs = (feats0 - feats1)^2) with its gradient -- the
  // non-adaptive in-loop form (--use_adaptive_perceptual_loss off, train.py:241-251); no latents, no latent gradient
  const bool plain = latents == nullptr;
  for (int i = threadIdx.x; i < 2 * C; i += 256) sdl[i] = 0.0f;
  for (int c = threadIdx.x; c < C; c += 256) {
    slin[c] = lin[c];
    if (plain) continue;
    const ChanParams P = chan_params(latents[c], latents[C + c], spline, n_knots, x_scale);
    LpChan L;
    L.e = 0.5f * P.alpha;
    L.inv_c = 1.0f / P.c;
    L.inv_c2 = 1.0f / (P.c * P.c);
    L.x2_inv_c3 = 1.0f / (P.c * P.c * P.c);
    L.inv_beta = 1.0f / P.beta;
    L.boa = P.beta / P.alpha;
    L.toa2 = 2.0f / (P.alpha * P.alpha);
    L.e_inv_b2 = L.e / (P.beta * P.beta);
    L.logc_plus_logz = P.logc_plus_logz; L.dlogz = P.dlogz; L.dalpha_dl = P.dalpha_dl; L.dc_dl = P.dc_dl;
    scp[c] = L;
  }
  __syncthreads();
  const int64_t npos = (int64_t)N * hw;
  const int64_t ngroups = (npos + PL - 1) / PL;
  float val = 0.0f;
  // latent gradients: a thread owns the same Q channels in every group, so their sums over this block's groups stay in registers
  // and meet the other PL position lanes ONCE after the loop (they were 2 x log2(PL) lane exchanges + an LDS update per channel and
  // group: a quarter of the launch's vector instructions)
  float acc_a[Q], acc_c[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) { acc_a[q] = 0.0f; acc_c[q] = 0.0f; }
  for (int64_t grp = bid; grp < ngroups; grp += nb) {
    const int64_t t = grp * PL + pl;
    const bool live = t < npos;
    const int n = live ? (int)(t / hw) : 0, p = live ? (int)(t - (int64_t)n * hw) : 0;
    const float* a0 = f0 + (int64_t)n * C * hw + p;
    const float* a1 = f1 + (int64_t)n * C * hw + p;
    float* g0 = df0 ? df0 + (int64_t)n * C * hw + p : nullptr;
    __bf16* gf = nullptr;
    const _Float16* gy = nullptr;
    if (T.dflat) {
      const int yy = p / T.fW, xx = p - yy * T.fW;
      const int64_t e = (kConvGuard + (int64_t)n * T.fS + (int64_t)(yy + 1) * (T.fW + 2) + xx + 1) * 8;
      gf = T.dflat + e;
      if (T.yact) gy = T.yact + e;
    }
    float u[Q], v[Q], dd[Q];
    float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int c = cl + CL * q;
      u[q] = live ? a0[(int64_t)c * hw] : 0.0f;
      v[q] = live ? a1[(int64_t)c * hw] : 0.0f;
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      s0 = fmaf(u[q], u[q], s0);
      s1 = fmaf(v[q], v[q], s1);
    }
    __syncthreads();                                      // red[] of the previous group fully consumed
    red[0][cl][pl] = s0;
    red[1][cl][pl] = s1;
    __syncthreads();
    s0 = 0.0f; s1 = 0.0f;
#pragma unroll 16
    for (int q = 0; q < CL; ++q) { s0 += red[0][q][pl]; s1 += red[1][q][pl]; }
    const float n0 = sqrtf(s0), n1 = sqrtf(s1);
    const float i0 = 1.0f / (n0 + 1e-10f), i1 = 1.0f / (n1 + 1e-10f);
    float dot = 0.0f;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int c = cl + CL * q;
      const LpChan P = scp[c];
      const float l = slin[c];
      float da = 0.0f, dc = 0.0f;
      dd[q] = 0.0f;
      if (live && plain) {
        const float x = u[q] * i0 - v[q] * i1;
        val += l * x * x;
        if (grad) {
          const float d = l * coef * 2.0f * x;
          dd[q] = d;
          dot = fmaf(d, u[q], dot);
        }
      } else if (live) {
        const float x = u[q] * i0 - v[q] * i1;
        const float xs = x * P.inv_c, ssx = xs * xs;
        const float uu = fmaf(ssx, P.inv_beta, 1.0f), lnu = __logf(uu), inv_uu = __builtin_amdgcn_rcpf(uu);
        const float ue = __expf(P.e * lnu), ue1 = ue * inv_uu;
        val += l * (P.boa * (ue - 1.0f) + P.logc_plus_logz);
        if (grad) {
          const float d = l * coef * (x * P.inv_c2) * ue1;              // dL/d(normalised f0)_c
          dd[q] = d;
          dot = fmaf(d, u[q], dot);
          da = l * coef * (-P.toa2 * (ue - 1.0f) + P.boa * ue * (0.5f * lnu + P.e_inv_b2 * ssx * inv_uu) + P.dlogz);
          dc = l * coef * (-(x * x) * P.x2_inv_c3 * ue1 + P.inv_c);
        }
      }
      acc_a[q] += da;
      acc_c[q] += dc;
    }
    red[2][cl][pl] = dot;
    __syncthreads();
    if (grad && live) {
      dot = 0.0f;
#pragma unroll 16
      for (int q = 0; q < CL; ++q) dot += red[2][q][pl];
      const float se = n0 + 1e-10f;
      const float k = dot / (fmaxf(n0, 1e-30f) * se * se);
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const int c = cl + CL * q;
        const float g = dd[q] * i0 - u[q] * k;
        if (gf) {                                   // inverse of conv_chan(): channel c -> (chunk, element) in the stored order
          const int r = c & 15, c8 = 4 * (c >> 5) + 2 * ((c >> 4) & 1) + ((r >> 2) & 1), j = ((r >> 3) << 2) | (r & 3);
          const int64_t e = (int64_t)c8 * T.flat_nposp * 8 + j;
          gf[e] = (__bf16)((!gy || (float)gy[e] > 0.0f) ? g : 0.0f);
        }
        else g0[(int64_t)c * hw] = g;
      }
    }
  }
  if (grad && !plain) {                                                  // uniform branch
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      float da = acc_a[q], dc = acc_c[q];
#pragma unroll
      for (int off = PL / 2; off > 0; off >>= 1) {                       // the PL position lanes of this channel
        da += __shfl_xor(da, off, 64);
        dc += __shfl_xor(dc, off, 64);
      }
      if (pl == 0) {                                                     // channel c belongs to this thread alone in the block
        const int c = cl + CL * q;
        sdl[c] = da * scp[c].dalpha_dl;
        sdl[C + c] = dc * scp[c].dc_dl;
      }
    }
  }
  __syncthreads();
  for (int off = 32; off > 0; off >>= 1) val += __shfl_xor(val, off, 64);
  if ((threadIdx.x & 63) == 0) tot[threadIdx.x >> 6] = val;
  __syncthreads();
  if (fix) {
    // Order-independent sums (round 4): every block adds its partials as fixed-point integers (integer addition is
    // associative, so the arrival order of the up to 256 blocks no longer shows in the result); the last arriver converts the
    // totals back, accumulates them into dlatent / loss and clears the accumulators for the next call.
    // Scales: the loss word keeps 2^-40 (|sum| < 2^23); the latent gradients -- coef x small lin weights: 1e-9 and below per
    // channel -- are kept at 2^-52 (|sum| < 2^11: they are bounded by a few units), so that the up to 192 per-block roundings stay below
    // 1e-5 of such a value.  Out-of-range values saturate instead of wrapping.
    constexpr double kFixLoss = 1099511627776.0, kFixLat = 4503599627370496.0;          // 2^40, 2^52
    auto tofix = [](float v, double scale) {
      const double x = fmin(fmax((double)v * scale, -4.0e18), 4.0e18);
      return (unsigned long long)__double2ll_rn(x);
    };
    if (grad && !plain)
      for (int i = threadIdx.x; i < 2 * C; i += 256) atomicAdd(fix + i, tofix(sdl[i], kFixLat));
    if (threadIdx.x == 0) atomicAdd(fix + 2 * C, tofix(coef * (tot[0] + tot[1] + tot[2] + tot[3]), kFixLoss));
    if (!block_last_arriver((unsigned*)(fix + 2 * C + 1), nb)) return;
    for (int i = threadIdx.x; i <= 2 * C; i += 256) {
      const long long s = (long long)__hip_atomic_exchange(fix + i, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float v = (float)((double)s * (i == 2 * C ? 1.0 / kFixLoss : 1.0 / kFixLat));
      if (i == 2 * C) atomicAdd(loss, v);                    // (the contextual branch adds to the same word from its own stream)
      else if (grad && !plain) dlatent[i] += v;
    }
    return;
  }
  if (grad && !plain)
    for (int i = threadIdx.x; i < 2 * C; i += 256) atomicAdd(dlatent + i, sdl[i]);
  if (threadIdx.x == 0) atomicAdd(loss, coef * (tot[0] + tot[1] + tot[2] + tot[3]));
}

// One launch for up to five taps (blockIdx.y = tap): the heads of LPIPS.forward are independent of each other, and in a row they were
// 100 us (15-25 us each) of the 'same' iteration's longer chain.
// MAXQ: the largest number of channels per thread among the instantiations a variant carries.  A kernel's register count is the
// maximum over everything inlined into it: with the 32-channel case in (512 channels at 16 positions per block: 256 VGPRs + 182 AGPRs)
// EVERY tap ran at one wave per SIMD = one block per CU, and the five taps of a grouped launch queued behind each other on each CU
// (found late in round 4 with -Rpass-analysis=kernel-resource-usage).  The host picks the smallest variant that holds all taps.
#ifndef NPP_LP_MINBLOCKS
#define NPP_LP_MINBLOCKS 4         // blocks per CU the 8-channel variant is compiled for (4: 128 VGPRs with 31 spilled; 3: 152, none)
#endif
template <int MAXQ>
__global__ __launch_bounds__(256, MAXQ <= 8 ? NPP_LP_MINBLOCKS : 1) void lpips_multi_kernel(LpMulti m) {
  extern __shared__ __attribute__((aligned(16))) char lp_smem[];
  const int y = blockIdx.y;
  LpTap T = m.t[0];
#pragma unroll
  for (int q = 1; q < kLpMaxTaps; ++q)
    if (q == y) T = m.t[q];
  const int bid = blockIdx.x;
  if (bid >= T.nb) return;                                 // (whole block: no barrier is skipped)
#define NPP_LP_CASE(Q, PL) lpips_layer_body<Q, PL>(T, m.N, m.spline, m.n_knots, m.x_scale, m.loss, lp_smem, bid, T.nb)
  if (T.few) {
    switch (T.C) {
      case 64: NPP_LP_CASE(1, 4); break;
      case 128: NPP_LP_CASE(2, 4); break;
      case 192: NPP_LP_CASE(3, 4); break;
      case 256: NPP_LP_CASE(4, 4); break;
      case 384: NPP_LP_CASE(6, 4); break;
      default: NPP_LP_CASE(8, 4); break;
    }
  } else {
    switch (T.C) {
      case 16: NPP_LP_CASE(1, 16); break;
      case 32: NPP_LP_CASE(2, 16); break;
      case 64: NPP_LP_CASE(4, 16); break;
      case 128: NPP_LP_CASE(8, 16); break;
      case 192: if constexpr (MAXQ >= 16) NPP_LP_CASE(12, 16); break;
      case 256: if constexpr (MAXQ >= 16) NPP_LP_CASE(16, 16); break;
      case 384: if constexpr (MAXQ >= 32) NPP_LP_CASE(24, 16); break;
      default: if constexpr (MAXQ >= 32) NPP_LP_CASE(32, 16); break;
    }
  }
#undef NPP_LP_CASE
}

}  // namespace npp

using namespace npp;

// 2 C + 1 fixed-point accumulators + the ticket counter (8 bytes each); zeroed once by the caller, owned by one stream
extern "C" int64_t npp_lpips_workspace_bytes(int C) { return (int64_t)(2 * C + 2) * 8; }

static int lp_fill(LpTap& T, const float* f0, const float* f1, int N, int C, int hw, const float* lin, const float* latents, float scale,
                   float* df0, float* dlatent, void* ws, const char* who, void* dflat = nullptr, int N_total = 0, int H = 0, int W = 0,
                   const void* yact = nullptr) {
  if (!f0 || !f1 || !lin || N < 1 || hw < 1 || C < 16 || C > kLpipsMaxC ||
      !(C == 16 || C == 32 || C == 64 || C == 128 || C == 192 || C == 256 || C == 384 || C == 512)) {
    set_error("%s: bad arguments (N=%d C=%d hw=%d; C one of 16, 32, 64, 128, 192, 256, 384, 512)", who, N, C, hw);
    return NPP_ERR_ARG;
  }
  // latents == NULL: the plain head (use_robust=False), with its gradient when df0 is given
  const bool grad = df0 || dflat;
  if (latents ? !grad != (dlatent == nullptr) : dlatent != nullptr) {
    set_error("%s: df0 and dlatent go together (no dlatent for the plain head)", who);
    return NPP_ERR_ARG;
  }
  if (dflat && (df0 || H < 1 || W < 1 || (int64_t)H * W != hw || N_total < N || C % 16 || W > kConvGuard - 3)) {
    set_error("%s: the flat gradient takes the place of df0 and needs the tap's geometry (N_total=%d >= N=%d, H=%d x W=%d = hw=%d, C %% 16)",
              who, N_total, N, H, W, hw);
    return NPP_ERR_ARG;
  }
  const int64_t nh = (int64_t)N * hw;
  // deep taps: 4 positions x 64 channel lanes per block.  (2048: VGG16's relu3_3 of two 96^2 patches -- 1152 positions x 256
  // channels -- takes 288 four-position groups of 4 channels per thread instead of 72 sixteen-position groups of 16: every tap of the
  // loop then fits the 8-channel variant of the kernel, four blocks per CU.)
  const bool few = nh <= 2048 && (C % 64) == 0;
  const int PLr = few ? 4 : 16;
  const int64_t groups = (nh + PLr - 1) / PLr;
  T.f0 = f0; T.f1 = f1; T.lin = lin; T.latents = latents; T.df0 = df0; T.dlatent = dlatent; T.fix = (unsigned long long*)ws;
  // Timing-only builds (tools/r4_lp_heads_probe.py, at one block per CU as the kernel then ran): without the tail (atomics + last arriver) 38.1 us of
  // 47.3, without the per-channel prologue 42.7, neither 33.1; a tap alone 15-20 us.
  // (at four blocks per CU: cap 128 / 160 / 192 / 224 / 256 = 30.6 / 29.7 / 30.2 / 30.0 / 32.1 us for the five taps alone; 384 .. 1152: 39.6 .. 56.5 -- every
  //  block ends in 2 C + 1 same-address atomics per tap)
  constexpr int nb_cap = 192;
  const int64_t per = (groups + nb_cap - 1) / nb_cap;      // groups per block; then the fewest blocks that need no more than that
  T.hw = hw; T.C = C; T.few = few ? 1 : 0; T.nb = (int)((groups + per - 1) / per);
  T.coef = scale / (float)nh;                             // spatial mean and batch mean folded with the caller's weight
  if (yact && !dflat) { set_error("%s: yact gates the flat gradient (dflat)", who); return NPP_ERR_ARG; }
  T.dflat = (__bf16*)dflat; T.yact = (const _Float16*)yact;
  if (dflat) { T.flat_nposp = conv_nposp(N_total, H, W); T.fW = W; T.fS = (H + 2) * (W + 2); }
  return NPP_OK;
}
static int lp_launch(LpMulti& m, void* stream, const char* who) {
  int nb = 0, smem = 0;
  for (int i = 0; i < m.n_taps; ++i) {
    nb = m.t[i].nb > nb ? m.t[i].nb : nb;
    const int b = lp_smem_bytes(m.t[i].C, m.t[i].few ? 4 : 16);
    smem = b > smem ? b : smem;
  }
  int maxq = 8;                                             // channels per thread of the widest tap (few: C / 64, else C / 16)
  for (int i = 0; i < m.n_taps; ++i) {
    const int q = m.t[i].few ? m.t[i].C / 64 : m.t[i].C / 16;
    maxq = q > maxq ? q : maxq;
  }
#define NPP_LP_GO(MQ)                                                                                                                  \
  do {                                                                                                                                 \
    static SmemOnce once;                                                                                                              \
    if (!smem_attr(once, (const void*)lpips_multi_kernel<MQ>, lp_smem_bytes(kLpipsMaxC, 4))) { set_error("%s: smem attribute", who); return NPP_ERR_LAUNCH; } \
    hipLaunchKernelGGL(lpips_multi_kernel<MQ>, dim3((unsigned)nb, (unsigned)m.n_taps), dim3(256), (size_t)smem, (hipStream_t)stream, m); \
  } while (0)
  if (maxq <= 8) NPP_LP_GO(8);
  else if (maxq <= 16) NPP_LP_GO(16);
  else NPP_LP_GO(32);
#undef NPP_LP_GO
  return check_launch(who);
}

extern "C" int npp_lpips_layer(const float* d_f0, const float* d_f1, int N, int C, int hw, const float* d_lin,
                               const float* d_latents, const float* d_spline, int n_knots, float x_scale, float scale,
                               float* d_loss, float* d_df0, float* d_dlatent, void* d_workspace, void* stream) {
  if (!d_loss || (d_latents && (!d_spline || n_knots < 2))) { set_error("npp_lpips_layer: bad arguments (loss / spline)"); return NPP_ERR_ARG; }
  LpMulti m{};
  int rc = lp_fill(m.t[0], d_f0, d_f1, N, C, hw, d_lin, d_latents, scale, d_df0, d_dlatent, d_workspace, "npp_lpips_layer");
  if (rc) return rc;
  m.spline = d_spline; m.loss = d_loss; m.n_taps = 1; m.N = N; m.n_knots = n_knots; m.x_scale = x_scale;
  return lp_launch(m, stream, "npp_lpips_layer");
}

// All taps of LPIPS.forward in ONE launch (lpips.py:99-133: the five heads are independent; `val` is their sum): taps[i] as the
// arguments of npp_lpips_layer for tap i; every tap needs a workspace of its OWN when workspaces are used (they run side by side).
extern "C" int npp_lpips_layers(int n_taps, const npp_lpips_tap* taps, int N, const float* d_spline, int n_knots, float x_scale, float scale,
                                float* d_loss, void* stream) {
  if (n_taps < 1 || n_taps > kLpMaxTaps || !taps || !d_loss) { set_error("npp_lpips_layers: n_taps=%d (1..%d)", n_taps, kLpMaxTaps); return NPP_ERR_ARG; }
  LpMulti m{};
  for (int i = 0; i < n_taps; ++i) {
    const npp_lpips_tap& t = taps[i];
    if (t.latents && (!d_spline || n_knots < 2)) { set_error("npp_lpips_layers: spline"); return NPP_ERR_ARG; }
    for (int j = 0; j < i; ++j)
      if (t.workspace && t.workspace == taps[j].workspace) { set_error("npp_lpips_layers: taps %d and %d share a workspace", j, i); return NPP_ERR_ARG; }
    int rc = lp_fill(m.t[i], t.f0, t.f1, N, t.C, t.hw, t.lin, t.latents, scale, t.df0, t.dlatent, t.workspace, "npp_lpips_layers",
                     t.dflat, t.N_total, t.H, t.W, t.yact);
    if (rc) return rc;
  }
  m.spline = d_spline; m.loss = d_loss; m.n_taps = n_taps; m.N = N; m.n_knots = n_knots; m.x_scale = x_scale;
  return lp_launch(m, stream, "npp_lpips_layers");
}
