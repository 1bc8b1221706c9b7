// npp_light16.hip -- NPP_Net_light's training chains on the 16-bit matrix pipe (VERDICT r3 "f1 in 16-bit"):
// the proposal-ranking fits of NPP_proposal/search.py:85-147 (one NPP_Net_light per candidate, models/networks.py:176-263 with
// len(freq_scales) == 1, D = 4, W = 256, snake) with bf16 operands, fp32 accumulation and fp32 master weights -- the numeric
// contract of the main loop's coordinate MLP (npp_mlp_fwd.hip / npp_mlp_bwd.hip), whose building blocks these kernels share
// (npp_common.h: weight-stream ring, mma_ring, fragment regions in LDS, W-format stashes).  npp_light.hip is the exact-fp32
// form of the same chains (v_mfma_f32_32x32x2_f32, 157 TFLOP/s peak); this one runs v_mfma_f32_32x32x16_bf16.
//
//   forward   x_per(20) -> 4 x [256, snake] -> feature_linear1 (256, linear) -> [f1 | x_pos(42)] -> pos_linears.0 (128, snake)
//             -> rgb_linear (3) -> sigmoid                                                         ONE launch, all candidates
//   backward  pixel loss -> d raw -> d z_p -> d f1 -> d z_3 .. d z_0 (the data gradients)          ONE launch
//   weight gradients: the main loop's grouped split-K launch (npp_mlp_wgrad.hip) over a job table for this network,
//   candidate = image of a stacked launch;  Adam + re-pack of both 16-bit packs: ONE launch
//
// A 256-thread workgroup owns 64 pixel rows of one candidate (blockIdx.y); GEMMs transposed (Z^T = W X^T), weights = A operand
// streamed from the candidate's bf16 pack through a 4-k-step register ring, activations = B operand as 16-byte fragments in LDS
// (an accumulator tile converted to bf16 IS two fragments of the next layer).  Stashes: W-format arrays (npp_layout.h): fp16
// pre-activations z of the snake layers (the backward derives 1 + sin 2z, the weight-gradient launch snake(z)), bf16 for the
// linear ones.
#include <stdlib.h>

#include "npp_common.h"
#include "npp_light_layout.h"

namespace npp {

static_assert(kW == 256 && kNB == 2 && kNT == 8, "the 16-bit light chains are built for W = 256, 64-row workgroups");

constexpr int kL16Threads = 256;
constexpr int kL16Region = kKSAct * kNB * 1024;           // 32 KiB: 256 features x 64 rows of bf16 fragments
constexpr int kL16RegionX = 4 * kNB * 1024;               // x_pos: 4 k-steps
constexpr int kL16SmemF = 2 * kL16Region + kL16RegionX;
constexpr int kL16SmemB = 2 * kL16Region + kRowTile * 3 * 4;

// packs of one candidate, 16-byte units: forward [layer][k-step][neuron tile][lane], then the transposed packs of the backward chain
enum { HF_L0 = 0, HF_L1, HF_L2, HF_L3, HF_F1, HF_POS, HF_N };
enum { HB_POS = 0, HB_F1, HB_L3, HB_L2, HB_L1, HB_N };
struct L16Pack {
  int32_t f_off[HF_N], f_ks[HF_N], f_nt[HF_N];
  int32_t b_off[HB_N], b_ks[HB_N];
  int32_t f_total, total;
};
__host__ __device__ inline L16Pack l16_pack_desc() {
  L16Pack d{};
  int off = 0;
  const int ks[HF_N] = {2, kKSAct, kKSAct, kKSAct, kKSAct, kL16KsHp}, nt[HF_N] = {kNT, kNT, kNT, kNT, kNT, kNT / 2};
  for (int l = 0; l < HF_N; ++l) { d.f_off[l] = off; d.f_ks[l] = ks[l]; d.f_nt[l] = nt[l]; off += ks[l] * nt[l] * 64; }
  d.f_total = off;
  const int bk[HB_N] = {kLPosOut / 16, kKSAct, kKSAct, kKSAct, kKSAct};
  for (int l = 0; l < HB_N; ++l) { d.b_off[l] = off; d.b_ks[l] = bk[l]; off += bk[l] * kNT * 64; }
  d.total = off;
  return d;
}
// npp_light_desc index of a pack entry: periodic 0..3, pos (4), feature1 (5), rgb (6)
__host__ __device__ inline int l16_fwd_layer(int l) { return l < 4 ? l : (l == HF_F1 ? 5 : 4); }
__host__ __device__ inline int l16_bwd_layer(int l) { return l == HB_POS ? 4 : (l == HB_F1 ? 5 : (l == HB_L3 ? 3 : (l == HB_L2 ? 2 : 1))); }

struct L16Args {
  npp_light_desc L;
  const float* params; int64_t params_stride;       // (C, params_stride) fp32 master weights
  const bf16x8* pack; int64_t pack_stride16;         // 16-byte units per candidate
  const float* x_per; const float* x_pos; const int64_t* idx; int64_t n_src;
  char* actF; int64_t act_stride;                    // forward stash, bytes per candidate
  char* dzF; int64_t dz_stride;                      // gradient stash
  float* pred;                                       // (C, B, 3)
  const float* dpred;                                // (C, B, 3), backward without the folded loss
  int64_t B;
  const float* gt; const float* latents; const float* spline; int n_knots; float x_scale;
  float* loss; float* dlatent;
  // (round 6) bit-reproducible form: every block leaves its seven loss / latent-gradient sums in part[(c n_wg + wg) * 8 + k] (no atomics;
  // npp_light16_adam_pack_det adds them in block order).  "multi" forms (candidate = one IMAGE's fit): elements per candidate of the
  // positional table, the row indices and the targets; 0 = shared
  float* part;
  int64_t x_pos_cs, idx_cs, gt_cs;
};

// ---- packs ----------------------------------------------------------------------------------------------------------------
// forward unit (k-step ks, tile nt, lane (m, hh)) element j = W[32 nt + m][16 ks + perm16(hh, j)]; backward unit (ks, tile t, lane)
// element j = W[16 ks + perm16(hh, j)][32 t + m]; zero outside the matrix
__global__ void light16_pack_kernel(L16Args a, L16Pack pd, bf16x8* __restrict__ out, int64_t out_stride16) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= pd.total) return;
  const float* P = a.params + (int64_t)blockIdx.y * a.params_stride;
  const bool bwd = u >= pd.f_total;
  int l = 0;
  if (!bwd) { for (int q = 1; q < HF_N; ++q) if (u >= pd.f_off[q]) l = q; }
  else { for (int q = 1; q < HB_N; ++q) if (u >= pd.b_off[q]) l = q; }
  const int r = u - (bwd ? pd.b_off[l] : pd.f_off[l]);
  const int nt_n = bwd ? kNT : pd.f_nt[l];
  const int lane = r & 63, nt = (r >> 6) % nt_n, ks = (r >> 6) / nt_n;
  const int m = nt * 32 + (lane & 31), hh = lane >> 5;
  const int li = bwd ? l16_bwd_layer(l) : l16_fwd_layer(l);
  const float* Wm = P + a.L.w_off[li];
  const int ld = a.L.ld[li], n_out = a.L.n_out[li], n_in = a.L.n_in[li];
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = 16 * ks + perm16(hh, j);
    float v = 0.0f;
    if (!bwd) { if (m < n_out && k < n_in) v = Wm[(int64_t)m * ld + k]; }
    else if (k < n_out && m < kLW) v = Wm[(int64_t)k * ld + m];
    o[j] = (__bf16)v;
  }
  out[(int64_t)blockIdx.y * out_stride16 + u] = o;
}

// ---- optimizer.step() + the packs of the next forward / backward in one launch ----------------------------------------------------
// Adam over the stacked fp32 blobs (adam_update, npp_common.h); the gradient of a parameter is the sum of the weight-gradient
// launch's split-K slabs in slab order (plain stores there, no atomics: bit-reproducible); every updated weight is scattered as bf16
// into both packs through the inverse of light16_pack_kernel's map.  The extra block column steps the candidate's six
// adaptive-loss latents and clears one loss word.
struct L16AdamArgs {
  npp_light_desc L;
  float *p, *m, *v; int64_t stride; int32_t n;
  const float* gslabs; int32_t n_slabs; int64_t slab_stride, slab_cand_stride;
  __bf16* pack; int64_t pack_stride16;
  float *lat, *lat_m, *lat_v, *dlat, *zero;
  float step_size, b1, b2, inv_sqrt_bc2, eps;
  const float* part; int32_t n_part; float* loss_cur;   // npp_light16_adam_pack_det: the blocks' sums of npp_light16_bwd_det, added in block order
};
__global__ __launch_bounds__(256) void light16_adam_pack_kernel(L16AdamArgs a, L16Pack pd) {
  const int c = blockIdx.y;
  if (blockIdx.x == gridDim.x - 1) {
    const int t = threadIdx.x;
    if (t < 6) {
      const int i = c * 6 + t;
      float g = a.dlat[i];
      if (a.part)
        for (int b = 0; b < a.n_part; ++b) g += a.part[((int64_t)c * a.n_part + b) * 8 + 1 + t];
      float m = a.lat_m[i], v = a.lat_v[i];
      a.lat[i] = adam_update(a.lat[i], m, v, g, a.step_size, a.b1, a.b2, a.inv_sqrt_bc2, a.eps);
      a.lat_m[i] = m; a.lat_v[i] = v; a.dlat[i] = 0.0f;
    } else if (t == 6 && a.zero) a.zero[c] = 0.0f;
    else if (t == 64 && a.part && a.loss_cur) {
      float l = 0.0f;
      for (int b = 0; b < a.n_part; ++b) l += a.part[((int64_t)c * a.n_part + b) * 8];
      a.loss_cur[c] += l;
    }
    return;
  }
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  const int64_t gi = (int64_t)c * a.stride + i;
  const float* gs = a.gslabs + (int64_t)c * a.slab_cand_stride + i;
  float g = gs[0];
  for (int s = 1; s < a.n_slabs; ++s) g += gs[(int64_t)s * a.slab_stride];
  float m = a.m[gi], v = a.v[gi];
  const float w = adam_update(a.p[gi], m, v, g, a.step_size, a.b1, a.b2, a.inv_sqrt_bc2, a.eps);
  a.p[gi] = w; a.m[gi] = m; a.v[gi] = v;
  int li = -1;
#pragma unroll
  for (int q = 0; q < 6; ++q)          // (biases and rgb_linear are read from the blob by the chains: nothing to scatter)
    if (i >= a.L.w_off[q] && i < a.L.w_off[q] + (int64_t)a.L.n_out[q] * a.L.ld[q]) li = q;
  if (li < 0) return;
  const int off = i - (int)a.L.w_off[li], ld = a.L.ld[li];
  const int row = off / ld, col = off - row * ld;
  if (col >= a.L.n_in[li]) return;                        // pad column of the stored matrix
  __bf16* pk = a.pack + (int64_t)c * a.pack_stride16 * 8;
  const __bf16 wb = (__bf16)w;
  const int lf = li < 4 ? HF_L0 + li : (li == 4 ? HF_POS : HF_F1);
  {
    const int c16 = col & 15;
    const int64_t unit = pd.f_off[lf] + ((int64_t)(col >> 4) * pd.f_nt[lf] + (row >> 5)) * 64 + (row & 31) + 32 * unperm_hh(c16);
    pk[unit * 8 + unperm_j(c16)] = wb;
  }
  if (li >= 1 && col < kLW) {                               // transposed pack: A[m = col][k = row]
    const int lb = li == 4 ? HB_POS : (li == 5 ? HB_F1 : (li == 3 ? HB_L3 : (li == 2 ? HB_L2 : HB_L1)));
    const int c16 = row & 15;
    const int64_t unit = pd.b_off[lb] + ((int64_t)(row >> 4) * kNT + (col >> 5)) * 64 + (col & 31) + 32 * unperm_hh(c16);
    pk[unit * 8 + unperm_j(c16)] = wb;
  }
}

// (Measured and dropped, round 4: a form with 16-byte pack stores -- a wave owns a 32-neuron x 16-column block = the 64 fragments of one
// forward k-step and, through a 1-KiB LDS transpose, 64 whole fragments of the transposed pack -- took 41 us for 9 candidates against
// 27 us for this element-wise kernel: its loads are 64-byte pieces of 32 different rows per instruction, and that costs more than the
// 2-byte scattered stores it removes.)

// ---- shared pieces ----------------------------------------------------------------------------------------------------------------
template <int NTW>
__device__ __forceinline__ void l16_bias(f32x16 (&acc)[NTW][kNB], const float* __restrict__ bias, int nt0, const Lane& L) {
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
    f32x16 bv;
#pragma unroll
    for (int r = 0; r < 16; ++r) bv[r] = bias[(nt0 + nt) * 32 + acc_row(r, L.h)];
#pragma unroll
    for (int bt = 0; bt < kNB; ++bt) acc[nt][bt] = bv;
  }
}
// forward epilogue: SNAKE: z -> fp16 stash, a = snake(z) -> fragments of the next layer; linear: the bf16 fragments are the stash.
// acc keeps the fp32 activation (pos_linears.0 -> rgb_linear reads it back)
template <bool SNAKE, int NTW>
__device__ __forceinline__ void l16_epi(f32x16 (&acc)[NTW][kNB], char* out, int nt0, char* arr, int arr_nks, int wg, const Lane& L) {
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
    const int ntg = nt0 + nt;
#pragma unroll
    for (int bt = 0; bt < kNB; ++bt) {
      if (SNAKE) {
#pragma unroll
        for (int s = 0; s < 2; ++s) stash_store(arr + wfmt_unit(arr_nks, wg, 2 * ntg + s, bt, L.b, L.h), pack_acc_f16(acc[nt][bt], s));
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][bt][r] = snake_fast(acc[nt][bt][r]);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 f = pack_acc(acc[nt][bt], s);
        if (out) lds_store_frag(out, 2 * ntg + s, bt, L.lane, f);
        if (!SNAKE) stash_store(arr + wfmt_unit(arr_nks, wg, 2 * ntg + s, bt, L.b, L.h), f);
      }
    }
  }
}
// one input fragment built from a row-major fp32 table: slot j of unit (k-step ks, lane half hh) = column 16 ks + perm16(hh, j)
__device__ __forceinline__ bf16x8 l16_in_frag(const float* __restrict__ row, int ks, int hh, int ncol) {
  bf16x8 f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int col = 16 * ks + perm16(hh, j);
    f[j] = (__bf16)(col < ncol ? row[col] : 0.0f);
  }
  return f;
}

// Work item (candidate, 64-row tile) of a workgroup.  Workgroups go round-robin over the 8 XCDs by linear id (observed, speed only); the
// items are numbered so that every XCD owns a CONTIGUOUS range of them in candidate-major order: an XCD then streams ~C / 8 candidates'
// packs (1.2 MB each) through its 4-MiB L2 instead of all C of them (11 MB at C = 9: every k-step of every layer a miss).  The grid
// is C * n_wg rounded up to a multiple of 8; surplus workgroups exit before any barrier.
__device__ __forceinline__ bool l16_item(int n_wg, int C, int& c, int& wg) {
  const int total = n_wg * C, per = (total + 7) >> 3;
  const int l = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
  if (((int)blockIdx.x >> 3) >= per || l >= total) return false;
  c = l / n_wg;
  wg = l - c * n_wg;
  return true;
}
// The weight-stream ring of npp_common.h (WRing / mma_ring) with the depth as a parameter.  What bounds these chains, measured with
// timing-only builds on 9 candidates x 2048 rows (288 workgroups; forward / backward launch, us):  as shipped 31.6 / 40.3;  no stash
// stores (NPP_DIAG_NOSTASH) 27.6 / 27.8;  no MFMAs (NPP_DIAG_L16_NOMFMA) 27.4 / 38.2;  neither 23.2 / 26.6;  ring depth 8 instead of
// 4, candidate-contiguous XCD numbering, bias prefetch: no change each.  I.e. the skeleton -- every workgroup streams its candidate's
// whole pack (0.7 MB forward) through one CU's L2 -> register path (~70 GB/s per CU: ~10 us), the 32 CUs that hold two of the 288
// workgroups take twice that -- not latency, not the matrix pipe (4 us) and, in the forward, not the stores; the backward's 53 MB of
// dz stores cost 12 us.  128-row workgroups (144 of them: one per CU, half the weight bytes per row) were then built (kernels templated on
// the batch tiles per workgroup, tests green) and measured: forward 32.5 -> 39.6 us, backward 40.9 -> 41.4 -- with one wave per SIMD a
// workgroup's MFMA + epilogue time per row adds to its weight stream instead of hiding under it.  Not kept.
#ifndef NPP_LIGHT16_RING
#define NPP_LIGHT16_RING 4
#endif
constexpr int kLRD = NPP_LIGHT16_RING;
template <int NTW> struct LRing { bf16x8 w[kLRD][NTW]; wrsrc_t rsrc; };
template <int NTW, int NT>
__device__ __forceinline__ void lslot_load(LRing<NTW>& r, int slot, wptr_t wp, int ks, int nt0, int lane) {
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
    const u32x4_t raw = __builtin_amdgcn_raw_buffer_load_b128(r.rsrc, lane * 16, (int)((wp + (uint32_t)((ks * NT + nt0 + nt) * 64)) * 16u), 0);
    r.w[slot][nt] = __builtin_bit_cast(bf16x8, raw);
  }
}
template <int NTW, int NT>
__device__ __forceinline__ void lring_fill(LRing<NTW>& r, wptr_t wp, int nt0, int lane) {
#pragma unroll
  for (int q = 0; q < kLRD; ++q) lslot_load<NTW, NT>(r, q, wp, q, nt0, lane);
}
// schedule positions [0, KSTOT) of a part with KSREAL real k-steps (KSTOT a multiple of the depth, so that every part starts at slot 0);
// after the MFMAs of position ks its slot is refilled with this part's k-step ks + depth or, past the part's end, the next part's
template <int KSREAL, int KSTOT, int NTW, int NT>
__device__ __forceinline__ void lmma(f32x16 (&acc)[NTW][kNB], const char* region, wptr_t wp, wptr_t next_wp, int nt0, const Lane& L, LRing<NTW>& ring) {
  static_assert(KSTOT % kLRD == 0 && KSREAL <= KSTOT, "ring schedule");
  bf16x8 xn[kNB];
#pragma unroll
  for (int bt = 0; bt < kNB; ++bt) xn[bt] = lds_frag(region, 0, bt, L.lane);
#pragma unroll
  for (int ks = 0; ks < KSTOT; ++ks) {
    const int slot = ks % kLRD;
    if (ks < KSREAL) {
      bf16x8 x[kNB];
#pragma unroll
      for (int bt = 0; bt < kNB; ++bt) x[bt] = xn[bt];
      if (ks + 1 < KSREAL) {
#pragma unroll
        for (int bt = 0; bt < kNB; ++bt) xn[bt] = lds_frag(region, ks + 1, bt, L.lane);
      }
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int bt = 0; bt < kNB; ++bt) {
          acc[nt][bt] = mfma_bf16(ring.w[slot][nt], x[bt], acc[nt][bt]);
        }
    }
    if (ks + kLRD < KSREAL) lslot_load<NTW, NT>(ring, slot, wp, ks + kLRD, nt0, L.lane);
    else if (ks + kLRD >= KSTOT && next_wp != kNoW) lslot_load<NTW, NT>(ring, slot, next_wp, ks + kLRD - KSTOT, nt0, L.lane);
    asm volatile("" ::: "memory");   // pin the refill here: no hoisting of later loads
  }
}
constexpr int l16_tot(int ks) { return (ks + kLRD - 1) / kLRD * kLRD; }

template <int NTW> struct L16Bias { float v[NTW][16]; };
template <int NTW>
__device__ __forceinline__ void l16_bias_fetch(L16Bias<NTW>& bp, const float* __restrict__ bias, int nt0, const Lane& L) {
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) bp.v[nt][r] = bias[(nt0 + nt) * 32 + acc_row(r, L.h)];
}
template <int NTW>
__device__ __forceinline__ void l16_bias_apply(f32x16 (&acc)[NTW][kNB], const L16Bias<NTW>& bp) {
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
    for (int bt = 0; bt < kNB; ++bt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][bt][r] = bp.v[nt][r];
}

// ---- forward ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kL16Threads, 2) void light16_fwd_kernel(L16Args a, L16Pack pd, int n_wg, int C) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* R0 = smem;
  char* R1 = smem + kL16Region;
  char* RX = smem + 2 * kL16Region;
  Lane L;
  L.tid = threadIdx.x;
  L.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  L.lane = threadIdx.x & 63;
  L.b = L.lane & 31;
  L.h = L.lane >> 5;
  L.n_wg = n_wg; L.xslot = 0; L.xcount = 1;
  int c, wg;
  if (!l16_item(n_wg, C, c, wg)) return;
  const int64_t B = a.B, row0 = (int64_t)wg * kRowTile;
  const float* P = a.params + (int64_t)c * a.params_stride;
  char* actF = a.actF + (int64_t)c * a.act_stride;
  const int nt0 = 2 * L.wave;
  auto arr = [&](int ks_off) -> char* { return actF + wfmt_array_base(ks_off, L.n_wg); };

  LRing<2> ring;
  LRing<1> ringp;
  ring.rsrc = ringp.rsrc = make_wrsrc(a.pack + (int64_t)c * a.pack_stride16, pd.total);
  lring_fill<2, kNT>(ring, (wptr_t)pd.f_off[HF_L0], nt0, L.lane);

  // inputs -> fragments: x_per (2 k-steps, region R1) and x_pos (4 k-steps, region RX), both also into their stash arrays
  {
    const int ks = L.tid >> 7, bt = (L.tid >> 6) & 1;
    const int64_t r = row0 + bt * 32 + L.b, src = a.idx ? a.idx[(int64_t)c * a.idx_cs + r] : r;
    const bf16x8 f = l16_in_frag(a.x_per + ((int64_t)c * a.n_src + src) * kLPer, ks, L.h, kLPer);
    lds_store_frag(R1, ks, bt, L.lane, f);
    stash_store(arr(L16A_XP) + wfmt_unit(2, wg, ks, bt, L.b, L.h), f);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int ksl = 2 * q + ks;
      const bf16x8 g = l16_in_frag(a.x_pos + (int64_t)c * a.x_pos_cs + src * kLPos, ksl, L.h, kLPos);
      lds_store_frag(RX, ksl, bt, L.lane, g);
      stash_store(arr(L16A_HP) + wfmt_unit(kL16KsHp, wg, kKSAct + ksl, bt, L.b, L.h), g);
    }
  }
  wg_barrier();

  f32x16 acc[2][kNB];
  constexpr int A = kKSAct;
  // (the NEXT layer's bias values are fetched before the current layer's epilogue: their L2 latency hides under it and the barrier)
  L16Bias<2> bn;
  L16Bias<1> bnp;
  // periodic_linears.0: 20 (32 slots) -> 256, snake: R1 -> R0
  l16_bias<2>(acc, P + a.L.b_off[0], nt0, L);
  lmma<2, l16_tot(2), 2, kNT>(acc, R1, (wptr_t)pd.f_off[HF_L0], (wptr_t)pd.f_off[HF_L1], nt0, L, ring);
  l16_bias_fetch<2>(bn, P + a.L.b_off[1], nt0, L);
  l16_epi<true, 2>(acc, R0, nt0, arr(L16A_Z0), A, wg, L);
  wg_barrier();
  // periodic_linears.1 .. 3: R0 -> R1 -> R0 -> R1
#pragma unroll
  for (int l = 1; l <= 3; ++l) {
    char* in = (l & 1) ? R0 : R1;
    char* out = (l & 1) ? R1 : R0;
    l16_bias_apply<2>(acc, bn);
    lmma<A, l16_tot(A), 2, kNT>(acc, in, (wptr_t)pd.f_off[HF_L0 + l], (wptr_t)pd.f_off[HF_L0 + l + 1], nt0, L, ring);
    l16_bias_fetch<2>(bn, P + a.L.b_off[l == 3 ? 5 : l + 1], nt0, L);
    l16_epi<true, 2>(acc, out, nt0, arr(L16A_Z0 + A * l), A, wg, L);
    wg_barrier();
  }
  // feature_linear1 (linear): R1 -> R0, also the first 16 k-steps of the [f1 | x_pos] stash
  l16_bias_apply<2>(acc, bn);
  lmma<A, l16_tot(A), 2, kNT>(acc, R1, (wptr_t)pd.f_off[HF_F1], kNoW, nt0, L, ring);
  lring_fill<1, kNT / 2>(ringp, (wptr_t)pd.f_off[HF_POS], L.wave, L.lane);
  l16_bias_fetch<1>(bnp, P + a.L.b_off[4], L.wave, L);
  l16_epi<false, 2>(acc, R0, nt0, arr(L16A_HP), kL16KsHp, wg, L);
  wg_barrier();
  // pos_linears.0: [f1 (R0) | x_pos (RX)] -> 128, snake; one neuron tile per wave
  f32x16 accp[1][kNB];
  constexpr wptr_t UP = (kNT / 2) * 64;
  l16_bias_apply<1>(accp, bnp);
  lmma<A, l16_tot(A), 1, kNT / 2>(accp, R0, (wptr_t)pd.f_off[HF_POS], (wptr_t)pd.f_off[HF_POS] + A * UP, L.wave, L, ringp);
  lmma<4, l16_tot(4), 1, kNT / 2>(accp, RX, (wptr_t)pd.f_off[HF_POS] + A * UP, kNoW, L.wave, L, ringp);
  l16_epi<true, 1>(accp, nullptr, L.wave, arr(L16A_ZP), kLPosOut / 16, wg, L);
  // rgb_linear 128 -> 3 + sigmoid (models/helpers.py:55-56): per-lane partial dot over its 16 neurons, lane halves by shuffle,
  // the four waves through LDS
  wg_barrier();                                  // every wave is done with R0 / RX
  float* sRGB = (float*)R0;                      // [4 waves][64 rows][3]
  {
    const float* Wr = P + a.L.w_off[6];
    const int ldr = a.L.ld[6];
    float part[kNB][3];
#pragma unroll
    for (int bt = 0; bt < kNB; ++bt)
#pragma unroll
      for (int q = 0; q < 3; ++q) part[bt][q] = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = L.wave * 32 + acc_row(r, L.h);
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const float w = Wr[q * ldr + k];
#pragma unroll
        for (int bt = 0; bt < kNB; ++bt) part[bt][q] = fmaf(w, accp[0][bt][r], part[bt][q]);
      }
    }
#pragma unroll
    for (int bt = 0; bt < kNB; ++bt)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const float v = part[bt][q] + __shfl_xor(part[bt][q], 32, 64);
        if (L.h == 0) sRGB[(L.wave * kRowTile + bt * 32 + L.b) * 3 + q] = v;
      }
  }
  wg_barrier();
  if (L.tid < kRowTile * 3) {
    const int row = L.tid / 3, q = L.tid - row * 3;
    float z = P[a.L.b_off[6] + q];
#pragma unroll
    for (int w = 0; w < 4; ++w) z += sRGB[(w * kRowTile + row) * 3 + q];
    a.pred[((int64_t)c * B + row0 + row) * 3 + q] = 1.0f / (1.0f + expf(-z));
  }
}

// ---- backward (data gradients) ------------------------------------------------------------------------------------------------
struct L16ZPre { f16x8 z[2][kNB][2]; };
__device__ __forceinline__ void l16_zfetch(L16ZPre& zp, const char* z_array, int wg, int kt0, const Lane& L) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int bt = 0; bt < kNB; ++bt)
#pragma unroll
      for (int s = 0; s < 2; ++s) zp.z[t][bt][s] = *(const f16x8*)(z_array + wfmt_unit(kKSAct, wg, 2 * (kt0 + t) + s, bt, L.b, L.h));
}
template <bool DERIV>
__device__ __forceinline__ void l16_bepi(f32x16 (&acc)[2][kNB], char* out, const L16ZPre* zp, char* dz_array, int wg, int kt0, const Lane& L) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int ntg = kt0 + t;
#pragma unroll
    for (int bt = 0; bt < kNB; ++bt) {
      f32x16 g = acc[t][bt];
      if (DERIV) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const f16x8 zf = zp->z[t][bt][s];
#pragma unroll
          for (int j = 0; j < 8; ++j) g[8 * s + j] *= 1.0f + __builtin_amdgcn_sinf((float)zf[j] * (2.0f * kInv2Pi));     // activations.py:29-35: 1 + sin 2z
        }
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 f = pack_acc(g, s);
        if (out) lds_store_frag(out, 2 * ntg + s, bt, L.lane, f);
        dz_store(dz_array + wfmt_unit(kKSAct, wg, 2 * ntg + s, bt, L.b, L.h), f);
      }
    }
  }
}
__device__ __forceinline__ void l16_zero(f32x16 (&acc)[2][kNB]) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int bt = 0; bt < kNB; ++bt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][bt][r] = 0.0f;
}

__global__ __launch_bounds__(kL16Threads, 2) void light16_bwd_kernel(L16Args a, L16Pack pd, int n_wg, int C) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* R0 = smem;
  char* R1 = smem + kL16Region;
  float* sD = (float*)(smem + 2 * kL16Region);       // d raw [64 rows][3]
  Lane L;
  L.tid = threadIdx.x;
  L.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  L.lane = threadIdx.x & 63;
  L.b = L.lane & 31;
  L.h = L.lane >> 5;
  L.n_wg = n_wg; L.xslot = 0; L.xcount = 1;
  const int tid = L.tid;
  int c, wg;
  if (!l16_item(n_wg, C, c, wg)) return;
  const int64_t B = a.B, row0 = (int64_t)wg * kRowTile;
  const float* P = a.params + (int64_t)c * a.params_stride;
  const char* actF = a.actF + (int64_t)c * a.act_stride;
  char* dzF = a.dzF + (int64_t)c * a.dz_stride;
  const int kt0 = 2 * L.wave;
  auto zs = [&](int ks_off) { return actF + wfmt_array_base(ks_off, L.n_wg); };
  auto dzr = [&](int ks_off) { return dzF + wfmt_array_base(ks_off, L.n_wg); };

  // everything the prologue needs from memory first: the weight ring of the first data-gradient part, rgb_linear's rows, z_p
  LRing<2> ring;
  ring.rsrc = make_wrsrc(a.pack + (int64_t)c * a.pack_stride16, pd.total);
  lring_fill<2, kNT>(ring, (wptr_t)pd.b_off[HB_POS], kt0, L.lane);
  f16x8 zp_pre[kNB][2];
#pragma unroll
  for (int bt = 0; bt < kNB; ++bt)
#pragma unroll
    for (int s = 0; s < 2; ++s) zp_pre[bt][s] = *(const f16x8*)(zs(L16A_ZP) + wfmt_unit(kLPosOut / 16, wg, 2 * L.wave + s, bt, L.b, L.h));
  float wr[3][16];
  {
    const float* Wr = P + a.L.w_off[6];
    const int ldr = a.L.ld[6];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) wr[q][r] = Wr[q * ldr + L.wave * 32 + acc_row(r, L.h)];
  }

  // d raw = d pred * pred (1 - pred); with the pixel loss folded in (a.gt): d pred = d img2mse(robust_loss_adaptive)/d pred right here
  // (models/mse_calculator.py:13-27 without a mask: the arithmetic of pixel_loss_body, npp_common.h), loss / latent gradients by atomics
  __shared__ ChanParams cp[3];
  __shared__ float sred[7];
  __shared__ float swv[kL16Threads / 64][7];
  if (a.gt) {
    if (tid < 3) cp[tid] = chan_params(a.latents[c * 6 + tid], a.latents[c * 6 + 3 + tid], a.spline, a.n_knots, a.x_scale);
    if (tid < 7) sred[tid] = 0.0f;
    wg_barrier();
  }
  float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f;                 // this thread's loss term and latent-gradient terms (channel tid % 3)
  if (tid < kRowTile * 3) {
    const int64_t g = ((int64_t)c * B + row0) * 3 + tid;
    const float p = a.pred[g];
    float dp;
    if (a.gt) {
      const int ch = tid % 3;
      const ChanParams q = cp[ch];
      const float inv = 1.0f / (3.0f * (float)B);
      const float x = p - a.gt[(int64_t)c * a.gt_cs + row0 * 3 + tid];
      const float xs = x / q.c, ssx = xs * xs;
      const float u = ssx / q.beta + 1.0f, e = 0.5f * q.alpha, lnu = logf(u);
      const float ue = expf(e * lnu), ue1 = ue / u;
      dp = inv * (x / (q.c * q.c)) * ue1;
      t0 = (q.beta / q.alpha) * (ue - 1.0f) + q.logc_plus_logz;
      t1 = -(2.0f / (q.alpha * q.alpha)) * (ue - 1.0f) + (q.beta / q.alpha) * ue * (0.5f * lnu + e * ssx / (q.beta * q.beta * u)) + q.dlogz;
      t2 = -(x * x) / (q.c * q.c * q.c) * ue1 + 1.0f / q.c;
      if (!a.part) {
        atomicAdd(&sred[0], t0);
        atomicAdd(&sred[1 + ch], t1);
        atomicAdd(&sred[4 + ch], t2);
      }
    } else {
      dp = a.dpred[g];
    }
    sD[tid] = dp * p * (1.0f - p);
  }
  if (a.gt && a.part) {
    // deterministic form (every wave, whole: threads past the 3 x 64 values carry zeros): the seven sums of a wave by shuffle butterflies
    // -- a fixed tree -- then the waves' results in wave order (csrc/npp_light.hip light_bwd_kernel does the same)
    const int ch = tid % 3, lane_ = tid & 63, wave_ = tid >> 6;
    const float v7[7] = {t0, ch == 0 ? t1 : 0.0f, ch == 1 ? t1 : 0.0f, ch == 2 ? t1 : 0.0f, ch == 0 ? t2 : 0.0f, ch == 1 ? t2 : 0.0f, ch == 2 ? t2 : 0.0f};
#pragma unroll
    for (int k7 = 0; k7 < 7; ++k7) {
      float v = v7[k7];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
      if (lane_ == 0) swv[wave_][k7] = v;
    }
  }
  wg_barrier();
  if (a.gt && tid < 7) {
    const float inv = 1.0f / (3.0f * (float)B);
    if (a.part) {
      float v = 0.0f;
#pragma unroll
      for (int w_ = 0; w_ < kL16Threads / 64; ++w_) v += swv[w_][tid];
      a.part[((int64_t)c * n_wg + wg) * 8 + tid] = tid == 0 ? v * inv : (tid < 4 ? inv * v * cp[tid - 1].dalpha_dl : inv * v * cp[tid - 4].dc_dl);
    } else {
      const float v = sred[tid];
      if (tid == 0) atomicAdd(a.loss + c, v * inv);
      else if (tid < 4) atomicAdd(a.dlatent + c * 6 + (tid - 1), inv * v * cp[tid - 1].dalpha_dl);
      else atomicAdd(a.dlatent + c * 6 + 3 + (tid - 4), inv * v * cp[tid - 4].dc_dl);
    }
  }
  // d raw as a 2-k-step W-format array (rgb_linear's weight gradient): features 0..2 real, the rest zero
  {
    const int q1 = tid >> 7, bt = (tid >> 6) & 1, row = bt * 32 + L.b;
    bf16x8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (__bf16)0.0f;
    if (q1 == 0 && L.h == 0) {
      f[0] = (__bf16)sD[row * 3 + 0];
      f[1] = (__bf16)sD[row * 3 + 1];
      f[2] = (__bf16)sD[row * 3 + 2];
    }
    dz_store(dzr(L16D_RAW) + wfmt_unit(2, wg, q1, bt, L.b, L.h), f);
  }
  // d a_p = d raw W_rgb, d z_p = d a_p * snake'(z_p): wave w owns pos_linears.0's neuron tile w -> fragments in R0 + the stash
#pragma unroll
  for (int bt = 0; bt < kNB; ++bt) {
    const int row = bt * 32 + L.b;
    const float g0 = sD[row * 3 + 0], g1 = sD[row * 3 + 1], g2 = sD[row * 3 + 2];
    f32x16 g;
#pragma unroll
    for (int r = 0; r < 16; ++r) g[r] = wr[0][r] * g0 + wr[1][r] * g1 + wr[2][r] * g2;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const f16x8 zf = zp_pre[bt][s];
#pragma unroll
      for (int j = 0; j < 8; ++j) g[8 * s + j] *= 1.0f + __builtin_amdgcn_sinf((float)zf[j] * (2.0f * kInv2Pi));
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const bf16x8 f = pack_acc(g, s);
      lds_store_frag(R0, 2 * L.wave + s, bt, L.lane, f);
      dz_store(dzr(L16D_ZP) + wfmt_unit(kLPosOut / 16, wg, 2 * L.wave + s, bt, L.b, L.h), f);
    }
  }
  wg_barrier();

  f32x16 acc[2][kNB];
  L16ZPre zpre;
  constexpr int KP = kLPosOut / 16, A = kKSAct;
  // d f1 = W_pos[:, :256]^T d z_p  (feature_linear1 is linear: this IS its d z; x_pos gets no gradient): R0 -> R1
  l16_zero(acc);
  lmma<KP, l16_tot(KP), 2, kNT>(acc, R0, (wptr_t)pd.b_off[HB_POS], (wptr_t)pd.b_off[HB_F1], kt0, L, ring);
  l16_bepi<false>(acc, R1, nullptr, dzr(L16D_F1), wg, kt0, L);
  wg_barrier();
  // d z_3 = (W_f1^T d f1) * snake'(z_3), d z_2 = (W_3^T d z_3) * snake'(z_2), ..., d z_0: R1 -> R0 -> R1 -> R0
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int l = 3 - j;                           // hidden layer whose d z this step produces
    char* in = (j & 1) ? R0 : R1;
    char* out = (j & 1) ? R1 : R0;
    l16_zero(acc);
    l16_zfetch(zpre, zs(L16A_Z0 + A * l), wg, kt0, L);
    lmma<A, l16_tot(A), 2, kNT>(acc, in, (wptr_t)pd.b_off[HB_F1 + j], j == 3 ? kNoW : (wptr_t)pd.b_off[HB_F1 + j + 1], kt0, L, ring);
    l16_bepi<true>(acc, j == 3 ? nullptr : out, &zpre, dzr(L16D_Z0 + A * l), wg, kt0, L);
    if (j != 3) wg_barrier();
  }
}

}  // namespace npp

using namespace npp;

static int l16_check(const npp_light_desc* L, const void* p0, const void* p1, int C, int64_t B, const char* who) {
  if (!L || !p0 || !p1 || C < 1 || C > 65535 || B < kRowTile || B % kRowTile || B / kRowTile > 65535 || (B / kRowTile) * C > 0x3fffffff) {
    set_error("%s: bad argument (C=%d B=%lld; B a positive multiple of %d)", who, C, (long long)B, kRowTile);
    return NPP_ERR_ARG;
  }
  const int n_out[7] = {kLW, kLW, kLW, kLW, kLPosOut, kLW, 3}, n_in[7] = {kLPer, kLW, kLW, kLW, kLW + kLPos, kLW, kLPosOut};
  for (int i = 0; i < 7; ++i)
    if (L->n_out[i] != n_out[i] || L->n_in[i] != n_in[i] || L->ld[i] < n_in[i] || L->w_off[i] < 0 || L->b_off[i] < 0) {
      set_error("%s: layer %d is %d x %d (ld %d): this build fuses NPP_Net_light(D=4, W=256) with 20 / 42 input columns only", who, i,
                L->n_out[i], L->n_in[i], L->ld[i]);
      return NPP_ERR_UNSUPPORTED;
    }
  return NPP_OK;
}

extern "C" int64_t npp_light16_pack_bytes(void) { return 16 * (int64_t)l16_pack_desc().total; }
extern "C" int64_t npp_light16_stash_bytes(int64_t B, int which) {
  if (B < kRowTile || B % kRowTile || which < 0 || which > 1) return NPP_ERR_ARG;
  return wfmt_array_base(which ? L16D_TOTAL : L16A_TOTAL, B / kRowTile);
}

extern "C" int npp_light16_pack(const npp_light_desc* L, const float* d_params, int64_t params_stride, int C, void* d_pack,
                                int64_t pack_stride_bytes, void* stream) {
  int rc = l16_check(L, d_params, d_pack, C, kRowTile, "npp_light16_pack");
  if (rc) return rc;
  const L16Pack pd = l16_pack_desc();
  if (pack_stride_bytes < 16 * (int64_t)pd.total || pack_stride_bytes % 16) { set_error("npp_light16_pack: pack stride %lld", (long long)pack_stride_bytes); return NPP_ERR_ARG; }
  L16Args a{};
  a.L = *L; a.params = d_params; a.params_stride = params_stride;
  hipLaunchKernelGGL(light16_pack_kernel, dim3((unsigned)((pd.total + 255) / 256), (unsigned)C), dim3(256), 0, (hipStream_t)stream, a, pd,
                     (bf16x8*)d_pack, pack_stride_bytes / 16);
  return check_launch("npp_light16_pack");
}

static int l16_fwd_go(const npp_light_desc* L, const float* d_params, int64_t params_stride, const void* d_pack,
                      int64_t pack_stride_bytes, const float* d_x_per, const float* d_x_pos, const int64_t* d_idx, int64_t n_src,
                      int C, int64_t B, void* d_actF, int64_t act_stride_bytes, float* d_pred, int64_t x_pos_cs, int64_t idx_cs, void* stream);
extern "C" int npp_light16_fwd(const npp_light_desc* L, const float* d_params, int64_t params_stride, const void* d_pack,
                               int64_t pack_stride_bytes, const float* d_x_per, const float* d_x_pos, const int64_t* d_idx, int64_t n_src,
                               int C, int64_t B, void* d_actF, int64_t act_stride_bytes, float* d_pred, void* stream) {
  return l16_fwd_go(L, d_params, params_stride, d_pack, pack_stride_bytes, d_x_per, d_x_pos, d_idx, n_src, C, B, d_actF, act_stride_bytes, d_pred, 0, 0,
                    stream);
}
// "multi" form (round 6): candidate c = one IMAGE's fit -- its own positional table d_x_pos (C, n_src, 42) and pixel rows d_idx (C, B)
extern "C" int npp_light16_fwd_multi(const npp_light_desc* L, const float* d_params, int64_t params_stride, const void* d_pack,
                                     int64_t pack_stride_bytes, const float* d_x_per, const float* d_x_pos, const int64_t* d_idx, int64_t n_src,
                                     int C, int64_t B, void* d_actF, int64_t act_stride_bytes, float* d_pred, void* stream) {
  if (!d_idx) { set_error("npp_light16_fwd_multi: null row indices"); return NPP_ERR_ARG; }
  return l16_fwd_go(L, d_params, params_stride, d_pack, pack_stride_bytes, d_x_per, d_x_pos, d_idx, n_src, C, B, d_actF, act_stride_bytes, d_pred,
                    n_src * kLPos, B, stream);
}
static int l16_fwd_go(const npp_light_desc* L, const float* d_params, int64_t params_stride, const void* d_pack,
                      int64_t pack_stride_bytes, const float* d_x_per, const float* d_x_pos, const int64_t* d_idx, int64_t n_src,
                      int C, int64_t B, void* d_actF, int64_t act_stride_bytes, float* d_pred, int64_t x_pos_cs, int64_t idx_cs, void* stream) {
  int rc = l16_check(L, d_params, d_pack, C, B, "npp_light16_fwd");
  if (rc) return rc;
  if (!d_x_per || !d_x_pos || !d_actF || !d_pred || (d_idx ? n_src < 1 : n_src != B) || pack_stride_bytes % 16 || act_stride_bytes % 16 ||
      act_stride_bytes < wfmt_array_base(L16A_TOTAL, B / kRowTile)) {
    set_error("npp_light16_fwd: null argument / n_src / strides");
    return NPP_ERR_ARG;
  }
  L16Args a{};
  a.L = *L; a.params = d_params; a.params_stride = params_stride; a.pack = (const bf16x8*)d_pack; a.pack_stride16 = pack_stride_bytes / 16;
  a.x_per = d_x_per; a.x_pos = d_x_pos; a.idx = d_idx; a.n_src = n_src; a.actF = (char*)d_actF; a.act_stride = act_stride_bytes;
  a.pred = d_pred; a.B = B; a.x_pos_cs = x_pos_cs; a.idx_cs = idx_cs;
  static SmemOnce once;
  if (!smem_attr(once, (const void*)light16_fwd_kernel, kL16SmemF)) { set_error("npp_light16_fwd: smem attribute"); return NPP_ERR_LAUNCH; }
  const int n_wg = (int)(B / kRowTile);
  hipLaunchKernelGGL(light16_fwd_kernel, dim3((unsigned)((n_wg * C + 7) / 8 * 8)), dim3(kL16Threads), kL16SmemF, (hipStream_t)stream, a,
                     l16_pack_desc(), n_wg, C);
  return check_launch("npp_light16_fwd");
}

static int l16_bwd_go(const npp_light_desc* L, const float* d_params, int64_t params_stride, const void* d_pack,
                      int64_t pack_stride_bytes, const void* d_actF, int64_t act_stride_bytes, const float* d_pred,
                      const float* d_dpred, const float* d_gt, const float* d_latents, const float* d_spline, int n_knots,
                      float x_scale, float* d_loss, float* d_dlatent, int C, int64_t B, void* d_dzF, int64_t dz_stride_bytes,
                      float* d_part, int64_t gt_cs, void* stream);
extern "C" int npp_light16_bwd(const npp_light_desc* L, const float* d_params, int64_t params_stride, const void* d_pack,
                               int64_t pack_stride_bytes, const void* d_actF, int64_t act_stride_bytes, const float* d_pred,
                               const float* d_dpred, const float* d_gt, const float* d_latents, const float* d_spline, int n_knots,
                               float x_scale, float* d_loss, float* d_dlatent, int C, int64_t B, void* d_dzF, int64_t dz_stride_bytes,
                               void* stream) {
  return l16_bwd_go(L, d_params, params_stride, d_pack, pack_stride_bytes, d_actF, act_stride_bytes, d_pred, d_dpred, d_gt, d_latents, d_spline,
                    n_knots, x_scale, d_loss, d_dlatent, C, B, d_dzF, dz_stride_bytes, nullptr, 0, stream);
}
// Bit-reproducible form (round 6): the pixel loss folded in (d_gt required), every block's seven loss / latent-gradient sums go to
// d_part (C, B / 64, 8) by plain stores; npp_light16_adam_pack_det adds them in block order.  gt_cs: elements per candidate of d_gt
// (0: one (B, 3) target shared by the candidates; 3 B: the "multi" form, candidate = one image's fit with its own targets (C, B, 3)).
extern "C" int npp_light16_bwd_det(const npp_light_desc* L, const float* d_params, int64_t params_stride, const void* d_pack,
                                   int64_t pack_stride_bytes, const void* d_actF, int64_t act_stride_bytes, const float* d_pred,
                                   const float* d_gt, int64_t gt_cs, const float* d_latents, const float* d_spline, int n_knots, float x_scale,
                                   float* d_part, int C, int64_t B, void* d_dzF, int64_t dz_stride_bytes, void* stream) {
  if (!d_gt || !d_part || (gt_cs != 0 && gt_cs != 3 * B)) { set_error("npp_light16_bwd_det: targets / partial-sum buffer / gt_cs"); return NPP_ERR_ARG; }
  return l16_bwd_go(L, d_params, params_stride, d_pack, pack_stride_bytes, d_actF, act_stride_bytes, d_pred, nullptr, d_gt, d_latents, d_spline,
                    n_knots, x_scale, nullptr, nullptr, C, B, d_dzF, dz_stride_bytes, d_part, gt_cs, stream);
}
static int l16_bwd_go(const npp_light_desc* L, const float* d_params, int64_t params_stride, const void* d_pack,
                      int64_t pack_stride_bytes, const void* d_actF, int64_t act_stride_bytes, const float* d_pred,
                      const float* d_dpred, const float* d_gt, const float* d_latents, const float* d_spline, int n_knots,
                      float x_scale, float* d_loss, float* d_dlatent, int C, int64_t B, void* d_dzF, int64_t dz_stride_bytes,
                      float* d_part, int64_t gt_cs, void* stream) {
  int rc = l16_check(L, d_params, d_pack, C, B, "npp_light16_bwd");
  if (rc) return rc;
  if (!d_actF || !d_pred || !d_dzF || (d_gt ? (!d_latents || !d_spline || n_knots < 2 || (!d_part && (!d_loss || !d_dlatent))) : !d_dpred) ||
      pack_stride_bytes % 16 || act_stride_bytes % 16 || dz_stride_bytes % 16 || act_stride_bytes < wfmt_array_base(L16A_TOTAL, B / kRowTile) ||
      dz_stride_bytes < wfmt_array_base(L16D_TOTAL, B / kRowTile)) {
    set_error("npp_light16_bwd: null argument (d_dpred, or d_gt with latents / spline / loss / dlatent) / strides");
    return NPP_ERR_ARG;
  }
  L16Args a{};
  a.L = *L; a.params = d_params; a.params_stride = params_stride; a.pack = (const bf16x8*)d_pack; a.pack_stride16 = pack_stride_bytes / 16;
  a.actF = (char*)d_actF; a.act_stride = act_stride_bytes; a.dzF = (char*)d_dzF; a.dz_stride = dz_stride_bytes;
  a.pred = (float*)d_pred; a.dpred = d_dpred; a.B = B;
  a.gt = d_gt; a.latents = d_latents; a.spline = d_spline; a.n_knots = n_knots; a.x_scale = x_scale; a.loss = d_loss; a.dlatent = d_dlatent;
  a.part = d_part; a.gt_cs = gt_cs;
  static SmemOnce once;
  if (!smem_attr(once, (const void*)light16_bwd_kernel, kL16SmemB)) { set_error("npp_light16_bwd: smem attribute"); return NPP_ERR_LAUNCH; }
  const int n_wg = (int)(B / kRowTile);
  hipLaunchKernelGGL(light16_bwd_kernel, dim3((unsigned)((n_wg * C + 7) / 8 * 8)), dim3(kL16Threads), kL16SmemB, (hipStream_t)stream, a,
                     l16_pack_desc(), n_wg, C);
  return check_launch("npp_light16_bwd");
}

static int l16_adam_go(const npp_light_desc* L, float* d_params, float* d_m, float* d_v, int64_t stride, int64_t n, int C,
                       const float* d_gslabs, int n_slabs, int64_t slab_stride, int64_t slab_cand_stride, void* d_pack,
                       int64_t pack_stride_bytes, float* d_lat, float* d_lat_m, float* d_lat_v, float* d_dlat, float* d_zero,
                       float lr, float beta1, float beta2, float eps, int step, const float* d_part, int n_part, float* d_loss_cur, void* stream);
extern "C" int npp_light16_adam_pack(const npp_light_desc* L, float* d_params, float* d_m, float* d_v, int64_t stride, int64_t n, int C,
                                     const float* d_gslabs, int n_slabs, int64_t slab_stride, int64_t slab_cand_stride, void* d_pack,
                                     int64_t pack_stride_bytes, float* d_lat, float* d_lat_m, float* d_lat_v, float* d_dlat, float* d_zero,
                                     float lr, float beta1, float beta2, float eps, int step, void* stream) {
  return l16_adam_go(L, d_params, d_m, d_v, stride, n, C, d_gslabs, n_slabs, slab_stride, slab_cand_stride, d_pack, pack_stride_bytes, d_lat, d_lat_m,
                     d_lat_v, d_dlat, d_zero, lr, beta1, beta2, eps, step, nullptr, 0, nullptr, stream);
}
// npp_light16_adam_pack after npp_light16_bwd_det: latent gradients = d_dlat + the blocks' sums of d_part (C, n_part, 8) in block order;
// d_loss_cur[c] += the blocks' loss terms (nullable)
extern "C" int npp_light16_adam_pack_det(const npp_light_desc* L, float* d_params, float* d_m, float* d_v, int64_t stride, int64_t n, int C,
                                         const float* d_gslabs, int n_slabs, int64_t slab_stride, int64_t slab_cand_stride, void* d_pack,
                                         int64_t pack_stride_bytes, float* d_lat, float* d_lat_m, float* d_lat_v, float* d_dlat, float* d_zero,
                                         float lr, float beta1, float beta2, float eps, int step, const float* d_part, int n_part,
                                         float* d_loss_cur, void* stream) {
  if (!d_part || n_part < 1) { set_error("npp_light16_adam_pack_det: partial sums"); return NPP_ERR_ARG; }
  return l16_adam_go(L, d_params, d_m, d_v, stride, n, C, d_gslabs, n_slabs, slab_stride, slab_cand_stride, d_pack, pack_stride_bytes, d_lat, d_lat_m,
                     d_lat_v, d_dlat, d_zero, lr, beta1, beta2, eps, step, d_part, n_part, d_loss_cur, stream);
}
static int l16_adam_go(const npp_light_desc* L, float* d_params, float* d_m, float* d_v, int64_t stride, int64_t n, int C,
                       const float* d_gslabs, int n_slabs, int64_t slab_stride, int64_t slab_cand_stride, void* d_pack,
                       int64_t pack_stride_bytes, float* d_lat, float* d_lat_m, float* d_lat_v, float* d_dlat, float* d_zero,
                       float lr, float beta1, float beta2, float eps, int step, const float* d_part, int n_part, float* d_loss_cur, void* stream) {
  int rc = l16_check(L, d_params, d_pack, C, kRowTile, "npp_light16_adam_pack");
  if (rc) return rc;
  const L16Pack pd = l16_pack_desc();
  if (!d_m || !d_v || !d_gslabs || !d_lat || !d_lat_m || !d_lat_v || !d_dlat || n < 1 || n > stride || n > 0x7fffffffLL || step < 1 ||
      n_slabs < 1 || slab_stride < n || slab_cand_stride < (int64_t)n_slabs * slab_stride || pack_stride_bytes < 16 * (int64_t)pd.total ||
      pack_stride_bytes % 16) {
    set_error("npp_light16_adam_pack: bad argument");
    return NPP_ERR_ARG;
  }
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  L16AdamArgs a{};
  a.L = *L; a.p = d_params; a.m = d_m; a.v = d_v; a.stride = stride; a.n = (int32_t)n;
  a.gslabs = d_gslabs; a.n_slabs = n_slabs; a.slab_stride = slab_stride; a.slab_cand_stride = slab_cand_stride;
  a.pack = (__bf16*)d_pack; a.pack_stride16 = pack_stride_bytes / 16;
  a.lat = d_lat; a.lat_m = d_lat_m; a.lat_v = d_lat_v; a.dlat = d_dlat; a.zero = d_zero;
  a.step_size = (float)((double)lr / bc1); a.b1 = beta1; a.b2 = beta2; a.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2)); a.eps = eps;
  a.part = d_part; a.n_part = n_part; a.loss_cur = d_loss_cur;
  hipLaunchKernelGGL(light16_adam_pack_kernel, dim3((unsigned)((n + 255) / 256 + 1), (unsigned)C), dim3(256), 0, (hipStream_t)stream, a, pd);
  return check_launch("npp_light16_adam_pack");
}
