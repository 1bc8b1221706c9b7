// npp_light.hip -- NPP_Net_light's training chains fused, exact fp32 (NPP_proposal/search.py:85-147 fits one
// NPP_Net_light per candidate; models/networks.py:176-263 with len(freq_scales) == 1, D = 4, W = 256, snake):
//
//   forward   x_per(20) -> 4 x [256, snake] -> feature_linear1 (256, linear) -> [f1 | x_pos(42)] -> pos_linears.0 (128, snake)
//             -> rgb_linear (3) -> sigmoid                                                         ONE launch
//   backward  dpred -> d raw -> d a_p -> d z_p -> d f1 -> d z_3 .. d z_0 (the data gradients)      ONE launch
//
// for ALL candidates of an image (blockIdx.y = candidate, each with its own weights) on the building blocks of the fp32 render
// chain (npp_chain32.h): a 256-thread workgroup owns 64 pixel rows, GEMMs transposed (Z^T = W X^T), weights streamed from a
// packed fp32 buffer as the A operand, activations in one LDS region [feature][64 rows] as the B operand, every contraction on
// v_mfma_f32_32x32x2_f32.  What the unfused path (npp_linear.hip, one launch per layer: 13 forward + data-gradient launches of
// 20-45 us for 9 stacked candidates) loses is not arithmetic but the per-launch ramp and the LDS staging of both operands;
// here an activation never leaves the chip between layers except as the stash the weight gradients need.
//
// Stashes are FEATURE-major ([feature][row], per candidate): an accumulator tile holds one row per lane, so a register of a
// tile is 32 consecutive rows of one feature -- a coalesced 128-byte store -- and the weight-gradient GEMMs (npp_linear.hip,
// batched, strided operands) take both operands contiguous along the rows they contract.
#include "npp_chain32.h"
#include "npp_light_layout.h"

namespace npp {

constexpr int light_region_bytes(int nb) { return kLHp * nb * 32 * 4; }   // the activation region: 77 824 B at 64 rows, 38 912 at 32
constexpr int kLThreads = 256;

// packed weights of one candidate (16-byte units): forward pack, then the transposed pack of the backward chain
enum { LF_L0 = 0, LF_L1, LF_L2, LF_L3, LF_F1, LF_POS, LF_N };
enum { LB_POS = 0, LB_F1, LB_L3, LB_L2, LB_L1, LB_N };
struct LightPackDesc {
  int32_t f_off[LF_N], f_groups[LF_N], f_nt[LF_N];
  int32_t b_off[LB_N], b_groups[LB_N];
  int32_t f_total, total;                       // units of the forward pack / of both
};
__host__ __device__ inline LightPackDesc light_pack_desc() {
  LightPackDesc d{};
  int off = 0;
  const int fg[LF_N] = {4, 32, 32, 32, 32, kLHp / 8}, fnt[LF_N] = {8, 8, 8, 8, 8, 4};
  for (int l = 0; l < LF_N; ++l) { d.f_off[l] = off; d.f_groups[l] = fg[l]; d.f_nt[l] = fnt[l]; off += fg[l] * fnt[l] * 64; }
  d.f_total = off;
  const int bg[LB_N] = {kLPosOut / 8, 32, 32, 32, 32};
  for (int l = 0; l < LB_N; ++l) { d.b_off[l] = off; d.b_groups[l] = bg[l]; off += bg[l] * 8 * 64; }
  d.total = off;
  return d;
}

struct LightArgs {
  npp_light_desc L;                             // parameter-blob offsets (include/npp_hip.h)
  const float* params; int64_t params_stride;   // (C, params_stride) fp32
  const float* pack; int64_t pack_stride;       // (C, pack_stride) fp32, pack_stride = 4 * LightPackDesc.total
  const float* x_per;                           // (C, B, 20)
  const float* x_pos;                           // (B, 42), shared
  float* stash;                                 // (C, LS_ROWS, B)
  float* pred;                                  // (C, B, 3)
  const float* dpred;                           // (C, B, 3)        backward only
  float* draw;                                  // (C, B, 3)        backward only
  float* dstash;                                // (C, LD_ROWS, B)  backward only
  int64_t B;
  const int64_t* idx; int64_t n_src;            // forward: rows idx[r] of x_per (C, n_src, 20) / x_pos (n_src, 42); idx null: rows r, n_src = B
  const float* gt;                              // backward with the pixel loss folded in: targets (B, 3) ... (null: d pred is an input)
  const float* latents; const float* spline; int n_knots; float x_scale;      // ... its adaptive-loss latents (C, 6) and spline table
  float* loss; float* dlatent;                  // ... and where the loss words (C) / latent gradients (C, 6) accumulate
  float* part;                                  // npp_light_bwd_det: (C, blocks, 8) -- every block leaves its seven sums here instead (no atomics)
  // "multi" forms (candidate = one IMAGE's fit: its own pixel rows, positional table and targets): elements per candidate, 0 = shared
  int64_t x_pos_cs, idx_cs, gt_cs;
};

// ---- packs ----------------------------------------------------------------------------------------------------------------
// unit u = [g][nt][lane]: element e = A[32 nt + (lane & 31)][8 g + e + 4 (lane >> 5)], A = W (forward) or W^T (backward),
// zero outside the matrix
__global__ void light_pack_kernel(LightArgs a, LightPackDesc pd, float* __restrict__ out, int64_t out_stride) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= pd.total) return;
  const float* P = a.params + (int64_t)blockIdx.y * a.params_stride;
  const bool bwd = u >= pd.f_total;
  int l = 0;
  if (!bwd) { for (int q = 1; q < LF_N; ++q) if (u >= pd.f_off[q]) l = q; }
  else { for (int q = 1; q < LB_N; ++q) if (u >= pd.b_off[q]) l = q; }
  const int r = u - (bwd ? pd.b_off[l] : pd.f_off[l]);
  const int nt_n = bwd ? 8 : pd.f_nt[l];
  const int lane = r & 63, nt = (r >> 6) % nt_n, g = (r >> 6) / nt_n;
  const int row = nt * 32 + (lane & 31), h = lane >> 5;
  // source matrix (out x in, leading dimension ld) of this pack entry
  const int fwd_layer[LF_N] = {0, 1, 2, 3, 5, 4};          // npp_light_desc index: periodic 0..3, pos (4), feature1 (5), rgb (6)
  const int bwd_layer[LB_N] = {4, 5, 3, 2, 1};
  const int li = bwd ? bwd_layer[l] : fwd_layer[l];
  const float* Wm = P + a.L.w_off[li];
  const int ld = a.L.ld[li], n_out = a.L.n_out[li], n_in = a.L.n_in[li];
  f32x4_t o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int col = 8 * g + e + 4 * h;
    float v = 0.0f;
    if (!bwd) { if (row < n_out && col < n_in) v = Wm[(int64_t)row * ld + col]; }
    else if (row < kLW && col < n_out) v = Wm[(int64_t)col * ld + row];        // A = W^T restricted to the first 256 inputs
    o[e] = v;
  }
  ((f32x4_t*)(out + (int64_t)blockIdx.y * out_stride))[u] = o;
}

// ---- optimizer.step() + zero_grad() + re-pack in one launch ---------------------------------------------------------------------
// Adam over the stacked blobs (torch.optim.Adam single-tensor maths, adam_update in npp_common.h), every updated weight scattered into
// the packs through the inverse of light_pack_kernel's map, the gradient cleared on the way; the extra block column (blockIdx.x ==
// gridDim.x - 1) steps the candidate's six adaptive-loss latents and clears one loss word.
struct LightAdamArgs {
  npp_light_desc L;
  float *p, *m, *v, *g; int64_t stride; int32_t n;         // (C, stride) blobs, n live floats per candidate
  float* pack; int64_t pack_stride;
  float *lat, *lat_m, *lat_v, *dlat, *zero;               // (C, 6) x 4, (C)
  float step_size, b1, b2, inv_sqrt_bc2, eps;
  const float* part; int32_t n_part; float* loss_cur;     // npp_light_adam_pack_det: the blocks' partial sums, added here in block order
};
__global__ __launch_bounds__(256) void light_adam_pack_kernel(LightAdamArgs a, LightPackDesc pd) {
  const int c = blockIdx.y;
  if (blockIdx.x == gridDim.x - 1) {
    const int t = threadIdx.x;
    if (t < 6) {
      const int i = c * 6 + t;
      float g = a.dlat[i];
      if (a.part)                                             // fixed order: bit-reproducible latent gradients
        for (int b = 0; b < a.n_part; ++b) g += a.part[((int64_t)c * a.n_part + b) * 8 + 1 + t];
      float m = a.lat_m[i], v = a.lat_v[i];
      a.lat[i] = adam_update(a.lat[i], m, v, g, a.step_size, a.b1, a.b2, a.inv_sqrt_bc2, a.eps);
      a.lat_m[i] = m; a.lat_v[i] = v; a.dlat[i] = 0.0f;
    } else if (t == 6 && a.zero) a.zero[c] = 0.0f;
    else if (t == 64 && a.part && a.loss_cur) {               // (another wave: the two sums run side by side)
      float l = 0.0f;
      for (int b = 0; b < a.n_part; ++b) l += a.part[((int64_t)c * a.n_part + b) * 8];
      a.loss_cur[c] += l;
    }
    return;
  }
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  const int64_t gi = (int64_t)c * a.stride + i;
  float m = a.m[gi], v = a.v[gi];
  const float w = adam_update(a.p[gi], m, v, a.g[gi], a.step_size, a.b1, a.b2, a.inv_sqrt_bc2, a.eps);
  a.p[gi] = w; a.m[gi] = m; a.v[gi] = v; a.g[gi] = 0.0f;
  // which weight is this?  (biases and rgb_linear are read from the blob by the chains: nothing to scatter)
  int li = -1;
#pragma unroll
  for (int q = 0; q < 6; ++q)
    if (i >= a.L.w_off[q] && i < a.L.w_off[q] + (int64_t)a.L.n_out[q] * a.L.ld[q]) li = q;
  if (li < 0) return;
  const int off = i - (int)a.L.w_off[li], ld = a.L.ld[li];
  const int row = off / ld, col = off - row * ld;
  float* pk = a.pack + (int64_t)c * a.pack_stride;
  const int lf = li < 4 ? LF_L0 + li : (li == 4 ? LF_POS : LF_F1);
  if (col < pd.f_groups[lf] * 8)
    pk[(int64_t)(pd.f_off[lf] + ((col >> 3) * pd.f_nt[lf] + (row >> 5)) * 64 + ((col & 7) >> 2) * 32 + (row & 31)) * 4 + (col & 3)] = w;
  if (li >= 1 && col < kLW) {                                  // transposed pack: A[m = col][k = row]
    const int lb = li == 4 ? LB_POS : (li == 5 ? LB_F1 : (li == 3 ? LB_L3 : (li == 2 ? LB_L2 : LB_L1)));
    pk[(int64_t)(pd.b_off[lb] + ((row >> 3) * 8 + (col >> 5)) * 64 + ((row & 7) >> 2) * 32 + (col & 31)) * 4 + (row & 3)] = w;
  }
}

// ---- forward ---------------------------------------------------------------------------------------------------------------
// epilogue of a hidden layer: z (+ bias already in acc) -> stash zT, h = snake(z) (or z) -> region (+ stash hT).  Only z is stashed for
// the snake layers: the weight-gradient GEMM forms h = snake(z) again while it stages the operand (npp_linear_bwd_weight_strided)
template <bool SNAKE, int NTW, int NB>
__device__ __forceinline__ void light_epi(f32x16 (&acc)[NTW][NB], char* region, float* __restrict__ zT, float* __restrict__ hT, uint32_t B,
                                          uint32_t row0, int nt0, int b, int h) {
  // 32-bit element indices off the (uniform) array bases: the launcher bounds rows x B below 2^31
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
    for (int bt = 0; bt < NB; ++bt) {
      const uint32_t f0 = (uint32_t)((nt0 + nt) * 32 + 4 * h);
      uint32_t g = f0 * B + row0 + (uint32_t)(bt * 32 + b);
      char* rg = region ? region + (f0 * (NB * 32) + bt * 32 + b) * 4 : nullptr;
#pragma unroll
      for (int r = 0; r < 16; ++r) {            // feature f0 + (r & 3) + 8 (r >> 2)
        const uint32_t gi = g + (uint32_t)((r & 3) + 8 * (r >> 2)) * B;
        float z = acc[nt][bt][r];
        if (zT) zT[gi] = z;
        if (SNAKE) z = snake_fast(z);
        acc[nt][bt][r] = z;
        if (hT) hT[gi] = z;
        if (region) *(float*)(rg + ((r & 3) + 8 * (r >> 2)) * (NB * 32) * 4) = z;
      }
    }
}

template <int NB>
__global__ __launch_bounds__(kLThreads, NB == 2 ? 2 : 3) void light_fwd_kernel(LightArgs a, LightPackDesc pd) {
  constexpr int RT = NB * 32;                   // pixel rows per workgroup
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* R = smem;
  const int tid = threadIdx.x, lane = tid & 63, b = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = blockIdx.y;
  const int64_t B = a.B, row0 = (int64_t)blockIdx.x * RT;
  const float* P = a.params + (int64_t)c * a.params_stride;
  float* S = a.stash + (int64_t)c * LS_ROWS * B;
  const int nt0 = 2 * wave;
  // x_per tile -> region features 0..31 (20 real, 12 zero: L0 is packed as 4 k-step groups)
  {
    const float* xp = a.x_per + (int64_t)c * a.n_src * kLPer;
    float* Rf = (float*)R;
    for (int i = tid; i < 32 * RT; i += kLThreads) {
      const int f = i / RT, row = i % RT;
      const int64_t src = a.idx ? a.idx[c * a.idx_cs + row0 + row] : row0 + row;
      const float v = f < kLPer ? xp[src * kLPer + f] : 0.0f;
      Rf[f * RT + row] = v;
      if (f < kLPer) S[(int64_t)(LS_XP + f) * B + row0 + row] = v;        // x_per^T: the first layer's weight-gradient operand
    }
  }
  wg_barrier();
  const wrsrc_t rsrc = make_wrsrc(a.pack + (int64_t)c * a.pack_stride, pd.total);
  const ActSrc32 act{R + ((4 * h) * RT + b) * 4};
  f32x16 acc[2][NB];
#pragma unroll 1
  for (int l = 0; l < 4; ++l) {
    bias32(acc, P + a.L.b_off[l], nt0, h);
    part32<2, 8>(acc, rsrc, (uint32_t)pd.f_off[LF_L0 + l], pd.f_groups[LF_L0 + l], nt0, lane, act);
    wg_barrier();                               // every wave has read its last operand of this layer
    light_epi<true>(acc, R, S + (int64_t)(LS_Z0 + 256 * l) * B, nullptr, (uint32_t)B, (uint32_t)row0, nt0, b, h);
    wg_barrier();
  }
  // feature_linear1 (linear) -> region rows 0..255 and the first 256 rows of hpT; x_pos (+ zero pad) behind it
  bias32(acc, P + a.L.b_off[5], nt0, h);
  part32<2, 8>(acc, rsrc, (uint32_t)pd.f_off[LF_F1], pd.f_groups[LF_F1], nt0, lane, act);
  wg_barrier();
  light_epi<false>(acc, R, nullptr, S + (int64_t)LS_HP * B, (uint32_t)B, (uint32_t)row0, nt0, b, h);
  {
    float* Rf = (float*)R;
    float* hp = S + (int64_t)LS_HP * B;
    for (int i = tid; i < (kLHp - kLW) * RT; i += kLThreads) {
      const int f = i / RT, row = i % RT;
      const float v = f < kLPos ? a.x_pos[c * a.x_pos_cs + (a.idx ? a.idx[c * a.idx_cs + row0 + row] : row0 + row) * kLPos + f] : 0.0f;
      Rf[(kLW + f) * RT + row] = v;
      hp[(int64_t)(kLW + f) * B + row0 + row] = v;
    }
  }
  wg_barrier();
  // pos_linears.0: 304 -> 128 (snake), one neuron tile per wave; a_p stays in registers for rgb_linear
  f32x16 accp[1][NB];
  bias32(accp, P + a.L.b_off[4], wave, h);
  part32<1, 4>(accp, rsrc, (uint32_t)pd.f_off[LF_POS], pd.f_groups[LF_POS], wave, lane, act);
  light_epi<true>(accp, nullptr, S + (int64_t)LS_ZP * B, nullptr, (uint32_t)B, (uint32_t)row0, wave, b, h);
  // rgb_linear 128 -> 3 + sigmoid (models/helpers.py:55-56)
  wg_barrier();
  float* sRGB = (float*)R;                      // [4 waves][64 rows][3]
  {
    const float* Wr = P + a.L.w_off[6];
    const int ldr = a.L.ld[6];
    float part[NB][3];
#pragma unroll
    for (int bt = 0; bt < NB; ++bt)
#pragma unroll
      for (int q = 0; q < 3; ++q) part[bt][q] = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = wave * 32 + acc_row(r, h);
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const float w = Wr[q * ldr + k];
#pragma unroll
        for (int bt = 0; bt < NB; ++bt) part[bt][q] = fmaf(w, accp[0][bt][r], part[bt][q]);
      }
    }
#pragma unroll
    for (int bt = 0; bt < NB; ++bt)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const float v = part[bt][q] + __shfl_xor(part[bt][q], 32, 64);
        if (h == 0) sRGB[(wave * RT + bt * 32 + b) * 3 + q] = v;
      }
  }
  wg_barrier();
  if (tid < RT * 3) {
    const int row = tid / 3, q = tid - row * 3;
    float z = P[a.L.b_off[6] + q];
#pragma unroll
    for (int w = 0; w < 4; ++w) z += sRGB[(w * RT + row) * 3 + q];
    a.pred[((int64_t)c * B + row0 + row) * 3 + q] = 1.0f / (1.0f + expf(-z));
  }
}

// ---- backward (data gradients) ------------------------------------------------------------------------------------------------
// epilogue: d h -> d z = d h * snake'(z) (z from the forward stash; DERIV false: d z = d h) -> gradient stash (+ region)
// PRE: the z tile was fetched into zpre before the MFMA loop of this layer (32-row workgroups: 32 registers, loads in flight under the
// whole contraction instead of a round trip at its end)
template <bool DERIV, bool PRE, int NB>
__device__ __forceinline__ void light_bepi(f32x16 (&acc)[2][NB], char* region, const float* __restrict__ zT, float* __restrict__ dT, uint32_t B,
                                           uint32_t row0, int nt0, int b, int h, const f32x16 (&zpre)[2][NB]) {
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int bt = 0; bt < NB; ++bt) {
      const uint32_t f0 = (uint32_t)((nt0 + nt) * 32 + 4 * h);
      const uint32_t g = f0 * B + row0 + (uint32_t)(bt * 32 + b);
      char* rg = region ? region + (f0 * (NB * 32) + bt * 32 + b) * 4 : nullptr;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint32_t gi = g + (uint32_t)((r & 3) + 8 * (r >> 2)) * B;
        float d = acc[nt][bt][r];
        if (DERIV) d *= 1.0f + __builtin_amdgcn_sinf((PRE ? zpre[nt][bt][r] : zT[gi]) * (2.0f * kInv2Pi));      // activations.py:29-35: 1 + sin 2z
        dT[gi] = d;
        if (region) *(float*)(rg + ((r & 3) + 8 * (r >> 2)) * (NB * 32) * 4) = d;
      }
    }
}
template <int NB>
__device__ __forceinline__ void light_zfetch(f32x16 (&zpre)[2][NB], const float* __restrict__ zT, uint32_t B, uint32_t row0, int nt0, int b, int h) {
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int bt = 0; bt < NB; ++bt) {
      const uint32_t g = (uint32_t)((nt0 + nt) * 32 + 4 * h) * B + row0 + (uint32_t)(bt * 32 + b);
#pragma unroll
      for (int r = 0; r < 16; ++r) zpre[nt][bt][r] = zT[g + (uint32_t)((r & 3) + 8 * (r >> 2)) * B];
    }
}
template <int NB>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[2][NB]) {
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int bt = 0; bt < NB; ++bt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][bt][r] = 0.0f;
}

template <int NB>
__global__ __launch_bounds__(kLThreads, NB == 2 ? 2 : 3) void light_bwd_kernel(LightArgs a, LightPackDesc pd) {
  constexpr int RT = NB * 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* R = smem;
  float* sD = (float*)(smem + kLW * RT * 4);          // d raw [64 rows][3] behind the 256-feature part of the region
  const int tid = threadIdx.x, lane = tid & 63, b = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = blockIdx.y;
  const int64_t B = a.B, row0 = (int64_t)blockIdx.x * RT;
  const float* P = a.params + (int64_t)c * a.params_stride;
  const float* S = a.stash + (int64_t)c * LS_ROWS * B;
  float* D = a.dstash + (int64_t)c * LD_ROWS * B;
  const int nt0 = 2 * wave;
  // d raw = d pred * pred (1 - pred); with the pixel loss folded in (a.gt): d pred = d img2mse(robust_loss_adaptive)/d pred right here
  // (models/mse_calculator.py:13-27 without a mask: the arithmetic of pixel_loss_body, npp_common.h), loss / latent gradients by atomics
  __shared__ ChanParams cp[3];
  __shared__ float sred[7];
  __shared__ float swv[kLThreads / 64][7];
  if (a.gt) {
    if (tid < 3) cp[tid] = chan_params(a.latents[c * 6 + tid], a.latents[c * 6 + 3 + tid], a.spline, a.n_knots, a.x_scale);
    if (tid < 7) sred[tid] = 0.0f;
    wg_barrier();
  }
  float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f;                 // this thread's loss term and latent-gradient terms (channel tid % 3)
  if (tid < RT * 3) {
    const int64_t g = ((int64_t)c * B + row0) * 3 + tid;
    const float p = a.pred[g];
    float dp;
    if (a.gt) {
      const int ch = tid % 3;
      const ChanParams q = cp[ch];
      const float inv = 1.0f / (3.0f * (float)B);
      const float x = p - a.gt[c * a.gt_cs + row0 * 3 + tid];
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
    const float d = dp * p * (1.0f - p);
    a.draw[g] = d;
    sD[tid] = d;
    D[(int64_t)(LD_RAW + tid % 3) * B + row0 + tid / 3] = d;       // d raw^T for rgb_linear's weight gradient
  }
  if (a.gt && a.part) {
    // deterministic form (every wave, whole: threads past the 3 RT values carry zeros): the seven sums of a wave by shuffle
    // butterflies -- a fixed tree, masked-out lanes add exact zeros -- then the waves' results in wave order
    const int ch = tid % 3;
    const float v7[7] = {t0, ch == 0 ? t1 : 0.0f, ch == 1 ? t1 : 0.0f, ch == 2 ? t1 : 0.0f, ch == 0 ? t2 : 0.0f, ch == 1 ? t2 : 0.0f, ch == 2 ? t2 : 0.0f};
#pragma unroll
    for (int k7 = 0; k7 < 7; ++k7) {
      float v = v7[k7];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
      if (lane == 0) swv[wave][k7] = v;
    }
  }
  wg_barrier();
  if (a.gt && tid < 7) {
    const float inv = 1.0f / (3.0f * (float)B);
    if (a.part) {
      float v = 0.0f;
#pragma unroll
      for (int w_ = 0; w_ < kLThreads / 64; ++w_) v += swv[w_][tid];                        // wave order
      const float o = tid == 0 ? v * inv : (tid < 4 ? inv * v * cp[tid - 1].dalpha_dl : inv * v * cp[tid - 4].dc_dl);
      a.part[((int64_t)c * gridDim.x + blockIdx.x) * 8 + tid] = o;                          // summed in block order by npp_light_adam_pack_det
    } else {
      const float v = sred[tid];
      if (tid == 0) atomicAdd(a.loss + c, v * inv);
      else if (tid < 4) atomicAdd(a.dlatent + c * 6 + (tid - 1), inv * v * cp[tid - 1].dalpha_dl);
      else atomicAdd(a.dlatent + c * 6 + 3 + (tid - 4), inv * v * cp[tid - 4].dc_dl);
    }
  }
  // d a_p = d raw W_rgb; d z_p = d a_p * snake'(z_p) -> region rows 0..127 + stash
  {
    const float* Wr = P + a.L.w_off[6];
    const int ldr = a.L.ld[6];
    const float* zp = S + (int64_t)LS_ZP * B;
    float* dzp = D + (int64_t)LD_ZP * B;
    float* Rf = (float*)R;
    for (int i = tid; i < kLPosOut * RT; i += kLThreads) {
      const int k = i / RT, row = i % RT;
      float d = sD[row * 3] * Wr[k];
      d = fmaf(sD[row * 3 + 1], Wr[ldr + k], d);
      d = fmaf(sD[row * 3 + 2], Wr[2 * ldr + k], d);
      const int64_t g = (int64_t)k * B + row0 + row;
      d *= 1.0f + __builtin_amdgcn_sinf(zp[g] * (2.0f * kInv2Pi));
      dzp[g] = d;
      Rf[k * RT + row] = d;
    }
  }
  wg_barrier();
  const wrsrc_t rsrc = make_wrsrc(a.pack + (int64_t)c * a.pack_stride, pd.total);
  const ActSrc32 act{R + ((4 * h) * RT + b) * 4};
  f32x16 acc[2][NB];
  // d f1 = W_pos[:, :256]^T d z_p   (feature_linear1 is linear: this IS its d z; x_pos gets no gradient)
  zero_acc(acc);
  part32<2, 8>(acc, rsrc, (uint32_t)pd.b_off[LB_POS], pd.b_groups[LB_POS], nt0, lane, act);
  wg_barrier();
  f32x16 zpre[2][NB];
  light_bepi<false, false>(acc, R, nullptr, D + (int64_t)LD_F1 * B, (uint32_t)B, (uint32_t)row0, nt0, b, h, zpre);
  wg_barrier();
  // d z_3 = (W_f1^T d f1) * snake'(z_3), d z_2 = (W_3^T d z_3) * snake'(z_2), ..., d z_0
#pragma unroll 1
  for (int j = 0; j < 4; ++j) {
    const int l = 3 - j;                        // hidden layer whose d z this step produces
    zero_acc(acc);
    constexpr bool kPre = NB == 1;
    if (kPre) light_zfetch(zpre, S + (int64_t)(LS_Z0 + 256 * l) * B, (uint32_t)B, (uint32_t)row0, nt0, b, h);
    part32<2, 8>(acc, rsrc, (uint32_t)pd.b_off[LB_F1 + j], pd.b_groups[LB_F1 + j], nt0, lane, act);
    wg_barrier();
    light_bepi<true, kPre>(acc, l > 0 ? R : nullptr, S + (int64_t)(LS_Z0 + 256 * l) * B, D + (int64_t)(LD_Z0 + 256 * l) * B, (uint32_t)B, (uint32_t)row0, nt0, b, h,
                           zpre);
    wg_barrier();
  }
}

}  // namespace npp

using namespace npp;

static int light_check(const npp_light_desc* L, const void* p0, const void* p1, int C, int64_t B, const char* who) {
  if (!L || !p0 || !p1 || C < 1 || C > 65535 || B < 32 || B % 32 || B * 512 >= 0x7fffffffLL) {
    set_error("%s: bad argument (C=%d B=%lld; B a positive multiple of 32)", who, C, (long long)B);
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

// Rows per workgroup: 64 (two batch tiles share every weight fragment) or 32.  A chain is one long dependent sequence per workgroup
// (forward: 98 us for 64 rows, 59 us for 32, alone on a CU), so what counts is the most loaded CU and how well the workgroups on it
// cover each other's epilogues.  Measured, 2048 rows per candidate (tools/r3_light_chain_probe.py, us, 64 / 32 rows):
//   candidates      1          4          8          9          12         16
//   forward      98 / 59   114 / 76   116 / 106  165 / 139  170 / 152  183 / 225
//   backward    117 / 64   167 / 96   168 / 121  221 / 151  249 / 174  293 / 231
// 32 rows until the chip holds more than ~3 (forward) / ~6 (backward) of them per CU, then the weight traffic of the narrow tile
// (every fragment feeds one MFMA instead of two) costs more than the balance gives.
static int light_rows_per_wg(int C, int64_t B, bool backward) {
  static const int forced = [] { const char* e = getenv("NPP_LIGHT_ROWS"); return e ? atoi(e) : 0; }();
  if (B % 64) return 32;
  if (forced == 32 || forced == 64) return forced;
  return C * (B / 32) <= (backward ? 1536 : 768) ? 32 : 64;
}

extern "C" int64_t npp_light_pack_floats(void) { return 4 * (int64_t)light_pack_desc().total; }
extern "C" int64_t npp_light_stash_rows(void) { return LS_ROWS; }
extern "C" int64_t npp_light_dstash_rows(void) { return LD_ROWS; }
extern "C" int npp_light_stash_row(int which) {
  const int rows[8] = {LS_Z0, LS_Z1, LS_Z2, LS_Z3, LS_HP, LS_ZP, LS_XP, LS_ROWS};
  return (which < 0 || which > 7) ? NPP_ERR_ARG : rows[which];
}
extern "C" int npp_light_dstash_row(int which) {
  const int rows[8] = {LD_Z0, LD_Z1, LD_Z2, LD_Z3, LD_F1, LD_ZP, LD_RAW, LD_ROWS};
  return (which < 0 || which > 7) ? NPP_ERR_ARG : rows[which];
}

extern "C" int npp_light_pack(const npp_light_desc* L, const float* d_params, int64_t params_stride, int C, float* d_pack, int64_t pack_stride,
                              void* stream) {
  int rc = light_check(L, d_params, d_pack, C, 32, "npp_light_pack");
  if (rc) return rc;
  const LightPackDesc pd = light_pack_desc();
  if (pack_stride < 4 * (int64_t)pd.total || pack_stride % 4) { set_error("npp_light_pack: pack_stride %lld", (long long)pack_stride); return NPP_ERR_ARG; }
  LightArgs a{};
  a.L = *L; a.params = d_params; a.params_stride = params_stride;
  hipLaunchKernelGGL(light_pack_kernel, dim3((unsigned)((pd.total + 255) / 256), (unsigned)C), dim3(256), 0, (hipStream_t)stream, a, pd, d_pack,
                     pack_stride);
  return check_launch("npp_light_pack");
}

static int light_fwd_go(const npp_light_desc* L, const float* d_params, int64_t params_stride, const float* d_pack, int64_t pack_stride,
                             const float* d_x_per, const float* d_x_pos, const int64_t* d_idx, int64_t n_src, int C, int64_t B, float* d_stash,
                             float* d_pred, void* stream, int64_t x_pos_cs, int64_t idx_cs) {
  int rc = light_check(L, d_params, d_pack, C, B, "npp_light_fwd");
  if (rc) return rc;
  if (!d_x_per || !d_x_pos || !d_stash || !d_pred || (d_idx ? n_src < 1 : n_src != B)) { set_error("npp_light_fwd: null argument / n_src"); return NPP_ERR_ARG; }
  LightArgs a{};
  a.L = *L; a.params = d_params; a.params_stride = params_stride; a.pack = d_pack; a.pack_stride = pack_stride;
  a.x_per = d_x_per; a.x_pos = d_x_pos; a.stash = d_stash; a.pred = d_pred; a.B = B; a.idx = d_idx; a.n_src = n_src; a.x_pos_cs = x_pos_cs; a.idx_cs = idx_cs;
  if (light_rows_per_wg(C, B, false) == 64) {
    static SmemOnce once;
    if (!smem_attr(once, (const void*)light_fwd_kernel<2>, light_region_bytes(2))) { set_error("npp_light_fwd: smem attribute"); return NPP_ERR_LAUNCH; }
    hipLaunchKernelGGL(light_fwd_kernel<2>, dim3((unsigned)(B / 64), (unsigned)C), dim3(kLThreads), light_region_bytes(2), (hipStream_t)stream, a,
                       light_pack_desc());
  } else {
    static SmemOnce once;
    if (!smem_attr(once, (const void*)light_fwd_kernel<1>, light_region_bytes(1))) { set_error("npp_light_fwd: smem attribute"); return NPP_ERR_LAUNCH; }
    hipLaunchKernelGGL(light_fwd_kernel<1>, dim3((unsigned)(B / 32), (unsigned)C), dim3(kLThreads), light_region_bytes(1), (hipStream_t)stream, a,
                       light_pack_desc());
  }
  return check_launch("npp_light_fwd");
}

extern "C" int npp_light_fwd(const npp_light_desc* L, const float* d_params, int64_t params_stride, const float* d_pack, int64_t pack_stride,
                             const float* d_x_per, const float* d_x_pos, const int64_t* d_idx, int64_t n_src, int C, int64_t B, float* d_stash,
                             float* d_pred, void* stream) {
  return light_fwd_go(L, d_params, params_stride, d_pack, pack_stride, d_x_per, d_x_pos, d_idx, n_src, C, B, d_stash, d_pred, stream, 0, 0);
}
// npp_light_fwd where every candidate has its OWN positional table and pixel rows (candidate c = image c's fit: the search of several
// images advanced in one launch sequence): d_x_pos (C, n_src, 42), d_idx (C, B) -- tables of fewer than n_src rows are padded
extern "C" int npp_light_fwd_multi(const npp_light_desc* L, const float* d_params, int64_t params_stride, const float* d_pack, int64_t pack_stride,
                                   const float* d_x_per, const float* d_x_pos, const int64_t* d_idx, int64_t n_src, int C, int64_t B,
                                   float* d_stash, float* d_pred, void* stream) {
  if (!d_idx) { set_error("npp_light_fwd_multi: d_idx is required"); return NPP_ERR_ARG; }
  return light_fwd_go(L, d_params, params_stride, d_pack, pack_stride, d_x_per, d_x_pos, d_idx, n_src, C, B, d_stash, d_pred, stream,
                      n_src * kLPos, B);
}
extern "C" int npp_light_part_blocks(int C, int64_t B) { return (C < 1 || B < 32 || B % 32) ? NPP_ERR_ARG : (int)(B / light_rows_per_wg(C, B, true)); }

static int light_bwd_go(const npp_light_desc* L, const float* d_params, int64_t params_stride, const float* d_pack, int64_t pack_stride,
                        const float* d_stash, const float* d_pred, const float* d_dpred, const float* d_gt, const float* d_latents,
                        const float* d_spline, int n_knots, float x_scale, float* d_loss, float* d_dlatent, int C, int64_t B,
                        float* d_draw, float* d_dstash, float* d_part, void* stream, int64_t gt_cs = 0);

extern "C" int npp_light_bwd(const npp_light_desc* L, const float* d_params, int64_t params_stride, const float* d_pack, int64_t pack_stride,
                             const float* d_stash, const float* d_pred, const float* d_dpred, const float* d_gt, const float* d_latents,
                             const float* d_spline, int n_knots, float x_scale, float* d_loss, float* d_dlatent, int C, int64_t B,
                             float* d_draw, float* d_dstash, void* stream) {
  return light_bwd_go(L, d_params, params_stride, d_pack, pack_stride, d_stash, d_pred, d_dpred, d_gt, d_latents, d_spline, n_knots, x_scale,
                      d_loss, d_dlatent, C, B, d_draw, d_dstash, nullptr, stream);
}

// npp_light_bwd with the folded pixel loss's sums left per block in d_part (C, npp_light_part_blocks(C, B), 8) instead of added to
// d_loss / d_dlatent by float atomics: npp_light_adam_pack_det adds them in block order -- candidate fits are bit-reproducible
extern "C" int npp_light_bwd_det(const npp_light_desc* L, const float* d_params, int64_t params_stride, const float* d_pack, int64_t pack_stride,
                                 const float* d_stash, const float* d_pred, const float* d_gt, const float* d_latents, const float* d_spline,
                                 int n_knots, float x_scale, float* d_part, int C, int64_t B, float* d_draw, float* d_dstash, void* stream) {
  if (!d_gt || !d_part) { set_error("npp_light_bwd_det: d_gt and d_part are required"); return NPP_ERR_ARG; }
  return light_bwd_go(L, d_params, params_stride, d_pack, pack_stride, d_stash, d_pred, nullptr, d_gt, d_latents, d_spline, n_knots, x_scale,
                      d_part, d_part, C, B, d_draw, d_dstash, d_part, stream);
}

// npp_light_bwd_det with targets per candidate, d_gt (C, B, 3) (see npp_light_fwd_multi)
extern "C" int npp_light_bwd_det_multi(const npp_light_desc* L, const float* d_params, int64_t params_stride, const float* d_pack,
                                       int64_t pack_stride, const float* d_stash, const float* d_pred, const float* d_gt, const float* d_latents,
                                       const float* d_spline, int n_knots, float x_scale, float* d_part, int C, int64_t B, float* d_draw,
                                       float* d_dstash, void* stream) {
  if (!d_gt || !d_part) { set_error("npp_light_bwd_det_multi: d_gt and d_part are required"); return NPP_ERR_ARG; }
  return light_bwd_go(L, d_params, params_stride, d_pack, pack_stride, d_stash, d_pred, nullptr, d_gt, d_latents, d_spline, n_knots, x_scale,
                      d_part, d_part, C, B, d_draw, d_dstash, d_part, stream, B * 3);
}

static int light_bwd_go(const npp_light_desc* L, const float* d_params, int64_t params_stride, const float* d_pack, int64_t pack_stride,
                        const float* d_stash, const float* d_pred, const float* d_dpred, const float* d_gt, const float* d_latents,
                        const float* d_spline, int n_knots, float x_scale, float* d_loss, float* d_dlatent, int C, int64_t B,
                        float* d_draw, float* d_dstash, float* d_part, void* stream, int64_t gt_cs) {
  int rc = light_check(L, d_params, d_pack, C, B, "npp_light_bwd");
  if (rc) return rc;
  if (!d_stash || !d_pred || !d_draw || !d_dstash || (d_gt ? (!d_latents || !d_spline || n_knots < 2 || !d_loss || !d_dlatent) : !d_dpred)) {
    set_error("npp_light_bwd: null argument (d_dpred, or d_gt with latents / spline / loss / dlatent)");
    return NPP_ERR_ARG;
  }
  LightArgs a{};
  a.L = *L; a.params = d_params; a.params_stride = params_stride; a.pack = d_pack; a.pack_stride = pack_stride;
  a.stash = (float*)d_stash; a.pred = (float*)d_pred; a.dpred = d_dpred; a.draw = d_draw; a.dstash = d_dstash; a.B = B;
  a.gt = d_gt; a.latents = d_latents; a.spline = d_spline; a.n_knots = n_knots; a.x_scale = x_scale; a.loss = d_loss; a.dlatent = d_dlatent;
  a.part = d_part; a.gt_cs = gt_cs;
  if (light_rows_per_wg(C, B, true) == 64) {
    static SmemOnce once;
    if (!smem_attr(once, (const void*)light_bwd_kernel<2>, light_region_bytes(2))) { set_error("npp_light_bwd: smem attribute"); return NPP_ERR_LAUNCH; }
    hipLaunchKernelGGL(light_bwd_kernel<2>, dim3((unsigned)(B / 64), (unsigned)C), dim3(kLThreads), light_region_bytes(2), (hipStream_t)stream, a,
                       light_pack_desc());
  } else {
    static SmemOnce once;
    if (!smem_attr(once, (const void*)light_bwd_kernel<1>, light_region_bytes(1))) { set_error("npp_light_bwd: smem attribute"); return NPP_ERR_LAUNCH; }
    hipLaunchKernelGGL(light_bwd_kernel<1>, dim3((unsigned)(B / 32), (unsigned)C), dim3(kLThreads), light_region_bytes(1), (hipStream_t)stream, a,
                       light_pack_desc());
  }
  return check_launch("npp_light_bwd");
}

static int light_adam_go(const npp_light_desc* L, float* d_params, float* d_m, float* d_v, float* d_grad, int64_t stride, int64_t n, int C,
                         float* d_pack, int64_t pack_stride, float* d_lat, float* d_lat_m, float* d_lat_v, float* d_dlat, float* d_zero,
                         float lr, float beta1, float beta2, float eps, int step, const float* d_part, int n_part, float* d_loss_cur, void* stream);
extern "C" int npp_light_adam_pack(const npp_light_desc* L, float* d_params, float* d_m, float* d_v, float* d_grad, int64_t stride, int64_t n, int C,
                                   float* d_pack, int64_t pack_stride, float* d_lat, float* d_lat_m, float* d_lat_v, float* d_dlat, float* d_zero,
                                   float lr, float beta1, float beta2, float eps, int step, void* stream) {
  return light_adam_go(L, d_params, d_m, d_v, d_grad, stride, n, C, d_pack, pack_stride, d_lat, d_lat_m, d_lat_v, d_dlat, d_zero, lr, beta1, beta2,
                       eps, step, nullptr, 0, nullptr, stream);
}
// npp_light_adam_pack after npp_light_bwd_det: the latent gradients are d_dlat (+) the n_part per-block sums of d_part in block order,
// and the iteration's loss word d_loss_cur[c] (nullable) receives the blocks' loss terms the same way
extern "C" int npp_light_adam_pack_det(const npp_light_desc* L, float* d_params, float* d_m, float* d_v, float* d_grad, int64_t stride, int64_t n,
                                       int C, float* d_pack, int64_t pack_stride, float* d_lat, float* d_lat_m, float* d_lat_v, float* d_dlat,
                                       float* d_zero, float lr, float beta1, float beta2, float eps, int step, const float* d_part, int n_part,
                                       float* d_loss_cur, void* stream) {
  if (!d_part || n_part < 1) { set_error("npp_light_adam_pack_det: d_part / n_part"); return NPP_ERR_ARG; }
  return light_adam_go(L, d_params, d_m, d_v, d_grad, stride, n, C, d_pack, pack_stride, d_lat, d_lat_m, d_lat_v, d_dlat, d_zero, lr, beta1, beta2,
                       eps, step, d_part, n_part, d_loss_cur, stream);
}
static int light_adam_go(const npp_light_desc* L, float* d_params, float* d_m, float* d_v, float* d_grad, int64_t stride, int64_t n, int C,
                         float* d_pack, int64_t pack_stride, float* d_lat, float* d_lat_m, float* d_lat_v, float* d_dlat, float* d_zero,
                         float lr, float beta1, float beta2, float eps, int step, const float* d_part, int n_part, float* d_loss_cur, void* stream) {
  int rc = light_check(L, d_params, d_pack, C, 32, "npp_light_adam_pack");
  if (rc) return rc;
  const LightPackDesc pd = light_pack_desc();
  if (!d_m || !d_v || !d_grad || !d_lat || !d_lat_m || !d_lat_v || !d_dlat || n < 1 || n > stride || n > 0x7fffffffLL || step < 1 ||
      pack_stride < 4 * (int64_t)pd.total) {
    set_error("npp_light_adam_pack: bad argument");
    return NPP_ERR_ARG;
  }
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  LightAdamArgs a{};
  a.L = *L; a.p = d_params; a.m = d_m; a.v = d_v; a.g = d_grad; a.stride = stride; a.n = (int32_t)n;
  a.pack = d_pack; a.pack_stride = pack_stride;
  a.lat = d_lat; a.lat_m = d_lat_m; a.lat_v = d_lat_v; a.dlat = d_dlat; a.zero = d_zero;
  a.step_size = (float)((double)lr / bc1); a.b1 = beta1; a.b2 = beta2; a.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2)); a.eps = eps;
  a.part = d_part; a.n_part = n_part; a.loss_cur = d_loss_cur;
  hipLaunchKernelGGL(light_adam_pack_kernel, dim3((unsigned)((n + 255) / 256 + 1), (unsigned)C), dim3(256), 0, (hipStream_t)stream, a, pd);
  return check_launch("npp_light_adam_pack");
}
