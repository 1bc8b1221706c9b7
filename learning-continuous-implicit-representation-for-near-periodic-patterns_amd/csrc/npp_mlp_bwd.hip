// npp_mlp_bwd.hip -- K3a: fused backward (dgrad) chain of the coordinate MLP, bf16 MFMA.
//
// What autograd does through models/networks.py:56-95 / :145-173 and the sigmoid of
// models/helpers.py:56, restricted to the activations: given dL/dpred it produces the
// pre-activation gradients dz_l of every layer as bf16 fragments in the W-format line
// layout of npp_layout.h (what npp_mlp_wgrad copies into LDS and reads transposed).  No gradient flows to the embedding
// inputs, so L0 and the embedding columns of L5 / S have no dgrad.
//
// Same transposed formulation as the forward kernel: dA^T[k][b] = W^T[k][n] dZ^T[n][b]
// with W^T pre-packed as MFMA A-operand fragments (npp_pack_weights, backward pack) and
// the accumulator tile of one layer reused, after x snake'(z) and conversion to bf16, as
// the B operand of the next.  snake'(z) = 1 + sin 2z is derived from the fp16 pre-activation
// fragments the forward kernel stashed (one 16-byte load per lane per fragment).
#include "npp_common.h"

namespace npp {

constexpr int kThreadsB = 32 * kNT;          // two neuron tiles per wave: 4 waves at W = 256, 8 at W = 512
constexpr int kKSP = kKSAct / 2;               // k-steps of the W/2-wide P layer
constexpr int kRegionBytesB = kKSAct * kNB * 1024;
constexpr int kSmemBwd = 2 * kRegionBytesB + kRowTile * 3 * 4;

struct BwdArgs {
  const float* dpred;
  const float* pred;
  int64_t Bp;
  const bf16x8* wb;
  const float* params;
  const char* actF;        // forward stash: fp16 z fragments of the snake layers (W-format)
  char* dzF;
  int32_t out_act;         // output nonlinearity the forward applied: 0 raw, 1 sigmoid, 2 tanh
  // npp_mlp_bwd_patch: dL/dpred of the patch rows [pg_row0, pg_row0 + n_p P^2) is FORMED here from the patch losses' image
  // gradients (what npp_patch_compose_bwd computes, csrc/npp_patch.hip) instead of being read -- one launch less per
  // iteration; the rows are also written to dpred so the buffer stays what loss.backward() would have left there
  const float* pg_dxa;     // (n_p k, 3, P, P) or null = plain npp_mlp_bwd
  const float* pg_dxb;     // nullable second gradient (LPIPS / style branch)
  const float* pg_fmask;   // (n_p, P, P)
  const float* pg_rmask;   // (n_p k, P, P)
  float* dpred_out;
  int64_t pg_row0;
  int32_t pg_np, pg_k, pg_P, pg_comp;
  // stacked launch (npp_mlp_bwd_patch_stack): image m at + m * stride; k / comp / the position of its patches in the stacked
  // trunk batch come from S.iter[m]; pg_dxb (LPIPS branch) is per image (2 n_p kmax, 3, P, P) and read when iter[m].with_lp
  Stack S;
  int64_t wb_stride16, params_stride, act_stride, dz_stride;
  int64_t crop_stride, cmask_stride, dxb_stride;   // floats per image: rgb crops [fake | real], mask crops, LPIPS gradient
};

__device__ __forceinline__ void zero_acc(f32x16 (&acc)[2][kNB]) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int bt = 0; bt < kNB; ++bt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][bt][r] = 0.0f;
}

// The z fragments a layer's epilogue needs (stash, cold in HBM), fetched BEFORE the layer's MFMA loop so that their
// latency is covered by it instead of stalling every epilogue (32 VGPRs).
// S8 (stash8): the forward left snake'(z) itself, one unsigned byte per element in the W8-format (npp_common.h kSd8Scale): 8-byte
// units, no sine here
template <bool S8> struct ZPre { f16x8 z[2][kNB][2]; };
template <> struct ZPre<true> { u32x2 z[2][kNB][2]; };
template <bool S8>
__device__ __forceinline__ void z_prefetch(ZPre<S8>& zp, const char* z_array, int wg, int kt0, const Lane& L) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int bt = 0; bt < kNB; ++bt)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if constexpr (S8) zp.z[t][bt][s] = *(const u32x2*)(z_array + wfmt8_unit(kKSAct, wg, 2 * (kt0 + t) + s, 32 * bt + L.b, L.h));
        else zp.z[t][bt][s] = *(const f16x8*)(z_array + wfmt_unit(kKSAct, wg, 2 * (kt0 + t) + s, bt, L.b, L.h));
      }
}

// dz = acc (x stashed snake derivative); fragments -> LDS for the next dgrad, rows -> dzT.
template <bool HAS_S, bool S8>
__device__ __forceinline__ void bwd_epilogue(f32x16 (&acc)[2][kNB], char* out, const ZPre<S8>* zp, char* dz_array,
                                             int wg, int kt0, const Lane& L) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int ntg = kt0 + t;
#pragma unroll
    for (int bt = 0; bt < kNB; ++bt) {
      f32x16 g = acc[t][bt];
      if (HAS_S) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if constexpr (S8) {
            const u32x2 sd = zp->z[t][bt][s];
#pragma unroll
            for (int j = 0; j < 8; ++j) g[8 * s + j] *= u8_byte_f32(sd[j >> 2], j & 3) * kSd8Inv;
          } else {
            const f16x8 zf = zp->z[t][bt][s];
#pragma unroll
            for (int j = 0; j < 8; ++j)      // snake'(z) = 1 + sin 2z  (activations.py:29-35)
              g[8 * s + j] *= 1.0f + __builtin_amdgcn_sinf((float)zf[j] * (2.0f * kInv2Pi));
          }
        }
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 f = pack_acc(g, s);
        if (out) lds_store_frag(out, 2 * ntg + s, bt, L.lane, f);
        if (S8) stash8_store(dz_array + wfmt8_unit(kKSAct, wg, 2 * ntg + s, 32 * bt + L.b, L.h), pack8_bf8_acc(g, s));
        else dz_store(dz_array + wfmt_unit(kKSAct, wg, 2 * ntg + s, bt, L.b, L.h), f);
      }
    }
  }
}

// S8 (npp_tune "stash8"): the gradients leave as bf8 in the W8-format (npp_layout.h).  The whole chain is linear in dL/draw, so the
// workgroup runs it on dL/draw * 2^(kDz8Lift - e), e = floor(log2 max |dL/draw|) over its 64 rows: the stored bytes then sit in
// bf8's range whatever the loss normalisation is, and the E8M0 byte of 2^(e - kDz8Lift) goes to the tile's scale word, which
// npp_mlp_wgrad8 hands to the matrix instruction as the block scale of the tile's 64-row contraction step (no arithmetic anywhere).
template <bool MULTI, bool S8>
__global__ __launch_bounds__(kThreadsB, 2) void mlp_bwd_kernel(BwdArgs A_in, NetDesc d, BwdDesc bd) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (S8) set_fp16_ovfl();
  BwdArgs A = A_in;
  int img_, wg, xslot_, xcount_;
  if (!stack_decode(A.S, img_, wg, xslot_, xcount_)) return;
  const int n_wg_ = A.S.M ? A.S.n_items : (int)gridDim.x;
  if (A.S.M) {
    const StackIter it = A.S.iter[img_];
    A.dpred += (int64_t)img_ * A.Bp * 3;
    A.pred += (int64_t)img_ * A.Bp * 3;
    A.wb += (int64_t)img_ * A.wb_stride16;
    A.params += (int64_t)img_ * A.params_stride;
    A.actF += (int64_t)img_ * A.act_stride;
    A.dzF += (int64_t)img_ * A.dz_stride;
    if (A.pg_dxa) {
      const int64_t pp = (int64_t)A.pg_P * A.pg_P;
      A.dpred_out += (int64_t)img_ * A.Bp * 3;
      A.pg_dxa += (int64_t)it.x0 * 3 * pp;                              // this image's prediction half in the stacked batch
      A.pg_dxb = it.with_lp ? A.pg_dxb + (int64_t)img_ * A.dxb_stride : nullptr;
      A.pg_fmask += (int64_t)img_ * A.cmask_stride;                     // mask crops: [fake (n_p) | real (n_p kmax)]
      A.pg_rmask = it.same ? A.pg_fmask : A.pg_fmask + (int64_t)A.pg_np * pp;
      A.pg_k = it.k; A.pg_comp = it.comp;
    }
  }
  char* R0 = smem;
  char* R1 = smem + kRegionBytesB;
  float* sDraw = (float*)(smem + 2 * kRegionBytesB);   // [64 rows][3]

  Lane L;
  L.tid = threadIdx.x;
  L.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  L.lane = threadIdx.x & 63;
  L.b = L.lane & 31;
  L.h = L.lane >> 5;
  L.n_wg = n_wg_; L.xslot = xslot_; L.xcount = xcount_;
  const int64_t row0 = (int64_t)wg * kRowTile, Bp = A.Bp;
  const float* P = A.params;
  const int kt0 = 2 * L.wave;
  auto zs = [&](int idx) { return A.actF + (S8 ? wfmt8_array_base(idx * kKSAct, L.n_wg) : wfmt_array_base(idx * kKSAct, L.n_wg)); };   // z (S8: snake' bytes) of layer idx
  auto dzr = [&](int idx) { return A.dzF + (S8 ? wfmt8_array_base(idx * kKSAct, L.n_wg) : wfmt_array_base(idx * kKSAct, L.n_wg)); };

  // Everything the prologue needs from memory is requested first (weight ring of the first dgrad, the rgb weights, the cold
  // z fragments of P), so that ONE latency is paid instead of one per dependent section.
  // the backward pack is cold like the forward one (npp_mlp_fwd.hip, NPP_FWD_PREFETCH_LINES): the workgroups of each XCD
  // request it whole at entry (the loop also has the preceding trunk launch request it: npp_conv3x3_pf)
  uint32_t pfv[2];
  {
    const int64_t lines = ((int64_t)d.wb_total16 * 16 + 127) / 128;
    const int64_t per_xcd_threads = (int64_t)L.xcount * kThreadsB;
    int64_t line = (int64_t)L.xslot * kThreadsB + threadIdx.x;
#pragma unroll
    for (int q = 0; q < 2; ++q, line += per_xcd_threads)
      pfv[q] = line < lines ? *(const volatile uint32_t*)((const char*)A.wb + line * 128) : 0u;
  }
  WRing<2> ring;                                        // weight-stream ring, chained across layers
  ring.rsrc = make_wrsrc(A.wb, d.wb_total16);
  auto wbl = [&](int v) -> wptr_t { return (wptr_t)bd.off16[v]; };
  wring_fill<2, kNT>(ring, wbl(BP1), kt0, L.lane);
  f16x8 zp_pre[kNB][2];
  u32x2 sdp_pre[kNB][2];
  {
    const char* zp = A.actF + (S8 ? wfmt8_array_base(kActKsAP, L.n_wg) : wfmt_array_base(kActKsAP, L.n_wg));
#pragma unroll
    for (int bt = 0; bt < kNB; ++bt)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (S8) sdp_pre[bt][s] = *(const u32x2*)(zp + wfmt8_unit(kKSAct / 2, wg, 2 * L.wave + s, 32 * bt + L.b, L.h));
        else zp_pre[bt][s] = *(const f16x8*)(zp + wfmt_unit(kKSAct / 2, wg, 2 * L.wave + s, bt, L.b, L.h));
      }
  }
  float wr[3][16];
  {
    const float* Wr = P + d.w_off[LRGB];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) wr[c][r] = Wr[c * (kW / 2) + L.wave * 32 + acc_row(r, L.h)];
  }

  // ---- sigmoid backward (helpers.py:56): draw = dpred * pred * (1 - pred); also dz_rgb^T
  if (L.tid < kRowTile * 3) {
    const int row = L.tid / 3, c = L.tid - row * 3;
    const float pr = A.pred[(row0 + row) * 3 + c];
    // helpers.py:55-60: sigmoid (1), tanh (2) or the raw network output (0, npp_mlp_bwd_act only)
    const float dact = A.out_act == 1 ? pr * (1.0f - pr) : (A.out_act == 2 ? 1.0f - pr * pr : 1.0f);
    float dp;
    const int64_t prow = row0 + row - A.pg_row0;
    const int64_t pp = (int64_t)A.pg_P * A.pg_P;
    if (A.pg_dxa && prow >= 0 && prow < (int64_t)A.pg_np * pp) {       // train.py:200-236 backwards (npp_patch_compose_bwd's sum)
      const int p = (int)(prow / pp);
      const int64_t q = prow - (int64_t)p * pp;
      const float gm = A.pg_comp ? 1.0f - A.pg_fmask[prow] : 1.0f;
      dp = 0.0f;
      for (int kk = 0; kk < A.pg_k; ++kk) {
        const int64_t pk = (int64_t)p * A.pg_k + kk;
        float dd = A.pg_dxa[(pk * 3 + c) * pp + q];
        if (A.pg_dxb) dd += A.pg_dxb[(pk * 3 + c) * pp + q];
        dp = fmaf(dd, A.pg_rmask[pk * pp + q] * gm, dp);
      }
      A.dpred_out[(row0 + row) * 3 + c] = dp;
    } else {
      dp = A.dpred[(row0 + row) * 3 + c];
    }
    const float g = dp * dact;
    sDraw[L.tid] = g;
  }
  wg_barrier();
  float gs = 1.0f;                                       // 2^(kDz8Lift - e): what the chain is run on (S8)
  if (S8) {
    // every wave forms the tile's max |dL/draw| itself (192 values, three per lane): no second barrier
    float m = fmaxf(fmaxf(fabsf(sDraw[L.lane]), fabsf(sDraw[64 + L.lane])), fabsf(sDraw[128 + L.lane]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    int e = (int)((__float_as_uint(m) >> 23) & 255u) - 127;          // floor(log2 m) of a normal m; zero / subnormal: -127
    e = e < -100 ? -100 : (e > 100 ? 100 : e);                        // (NaN / infinity in dL/dpred: the run is lost anyway)
    gs = __uint_as_float((uint32_t)(127 + kDz8Lift - e) << 23);
    if (L.tid == 0) ((int32_t*)(A.dzF + dz8_scale_base(L.n_wg)))[wg] = 127 + e - kDz8Lift;
  }
  // dz_rgb as a 2-k-step W-format array: features 0..2 real, the rest zero
  {
    const int q1 = L.tid >> 7, bt = (L.tid >> 6) & 1, row = bt * 32 + L.b;        // waves 0..3: (k-step q1, batch tile bt)
    const bool real = q1 == 0 && L.h == 0;
    const float r0 = real ? sDraw[row * 3 + 0] * gs : 0.0f, r1 = real ? sDraw[row * 3 + 1] * gs : 0.0f, r2 = real ? sDraw[row * 3 + 2] * gs : 0.0f;
    if (S8) {
      if (q1 < 2) stash8_store(A.dzF + wfmt8_array_base(kDzKsRgb, L.n_wg) + wfmt8_unit(2, wg, q1, row, L.h), pack8_bf8(r0, r1, r2, 0.f, 0.f, 0.f, 0.f, 0.f));
    } else {
      bf16x8 f;
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = (__bf16)0.0f;
      f[0] = (__bf16)r0; f[1] = (__bf16)r1; f[2] = (__bf16)r2;
      if (q1 < 2) dz_store(A.dzF + wfmt_array_base(kDzKsRgb, L.n_wg) + wfmt_unit(2, wg, q1, bt, L.b, L.h), f);
    }
  }

  // ---- rgb_linear dgrad (VALU, 3 outputs) fused with the snake derivative of P:
  //      wave w owns P's neuron tile w.  dz_p fragments -> R0 (8 k-steps), rows -> dzT.
  {
#pragma unroll
    for (int bt = 0; bt < kNB; ++bt) {
      const int row = bt * 32 + L.b;
      const float g0 = sDraw[row * 3 + 0] * gs, g1 = sDraw[row * 3 + 1] * gs, g2 = sDraw[row * 3 + 2] * gs;
      f32x16 g;
#pragma unroll
      for (int r = 0; r < 16; ++r) g[r] = wr[0][r] * g0 + wr[1][r] * g1 + wr[2][r] * g2;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (S8) {
          const u32x2 sd = sdp_pre[bt][s];
#pragma unroll
          for (int j = 0; j < 8; ++j) g[8 * s + j] *= u8_byte_f32(sd[j >> 2], j & 3) * kSd8Inv;
        } else {
          const f16x8 zf = zp_pre[bt][s];
#pragma unroll
          for (int j = 0; j < 8; ++j) g[8 * s + j] *= 1.0f + __builtin_amdgcn_sinf((float)zf[j] * (2.0f * kInv2Pi));
        }
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 f = pack_acc(g, s);
        lds_store_frag(R0, 2 * L.wave + s, bt, L.lane, f);
        if (S8) stash8_store(A.dzF + wfmt8_array_base(kDzKsP, L.n_wg) + wfmt8_unit(kKSAct / 2, wg, 2 * L.wave + s, 32 * bt + L.b, L.h), pack8_bf8_acc(g, s));
        else dz_store(A.dzF + wfmt_array_base(kDzKsP, L.n_wg) + wfmt_unit(kKSAct / 2, wg, 2 * L.wave + s, bt, L.b, L.h), f);
      }
    }
  }
  wg_barrier();

  f32x16 acc[2][kNB], acc1[2][kNB];
  ZPre<S8> zpre;

  // ---- P dgrad: d[f1 ; f2] = W_P^T dz_p  (contraction over 128 neurons = 8 k-steps)
  zero_acc(acc1);
  mma_ring<0, kKSP, kKSP, kKSP, 2, kNT>(acc1, R0, 0, wbl(BP1), MULTI ? wbl(BP2) : wbl(BF1), kt0, L, ring);   // df1, part 1
  if (MULTI) {
    zero_acc(acc);
    mma_ring<0, kKSP, kKSP, kKSP, 2, kNT>(acc, R0, 0, wbl(BP2), wbl(BF2), kt0, L, ring);   // df2 = dz_f2 (F2 is linear)
    bwd_epilogue<false, S8>(acc, R1, nullptr, dzr(kDzF2), wg, kt0, L);
    wg_barrier();
    // ---- F2 dgrad -> x snake'(z_S) -> dz_s -> R0
    zero_acc(acc);
    z_prefetch(zpre, zs(kActAS), wg, kt0, L);
    mma_ring<0, kKSAct, kKSAct, kKSAct, 2, kNT>(acc, R1, 0, wbl(BF2), wbl(BS), kt0, L, ring);
    bwd_epilogue<true, S8>(acc, R0, &zpre, dzr(kDzS), wg, kt0, L);
    wg_barrier();
    // ---- S dgrad (f1 columns only; aux columns are raw embedding, no gradient) added
    //      onto the P part: df1 complete = dz_f1 (F1 is linear) -> R1
    mma_ring<0, kKSAct, kKSAct, kKSAct, 2, kNT>(acc1, R0, 0, wbl(BS), wbl(BF1), kt0, L, ring);
  }
  bwd_epilogue<false, S8>(acc1, R1, nullptr, dzr(kDzF1), wg, kt0, L);
  wg_barrier();

  // ---- F1, L7 .. L1 dgrads, ping-pong R1 -> R0 -> R1 ...; each output is multiplied by
  //      the snake derivative of the layer that produced that activation.
  //      virtual layer v: BF1 -> dz_7, B7 -> dz_6, ..., B1 -> dz_0
#pragma unroll
  for (int v = BF1; v <= B1; ++v) {
    const int out_layer = 7 - (v - BF1);            // L7 .. L0
    char* in = ((v - BF1) & 1) ? R0 : R1;
    char* out = ((v - BF1) & 1) ? R1 : R0;
    zero_acc(acc);
    z_prefetch(zpre, zs(out_layer), wg, kt0, L);
    mma_ring<0, kKSAct, kKSAct, kKSAct, 2, kNT>(acc, in, 0, wbl(v), v == B1 ? kNoW : wbl(v + 1), kt0, L, ring);
    bwd_epilogue<true, S8>(acc, v == B1 ? nullptr : out, &zpre, dzr(out_layer), wg, kt0, L);
    if (v != B1) wg_barrier();
  }
  asm volatile("" :: "v"(pfv[0]), "v"(pfv[1]));
}

}  // namespace npp

using namespace npp;

static int bwd_go(const BwdArgs& A, int K, void* stream);

static int bwd_launch(const float* d_dpred, const float* d_pred, int64_t Bp, int K, int width, const void* d_wb,
                      const float* d_params, const void* d_actT, void* d_dzT, int out_act, void* stream,
                      const npp_patch_grad* pg = nullptr) {
  if (K < 1 || K > NPP_MAX_K) { set_error("npp_mlp_bwd: K=%d", K); return NPP_ERR_ARG; }
  if (width != NPP_WIDTH) { set_error("npp_mlp_bwd: width %d unsupported (build is %d)", width, NPP_WIDTH); return NPP_ERR_UNSUPPORTED; }
  if (Bp <= 0 || Bp % kRowTile) { set_error("npp_mlp_bwd: Bp=%lld must be a positive multiple of %d", (long long)Bp, kRowTile); return NPP_ERR_ARG; }
  if (!d_dpred || !d_pred || !d_wb || !d_params || !d_actT || !d_dzT) { set_error("npp_mlp_bwd: null pointer"); return NPP_ERR_ARG; }
  if (out_act < 0 || out_act > 2) { set_error("npp_mlp_bwd: out_act=%d", out_act); return NPP_ERR_ARG; }
  BwdArgs A{d_dpred, d_pred, Bp, (const bf16x8*)d_wb, d_params, (const char*)d_actT, (char*)d_dzT, out_act};
  if (pg) {
    if (!pg->dx_a || !pg->fmask || !pg->rmask || pg->n_p < 1 || pg->k < 1 || pg->P < 1 || pg->row0 < 0 ||
        pg->row0 + (int64_t)pg->n_p * pg->P * pg->P > Bp) {
      set_error("npp_mlp_bwd_patch: bad patch-gradient description (n_p=%d k=%d P=%d row0=%lld)", pg->n_p, pg->k, pg->P, (long long)pg->row0);
      return NPP_ERR_ARG;
    }
    A.pg_dxa = pg->dx_a; A.pg_dxb = pg->dx_b; A.pg_fmask = pg->fmask; A.pg_rmask = pg->rmask;
    A.dpred_out = const_cast<float*>(d_dpred);
    A.pg_row0 = pg->row0; A.pg_np = pg->n_p; A.pg_k = pg->k; A.pg_P = pg->P; A.pg_comp = pg->comp;
  }
  return bwd_go(A, K, stream);
}

static int bwd_go(const BwdArgs& A, int K, void* stream) {
  const NetDesc d = make_desc(K);
  const BwdDesc bd = make_bwd_desc(K);
  const dim3 grid(A.S.M ? stack_grid(A.S) : (unsigned)(A.Bp / kRowTile)), block(kThreadsB);
  hipStream_t s = (hipStream_t)stream;
#define NPP_LAUNCH_B(M, S8)                                                                        \
  do {                                                                                             \
    static SmemOnce once;                                                                          \
    if (!smem_attr(once, (const void*)mlp_bwd_kernel<M, S8>, kSmemBwd)) {                          \
      set_error("npp_mlp_bwd: smem attribute"); return NPP_ERR_LAUNCH;                             \
    }                                                                                              \
    hipLaunchKernelGGL((mlp_bwd_kernel<M, S8>), grid, block, kSmemBwd, s, A, d, bd);               \
  } while (0)
  const bool s8 = __atomic_load_n(&g_tune.stash8, __ATOMIC_RELAXED) != 0;
  if (K > 1) { if (s8) NPP_LAUNCH_B(true, true); else NPP_LAUNCH_B(true, false); }
  else { if (s8) NPP_LAUNCH_B(false, true); else NPP_LAUNCH_B(false, false); }
#undef NPP_LAUNCH_B
  return check_launch("npp_mlp_bwd");
}

extern "C" int npp_mlp_bwd(const float* d_dpred, const float* d_pred, int64_t Bp, int K, int width, const void* d_wb,
                           const float* d_params, const void* d_actT, void* d_dzT, void* stream) {
  return bwd_launch(d_dpred, d_pred, Bp, K, width, d_wb, d_params, d_actT, d_dzT, 1, stream);
}

extern "C" int npp_mlp_bwd_act(const float* d_dout, const float* d_out, int64_t Bp, int K, int width, const void* d_wb,
                               const float* d_params, const void* d_actT, void* d_dzT, int out_act, void* stream) {
  return bwd_launch(d_dout, d_out, Bp, K, width, d_wb, d_params, d_actT, d_dzT, out_act, stream);
}

extern "C" int npp_mlp_bwd_patch(float* d_dpred, const float* d_pred, int64_t Bp, int K, int width, const void* d_wb,
                                 const float* d_params, const void* d_actT, void* d_dzT, const npp_patch_grad* pg, void* stream) {
  if (!pg) { set_error("npp_mlp_bwd_patch: null patch-gradient description"); return NPP_ERR_ARG; }
  return bwd_launch(d_dpred, d_pred, Bp, K, width, d_wb, d_params, d_actT, d_dzT, 1, stream, pg);
}

// npp_mlp_bwd_patch behind another output nonlinearity (out_act: 1 sigmoid = npp_mlp_bwd_patch, 2 tanh, 0 none)
extern "C" int npp_mlp_bwd_patch_act(float* d_dpred, const float* d_pred, int64_t Bp, int K, int width, const void* d_wb,
                                     const float* d_params, const void* d_actT, void* d_dzT, const npp_patch_grad* pg, int out_act,
                                     void* stream) {
  if (!pg) { set_error("npp_mlp_bwd_patch_act: null patch-gradient description"); return NPP_ERR_ARG; }
  return bwd_launch(d_dpred, d_pred, Bp, K, width, d_wb, d_params, d_actT, d_dzT, out_act, stream, pg);
}

// stacked form: M images per launch (npp_common.h "stacked launches"); always the patch form (npp_mlp_bwd_patch)
extern "C" int npp_mlp_bwd_patch_stack(float* d_dpred, const float* d_pred, int64_t Bp, int M, int K, int width, const void* d_wb,
                                       int64_t wb_stride_bytes, const float* d_params, int64_t params_stride, const void* d_actT,
                                       int64_t act_stride_bytes, void* d_dzT, int64_t dz_stride_bytes, const float* d_dx_a,
                                       const float* d_dx_b, int64_t dxb_stride, const float* d_cmask, int64_t cmask_stride,
                                       int64_t row0, int n_p, int P, const void* d_iter, void* stream) {
  if (K < 1 || K > NPP_MAX_K || M < 1 || M > NPP_MAX_STACK) { set_error("npp_mlp_bwd_patch_stack: K=%d M=%d", K, M); return NPP_ERR_ARG; }
  if (width != NPP_WIDTH) { set_error("npp_mlp_bwd_patch_stack: width %d unsupported (build is %d)", width, NPP_WIDTH); return NPP_ERR_UNSUPPORTED; }
  if (Bp <= 0 || Bp % kRowTile || !d_dpred || !d_pred || !d_wb || !d_params || !d_actT || !d_dzT || !d_dx_a || !d_cmask || !d_iter ||
      n_p < 1 || P < 1 || row0 < 0 || row0 + (int64_t)n_p * P * P > Bp || wb_stride_bytes % 16 || act_stride_bytes % 16 ||
      dz_stride_bytes % 16) {
    set_error("npp_mlp_bwd_patch_stack: bad arguments (Bp=%lld n_p=%d P=%d row0=%lld)", (long long)Bp, n_p, P, (long long)row0);
    return NPP_ERR_ARG;
  }
  BwdArgs A{d_dpred, d_pred, Bp, (const bf16x8*)d_wb, d_params, (const char*)d_actT, (char*)d_dzT, 1};
  A.pg_dxa = d_dx_a; A.pg_dxb = d_dx_b; A.pg_fmask = d_cmask; A.pg_rmask = d_cmask; A.dpred_out = d_dpred;
  A.pg_row0 = row0; A.pg_np = n_p; A.pg_k = 1; A.pg_P = P; A.pg_comp = 0;
  A.S = make_stack(M, (int)(Bp / kRowTile), d_iter);
  A.wb_stride16 = wb_stride_bytes / 16; A.params_stride = params_stride; A.act_stride = act_stride_bytes; A.dz_stride = dz_stride_bytes;
  A.cmask_stride = cmask_stride; A.dxb_stride = dxb_stride;
  return bwd_go(A, K, stream);
}
