// npp_layout.h -- index maps shared by host code and device kernels: the parameter
// blob (reference layout), the bf16 MFMA-fragment weight packs, and the training
// stash arrays.  Single source of truth; the CPU tests exercise it through
// npp_pack_weights_host() against a NumPy model of the MFMA lane maps.
//
// Network restated (models/networks.py:56-95, K>1; :145-173, K==1), W=256, D=8:
//   L0: emb0(462)->256   L1..L4: 256->256   L5: [emb0(462), h(256)]->256 (cat :71)
//   L6,L7: 256->256      F1 = feature_linear1 (no activation)
//   S  = scale_linears[0]: [f1(256), aux((K-1)*462)]->256 (cat :76)   (K>1)
//   F2 = feature_linear2 (no activation)                               (K>1)
//   P  = pos_linears[0]: [f1, f2](512)->128 (cat :85)  /  f1(256)->128 (K==1)
//   RGB = rgb_linear 128->3, then sigmoid (models/helpers.py:55-56)
// snake after L0..L7, S, P (models/activations.py:29-35).
#pragma once
#include <stdint.h>
#include "npp_hip.h"

#if defined(__HIPCC__)
#define NPP_HD __host__ __device__ inline
#else
#define NPP_HD inline
#endif

namespace npp {

constexpr int kW = NPP_WIDTH;       // 256
constexpr int kE = NPP_E;           // 462
constexpr int kNT = kW / 32;        // 8 neuron tiles of 32
constexpr int kKSAct = kW / 16;     // 16 k-steps per 256 activation features
constexpr int kKSEmb = 30;          // k-steps per proposal embedding (480 slots >= 462)
constexpr int kEmbSlots = kKSEmb * 16;
constexpr int kRowTile = NPP_ROW_TILE;  // 64 rows per workgroup = NB * 32
constexpr int kNB = kRowTile / 32;      // batch tiles per workgroup

enum Layer : int { L0 = 0, L1, L2, L3, L4, L5, L6, L7, LF1, LS, LF2, LP, LRGB, kNumLayers };

// ---- MFMA 32x32x16 bf16 lane maps (cdna_hip_programming.md section 3) ----------
// accumulator: lane l holds column (l & 31); register r holds row acc_row(r, l >> 5)
NPP_HD int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
// An accumulator tile converted to bf16 (registers 8s..8s+7) is the B operand of
// k-step s; element j of lane-half h is then row 16s + perm16(h, j).  All activation
// operands in this library use that order, so A-operand (weight) fragments are packed
// with the same k order.
NPP_HD int perm16(int h, int j) { return 8 * (j >> 2) + 4 * h + (j & 3); }
// inverse of perm16 on the 16 columns of a k-step: column c -> (hh, j)
NPP_HD int unperm_hh(int c) { return (c >> 2) & 1; }
NPP_HD int unperm_j(int c) { return ((c >> 3) << 2) | (c & 3); }

// ---- embedding slot order ------------------------------------------------------
// One proposal = 30 k-steps of 16 slots.  k-steps 0..27: t = 8*ks + j enumerates
// (freq fj = t / 22, warp index i = t % 22); lane-half 0 holds sin(f v_i), half 1
// holds cos(f v_i) (same argument, one range reduction).  k-step 28: v_0..v_15,
// k-step 29: v_16..v_21.  Returns the reference column in [0,462)
// (models/embedder.py:41-44,56: block 0 identity, 1+2f sin, 2+2f cos) or -1 (padding).
NPP_HD int emb_col(int ks, int h, int j) {
  if (ks < 28) {
    int t = 8 * ks + j;
    if (t >= 220) return -1;
    int fj = t / 22, i = t % 22;
    return (1 + 2 * fj + h) * 22 + i;
  }
  if (ks == 28) return 8 * h + j;
  return (h == 0 && j < 6) ? 16 + j : -1;
}

struct NetDesc {
  int32_t K;
  int32_t present[kNumLayers];
  int32_t n_out[kNumLayers], n_in[kNumLayers];       // reference shapes
  int64_t w_off[kNumLayers], b_off[kNumLayers];      // floats into the blob
  int64_t total_params;
  // forward pack: [ks][nt][64 lanes][8] bf16 per layer, offsets in 16-byte units
  int32_t ks_f[kNumLayers], nt_f[kNumLayers];
  int64_t wf_off[kNumLayers];
  int64_t wf_total16;
  // backward (dgrad) pack: virtual layers, each 8 output tiles x ns k-steps
  int64_t wb_total16;
};

// virtual dgrad layers, in the order the backward kernel runs them
enum BwdLayer : int { BP1 = 0, BP2, BF2, BS, BF1, B7, B6, B5, B4, B3, B2, B1, kNumBwd };
struct BwdDesc {
  int32_t present[kNumBwd];
  int32_t layer[kNumBwd];     // forward layer whose weight it transposes
  int32_t col0[kNumBwd];      // first reference input column of the 256 outputs
  int32_t ns[kNumBwd];        // k-steps over the layer's output neurons
  int64_t off16[kNumBwd];     // offset in 16-byte units
};

NPP_HD int64_t bwd_total16(int K);
NPP_HD NetDesc make_desc(int K) {
  NetDesc d{};
  d.K = K;
  const int multi = K > 1;
  int64_t off = 0, off16 = 0;
  for (int l = 0; l < kNumLayers; ++l) {
    int nout = kW, nin = kW, ks = kKSAct, nt = kNT, present = 1;
    switch (l) {
      case L0: nin = kE; ks = kKSEmb; break;
      case L5: nin = kE + kW; ks = kKSEmb + kKSAct; break;
      case LS: nin = kW + (K - 1) * kE; ks = kKSAct + (K - 1) * kKSEmb; present = multi; break;
      case LF2: present = multi; break;
      case LP: nout = kW / 2; nin = multi ? 2 * kW : kW; ks = multi ? 2 * kKSAct : kKSAct; nt = kNT / 2; break;
      case LRGB: nout = 3; nin = kW / 2; ks = 0; nt = 0; break;
      default: break;
    }
    d.present[l] = present;
    d.n_out[l] = nout; d.n_in[l] = nin; d.ks_f[l] = ks; d.nt_f[l] = nt;
    if (!present) { d.w_off[l] = d.b_off[l] = -1; d.wf_off[l] = -1; continue; }
    d.w_off[l] = off; off += (int64_t)nout * nin;
    d.b_off[l] = off; off += nout;
    d.wf_off[l] = off16; off16 += (int64_t)ks * nt * 64;
  }
  d.total_params = off;
  d.wf_total16 = off16;
  d.wb_total16 = bwd_total16(K);
  return d;
}

NPP_HD BwdDesc make_bwd_desc(int K) {
  BwdDesc b{};
  const int multi = K > 1;
  const int lay[kNumBwd] = {LP, LP, LF2, LS, LF1, L7, L6, L5, L4, L3, L2, L1};
  int64_t off = 0;
  for (int v = 0; v < kNumBwd; ++v) {
    b.layer[v] = lay[v];
    b.col0[v] = (v == BP2) ? kW : (v == B5 ? kE : 0);
    b.ns[v] = (v == BP1 || v == BP2) ? (kW / 2) / 16 : kKSAct;
    b.present[v] = (v == BP2 || v == BF2 || v == BS) ? multi : 1;
    b.off16[v] = off;
    if (b.present[v]) off += (int64_t)b.ns[v] * kNT * 64;
  }
  return b;
}
NPP_HD int64_t bwd_total16(int K) {
  BwdDesc b = make_bwd_desc(K);
  int64_t t = 0;
  for (int v = 0; v < kNumBwd; ++v) if (b.present[v]) t += (int64_t)b.ns[v] * kNT * 64;
  return t;
}

// Reference input column of forward-pack element (layer l, k-step ks, lane half h,
// element j), or -1 for zero padding.
NPP_HD int fwd_col(int K, int l, int ks, int h, int j) {
  switch (l) {
    case L0: return emb_col(ks, h, j);
    case L5: return ks < kKSEmb ? emb_col(ks, h, j) : kE + 16 * (ks - kKSEmb) + perm16(h, j);
    case LS: {
      if (ks < kKSAct) return 16 * ks + perm16(h, j);
      int q = ks - kKSAct, p = q / kKSEmb, c = emb_col(q % kKSEmb, h, j);
      return c < 0 ? -1 : kW + p * kE + c;
    }
    default: return 16 * ks + perm16(h, j);
  }
}

// ---- inverse of the two bf16 packs: where parameter W_l[n][k] lives (used by the fused Adam + re-pack launch, which walks
// the blob in parameter order and scatters every updated weight into both packs) --------------------------------------
// slot of reference embedding column c in [0, 462): (k-step, lane half, element), inverse of emb_col
NPP_HD void emb_slot(int c, int& ks, int& h, int& j) {
  const int blk = c / 22, i = c - blk * 22;
  if (blk == 0) {
    if (i < 16) { ks = 28; h = i >> 3; j = i & 7; } else { ks = 29; h = 0; j = i - 16; }
  } else {
    const int t = ((blk - 1) >> 1) * 22 + i;
    ks = t >> 3; h = (blk - 1) & 1; j = t & 7;
  }
}
// forward pack: index of the 16-bit element (16-byte unit * 8 + j) holding W_l[n][k]; -1: the layer has no forward pack (rgb)
NPP_HD int64_t fwd_pack_pos(const NetDesc& d, int l, int n, int k) {
  if (l == LRGB || !d.present[l]) return -1;
  int ks, h, j;
  bool emb = false;
  int ke = k, ks0 = 0;
  if (l == L0) { emb = true; }
  else if (l == L5) { if (k < kE) emb = true; else { ke = k - kE; ks0 = kKSEmb; } }
  else if (l == LS) {
    if (k >= kW) { const int q = k - kW, p = q / kE; emb = true; ke = q - p * kE; ks0 = kKSAct + p * kKSEmb; }
  }
  if (emb) { emb_slot(ke, ks, h, j); ks += ks0; }
  else { const int c16 = ke & 15; ks = ks0 + (ke >> 4); h = unperm_hh(c16); j = unperm_j(c16); }
  const int64_t unit = d.wf_off[l] + ((int64_t)ks * d.nt_f[l] + (n >> 5)) * 64 + (n & 31) + 32 * h;
  return unit * 8 + j;
}
// backward (transposed) pack: element index of W_l[n][k], or -1 when that weight has no data-gradient use
// (L0, the embedding columns of L5 / S, rgb)
NPP_HD int64_t bwd_pack_pos(const BwdDesc& b, int l, int n, int k) {
  int v = -1;
  switch (l) {
    case LP: v = k < kW ? BP1 : BP2; break;
    case LF2: v = BF2; break;
    case LS: v = k < kW ? BS : -1; break;
    case LF1: v = BF1; break;
    case L7: v = B7; break;
    case L6: v = B6; break;
    case L5: v = k >= kE ? B5 : -1; break;
    case L4: v = B4; break;
    case L3: v = B3; break;
    case L2: v = B2; break;
    case L1: v = B1; break;
    default: break;
  }
  if (v < 0 || !b.present[v]) return -1;
  const int kc = k - b.col0[v], c16 = n & 15;
  const int64_t unit = b.off16[v] + ((int64_t)(n >> 4) * kNT + (kc >> 5)) * 64 + (kc & 31) + 32 * unperm_hh(c16);
  return unit * 8 + unperm_j(c16);
}

// ---- exact-fp32 forward pack (npp_mlp_fwd32.hip, v_mfma_f32_32x32x2_f32) ---------------------------------
// A k-step contracts TWO input features (lane half h = 0 / 1).  Activation k-step s of a 256-feature input: features
// (8 (s / 4) + s % 4, that + 4) -- register r = s % 16 of accumulator tile s / 16 in both lane halves (acc_row).  Embedding
// k-step q of a proposal: q < 220 -> (sin, cos) block columns of (frequency q / 22, warped coordinate q % 22); 220..230 ->
// the raw block, two coordinates per k-step; 231 is padding (232 = 58 groups of 4).  One 16-byte pack unit carries the four
// k-steps of a GROUP for one lane: [layer][group][neuron tile][lane][4].
constexpr int kKSEmb32 = 232, kGE32 = kKSEmb32 / 4, kGA32 = kW / 8;
NPP_HD int emb32_col(int q, int h) {
  if (q < 220) return (1 + 2 * (q / 22) + h) * 22 + q % 22;
  const int i = 2 * (q - 220) + h;
  return (q < 231 && i < 22) ? i : -1;
}
NPP_HD int act32_col(int s, int h) { return 8 * (s >> 2) + (s & 3) + 4 * h; }
struct Desc32 {
  int32_t present[kNumLayers], nt[kNumLayers], groups[kNumLayers];
  int64_t off16[kNumLayers];
  int64_t total16;
};
NPP_HD Desc32 make_desc32(int K) {
  Desc32 d{};
  const int multi = K > 1;
  int64_t off = 0;
  for (int l = 0; l < kNumLayers; ++l) {
    int g = kGA32, nt = kNT, present = 1;
    switch (l) {
      case L0: g = kGE32; break;
      case L5: g = kGE32 + kGA32; break;
      case LS: g = kGA32 + (K - 1) * kGE32; present = multi; break;
      case LF2: present = multi; break;
      case LP: g = multi ? 2 * kGA32 : kGA32; nt = kNT / 2; break;
      case LRGB: g = 0; nt = 0; present = 0; break;
      default: break;
    }
    d.present[l] = present; d.nt[l] = nt; d.groups[l] = g;
    d.off16[l] = present ? off : -1;
    if (present) off += (int64_t)g * nt * 64;
  }
  d.total16 = off;
  return d;
}
// reference input column of fp32-pack element (layer l, k-step ks, lane half h), or -1 (zero weight)
NPP_HD int col32(int K, int l, int ks, int h) {
  switch (l) {
    case L0: return emb32_col(ks, h);
    case L5: return ks < kKSEmb32 ? emb32_col(ks, h) : kE + act32_col(ks - kKSEmb32, h);
    case LS: {
      if (ks < kW / 2) return act32_col(ks, h);
      const int q = ks - kW / 2, p = q / kKSEmb32, c = emb32_col(q % kKSEmb32, h);
      return c < 0 ? -1 : kW + p * kE + c;
    }
    case LP: return K > 1 ? (ks < kW / 2 ? kW + act32_col(ks, h) : act32_col(ks - kW / 2, h)) : act32_col(ks, h);   // [f2 | f1]
    default: return act32_col(ks, h);
  }
}

// ---- training stash for wgrad: "W-format" fragment arrays -----------------------------
// Every layer input (actF) and every pre-activation gradient (dzF) is stored as the very
// 16-byte fragments the fused kernels hold in registers: unit (k-step ks of 16 features,
// batch tile bt, row b, lane-half hh) = 8 bf16 = features 16 ks + perm16(hh, 0..7) of one
// row.  An array with NKS k-steps occupies NKS * 2 KiB per 64-row workgroup tile:
//   [wg][ks pair][bt][8 lines of 256 B], line = [ks & 1][hh][row & 3][16 B]
// so that (a) the two 16-byte stores a wave issues per accumulator tile fill whole 256-B
// lines, (b) a 128-feature operand tile of one workgroup tile is ONE contiguous 16-KiB
// chunk (linear copy into LDS), and (c) the transposed reads (ds_read_b64_tr_b16) by which
// npp_mlp_wgrad turns rows-of-features into the batch-contiguous MFMA operands hit 64
// distinct banks per 32-lane half.
NPP_HD int64_t wfmt_unit(int nks, int64_t wg, int ks, int bt, int b, int hh) {
  return ((((wg * (nks >> 1) + (ks >> 1)) * 2 + bt) * 8 + (b >> 2)) * 256) + (ks & 1) * 128 + hh * 64 + (b & 3) * 16;
}
// k-step offsets of the arrays inside actF / dzF (array base = offset * n_wg * 2 KiB)
constexpr int kActF1 = 8, kActAS = 9, kActF2 = 10;          // 256-wide arrays: index * 16
constexpr int kActKsAP = 11 * kKSAct;                        // a_p (128 wide, 8 k-steps)
constexpr int kActKsEmb0 = kActKsAP + kKSAct / 2;            // 184: proposal p at + 30 p
NPP_HD int act_total_ks(int K) { return kActKsEmb0 + K * kKSEmb; }
constexpr int kDzF1 = 8, kDzS = 9, kDzF2 = 10;
constexpr int kDzKsP = 11 * kKSAct;                          // dz_p (8 k-steps)
constexpr int kDzKsRgb = kDzKsP + kKSAct / 2;                // dz_rgb: 2 k-steps, 3 features used
constexpr int kDzTotalKs = kDzKsRgb + 2;
// split-K slabs of the weight gradient: slab s starts at s * slab_stride_of(total parameters) floats -- rounded up to a multiple
// of 4 so that every slab is 16-byte aligned (wgrad stores and Adam loads whole float4s; the parameter count itself is odd)
NPP_HD int64_t slab_stride_of(int64_t total) { return (total + 3) / 4 * 4; }
NPP_HD int64_t wfmt_array_base(int ks_off, int64_t n_wg) { return (int64_t)ks_off * n_wg * 2048; }

// ---- 8-bit training stash (round 6, npp_tune "stash8"): "W8-format" --------------------------------------------------------
// What npp_mlp_wgrad8 contracts: the pre-activation gradients as bf8 (e5m2) and the layer inputs (snake(z) of the snake layers, f1,
// f2, the embedding slots) as fp8 (e4m3, OCP), both operands of v_mfma_scale_f32_32x32x64_f8f6f4 (K = the 64 rows of a workgroup
// tile per instruction, twice the bf16 rate).  One unit = the same 8 features of one row as a W-format unit (features
// 16 ks + perm16(hh, 0..7)), now 8 bytes.  An array with NKS k-steps occupies NKS * 1 KiB per 64-row workgroup tile:
//   [wg][ks pair][row group of 8 rows (8)][line of 256 B], line = [ks & 1][hh][row & 7][8 B]
// so that (a) the (ks pair) chunk of a workgroup tile -- 32 features x 64 rows -- is ONE contiguous 2-KiB run (two 1-KiB LDS-DMA
// pieces), (b) a wave's 8-byte stores of one k-step fill 128 contiguous bytes of four lines, and (c) the transposing read
// ds_read_b64_tr_b8 (lane 2q + p of a 16-lane group addresses row q, lane-half chunk p; lane i receives, for e = 0..7, row e of
// byte i & 7 of chunk i >> 3: tools/micro/fp8_probe.hip) takes the 32 chunks of one line per 32-lane half: conflict-free.
// After that read lane l & 31 of a wave holds feature w8_feat(l & 31) of the 32-feature pair (bits 2 and 3 swapped: perm16 order
// inside the chunks), the same in both operands; the output maps of npp_mlp_wgrad8 undo it.
NPP_HD int64_t wfmt8_unit(int nks, int64_t wg, int ks, int row64, int hh) {
  return (((wg * (nks >> 1) + (ks >> 1)) * 8 + (row64 >> 3)) * 256) + (ks & 1) * 128 + hh * 64 + (row64 & 7) * 8;
}
NPP_HD int64_t wfmt8_array_base(int ks_off, int64_t n_wg) { return (int64_t)ks_off * n_wg * 1024; }
NPP_HD int w8_feat(int lane32) { return (lane32 & 0x13) | ((lane32 & 4) << 1) | ((lane32 & 8) >> 1); }
// actF in stash8 mode: the 16-bit W-format region keeps the fp16 pre-activations the backward chain reads (f1 / f2 / embedding
// arrays of that region stay unwritten); the 8-bit arrays follow it with the SAME k-step offsets (kActF1 ..., kActKsEmb0 + 30 p)
NPP_HD int64_t act8_region_base(int K, int64_t n_wg) { return wfmt_array_base(act_total_ks(K), n_wg); }
// dzF in stash8 mode: the bf8 arrays with the k-step offsets of the 16-bit layout, then one int32 per workgroup tile: the E8M0
// byte (127 + e) of the power of two 2^e that turns the stored values back into gradients (the backward chain runs on
// dL/dpred * 2^-e, e chosen per 64-row tile from max |dL/draw|: npp_mlp_bwd.hip)
NPP_HD int64_t dz8_scale_base(int64_t n_wg) { return wfmt8_array_base(kDzTotalKs, n_wg); }
constexpr int kDz8Lift = 11;      // stored = dz * 2^(kDz8Lift - floor(log2 max |draw| of the tile))
}  // namespace npp
