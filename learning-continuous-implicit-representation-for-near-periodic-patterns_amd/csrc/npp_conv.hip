// npp_conv.hip -- rows a11 / a13 (trunks): the frozen VGG19[0:18] (contextual loss,
// externel_lib/contextual_loss/modules/vgg.py:7-48) and VGG16 (LPIPS,
// externel_lib/lpips/pretrained_networks.py:96-134) convolution stacks, forward and
// data-gradient, as 16-bit-MFMA implicit GEMMs (fp32 accumulation).  Numeric contract: FORWARD operands
// (activations, weights) are fp16 -- 11 significand bits, the class of the TF32 convolutions cuDNN runs
// the reference's trunks with by default -- saturated at +-65504; GRADIENT operands are bf16 (range
// over precision: nothing here can under/overflow), ReLU gates are taken from the fp16 activations.
// The reference runs them through torchvision/cuDNN (F.conv2d 3x3 pad 1 + ReLU, MaxPool2d(2,2));
// weights are frozen (vgg.py:26-28), so only dL/dinput is ever needed.
//
// Layout ("flat padded"): an activation tensor (N, C, H, W) is stored as
//   [C/8 chunks][NPOSP positions][8 channels] fp16 | bf16,  16 bytes per (chunk, position) unit,
// where the position axis enumerates ALL images with a one-pixel zero border each,
// pos = kConvGuard + n*(H+2)*(W+2) + y*(W+2) + x  (y, x in [0,H+2) x [0,W+2), interior 1..H, 1..W),
// plus kConvGuard zero units on both ends.  Consequences:
//   * a 3x3 tap is a CONSTANT position shift (ky-1)*(W+2) + (kx-1): no boundary predicates,
//     the zero padding of conv2d(padding=1) is physically there;
//   * the MFMA B operand (16 channels x 32 positions) of one tap is two contiguous 512-byte
//     runs, fetched with ONE buffer_load_dwordx4 per lane straight from global memory / L2;
//   * the GEMM is computed transposed, Z^T[co][pos] = W[co][tap,ci] X^T[tap,ci][pos], weights
//     pre-packed as A-operand fragments (frozen: packed once), and an accumulator tile converted
//     to bf16 is stored as two 16-byte units per lane -- which makes the channel order inside a
//     chunk the perm16 order of npp_layout.h; the packers absorb that permutation.
// Border / tail positions are recomputed as zeros by every kernel, so a tensor is always a
// valid conv input.  The data-gradient of a conv is the same kernel with the flipped,
// transposed weight pack and a ReLU-mask epilogue (dZ = dY * [Y > 0]).
//
// Algorithmic work: 2 * 9 * Cin * Cout FLOP per interior output position (border and padded
// positions, and the 3 -> 16 channel padding of the first layer, are not counted).
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#include "npp_common.h"
#include "npp_trunk_layout.h"
#ifdef NPP_DIAG
#include "npp_diag.h"          // tools/npp_diag.h: diagnostic builds only (in-kernel time stamps)
#else
#define NPP_DIAG_FIELD
#define NPP_DIAG_FILL(a) do { } while (0)
#define NPP_STAMP(a, k) do { } while (0)
#define NPP_STAMP_DRAIN() do { } while (0)
#endif

namespace npp {

// k-steps of operand fragments in flight per wave (3 or 9: must divide the 9 taps).  9 = a whole input-channel step ahead.
// Timed alone in a loop (operands warm in L2) the depth makes no difference; INSIDE the iteration, where a layer's weights
// were last touched 0.7 ms / 1.5 GB of traffic ago and its input was just written from another XCD, depth 9 is worth
// 19 us per iteration (0.761 -> 0.742 ms, same-box A/B).
#ifndef NPP_CONV_RING
#define NPP_CONV_RING 9
#endif
constexpr int kConvRing = NPP_CONV_RING;
static_assert(kConvRing == 3 || kConvRing == 9, "ring depth");

enum ConvMode : int { kConvFwd = 0, kConvDgradMask = 1, kConvDgradLin = 2, kConvFwdPool = 3, kConvDgradPool = 4 };   // 3, 4: internal
// (npp_conv3x3_pool / npp_conv3x3_dgrad_pool: instantiations of their own, so that the folds' prefetch registers -- up to 256 VGPRs in the
// one-wave tiles -- are not carried by the plain modes)

struct ConvArgs {
  const void* x;         // flat input, Cin channels
  const void* pack;      // A fragments [cot][ci_step][tap][64 lanes][8]
  const float* bias;     // fwd only
  const void* mask;      // dgrad-mask: flat tensor of the layer output whose ReLU gates this gradient
  void* y;               // flat output (nullable when only the tap is wanted)
  float* tap;            // optional fp32 (N, Ctap, H, W) copy of the result (interior only)
  float tap_scale[4];
  int32_t N, H, W, Wp, S, CI, cout_chunks, Ctap, has_scale, pos_tiles;
  int64_t nposp, npos_valid;
  uint32_t x_bytes, pack_bytes;
  const char* pf;        // optional: the NEXT launch's weight pack, requested into this XCD's L2 (one dword per 128-byte line)
  int64_t pf_bytes;
  // dgrad-lin with the max-pool backward folded into the epilogue (npp_conv3x3_dgrad_pool): this launch's output is the gradient
  // of a POOLED tensor; instead of storing it, every interior position routes its value to the first maximum of its 2 x 2
  // window of the pre-pool activation pool_x (fp16, (2H, 2W) geometry), adds the optional tap gradient pool_add and applies the
  // pre-pool ReLU gate -- what maxpool2_bwd_kernel does in a launch of its own, bit for bit (the value is rounded to bf16 first).
  const void* pool_x;
  const void* pool_add;
  void* pool_dz;
  int64_t pool_nposp;
  // forward with MaxPool2d(2,2) folded in (kConvFwdPool, npp_conv3x3_pool): a position tile is 16 columns x 2 rows of one image
  // (lane b: row 2 rp + 1 + (b >> 4), column 16 cb + 1 + (b & 15)), so a pool window is lanes {b, b^1, b^16, b^17} of one tile;
  // pool_dz = the pooled fp16 tensor (geometry (H/2, W/2), pool_nposp units per chunk), pool_tiles = n_run * H/2 * pool_cbn.
  int32_t pool_tiles, pool_cbn;
  // weight-stationary numbering (conv3x3_kernel, 1-D grid): block -> XCD is linear id % 8; with the natural (position, channel-group)
  // grid every XCD runs every channel group and pulls the WHOLE weight pack into its L2 -- 8 x 4.7 MB from HBM for a 512 -> 512 layer
  // on a handful of position tiles (VGG16 conv4_x / conv5_x on the LPIPS branch's four patches: the launch is that traffic).  Here
  // channel group g lives on XCD g % 8: an XCD fetches one eighth of the pack and all (few) positions.
  int32_t wstat, wstat_npos, wstat_ncg;
  NPP_DIAG_FIELD
};

__device__ __forceinline__ f32x16 mfma16(const bf16x8& a, const bf16x8& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma16(const f16x8& a, const f16x8& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
template <bool F16> struct OpT { typedef bf16x8 frag; typedef __bf16 elem; };
template <> struct OpT<true> { typedef f16x8 frag; typedef _Float16 elem; };

// One workgroup = S waves that share ONE output tile (CT x PT MFMA tiles = 32 CT output channels x 32 PT
// positions) and split the contraction: wave w accumulates input-channel steps [w CI/S, (w+1) CI/S) x 9 taps,
// the partial tiles meet in LDS and each wave finishes (epilogue) a share of the tiles.  Why: the trunk layers of
// the loop are small (a 256 -> 256 layer on twelve 24 x 24 maps is 2032 MFMA tiles for 1024 SIMDs); with one small
// tile per wave every wave re-streams its whole weight slice (1.5 KiB of operands per MFMA through L1: measured
// 16 % of MFMA peak, L2 -> L1 bound), with big tiles there are too few waves.  Split-K keeps >= 256 workgroups
// with 2 x 4 tiles (0.75 KiB per MFMA) and every fragment is fetched by exactly one wave.
template <int CT, int PT, int S, int MODE>
__global__ __launch_bounds__(64 * S) void conv3x3_kernel(ConvArgs a) {
  constexpr bool FWD = MODE == kConvFwd || MODE == kConvFwdPool, FPOOL = MODE == kConvFwdPool;
  typedef typename OpT<FWD>::frag frag_t;                    // forward: fp16 operands; gradients: bf16
  typedef typename OpT<FWD>::elem elem_t;
  extern __shared__ __attribute__((aligned(16))) float red[];   // [S][CT*PT][16 regs][64 lanes] when S > 1
  NPP_STAMP(a, 0);
  NPP_STAMP(a, 1);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = lane & 31, h = lane >> 5;
  // XCD-aware: consecutive position blocks (which share halo rows) go to the same XCD / L2
  int bid = blockIdx.x, cgrp = blockIdx.y;
  const int nb = gridDim.x;
  if (a.wstat) {
    const int xcd = bid & 7, slot = bid >> 3;
    cgrp = xcd + 8 * (slot / a.wstat_npos);
    if (cgrp >= a.wstat_ncg) return;                      // (whole workgroup)
    bid = slot % a.wstat_npos;
  } else if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);
  // The next layer's weights were last read an iteration ago (1.1 GB of stash traffic since): its first workgroups would
  // take them from HBM.  Every XCD's workgroups of THIS launch together touch one dword per line of that pack; the value is
  // consumed at the very end (loads return in order, so it costs no extra wait).
  uint32_t pf_val = 0;
  if (a.pf) {
    const int64_t line = ((int64_t)((blockIdx.x >> 3) + blockIdx.y * ((gridDim.x + 7) >> 3)) * blockDim.x + threadIdx.x) * 128;
    if (line + 4 <= a.pf_bytes) pf_val = *(const volatile uint32_t*)(a.pf + line);
  }
  const int tile0 = bid * PT;
  const int cot0 = cgrp * CT;
  const int KS = a.CI * 9;
  const int ci_per = a.CI / S, ci_beg = wave * ci_per, ci_end = ci_beg + ci_per;

  const wrsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.pack), 0, (int)a.pack_bytes, 0x00020000);
  const wrsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  const int voffA = lane * 16;
  const int voffB = (int)(((int64_t)h * a.nposp + kConvGuard + (int64_t)tile0 * 32 + b - (a.Wp + 1)) * 16);
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  // flat position of my column in position tile pt of this workgroup; ok: it is a position this launch owns
  auto tile_pos = [&](int pt, bool& ok) -> int64_t {
    if (!FPOOL) { ok = true; return (int64_t)(tile0 + pt) * 32 + b; }
    const int T = tile0 + pt, per_img = (a.H >> 1) * a.pool_cbn;
    const int n = T / per_img, r = T - n * per_img, rp = r / a.pool_cbn, cb = r - rp * a.pool_cbn;
    const int row = 2 * rp + 1 + (b >> 4), col = 16 * cb + 1 + (b & 15);
    ok = T < a.pool_tiles && col <= a.W;
    return T < a.pool_tiles ? (int64_t)n * a.S + (int64_t)row * a.Wp + col : 0;
  };
  int voffBp[FPOOL ? PT : 1];
  if (FPOOL) {
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      bool ok;
      voffBp[pt] = (int)(((int64_t)h * a.nposp + kConvGuard + tile_pos(pt, ok) - (a.Wp + 1)) * 16);
    }
  }

  frag_t A[kConvRing][CT], B[kConvRing][PT];
  auto load = [&](int slot, int ci, int tap) {
    const int ks = ci * 9 + tap;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const u32x4_t raw = __builtin_amdgcn_raw_buffer_load_b128(rA, voffA, (int)((uint32_t)((cot0 + ct) * KS + ks) * 1024u), 0);
      A[slot][ct] = __builtin_bit_cast(frag_t, raw);
    }
    const int shift = (tap / 3) * a.Wp + (tap % 3);
    const uint32_t soff = (uint32_t)(((int64_t)2 * ci * a.nposp + shift) * 16);
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      const u32x4_t raw = FPOOL ? __builtin_amdgcn_raw_buffer_load_b128(rB, voffBp[FPOOL ? pt : 0], (int)soff, 0)
                                : __builtin_amdgcn_raw_buffer_load_b128(rB, voffB, (int)(soff + (uint32_t)pt * 512u), 0);
      B[slot][pt] = __builtin_bit_cast(frag_t, raw);
    }
  };

  f32x16 acc[CT][PT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ct][pt][r] = 0.0f;

  // Epilogue operands fetched BEFORE the contraction (their latency hides under it instead of stalling the short
  // epilogue of a 10-us kernel): the ReLU-gate units of the tiles this wave will finish (dgrad; written by the forward
  // pass long ago -> cold) and the bias values of its output channels (forward).
  constexpr int NT = CT * PT;
  constexpr int NF = S == 1 ? NT : (NT + S - 1) / S;          // tiles finished by one wave
  f16x8 gate[MODE == kConvDgradMask ? NF : 1][2];
  float bias_r[FWD ? NF : 1][16];
  constexpr bool kPoolable = MODE == kConvDgradPool;
  f16x8 pwin[kPoolable ? NF : 1][2][4];                    // folded pool backward: the four pre-pool units of my window per chunk
  bf16x8 padd[kPoolable ? NF : 1][2][4];
  constexpr bool pool_fold = kPoolable;
#pragma unroll
  for (int q = 0; q < NF; ++q) {
    const int t = S == 1 ? q : wave + q * S;
    if (t < NT) {
      const int ct = t / PT, pt = t % PT;
      if (MODE == kConvDgradMask) {
        const int64_t p = (int64_t)(tile0 + pt) * 32 + b;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int chunk = 4 * (cot0 + ct) + 2 * s + h;
          if (chunk < a.cout_chunks) gate[q][s] = ((const f16x8*)a.mask)[(int64_t)chunk * a.nposp + kConvGuard + p];
        }
      }
      if (FWD) {
#pragma unroll
        for (int r = 0; r < 16; ++r) bias_r[q][r] = a.bias[32 * (cot0 + ct) + acc_row(r, h)];
      }
      if (kPoolable && pool_fold) {
        const int64_t p = (int64_t)(tile0 + pt) * 32 + b;
        const int n = (int)((uint32_t)p / (uint32_t)a.S);
        const int r0 = (int)(p - (int64_t)n * a.S);
        const int yy = r0 / a.Wp, xx = r0 - yy * a.Wp;
        const bool interior = p < a.npos_valid && yy >= 1 && yy <= a.H && xx >= 1 && xx <= a.W;
        const int Wf = 2 * a.W + 2;
        const int64_t qpos = (int64_t)n * (2 * a.H + 2) * Wf + (int64_t)(2 * yy - 1) * Wf + (2 * xx - 1);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int chunk = 4 * (cot0 + ct) + 2 * s + h;
          if (interior && chunk < a.cout_chunks) {
            const int64_t u0 = (int64_t)chunk * a.pool_nposp + kConvGuard + qpos;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int64_t u = u0 + (k >> 1) * Wf + (k & 1);
              pwin[q][s][k] = ((const f16x8*)a.pool_x)[u];
              if (a.pool_add) padd[q][s][k] = ((const bf16x8*)a.pool_add)[u];
            }
          }
        }
      }
    }
  }

#pragma unroll
  for (int q = 0; q < kConvRing; ++q) load(q, ci_beg, q);
  NPP_STAMP(a, 2);
  for (int ci = ci_beg; ci < ci_end; ++ci) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int slot = tap % kConvRing;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = mfma16(A[slot][ct], B[slot][pt], acc[ct][pt]);
      if (tap + kConvRing < 9) load(slot, ci, tap + kConvRing);
      else if (ci + 1 < ci_end) load(slot, ci + 1, tap + kConvRing - 9);
      asm volatile("" ::: "memory");
    }
  }

  NPP_STAMP(a, 3);
  // ---- epilogue of one 32 x 32 tile (ct, pt): bias / ReLU / gate, 16-bit store, optional fp32 tap -------
  auto finish = [&](const f32x16& v16, int ct, int pt, int q) {
    bool mine;
    const int64_t p = tile_pos(pt, mine);
    const int n = (int)((uint32_t)p / (uint32_t)a.S);
    const int r0 = (int)(p - (int64_t)n * a.S);
    const int yy = r0 / a.Wp, xx = r0 - yy * a.Wp;
    const bool interior = mine && p < a.npos_valid && yy >= 1 && yy <= a.H && xx >= 1 && xx <= a.W;
    const int64_t tap_base = (((int64_t)n * a.Ctap) * a.H + (yy - 1)) * a.W + (xx - 1);
    const int cot = cot0 + ct;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int chunk = 4 * cot + 2 * s + h;
      if (chunk >= a.cout_chunks) continue;
      const int64_t unit = (int64_t)chunk * a.nposp + kConvGuard + p;
      if (kPoolable && pool_fold) {
        if (!interior) continue;                              // the pre-pool tensor's border stays as it is: zero
        const int Wf = 2 * a.W + 2;
        const int64_t u0 = (int64_t)chunk * a.pool_nposp + kConvGuard + (int64_t)n * (2 * a.H + 2) * Wf +
                           (int64_t)(2 * yy - 1) * Wf + (2 * xx - 1);
        bf16x8 o4[4];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float d = (float)(__bf16)v16[8 * s + j];      // what the separate launch reads back from the pooled gradient tensor
          const float v0 = (float)pwin[q][s][0][j], v1 = (float)pwin[q][s][1][j], v2 = (float)pwin[q][s][2][j],
                      v3 = (float)pwin[q][s][3][j];
          int am = 0;
          float mx = v0;
          if (v1 > mx) { mx = v1; am = 1; }
          if (v2 > mx) { mx = v2; am = 2; }
          if (v3 > mx) { mx = v3; am = 3; }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            float g = am == k ? d : 0.0f;
            if (a.pool_add) g += (float)padd[q][s][k][j];
            o4[k][j] = (__bf16)((float)pwin[q][s][k][j] > 0.0f ? g : 0.0f);
          }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) ((bf16x8*)a.pool_dz)[u0 + (k >> 1) * Wf + (k & 1)] = o4[k];
        continue;
      }
      f16x8 m;
      if (MODE == kConvDgradMask) m = gate[q][s];
      frag_t o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int co = 32 * cot + acc_row(8 * s + j, h);
        float v = v16[8 * s + j];
        if (FWD) v = fminf(fmaxf(v + bias_r[q][8 * s + j], 0.0f), 65504.0f);
        if (MODE == kConvDgradMask) v = (float)m[j] > 0.0f ? v : 0.0f;
        v = interior ? v : 0.0f;
        o[j] = (elem_t)v;
        if (a.tap && interior && co < a.Ctap)
          a.tap[tap_base + (int64_t)co * a.H * a.W] = a.has_scale ? v * a.tap_scale[co & 3] : v;
      }
      if (a.y && (!FPOOL || interior)) ((frag_t*)a.y)[unit] = o;     // (two-row tiles own interior positions only: borders stay zero)
      if (FPOOL) {
        // the 2 x 2 window's maximum: lanes b, b ^ 1 (next column), b ^ 16 (next row); values are post-ReLU fp16 (>= 0, columns
        // beyond W hold 0), so the maximum of the rounded values is the rounded maximum: what maxpool2_fwd_kernel stores
        typedef int i32x4_t __attribute__((ext_vector_type(4)));
        i32x4_t w = __builtin_bit_cast(i32x4_t, o);
#pragma unroll
        for (int step = 0; step < 2; ++step) {
          i32x4_t t;
#pragma unroll
          for (int e = 0; e < 4; ++e) t[e] = __shfl_xor(w[e], step == 0 ? 1 : 16, 64);
          const f16x8 mine8 = __builtin_bit_cast(f16x8, w), other8 = __builtin_bit_cast(f16x8, t);
          f16x8 mx;
#pragma unroll
          for (int j = 0; j < 8; ++j) mx[j] = other8[j] > mine8[j] ? other8[j] : mine8[j];
          w = __builtin_bit_cast(i32x4_t, mx);
        }
        if (interior && (b & 17) == 0) {
          const int Ho = a.H >> 1, Wo = a.W >> 1;
          const int64_t up = (int64_t)chunk * a.pool_nposp + kConvGuard + (int64_t)n * (Ho + 2) * (Wo + 2) +
                             (int64_t)((yy + 1) >> 1) * (Wo + 2) + ((xx + 1) >> 1);
          ((f16x8*)a.pool_dz)[up] = __builtin_bit_cast(f16x8, w);
        }
      }
    }
  };

  if (S == 1) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) finish(acc[ct][pt], ct, pt, ct * PT + pt);
  } else {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int pt = 0; pt < PT; ++pt)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wave * NT + ct * PT + pt) * 16 + r) * 64 + lane] = acc[ct][pt][r];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NF; ++q) {                        // tile t is finished by wave t % S
      const int t = wave + q * S;
      if (t >= NT) break;
      f32x16 v;
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = 0.0f;
      for (int w2 = 0; w2 < S; ++w2)
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] += red[((w2 * NT + t) * 16 + r) * 64 + lane];
      finish(v, t / PT, t % PT, q);
    }
  }
  NPP_STAMP(a, 4);
  NPP_STAMP_DRAIN();
  NPP_STAMP(a, 5);
  asm volatile("" :: "v"(pf_val));
}

// ---- window-staged form for the position-rich layers (round 4) -----------------------------------------------
// conv1_1 / conv1_2 / conv2_x see 30-115 k positions and only 1-8 input-channel steps: with operands straight from L2 every
// MFMA pulls 1 KiB through L1 (0.75 at best), ~260 MB per layer at the ~70 GB/s a CU draws, i.e. the layer is bound by L2 -> CU
// bandwidth at 4-5 x its MFMA time.  Here a workgroup of 4 waves owns CT x 32 output channels x 256 consecutive flat positions
// and per input-channel step stages ONCE, through registers into LDS (global_load -> ds_write_b128, double buffered):
//   * the step's CT x 9 weight fragments (CT x 9 KiB, shared by the 4 waves), and
//   * ONE input window [p0 - (Wp + 1), p0 + 256 + (Wp + 1)) of the step's two 8-channel chunks: the 9 taps are 9 shifted
//     ds_read_b128 of that window (a tap is a constant position shift in the flat layout) instead of 9 global fetches.
// Global bytes per MFMA: (CT x 9 KiB + 2 x (256 + 2 Wp + 2) x 16 B) / (72 CT) = 0.23 KiB at CT = 2, Wp = 98 (1.0 before).
// LDS per step and workgroup: 33 KiB written, 144 KiB read (256 B/clk) against 1152 cycles of MFMA per SIMD: not array-bound.
// Round 3 built this with LDS-DMA and measured 2 x SLOWER (LDS-DMA lands ~16 GB/s per CU here); the register path was only argued
// about.  Measured: see DESIGN.md section 7 / profiles/r04_conv_window_ab.txt.
constexpr int kWinPos = 256;                       // positions per workgroup (8 position tiles: 2 per wave)
#ifndef NPP_CONV_WIN_DEPTH
#define NPP_CONV_WIN_DEPTH 1       // 2 measured: no change (profiles/r04_conv_window_ab.txt 4.)
#endif
constexpr int kWinDepth = NPP_CONV_WIN_DEPTH;      // register stages of the operand prefetch (1 or 2)
static_assert(kWinDepth == 1 || kWinDepth == 2, "window prefetch depth");
constexpr int kWinMaxUnits = 640;                  // window units per chunk the register staging is sized for: Wp <= 191
template <int CT>
constexpr int win_lds_bytes() { return 2 * (CT * 9 * 1024 + 2 * kWinMaxUnits * 16); }

// (Round 4 also had a form that pooled its INPUT while staging the window -- pool1 inside conv2_1 --: bit-identical, trunk forward
//  130.5 -> 133.8 us, removed in round 5; pool1 now rides in the fused pair of npp_conv_pair.hip.)
template <int CT, int MODE>
__global__ __launch_bounds__(256, 2) void conv3x3_win_kernel(ConvArgs a) {
  typedef typename OpT<MODE == kConvFwd>::frag frag_t;
  typedef typename OpT<MODE == kConvFwd>::elem elem_t;
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) char wlds[];
  NPP_STAMP(a, 0);
  NPP_STAMP(a, 1);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = lane & 31, h = lane >> 5;
  int bid = blockIdx.x;
  const int nb = gridDim.x;
  if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);      // consecutive position blocks (shared halo rows) on one XCD
  uint32_t pf_val = 0;
  if (a.pf) {
    const int64_t line = ((int64_t)((blockIdx.x >> 3) + blockIdx.y * ((gridDim.x + 7) >> 3)) * blockDim.x + threadIdx.x) * 128;
    if (line + 4 <= a.pf_bytes) pf_val = *(const volatile uint32_t*)(a.pf + line);
  }
  const int tile0 = bid * (kWinPos / 32);
  const int cot0 = blockIdx.y * CT;
  const int KS = a.CI * 9;
  const int halo = a.Wp + 1, WIN = kWinPos + 2 * halo;
  constexpr int kA = CT * 9 * 1024;                               // bytes of weight fragments per step
  constexpr int kBuf = kA + 2 * kWinMaxUnits * 16;
  constexpr int NA = (CT * 9 * 64 + 255) / 256;                    // 16-byte units per thread: weights
  constexpr int NW = (2 * kWinMaxUnits + 255) / 256;               // ... window (5)
  const wrsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.pack), 0, (int)a.pack_bytes, 0x00020000);
  const wrsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  // per-thread source offsets (bytes) of its units inside a step; -1 = no unit
  int offA[NA], offW[NW], dstW[NW];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int u = tid + 256 * i;                                   // unit (ct, tap, lane) of the step
    const int ct = u / 576, r = u - ct * 576;
    offA[i] = u < CT * 576 ? ((cot0 + ct) * KS) * 1024 + r * 16 : -1;
  }
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    const int u = tid + 256 * i;
    const int chunk = u >= WIN, pos = u - chunk * WIN;
    offW[i] = u < 2 * WIN ? (int)((((int64_t)chunk * a.nposp) + kConvGuard + (int64_t)tile0 * 32 - halo + pos) * 16) : -1;
    dstW[i] = kA + (chunk * kWinMaxUnits + pos) * 16;
  }
  // Register stages of the operand prefetch (NPP_CONV_WIN_DEPTH): with 2 the operands of step ci + 2 are requested while step ci is
  // multiplied.  Built on the hypothesis that a step lasts as long as its loads (the input was just written by the previous layer from
  // other XCDs); measured in the iteration, same box: trunk forward 132.1 us with one stage, 132.2 with two -- not the bound.
  struct Stage { u32x4_t a[NA], w[NW]; };
  Stage st[kWinDepth];
  auto gload = [&](int ci, Stage& r) {
#pragma unroll
    for (int i = 0; i < NA; ++i)
      if (offA[i] >= 0 || NA * 256 == CT * 576) r.a[i] = __builtin_amdgcn_raw_buffer_load_b128(rA, offA[i] < 0 ? 0 : offA[i], ci * 9 * 1024, 0);
    const uint32_t soff = (uint32_t)((int64_t)2 * ci * a.nposp * 16);
#pragma unroll
    for (int i = 0; i < NW; ++i) r.w[i] = __builtin_amdgcn_raw_buffer_load_b128(rB, offW[i] < 0 ? 0 : offW[i], (int)soff, 0);
  };
  auto sstore = [&](int buf, const Stage& r) {
    char* base = wlds + buf * kBuf;
#pragma unroll
    for (int i = 0; i < NA; ++i)
      if (offA[i] >= 0) *(u32x4_t*)(base + (tid + 256 * i) * 16) = r.a[i];
#pragma unroll
    for (int i = 0; i < NW; ++i)
      if (offW[i] >= 0) *(u32x4_t*)(base + dstW[i]) = r.w[i];
  };
  f32x16 acc[CT][2];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int pt = 0; pt < 2; ++pt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ct][pt][r] = 0.0f;
  // epilogue operands requested before the contraction (ReLU-gate units / bias values), like conv3x3_kernel
  f16x8 gate[MODE == kConvDgradMask ? CT * 2 : 1][2];
  float bias_r[MODE == kConvFwd ? CT : 1][16];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    if (MODE == kConvFwd) {
#pragma unroll
      for (int r = 0; r < 16; ++r) bias_r[ct][r] = a.bias[32 * (cot0 + ct) + acc_row(r, h)];
    }
    if (MODE == kConvDgradMask) {
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) {
        const int64_t p = (int64_t)(tile0 + 2 * wave + pt) * 32 + b;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int chunk = 4 * (cot0 + ct) + 2 * s + h;
          if (chunk < a.cout_chunks) gate[ct * 2 + pt][s] = ((const f16x8*)a.mask)[(int64_t)chunk * a.nposp + kConvGuard + p];
        }
      }
    }
  }
  gload(0, st[0]);
  if (kWinDepth > 1 && a.CI > 1) gload(1, st[kWinDepth - 1]);
  sstore(0, st[0]);
  __syncthreads();
  NPP_STAMP(a, 2);
  int buf = 0;
  const int posw = (kWinPos / 4) * wave + b;                        // this lane's first position inside the workgroup's 256
  // one step; PAR = ci % kWinDepth at compile time (the stage of an in-flight load must be a compile-time name)
  auto step = [&](int ci, auto par_) {
    constexpr int PAR = decltype(par_)::value, NXT = (PAR + 1) % kWinDepth;
    const bool has_next = ci + 1 < a.CI;
    if (kWinDepth == 1) { if (has_next) gload(ci + 1, st[0]); }
    else if (ci + 2 < a.CI) gload(ci + 2, st[PAR]);                 // stage PAR went to LDS before this step began
    const char* bA = wlds + buf * kBuf;
    const char* bW = bA + kA + h * kWinMaxUnits * 16;
    // (round 5) tap + 1's fragments are requested before tap's MFMAs: see conv3x3_wink_kernel
    frag_t A[2][CT], B[2][2];
    auto lread = [&](int set, int tap) {
      const int shift = (tap / 3) * a.Wp + (tap % 3);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) A[set][ct] = *(const frag_t*)(bA + ((ct * 9 + tap) * 64 + lane) * 16);
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) B[set][pt] = *(const frag_t*)(bW + (posw + 32 * pt + shift) * 16);
    };
    lread(0, 0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (tap + 1 < 9) lread((tap + 1) & 1, tap + 1);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) acc[ct][pt] = mfma16(A[tap & 1][ct], B[tap & 1][pt], acc[ct][pt]);
    }
    if (has_next) sstore(buf ^ 1, st[NXT]);
    __syncthreads();
    buf ^= 1;
  };
  for (int ci = 0; ci < a.CI; ci += kWinDepth) {
    step(ci, std::integral_constant<int, 0>{});
    if (kWinDepth > 1 && ci + 1 < a.CI) step(ci + 1, std::integral_constant<int, kWinDepth - 1>{});
  }
  NPP_STAMP(a, 3);
  // ---- epilogue: the same per-tile finish as conv3x3_kernel
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      const int64_t p = (int64_t)(tile0 + 2 * wave + pt) * 32 + b;
      const int n = (int)((uint32_t)p / (uint32_t)a.S);
      const int r0 = (int)(p - (int64_t)n * a.S);
      const int yy = r0 / a.Wp, xx = r0 - yy * a.Wp;
      const bool interior = p < a.npos_valid && yy >= 1 && yy <= a.H && xx >= 1 && xx <= a.W;
      const int64_t tap_base = (((int64_t)n * a.Ctap) * a.H + (yy - 1)) * a.W + (xx - 1);
      const int cot = cot0 + ct;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int chunk = 4 * cot + 2 * s + h;
        if (chunk >= a.cout_chunks) continue;
        const int64_t unit = (int64_t)chunk * a.nposp + kConvGuard + p;
        frag_t o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int co = 32 * cot + acc_row(8 * s + j, h);
          float v = acc[ct][pt][8 * s + j];
          if (MODE == kConvFwd) v = fminf(fmaxf(v + bias_r[ct][8 * s + j], 0.0f), 65504.0f);
          if (MODE == kConvDgradMask) v = (float)gate[ct * 2 + pt][s][j] > 0.0f ? v : 0.0f;
          v = interior ? v : 0.0f;
          o[j] = (elem_t)v;
          if (a.tap && interior && co < a.Ctap)
            a.tap[tap_base + (int64_t)co * a.H * a.W] = a.has_scale ? v * a.tap_scale[co & 3] : v;
        }
        if (a.y) ((frag_t*)a.y)[unit] = o;
      }
    }
  NPP_STAMP(a, 4);
  NPP_STAMP_DRAIN();
  NPP_STAMP(a, 5);
  asm volatile("" :: "v"(pf_val));
}

// ---- window-staged form with the contraction split over wave GROUPS (round 5) -------------------------------------
// What bounds the channel-rich layers of the loop (relu3_x: 256 -> 256 on 8 k positions, conv2_2) in conv3x3_kernel is the operand
// stream of a CU: a 2 x 2-tile workgroup pulls 64 couts x 2304 k x 2 B = 295 KB of weights AND 9 taps x 64 positions x 256 channels
// x 2 B = 295 KB of activations through the vector memory pipe for 144 MFMAs per wave -- two such workgroups per CU are 1.18 MB at
// the ~64 B/clk a CU draws = 9 us of a 17-us launch, against 4.6 us of matrix time.  The window form (one input window per
// channel step in LDS, the nine taps as shifted LDS reads) cuts the activation half 6 x, and a workgroup that owns 128 / 256
// positions instead of 64 halves / quarters the weight half per MFMA -- but with 4 waves per 256 positions the deep layers gave
// only 128 workgroups of one wave per SIMD (round 4: slower).  Here a workgroup is KG groups of four waves: group g contracts the
// channel steps ci = g (mod KG) from its OWN double-buffered LDS stage (weights of the step + the step's window, staged by the
// group's 256 threads), all groups cover the same CT x 32 output channels x 128 NPT positions, and the partial tiles meet in LDS at
// the end (fixed group order: deterministic).  relu3_x: 256 workgroups of 8 waves, one per CU, 388 KB of operands each.
constexpr int kWinkMaxWp = 98;                      // widest padded row the staging registers are sized for (96-pixel maps)
template <int CT, int NPT, int KG, int MODE>
__global__ __launch_bounds__(256 * KG) void conv3x3_wink_kernel(ConvArgs a, int wu /* window units per chunk in LDS (>= WIN) */) {
  constexpr bool FWD = MODE == kConvFwd;
  typedef typename OpT<FWD>::frag frag_t;
  typedef typename OpT<FWD>::elem elem_t;
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) char klds[];
  NPP_STAMP(a, 0);
  NPP_STAMP(a, 1);
  constexpr int WP = 128 * NPT;                                      // positions of a workgroup
  constexpr int NT = CT * NPT;                                       // output tiles of a wave
  static_assert(NT % KG == 0, "tiles must split over the groups");
  constexpr int NF = NT / KG;                                        // tiles a wave finishes
  const int tid = threadIdx.x, lane = tid & 63, tg = tid & 255;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pw = wave & 3, kg = wave >> 2;
  const int b = lane & 31, h = lane >> 5;
  int bid = blockIdx.x;
  const int nb = gridDim.x;
  if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);      // consecutive position blocks (shared halo rows) on one XCD
  uint32_t pf_val = 0;
  if (a.pf) {
    const int64_t line = ((int64_t)((blockIdx.x >> 3) + blockIdx.y * ((gridDim.x + 7) >> 3)) * blockDim.x + threadIdx.x) * 128;
    if (line + 4 <= a.pf_bytes) pf_val = *(const volatile uint32_t*)(a.pf + line);
  }
  const int tile0 = bid * (WP / 32);
  const int cot0 = blockIdx.y * CT;
  const int KS = a.CI * 9;
  const int halo = a.Wp + 1, WIN = WP + 2 * halo;
  constexpr int kA = CT * 9 * 1024;                                 // bytes of weight fragments per channel step
  const int kBuf = kA + 2 * wu * 16;                                // one stage: weights + the two chunks' windows
  constexpr int NA = (CT * 9 * 64 + 255) / 256;
  constexpr int NW = (2 * (WP + 2 * (kWinkMaxWp + 1)) + 255) / 256;
  const wrsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.pack), 0, (int)a.pack_bytes, 0x00020000);
  const wrsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  int offA[NA], offW[NW], dstW[NW];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int u = tg + 256 * i;                                     // unit (ct, tap, lane) of the step
    const int ct = u / 576, r = u - ct * 576;
    offA[i] = u < CT * 576 ? ((cot0 + ct) * KS) * 1024 + r * 16 : -1;
  }
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    const int u = tg + 256 * i;
    const int chunk = u >= WIN, pos = u - chunk * WIN;
    offW[i] = u < 2 * WIN ? (int)((((int64_t)chunk * a.nposp) + kConvGuard + (int64_t)tile0 * 32 - halo + pos) * 16) : -1;
    dstW[i] = kA + (chunk * wu + pos) * 16;
  }
  struct Stage { u32x4_t a[NA], w[NW]; };
  Stage st;
  auto gload = [&](int ci, Stage& r) {
#pragma unroll
    for (int i = 0; i < NA; ++i)
      if (offA[i] >= 0 || NA * 256 == CT * 576) r.a[i] = __builtin_amdgcn_raw_buffer_load_b128(rA, offA[i] < 0 ? 0 : offA[i], ci * 9 * 1024, 0);
    const uint32_t soff = (uint32_t)((int64_t)2 * ci * a.nposp * 16);
#pragma unroll
    for (int i = 0; i < NW; ++i) r.w[i] = __builtin_amdgcn_raw_buffer_load_b128(rB, offW[i] < 0 ? 0 : offW[i], (int)soff, 0);
  };
  char* const gbase = klds + kg * 2 * kBuf;                          // this group's two stages
  auto sstore = [&](int buf, const Stage& r) {
    char* base = gbase + buf * kBuf;
#pragma unroll
    for (int i = 0; i < NA; ++i)
      if (offA[i] >= 0) *(u32x4_t*)(base + (tg + 256 * i) * 16) = r.a[i];
#pragma unroll
    for (int i = 0; i < NW; ++i)
      if (offW[i] >= 0) *(u32x4_t*)(base + dstW[i]) = r.w[i];
  };
  f32x16 acc[CT][NPT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ct][pt][r] = 0.0f;
  // epilogue operands of the tiles this wave finishes (tile t = ct * NPT + pt with t % KG == kg), requested before the contraction
  f16x8 gate[MODE == kConvDgradMask ? NF : 1][2];
  float bias_r[FWD ? NF : 1][16];
#pragma unroll
  for (int q = 0; q < NF; ++q) {
    const int t = kg + q * KG, ct = t / NPT, pt = t % NPT;
    if (FWD) {
#pragma unroll
      for (int r = 0; r < 16; ++r) bias_r[q][r] = a.bias[32 * (cot0 + ct) + acc_row(r, h)];
    }
    if (MODE == kConvDgradMask) {
      const int64_t p = (int64_t)(tile0 + NPT * pw + pt) * 32 + b;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int chunk = 4 * (cot0 + ct) + 2 * s + h;
        if (chunk < a.cout_chunks) gate[q][s] = ((const f16x8*)a.mask)[(int64_t)chunk * a.nposp + kConvGuard + p];
      }
    }
  }
  gload(kg, st);
  sstore(0, st);
  __syncthreads();
  NPP_STAMP(a, 2);
  int buf = 0;
  const int posw = (WP / 4) * pw + b;                               // this lane's first position inside the workgroup's WP
  for (int ci = kg; ci < a.CI; ci += KG) {
    const bool has_next = ci + KG < a.CI;
    if (has_next) gload(ci + KG, st);
    const char* bA = gbase + buf * kBuf;
    const char* bW = bA + kA + h * wu * 16;
    // the fragments of tap + 1 are requested BEFORE the MFMAs of tap (two register sets): an LDS read takes ~130-200 cycles under
    // load against 64 NPT cycles of matrix work per tap -- read-wait-multiply per tap left the pipe idle 55 % of the loop
    frag_t A[2][CT], B[2][NPT];
    auto lread = [&](int set, int tap) {
      const int shift = (tap / 3) * a.Wp + (tap % 3);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) A[set][ct] = *(const frag_t*)(bA + ((ct * 9 + tap) * 64 + lane) * 16);
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt) B[set][pt] = *(const frag_t*)(bW + (posw + 32 * pt + shift) * 16);
    };
    lread(0, 0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (tap + 1 < 9) lread((tap + 1) & 1, tap + 1);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) acc[ct][pt] = mfma16(A[tap & 1][ct], B[tap & 1][pt], acc[ct][pt]);
    }
    if (has_next) sstore(buf ^ 1, st);
    __syncthreads();
    buf ^= 1;
  }
  NPP_STAMP(a, 3);
  // ---- the groups' partial tiles meet in LDS (the stages are dead behind the last barrier): slot (pw, g, t) = 4 KiB ----
  float* red = (float*)klds;
  if (KG > 1) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (t % KG == kg) continue;                                   // (mine to finish: stays in registers)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(((pw * KG + kg) * NT + t) * 16 + r) * 64 + lane] = acc[t / NPT][t % NPT][r];
    }
    __syncthreads();
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {                                    // (t compile-time: a run-time tile index would put acc in scratch)
    if (t % KG != kg) continue;
    const int q = t / KG, ct = t / NPT, pt = t % NPT;
    f32x16 v;
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = 0.0f;
#pragma unroll
    for (int g = 0; g < KG; ++g) {                                  // fixed order: group 0 first
      if (g == kg) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] += acc[ct][pt][r];
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] += red[(((pw * KG + g) * NT + t) * 16 + r) * 64 + lane];
      }
    }
    const int64_t p = (int64_t)(tile0 + NPT * pw + pt) * 32 + b;
    const int n = (int)((uint32_t)p / (uint32_t)a.S);
    const int r0 = (int)(p - (int64_t)n * a.S);
    const int yy = r0 / a.Wp, xx = r0 - yy * a.Wp;
    const bool interior = p < a.npos_valid && yy >= 1 && yy <= a.H && xx >= 1 && xx <= a.W;
    const int64_t tap_base = (((int64_t)n * a.Ctap) * a.H + (yy - 1)) * a.W + (xx - 1);
    const int cot = cot0 + ct;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int chunk = 4 * cot + 2 * s + h;
      if (chunk >= a.cout_chunks) continue;
      const int64_t unit = (int64_t)chunk * a.nposp + kConvGuard + p;
      frag_t o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int co = 32 * cot + acc_row(8 * s + j, h);
        float x = v[8 * s + j];
        if (FWD) x = fminf(fmaxf(x + bias_r[q][8 * s + j], 0.0f), 65504.0f);
        if (MODE == kConvDgradMask) x = (float)gate[q][s][j] > 0.0f ? x : 0.0f;
        x = interior ? x : 0.0f;
        o[j] = (elem_t)x;
        if (a.tap && interior && co < a.Ctap)
          a.tap[tap_base + (int64_t)co * a.H * a.W] = a.has_scale ? x * a.tap_scale[co & 3] : x;
      }
      if (a.y) ((frag_t*)a.y)[unit] = o;
    }
  }
  NPP_STAMP(a, 4);
  NPP_STAMP_DRAIN();
  NPP_STAMP(a, 5);
  asm volatile("" :: "v"(pf_val));
}

// ---- weight packers (run once per trunk: the weights are frozen) -------------------------
// w: torch Conv2d weight (Cout, Cin, 3, 3) fp32.
// forward pack unit (cot, ci_step, tap, lane=(m,h)) element j = w[32 cot + m][chan_in(2 ci_step + h, j)][ky][kx]
// dgrad  pack unit (cit, co_step, tap, lane=(m,h)) element j = w[conv_chan(2 co_step + h, j)][32 cit + m][2-ky][2-kx]
__global__ void conv_pack_kernel(const float* __restrict__ w, int Cin, int Cout, int in_natural, int CIp /*padded Cin*/,
                                 f16x8* __restrict__ pf, int64_t nf, bf16x8* __restrict__ pb, int64_t nbk) {
  const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (u < nf) {
    const int lane = (int)(u & 63);
    int64_t r = u >> 6;
    const int tap = (int)(r % 9); r /= 9;
    const int CI = CIp / 16;
    const int ci_step = (int)(r % CI);
    const int cot = (int)(r / CI);
    const int m = lane & 31, h = lane >> 5, co = 32 * cot + m;
    f16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c8 = 2 * ci_step + h;
      const int ci = in_natural ? c8 * 8 + j : conv_chan(c8, j);
      const float v = (co < Cout && ci < Cin) ? w[((int64_t)co * Cin + ci) * 9 + tap] : 0.0f;
      o[j] = (_Float16)fminf(fmaxf(v, -65504.0f), 65504.0f);
    }
    pf[u] = o;
  } else if (u < nf + nbk) {
    const int64_t ub = u - nf;
    const int lane = (int)(ub & 63);
    int64_t r = ub >> 6;
    const int tap = (int)(r % 9); r /= 9;
    const int CO = Cout / 16;
    const int co_step = (int)(r % CO);
    const int cit = (int)(r / CO);
    const int m = lane & 31, h = lane >> 5, ci = 32 * cit + m;
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int co = conv_chan(2 * co_step + h, j);
      const float v = (ci < Cin && co < Cout) ? w[((int64_t)co * Cin + ci) * 9 + (8 - tap)] : 0.0f;
      o[j] = (__bf16)v;
    }
    pb[ub] = o;
  }
}

// ---- image in: (N,3,H,W) fp32 -> flat C=16 (channels 0..2 natural order, rest 0), x*scale + shift ----
__global__ void trunk_image_in_kernel(const float* __restrict__ img, int N, int H, int W, float s0, float s1, float s2,
                                      float b0, float b1, float b2, f16x8* __restrict__ out, int64_t nposp,
                                      int64_t npos_round) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npos_round) return;
  const int Wp = W + 2, S = (H + 2) * Wp;
  const int n = (int)(p / S), r = (int)(p - (int64_t)n * S), y = r / Wp, x = r - y * Wp;
  f16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (_Float16)0.0f;
  const f16x8 z = o;
  if (n < N && y >= 1 && y <= H && x >= 1 && x <= W) {
    const int64_t base = ((int64_t)n * 3 * H + (y - 1)) * W + (x - 1);
    const int64_t cs = (int64_t)H * W;
    o[0] = trunk_in_f16(img[base], s0, b0);
    o[1] = trunk_in_f16(img[base + cs], s1, b1);
    o[2] = trunk_in_f16(img[base + 2 * cs], s2, b2);
  }
  out[kConvGuard + p] = o;
  out[nposp + kConvGuard + p] = z;
}

// ---- patch plumbing + image in, fused (row a10 feeding rows a11 / a13): the batch [x | y] of npp_patch_compose_fwd
// written straight into the trunk's flat input (x*scale + shift, fp16), optionally also as the fp32 (2 n_p k,3,P,P)
// tensor the other trunks of the iteration read, and the iteration's patch-loss accumulator zeroed by the way.
__global__ void trunk_patch_in_kernel(const float* __restrict__ pred, const float* __restrict__ fake,
                                      const float* __restrict__ fmask, const float* __restrict__ real,
                                      const float* __restrict__ rmask, int n_p, int k, int P, int comp, float s0, float s1,
                                      float s2, float b0, float b1, float b2, f16x8* __restrict__ out, int64_t nposp,
                                      int64_t npos_round, float* __restrict__ xy, float* __restrict__ zero, int n_zero,
                                      int which, PixelLossArgs pl, int nb_loss) {
  // npp_trunk_patch_in_loss: the first nb_loss blocks are the adaptive pixel loss of the iteration (the other consumer of the
  // prediction; independent of the patch rows) -- one launch instead of two dependent ones
  if ((int)blockIdx.x < nb_loss) {
    pixel_loss_body(pl, (int)blockIdx.x, nb_loss);
    return;
  }
  const int64_t p = (int64_t)((int)blockIdx.x - nb_loss) * blockDim.x + threadIdx.x;
  if (p < n_zero) zero[p] = 0.0f;
  if (p >= npos_round) return;
  const int Wp = P + 2, S = (P + 2) * Wp, nk = n_p * k;
  const int nl = (int)(p / S), r = (int)(p - (int64_t)nl * S), y = r / Wp, x = r - y * Wp;
  const int n_img = which ? nk : 2 * nk;                     // images in the flat tensor
  const int n = which == 2 ? nl + nk : nl;                   // index in the [x | y] batch
  f16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (_Float16)0.0f;
  const f16x8 z = o;
  if (nl < n_img && y >= 1 && y <= P && x >= 1 && x <= P) {
    const int64_t pp = (int64_t)P * P, q = (int64_t)(y - 1) * P + (x - 1);
    const int pk = n < nk ? n : n - nk;
    const float rm = rmask[(int64_t)pk * pp + q];
    float v[3];
    if (n < nk) {                                                                 // prediction half
      const int pi = pk / k;
      const float fm = comp ? fmask[(int64_t)pi * pp + q] : 0.0f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float pv = pred[((int64_t)pi * pp + q) * 3 + c];
        const float u = comp ? fake[((int64_t)pi * 3 + c) * pp + q] * fm + pv * (1.0f - fm) : pv;   // train.py:230-231
        v[c] = u * rm;                                                                              // :232-233
      }
    } else {                                                                      // real half, :235-236
#pragma unroll
      for (int c = 0; c < 3; ++c) v[c] = real[((int64_t)pk * 3 + c) * pp + q] * rm;
    }
    if (xy) {
#pragma unroll
      for (int c = 0; c < 3; ++c) xy[((int64_t)n * 3 + c) * pp + q] = v[c];
    }
    o[0] = trunk_in_f16(v[0], s0, b0);
    o[1] = trunk_in_f16(v[1], s1, b1);
    o[2] = trunk_in_f16(v[2], s2, b2);
  }
  out[kConvGuard + p] = o;
  out[nposp + kConvGuard + p] = z;
}

// ---- the same for M stacked images (npp_common.h "stacked launches"): the trunk batch is [x_0 .. x_{M-1} | y_0 .. y_{M-1}] (image
// m's prediction half at batch index iter[m].x0, its real half at X + x0, X = sum of the nk), so that the data-gradient pass
// runs on the leading X images; the flat tensor has the FIXED geometry of the largest batch (2 M n_p kmax images), of which
// 2 X are written and run.  Crops per image: rgb [fake (n_p) | real (n_p kmax)], masks likewise; 'same' iterations take the
// fake crops as the real ones (sampler.py:338).  The first M * nb_loss blocks are the images' adaptive pixel losses.
struct PatchInStack {
  const float* pred;        // (M, Bp, 3)
  const float* crops;       // (M, n_p + n_p kmax, 3, P, P)
  const float* cmasks;      // (M, n_p + n_p kmax, P, P)
  int64_t Bp, row0, crop_stride, cmask_stride, xy_stride;
  int32_t M, n_p, P, X;
  float s0, s1, s2, b0, b1, b2;
  f16x8* out;
  int64_t nposp, npos_round;
  float* xy;                // (M, 2 n_p kmax, 3, P, P) fp32 [x | y] of the images whose iter.with_lp is set (nullable)
  float* zero;              // M patch-loss accumulators
  const StackIter* iter;
  PixelLossArgs pl;         // image 0; the others at + m * the strides below
  int64_t gt_stride, scratch_stride;
  int32_t lat_stride, loss_stride, nb_loss, pad;
};
__global__ void trunk_patch_in_stack_kernel(PatchInStack a) {
  if ((int)blockIdx.x < a.M * a.nb_loss) {
    const int m = (int)blockIdx.x / a.nb_loss, b = (int)blockIdx.x - m * a.nb_loss;
    if (!a.iter[m].active) return;
    PixelLossArgs pl = a.pl;
    pl.pred += (int64_t)m * a.Bp * 3; pl.dpred += (int64_t)m * a.Bp * 3; pl.gt += (int64_t)m * a.gt_stride;
    if (pl.mask) pl.mask += (int64_t)m * pl.N;                       // per-pixel loss weights (remapping): (M, N) contiguous
    pl.latents += m * a.lat_stride; pl.dlatent += m * a.lat_stride; pl.loss_out += m * a.loss_stride;
    if (pl.scratch) pl.scratch += (int64_t)m * a.scratch_stride;
    pixel_loss_body(pl, b, a.nb_loss);
    return;
  }
  const int64_t p = (int64_t)((int)blockIdx.x - a.M * a.nb_loss) * blockDim.x + threadIdx.x;
  if (p < a.M) a.zero[p] = 0.0f;
  if (p >= a.npos_round) return;
  const int P = a.P, Wp = P + 2, S = (P + 2) * Wp;
  const int nl = (int)(p / S), r = (int)(p - (int64_t)nl * S), y = r / Wp, x = r - y * Wp;
  f16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (_Float16)0.0f;
  const f16x8 z = o;
  if (nl < 2 * a.X && y >= 1 && y <= P && x >= 1 && x <= P) {
    const bool real_half = nl >= a.X;
    const int nb = real_half ? nl - a.X : nl;
    int m = 0;
    for (int j = 1; j < a.M; ++j)
      if (a.iter[j].nk > 0 && nb >= a.iter[j].x0) m = j;          // (x0 ascending; images sitting out have nk = 0)
    const StackIter it = a.iter[m];
    const int pk = nb - it.x0;                                     // patch (p, kk) of image m, < nk
    const int64_t pp = (int64_t)P * P, q = (int64_t)(y - 1) * P + (x - 1);
    const float* fake = a.crops + (int64_t)m * a.crop_stride;
    const float* fmask = a.cmasks + (int64_t)m * a.cmask_stride;
    const float* real = it.same ? fake : fake + (int64_t)a.n_p * 3 * pp;
    const float* rmask = it.same ? fmask : fmask + (int64_t)a.n_p * pp;
    const float rm = rmask[(int64_t)pk * pp + q];
    float v[3];
    if (!real_half) {
      const int pi = pk / it.k;
      const float fm = it.comp ? fmask[(int64_t)pi * pp + q] : 0.0f;
      const float* pr = a.pred + ((int64_t)m * a.Bp + a.row0) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float pv = pr[((int64_t)pi * pp + q) * 3 + c];
        const float u = it.comp ? fake[((int64_t)pi * 3 + c) * pp + q] * fm + pv * (1.0f - fm) : pv;   // train.py:230-231
        v[c] = u * rm;
      }
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) v[c] = real[((int64_t)pk * 3 + c) * pp + q] * rm;
    }
    if (a.xy && it.with_lp) {
      float* xy = a.xy + (int64_t)m * a.xy_stride;
      const int n = real_half ? pk + it.nk : pk;
#pragma unroll
      for (int c = 0; c < 3; ++c) xy[((int64_t)n * 3 + c) * pp + q] = v[c];
    }
    o[0] = trunk_in_f16(v[0], a.s0, a.b0);
    o[1] = trunk_in_f16(v[1], a.s1, a.b1);
    o[2] = trunk_in_f16(v[2], a.s2, a.b2);
  }
  a.out[kConvGuard + p] = o;
  a.out[a.nposp + kConvGuard + p] = z;
}

__device__ __forceinline__ void unit_decode(int64_t p, int H, int W, int& n, int& y, int& x) {
  const int Wp = W + 2, S = (H + 2) * Wp;
  n = (int)(p / S);
  const int r = (int)(p - (int64_t)n * S);
  y = r / Wp;
  x = r - y * Wp;
}

// ---- MaxPool2d(2,2) forward on flat tensors: (N,C,H,W) -> (N,C,H/2,W/2) ------------------
__global__ void maxpool2_fwd_kernel(const f16x8* __restrict__ in, int N, int H, int W, int64_t nposp_in,
                                    f16x8* __restrict__ out, int64_t nposp_out, int64_t npos_round_out, int chunks) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= npos_round_out * chunks) return;
  const int c8 = (int)(t / npos_round_out);
  const int64_t p = t - (int64_t)c8 * npos_round_out;
  const int Ho = H / 2, Wo = W / 2;
  int n, y, x;
  unit_decode(p, Ho, Wo, n, y, x);
  f16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (_Float16)0.0f;
  if (n < N && y >= 1 && y <= Ho && x >= 1 && x <= Wo) {
    const int Wp = W + 2;
    const int64_t q = (int64_t)c8 * nposp_in + kConvGuard + (int64_t)n * (H + 2) * Wp + (int64_t)(2 * y - 1) * Wp + (2 * x - 1);
    const f16x8 a0 = in[q], a1 = in[q + 1], a2 = in[q + Wp], a3 = in[q + Wp + 1];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      o[j] = (_Float16)fmaxf(fmaxf((float)a0[j], (float)a1[j]), fmaxf((float)a2[j], (float)a3[j]));
  }
  out[(int64_t)c8 * nposp_out + kConvGuard + p] = o;
}

// ---- MaxPool2d(2,2) backward fused with the ReLU gate of the pre-pool layer ----------------
// dz[pre-pool] = (route(dy) + addend) * [x > 0]; the gradient goes to the FIRST maximum of the window in
// scan order (torch's max_pool2d keeps `val > maxval`).  addend: optional tap gradient on the pre-pool tensor.
__global__ void maxpool2_bwd_kernel(const bf16x8* __restrict__ dy, const f16x8* __restrict__ xin,
                                    const bf16x8* __restrict__ addend, int N, int H, int W, int64_t nposp_in,
                                    int64_t nposp_out, int64_t npos_range, bf16x8* __restrict__ dz, int chunks) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= npos_range * chunks) return;
  const int c8 = (int)(t / npos_range);
  const int64_t p = t - (int64_t)c8 * npos_range;
  int n, y, x;
  unit_decode(p, H, W, n, y, x);
  const int64_t u = (int64_t)c8 * nposp_in + kConvGuard + p;
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (__bf16)0.0f;
  if (n < N && y >= 1 && y <= H && x >= 1 && x <= W) {
    const int Ho = H / 2, Wo = W / 2, Wp = W + 2;
    const int yo = (y - 1) / 2 + 1, xo = (x - 1) / 2 + 1;
    const f16x8 me = xin[u];
    float g[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = 0.0f;
    if (yo <= Ho && xo <= Wo) {
      const int wy = (y - 1) & 1, wx = (x - 1) & 1, k = wy * 2 + wx;     // my slot in the window
      const int64_t q = u - wy * Wp - wx;
      const f16x8 w0 = xin[q], w1 = xin[q + 1], w2 = xin[q + Wp], w3 = xin[q + Wp + 1];
      const bf16x8 d = dy[(int64_t)c8 * nposp_out + kConvGuard + (int64_t)n * (Ho + 2) * (Wo + 2) + (int64_t)yo * (Wo + 2) + xo];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v0 = (float)w0[j], v1 = (float)w1[j], v2 = (float)w2[j], v3 = (float)w3[j];
        int am = 0;
        float mx = v0;
        if (v1 > mx) { mx = v1; am = 1; }
        if (v2 > mx) { mx = v2; am = 2; }
        if (v3 > mx) { mx = v3; am = 3; }
        g[j] = am == k ? (float)d[j] : 0.0f;
      }
    }
    if (addend) {
      const bf16x8 ad = addend[u];
#pragma unroll
      for (int j = 0; j < 8; ++j) g[j] += (float)ad[j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (__bf16)((float)me[j] > 0.0f ? g[j] : 0.0f);
  }
  dz[u] = o;
}

// ---- tap gradient in: df (Ng,C,H,W) fp32 -> flat bf16; gated by [y > 0] when y is given ------------
template <bool F16OUT>
__global__ void trunk_grad_in_kernel(const float* __restrict__ df, const f16x8* __restrict__ yact, int Ng, int C, int H,
                                     int W, int64_t nposp, int64_t npos_range, void* __restrict__ dz_, int accumulate,
                                     const char* pf, int64_t pf_bytes) {
  typedef typename OpT<F16OUT>::frag frag_t;
  typedef typename OpT<F16OUT>::elem elem_t;
  frag_t* dz = (frag_t*)dz_;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pf) {              // the first data-gradient layer's weight pack -> every XCD's L2 (see conv3x3_kernel)
    const int64_t line = ((int64_t)(blockIdx.x >> 3) * blockDim.x + threadIdx.x) * 128;
    if (line + 4 <= pf_bytes) { const uint32_t v = *(const volatile uint32_t*)(pf + line); asm volatile("" :: "v"(v)); }
  }
  const int chunks = C / 8;
  if (t >= npos_range * chunks) return;
  const int c8 = (int)(t / npos_range);
  const int64_t p = t - (int64_t)c8 * npos_range;
  int n, y, x;
  unit_decode(p, H, W, n, y, x);
  const int64_t u = (int64_t)c8 * nposp + kConvGuard + p;
  frag_t o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (elem_t)0.0f;
  if (n < Ng && y >= 1 && y <= H && x >= 1 && x <= W) {
    f16x8 m;
    if (yact) m = yact[u];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = conv_chan(c8, j);
      const float v = df[(((int64_t)n * C + c) * H + (y - 1)) * W + (x - 1)];
      o[j] = (elem_t)((!yact || (float)m[j] > 0.0f) ? v : 0.0f);
    }
    if (accumulate) {                                   // tap gradient added onto a gradient that is already there
      const frag_t old = dz[u];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (elem_t)((float)o[j] + (float)old[j]);
    }
  } else if (accumulate) {
    return;                                             // borders / other images: leave what is there
  }
  dz[u] = o;
}

// ---- flat -> (N,C,H,W) fp32 (tests / taps of tensors that were not exported by the conv epilogue) ----
template <bool F16IN>
__global__ void trunk_export_kernel(const void* __restrict__ act_, int N, int C, int H, int W, int64_t nposp,
                                    float* __restrict__ out) {
  typedef typename OpT<F16IN>::frag frag_t;
  const frag_t* act = (const frag_t*)act_;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int chunks = C / 8;
  const int64_t hw = (int64_t)H * W;
  if (t >= (int64_t)N * hw * chunks) return;
  const int c8 = (int)(t / (N * hw));
  const int64_t r = t - (int64_t)c8 * N * hw;
  const int n = (int)(r / hw);
  const int q = (int)(r - (int64_t)n * hw), y = q / W, x = q - y * W;
  const frag_t v = act[(int64_t)c8 * nposp + kConvGuard + (int64_t)n * (H + 2) * (W + 2) + (int64_t)(y + 1) * (W + 2) + (x + 1)];
#pragma unroll
  for (int j = 0; j < 8; ++j) out[(((int64_t)n * C + conv_chan(c8, j)) * H + y) * W + x] = (float)v[j];
}

}  // namespace npp

using namespace npp;

#ifdef NPP_DIAG
namespace npp { unsigned long long* g_diag_stamps = nullptr; long long g_diag_n = 0; }
extern "C" void npp_diag_set_stamps(unsigned long long* buf, long long n_words) { npp::g_diag_stamps = buf; npp::g_diag_n = n_words; }
#endif

static int conv_geom_check(int N, int H, int W, const char* who) {
  if (N < 1 || H < 1 || W < 1 || W + 3 > kConvGuard) {
    set_error("%s: bad geometry N=%d H=%d W=%d (W <= %d)", who, N, H, W, kConvGuard - 3);
    return NPP_ERR_ARG;
  }
  if (conv_nposp(N, H, W) * 16 * 64 > 0x7fffffffLL) {   // 512 channels * nposp * 16 B must fit a 32-bit buffer descriptor
    set_error("%s: tensor too large for one launch (N*(H+2)*(W+2) = %lld positions)", who, (long long)conv_npos_round(N, H, W));
    return NPP_ERR_ARG;
  }
  return NPP_OK;
}

extern "C" int64_t npp_trunk_nposp(int N, int H, int W) {
  if (N < 1 || H < 1 || W < 1) return NPP_ERR_ARG;
  return conv_nposp(N, H, W);
}

extern "C" int64_t npp_trunk_act_bytes(int N, int C, int H, int W) {
  if (N < 1 || H < 1 || W < 1 || C < 1) return NPP_ERR_ARG;
  return (int64_t)((C + 15) / 16 * 2) * conv_nposp(N, H, W) * 16;
}

extern "C" int64_t npp_conv_pack_bytes(int Cin, int Cout, int which) {
  if (Cin < 1 || Cout < 1 || Cout % 16) return NPP_ERR_ARG;
  const int CIp = (Cin + 15) / 16 * 16;
  if (which == 0) return (int64_t)((Cout + 31) / 32) * (CIp / 16) * 9 * 1024;
  if (which == 1) return (int64_t)((CIp + 31) / 32) * (Cout / 16) * 9 * 1024;
  return NPP_ERR_ARG;
}

extern "C" int npp_conv_pack(const float* d_w, int Cin, int Cout, int in_natural, void* d_pack_fwd, void* d_pack_bwd,
                             void* stream) {
  if (!d_w || !d_pack_fwd || !d_pack_bwd || Cin < 1 || Cout < 16 || Cout % 16) {
    set_error("npp_conv_pack: bad argument (Cin=%d Cout=%d)", Cin, Cout);
    return NPP_ERR_ARG;
  }
  if (!in_natural && Cin % 16) { set_error("npp_conv_pack: Cin=%d must be a multiple of 16 unless in_natural", Cin); return NPP_ERR_ARG; }
  const int CIp = (Cin + 15) / 16 * 16;
  const int64_t nf = npp_conv_pack_bytes(Cin, Cout, 0) / 16, nb = npp_conv_pack_bytes(Cin, Cout, 1) / 16;
  const int64_t n = nf + nb;
  hipLaunchKernelGGL(conv_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_w, Cin, Cout,
                     in_natural, CIp, (f16x8*)d_pack_fwd, nf, (bf16x8*)d_pack_bwd, nb);
  return check_launch("npp_conv_pack");
}

extern "C" int npp_trunk_image_in(const float* d_img_nchw, int N, int H, int W, const float scale[3], const float shift[3],
                                  void* d_x0, void* stream) {
  int rc = conv_geom_check(N, H, W, "npp_trunk_image_in");
  if (rc) return rc;
  if (!d_img_nchw || !d_x0 || !scale || !shift) { set_error("npp_trunk_image_in: null pointer"); return NPP_ERR_ARG; }
  const int64_t nr = conv_npos_round(N, H, W);
  hipLaunchKernelGGL(trunk_image_in_kernel, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_img_nchw,
                     N, H, W, scale[0], scale[1], scale[2], shift[0], shift[1], shift[2], (f16x8*)d_x0, conv_nposp(N, H, W), nr);
  return check_launch("npp_trunk_image_in");
}

static int patch_in_launch(const float* d_pred_rows, const float* d_fake, const float* d_fmask, const float* d_real,
                           const float* d_rmask, int n_p, int k, int P, int comp, const float scale[3], const float shift[3],
                           void* d_x0, float* d_xy, float* d_zero, int n_zero, int which, const PixelLossArgs* pl, void* stream,
                           const char* who) {
  if (n_p < 1 || k < 1 || n_zero < 0 || n_zero > 256 || which < 0 || which > 2) {
    set_error("%s: bad n_p=%d k=%d n_zero=%d which=%d", who, n_p, k, n_zero, which);
    return NPP_ERR_ARG;
  }
  const int N = (which ? 1 : 2) * n_p * k;
  int rc = conv_geom_check(N, P, P, who);
  if (rc) return rc;
  const bool need_x = which != 2, need_y = which != 1;
  if ((need_x && !d_pred_rows) || (need_y && !d_real) || !d_rmask || !d_x0 || !scale || !shift ||
      (need_x && comp && (!d_fake || !d_fmask)) || (n_zero && !d_zero)) {
    set_error("%s: null pointer", who);
    return NPP_ERR_ARG;
  }
  if (pl && (pl->N <= 0 || !pl->pred || !pl->gt || !pl->loss_out || !pl->dpred || pl->quad < 0.0f ||
             (pl->quad == 0.0f && (!pl->latents || !pl->spline || !pl->dlatent || pl->n_knots < 2)))) {
    set_error("%s: bad pixel-loss arguments (N=%lld)", who, (long long)(pl ? pl->N : 0));
    return NPP_ERR_ARG;
  }
  const int64_t nr = conv_npos_round(N, P, P);
  const int nb_loss = pl ? pixel_loss_blocks(pl->N) : 0;
  hipLaunchKernelGGL(trunk_patch_in_kernel, dim3((unsigned)((nr + 255) / 256 + nb_loss)), dim3(256), 0, (hipStream_t)stream,
                     d_pred_rows, d_fake, d_fmask, d_real, d_rmask, n_p, k, P, comp, scale[0], scale[1], scale[2], shift[0], shift[1],
                     shift[2], (f16x8*)d_x0, conv_nposp(N, P, P), nr, d_xy, d_zero, n_zero, which, pl ? *pl : PixelLossArgs{}, nb_loss);
  return check_launch(who);
}

extern "C" int npp_trunk_patch_in(const float* d_pred_rows, const float* d_fake, const float* d_fmask, const float* d_real,
                                  const float* d_rmask, int n_p, int k, int P, int comp, const float scale[3],
                                  const float shift[3], void* d_x0, float* d_xy, float* d_zero, int n_zero, int which,
                                  void* stream) {
  return patch_in_launch(d_pred_rows, d_fake, d_fmask, d_real, d_rmask, n_p, k, P, comp, scale, shift, d_x0, d_xy, d_zero, n_zero,
                         which, nullptr, stream, "npp_trunk_patch_in");
}

// npp_trunk_patch_in + npp_pixel_loss in ONE launch (both read the prediction of the forward launch before them).
extern "C" int npp_trunk_patch_in_loss(const float* d_pred_rows, const float* d_fake, const float* d_fmask, const float* d_real,
                                       const float* d_rmask, int n_p, int k, int P, int comp, const float scale[3],
                                       const float shift[3], void* d_x0, float* d_xy, float* d_zero, int n_zero, int which,
                                       const npp_pixel_loss_args* loss, void* stream) {
  if (!loss) { set_error("npp_trunk_patch_in_loss: null pixel-loss arguments"); return NPP_ERR_ARG; }
  const PixelLossArgs pl{loss->pred, loss->gt, loss->mask, loss->N, loss->latents, loss->spline, loss->n_knots, loss->x_scale,
                         loss->weight, loss->loss, loss->dpred, loss->dlatent, loss->scratch, loss->quad};
  return patch_in_launch(d_pred_rows, d_fake, d_fmask, d_real, d_rmask, n_p, k, P, comp, scale, shift, d_x0, d_xy, d_zero, n_zero,
                         which, &pl, stream, "npp_trunk_patch_in_loss");
}

// Stacked form of npp_trunk_patch_in_loss (M images; see trunk_patch_in_stack_kernel).  N_total = images of the flat tensor's
// geometry (2 M n_p kmax), X = sum of the images' n_p k this iteration (host-side sum of the npp_stack_iter entries).
extern "C" int npp_trunk_patch_in_loss_stack(const float* d_pred, int64_t Bp, int64_t row0, const float* d_crops, int64_t crop_stride,
                                             const float* d_cmasks, int64_t cmask_stride, int M, int n_p, int P, int X, int N_total,
                                             const float scale[3], const float shift[3], void* d_x0, float* d_xy, int64_t xy_stride,
                                             float* d_zero, const void* d_iter, const npp_pixel_loss_args* loss, int64_t gt_stride,
                                             int lat_stride, int loss_stride, int64_t scratch_stride, void* stream) {
  const char* who = "npp_trunk_patch_in_loss_stack";
  if (M < 1 || M > NPP_MAX_STACK || n_p < 1 || X < 0 || 2 * X > N_total || !d_pred || !d_crops || !d_cmasks || !d_x0 || !d_zero ||
      !d_iter || !scale || !shift || !loss || row0 < 0 || row0 + (int64_t)n_p * P * P > Bp) {
    set_error("%s: bad arguments (M=%d n_p=%d X=%d N_total=%d)", who, M, n_p, X, N_total);
    return NPP_ERR_ARG;
  }
  int rc = conv_geom_check(N_total, P, P, who);
  if (rc) return rc;
  if (loss->N <= 0 || !loss->pred || !loss->gt || !loss->loss || !loss->dpred || loss->quad < 0.0f ||
      (loss->quad == 0.0f && (!loss->latents || !loss->spline || !loss->dlatent || loss->n_knots < 2))) {
    set_error("%s: bad pixel-loss arguments (N=%lld)", who, (long long)loss->N);
    return NPP_ERR_ARG;
  }
  PatchInStack a{};
  a.pred = d_pred; a.crops = d_crops; a.cmasks = d_cmasks; a.Bp = Bp; a.row0 = row0; a.crop_stride = crop_stride;
  a.cmask_stride = cmask_stride; a.xy_stride = xy_stride; a.M = M; a.n_p = n_p; a.P = P; a.X = X;
  a.s0 = scale[0]; a.s1 = scale[1]; a.s2 = scale[2]; a.b0 = shift[0]; a.b1 = shift[1]; a.b2 = shift[2];
  a.out = (f16x8*)d_x0; a.nposp = conv_nposp(N_total, P, P);
  a.npos_round = X > 0 ? conv_npos_round(2 * X, P, P) : 0;
  a.xy = d_xy; a.zero = d_zero; a.iter = (const StackIter*)d_iter;
  a.pl = PixelLossArgs{loss->pred, loss->gt, loss->mask, loss->N, loss->latents, loss->spline, loss->n_knots, loss->x_scale,
                       loss->weight, loss->loss, loss->dpred, loss->dlatent, loss->scratch, loss->quad};
  a.gt_stride = gt_stride; a.lat_stride = lat_stride; a.loss_stride = loss_stride; a.scratch_stride = scratch_stride;
  a.nb_loss = pixel_loss_blocks(loss->N);
  const int64_t nblk = (a.npos_round > M ? a.npos_round : M) ;
  hipLaunchKernelGGL(trunk_patch_in_stack_kernel, dim3((unsigned)((nblk + 255) / 256 + (int64_t)M * a.nb_loss)), dim3(256), 0,
                     (hipStream_t)stream, a);
  return check_launch(who);
}

template <int CT, int PT, int S>
static int conv_launch_mode(const ConvArgs& a, int mode, dim3 grid, hipStream_t s) {
  const size_t smem = S > 1 ? (size_t)S * CT * PT * 16 * 64 * sizeof(float) : 0;
#define NPP_CONV_GO(M)                                                                                        \
  do {                                                                                                        \
    static SmemOnce once;                                                                                     \
    if (smem > 48 * 1024 && !smem_attr(once, (const void*)conv3x3_kernel<CT, PT, S, M>, (int)smem)) {          \
      set_error("npp_conv3x3: smem attribute"); return NPP_ERR_LAUNCH;                                        \
    }                                                                                                         \
    hipLaunchKernelGGL((conv3x3_kernel<CT, PT, S, M>), grid, dim3(64 * S), smem, s, a);                       \
  } while (0)
  if (mode == kConvFwd) NPP_CONV_GO(kConvFwd);
  else if (mode == kConvDgradMask) NPP_CONV_GO(kConvDgradMask);
  else if (mode == kConvFwdPool) NPP_CONV_GO(kConvFwdPool);
  else if (mode == kConvDgradPool) NPP_CONV_GO(kConvDgradPool);
  else NPP_CONV_GO(kConvDgradLin);
#undef NPP_CONV_GO
  return NPP_OK;
}

// mode 0: y = relu(conv(x) + bias)            (forward layer)
// mode 1: y = conv_T(x) * [mask > 0]          (data gradient through a conv into a ReLU layer's pre-activation)
// mode 2: y = conv_T(x)                       (data gradient into a pooled tensor / the image)
// N_total fixes the geometry of the buffers, n_run <= N_total the leading images actually computed.
struct PoolFold { const void* x; const void* add; void* dz; void* ypool; };
static int conv3x3_impl(const void* d_x, int N_total, int n_run, int H, int W, int Cin, int Cout, const void* d_pack,
                        const float* d_bias, int mode, const void* d_mask, void* d_y, float* d_tap, int Ctap,
                        const float* tap_scale, const void* d_next_pack, int64_t next_pack_bytes, void* stream,
                        const PoolFold* fold = nullptr);
extern "C" int npp_conv3x3(const void* d_x, int N_total, int n_run, int H, int W, int Cin, int Cout, const void* d_pack,
                           const float* d_bias, int mode, const void* d_mask, void* d_y, float* d_tap, int Ctap,
                           const float* tap_scale, void* stream) {
  return conv3x3_impl(d_x, N_total, n_run, H, W, Cin, Cout, d_pack, d_bias, mode, d_mask, d_y, d_tap, Ctap, tap_scale, nullptr, 0, stream);
}
// The same launch additionally requesting the weight pack of the launch that FOLLOWS it into L2 (d_next_pack, bytes).
extern "C" int npp_conv3x3_pf(const void* d_x, int N_total, int n_run, int H, int W, int Cin, int Cout, const void* d_pack,
                              const float* d_bias, int mode, const void* d_mask, void* d_y, float* d_tap, int Ctap,
                              const float* tap_scale, const void* d_next_pack, int64_t next_pack_bytes, void* stream) {
  return conv3x3_impl(d_x, N_total, n_run, H, W, Cin, Cout, d_pack, d_bias, mode, d_mask, d_y, d_tap, Ctap, tap_scale, d_next_pack,
                      next_pack_bytes, stream);
}
// The data gradient of a convolution whose input is a POOLED tensor, with MaxPool2d(2,2)'s backward, the pre-pool layer's ReLU
// gate and its optional tap gradient folded into the epilogue (= npp_conv3x3 mode 2 into a scratch tensor followed by
// npp_maxpool2_bwd, bit for bit, in one launch).  H, W: the pooled geometry (this convolution's); d_xpre / d_addend / d_dz: flat
// tensors of the pre-pool layer, geometry (N_total, Cout, 2H, 2W); d_dz's border must be zero and stays untouched.
extern "C" int npp_conv3x3_dgrad_pool(const void* d_x, int N_total, int n_run, int H, int W, int Cin, int Cout, const void* d_pack,
                                      const void* d_xpre, const void* d_addend, void* d_dz, const void* d_next_pack,
                                      int64_t next_pack_bytes, void* stream) {
  if (!d_xpre || !d_dz) { set_error("npp_conv3x3_dgrad_pool: null pre-pool tensor"); return NPP_ERR_ARG; }
  int rc = conv_geom_check(N_total, 2 * H, 2 * W, "npp_conv3x3_dgrad_pool");
  if (rc) return rc;
  const PoolFold f{d_xpre, d_addend, d_dz, nullptr};
  return conv3x3_impl(d_x, N_total, n_run, H, W, Cin, Cout, d_pack, nullptr, kConvDgradLin, nullptr, d_dz, nullptr, 0, nullptr,
                      d_next_pack, d_next_pack ? next_pack_bytes : 0, stream, &f);
}
// Forward layer with the nn.MaxPool2d(2,2) that follows it folded in: y = relu(conv(x) + bias) as npp_conv3x3 mode 0 (flat fp16
// tensor + optional fp32 tap) AND d_ypool = maxpool(y) (geometry (N_total, Cout, H/2, W/2)) in one launch; H and W even.  Position
// tiles are 16 columns x 2 rows, so a pool window sits in four lanes of one tile; bit-identical to the two launches.
extern "C" int npp_conv3x3_pool(const void* d_x, int N_total, int n_run, int H, int W, int Cin, int Cout, const void* d_pack,
                                const float* d_bias, void* d_y, void* d_ypool, float* d_tap, int Ctap, const float* tap_scale,
                                const void* d_next_pack, int64_t next_pack_bytes, void* stream) {
  if (!d_ypool || !d_y || (H & 1) || (W & 1)) { set_error("npp_conv3x3_pool: needs d_y, d_ypool and even H, W (H=%d W=%d)", H, W); return NPP_ERR_ARG; }
  const PoolFold f{nullptr, nullptr, nullptr, d_ypool};
  return conv3x3_impl(d_x, N_total, n_run, H, W, Cin, Cout, d_pack, d_bias, kConvFwd, nullptr, d_y, d_tap, Ctap, tap_scale,
                      d_next_pack, d_next_pack ? next_pack_bytes : 0, stream, &f);
}
static int conv3x3_impl(const void* d_x, int N_total, int n_run, int H, int W, int Cin, int Cout, const void* d_pack,
                        const float* d_bias, int mode, const void* d_mask, void* d_y, float* d_tap, int Ctap,
                        const float* tap_scale, const void* d_next_pack, int64_t next_pack_bytes, void* stream,
                        const PoolFold* fold) {
  int rc = conv_geom_check(N_total, H, W, "npp_conv3x3");
  if (rc) return rc;
  if (!d_x || !d_pack || (!d_y && !d_tap) || mode < 0 || mode > 2 || n_run < 1 || n_run > N_total) {
    set_error("npp_conv3x3: bad argument (mode=%d n_run=%d N=%d)", mode, n_run, N_total);
    return NPP_ERR_ARG;
  }
  if (Cin % 16 || Cout % 16 || Cin < 16 || Cout < 16 || Cin > 512 || Cout > 512) {
    set_error("npp_conv3x3: Cin=%d / Cout=%d must be multiples of 16 in [16, 512] (pad the image to 16 channels)", Cin, Cout);
    return NPP_ERR_ARG;
  }
  if ((mode == kConvFwd && !d_bias) || (mode == kConvDgradMask && !d_mask)) { set_error("npp_conv3x3: mode %d needs bias / mask", mode); return NPP_ERR_ARG; }
  if (d_tap && (Ctap < 1 || Ctap > Cout || (tap_scale && Ctap > 4))) { set_error("npp_conv3x3: bad Ctap=%d", Ctap); return NPP_ERR_ARG; }
  ConvArgs a{};
  a.x = d_x; a.pack = d_pack; a.bias = d_bias; a.mask = d_mask; a.y = d_y; a.tap = d_tap;
  a.N = N_total; a.H = H; a.W = W; a.Wp = W + 2; a.S = (H + 2) * (W + 2); a.CI = Cin / 16; a.cout_chunks = Cout / 8;
  a.Ctap = d_tap ? Ctap : 0; a.has_scale = tap_scale != nullptr;
  for (int i = 0; i < 4; ++i) a.tap_scale[i] = (tap_scale && i < Ctap) ? tap_scale[i] : 1.0f;
  a.nposp = conv_nposp(N_total, H, W);
  a.npos_valid = (int64_t)N_total * a.S;
  const int64_t range = n_run == N_total ? conv_npos_round(N_total, H, W)
                                         : ((int64_t)n_run * a.S + kPosRound - 1) / kPosRound * kPosRound;
  a.pos_tiles = (int)(range / 32);
  a.x_bytes = (uint32_t)((int64_t)(Cin / 8) * a.nposp * 16);
  const int cot_n = (Cout + 31) / 32;
  a.pack_bytes = (uint32_t)((int64_t)cot_n * a.CI * 9 * 1024);
  a.pf = (const char*)d_next_pack;
  a.pf_bytes = d_next_pack ? next_pack_bytes : 0;
  NPP_DIAG_FILL(a);
  if (fold && fold->ypool) {                       // forward + pool: two-row position tiles over the n_run images' interiors
    a.pool_dz = fold->ypool;
    a.pool_nposp = conv_nposp(N_total, H / 2, W / 2);
    a.pool_cbn = (W + 15) / 16;
    a.pool_tiles = n_run * (H / 2) * a.pool_cbn;
    a.pos_tiles = (a.pool_tiles + 1) / 2 * 2;
    mode = kConvFwdPool;
  } else if (fold) {
    a.pool_x = fold->x; a.pool_add = fold->add; a.pool_dz = fold->dz;
    a.pool_nposp = conv_nposp(N_total, 2 * H, 2 * W);
    a.y = nullptr;
    mode = kConvDgradPool;
  }
  hipStream_t s = (hipStream_t)stream;
  // Tile choice: the largest output tile per workgroup (fewest operand bytes per MFMA) that still yields about one
  // workgroup per CU, with the contraction split over S waves (CI % S == 0).  NPP_CONV_TILE="ct,pt,s" forces one
  // (diagnostics: tools/conv_probe.py).
  // Measured on MI355X (tools/conv_probe.py, 12 x 96^2 VGG shapes): 2x2 tiles with S = 4 win wherever the channel steps
  // split four ways (c2_2 21 -> 18 us, c3_x 20 -> 14.5, c4_x 42 -> 25, c5_x 38 -> 15 with 1x1 S = 4); position-rich layers
  // (>= 1024 workgroups without a split) are better off unsplit; 2x4 tiles never won at these sizes.
  struct Cand { int ct, pt, s; int64_t min_wgs; };
  static const Cand cands[] = {{2, 2, 1, 1024}, {2, 2, 4, 200}, {2, 1, 4, 200}, {1, 1, 4, 100}, {1, 1, 8, 1}, {2, 1, 1, 1}, {1, 1, 1, 1}};
  auto feasible = [&](const Cand& c) { return cot_n % c.ct == 0 && a.CI % c.s == 0 && a.pos_tiles % c.pt == 0; };
  auto wgs = [&](const Cand& c) { return (int64_t)(a.pos_tiles / c.pt) * (cot_n / c.ct); };
  Cand pick = {1, 1, 1, 1};
  static const char* force = getenv("NPP_CONV_TILE");
  bool found = false;
  if (force && strlen(force) == 5 && force[1] == ',' && force[3] == ',') {
    for (const Cand& c : cands)
      if (c.ct == force[0] - '0' && c.pt == force[2] - '0' && c.s == force[4] - '0' && feasible(c)) { pick = c; found = true; }
  }
  const bool forced = found;
  if (!found && a.CI == 1) {                    // the image layer: 9 k-steps, nothing to split
    const Cand c = {2, 1, 1, 1};
    if (feasible(c)) { pick = c; found = true; }
  }
  for (const Cand& c : cands) {
    if (found) break;
    // (two-row tiles of the pool fold waste columns on narrow maps -- 12 of 16 on VGG16's conv4_3 -- which must not push the
    //  launch to a smaller tile than the plain layer beside it runs with: 192 workgroups of 2 x 1 tiles beat 384 of 1 x 1)
    const int64_t need = mode == kConvFwdPool ? c.min_wgs * 9 / 10 : c.min_wgs;
    if (feasible(c) && wgs(c) >= need) { pick = c; found = true; }
  }
  // window form with the contraction split over wave groups (conv3x3_wink_kernel): channel-rich layers (>= 8 channel steps) whose
  // positions give about one 8-wave workgroup per CU; NPP_CONV_WINK=0: never (A/B comparator), 2: wherever feasible (probe)
  const int wink_mode = __atomic_load_n(&g_tune.conv_wink, __ATOMIC_RELAXED);
  if (wink_mode && !forced && !fold && cot_n % 2 == 0 && a.Wp <= kWinkMaxWp && a.CI % 2 == 0 && (a.CI >= 8 || wink_mode == 2) &&
      (mode == kConvFwd || mode == kConvDgradMask || mode == kConvDgradLin)) {
    const int64_t wgs2 = (int64_t)(a.pos_tiles / 8) * (cot_n / 2), wgs1 = (int64_t)(a.pos_tiles / 4) * (cot_n / 2);
    const int npt = (a.pos_tiles % 8 == 0 && wgs2 >= 200) ? 2 : ((a.pos_tiles % 4 == 0 && (wgs1 >= 200 || wink_mode == 2)) ? 1 : 0);
    if (npt) {
      const int WPk = 128 * npt, wu = (WPk + 2 * (a.Wp + 1) + 7) / 8 * 8;
      const int stage = 2 * 9 * 1024 + 2 * wu * 16;
      const int red = 4 * 2 * (2 * npt) * 4096;
      const int smem = 2 * 2 * stage > red ? 2 * 2 * stage : red;
      const dim3 kgrid((unsigned)(a.pos_tiles / (4 * npt)), (unsigned)(cot_n / 2));
#define NPP_WINK_GO(NPT_, M)                                                                                                    \
      do {                                                                                                                        \
        static SmemOnce once;                                                                                                     \
        if (!smem_attr(once, (const void*)conv3x3_wink_kernel<2, NPT_, 2, M>, 160 * 1024)) { set_error("npp_conv3x3: smem attribute"); return NPP_ERR_LAUNCH; } \
        hipLaunchKernelGGL((conv3x3_wink_kernel<2, NPT_, 2, M>), kgrid, dim3(512), smem, s, a, wu);                               \
      } while (0)
      if (smem <= 160 * 1024) {
        if (npt == 2) { if (mode == kConvFwd) NPP_WINK_GO(2, kConvFwd); else if (mode == kConvDgradMask) NPP_WINK_GO(2, kConvDgradMask); else NPP_WINK_GO(2, kConvDgradLin); }
        else { if (mode == kConvFwd) NPP_WINK_GO(1, kConvFwd); else if (mode == kConvDgradMask) NPP_WINK_GO(1, kConvDgradMask); else NPP_WINK_GO(1, kConvDgradLin); }
        return check_launch("npp_conv3x3");
      }
#undef NPP_WINK_GO
    }
  }
  // window-staged form (conv3x3_win_kernel): few input-channel steps, many positions, 64-channel output blocks
  const int win_mode = __atomic_load_n(&g_tune.conv_win, __ATOMIC_RELAXED);     // 0: never (A/B comparator)
  const int64_t win_wgs = (int64_t)(a.pos_tiles / 8) * (cot_n / 2);
  if (win_mode && !forced && !fold && a.CI <= 8 && cot_n % 2 == 0 && a.pos_tiles % 8 == 0 && 2 * (a.Wp + 1) + kWinPos <= kWinMaxUnits &&
      ((win_wgs >= 200 && mode == kConvFwd && a.CI <= 4) || win_mode == 2)) {       // measured: only these layers gain (conv1_1, conv1_2, conv2_1 forward)
    const dim3 wgrid((unsigned)(a.pos_tiles / 8), (unsigned)(cot_n / 2));
    constexpr int smem = win_lds_bytes<2>();
#define NPP_WIN_GO(M)                                                                                           \
    do {                                                                                                          \
      static SmemOnce once;                                                                                       \
      if (!smem_attr(once, (const void*)conv3x3_win_kernel<2, M>, smem)) { set_error("npp_conv3x3: smem attribute"); return NPP_ERR_LAUNCH; } \
      hipLaunchKernelGGL((conv3x3_win_kernel<2, M>), wgrid, dim3(256), smem, s, a);                               \
    } while (0)
    if (mode == kConvFwd) NPP_WIN_GO(kConvFwd);
    else if (mode == kConvDgradMask) NPP_WIN_GO(kConvDgradMask);
    else NPP_WIN_GO(kConvDgradLin);
#undef NPP_WIN_GO
    return check_launch("npp_conv3x3");
  }
  dim3 grid((unsigned)(a.pos_tiles / pick.pt), (unsigned)(cot_n / pick.ct));
  // weight-stationary numbering where the pack outweighs the activations the launch reads and there are channel groups for all XCDs
  const int wstat_mode = __atomic_load_n(&g_tune.conv_wstat, __ATOMIC_RELAXED);     // 0: never (A/B comparator)
  const int64_t act_bytes = (int64_t)(Cin / 8) * range * 16;
  if (wstat_mode && (int)grid.y >= 8 && (int64_t)a.pack_bytes > 2 * act_bytes) {
    a.wstat = 1; a.wstat_npos = (int)grid.x; a.wstat_ncg = (int)grid.y;
    a.pf = nullptr; a.pf_bytes = 0;                       // (the next pack is partitioned the same way: nothing to request into EVERY L2)
    grid = dim3((unsigned)(8 * (((int)grid.y + 7) / 8) * (int)grid.x), 1);
  }
  int lrc = NPP_OK;
#define NPP_CONV_CASE(CT_, PT_, S_) if (pick.ct == CT_ && pick.pt == PT_ && pick.s == S_) lrc = conv_launch_mode<CT_, PT_, S_>(a, mode, grid, s)
  NPP_CONV_CASE(2, 2, 4); else NPP_CONV_CASE(2, 1, 4); else NPP_CONV_CASE(1, 1, 8); else NPP_CONV_CASE(1, 1, 4);
  else NPP_CONV_CASE(2, 2, 1); else NPP_CONV_CASE(2, 1, 1); else NPP_CONV_CASE(1, 1, 1);
#undef NPP_CONV_CASE
  if (lrc) return lrc;
  return check_launch("npp_conv3x3");
}

extern "C" int npp_maxpool2_fwd(const void* d_x, int N, int H, int W, int C, void* d_y, void* stream) {
  int rc = conv_geom_check(N, H, W, "npp_maxpool2_fwd");
  if (rc) return rc;
  if (!d_x || !d_y || C % 16 || H < 2 || W < 2) { set_error("npp_maxpool2_fwd: bad argument"); return NPP_ERR_ARG; }
  const int64_t nr = conv_npos_round(N, H / 2, W / 2), n = nr * (C / 8);
  hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const f16x8*)d_x,
                     N, H, W, conv_nposp(N, H, W), (f16x8*)d_y, conv_nposp(N, H / 2, W / 2), nr, C / 8);
  return check_launch("npp_maxpool2_fwd");
}

extern "C" int npp_maxpool2_bwd(const void* d_dy, const void* d_x, const void* d_addend, int N_total, int n_run, int H, int W,
                                int C, void* d_dz, void* stream) {
  int rc = conv_geom_check(N_total, H, W, "npp_maxpool2_bwd");
  if (rc) return rc;
  if (!d_dy || !d_x || !d_dz || C % 16 || H < 2 || W < 2 || n_run < 1 || n_run > N_total) { set_error("npp_maxpool2_bwd: bad argument"); return NPP_ERR_ARG; }
  const int64_t S = (int64_t)(H + 2) * (W + 2);
  const int64_t range = n_run == N_total ? conv_npos_round(N_total, H, W) : ((int64_t)n_run * S + kPosRound - 1) / kPosRound * kPosRound;
  const int64_t n = range * (C / 8);
  hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16x8*)d_dy,
                     (const f16x8*)d_x, (const bf16x8*)d_addend, n_run, H, W, conv_nposp(N_total, H, W),
                     conv_nposp(N_total, H / 2, W / 2), range, (bf16x8*)d_dz, C / 8);
  return check_launch("npp_maxpool2_bwd");
}

static int grad_in_impl(const float* d_df_nchw, const void* d_y, int N_total, int n_run, int C, int H, int W, void* d_dz, int as_f16,
                        int accumulate, const void* d_next_pack, int64_t next_pack_bytes, void* stream);
extern "C" int npp_trunk_grad_in(const float* d_df_nchw, const void* d_y, int N_total, int n_run, int C, int H, int W,
                                 void* d_dz, int as_f16, int accumulate, void* stream) {
  return grad_in_impl(d_df_nchw, d_y, N_total, n_run, C, H, W, d_dz, as_f16, accumulate, nullptr, 0, stream);
}
extern "C" int npp_trunk_grad_in_pf(const float* d_df_nchw, const void* d_y, int N_total, int n_run, int C, int H, int W,
                                    void* d_dz, int as_f16, int accumulate, const void* d_next_pack, int64_t next_pack_bytes,
                                    void* stream) {
  return grad_in_impl(d_df_nchw, d_y, N_total, n_run, C, H, W, d_dz, as_f16, accumulate, d_next_pack, next_pack_bytes, stream);
}
static int grad_in_impl(const float* d_df_nchw, const void* d_y, int N_total, int n_run, int C, int H, int W, void* d_dz, int as_f16,
                        int accumulate, const void* d_next_pack, int64_t next_pack_bytes, void* stream) {
  int rc = conv_geom_check(N_total, H, W, "npp_trunk_grad_in");
  if (rc) return rc;
  if (!d_df_nchw || !d_dz || C % 16 || n_run < 1 || n_run > N_total) { set_error("npp_trunk_grad_in: bad argument"); return NPP_ERR_ARG; }
  const int64_t S = (int64_t)(H + 2) * (W + 2);
  const int64_t range = n_run == N_total ? conv_npos_round(N_total, H, W) : ((int64_t)n_run * S + kPosRound - 1) / kPosRound * kPosRound;
  const int64_t n = range * (C / 8);
  if (as_f16)
    hipLaunchKernelGGL(trunk_grad_in_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_df_nchw,
                       (const f16x8*)d_y, n_run, C, H, W, conv_nposp(N_total, H, W), range, d_dz, accumulate, (const char*)d_next_pack,
                       d_next_pack ? next_pack_bytes : 0);
  else
    hipLaunchKernelGGL(trunk_grad_in_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_df_nchw,
                       (const f16x8*)d_y, n_run, C, H, W, conv_nposp(N_total, H, W), range, d_dz, accumulate, (const char*)d_next_pack,
                       d_next_pack ? next_pack_bytes : 0);
  return check_launch("npp_trunk_grad_in");
}

extern "C" int npp_trunk_export(const void* d_act, int N_total, int n_run, int C, int H, int W, float* d_out_nchw, int is_f16,
                                void* stream) {
  int rc = conv_geom_check(N_total, H, W, "npp_trunk_export");
  if (rc) return rc;
  if (!d_act || !d_out_nchw || C % 16 || n_run < 1 || n_run > N_total) { set_error("npp_trunk_export: bad argument"); return NPP_ERR_ARG; }
  const int64_t n = (int64_t)n_run * H * W * (C / 8);
  if (is_f16)
    hipLaunchKernelGGL(trunk_export_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_act,
                       n_run, C, H, W, conv_nposp(N_total, H, W), d_out_nchw);
  else
    hipLaunchKernelGGL(trunk_export_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_act,
                       n_run, C, H, W, conv_nposp(N_total, H, W), d_out_nchw);
  return check_launch("npp_trunk_export");
}
