// npp_mlp_fwd.hip -- K2: fused embedder + coordinate MLP forward (+ sigmoid), bf16 MFMA.
//
// Replaces, in one launch, what the reference does with a table gather, 13 F.linear
// GEMMs, 10 snake kernels, 3 cats and a sigmoid (NPP_completion/train.py:166-181 ->
// models/helpers.py:41-62 -> models/networks.py:56-95 / :145-173).
//
// Structure: one 256-thread workgroup owns 64 pixel rows (two
// 32-column batch tiles).  The GEMMs are computed TRANSPOSED, Z^T[n][b] = W[n][k] X^T[k][b],
// with bf16 MFMAs: the weights are the A operand (pre-packed in fragment
// order, streamed straight from L2 with one coalesced 1-KiB load per fragment), the
// activations are the B operand.  A 32x32 accumulator tile converted to bf16 IS the B
// operand of the next layer's k-steps (no transpose), so activations move between layers
// as 16-byte-per-lane fragments through LDS (slot layout of v_mfma_f32_32x32x16_bf16; the W = 256
// build issues the same work as v_mfma_f32_16x16x32_bf16 on gathered slots, see NPP_FWD_MFMA16 below): each of the 4 waves owns 2 of the 8 neuron
// tiles and needs the other waves' tiles as its next input.  The 462-wide embedding
// inputs are never materialised: per proposal the 22 warped coordinates of the 64 rows
// go to LDS (fp32) and sin/cos Fourier fragments are generated chunk-wise (8 k-steps)
// into a double-buffered LDS ring by all four waves, overlapped with the MFMAs of the
// previous chunk.  rgb_linear (128->3) is a VALU dot + wave shuffle + LDS reduction.
//
// In training mode the kernel also writes, as 16-byte fragments in the "W-format" line layout of
// npp_layout.h (two coalesced streaming stores per accumulator tile): the pre-activation z of
// every snake layer in fp16 (npp_mlp_bwd derives 1 + sin 2z from it, npp_mlp_wgrad derives the
// layer input snake(z) while staging), the linear outputs f1 / f2 in bf16, and the embedding
// slots in bf16.
//
// Algorithmic work: 2 * ((K+1)*462*256 + 11*256^2 + 384) FLOP per row;
// the zero padding of 462 -> 480 slots per proposal is not counted.
#include "npp_common.h"
#include <cstring>

namespace npp {

EmbedDev make_embed_dev(const npp_embed_cfg& c);
int check_embed_cfg(const npp_embed_cfg* c, const char* who);

// Waves per 64-row workgroup: 4 (each owns 2 of the 8 neuron tiles; <= 256 VGPRs, 2 waves per SIMD with two workgroups
// per CU) or 8 (1 tile each; <= 128 VGPRs, 4 waves per SIMD: more independent instruction streams to cover the LDS
// hand-off / barrier / weight-load latencies of the layer chain).  Same LDS image, same stash layout either way.
#ifndef NPP_FWD_WAVES
#define NPP_FWD_WAVES (NPP_WIDTH / 64)       // two neuron tiles per wave: 4 waves at W = 256, 8 at W = 512
#endif
#ifndef NPP_FWD_SLICED
#define NPP_FWD_SLICED 1
#endif
#ifndef NPP_FWD_PREFETCH_LINES      // lines of the weight pack each thread requests at kernel entry (0 = off): measured
                                    // in the complete iteration, same box: 0 lines 0.7150 ms, 2 lines 0.7074, 4 lines 0.7052
#define NPP_FWD_PREFETCH_LINES 4
#endif
#ifndef NPP_FWD_SLICED_PLAIN        // scheduling groups also in the plain parts that carry a chunk-0 generation
#define NPP_FWD_SLICED_PLAIN 1
#endif
// Experiment kept buildable (-DNPP_FWD_OVERLAP_PROLOGUE=1; parity tests green with it): the exposed prologue of an embedding
// pass (warped coordinates + barrier + chunk 0 + barrier = 25 % of a pass by the in-kernel stamps) moved into the MFMA gaps of
// a preceding plain part (L5: h part first, sV reused from L0; S: f1 part split around a barrier; proposal p + 1's coordinates
// under the last chunk of proposal p).  Measured (same-box A/B, 1024^2 K = 3 render): 2.550 -> 2.532 ms at W = 256 -- the
// stamps of ONE workgroup shrink, the CU's throughput does not, because the co-resident workgroup was already using the SIMDs
// during those prologues -- and 1.97 -> 2.31 ms at W = 512 (register spills).  Default off.  DESIGN.md section 4.
#ifndef NPP_FWD_OVERLAP_PROLOGUE
#define NPP_FWD_OVERLAP_PROLOGUE 0
#endif
constexpr bool kOverlapPro = NPP_FWD_OVERLAP_PROLOGUE != 0;
// MFMA shape of the forward chain.  1 (default at W = 256): v_mfma_f32_16x16x32_bf16 -- the same FLOPs per cycle as 32x32x16, but
// the chip holds a higher clock on it under load (MI355X_MICROARCH.md 'DVFS give-back' item 7).  Measured, same-box A/B, 1024^2
// K = 3 render: 2.510 -> 2.434 ms (-3 %; a timing-only substitution that kept the old operand registers had promised -8 %), the
// training forward unchanged (89-90 us: bound by its stores and the row quantisation); at W = 512 the render is unchanged and the
// training kernel reaches the 256-register limit (21 spills, 239 -> 264 us), so that build keeps 32x32x16.  Nothing outside this
// file changes: LDS fragments, weight pack and stash
// keep the 32x32x16 slot layout (16 bytes = 8 features of one row); the 16x16x32 operands GATHER slots from two consecutive
// k-steps (lane (n, g): row 16 rt + n, k-step 2 KB + (g >> 1), half g & 1), and an accumulator tile (4 neurons 4 g + r of 16
// rows per lane) becomes next-layer slots after ONE v_permlane32_swap per dword (the two halves of a slot sit in lanes l, l + 32).
// A wave's f32x16 accumulator [nt][bt] is reinterpreted as four 16 x 16 tiles: element 4 (2 mi + ri) + r = neuron
// 32 nt + 16 mi + 4 g + r of row 32 bt + 16 ri + (lane & 15).
#ifndef NPP_FWD_MFMA16
#define NPP_FWD_MFMA16 (NPP_WIDTH == 256)
#endif
constexpr bool kM16 = NPP_FWD_MFMA16 != 0;
typedef float f32x4 __attribute__((ext_vector_type(4)));
// neuron (within the wave's 32-neuron tile) of accumulator element idx for this lane
__device__ __forceinline__ int acc_nrow(int idx, const Lane& L) {
  return kM16 ? 16 * (idx >> 3) + 4 * (L.lane >> 4) + (idx & 3) : acc_row(idx, L.h);
}
constexpr int kWavesF = NPP_FWD_WAVES;
constexpr int kNTW = kNT / kWavesF;                             // neuron tiles per wave in 256-wide layers
constexpr int kThreads = 64 * kWavesF;
static_assert(kWavesF == 4 || kWavesF == 8, "4 or 8 waves per workgroup");
static_assert(kNT % kWavesF == 0 && kNT / 2 <= kWavesF, "every wave owns kNT / kWavesF neuron tiles; P's kNT / 2 tiles go one per wave");
constexpr int kFragBytes = 1024;                               // 64 lanes x 16 B
constexpr int kRegionBytes = kKSAct * kNB * kFragBytes;        // 32 KiB: 256 feats x 64 rows bf16
constexpr int kChunkKS = 8;                                    // k-steps per embedding chunk
constexpr int kChunkBytes = kChunkKS * kNB * kFragBytes;       // 16 KiB, two of them = one region
constexpr int kNChunks = (kKSEmb + kChunkKS - 1) / kChunkKS;   // 4 (8,8,8,6)
constexpr int kVRows = 22 + 8;                                 // rows 22..29 repeat rows 0..7 (see gen_emb_pair)
constexpr int kSmemV = kVRows * kRowTile * 4;                  // warped coords of one proposal
constexpr int kSmemE = (sizeof(EmbedDev) + 15) / 16 * 16;
// Branch-free embedding generation is driven by two small LDS tables built once per workgroup:
//  WarpEnt[K][22]: per warped coordinate cos/sin(theta), period, 1/period, phase, or the
//                linear form for the two normalised raw coordinates.
struct WarpEnt { float cs, sn, per, inv_per, phase, lin; };   // lin: value = t - 1 (normalised raw coordinate)
constexpr int kSmemWarp = NPP_MAX_K * 22 * (int)sizeof(WarpEnt);                // 3520
// the 4-wave rgb partial sums reuse region R0 after a barrier (everything else is dead by then)
constexpr int kSmemFwd = 2 * kRegionBytes + kSmemV + 2 * kRowTile * 4 + kSmemE + kSmemWarp;
// W = 256: two workgroups per CU (<= 80 KiB each); W = 512: the two 64-KiB regions leave room for one
constexpr int kWgPerCuF = kSmemFwd <= 80 * 1024 ? 2 : 1;
static_assert(kSmemFwd <= 160 * 1024, "LDS");
constexpr int kWavesPerSimdF = kWgPerCuF * kWavesF / 4;          // 2 (256 VGPRs) except the 8-wave W = 256 form (4, 128 VGPRs)


struct FwdArgs {
  const int32_t* coords;
  int64_t Bp;
  const bf16x8* wf;
  const float* params;
  float* pred;
  char* actF;        // nullable: W-format fragment arrays (npp_layout.h)
  // compatibility form (npp_mlp_fwd_emb): a materialised (Bp, K*462) fp32 embedding is the input, as
  // NPP_Net.forward(None, x_periodic) receives it (models/networks.py:56); out_act: helpers.py:55-60
  const float* emb;
  int64_t emb_ld;
  int32_t out_act;   // 0 raw, 1 sigmoid, 2 tanh
  // stacked launch (npp_mlp_fwd_stack): image m reads / writes every array at + m * stride; its embedder constants come
  // from estack[m] (device memory) instead of the kernel argument
  Stack S;
  const EmbedDev* estack;
  int64_t wf_stride16, params_stride, act_stride;     // 16-byte units / floats / bytes per image
};

struct EmbTabs {
  const float* freq_rev;   // LDS copy of EmbedDev::freq_rev[10]
  const WarpEnt* warp;
  const float* emb;        // compatibility form: materialised embedding rows instead of coordinates
  int64_t emb_ld;
};

// acc[nt][bt] <- bias of this wave's neuron tiles (row constants as the initial accumulator)
template <int NTW>
__device__ __forceinline__ void init_bias(f32x16 (&acc)[NTW][kNB], const float* __restrict__ bias, int nt0,
                                          const Lane& L) {
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
    f32x16 bv;
#pragma unroll
    for (int r = 0; r < 16; ++r) bv[r] = bias[(nt0 + nt) * 32 + acc_nrow(r, L)];
#pragma unroll
    for (int bt = 0; bt < kNB; ++bt) acc[nt][bt] = bv;
  }
}

// The same in two halves: the NEXT layer's bias values are fetched before the current layer's epilogue (their L2 latency
// hides under it and the barrier) and moved into the accumulators where init_bias used to load them.
template <int NTW> struct BiasPre { float v[NTW][16]; };
template <int NTW>
__device__ __forceinline__ void bias_fetch(BiasPre<NTW>& bp, const float* __restrict__ bias, int nt0, const Lane& L) {
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) bp.v[nt][r] = bias[(nt0 + nt) * 32 + acc_nrow(r, L)];
}
template <int NTW>
__device__ __forceinline__ void bias_apply(f32x16 (&acc)[NTW][kNB], const BiasPre<NTW>& bp) {
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
    for (int bt = 0; bt < kNB; ++bt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][bt][r] = bp.v[nt][r];
}

// The 22 warped coordinates (a1, models/embedder.py:110-133) of proposal p for the 64 rows ->
// sV[i][row] (fp32).  Table-driven: wave w takes i = w, w+4, ...; the entry is wave-uniform.
// ---- 16x16x32 form of the weight ring and of mma_ring (npp_common.h holds the 32x32x16 originals, which the backward chain uses)
// The ring keeps its 4 k-steps x NTW fragments of storage; fragment (k-step pair KB, 16-neuron tile mt) lives in
// w[2 (KB & 1) + (mt & 1)][mt >> 1] and is ONE buffer load that gathers 4 runs of 16 slots from the pack's k-steps 2 KB, 2 KB + 1.
template <int NTW, int NT>
__device__ __forceinline__ void wpair_load(WRing<NTW>& r, int half, wptr_t wp, int ks /*even*/, int nt0, const Lane& L) {
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  const int n16 = L.lane & 15, g = L.lane >> 4;
  const int voff = (n16 + 32 * (g & 1)) * 16 + (g >> 1) * NT * 1024;          // lane part: slot, and the second k-step of the pair
#pragma unroll
  for (int mt = 0; mt < 2 * NTW; ++mt) {
    const u32x4_t raw = __builtin_amdgcn_raw_buffer_load_b128(
        r.rsrc, voff + (mt & 1) * 256, (int)((wp + (uint32_t)((ks * NT + nt0 + (mt >> 1)) * 64)) * 16u), 0);
    r.w[2 * half + (mt & 1)][mt >> 1] = __builtin_bit_cast(bf16x8, raw);
  }
}
template <int NTW, int NT>
__device__ __forceinline__ void wring_fill16(WRing<NTW>& r, wptr_t wp, int nt0, const Lane& L) {
  wpair_load<NTW, NT>(r, 0, wp, 0, nt0, L);
  wpair_load<NTW, NT>(r, 1, wp, 2, nt0, L);
}
// B operand of k-step pair starting at LDS k-step ks (even), 16-row tile rt
__device__ __forceinline__ bf16x8 lds_frag16(const char* region, int ks, int rt, const Lane& L) {
  const int n16 = L.lane & 15, g = L.lane >> 4;
  return *(const bf16x8*)(region + ((ks + (g >> 1)) * kNB + (rt >> 1)) * 1024 + (16 * (rt & 1) + n16 + 32 * (g & 1)) * 16);
}
__device__ __forceinline__ void mfma16_sub(f32x16& c, int q, const bf16x8& a, const bf16x8& b) {
  f32x4 t = {c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]};
  t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, t, 0, 0, 0);
  c[4 * q] = t[0]; c[4 * q + 1] = t[1]; c[4 * q + 2] = t[2]; c[4 * q + 3] = t[3];
}
// Same contract as mma_ring (positions in 32x32x16 k-steps, all even here; hook called once per k-step position).
template <int KS0, int KS1, int KSREAL, int KSTOT, int NTW, int NT, typename Hook = NoHook, bool SLICED = false>
__device__ __forceinline__ void mma_ring16(f32x16 (&acc)[NTW][kNB], const char* region, int ks_lds0,
                                           wptr_t wp, wptr_t next_wp, int nt0,
                                           const Lane& L, WRing<NTW>& ring, const Hook& hook = Hook()) {
  static_assert(KS0 % 2 == 0 && KS1 % 2 == 0 && KSREAL % 2 == 0 && KSTOT % kRD == 0 && kRD == 4 && KSREAL <= KSTOT && KS1 <= KSTOT,
                "ring schedule (k-step pairs)");
  constexpr int KSE = KS1 < KSREAL ? KS1 : KSREAL;
  constexpr int RT = 2 * kNB;
  bf16x8 xn[RT];
  if (KS0 < KSE) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) xn[rt] = lds_frag16(region, ks_lds0, rt, L);
  }
#pragma unroll
  for (int ks = KS0; ks < KS1; ks += 2) {
    const int half = (ks >> 1) & 1;
    if (ks < KSREAL) {
      bf16x8 x[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) x[rt] = xn[rt];
      if (ks + 2 < KSE) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) xn[rt] = lds_frag16(region, ks_lds0 + ks + 2 - KS0, rt, L);
      }
#pragma unroll
      for (int mt = 0; mt < 2 * NTW; ++mt)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
          mfma16_sub(acc[mt >> 1][rt >> 1], 2 * (mt & 1) + (rt & 1), ring.w[2 * half + (mt & 1)][mt >> 1], x[rt]);
    }
    hook(ks - KS0);
    hook(ks + 1 - KS0);
    if (ks + kRD < KSREAL) wpair_load<NTW, NT>(ring, half, wp, ks + kRD, nt0, L);
    else if (ks + kRD >= KSTOT && next_wp != kNoW) wpair_load<NTW, NT>(ring, half, next_wp, ks + kRD - KSTOT, nt0, L);
    if (SLICED && ks < KSREAL) {
#pragma unroll
      for (int gq = 0; gq < 4 * NTW * kNB; ++gq) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, (NPP_SLICE_VALU_PER_GAP + 1) / 2, 0);
        if (gq & 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
    }
    asm volatile("" ::: "memory");
  }
}
#if NPP_FWD_MFMA16
#define MMA_RING mma_ring16
#define WRING_FILL(NTW_, NT_, ring_, wp_, nt0_, L_) wring_fill16<NTW_, NT_>(ring_, wp_, nt0_, L_)
#else
#define MMA_RING mma_ring
#define WRING_FILL(NTW_, NT_, ring_, wp_, nt0_, L_) wring_fill<NTW_, NT_>(ring_, wp_, nt0_, (L_).lane)
#endif

constexpr int kWarpPer = (22 + kWavesF - 1) / kWavesF;        // warped coordinates computed by one wave (6 or 3)
__device__ __forceinline__ void gen_warp(const WarpEnt* tw, int p, float* sV, const float* sY, const float* sX,
                                         const Lane& L, int q_only = -1) {
  const float y = sY[L.lane], x = sX[L.lane];
#pragma unroll
  for (int q = 0; q < kWarpPer; ++q) {
    const int i = L.wave + kWavesF * q;
    if (i < 22 && (q_only < 0 || q == q_only)) {
      const WarpEnt w = tw[p * 22 + i];
      // y*cos + x*sin rounded like the reference's two torch ops; the linear entries put the
      // normalised coordinate (x/W - .5)*2 into the same form (cs or sn = 2/res, bias = -1)
      const float t = __fadd_rn(__fmul_rn(y, w.cs), __fmul_rn(x, w.sn));
      const float qf = floorf(t * w.inv_per);
      float r = fmaf(-qf, w.per, t);                  // torch.remainder: result in [0, per)
      r = r < 0.0f ? r + w.per : r;
      r = r >= w.per ? r - w.per : r;
      const float sv = __builtin_amdgcn_sinf(fmaf(r, w.inv_per, w.phase));
      const float val = w.lin != 0.0f ? t - 1.0f : sv;
      sV[i * kRowTile + L.lane] = val;
      if (i < kVRows - 22) sV[(22 + i) * kRowTile + L.lane] = val;      // wrap rows (wave-uniform branch)
    }
  }
}

// J < jb ? a : b on the scalar unit (hipcc turns the C++ select of two uniform floats into v_cndmask per value)
template <int J>
__device__ __forceinline__ float scalar_pick(int jb, int a, int b) {
  int r;
  asm("s_cmp_gt_i32 %1, %2\n\ts_cselect_b32 %0, %3, %4" : "=s"(r) : "s"(jb), "n"(J), "s"(a), "s"(b) : "scc");
  return __builtin_bit_cast(float, r);
}
template <int JP>
__device__ __forceinline__ void gen_emb_quad(const float* v, int jb, int fr0, int fr1, float ph, bf16x8 (&f)[kNB]) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  const float fa = scalar_pick<2 * JP>(jb, fr0, fr1), fb = scalar_pick<2 * JP + 1>(jb, fr0, fr1);
#pragma unroll
  for (int bt = 0; bt < kNB; ++bt) {
    f32x2 x;
    x[0] = __builtin_amdgcn_sinf(fmaf(v[(2 * JP) * kRowTile + bt * 32], fa, ph));
    x[1] = __builtin_amdgcn_sinf(fmaf(v[(2 * JP + 1) * kRowTile + bt * 32], fb, ph));
    const bf16x2 pk = __builtin_convertvector(x, bf16x2);
    f[bt][2 * JP] = pk[0];
    f[bt][2 * JP + 1] = pk[1];
  }
}
// The two embedding fragments (batch tiles 0, 1) of k-step ks.  ks is wave-uniform, so the slot ->
// (Fourier frequency, warped coordinate) map of npp_layout.h emb_col() is scalar arithmetic and the
// frequency is one broadcast LDS read; per value the vector ALU does fma + v_sin + half a pack.
// Lane-half 1 adds a quarter revolution (cos).  Padding slots (t >= 220 of k-step 27) receive
// some finite sin value: their packed weights are zero and npp_mlp_wgrad drops their columns.
__device__ __forceinline__ void gen_emb_pair(const float* sFr, const float* sV, int ks, bf16x8 (&f)[kNB], const Lane& L) {
  if (ks < 28) {
    // the 8 slots t = 8 ks + j span at most two frequencies: slots j >= jb belong to fj0 + 1 and their coordinate index
    // wraps to i0 + j - 22, which the 8 wrap rows of sV turn into the plain row i0 + j.  So per k-step: two frequency
    // reads, one row base; per value: fma + v_sin + half a pack + half a ds_read2 (immediate offsets only).
    const int t0 = 8 * ks;
    const int fj0 = (t0 * 2979) >> 16;           // t0 / 22, exact for t0 < 240
    const int i0 = t0 - 22 * fj0, jb = 22 - i0;
    const int fj1 = fj0 + 1 > NPP_N_FREQ - 1 ? NPP_N_FREQ - 1 : fj0 + 1;
    const int fr0 = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sFr[fj0]));      // scalar registers
    const int fr1 = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sFr[fj1]));
    const float* v = sV + i0 * kRowTile + L.b;
    const float ph = L.h ? 0.25f : 0.0f;
    gen_emb_quad<0>(v, jb, fr0, fr1, ph, f);
    gen_emb_quad<1>(v, jb, fr0, fr1, ph, f);
    gen_emb_quad<2>(v, jb, fr0, fr1, ph, f);
    gen_emb_quad<3>(v, jb, fr0, fr1, ph, f);
  } else {                                       // identity block: v_0..v_15 | v_16..v_21, 0...
    const float* vrow = sV + L.b;
    const int i0 = 16 * (ks - 28) + 8 * L.h;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool ok = i0 + j < 22;
      const int i = ok ? i0 + j : 0;
#pragma unroll
      for (int bt = 0; bt < kNB; ++bt) {
        const float v = vrow[i * kRowTile + bt * 32];
        f[bt][j] = (__bf16)(ok ? v : 0.0f);
      }
    }
  }
}

// Same two fragments read from a materialised reference-layout embedding row (compatibility form):
// slot -> reference column through npp_layout.h emb_col(), padding slots are zero.
__device__ __forceinline__ void load_emb_pair(const float* __restrict__ emb, int64_t ld, int64_t row0, int p, int ks,
                                              bf16x8 (&f)[kNB], const Lane& L) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int col = emb_col(ks, L.h, j);
#pragma unroll
    for (int bt = 0; bt < kNB; ++bt) {
      const float v = col >= 0 ? emb[(row0 + bt * 32 + L.b) * ld + p * kE + col] : 0.0f;
      f[bt][j] = (__bf16)v;
    }
  }
}

// The generator cut into slices of one QUAD (slots 2 JP, 2 JP + 1 of both batch tiles = 4 values: one ds_read2 per slot,
// 4 fma, 4 v_sin, 2 packs) so that MMA_RING<..., SLICED> can thread it through the MFMAs of one schedule position.  The LDS
// reads of a quad are issued one position before its arithmetic.  Only for chunks whose k-steps are all regular
// (k-step < 28, hence < kKSEmb): no branch anywhere, so a position stays one scheduling region.
// STORE_EMB: 0 no stash, 1 bf16 W-format, 2 fp8 W8-format (emb_base / lane_off then address the 8-bit array)
template <int STORE_EMB, int KPER>
struct SlicedGen {
  const float* sFr;
  const float* vlane;      // sV + (lane & 31)
  char* dst;               // LDS chunk buffer the fragments go to
  char* emb_base;          // stash (uniform part) or null
  uint32_t lane_off;
  int lane, ksl0, ks0;     // this wave's first k-step of the chunk: local index, global index
  float ph;
  mutable const float* v;
  mutable int fr0, fr1, jb;
  mutable float val[2][kNB];
  mutable bf16x8 f[kNB];

  __device__ __forceinline__ void setup(int q2) const {
    const int t0 = 8 * (ks0 + q2);
    const int fj0 = (t0 * 2979) >> 16;
    const int i0 = t0 - 22 * fj0;
    jb = 22 - i0;
    const int fj1 = fj0 + 1 > NPP_N_FREQ - 1 ? NPP_N_FREQ - 1 : fj0 + 1;
    fr0 = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sFr[fj0]));
    fr1 = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sFr[fj1]));
    v = vlane + i0 * kRowTile;
  }
  template <int JP> __device__ __forceinline__ void issue() const {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int bt = 0; bt < kNB; ++bt) val[jj][bt] = v[(2 * JP + jj) * kRowTile + bt * 32];
  }
  template <int JP> __device__ __forceinline__ void compute() const {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const float fa = scalar_pick<2 * JP>(jb, fr0, fr1), fb = scalar_pick<2 * JP + 1>(jb, fr0, fr1);
#pragma unroll
    for (int bt = 0; bt < kNB; ++bt) {
      f32x2 x;
      x[0] = __builtin_amdgcn_sinf(fmaf(val[0][bt], fa, ph));
      x[1] = __builtin_amdgcn_sinf(fmaf(val[1][bt], fb, ph));
      const bf16x2 pk = __builtin_convertvector(x, bf16x2);
      f[bt][2 * JP] = pk[0];
      f[bt][2 * JP + 1] = pk[1];
    }
  }
  __device__ __forceinline__ void finish(int q2) const {
#pragma unroll
    for (int bt = 0; bt < kNB; ++bt) {
      lds_store_frag(dst, ksl0 + q2, bt, lane, f[bt]);
      if (STORE_EMB == 1) stash_store(emb_base + (uint32_t)wfmt_unit(kKSEmb, 0, ks0 + q2, bt, 0, 0) + lane_off, f[bt]);
      if (STORE_EMB == 2) stash8_store(emb_base + (uint32_t)wfmt8_unit(kKSEmb, 0, ks0 + q2, 32 * bt, 0) + lane_off, pack8_fp8_bf16(f[bt]));
    }
  }
  template <int JP> __device__ __forceinline__ void quad(int q2) const {
    compute<JP>();
    if (JP == 3) {
      finish(q2);
      if (q2 + 1 < KPER) { setup(q2 + 1); issue<0>(); }
    } else {
      issue<(JP + 1) & 3>();
    }
  }
  __device__ __forceinline__ void prologue() const { setup(0); issue<0>(); }
  // schedule position pos (0..7) of the chunk being multiplied: KPER = 2 -> quad pos & 3 of k-step pos >> 2;
  // KPER = 1 -> quad pos >> 1 at the even positions
  __device__ __forceinline__ void operator()(int pos) const {
    const int qi = KPER == 2 ? pos : (pos >> 1);
    if ((KPER == 1 && (pos & 1)) || qi >= 4 * KPER) return;
    const int q2 = qi >> 2;
    switch (qi & 3) {
      case 0: quad<0>(q2); break;
      case 1: quad<1>(q2); break;
      case 2: quad<2>(q2); break;
      default: quad<3>(q2); break;
    }
  }
};

// Accumulate one proposal's 30 embedding k-steps.  lds_ring = 32 KiB LDS (two 16 KiB chunk
// buffers).  Caller guarantees sV is free to overwrite and lds_ring is free; on return every
// wave has passed a barrier after its last lds_ring / sV read.  The weight ring holds the
// first 4 k-steps on entry and the first 4 k-steps of next_wp on return.
//  have_warp: sV already holds proposal p's warped coordinates; have_chunk0: chunk 0 of the LDS ring was generated by the
//  caller (sliced_chunk0 below, in the gaps of a preceding plain part) and a barrier has been passed since;
//  next_warp_p >= 0: the warped coordinates of THAT proposal are computed in the gaps of the last chunk's MFMAs (sV is idle
//  then), so the next pass starts with have_warp.  All three are workgroup-uniform.
template <int STORE_EMB, int NTW, int NT, bool EMB_IN = false>
__device__ __forceinline__ void mma_embedding(f32x16 (&acc)[NTW][kNB], const EmbTabs& e, int p, char* lds_ring,
                                              float* sV, const float* sY, const float* sX,
                                              wptr_t wp, wptr_t next_wp,
                                              int nt0, char* actF, int wg, const Lane& L, WRing<NTW>& ring,
                                              bool have_warp = false, bool have_chunk0 = false, int next_warp_p = -1) {
  if (!EMB_IN && !have_warp) {
    gen_warp(e.warp, p, sV, sY, sX, L);
    wg_barrier();
  } else if (EMB_IN) {
    wg_barrier();
  }
  // stash address = uniform (array, workgroup, k-step, batch tile) part + this lane's 32-bit offset (npp_layout.h wfmt_unit)
  // (STORE_EMB == 2: actF is the base of the 8-bit region, npp_layout.h act8_region_base)
  char* emb_base = STORE_EMB == 2 ? actF + wfmt8_array_base(kActKsEmb0 + p * kKSEmb, L.n_wg) + wfmt8_unit(kKSEmb, wg, 0, 0, 0)
                   : STORE_EMB ? actF + wfmt_array_base(kActKsEmb0 + p * kKSEmb, L.n_wg) + wfmt_unit(kKSEmb, wg, 0, 0, 0, 0) : nullptr;
  const uint32_t lane_off = STORE_EMB == 2 ? (uint32_t)wfmt8_unit(kKSEmb, 0, 0, L.b, L.h) : (uint32_t)wfmt_unit(kKSEmb, 0, 0, 0, L.b, L.h);
  // this wave's k-step q (0, 1) of chunk c: generate, hand to the LDS ring, stash for wgrad
  constexpr int kPer = kChunkKS / kWavesF;                       // k-steps of a chunk generated by one wave (2 or 1)
  auto gen_pair = [&](int c, int q) {
    const int ksl = kPer * L.wave + q, ks = kChunkKS * c + ksl;    // wave-uniform
    if (ks < kKSEmb) {
      bf16x8 f[kNB];
      if (EMB_IN) load_emb_pair(e.emb, e.emb_ld, (int64_t)wg * kRowTile, p, ks, f, L);
      else gen_emb_pair(e.freq_rev, sV, ks, f, L);
#pragma unroll
      for (int bt = 0; bt < kNB; ++bt) {
        lds_store_frag(lds_ring + (c & 1) * kChunkBytes, ksl, bt, L.lane, f[bt]);
        if (STORE_EMB == 1) stash_store(emb_base + (uint32_t)wfmt_unit(kKSEmb, 0, ks, bt, 0, 0) + lane_off, f[bt]);
        if (STORE_EMB == 2) stash8_store(emb_base + (uint32_t)wfmt8_unit(kKSEmb, 0, ks, 32 * bt, 0) + lane_off, pack8_fp8_bf16(f[bt]));
      }
    }
  };
  if (!have_chunk0) {
    gen_pair(0, 0);
    if (kPer == 2) gen_pair(0, 1);
    wg_barrier();
  }
  // chunk c is multiplied while chunk c+1 is generated: this wave's two k-steps of the next chunk
  // are produced at schedule positions 0 and 4 of the current one, so their LDS reads / v_sin /
  // stores issue between the MFMAs.
  auto hook_for = [&](int c_next) {
    return [&, c_next](int pos) {
      if (pos == 0) gen_pair(c_next, 0);
      else if (pos == 4 && kPer == 2) gen_pair(c_next, 1);
    };
  };
  // chunks 1 and 2 hold regular k-steps only (8 .. 23 < 28): their generation is sliced into the MFMA gaps
  auto sliced_for = [&](int c_next) {
    SlicedGen<STORE_EMB, kPer> g;
    g.sFr = e.freq_rev; g.vlane = sV + L.b; g.dst = lds_ring + (c_next & 1) * kChunkBytes; g.emb_base = emb_base;
    g.lane_off = lane_off; g.lane = L.lane; g.ksl0 = kPer * L.wave; g.ks0 = kChunkKS * c_next + kPer * L.wave;
    g.ph = L.h ? 0.25f : 0.0f;
    return g;
  };
#if NPP_FWD_SLICED
  if (!EMB_IN) {
    const auto g1 = sliced_for(1);
    g1.prologue();
    MMA_RING<0, 8, kKSEmb, 32, NTW, NT, decltype(g1), true>(acc, lds_ring, 0, wp, next_wp, nt0, L, ring, g1);
    wg_barrier();
    const auto g2 = sliced_for(2);
    g2.prologue();
    MMA_RING<8, 16, kKSEmb, 32, NTW, NT, decltype(g2), true>(acc, lds_ring + kChunkBytes, 0, wp, next_wp, nt0, L, ring, g2);
    wg_barrier();
  } else
#endif
  {
    MMA_RING<0, 8, kKSEmb, 32, NTW, NT>(acc, lds_ring, 0, wp, next_wp, nt0, L, ring, hook_for(1));
    wg_barrier();
    MMA_RING<8, 16, kKSEmb, 32, NTW, NT>(acc, lds_ring + kChunkBytes, 0, wp, next_wp, nt0, L, ring, hook_for(2));
    wg_barrier();
  }
  MMA_RING<16, 24, kKSEmb, 32, NTW, NT>(acc, lds_ring, 0, wp, next_wp, nt0, L, ring, hook_for(3));
  wg_barrier();
  if (!EMB_IN && next_warp_p >= 0) {
    auto warp_hook = [&](int pos) { if (pos < kWarpPer) gen_warp(e.warp, next_warp_p, sV, sY, sX, L, pos); };
    MMA_RING<24, 32, kKSEmb, 32, NTW, NT>(acc, lds_ring + kChunkBytes, 0, wp, next_wp, nt0, L, ring, warp_hook);
  } else {
    MMA_RING<24, 32, kKSEmb, 32, NTW, NT>(acc, lds_ring + kChunkBytes, 0, wp, next_wp, nt0, L, ring);
  }
  wg_barrier();
}

// Chunk 0 of proposal p's embedding generated in the MFMA gaps of a PLAIN part (KS0..KS1 of the 16 k-steps of `region`) that
// precedes the embedding part of the same layer: replaces the exposed prologue of mma_embedding.  sV must hold p's warped
// coordinates.  The caller passes a barrier before the embedding part reads the chunk.
template <int STORE_EMB, int KS0, int KS1, int NTW, int NT>
__device__ __forceinline__ void mma_plain_gen_chunk0(f32x16 (&acc)[NTW][kNB], const char* region, const EmbTabs& e, int p,
                                                     char* lds_ring, const float* sV, wptr_t wp, wptr_t next_wp, int nt0,
                                                     char* actF, int wg, const Lane& L, WRing<NTW>& ring) {
  constexpr int kPer = kChunkKS / kWavesF;
  SlicedGen<STORE_EMB, kPer> g;
  g.sFr = e.freq_rev; g.vlane = sV + L.b; g.dst = lds_ring;
  g.emb_base = STORE_EMB == 2 ? actF + wfmt8_array_base(kActKsEmb0 + p * kKSEmb, L.n_wg) + wfmt8_unit(kKSEmb, wg, 0, 0, 0)
               : STORE_EMB ? actF + wfmt_array_base(kActKsEmb0 + p * kKSEmb, L.n_wg) + wfmt_unit(kKSEmb, wg, 0, 0, 0, 0) : nullptr;
  g.lane_off = STORE_EMB == 2 ? (uint32_t)wfmt8_unit(kKSEmb, 0, 0, L.b, L.h) : (uint32_t)wfmt_unit(kKSEmb, 0, 0, 0, L.b, L.h);
  g.lane = L.lane; g.ksl0 = kPer * L.wave; g.ks0 = kPer * L.wave;
  g.ph = L.h ? 0.25f : 0.0f;
  g.prologue();
  MMA_RING<KS0, KS1, kKSAct, kKSAct, NTW, NT, decltype(g), NPP_FWD_SLICED_PLAIN != 0>(acc, region, KS0, wp, next_wp, nt0, L, ring, g);
}

// Epilogue of a 256-wide (NTW=2 per wave) or 128-wide (NTW=1) layer.
//  SNAKE: apply x + sin^2 x; else linear.   out: LDS region that receives the bf16
//  fragments (k-step 2*ntile+s of the next layer), may be null.
//  TRAIN: stash z (fp16, snake layers) or the linear output (bf16) in W-format.
// 16x16x32 accumulator layout: the two halves (elements 0..3 | 4..7) of a next-layer slot sit in lanes l and l + 32, for the row
// tiles ri = 0 and ri = 1 alike: one v_permlane32_swap per dword gives the lower lanes the whole slot of tile ri = 0 and the upper
// lanes that of ri = 1 (cdna_hip_programming.md T21), i.e. slot 16 (g >> 1) + (lane & 15) + 32 (g & 1) of fragment
// (k-step 2 ntg + mi, batch tile bt): one 16-byte LDS / stash store per lane and fragment, as in the 32x32x16 form.
template <typename E2>   // E2 = 2-element vector of __bf16 or _Float16
__device__ __forceinline__ void slot_from_halves(const float (&lo)[4], const float (&hi)[4], uint32_t (&slot)[4]) {
  typedef float f32x2v __attribute__((ext_vector_type(2)));
  uint32_t pa[2], pb[2];
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    pa[d] = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2v{lo[2 * d], lo[2 * d + 1]}), E2));
    pb[d] = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2v{hi[2 * d], hi[2 * d + 1]}), E2));
    const auto r = __builtin_amdgcn_permlane32_swap(pa[d], pb[d], false, false);
    pa[d] = r[0];
    pb[d] = r[1];
  }
  slot[0] = pa[0]; slot[1] = pa[1]; slot[2] = pb[0]; slot[3] = pb[1];
}

// TRAIN: 0 inference, 1 the 16-bit stash, 2 (npp_tune "stash8") two W8 arrays per layer instead: the layer's output (snake(z) or
// the linear output) as fp8 units in `stash8_array` -- what npp_mlp_wgrad8 contracts -- and, for a snake layer, snake'(z) = 1 +
// sin 2z as unsigned bytes (npp_common.h kSd8Scale) in `stash_array` -- what npp_mlp_bwd multiplies by: 2 bytes per element
// where the 16-bit stash writes the 2-byte fp16 z that both consumers re-derive from
__device__ __forceinline__ uint32_t pack4_fp8(const float (&v)[4]) {
  int x = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
  return (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], x, true);
}
template <bool SNAKE, int TRAIN, int NTW>
__device__ __forceinline__ void epilogue(f32x16 (&acc)[NTW][kNB], char* out, int nt0, int ntl /*tiles in layer*/,
                                         char* stash_array, int wg, const Lane& L, bf16x8 (*keep)[kNB][2] = nullptr,
                                         char* stash8_array = nullptr) {
  typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
  typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
  const int g16 = L.lane >> 4;
  const int slot_lane = 16 * (g16 >> 1) + (L.lane & 15) + 32 * (g16 & 1);       // kM16: the slot this lane ends up holding
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
    const int ntg = nt0 + nt;
#pragma unroll
    for (int bt = 0; bt < kNB; ++bt) {
      if (TRAIN == 1 && SNAKE) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if (kM16) {
            float lo[4], hi[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { lo[r] = acc[nt][bt][8 * s + r]; hi[r] = acc[nt][bt][8 * s + 4 + r]; }
            uint32_t sl[4];
            slot_from_halves<f16x2v>(lo, hi, sl);
            stash_store(stash_array + wfmt_unit(2 * ntl, wg, 2 * ntg + s, bt, slot_lane & 31, slot_lane >> 5), __builtin_bit_cast(f16x8, sl));
          } else {
            stash_store(stash_array + wfmt_unit(2 * ntl, wg, 2 * ntg + s, bt, L.b, L.h), pack_acc_f16(acc[nt][bt], s));
          }
        }
      }
      // snake on aligned register pairs: v_pk_mul (z / 2pi), 2 x v_sin, v_pk_fma (s*s + z), v_cvt_pk
      f32x16 a, sd;                 // sd (stash8, snake): 127.5 * snake'(z) = 127.5 + 255 sin z cos z
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        f32x2 z = {acc[nt][bt][r], acc[nt][bt][r + 1]};
        if (SNAKE) {
          const f32x2 rev = z * kInv2Pi;
          const f32x2 sn = {__builtin_amdgcn_sinf(rev[0]), __builtin_amdgcn_sinf(rev[1])};
          if (TRAIN == 2) {
            const f32x2 cs = {__builtin_amdgcn_cosf(rev[0]), __builtin_amdgcn_cosf(rev[1])};
            const f32x2 d = __builtin_elementwise_fma(sn * cs, (f32x2){2.0f * kSd8Scale, 2.0f * kSd8Scale}, (f32x2){kSd8Scale, kSd8Scale});
            sd[r] = d[0];
            sd[r + 1] = d[1];
          }
          z = __builtin_elementwise_fma(sn, sn, z);
        }
        a[r] = z[0];
        a[r + 1] = z[1];
      }
      acc[nt][bt] = a;   // callers that need the fp32 activation (P -> rgb) read it back
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 f;
        int sl_lane = L.lane;
        if (kM16) {          // s = mi: elements 8 mi + 0..3 belong to row tile ri = 0, 8 mi + 4..7 to ri = 1
          float lo[4], hi[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) { lo[r] = a[8 * s + r]; hi[r] = a[8 * s + 4 + r]; }
          uint32_t sl[4];
          slot_from_halves<bf16x2v>(lo, hi, sl);
          f = __builtin_bit_cast(bf16x8, sl);
          sl_lane = slot_lane;
        } else {
          f = pack_acc(a, s);
        }
        if (out) lds_store_frag(out, 2 * ntg + s, bt, sl_lane, f);
        if (keep) keep[nt][bt][s] = f;
        if (TRAIN == 1 && !SNAKE) stash_store(stash_array + wfmt_unit(2 * ntl, wg, 2 * ntg + s, bt, sl_lane & 31, sl_lane >> 5), f);
        if (TRAIN == 2) {
          u32x2 u;
          if (kM16) {
            float lo[4], hi[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { lo[r] = a[8 * s + r]; hi[r] = a[8 * s + 4 + r]; }
            const auto r2 = __builtin_amdgcn_permlane32_swap(pack4_fp8(lo), pack4_fp8(hi), false, false);
            u = u32x2{r2[0], r2[1]};
          } else {
            u = pack8_fp8(a[8 * s], a[8 * s + 1], a[8 * s + 2], a[8 * s + 3], a[8 * s + 4], a[8 * s + 5], a[8 * s + 6], a[8 * s + 7]);
          }
          stash8_store(stash8_array + wfmt8_unit(2 * ntl, wg, 2 * ntg + s, 32 * bt + (sl_lane & 31), sl_lane >> 5), u);
          if (SNAKE) {
            u32x2 d;
            if (kM16) {
              const auto r2 = __builtin_amdgcn_permlane32_swap(pack4_u8(sd[8 * s], sd[8 * s + 1], sd[8 * s + 2], sd[8 * s + 3]),
                                                               pack4_u8(sd[8 * s + 4], sd[8 * s + 5], sd[8 * s + 6], sd[8 * s + 7]), false, false);
              d = u32x2{r2[0], r2[1]};
            } else {
              d = u32x2{pack4_u8(sd[8 * s], sd[8 * s + 1], sd[8 * s + 2], sd[8 * s + 3]), pack4_u8(sd[8 * s + 4], sd[8 * s + 5], sd[8 * s + 6], sd[8 * s + 7])};
            }
            stash8_store(stash_array + wfmt8_unit(2 * ntl, wg, 2 * ntg + s, 32 * bt + (sl_lane & 31), sl_lane >> 5), d);
          }
        }
      }
    }
  }
}

// STACK: the stacked-launch form (npp_mlp_fwd_stack).  A template parameter, not a run-time branch: the plain launch keeps reading
// its pointers from the kernel-argument segment where it needs them -- as locals they cost the training forward 3-4 us of SGPR
// pressure in a kernel that already spills 40 of them.
// ACT: the output nonlinearity is read from A_.out_act (npp_mlp_fwd_act: tanh / raw); false = the sigmoid, compiled in -- a template
// parameter for the same reason as STACK (the run-time test cost the default launch 0.9 us in a same-box A/B)
template <int TRAIN, bool MULTI, bool EMB_IN = false, bool STACK = false, bool ACT = false>
__global__ __launch_bounds__(kThreads, kWavesPerSimdF) void mlp_fwd_kernel(FwdArgs A_, EmbedDev e_arg, NetDesc d) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (TRAIN == 2) set_fp16_ovfl();
  int img_ = 0, wg = blockIdx.x, xslot_ = blockIdx.x >> 3, xcount_ = ((int)gridDim.x + 7) >> 3;
  if (STACK && !stack_decode(A_.S, img_, wg, xslot_, xcount_)) return;
  const int n_wg_ = STACK ? A_.S.n_items : (int)gridDim.x;
  // per-image offsets of a stacked launch (compile-time zeros otherwise)
  const int64_t o_coords = STACK ? (int64_t)img_ * A_.Bp * 2 : 0, o_wf = STACK ? (int64_t)img_ * A_.wf_stride16 : 0;
  const int64_t o_params = STACK ? (int64_t)img_ * A_.params_stride : 0, o_pred = STACK ? (int64_t)img_ * A_.Bp * 3 : 0;
  const int64_t o_act = STACK ? (int64_t)img_ * A_.act_stride : 0;
#define s_coords (A_.coords + o_coords)
#define s_wf (A_.wf + o_wf)
#define s_params (A_.params + o_params)
#define s_pred (A_.pred + o_pred)
#define s_actF (A_.actF + o_act)
  char* R0 = smem;
  char* R1 = smem + kRegionBytes;
  float* sV = (float*)(smem + 2 * kRegionBytes);
  float* sY = sV + kVRows * kRowTile;
  float* sX = sY + kRowTile;
  // The embedder constants are indexed with run-time (proposal, orientation, offset)
  // indices: keep them in LDS, copied with compile-time indices so the by-value kernel
  // argument never needs a scratch copy.
  EmbedDev& ed = *(EmbedDev*)(sX + kRowTile);
  WarpEnt* tWarp = (WarpEnt*)((char*)&ed + kSmemE);
  float* sRGB = (float*)R0;             // [4 waves][64 rows][3], reused after the last barrier
  if (STACK) {
    // the image's constants come from global memory: one dword per thread, ONE load latency (thread 0 alone walked them in a rolled
    // loop of dependent load -> LDS store round trips: ~0.2 us each x sizeof / 4 on the critical path of every workgroup)
    uint32_t* dst = (uint32_t*)&ed;
    const uint32_t* src = (const uint32_t*)(A_.estack + img_);
    for (int i = threadIdx.x; i < (int)(sizeof(EmbedDev) / 4); i += kThreads) dst[i] = src[i];
  } else if (threadIdx.x == 0) {
    uint32_t* dst = (uint32_t*)&ed;
    const uint32_t* src = (const uint32_t*)&e_arg;
#pragma unroll
    for (int i = 0; i < (int)(sizeof(EmbedDev) / 4); ++i) dst[i] = src[i];
  }
  wg_barrier();
  for (int wi = threadIdx.x; wi < ed.K * 22; wi += kThreads) {
    const int p = wi / 22, i = wi - p * 22, ori = i >= 11, ii = ori ? i - 11 : i;
    WarpEnt w{0.0f, 0.0f, 1.0f, 1.0f, 0.0f, 0.0f};
    if (ii == 0) {                                          // (x / res[1] - 0.5) * 2, (y / res[0] - 0.5) * 2
      w.lin = 1.0f;
      if (ori) w.cs = 2.0f * ed.inv_h; else w.sn = 2.0f * ed.inv_w;
    } else {
      const int o = (ii - 1) >> 1;
      w.cs = ed.cs[p][ori]; w.sn = ed.sn[p][ori];
      w.per = ed.per[p][ori][o]; w.inv_per = 1.0f / w.per;
      w.phase = ((ii - 1) & 1) ? 0.25f : 0.0f;
    }
    tWarp[wi] = w;
  }
  const EmbTabs e{ed.freq_rev, tWarp, A_.emb, A_.emb_ld};
  Lane L;
  L.tid = threadIdx.x;
  L.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  L.lane = threadIdx.x & 63;
  L.b = L.lane & 31;
  L.h = L.lane >> 5;
  L.n_wg = n_wg_; L.xslot = xslot_; L.xcount = xcount_;
  const int64_t row0 = (int64_t)wg * kRowTile;
  const int64_t Bp = A_.Bp;
  const float* P = s_params;
  const int nt0 = kNTW * L.wave;         // this wave's neuron tiles in 256-wide layers

  if (!EMB_IN && L.tid < kRowTile) {
    const int2 c = ((const int2*)s_coords)[row0 + L.tid];
    sY[L.tid] = (float)c.x;              // (row=y, col=x)
    sX[L.tid] = (float)c.y;
  }
  wg_barrier();

  // (stash8: the snake'(z) byte arrays live where the 16-bit layout has its fp16 z arrays, at W8 size)
  auto arow = [&](int idx) -> char* {
    return TRAIN == 2 ? s_actF + wfmt8_array_base(idx * kKSAct, L.n_wg) : TRAIN ? s_actF + wfmt_array_base(idx * kKSAct, L.n_wg) : nullptr;
  };
  // stash8: the 8-bit region behind the 16-bit one (same k-step table), and what the embedding passes stash into
  auto arow8 = [&](int idx) -> char* { return TRAIN == 2 ? s_actF + act8_region_base(d.K, L.n_wg) + wfmt8_array_base(idx * kKSAct, L.n_wg) : nullptr; };
#define s_actE (TRAIN == 2 ? s_actF + act8_region_base(d.K, L.n_wg) : s_actF)

  f32x16 acc[kNTW][kNB];
  WRing<kNTW> ring;                           // weight-stream register ring, live across layers
  WRing<1> ringp;                          // same for P (one neuron tile per wave)
  ring.rsrc = ringp.rsrc = make_wrsrc(s_wf, d.wf_total16);
  constexpr wptr_t U = kNT * 64;           // 16-byte units per k-step of a 256-wide layer
  constexpr wptr_t UP = (kNT / 2) * 64;
  constexpr int A = kKSAct;                // 16 k-steps per 256 features
  auto wl = [&](int l) -> wptr_t { return (wptr_t)d.wf_off[l]; };

  // In a training step the weight pack was re-written by npp_pack_weights just before this launch and is in no XCD's L2 for
  // reading: all ~52 workgroups of an XCD walk the layers together, so every k-step of every layer would be a first touch
  // (the ring looks 4 k-steps = ~1 us ahead, less than a miss).  The workgroups of each XCD request the WHOLE pack up front,
  // one dword per 128-byte line, while the embedding prologue runs; the values are consumed after the last layer.
  uint32_t pfv[NPP_FWD_PREFETCH_LINES > 0 ? NPP_FWD_PREFETCH_LINES : 1] = {0};
  {
    const int64_t lines = ((int64_t)d.wf_total16 * 16 + 127) / 128;
    const int64_t per_xcd_threads = (int64_t)L.xcount * kThreads;
    int64_t line = (int64_t)L.xslot * kThreads + L.tid;
#pragma unroll
    for (int q = 0; q < NPP_FWD_PREFETCH_LINES; ++q, line += per_xcd_threads)
      pfv[q] = line < lines ? *(const volatile uint32_t*)((const char*)s_wf + line * 128) : 0u;
  }
  // ---- L0: emb(p0) -> 256, snake.  LDS ring = R1, out -> R0
  WRING_FILL(kNTW, kNT, ring, wl(L0), nt0, L);
  init_bias<kNTW>(acc, P + d.b_off[L0], nt0, L);
  mma_embedding<TRAIN, kNTW, kNT, EMB_IN>(acc, e, 0, R1, sV, sY, sX, wl(L0), wl(L1), nt0, s_actE, wg, L, ring);
  BiasPre<kNTW> bn;
  BiasPre<1> bnp;
  bias_fetch<kNTW>(bn, P + d.b_off[L1], nt0, L);
  epilogue<true, TRAIN, kNTW>(acc, R0, nt0, kNT, arow(0), wg, L, nullptr, arow8(0));
  wg_barrier();

  // ---- L1..L4: 256 -> 256, snake, ping-pong R0 -> R1 -> R0 -> R1 -> R0
#pragma unroll
  for (int l = L1; l <= L4; ++l) {
    char* in = (l & 1) ? R0 : R1;
    char* out = (l & 1) ? R1 : R0;
    bias_apply<kNTW>(acc, bn);
    MMA_RING<0, A, A, A, kNTW, kNT>(acc, in, 0, wl(l), (l == L4 && !EMB_IN && kOverlapPro) ? wl(L5) + kKSEmb * U : wl(l + 1), nt0, L, ring);
    bias_fetch<kNTW>(bn, P + d.b_off[l + 1], nt0, L);
    epilogue<true, TRAIN, kNTW>(acc, out, nt0, kNT, arow(l), wg, L, nullptr, arow8(l));
    wg_barrier();
  }

  // ---- L5: [emb(p0) (LDS ring R1), h (R0)] -> 256, snake, out -> R1 (the LDS ring is idle
  //      again after mma_embedding's final barrier)
  bias_apply<kNTW>(acc, bn);
  if (!EMB_IN && kOverlapPro) {
    // h part FIRST: sV still holds proposal 0's warped coordinates (written for L0, nothing else touches it), so chunk 0 of
    // the embedding part is generated in the gaps of these 16 k-steps instead of in an exposed prologue
    mma_plain_gen_chunk0<0, 0, A, kNTW, kNT>(acc, R0, e, 0, R1, sV, wl(L5) + kKSEmb * U, wl(L5), nt0, s_actF, wg, L, ring);
    wg_barrier();
    mma_embedding<0, kNTW, kNT, EMB_IN>(acc, e, 0, R1, sV, sY, sX, wl(L5), wl(L6), nt0, s_actF, wg, L, ring, true, true);
  } else {
    mma_embedding<0, kNTW, kNT, EMB_IN>(acc, e, 0, R1, sV, sY, sX, wl(L5), wl(L5) + kKSEmb * U, nt0, s_actF, wg, L, ring);
    MMA_RING<0, A, A, A, kNTW, kNT>(acc, R0, 0, wl(L5) + kKSEmb * U, wl(L6), nt0, L, ring);
  }
  bias_fetch<kNTW>(bn, P + d.b_off[L6], nt0, L);
  epilogue<true, TRAIN, kNTW>(acc, R1, nt0, kNT, arow(5), wg, L, nullptr, arow8(5));
  wg_barrier();

  // ---- L6: R1 -> R0, L7: R0 -> R1
  bias_apply<kNTW>(acc, bn);
  MMA_RING<0, A, A, A, kNTW, kNT>(acc, R1, 0, wl(L6), wl(L7), nt0, L, ring);
  bias_fetch<kNTW>(bn, P + d.b_off[L7], nt0, L);
  epilogue<true, TRAIN, kNTW>(acc, R0, nt0, kNT, arow(6), wg, L, nullptr, arow8(6));
  wg_barrier();
  bias_apply<kNTW>(acc, bn);
  MMA_RING<0, A, A, A, kNTW, kNT>(acc, R0, 0, wl(L7), wl(LF1), nt0, L, ring);
  bias_fetch<kNTW>(bn, P + d.b_off[LF1], nt0, L);
  epilogue<true, TRAIN, kNTW>(acc, R1, nt0, kNT, arow(7), wg, L, nullptr, arow8(7));
  wg_barrier();

  // ---- F1 = feature_linear1 (linear): R1 -> R0; its fragments are also kept in
  //      registers because P needs f1 again after S and F2 have recycled the regions.
  bias_apply<kNTW>(acc, bn);
  MMA_RING<0, A, A, A, kNTW, kNT>(acc, R1, 0, wl(LF1), MULTI ? wl(LS) : kNoW, nt0, L, ring);
  const bool p_wave = L.wave < kNT / 2;            // P has 4 neuron tiles: with 8 waves the upper four only keep the barriers
  if (!MULTI && p_wave) WRING_FILL(1, kNT / 2, ringp, wl(LP), L.wave, L);
  if (MULTI) bias_fetch<kNTW>(bn, P + d.b_off[LS], nt0, L);
  else if (p_wave) bias_fetch<1>(bnp, P + d.b_off[LP], L.wave, L);
  epilogue<false, TRAIN, kNTW>(acc, R0, nt0, kNT, arow(kActF1), wg, L, nullptr, arow8(kActF1));
  wg_barrier();

  f32x16 accp[1][kNB];
  if (MULTI) {
    // ---- S = scale_linears[0]: [f1 (R0), emb(p1..pK-1) (LDS ring R1)] -> 256, snake, out -> R1
    bias_apply<kNTW>(acc, bn);
    if (!EMB_IN && kOverlapPro) {
      // f1 part: proposal 1's warped coordinates are computed in the gaps of its first half (sV is idle since L5), a
      // barrier publishes them, chunk 0 of proposal 1 is generated in the gaps of the second half
      auto warp_hook = [&](int pos) { if (pos < kWarpPer) gen_warp(e.warp, 1, sV, sY, sX, L, pos); };
      MMA_RING<0, A / 2, A, A, kNTW, kNT>(acc, R0, 0, wl(LS), wl(LS) + A * U, nt0, L, ring, warp_hook);
      wg_barrier();
      mma_plain_gen_chunk0<TRAIN, A / 2, A, kNTW, kNT>(acc, R0, e, 1, R1, sV, wl(LS), wl(LS) + A * U, nt0, s_actE, wg, L, ring);
      wg_barrier();
    } else {
      MMA_RING<0, A, A, A, kNTW, kNT>(acc, R0, 0, wl(LS), wl(LS) + A * U, nt0, L, ring);
    }
    for (int p = 1; p < d.K; ++p) {
      const wptr_t wpp = wl(LS) + (wptr_t)(A + (p - 1) * kKSEmb) * U;
      mma_embedding<TRAIN, kNTW, kNT, EMB_IN>(acc, e, p, R1, sV, sY, sX, wpp, (p + 1 < d.K) ? wpp + kKSEmb * U : kNoW, nt0,
                                   s_actE, wg, L, ring, /*have_warp=*/!EMB_IN && kOverlapPro, /*have_chunk0=*/!EMB_IN && kOverlapPro && p == 1,
                                   /*next_warp_p=*/(kOverlapPro && p + 1 < d.K) ? p + 1 : -1);
    }
    WRING_FILL(kNTW, kNT, ring, wl(LF2), nt0, L);        // flies under the epilogue
    bias_fetch<kNTW>(bn, P + d.b_off[LF2], nt0, L);
    epilogue<true, TRAIN, kNTW>(acc, R1, nt0, kNT, arow(kActAS), wg, L, nullptr, arow8(kActAS));
    wg_barrier();
    // ---- F2 = feature_linear2 (linear): R1 -> R0
    bias_apply<kNTW>(acc, bn);
    MMA_RING<0, A, A, A, kNTW, kNT>(acc, R1, 0, wl(LF2), kNoW, nt0, L, ring);
    if (p_wave) {
      WRING_FILL(1, kNT / 2, ringp, wl(LP), L.wave, L);
      bias_fetch<1>(bnp, P + d.b_off[LP], L.wave, L);
    }
    // P needs f1 again, and this epilogue recycles its region: the wave takes ITS f1 fragments (the only ones its own f2 stores
    // overwrite) out of R0 first and hands them back through R1 after the barrier -- live across one epilogue, not across S and F2
    bf16x8 f1keep[kNTW][kNB][2];
#pragma unroll
    for (int nt = 0; nt < kNTW; ++nt)
#pragma unroll
      for (int bt = 0; bt < kNB; ++bt)
#pragma unroll
        for (int s = 0; s < 2; ++s) f1keep[nt][bt][s] = lds_frag(R0, 2 * (nt0 + nt) + s, bt, L.lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the reads have landed before the stores below are issued
    epilogue<false, TRAIN, kNTW>(acc, R0, nt0, kNT, arow(kActF2), wg, L, nullptr, arow8(kActF2));
    wg_barrier();
    // f1 back into LDS (R1 is idle: every wave passed the barrier after reading a_s)
#pragma unroll
    for (int nt = 0; nt < kNTW; ++nt)
#pragma unroll
      for (int bt = 0; bt < kNB; ++bt)
#pragma unroll
        for (int s = 0; s < 2; ++s) lds_store_frag(R1, 2 * (nt0 + nt) + s, bt, L.lane, f1keep[nt][bt][s]);
    wg_barrier();
    // ---- P = pos_linears[0]: [f1 (R1), f2 (R0)] -> 128, snake; one neuron tile per wave
    if (p_wave) {
      bias_apply<1>(accp, bnp);
      MMA_RING<0, A, A, A, 1, kNT / 2>(accp, R1, 0, wl(LP), wl(LP) + A * UP, L.wave, L, ringp);
      MMA_RING<0, A, A, A, 1, kNT / 2>(accp, R0, 0, wl(LP) + A * UP, kNoW, L.wave, L, ringp);
    }
  } else if (p_wave) {
    // ---- NPP_Net_top1: P reads f1 (R0) directly (networks.py:162-170)
    bias_apply<1>(accp, bnp);
    MMA_RING<0, A, A, A, 1, kNT / 2>(accp, R0, 0, wl(LP), kNoW, L.wave, L, ringp);
  }
  if (p_wave)
    epilogue<true, TRAIN, 1>(accp, nullptr, L.wave, kNT / 2,
                             TRAIN == 2 ? s_actF + wfmt8_array_base(kActKsAP, L.n_wg) : TRAIN ? s_actF + wfmt_array_base(kActKsAP, L.n_wg) : nullptr,
                             wg, L, nullptr, TRAIN == 2 ? s_actF + act8_region_base(d.K, L.n_wg) + wfmt8_array_base(kActKsAP, L.n_wg) : nullptr);

  // ---- rgb_linear 128 -> 3 + sigmoid: per-lane partial dot over its 16 features,
  //      half-wave exchange by shuffle, 4-wave reduction through LDS.
  wg_barrier();       // every wave is done with R0 / R1: R0 now carries the rgb partial sums
  if (p_wave) {
    const float* Wr = P + d.w_off[LRGB];
    // partial dot of this lane's 16 P-neurons per output row it holds: 32x32x16 form: one row per batch tile (reduce over the
    // two lane halves); 16x16x32 form: rows 16 ri + (lane & 15) of each batch tile (reduce over the four 16-lane groups)
    constexpr int RI = kM16 ? 2 : 1;
    float part[kNB][RI][3];
#pragma unroll
    for (int bt = 0; bt < kNB; ++bt)
#pragma unroll
      for (int ri = 0; ri < RI; ++ri)
#pragma unroll
        for (int c = 0; c < 3; ++c) part[bt][ri][c] = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = L.wave * 32 + acc_nrow(r, L);
      const int ri = kM16 ? (r >> 2) & 1 : 0;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float w = Wr[c * (kW / 2) + k];
#pragma unroll
        for (int bt = 0; bt < kNB; ++bt) part[bt][ri][c] = fmaf(w, accp[0][bt][r], part[bt][ri][c]);
      }
    }
#pragma unroll
    for (int bt = 0; bt < kNB; ++bt)
#pragma unroll
      for (int ri = 0; ri < RI; ++ri)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float v = part[bt][ri][c] + __shfl_xor(part[bt][ri][c], 32, 64);
          if (kM16) {
            v += __shfl_xor(v, 16, 64);
            if ((L.lane >> 4) == 0) sRGB[(L.wave * kRowTile + bt * 32 + ri * 16 + (L.lane & 15)) * 3 + c] = v;
          } else if (L.h == 0) {
            sRGB[(L.wave * kRowTile + bt * 32 + L.b) * 3 + c] = v;
          }
        }
  }
  {
    wg_barrier();
    if (L.tid < kRowTile * 3) {
      const int row = L.tid / 3, c = L.tid - row * 3;
      float z = P[d.b_off[LRGB] + c];
#pragma unroll
      for (int w = 0; w < kNT / 2; ++w) z += sRGB[(w * kRowTile + row) * 3 + c];   // P's neuron tiles, in order
      float o = 1.0f / (1.0f + __expf(-z));                         // helpers.py:56 sigmoid
      if ((EMB_IN || ACT) && A_.out_act != 1) o = A_.out_act == 2 ? tanhf(z) : z;   // helpers.py:57-58 tanh (--normalize_type 2) / raw network output
      s_pred[(row0 + row) * 3 + c] = o;
    }
  }
#pragma unroll
  for (int q = 0; q < NPP_FWD_PREFETCH_LINES; ++q) asm volatile("" :: "v"(pfv[q]));
}

}  // namespace npp

using namespace npp;


#undef s_coords
#undef s_wf
#undef s_params
#undef s_pred
#undef s_actF
#undef s_actE

static int fwd_launch(const FwdArgs& A, const EmbedDev& e, const NetDesc& d, bool emb_in, void* stream, const char* who) {
  const dim3 grid(A.S.M ? stack_grid(A.S) : (unsigned)(A.Bp / kRowTile)), block(kThreads);
  const bool train = A.actF != nullptr, multi = d.K > 1;
  const bool s8 = __atomic_load_n(&g_tune.stash8, __ATOMIC_RELAXED) != 0;     // npp_tune "stash8": the 8-bit training stash
  hipStream_t s = (hipStream_t)stream;
#define NPP_LAUNCH(T, M, E)                                                                       \
  do {                                                                                            \
    static SmemOnce once;                                                                         \
    if (!smem_attr(once, (const void*)mlp_fwd_kernel<T, M, E>, kSmemFwd)) {                       \
      set_error("%s: smem attribute", who); return NPP_ERR_LAUNCH;                                \
    }                                                                                             \
    hipLaunchKernelGGL((mlp_fwd_kernel<T, M, E>), grid, block, kSmemFwd, s, A, e, d);             \
  } while (0)
#define NPP_LAUNCH2(E)                                                                            \
  do {                                                                                            \
    if (train && s8) { if (multi) NPP_LAUNCH(2, true, E); else NPP_LAUNCH(2, false, E); }         \
    else if (train) { if (multi) NPP_LAUNCH(1, true, E); else NPP_LAUNCH(1, false, E); }          \
    else { if (multi) NPP_LAUNCH(0, true, E); else NPP_LAUNCH(0, false, E); }                     \
  } while (0)
  if (A.S.M) {                            // stacked launch: training form, coordinates in
#define NPP_LAUNCH_STACK(T, M)                                                                                 \
    do {                                                                                                         \
      static SmemOnce once;                                                                                      \
      if (!smem_attr(once, (const void*)mlp_fwd_kernel<T, M, false, true>, kSmemFwd)) {                          \
        set_error("%s: smem attribute", who); return NPP_ERR_LAUNCH;                                             \
      }                                                                                                          \
      hipLaunchKernelGGL((mlp_fwd_kernel<T, M, false, true>), grid, block, kSmemFwd, s, A, e, d);                \
    } while (0)
    if (s8) { if (multi) NPP_LAUNCH_STACK(2, true); else NPP_LAUNCH_STACK(2, false); }
    else { if (multi) NPP_LAUNCH_STACK(1, true); else NPP_LAUNCH_STACK(1, false); }
#undef NPP_LAUNCH_STACK
  } else if (emb_in) NPP_LAUNCH2(true);
  else if (A.out_act != 1) {               // npp_mlp_fwd_act: the instantiations that read the nonlinearity from the arguments
#define NPP_LAUNCH_ACT(T, M)                                                                                   \
    do {                                                                                                         \
      static SmemOnce once;                                                                                      \
      if (!smem_attr(once, (const void*)mlp_fwd_kernel<T, M, false, false, true>, kSmemFwd)) {                   \
        set_error("%s: smem attribute", who); return NPP_ERR_LAUNCH;                                             \
      }                                                                                                          \
      hipLaunchKernelGGL((mlp_fwd_kernel<T, M, false, false, true>), grid, block, kSmemFwd, s, A, e, d);         \
    } while (0)
    if (train && s8) { if (multi) NPP_LAUNCH_ACT(2, true); else NPP_LAUNCH_ACT(2, false); }
    else if (train) { if (multi) NPP_LAUNCH_ACT(1, true); else NPP_LAUNCH_ACT(1, false); }
    else { if (multi) NPP_LAUNCH_ACT(0, true); else NPP_LAUNCH_ACT(0, false); }
#undef NPP_LAUNCH_ACT
  } else NPP_LAUNCH2(false);
#undef NPP_LAUNCH2
#undef NPP_LAUNCH
  return check_launch(who);
}

static int fwd_check(int64_t Bp, int width, const void* in, const void* d_wf, const float* d_params, const float* d_pred,
                     const char* who) {
  if (width != NPP_WIDTH) { set_error("%s: width %d unsupported (build is %d)", who, width, NPP_WIDTH); return NPP_ERR_UNSUPPORTED; }
  if (Bp <= 0 || Bp % kRowTile) { set_error("%s: Bp=%lld must be a positive multiple of %d", who, (long long)Bp, kRowTile); return NPP_ERR_ARG; }
  if (!in || !d_wf || !d_params || !d_pred) { set_error("%s: null pointer", who); return NPP_ERR_ARG; }
  if (Bp / kRowTile > 0x7fffffffLL) { set_error("%s: Bp too large", who); return NPP_ERR_ARG; }
  return NPP_OK;
}

extern "C" int npp_mlp_fwd(const int32_t* d_coords_yx, int64_t Bp, const npp_embed_cfg* cfg, int width,
                           const void* d_wf, const float* d_params, float* d_pred, void* d_actT, void* stream) {
  int rc = check_embed_cfg(cfg, "npp_mlp_fwd");
  if (rc) return rc;
  if ((rc = fwd_check(Bp, width, d_coords_yx, d_wf, d_params, d_pred, "npp_mlp_fwd"))) return rc;
  FwdArgs A{};
  A.coords = d_coords_yx; A.Bp = Bp; A.wf = (const bf16x8*)d_wf; A.params = d_params; A.pred = d_pred;
  A.actF = (char*)d_actT; A.out_act = 1;
  return fwd_launch(A, make_embed_dev(*cfg), make_desc(cfg->K), false, stream, "npp_mlp_fwd");
}

// npp_mlp_fwd with the output nonlinearity of render() as an argument (models/helpers.py:55-60): 1 sigmoid (--normalize_type 1, what
// npp_mlp_fwd applies), 2 tanh (--normalize_type 2: images in [-1, 1], loaders.py:56), 0 the raw network output
extern "C" int npp_mlp_fwd_act(const int32_t* d_coords_yx, int64_t Bp, const npp_embed_cfg* cfg, int width, const void* d_wf,
                               const float* d_params, float* d_pred, void* d_actT, int out_act, void* stream) {
  int rc = check_embed_cfg(cfg, "npp_mlp_fwd_act");
  if (rc) return rc;
  if ((rc = fwd_check(Bp, width, d_coords_yx, d_wf, d_params, d_pred, "npp_mlp_fwd_act"))) return rc;
  if (out_act < 0 || out_act > 2) { set_error("npp_mlp_fwd_act: out_act=%d", out_act); return NPP_ERR_ARG; }
  FwdArgs A{};
  A.coords = d_coords_yx; A.Bp = Bp; A.wf = (const bf16x8*)d_wf; A.params = d_params; A.pred = d_pred;
  A.actF = (char*)d_actT; A.out_act = out_act;
  return fwd_launch(A, make_embed_dev(*cfg), make_desc(cfg->K), false, stream, "npp_mlp_fwd_act");
}

extern "C" int npp_mlp_fwd_emb(const float* d_emb, int64_t ld, int64_t Bp, int K, int width, const void* d_wf,
                               const float* d_params, float* d_out, void* d_actT, int out_act, void* stream) {
  int rc = fwd_check(Bp, width, d_emb, d_wf, d_params, d_out, "npp_mlp_fwd_emb");
  if (rc) return rc;
  if (K < 1 || K > NPP_MAX_K || ld < (int64_t)K * kE || out_act < 0 || out_act > 2) {
    set_error("npp_mlp_fwd_emb: bad K=%d / ld=%lld / out_act=%d", K, (long long)ld, out_act);
    return NPP_ERR_ARG;
  }
  FwdArgs A{};
  A.Bp = Bp; A.wf = (const bf16x8*)d_wf; A.params = d_params; A.pred = d_out; A.actF = (char*)d_actT;
  A.emb = d_emb; A.emb_ld = ld; A.out_act = out_act;
  EmbedDev e{};
  e.K = K;
  return fwd_launch(A, e, make_desc(K), true, stream, "npp_mlp_fwd_emb");
}

// ---- stacked form: M images per launch (npp_common.h "stacked launches") ----------------------------------------------------
extern "C" int npp_embed_dev_bytes(void) { return (int)sizeof(EmbedDev); }

extern "C" int npp_embed_dev_build(const npp_embed_cfg* cfg, void* host_out) {
  int rc = check_embed_cfg(cfg, "npp_embed_dev_build");
  if (rc) return rc;
  if (!host_out) { set_error("npp_embed_dev_build: null pointer"); return NPP_ERR_ARG; }
  const EmbedDev e = make_embed_dev(*cfg);
  memcpy(host_out, &e, sizeof(e));
  return NPP_OK;
}

extern "C" int npp_mlp_fwd_stack(const int32_t* d_coords_yx, int64_t Bp, const void* d_embed_dev, int M, int K, int width,
                                 const void* d_wf, int64_t wf_stride_bytes, const float* d_params, int64_t params_stride,
                                 float* d_pred, void* d_actT, int64_t act_stride_bytes, const void* d_iter, void* stream) {
  int rc = fwd_check(Bp, width, d_coords_yx, d_wf, d_params, d_pred, "npp_mlp_fwd_stack");
  if (rc) return rc;
  if (M < 1 || M > NPP_MAX_STACK || K < 1 || K > NPP_MAX_K || !d_embed_dev || wf_stride_bytes % 16 || act_stride_bytes % 16 ||
      wf_stride_bytes < 16 * make_desc(K).wf_total16 || params_stride < make_desc(K).total_params) {
    set_error("npp_mlp_fwd_stack: bad M=%d / K=%d / strides", M, K);
    return NPP_ERR_ARG;
  }
  FwdArgs A{};
  A.coords = d_coords_yx; A.Bp = Bp; A.wf = (const bf16x8*)d_wf; A.params = d_params; A.pred = d_pred;
  A.actF = (char*)d_actT; A.out_act = 1;
  A.S = make_stack(M, (int)(Bp / kRowTile), d_iter);
  A.estack = (const EmbedDev*)d_embed_dev;
  A.wf_stride16 = wf_stride_bytes / 16; A.params_stride = params_stride; A.act_stride = act_stride_bytes;
  EmbedDev e{};
  e.K = K;
  return fwd_launch(A, e, make_desc(K), false, stream, "npp_mlp_fwd_stack");
}
