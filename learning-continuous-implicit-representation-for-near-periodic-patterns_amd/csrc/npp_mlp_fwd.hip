// npp_mlp_fwd.hip -- K2: fused embedder + coordinate MLP forward (+ sigmoid), bf16 MFMA.
//
// Replaces, in one launch, what the reference does with a table gather, 13 F.linear
// GEMMs, 10 snake kernels, 3 cats and a sigmoid (NPP_completion/train.py:166-181 ->
// models/helpers.py:41-62 -> models/networks.py:56-95 / :145-173).
//
// Structure: one 256-thread workgroup owns 64 pixel rows (two
// 32-column batch tiles).  The GEMMs are computed TRANSPOSED, Z^T[n][b] = W[n][k] X^T[k][b],
// with v_mfma_f32_32x32x16_bf16: the weights are the A operand (pre-packed in fragment
// order, streamed straight from L2 with one coalesced 1-KiB load per fragment), the
// activations are the B operand.  A 32x32 accumulator tile converted to bf16 IS the B
// operand of the next layer's k-steps (no transpose), so activations move between layers
// as 16-byte-per-lane fragments through LDS: each of the 4 waves owns 2 of the 8 neuron
// tiles and needs the other waves' tiles as its next input.  The 462-wide embedding
// inputs are never materialised: per proposal the 22 warped coordinates of the 64 rows
// go to LDS (fp32) and sin/cos Fourier fragments are generated chunk-wise (8 k-steps)
// into a double-buffered LDS ring by all four waves, overlapped with the MFMAs of the
// previous chunk.  rgb_linear (128->3) is a VALU dot + wave shuffle + LDS reduction.
//
// In training mode the kernel also writes (a) the snake derivative 1+sin(2z) in bf16
// fragment order (read back 1:1 by npp_mlp_bwd) and (b) every layer input -- including
// the embedding slots -- as the same 16-byte fragments it exchanges through LDS, in the
// "W-format" line layout of npp_layout.h that npp_mlp_wgrad copies linearly into LDS and
// reads transposed (two coalesced 16-byte stores per accumulator tile).
//
// Algorithmic work: 2 * ((K+1)*462*256 + 11*256^2 + 384) FLOP per row;
// the zero padding of 462 -> 480 slots per proposal is not counted.
#include "npp_common.h"

namespace npp {

EmbedDev make_embed_dev(const npp_embed_cfg& c);
int check_embed_cfg(const npp_embed_cfg* c, const char* who);

constexpr int kThreads = 256;
constexpr int kFragBytes = 1024;                               // 64 lanes x 16 B
constexpr int kRegionBytes = kKSAct * kNB * kFragBytes;        // 32 KiB: 256 feats x 64 rows bf16
constexpr int kChunkKS = 8;                                    // k-steps per embedding chunk
constexpr int kChunkBytes = kChunkKS * kNB * kFragBytes;       // 16 KiB, two of them = one region
constexpr int kNChunks = (kKSEmb + kChunkKS - 1) / kChunkKS;   // 4 (8,8,8,6)
constexpr int kSmemV = 22 * kRowTile * 4;                      // warped coords of one proposal
constexpr int kSmemE = (sizeof(EmbedDev) + 15) / 16 * 16;
constexpr int kSmemFwd = 2 * kRegionBytes + kSmemV + 2 * kRowTile * 4 + 4 * kRowTile * 3 * 4 + kSmemE;

struct FwdArgs {
  const int32_t* coords;
  int64_t Bp;
  const bf16x8* wf;
  const float* params;
  float* pred;
  bf16x8* sstash;    // nullable
  char* actF;        // nullable: W-format fragment arrays (npp_layout.h)
};

struct Lane {
  int tid, wave, lane, b, h;
};

__device__ __forceinline__ bf16x8 lds_frag(const char* region, int ks, int bt, int lane) {
  return *(const bf16x8*)(region + ((ks * kNB + bt) * 64 + lane) * 16);
}
__device__ __forceinline__ void lds_store_frag(char* region, int ks, int bt, int lane, const bf16x8& v) {
  *(bf16x8*)(region + ((ks * kNB + bt) * 64 + lane) * 16) = v;
}

// acc[nt][bt] <- bias of this wave's neuron tiles (row constants as the initial accumulator)
template <int NTW>
__device__ __forceinline__ void init_bias(f32x16 (&acc)[NTW][kNB], const float* __restrict__ bias, int nt0,
                                          const Lane& L) {
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
    f32x16 bv;
#pragma unroll
    for (int r = 0; r < 16; ++r) bv[r] = bias[(nt0 + nt) * 32 + acc_row(r, L.h)];
#pragma unroll
    for (int bt = 0; bt < kNB; ++bt) acc[nt][bt] = bv;
  }
}

// ---- weight stream: a rolling register ring of 4 k-steps --------------------------------
// With only Bp/64 workgroups in flight the kernel is latency-bound unless the L2 -> register
// weight stream runs ahead of the MFMAs.  Each wave keeps the weight fragments of the next
// 4 k-steps in a ring; right after the MFMAs of k-step ks have been issued, their slot is
// refilled with k-step ks+4 (16 MFMAs = 512+ cycles ahead).  The ring runs across part and
// layer boundaries (next_wp), so loads also fly under the epilogue and the barrier; START
// is the (compile-time) slot of this part's first k-step.
template <int NTW>
struct WRing {
  bf16x8 w[4][NTW];
};

template <int NTW, int NT>
__device__ __forceinline__ void wslot_load(WRing<NTW>& r, int slot, const bf16x8* __restrict__ wp, int ks, int nt0,
                                           int lane) {
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) r.w[slot][nt] = wp[((int64_t)ks * NT + nt0 + nt) * 64 + lane];
}
// fresh fill of the ring with k-steps 0..3 of wp (START = 0 for the consumer)
template <int NTW, int NT>
__device__ __forceinline__ void wring_fill(WRing<NTW>& r, const bf16x8* __restrict__ wp, int nt0, int lane) {
#pragma unroll
  for (int q = 0; q < 4; ++q) wslot_load<NTW, NT>(r, q, wp, q, nt0, lane);
}

// Ring schedule positions [KS0, KS1) of a part with KSREAL real k-steps (weights at wp) padded
// to KSTOT (a multiple of 4) schedule positions, so that every part starts at ring slot 0:
// positions >= KSREAL issue no MFMA and load nothing, they only keep the refill cadence.
// Activation fragments of k-step ks sit at LDS k-step (ks - KS0 + ks_lds0) of `region`.
template <int KS0, int KS1, int KSREAL, int KSTOT, int NTW, int NT>
__device__ __forceinline__ void mma_ring(f32x16 (&acc)[NTW][kNB], const char* region, int ks_lds0,
                                         const bf16x8* __restrict__ wp, const bf16x8* __restrict__ next_wp, int nt0,
                                         const Lane& L, WRing<NTW>& ring) {
  static_assert(KSTOT % 4 == 0 && KSREAL <= KSTOT && KS1 <= KSTOT, "ring schedule");
#pragma unroll
  for (int ks = KS0; ks < KS1; ++ks) {
    const int slot = ks & 3;
    if (ks < KSREAL) {
      bf16x8 x[kNB];
#pragma unroll
      for (int bt = 0; bt < kNB; ++bt) x[bt] = lds_frag(region, ks_lds0 + ks - KS0, bt, L.lane);
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int bt = 0; bt < kNB; ++bt) acc[nt][bt] = mfma_bf16(ring.w[slot][nt], x[bt], acc[nt][bt]);
    }
    if (ks + 4 < KSREAL) wslot_load<NTW, NT>(ring, slot, wp, ks + 4, nt0, L.lane);
    else if (ks + 4 >= KSTOT && next_wp) wslot_load<NTW, NT>(ring, slot, next_wp, ks + 4 - KSTOT, nt0, L.lane);
    asm volatile("" ::: "memory");   // pin the refill here: no hoisting of later loads
  }
}

// The 22 warped coordinates (a1) of proposal p for the 64 rows -> sV[i][row] (fp32).
__device__ __forceinline__ void gen_warp(const EmbedDev& e, int p, float* sV, const float* sY, const float* sX,
                                         const Lane& L) {
  const float y = sY[L.lane], x = sX[L.lane];
  for (int i = L.wave; i < 22; i += 4) sV[i * kRowTile + L.lane] = warp_value<false>(e, p, i, y, x);
}

// One embedding fragment: k-step ks (0..29) of a proposal, batch tile bt.  Slot order:
// npp_layout.h emb_col().  cos(x) = sin(x + 1/4 rev): one transcendental per element.
__device__ __forceinline__ bf16x8 gen_emb_frag(const EmbedDev& e, const float* sV, int ks, int bt, const Lane& L) {
  bf16x8 f;
  const int row = bt * 32 + L.b;
  if (ks < 28) {
    const float ph = L.h ? 0.25f : 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int t = 8 * ks + j;
      float val = 0.0f;
      if (t < 220) {
        const int fj = t / 22, i = t - fj * 22;
        val = __builtin_amdgcn_sinf(fmaf(sV[i * kRowTile + row], e.freq_rev[fj], ph));
      }
      f[j] = (__bf16)val;
    }
  } else if (ks == 28) {
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (__bf16)sV[(8 * L.h + j) * kRowTile + row];
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (__bf16)((L.h == 0 && j < 6) ? sV[(16 + j) * kRowTile + row] : 0.0f);
  }
  return f;
}

// Accumulate one proposal's 30 embedding k-steps.  lds_ring = 32 KiB LDS (two 16 KiB chunk
// buffers).  Caller guarantees sV is free to overwrite and lds_ring is free; on return every
// wave has passed a barrier after its last lds_ring / sV read.  The weight ring holds the
// first 4 k-steps on entry and the first 4 k-steps of next_wp on return.
template <bool STORE_EMB, int NTW, int NT>
__device__ __forceinline__ void mma_embedding(f32x16 (&acc)[NTW][kNB], const EmbedDev& e, int p, char* lds_ring,
                                              float* sV, const float* sY, const float* sX,
                                              const bf16x8* __restrict__ wp, const bf16x8* __restrict__ next_wp,
                                              int nt0, char* actF, int wg, const Lane& L, WRing<NTW>& ring) {
  gen_warp(e, p, sV, sY, sX, L);
  wg_barrier();
  auto gen_chunk = [&](int c) {
    char* buf = lds_ring + (c & 1) * kChunkBytes;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int ksl = 2 * L.wave + q;        // wave-uniform
      const int ks = kChunkKS * c + ksl;
      if (ks < kKSEmb) {
#pragma unroll
        for (int bt = 0; bt < kNB; ++bt) {
          const bf16x8 f = gen_emb_frag(e, sV, ks, bt, L);
          lds_store_frag(buf, ksl, bt, L.lane, f);
          if (STORE_EMB)
            *(bf16x8*)(actF + wfmt_array_base(kActKsEmb0 + p * kKSEmb, gridDim.x) +
                       wfmt_unit(kKSEmb, wg, ks, bt, L.b, L.h)) = f;
        }
      }
    }
  };
  gen_chunk(0);
  wg_barrier();
  gen_chunk(1);
  mma_ring<0, 8, kKSEmb, 32, NTW, NT>(acc, lds_ring, 0, wp, next_wp, nt0, L, ring);
  wg_barrier();
  gen_chunk(2);
  mma_ring<8, 16, kKSEmb, 32, NTW, NT>(acc, lds_ring + kChunkBytes, 0, wp, next_wp, nt0, L, ring);
  wg_barrier();
  gen_chunk(3);
  mma_ring<16, 24, kKSEmb, 32, NTW, NT>(acc, lds_ring, 0, wp, next_wp, nt0, L, ring);
  wg_barrier();
  mma_ring<24, 32, kKSEmb, 32, NTW, NT>(acc, lds_ring + kChunkBytes, 0, wp, next_wp, nt0, L, ring);
  wg_barrier();
}

// Epilogue of a 256-wide (NTW=2 per wave) or 128-wide (NTW=1) layer.
//  SNAKE: apply x + sin^2 x; else linear.   out: LDS region that receives the bf16
//  fragments (k-step 2*ntile+s of the next layer), may be null.
//  TRAIN: stash the derivative (fragment order) and the activation (feature-major).
template <bool SNAKE, bool TRAIN, int NTW>
__device__ __forceinline__ void epilogue(f32x16 (&acc)[NTW][kNB], char* out, int nt0, int ntl /*tiles in layer*/,
                                         bf16x8* sstash_layer, char* actF_array, int wg, const Lane& L,
                                         bf16x8 (*keep)[kNB][2] = nullptr) {
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
    const int ntg = nt0 + nt;
#pragma unroll
    for (int bt = 0; bt < kNB; ++bt) {
      f32x16 a, ds;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float z = acc[nt][bt][r];
        if (SNAKE) {
          if (TRAIN) { float av, dv; snake_fast2(z, av, dv); a[r] = av; ds[r] = dv; }
          else a[r] = snake_fast(z);
        } else {
          a[r] = z;
        }
      }
      acc[nt][bt] = a;   // callers that need the fp32 activation (P -> rgb) read it back
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 f = pack_acc(a, s);
        if (out) lds_store_frag(out, 2 * ntg + s, bt, L.lane, f);
        if (keep) keep[nt][bt][s] = f;
        if (TRAIN && SNAKE) sstash_layer[((((int64_t)wg * ntl + ntg) * kNB + bt) * 2 + s) * 64 + L.lane] = pack_acc(ds, s);
        if (TRAIN) *(bf16x8*)(actF_array + wfmt_unit(2 * ntl, wg, 2 * ntg + s, bt, L.b, L.h)) = f;
      }
    }
  }
}

template <bool TRAIN, bool MULTI>
__global__ __launch_bounds__(kThreads, 2) void mlp_fwd_kernel(FwdArgs A_, EmbedDev e_arg, NetDesc d) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* R0 = smem;
  char* R1 = smem + kRegionBytes;
  float* sV = (float*)(smem + 2 * kRegionBytes);
  float* sY = sV + 22 * kRowTile;
  float* sX = sY + kRowTile;
  float* sRGB = sX + kRowTile;          // [4 waves][64 rows][3]
  // The embedder constants are indexed with run-time (proposal, orientation, offset)
  // indices: keep them in LDS, copied with compile-time indices so the by-value kernel
  // argument never needs a scratch copy.
  EmbedDev& e = *(EmbedDev*)(sRGB + 4 * kRowTile * 3);
  if (threadIdx.x == 0) {
    const uint32_t* src = (const uint32_t*)&e_arg;
    uint32_t* dst = (uint32_t*)&e;
#pragma unroll
    for (int i = 0; i < (int)(sizeof(EmbedDev) / 4); ++i) dst[i] = src[i];
  }

  Lane L;
  L.tid = threadIdx.x;
  L.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  L.lane = threadIdx.x & 63;
  L.b = L.lane & 31;
  L.h = L.lane >> 5;
  const int wg = blockIdx.x;
  const int64_t row0 = (int64_t)wg * kRowTile;
  const int64_t Bp = A_.Bp;
  const float* P = A_.params;
  const bf16x8* wf = A_.wf;
  const int nt0 = 2 * L.wave;            // this wave's neuron tiles in 256-wide layers

  if (L.tid < kRowTile) {
    const int2 c = ((const int2*)A_.coords)[row0 + L.tid];
    sY[L.tid] = (float)c.x;              // (row=y, col=x)
    sX[L.tid] = (float)c.y;
  }
  wg_barrier();

  auto ss = [&](int slot) -> bf16x8* {
    return TRAIN ? (bf16x8*)((char*)A_.sstash + sstash_off_bytes(slot, Bp)) : nullptr;
  };
  auto arow = [&](int idx) -> char* { return TRAIN ? A_.actF + wfmt_array_base(idx * kKSAct, gridDim.x) : nullptr; };

  f32x16 acc[2][kNB];
  WRing<2> ring;                           // weight-stream register ring, live across layers
  WRing<1> ringp;                          // same for P (one neuron tile per wave)
  constexpr int64_t U = (int64_t)kNT * 64; // 16-byte units per k-step of a 256-wide layer
  constexpr int64_t UP = (int64_t)(kNT / 2) * 64;
  constexpr int A = kKSAct;                // 16 k-steps per 256 features
  auto wl = [&](int l) -> const bf16x8* { return wf + d.wf_off[l]; };

  // ---- L0: emb(p0) -> 256, snake.  LDS ring = R1, out -> R0
  wring_fill<2, kNT>(ring, wl(L0), nt0, L.lane);
  init_bias<2>(acc, P + d.b_off[L0], nt0, L);
  mma_embedding<TRAIN, 2, kNT>(acc, e, 0, R1, sV, sY, sX, wl(L0), wl(L1), nt0, A_.actF, wg, L, ring);
  epilogue<true, TRAIN, 2>(acc, R0, nt0, kNT, ss(0), arow(0), wg, L);
  wg_barrier();

  // ---- L1..L4: 256 -> 256, snake, ping-pong R0 -> R1 -> R0 -> R1 -> R0
#pragma unroll
  for (int l = L1; l <= L4; ++l) {
    char* in = (l & 1) ? R0 : R1;
    char* out = (l & 1) ? R1 : R0;
    init_bias<2>(acc, P + d.b_off[l], nt0, L);
    mma_ring<0, A, A, A, 2, kNT>(acc, in, 0, wl(l), wl(l + 1), nt0, L, ring);
    epilogue<true, TRAIN, 2>(acc, out, nt0, kNT, ss(l), arow(l), wg, L);
    wg_barrier();
  }

  // ---- L5: [emb(p0) (LDS ring R1), h (R0)] -> 256, snake, out -> R1 (the LDS ring is idle
  //      again after mma_embedding's final barrier)
  init_bias<2>(acc, P + d.b_off[L5], nt0, L);
  mma_embedding<false, 2, kNT>(acc, e, 0, R1, sV, sY, sX, wl(L5), wl(L5) + kKSEmb * U, nt0, A_.actF, wg, L, ring);
  mma_ring<0, A, A, A, 2, kNT>(acc, R0, 0, wl(L5) + kKSEmb * U, wl(L6), nt0, L, ring);
  epilogue<true, TRAIN, 2>(acc, R1, nt0, kNT, ss(5), arow(5), wg, L);
  wg_barrier();

  // ---- L6: R1 -> R0, L7: R0 -> R1
  init_bias<2>(acc, P + d.b_off[L6], nt0, L);
  mma_ring<0, A, A, A, 2, kNT>(acc, R1, 0, wl(L6), wl(L7), nt0, L, ring);
  epilogue<true, TRAIN, 2>(acc, R0, nt0, kNT, ss(6), arow(6), wg, L);
  wg_barrier();
  init_bias<2>(acc, P + d.b_off[L7], nt0, L);
  mma_ring<0, A, A, A, 2, kNT>(acc, R0, 0, wl(L7), wl(LF1), nt0, L, ring);
  epilogue<true, TRAIN, 2>(acc, R1, nt0, kNT, ss(7), arow(7), wg, L);
  wg_barrier();

  // ---- F1 = feature_linear1 (linear): R1 -> R0; its fragments are also kept in
  //      registers because P needs f1 again after S and F2 have recycled the regions.
  bf16x8 f1keep[2][kNB][2];
  init_bias<2>(acc, P + d.b_off[LF1], nt0, L);
  mma_ring<0, A, A, A, 2, kNT>(acc, R1, 0, wl(LF1), MULTI ? wl(LS) : nullptr, nt0, L, ring);
  if (!MULTI) wring_fill<1, kNT / 2>(ringp, wl(LP), L.wave, L.lane);
  epilogue<false, TRAIN, 2>(acc, R0, nt0, kNT, nullptr, arow(kActF1), wg, L, MULTI ? f1keep : nullptr);
  wg_barrier();

  f32x16 accp[1][kNB];
  if (MULTI) {
    // ---- S = scale_linears[0]: [f1 (R0), emb(p1..pK-1) (LDS ring R1)] -> 256, snake, out -> R1
    init_bias<2>(acc, P + d.b_off[LS], nt0, L);
    mma_ring<0, A, A, A, 2, kNT>(acc, R0, 0, wl(LS), wl(LS) + A * U, nt0, L, ring);
    for (int p = 1; p < d.K; ++p) {
      const bf16x8* wpp = wl(LS) + (int64_t)(A + (p - 1) * kKSEmb) * U;
      mma_embedding<TRAIN, 2, kNT>(acc, e, p, R1, sV, sY, sX, wpp, (p + 1 < d.K) ? wpp + kKSEmb * U : nullptr, nt0,
                                   A_.actF, wg, L, ring);
    }
    wring_fill<2, kNT>(ring, wl(LF2), nt0, L.lane);        // flies under the epilogue
    epilogue<true, TRAIN, 2>(acc, R1, nt0, kNT, ss(8), arow(kActAS), wg, L);
    wg_barrier();
    // ---- F2 = feature_linear2 (linear): R1 -> R0
    init_bias<2>(acc, P + d.b_off[LF2], nt0, L);
    mma_ring<0, A, A, A, 2, kNT>(acc, R1, 0, wl(LF2), nullptr, nt0, L, ring);
    wring_fill<1, kNT / 2>(ringp, wl(LP), L.wave, L.lane);
    epilogue<false, TRAIN, 2>(acc, R0, nt0, kNT, nullptr, arow(kActF2), wg, L);
    wg_barrier();
    // f1 back into LDS (R1 is idle: every wave passed the barrier after reading a_s)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int bt = 0; bt < kNB; ++bt)
#pragma unroll
        for (int s = 0; s < 2; ++s) lds_store_frag(R1, 2 * (nt0 + nt) + s, bt, L.lane, f1keep[nt][bt][s]);
    wg_barrier();
    // ---- P = pos_linears[0]: [f1 (R1), f2 (R0)] -> 128, snake; one neuron tile per wave
    init_bias<1>(accp, P + d.b_off[LP], L.wave, L);
    mma_ring<0, A, A, A, 1, kNT / 2>(accp, R1, 0, wl(LP), wl(LP) + A * UP, L.wave, L, ringp);
    mma_ring<0, A, A, A, 1, kNT / 2>(accp, R0, 0, wl(LP) + A * UP, nullptr, L.wave, L, ringp);
  } else {
    // ---- NPP_Net_top1: P reads f1 (R0) directly (networks.py:162-170)
    init_bias<1>(accp, P + d.b_off[LP], L.wave, L);
    mma_ring<0, A, A, A, 1, kNT / 2>(accp, R0, 0, wl(LP), nullptr, L.wave, L, ringp);
  }
  epilogue<true, TRAIN, 1>(accp, nullptr, L.wave, kNT / 2, ss(9),
                           TRAIN ? A_.actF + wfmt_array_base(kActKsAP, gridDim.x) : nullptr, wg, L);

  // ---- rgb_linear 128 -> 3 + sigmoid: per-lane partial dot over its 16 features,
  //      half-wave exchange by shuffle, 4-wave reduction through LDS.
  {
    const float* Wr = P + d.w_off[LRGB];
    float part[kNB][3];
#pragma unroll
    for (int bt = 0; bt < kNB; ++bt)
#pragma unroll
      for (int c = 0; c < 3; ++c) part[bt][c] = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = L.wave * 32 + acc_row(r, L.h);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float w = Wr[c * (kW / 2) + k];
#pragma unroll
        for (int bt = 0; bt < kNB; ++bt) part[bt][c] = fmaf(w, accp[0][bt][r], part[bt][c]);
      }
    }
#pragma unroll
    for (int bt = 0; bt < kNB; ++bt)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float v = part[bt][c] + __shfl_xor(part[bt][c], 32, 64);
        if (L.h == 0) sRGB[(L.wave * kRowTile + bt * 32 + L.b) * 3 + c] = v;
      }
    wg_barrier();
    if (L.tid < kRowTile * 3) {
      const int row = L.tid / 3, c = L.tid - row * 3;
      float z = P[d.b_off[LRGB] + c];
#pragma unroll
      for (int w = 0; w < 4; ++w) z += sRGB[(w * kRowTile + row) * 3 + c];
      A_.pred[(row0 + row) * 3 + c] = 1.0f / (1.0f + __expf(-z));   // helpers.py:56 sigmoid
    }
  }
}

}  // namespace npp

using namespace npp;

extern "C" int npp_mlp_fwd(const int32_t* d_coords_yx, int64_t Bp, const npp_embed_cfg* cfg, int width,
                           const void* d_wf, const float* d_params, float* d_pred, void* d_sstash, void* d_actT,
                           void* stream) {
  int rc = check_embed_cfg(cfg, "npp_mlp_fwd");
  if (rc) return rc;
  if (width != NPP_WIDTH) { set_error("npp_mlp_fwd: width %d unsupported (build is %d)", width, NPP_WIDTH); return NPP_ERR_UNSUPPORTED; }
  if (Bp <= 0 || Bp % kRowTile) { set_error("npp_mlp_fwd: Bp=%lld must be a positive multiple of %d", (long long)Bp, kRowTile); return NPP_ERR_ARG; }
  if (!d_coords_yx || !d_wf || !d_params || !d_pred) { set_error("npp_mlp_fwd: null pointer"); return NPP_ERR_ARG; }
  if ((d_sstash == nullptr) != (d_actT == nullptr)) { set_error("npp_mlp_fwd: sstash and actT must both be given or both be NULL"); return NPP_ERR_ARG; }
  if (Bp / kRowTile > 0x7fffffffLL) { set_error("npp_mlp_fwd: Bp too large"); return NPP_ERR_ARG; }
  const EmbedDev e = make_embed_dev(*cfg);
  const NetDesc d = make_desc(cfg->K);
  FwdArgs A{d_coords_yx, Bp, (const bf16x8*)d_wf, d_params, d_pred, (bf16x8*)d_sstash, (char*)d_actT};
  const dim3 grid((unsigned)(Bp / kRowTile)), block(kThreads);
  const bool train = d_sstash != nullptr, multi = cfg->K > 1;
  hipStream_t s = (hipStream_t)stream;
#define NPP_LAUNCH(T, M)                                                                          \
  do {                                                                                            \
    static bool attr_set = false;                                                                 \
    if (!attr_set) {                                                                              \
      hipError_t ea = hipFuncSetAttribute((const void*)mlp_fwd_kernel<T, M>,                      \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, kSmemFwd);  \
      if (ea != hipSuccess) { set_error("npp_mlp_fwd: smem attr: %s", hipGetErrorString(ea)); return NPP_ERR_LAUNCH; } \
      attr_set = true;                                                                            \
    }                                                                                             \
    hipLaunchKernelGGL((mlp_fwd_kernel<T, M>), grid, block, kSmemFwd, s, A, e, d);                \
  } while (0)
  if (train) { if (multi) NPP_LAUNCH(true, true); else NPP_LAUNCH(true, false); }
  else { if (multi) NPP_LAUNCH(false, true); else NPP_LAUNCH(false, false); }
#undef NPP_LAUNCH
  return check_launch("npp_mlp_fwd");
}
