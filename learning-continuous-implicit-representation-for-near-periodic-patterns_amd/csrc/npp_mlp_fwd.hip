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
// the embedding slots -- feature-major [feature][row] in bf16, which is the k-contiguous
// operand layout npp_mlp_wgrad needs.
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
  __bf16* actT;      // nullable
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

// KS k-steps whose activation fragments sit in an LDS region; weight fragments of this
// wave's NTW tiles streamed from the forward pack ([ks][NT][64] units of 16 B).
template <int KS, int NTW, int NT>
__device__ __forceinline__ void mma_region(f32x16 (&acc)[NTW][kNB], const char* region, int ks_lds0,
                                           const bf16x8* __restrict__ wp, int nt0, const Lane& L) {
#pragma unroll 2
  for (int ks = 0; ks < KS; ++ks) {
    bf16x8 w[NTW], x[kNB];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) w[nt] = wp[(ks * NT + nt0 + nt) * 64 + L.lane];
#pragma unroll
    for (int bt = 0; bt < kNB; ++bt) x[bt] = lds_frag(region, ks_lds0 + ks, bt, L.lane);
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
      for (int bt = 0; bt < kNB; ++bt) acc[nt][bt] = mfma_bf16(w[nt], x[bt], acc[nt][bt]);
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

// Accumulate one proposal's 30 embedding k-steps.  ring = 32 KiB LDS (two 16 KiB chunk
// buffers).  Caller guarantees sV is free to overwrite and ring is free; on return every
// wave has passed a barrier after its last ring / sV read.
template <bool STORE_EMB, int NTW, int NT>
__device__ __forceinline__ void mma_embedding(f32x16 (&acc)[NTW][kNB], const EmbedDev& e, int p, char* ring,
                                              float* sV, const float* sY, const float* sX,
                                              const bf16x8* __restrict__ wp, int nt0, __bf16* actT, int64_t Bp,
                                              int64_t row0, const Lane& L) {
  gen_warp(e, p, sV, sY, sX, L);
  __syncthreads();
  auto gen_chunk = [&](int c) {
    char* buf = ring + (c & 1) * kChunkBytes;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int ksl = 2 * L.wave + q;        // wave-uniform
      const int ks = kChunkKS * c + ksl;
      if (ks < kKSEmb) {
#pragma unroll
        for (int bt = 0; bt < kNB; ++bt) {
          const bf16x8 f = gen_emb_frag(e, sV, ks, bt, L);
          lds_store_frag(buf, ksl, bt, L.lane, f);
          if (STORE_EMB) {
            __bf16* dst = actT + ((int64_t)(kActEmbRow0 + p * kEmbSlots + ks * 16 + L.h * 8)) * Bp + row0 + bt * 32 + L.b;
#pragma unroll
            for (int j = 0; j < 8; ++j) dst[(int64_t)j * Bp] = f[j];
          }
        }
      }
    }
  };
  gen_chunk(0);
  __syncthreads();
#pragma unroll 1
  for (int c = 0; c < kNChunks; ++c) {
    if (c + 1 < kNChunks) gen_chunk(c + 1);
    const char* buf = ring + (c & 1) * kChunkBytes;
    const int nks = (c == kNChunks - 1) ? (kKSEmb - kChunkKS * (kNChunks - 1)) : kChunkKS;
#pragma unroll 2
    for (int ksl = 0; ksl < nks; ++ksl) {
      const int ks = kChunkKS * c + ksl;
      bf16x8 w[NTW], x[kNB];
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) w[nt] = wp[(ks * NT + nt0 + nt) * 64 + L.lane];
#pragma unroll
      for (int bt = 0; bt < kNB; ++bt) x[bt] = lds_frag(buf, ksl, bt, L.lane);
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int bt = 0; bt < kNB; ++bt) acc[nt][bt] = mfma_bf16(w[nt], x[bt], acc[nt][bt]);
    }
    __syncthreads();
  }
}

// Epilogue of a 256-wide (NTW=2 per wave) or 128-wide (NTW=1) layer.
//  SNAKE: apply x + sin^2 x; else linear.   out: LDS region that receives the bf16
//  fragments (k-step 2*ntile+s of the next layer), may be null.
//  TRAIN: stash the derivative (fragment order) and the activation (feature-major).
template <bool SNAKE, bool TRAIN, int NTW>
__device__ __forceinline__ void epilogue(f32x16 (&acc)[NTW][kNB], char* out, int nt0, int ntl /*tiles in layer*/,
                                         bf16x8* sstash_layer, __bf16* actT_rows, int64_t Bp, int64_t row0,
                                         int wg, const Lane& L, bf16x8 (*keep)[kNB][2] = nullptr) {
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
      }
      if (TRAIN) {
        __bf16* dst = actT_rows + (int64_t)(ntg * 32) * Bp + row0 + bt * 32 + L.b;
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[(int64_t)acc_row(r, L.h) * Bp] = (__bf16)a[r];
      }
    }
  }
}

template <bool TRAIN, bool MULTI>
__global__ __launch_bounds__(kThreads, 2) void mlp_fwd_kernel(FwdArgs A, EmbedDev e_arg, NetDesc d) {
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
  const int64_t Bp = A.Bp;
  const float* P = A.params;
  const bf16x8* wf = A.wf;
  const int nt0 = 2 * L.wave;            // this wave's neuron tiles in 256-wide layers

  if (L.tid < kRowTile) {
    const int2 c = ((const int2*)A.coords)[row0 + L.tid];
    sY[L.tid] = (float)c.x;              // (row=y, col=x)
    sX[L.tid] = (float)c.y;
  }
  __syncthreads();

  auto ss = [&](int slot) -> bf16x8* {
    return TRAIN ? (bf16x8*)((char*)A.sstash + sstash_off_bytes(slot, Bp)) : nullptr;
  };
  auto arow = [&](int idx) -> __bf16* { return TRAIN ? A.actT + (int64_t)idx * kW * Bp : nullptr; };

  f32x16 acc[2][kNB];

  // ---- L0: emb(p0) -> 256, snake.  ring = R1, out -> R0
  init_bias<2>(acc, P + d.b_off[L0], nt0, L);
  mma_embedding<TRAIN, 2, kNT>(acc, e, 0, R1, sV, sY, sX, wf + d.wf_off[L0], nt0, A.actT, Bp, row0, L);
  epilogue<true, TRAIN, 2>(acc, R0, nt0, kNT, ss(0), arow(0), Bp, row0, wg, L);
  __syncthreads();

  // ---- L1..L4: 256 -> 256, snake, ping-pong R0 -> R1 -> R0 -> R1 -> R0
#pragma unroll
  for (int l = L1; l <= L4; ++l) {
    char* in = (l & 1) ? R0 : R1;
    char* out = (l & 1) ? R1 : R0;
    init_bias<2>(acc, P + d.b_off[l], nt0, L);
    mma_region<kKSAct, 2, kNT>(acc, in, 0, wf + d.wf_off[l], nt0, L);
    epilogue<true, TRAIN, 2>(acc, out, nt0, kNT, ss(l), arow(l), Bp, row0, wg, L);
    __syncthreads();
  }

  // ---- L5: [emb(p0) (ring R1), h (R0)] -> 256, snake, out -> R1 (ring is idle again
  //      after mma_embedding's final barrier)
  init_bias<2>(acc, P + d.b_off[L5], nt0, L);
  mma_embedding<false, 2, kNT>(acc, e, 0, R1, sV, sY, sX, wf + d.wf_off[L5], nt0, A.actT, Bp, row0, L);
  mma_region<kKSAct, 2, kNT>(acc, R0, 0, wf + d.wf_off[L5] + (int64_t)kKSEmb * kNT * 64, nt0, L);
  epilogue<true, TRAIN, 2>(acc, R1, nt0, kNT, ss(5), arow(5), Bp, row0, wg, L);
  __syncthreads();

  // ---- L6: R1 -> R0, L7: R0 -> R1
  init_bias<2>(acc, P + d.b_off[L6], nt0, L);
  mma_region<kKSAct, 2, kNT>(acc, R1, 0, wf + d.wf_off[L6], nt0, L);
  epilogue<true, TRAIN, 2>(acc, R0, nt0, kNT, ss(6), arow(6), Bp, row0, wg, L);
  __syncthreads();
  init_bias<2>(acc, P + d.b_off[L7], nt0, L);
  mma_region<kKSAct, 2, kNT>(acc, R0, 0, wf + d.wf_off[L7], nt0, L);
  epilogue<true, TRAIN, 2>(acc, R1, nt0, kNT, ss(7), arow(7), Bp, row0, wg, L);
  __syncthreads();

  // ---- F1 = feature_linear1 (linear): R1 -> R0; its fragments are also kept in
  //      registers because P needs f1 again after S and F2 have recycled the regions.
  bf16x8 f1keep[2][kNB][2];
  init_bias<2>(acc, P + d.b_off[LF1], nt0, L);
  mma_region<kKSAct, 2, kNT>(acc, R1, 0, wf + d.wf_off[LF1], nt0, L);
  epilogue<false, TRAIN, 2>(acc, R0, nt0, kNT, nullptr, arow(kActF1), Bp, row0, wg, L, MULTI ? f1keep : nullptr);
  __syncthreads();

  f32x16 accp[1][kNB];
  if (MULTI) {
    // ---- S = scale_linears[0]: [f1 (R0), emb(p1..pK-1) (ring R1)] -> 256, snake, out -> R1
    init_bias<2>(acc, P + d.b_off[LS], nt0, L);
    mma_region<kKSAct, 2, kNT>(acc, R0, 0, wf + d.wf_off[LS], nt0, L);
    for (int p = 1; p < d.K; ++p)
      mma_embedding<TRAIN, 2, kNT>(acc, e, p, R1, sV, sY, sX,
                                   wf + d.wf_off[LS] + (int64_t)(kKSAct + (p - 1) * kKSEmb) * kNT * 64, nt0,
                                   A.actT, Bp, row0, L);
    epilogue<true, TRAIN, 2>(acc, R1, nt0, kNT, ss(8), arow(kActAS), Bp, row0, wg, L);
    __syncthreads();
    // ---- F2 = feature_linear2 (linear): R1 -> R0
    init_bias<2>(acc, P + d.b_off[LF2], nt0, L);
    mma_region<kKSAct, 2, kNT>(acc, R1, 0, wf + d.wf_off[LF2], nt0, L);
    epilogue<false, TRAIN, 2>(acc, R0, nt0, kNT, nullptr, arow(kActF2), Bp, row0, wg, L);
    __syncthreads();
    // f1 back into LDS (R1 is idle: every wave passed the barrier after reading a_s)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int bt = 0; bt < kNB; ++bt)
#pragma unroll
        for (int s = 0; s < 2; ++s) lds_store_frag(R1, 2 * (nt0 + nt) + s, bt, L.lane, f1keep[nt][bt][s]);
    __syncthreads();
    // ---- P = pos_linears[0]: [f1 (R1), f2 (R0)] -> 128, snake; one neuron tile per wave
    init_bias<1>(accp, P + d.b_off[LP], L.wave, L);
    mma_region<kKSAct, 1, kNT / 2>(accp, R1, 0, wf + d.wf_off[LP], L.wave, L);
    mma_region<kKSAct, 1, kNT / 2>(accp, R0, 0, wf + d.wf_off[LP] + (int64_t)kKSAct * (kNT / 2) * 64, L.wave, L);
  } else {
    // ---- NPP_Net_top1: P reads f1 (R0) directly (networks.py:162-170)
    init_bias<1>(accp, P + d.b_off[LP], L.wave, L);
    mma_region<kKSAct, 1, kNT / 2>(accp, R0, 0, wf + d.wf_off[LP], L.wave, L);
  }
  epilogue<true, TRAIN, 1>(accp, nullptr, L.wave, kNT / 2, ss(9), TRAIN ? A.actT + (int64_t)kActAP * kW * Bp : nullptr,
                           Bp, row0, wg, L);

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
    __syncthreads();
    if (L.tid < kRowTile * 3) {
      const int row = L.tid / 3, c = L.tid - row * 3;
      float z = P[d.b_off[LRGB] + c];
#pragma unroll
      for (int w = 0; w < 4; ++w) z += sRGB[(w * kRowTile + row) * 3 + c];
      A.pred[(row0 + row) * 3 + c] = 1.0f / (1.0f + __expf(-z));   // helpers.py:56 sigmoid
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
  FwdArgs A{d_coords_yx, Bp, (const bf16x8*)d_wf, d_params, d_pred, (bf16x8*)d_sstash, (__bf16*)d_actT};
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
