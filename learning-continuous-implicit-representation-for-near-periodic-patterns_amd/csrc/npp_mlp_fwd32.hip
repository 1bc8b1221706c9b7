// npp_mlp_fwd32.hip -- the fused embedder + coordinate-MLP forward in EXACT fp32 (BASELINE config c4: "1024 x 1024 ...
// full-resolution coord grid, fp32").  Same function as npp_mlp_fwd.hip (models/embedder.py:102-148 + :11-56,
// models/networks.py:56-95 / :145-173, models/helpers.py:55-56), same fused structure, but every contraction runs on
// v_mfma_f32_32x32x2_f32: f32 operands, f32 accumulate, bit for bit a k-ordered fmaf chain (157 TFLOP/s peak, 1/16 of the bf16
// rate) -- the reference's own arithmetic type, no bf16 rounding anywhere.  Inference only (the full-image render of
// train.py:270-331); training keeps the bf16 chain (BASELINE c2).
//
// Structure: one 256-thread workgroup = 64 pixel rows (two 32-column batch tiles), GEMMs transposed (Z^T = W X^T): the
// weights are the A operand (lane l: W[32 nt + (l & 31)][k-step's feature l >> 5]), streamed from L2 as 16-byte loads that
// carry FOUR consecutive k-steps each; the activations are the B operand (lane l: X[feature l >> 5][row l & 31]).  A 32 x 32
// accumulator tile holds feature acc_row(r, h) of row (lane & 31) in register r of lane half h, so register r of a tile IS
// the B operand of the k-step that contracts features (acc_row(r, 0), acc_row(r, 1)) = (n, n + 4): hidden activations cross
// LDS as plain fp32 [feature][64 rows] (one region, written in place behind a barrier), read back with one conflict-free
// ds_read_b32 per k-step and batch tile.  The 462 Fourier features of a proposal are never materialised: a k-step
// contracts (sin(f v_i), cos(f v_i)) -- lane half 0 / 1, one v_sin each -- and every wave generates its own B operands in
// registers right before the MFMAs that consume them (4 x redundant vector work that hides under the 64-cycle MFMAs, and no
// LDS ring, no barrier inside an embedding pass): 231 k-steps per proposal, no padding slots.
//
// Algorithmic work: 2 * ((K+1)*462*256 + 11*256^2 + 384) FLOP per row.
#include "npp_chain32.h"

namespace npp {

EmbedDev make_embed_dev(const npp_embed_cfg& c);
int check_embed_cfg(const npp_embed_cfg* c, const char* who);

constexpr int kT32 = 32 * kNT;                     // threads: 2 neuron tiles per wave (4 waves at W = 256, 8 at W = 512)
constexpr int kRegion32 = kW * kRowTile * 4;       // 64 KiB: 256 features x 64 rows fp32
struct WarpEnt32 { float cs, sn, per, inv_per, phase, lin; };
constexpr int kSmem32 = kRegion32 + 22 * kRowTile * 4 + 2 * kRowTile * 4 + ((sizeof(EmbedDev) + 15) / 16) * 16 +
                        NPP_MAX_K * 22 * (int)sizeof(WarpEnt32);

struct Fwd32Args {
  const int32_t* coords;
  int64_t Bp;
  const float* w32;          // fp32 pack (npp_layout.h: Desc32)
  const float* params;
  float* pred;
  int32_t out_act;
};

// B operands generated from the 22 warped coordinates of one proposal (sV[i][row], fp32): k-step q < 220 contracts
// (sin, cos)(f_{q / 22} v_{q % 22}); k-steps 220..230 the raw block (v_{2 (q - 220)}, v_{2 (q - 220) + 1}); k-step 231 is padding
struct EmbSrc32 {
  const float* sV;           // + (lane & 31)
  const float* fr;           // freq / 2 pi
  int h;
  __device__ __forceinline__ void frag(int g, int e, float (&b)[kNB]) const {
    const int q = 4 * g + e;                       // wave-uniform
    if (q < 220) {
      const int fj = (q * 2979) >> 16, i = q - 22 * fj;             // q / 22 exact for q < 240
      const float f = fr[fj], ph = h ? 0.25f : 0.0f;
#pragma unroll
      for (int bt = 0; bt < kNB; ++bt) b[bt] = __builtin_amdgcn_sinf(fmaf(sV[i * kRowTile + bt * 32], f, ph));
    } else {
      const int i = 2 * (q - 220) + h;
      const bool ok = i < 22;
#pragma unroll
      for (int bt = 0; bt < kNB; ++bt) b[bt] = ok ? sV[(ok ? i : 0) * kRowTile + bt * 32] : 0.0f;
    }
  }
};

template <bool MULTI>
__global__ __launch_bounds__(kT32, 2) void mlp_fwd32_kernel(Fwd32Args A_, EmbedDev e_arg, NetDesc d, Desc32 d32) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* R = smem;
  float* sV = (float*)(smem + kRegion32);
  float* sY = sV + 22 * kRowTile;
  float* sX = sY + kRowTile;
  EmbedDev& ed = *(EmbedDev*)(sX + kRowTile);
  WarpEnt32* tWarp = (WarpEnt32*)((char*)&ed + ((sizeof(EmbedDev) + 15) / 16) * 16);
  if (threadIdx.x == 0) {
    const uint32_t* src = (const uint32_t*)&e_arg;
    uint32_t* dst = (uint32_t*)&ed;
#pragma unroll
    for (int i = 0; i < (int)(sizeof(EmbedDev) / 4); ++i) dst[i] = src[i];
  }
  wg_barrier();
  for (int wi = threadIdx.x; wi < ed.K * 22; wi += kT32) {
    const int p = wi / 22, i = wi - p * 22, ori = i >= 11, ii = ori ? i - 11 : i;
    WarpEnt32 w{0.0f, 0.0f, 1.0f, 1.0f, 0.0f, 0.0f};
    if (ii == 0) {
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
  const int tid = threadIdx.x, lane = tid & 63, b = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t row0 = (int64_t)blockIdx.x * kRowTile;
  const float* P = A_.params;
  const int nt0 = 2 * wave;
  if (tid < kRowTile) {
    const int2 c = ((const int2*)A_.coords)[row0 + tid];
    sY[tid] = (float)c.x;
    sX[tid] = (float)c.y;
  }
  wg_barrier();

  // the 22 warped coordinates of proposal p -> sV (the same arithmetic as the bf16 chain's gen_warp)
  auto gen_warp = [&](int p) {
    const float y = sY[lane], x = sX[lane];
    for (int i = wave; i < 22; i += kT32 / 64) {
      const WarpEnt32 w = tWarp[p * 22 + i];
      const float t = __fadd_rn(__fmul_rn(y, w.cs), __fmul_rn(x, w.sn));
      const float qf = floorf(t * w.inv_per);
      float r = fmaf(-qf, w.per, t);
      r = r < 0.0f ? r + w.per : r;
      r = r >= w.per ? r - w.per : r;
      const float sv = __builtin_amdgcn_sinf(fmaf(r, w.inv_per, w.phase));
      sV[i * kRowTile + lane] = w.lin != 0.0f ? t - 1.0f : sv;
    }
  };

  const wrsrc_t rsrc = make_wrsrc(A_.w32, d32.total16);
  const ActSrc32 act{R + ((4 * h) * kRowTile + b) * 4};
  const EmbSrc32 emb{sV + b, ed.freq_rev, h};
  auto w32 = [&](int l) -> uint32_t { return (uint32_t)d32.off16[l]; };
  constexpr int GA = kW / 8;                       // 32 groups per 256 activation features
  constexpr int GE = 58;                           // 232 k-steps per proposal
  constexpr uint32_t U = kNT * 64;                 // 16-byte units per k-step group of a 256-wide layer

  f32x16 acc[2][kNB];
  // ---- L0
  gen_warp(0);
  bias32(acc, P + d.b_off[L0], nt0, h);
  wg_barrier();
  part32<2, kNT>(acc, rsrc, w32(L0), GE, nt0, lane, emb);
  epi32<true, 2>(acc, R, nt0, b, h);               // nobody reads R yet
  wg_barrier();
  // ---- L1..L4 (in place: read R, barrier, write R)
#pragma unroll 1
  for (int l = L1; l <= L4; ++l) {
    bias32(acc, P + d.b_off[l], nt0, h);
    part32<2, kNT>(acc, rsrc, w32(l), GA, nt0, lane, act);
    wg_barrier();
    epi32<true, 2>(acc, R, nt0, b, h);
    wg_barrier();
  }
  // ---- L5 = [emb(p0), h]   (sV still holds proposal 0)
  bias32(acc, P + d.b_off[L5], nt0, h);
  part32<2, kNT>(acc, rsrc, w32(L5), GE, nt0, lane, emb);
  part32<2, kNT>(acc, rsrc, w32(L5) + GE * U, GA, nt0, lane, act);
  wg_barrier();
  epi32<true, 2>(acc, R, nt0, b, h);
  wg_barrier();
#pragma unroll 1
  for (int l = L6; l <= L7; ++l) {
    bias32(acc, P + d.b_off[l], nt0, h);
    part32<2, kNT>(acc, rsrc, w32(l), GA, nt0, lane, act);
    wg_barrier();
    epi32<true, 2>(acc, R, nt0, b, h);
    wg_barrier();
  }
  // ---- F1 (linear); its tiles stay in registers: P needs f1 again after S / F2 have recycled the region
  f32x16 f1[2][kNB];
  bias32(f1, P + d.b_off[LF1], nt0, h);
  part32<2, kNT>(f1, rsrc, w32(LF1), GA, nt0, lane, act);
  wg_barrier();
  epi32<false, 2>(f1, R, nt0, b, h);
  wg_barrier();

  f32x16 accp[1][kNB];
  if (MULTI) {
    // ---- S = [f1 (R), emb(p1..)]
    bias32(acc, P + d.b_off[LS], nt0, h);
    part32<2, kNT>(acc, rsrc, w32(LS), GA, nt0, lane, act);
    for (int p = 1; p < d.K; ++p) {
      wg_barrier();                                // every wave is done with the previous proposal's sV
      gen_warp(p);
      wg_barrier();
      part32<2, kNT>(acc, rsrc, w32(LS) + (uint32_t)(GA + (p - 1) * GE) * U, GE, nt0, lane, emb);
    }
    wg_barrier();
    epi32<true, 2>(acc, R, nt0, b, h);             // a_s
    wg_barrier();
    // ---- F2 (linear) -> R (in place)
    bias32(acc, P + d.b_off[LF2], nt0, h);
    part32<2, kNT>(acc, rsrc, w32(LF2), GA, nt0, lane, act);
    wg_barrier();
    epi32<false, 2>(acc, R, nt0, b, h);            // f2
    wg_barrier();
    // ---- P = [f1, f2] -> 128: one neuron tile per wave; pack order: the f2 k-steps first, then f1
    bias32(accp, P + d.b_off[LP], wave, h);
    part32<1, kNT / 2>(accp, rsrc, w32(LP), GA, wave, lane, act);
    wg_barrier();
    epi32<false, 2>(f1, R, nt0, b, h);             // f1 back into the region
    wg_barrier();
    part32<1, kNT / 2>(accp, rsrc, w32(LP) + GA * (U / 2), GA, wave, lane, act);
  } else {
    bias32(accp, P + d.b_off[LP], wave, h);
    part32<1, kNT / 2>(accp, rsrc, w32(LP), GA, wave, lane, act);
  }
  epi32<true, 1>(accp, nullptr, wave, b, h);       // a_p stays in registers

  // ---- rgb_linear 128 -> 3 + output activation (models/helpers.py:55-58)
  wg_barrier();
  float* sRGB = (float*)R;                         // [4 waves][64 rows][3]
  {
    const float* Wr = P + d.w_off[LRGB];
    float part[kNB][3];
#pragma unroll
    for (int bt = 0; bt < kNB; ++bt)
#pragma unroll
      for (int c = 0; c < 3; ++c) part[bt][c] = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = wave * 32 + acc_row(r, h);
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
        if (h == 0) sRGB[(wave * kRowTile + bt * 32 + b) * 3 + c] = v;
      }
  }
  wg_barrier();
  if (tid < kRowTile * 3) {
    const int row = tid / 3, c = tid - row * 3;
    float z = P[d.b_off[LRGB] + c];
#pragma unroll
    for (int w = 0; w < kNT / 2; ++w) z += sRGB[(w * kRowTile + row) * 3 + c];
    const float o = A_.out_act == 1 ? 1.0f / (1.0f + expf(-z)) : (A_.out_act == 2 ? tanhf(z) : z);
    A_.pred[(row0 + row) * 3 + c] = o;
  }
}

// fp32 pack: unit u (16 bytes) = [layer][k-step group g][neuron tile nt][lane]: element e = W[32 nt + (lane & 31)][col32(l, 4 g + e, lane >> 5)]
__global__ void pack32_kernel(const float* __restrict__ P, float* __restrict__ out, NetDesc d, Desc32 d32) {
  const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= d32.total16) return;
  int l = 0;
  for (int q = 0; q < kNumLayers; ++q)
    if (d32.present[q] && u >= d32.off16[q]) l = q;
  const int64_t r = u - d32.off16[l];
  const int lane = (int)(r & 63);
  const int nt = (int)((r >> 6) % d32.nt[l]);
  const int g = (int)((r >> 6) / d32.nt[l]);
  const int row = nt * 32 + (lane & 31), h = lane >> 5;
  f32x4_t o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int col = col32(d.K, l, 4 * g + e, h);
    o[e] = col < 0 ? 0.0f : P[d.w_off[l] + (int64_t)row * d.n_in[l] + col];
  }
  ((f32x4_t*)out)[u] = o;
}

}  // namespace npp

using namespace npp;

extern "C" int64_t npp_pack32_bytes(int K, int width) {
  if (K < 1 || K > NPP_MAX_K || width != NPP_WIDTH) { set_error("npp_pack32_bytes: K=%d width=%d", K, width); return -1; }
  return make_desc32(K).total16 * 16;
}

extern "C" int npp_pack_weights32(const float* d_params, void* d_w32, int K, int width, void* stream) {
  if (K < 1 || K > NPP_MAX_K || width != NPP_WIDTH || !d_params || !d_w32) { set_error("npp_pack_weights32: bad argument"); return NPP_ERR_ARG; }
  const Desc32 d32 = make_desc32(K);
  hipLaunchKernelGGL(pack32_kernel, dim3((unsigned)((d32.total16 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_params,
                     (float*)d_w32, make_desc(K), d32);
  return check_launch("npp_pack_weights32");
}

extern "C" int npp_mlp_fwd32(const int32_t* d_coords_yx, int64_t Bp, const npp_embed_cfg* cfg, int width, const void* d_w32,
                             const float* d_params, float* d_out, int out_act, void* stream) {
  int rc = check_embed_cfg(cfg, "npp_mlp_fwd32");
  if (rc) return rc;
  if (width != NPP_WIDTH) { set_error("npp_mlp_fwd32: width %d unsupported (build is %d)", width, NPP_WIDTH); return NPP_ERR_UNSUPPORTED; }
  if (Bp <= 0 || Bp % kRowTile || Bp / kRowTile > 0x7fffffffLL) { set_error("npp_mlp_fwd32: Bp=%lld must be a positive multiple of %d", (long long)Bp, kRowTile); return NPP_ERR_ARG; }
  if (!d_coords_yx || !d_w32 || !d_params || !d_out || out_act < 0 || out_act > 2) { set_error("npp_mlp_fwd32: bad argument"); return NPP_ERR_ARG; }
  Fwd32Args A{d_coords_yx, Bp, (const float*)d_w32, d_params, d_out, out_act};
  const EmbedDev e = make_embed_dev(*cfg);
  const NetDesc d = make_desc(cfg->K);
  const Desc32 d32 = make_desc32(cfg->K);
  const dim3 grid((unsigned)(Bp / kRowTile)), block(kT32);
  hipStream_t s = (hipStream_t)stream;
  if (cfg->K > 1) {
    static SmemOnce once;
    if (!smem_attr(once, (const void*)mlp_fwd32_kernel<true>, kSmem32)) { set_error("npp_mlp_fwd32: smem attribute"); return NPP_ERR_LAUNCH; }
    hipLaunchKernelGGL((mlp_fwd32_kernel<true>), grid, block, kSmem32, s, A, e, d, d32);
  } else {
    static SmemOnce once;
    if (!smem_attr(once, (const void*)mlp_fwd32_kernel<false>, kSmem32)) { set_error("npp_mlp_fwd32: smem attribute"); return NPP_ERR_LAUNCH; }
    hipLaunchKernelGGL((mlp_fwd32_kernel<false>), grid, block, kSmem32, s, A, e, d, d32);
  }
  return check_launch("npp_mlp_fwd32");
}
