// npp_embed.hip -- K1: stand-alone periodicity-aware embedder (HBM-write-bound).
// Replaces Embedder_periodic.embed + Embedder.embed + the K-way cat
// (models/embedder.py:140-148, :51-56; NPP_completion/train.py:93-105).
//
// Algorithmic bytes per pixel: 8 B coords in + 4*K*462 B out (fp32) or 2*K*462 B (bf16).
#include "npp_common.h"
#include <math.h>

namespace npp {

EmbedDev make_embed_dev(const npp_embed_cfg& c) {
  EmbedDev e{};
  e.K = c.K; e.H = c.H; e.W = c.W;
  e.inv_w = 1.0f / (float)c.W; e.inv_h = 1.0f / (float)c.H;
  for (int k = 0; k < c.K && k < NPP_MAX_K; ++k)
    for (int o = 0; o < 2; ++o) {
      // torch.deg2rad (f32 multiply by pi/180) then torch.cos/sin in f32 (embedder.py:122-127)
      const float th = c.angles_deg[k][o] * (float)(M_PI / 180.0);
      e.cs[k][o] = cosf(th);
      e.sn[k][o] = sinf(th);
      for (int j = 0; j < NPP_N_OFF; ++j) e.per[k][o][j] = c.periods[k][o] + c.offsets[j];
    }
  for (int j = 0; j < NPP_N_FREQ; ++j) {
    e.freq[j] = c.freqs[j];
    e.freq_rev[j] = (float)((double)c.freqs[j] / (2.0 * M_PI));
  }
  return e;
}

int check_embed_cfg(const npp_embed_cfg* c, const char* who) {
  if (!c) { set_error("%s: null cfg", who); return NPP_ERR_ARG; }
  if (c->K < 1 || c->K > NPP_MAX_K || c->H < 1 || c->W < 1) {
    set_error("%s: bad cfg K=%d H=%d W=%d", who, c->K, c->H, c->W);
    return NPP_ERR_ARG;
  }
  for (int k = 0; k < c->K; ++k)
    for (int o = 0; o < 2; ++o)
      for (int j = 0; j < NPP_N_OFF; ++j)
        if (!(c->periods[k][o] + c->offsets[j] > 0.0f)) {
          set_error("%s: period %g + offset %g must be > 0", who, c->periods[k][o], c->offsets[j]);
          return NPP_ERR_ARG;
        }
  return NPP_OK;
}

#ifndef NPP_EMB_ROWS
#define NPP_EMB_ROWS 32
#endif
constexpr int kEmbRows = NPP_EMB_ROWS;     // rows per workgroup (one per lane in phase 1; 32: the upper half-waves idle there)
constexpr int kEmbThreads = 256;
#ifndef NPP_EMB_STAGE
#define NPP_EMB_STAGE 2
#endif
constexpr int kEmbStage = NPP_EMB_STAGE;     // rows staged in LDS per pass of the exact-fp32 form (2 x 5 x 462 floats = 18 KiB; measured
                                              // 1024^2 K=3: 2 rows 1.47 ms, 4 rows 1.62, 8 rows 2.69 -- occupancy; 64 rows per workgroup 1.56)
constexpr int kSvStride = NPP_MAX_K * 22 + 1;   // odd stride: conflict-free row-per-lane writes

// Per output column (reference order, models/embedder.py:41-44,56): index of its warped coordinate,
// Fourier frequency and kind -- built once per workgroup so the streaming loop does no index
// arithmetic beyond one 8-byte LDS read per value.
struct ColEnt { float f; int code; };            // code = v index | identity << 16 | cos << 17

// Phase 1: the K*22 warped values of 64 rows into LDS: lane = row, the warp index is
// wave-uniform (wave w takes jobs w, w+4, ...) so the constant tables are read through
// scalar loads.  Phase 2: every thread produces column PAIRS (462 is even) so stores are
// 8 B (fp32) / 4 B (bf16) and fully coalesced.
template <bool PRECISE, bool BF16OUT>
__global__ __launch_bounds__(kEmbThreads) void embed_kernel(const int32_t* __restrict__ coords,
                                                            int64_t N, EmbedDev e,
                                                            void* __restrict__ out) {
  __shared__ float sv[kEmbRows * kSvStride];
  constexpr bool kPairForm = PRECISE && !BF16OUT;             // exact-fp32 form: units instead of the column table
  __shared__ ColEnt tab[kPairForm ? 1 : NPP_MAX_K * kE];
  __shared__ float sfreq[NPP_N_FREQ];   // lane-varying index: keep it out of the kernarg struct
  typedef __attribute__((ext_vector_type(4))) float f32x4v;
  __shared__ __attribute__((aligned(16))) float sstage[kPairForm ? kEmbStage * NPP_MAX_K * kE : 4];
  const int tid = threadIdx.x;
  if (tid == 0) {
#pragma unroll
    for (int j = 0; j < NPP_N_FREQ; ++j) sfreq[j] = PRECISE ? e.freq[j] : e.freq_rev[j];
  }
  __syncthreads();
  const int K = e.K;
  for (int c = tid; !kPairForm && c < K * kE; c += kEmbThreads) {
    const int p = c / kE, cc = c - p * kE, blk = cc / 22, i = cc - blk * 22;
    ColEnt en;
    en.code = (p * 22 + i) | (blk == 0 ? 1 << 16 : 0) | ((blk > 0 && ((blk - 1) & 1)) ? 1 << 17 : 0);
    en.f = blk == 0 ? 0.0f : sfreq[(blk - 1) >> 1];
    tab[c] = en;
  }
  const int64_t row0 = (int64_t)blockIdx.x * kEmbRows;
  {
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int64_t r = row0 + lane;
    const int2 c = (r < N && lane < kEmbRows) ? ((const int2*)coords)[r] : make_int2(0, 0);
    const float y = (float)c.x, x = (float)c.y;   // (row=y, col=x): first member is y
    for (int job = wave; job < K * 22; job += 4)
      if (lane < kEmbRows) sv[lane * kSvStride + job] = warp_value<PRECISE>(e, job / 22, job % 22, y, x);
  }
  __syncthreads();
  if (PRECISE && !BF16OUT) {
    // Phase 2, exact-fp32 form (BASELINE config c4): the arithmetic, not the stores, bounded this path (2.4 -> 3.3 TB/s with
    // the branch-free polynomial; ~20 vector instructions per value).  sin(f v) and cos(f v) of one (frequency, coordinate)
    // pair sit 22 columns apart in a row, so a thread takes a UNIT = (row, proposal, block, coordinate pair i, i + 1) and
    // produces both functions of both arguments from two range reductions: 11 instead of 20 instructions per value.  The
    // values of kEmbStage rows are staged in LDS (the 8-byte pieces of a unit are 88 bytes apart: written straight to
    // memory they ran at 1.9 TB/s) and leave as the same 16-byte streaming stores as the other forms: 3.3 -> 4.0 TB/s.
    typedef __attribute__((ext_vector_type(2))) float f32x2v;
    float* stage = sstage;
    const int rowlen = K * kE;
    const int upr = K * 121;                                  // units per row: per proposal 11 identity pairs + 10 x 11
    for (int r0 = 0; r0 < kEmbRows; r0 += kEmbStage) {
      int t = tid, rr = 0;
      while (t >= upr) { t -= upr; ++rr; }
      while (rr < kEmbStage) {
        const int p = t / 121, tt = t - p * 121, blk = tt / 11, i = 2 * (tt - blk * 11);
        const float v0 = sv[(r0 + rr) * kSvStride + p * 22 + i], v1 = sv[(r0 + rr) * kSvStride + p * 22 + i + 1];
        float* orow = stage + rr * rowlen + p * kE;
        if (blk == 0) {
          *(f32x2v*)(orow + i) = f32x2v{v0, v1};
        } else {
          const float fr = sfreq[blk - 1];
          float s0, c0, s1, c1;
          sincos_pi2_both(v0 * fr, s0, c0);
          sincos_pi2_both(v1 * fr, s1, c1);
          *(f32x2v*)(orow + (2 * blk - 1) * 22 + i) = f32x2v{s0, s1};
          *(f32x2v*)(orow + (2 * blk) * 22 + i) = f32x2v{c0, c1};
        }
        t += kEmbThreads;
        while (t >= upr) { t -= upr; ++rr; }
      }
      __syncthreads();
      const int n4 = kEmbStage * rowlen / 4;                   // rowlen is even, kEmbStage a multiple of 2
      const int64_t base = (row0 + r0) * (int64_t)rowlen;
      const int64_t lim = N * (int64_t)rowlen;
      for (int q = tid; q < n4; q += kEmbThreads) {
        const f32x4v w = *(const f32x4v*)(stage + 4 * q);
        if (base + 4 * q + 3 < lim) __builtin_nontemporal_store(w, (f32x4v*)((float*)out + base + 4 * q));
        else
          for (int u = 0; u < 4; ++u)
            if (base + 4 * q + u < lim) ((float*)out)[base + 4 * q + u] = w[u];
      }
      __syncthreads();
    }
    return;
  }
  // Phase 2: the 64 x (K*462) outputs of this workgroup are one contiguous range of the output
  // array: each thread produces 4 consecutive values (16-byte fp32 / 8-byte bf16 streaming stores,
  // no ragged last pass over a row); (row, column) advance incrementally, without divisions.
  const int rowlen = K * kE;
  const int total = kEmbRows * rowlen;                       // multiple of 4 (64 rows)
  int f = 4 * tid;
  int r = f / rowlen, c = f - r * rowlen;
  for (; f < total; f += 4 * kEmbThreads) {
    float o4[4];
    bool ok[4];
    int rr = r, cc = c;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const ColEnt en = tab[cc];
      const float v = sv[rr * kSvStride + (en.code & 0xffff)];
      float val;
      if (PRECISE) {
        const float a = v * en.f;
        val = (en.code & (1 << 16)) ? v : sincos_pi2(a, (en.code & (1 << 17)) != 0);
      } else {
        const float sn = __builtin_amdgcn_sinf(fmaf(v, en.f, (en.code & (1 << 17)) ? 0.25f : 0.0f));
        val = (en.code & (1 << 16)) ? v : sn;
      }
      o4[u] = val;
      ok[u] = row0 + rr < N;
      if (++cc == rowlen) { cc = 0; ++rr; }
    }
    const int64_t idx = row0 * (int64_t)rowlen + f;
    if (ok[3]) {                                             // rows ascend: last valid => all valid
      if (BF16OUT) {
        bf16x4 w;
#pragma unroll
        for (int u = 0; u < 4; ++u) w[u] = (__bf16)o4[u];
        __builtin_nontemporal_store(w, (bf16x4*)((__bf16*)out + idx));
      } else {
        f32x4v w;
#pragma unroll
        for (int u = 0; u < 4; ++u) w[u] = o4[u];
        __builtin_nontemporal_store(w, (f32x4v*)((float*)out + idx));
      }
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (ok[u]) {
          if (BF16OUT) ((__bf16*)out)[idx + u] = (__bf16)o4[u];
          else ((float*)out)[idx + u] = o4[u];
        }
    }
    c += 4 * kEmbThreads;
    while (c >= rowlen) { c -= rowlen; ++r; }
  }
}

__global__ void warp_kernel(const int32_t* __restrict__ coords, int64_t N, EmbedDev e,
                            float* __restrict__ out) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= N) return;
  const int2 c = ((const int2*)coords)[r];
  const float y = (float)c.x, x = (float)c.y;
  for (int p = 0; p < e.K; ++p)
    for (int i = 0; i < 22; ++i) out[r * (e.K * 22) + p * 22 + i] = warp_value<true>(e, p, i, y, x);
}

}  // namespace npp

using namespace npp;

extern "C" int npp_embed_fwd(const int32_t* d_coords_yx, int64_t N, const npp_embed_cfg* cfg, void* d_out,
                             int out_dtype, int precise, void* stream) {
  int rc = check_embed_cfg(cfg, "npp_embed_fwd");
  if (rc) return rc;
  if (N < 0 || (N > 0 && (!d_coords_yx || !d_out))) { set_error("npp_embed_fwd: bad N/pointers"); return NPP_ERR_ARG; }
  if (out_dtype != 0 && out_dtype != 1) { set_error("npp_embed_fwd: out_dtype %d", out_dtype); return NPP_ERR_ARG; }
  if (N == 0) return NPP_OK;
  const EmbedDev e = make_embed_dev(*cfg);
  const dim3 grid((unsigned)((N + kEmbRows - 1) / kEmbRows)), block(kEmbThreads);
  hipStream_t s = (hipStream_t)stream;
  if (precise) {
    if (out_dtype) hipLaunchKernelGGL((embed_kernel<true, true>), grid, block, 0, s, d_coords_yx, N, e, d_out);
    else hipLaunchKernelGGL((embed_kernel<true, false>), grid, block, 0, s, d_coords_yx, N, e, d_out);
  } else {
    if (out_dtype) hipLaunchKernelGGL((embed_kernel<false, true>), grid, block, 0, s, d_coords_yx, N, e, d_out);
    else hipLaunchKernelGGL((embed_kernel<false, false>), grid, block, 0, s, d_coords_yx, N, e, d_out);
  }
  return check_launch("npp_embed_fwd");
}

extern "C" int npp_warp_fwd(const int32_t* d_coords_yx, int64_t N, const npp_embed_cfg* cfg, float* d_out,
                            void* stream) {
  int rc = check_embed_cfg(cfg, "npp_warp_fwd");
  if (rc) return rc;
  if (N < 0 || (N > 0 && (!d_coords_yx || !d_out))) { set_error("npp_warp_fwd: bad N/pointers"); return NPP_ERR_ARG; }
  if (N == 0) return NPP_OK;
  const EmbedDev e = make_embed_dev(*cfg);
  hipLaunchKernelGGL(warp_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     d_coords_yx, N, e, d_out);
  return check_launch("npp_warp_fwd");
}
