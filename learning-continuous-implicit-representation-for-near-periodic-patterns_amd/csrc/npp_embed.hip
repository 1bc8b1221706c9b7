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

constexpr int kEmbRows = 64;     // rows per workgroup (one per lane in phase 1)
constexpr int kEmbThreads = 256;
constexpr int kSvStride = NPP_MAX_K * 22 + 1;   // odd stride: conflict-free row-per-lane writes

// Phase 1: the K*22 warped values of 64 rows into LDS: lane = row, the warp index is
// wave-uniform (wave w takes jobs w, w+4, ...) so the constant tables are read through
// scalar loads.  Phase 2: every thread produces column PAIRS (462 is even) so stores are
// 8 B (fp32) / 4 B (bf16) and fully coalesced.
template <bool PRECISE, bool BF16OUT>
__global__ __launch_bounds__(kEmbThreads) void embed_kernel(const int32_t* __restrict__ coords,
                                                            int64_t N, EmbedDev e,
                                                            void* __restrict__ out) {
  __shared__ float sv[kEmbRows * kSvStride];
  __shared__ float sfreq[NPP_N_FREQ];   // lane-varying index: keep it out of the kernarg struct
  const int tid = threadIdx.x;
  if (tid == 0) {
#pragma unroll
    for (int j = 0; j < NPP_N_FREQ; ++j) sfreq[j] = PRECISE ? e.freq[j] : e.freq_rev[j];
  }
  const int64_t row0 = (int64_t)blockIdx.x * kEmbRows;
  const int K = e.K;
  {
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int64_t r = row0 + lane;
    const int2 c = r < N ? ((const int2*)coords)[r] : make_int2(0, 0);
    const float y = (float)c.x, x = (float)c.y;   // (row=y, col=x): first member is y
    for (int job = wave; job < K * 22; job += 4)
      sv[lane * kSvStride + job] = warp_value<PRECISE>(e, job / 22, job % 22, y, x);
  }
  __syncthreads();
  const int pairs = K * (kE / 2);
  for (int r = 0; r < kEmbRows; ++r) {
    const int64_t row = row0 + r;
    if (row >= N) break;
    for (int cp = tid; cp < pairs; cp += kEmbThreads) {
      const int c = cp * 2;
      const int p = c / kE, cc = c - p * kE;
      float o2[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int c1 = cc + u;
        const int blk = c1 / 22, i = c1 - blk * 22;
        const float v = sv[r * kSvStride + p * 22 + i];
        float val = v;
        if (blk > 0) {
          const int fj = (blk - 1) >> 1;
          const bool is_cos = ((blk - 1) & 1) != 0;
          if (PRECISE) {
            const float a = v * sfreq[fj];
            val = is_cos ? cosf(a) : sinf(a);
          } else {
            const float a = v * sfreq[fj];
            val = is_cos ? __builtin_amdgcn_cosf(a) : __builtin_amdgcn_sinf(a);
          }
        }
        o2[u] = val;
      }
      const int64_t idx = row * (int64_t)(K * kE) + c;
      if (BF16OUT) {
        bf16x2 w;
        w[0] = (__bf16)o2[0];
        w[1] = (__bf16)o2[1];
        *(bf16x2*)((__bf16*)out + idx) = w;
      } else {
        *(float2*)((float*)out + idx) = make_float2(o2[0], o2[1]);
      }
    }
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
