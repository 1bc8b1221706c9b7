// npp_mlp_wgrad.hip -- K3b: weight / bias gradients of every layer in ONE grouped,
// split-K bf16 MFMA GEMM launch.
//
// dW_l[n][k] = sum_b dz_l[n][b] * a_{l-1}[k][b]   (autograd of F.linear, networks.py:56-95)
// db_l[n]    = sum_b dz_l[n][b]
// Both operands come feature-major ([feature][row], written by npp_mlp_fwd / npp_mlp_bwd),
// i.e. contiguous along the contraction index b: a plain "NT" GEMM with M = layer outputs,
// N = layer inputs (or the 480 embedding slots of a proposal), K = padded batch.
// A job = one (dz block, input block) pair; jobs are cut into 128x128 output tiles and the
// batch is split over gridDim.y; every (tile, split) writes its partial sums with plain
// stores into slab `split` of the gradient buffer, in the reference's parameter layout
// ([out][in] row-major).  npp_adam_step adds the slabs, so there are no atomics and the
// result is bitwise reproducible.
//
// Algorithmic work: 2 * sum_l n_out*n_in FLOP per batch row (embedding pad slots and the
// 128-row padding of the 3-row rgb job are not counted).
#include "npp_common.h"

namespace npp {

constexpr int kWT = 128;            // output tile (both dims)
constexpr int kWBK = 64;            // batch rows per main-loop step
constexpr int kWThreads = 256;
constexpr int kWTileBytes = kWT * kWBK * 2;          // 16 KiB per operand tile
constexpr int kSmemW = 4 * kWTileBytes;              // A,B double buffered = 64 KiB
constexpr int kMaxJobs = 24;

struct WJob {
  int32_t dz_row0, m;          // rows of dzT, number of valid rows (layer outputs)
  int32_t src_row0, n;         // rows of actT, number of valid rows (inputs / emb slots)
  int32_t colmode, col0;       // 0: col = col0 + idx ; 1: col = col0 + emb_col(slot idx), pad slots skipped
  int32_t ld, bias_on;         // reference row stride (n_in) ; 1 = this job also produces db
  int64_t w_off, b_off;        // float offsets in the parameter blob
  int32_t tile0, tiles_n;      // first global tile index, tiles along n
};

struct WArgs {
  const __bf16* dzT;
  const __bf16* actT;
  int64_t Bp;
  float* gslabs;
  int64_t slab_stride;
  int32_t njobs, kchunk;
  WJob jobs[kMaxJobs];
};

__device__ __forceinline__ uint32_t swz(int row, int ch) { return (uint32_t)(row * 128 + ((ch ^ (row & 7)) << 4)); }

__global__ __launch_bounds__(kWThreads, 2) void wgrad_kernel(WArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int m_l = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;

  // locate the job of this tile: compile-time indices into the kernel-argument table so
  // it is read with scalar loads (a run-time index would force a scratch copy of it)
  WJob J = A.jobs[0];
#pragma unroll
  for (int j = 1; j < kMaxJobs; ++j)
    if (j < A.njobs && (int)blockIdx.x >= A.jobs[j].tile0) J = A.jobs[j];
  const int t_local = blockIdx.x - J.tile0;
  const int tm = t_local / J.tiles_n, tn = t_local - tm * J.tiles_n;
  const int m_valid = min(kWT, J.m - tm * kWT), n_valid = min(kWT, J.n - tn * kWT);
  const int64_t Bp = A.Bp;
  const int64_t k_begin = (int64_t)blockIdx.y * A.kchunk;
  const int64_t k_end = min(Bp, k_begin + (int64_t)A.kchunk);
  const __bf16* gA = A.dzT + (int64_t)(J.dz_row0 + tm * kWT) * Bp;
  const __bf16* gB = A.actT + (int64_t)(J.src_row0 + tn * kWT) * Bp;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  float bsum[2] = {0.0f, 0.0f};
  const bool do_bias = J.bias_on && tn == 0 && wn == 0;

  // staging: 1024 16-byte chunks per operand tile, 4 per thread
  u32x4 ra[4], rb[4];
  auto gload = [&](int64_t k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int id = tid + kWThreads * i, row = id >> 3, ch = id & 7;
      const u32x4 z = {0u, 0u, 0u, 0u};
      ra[i] = row < m_valid ? *(const u32x4*)(gA + (int64_t)row * Bp + k0 + ch * 8) : z;
      rb[i] = row < n_valid ? *(const u32x4*)(gB + (int64_t)row * Bp + k0 + ch * 8) : z;
    }
  };
  auto sstore = [&](int buf) {
    char* sA = smem + buf * 2 * kWTileBytes;
    char* sB = sA + kWTileBytes;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int id = tid + kWThreads * i, row = id >> 3, ch = id & 7;
      *(u32x4*)(sA + swz(row, ch)) = ra[i];
      *(u32x4*)(sB + swz(row, ch)) = rb[i];
    }
  };

  int buf = 0;
  if (k_begin < k_end) {
    gload(k_begin);
    sstore(0);
  }
  __syncthreads();
  for (int64_t k0 = k_begin; k0 < k_end; k0 += kWBK) {
    const bool has_next = k0 + kWBK < k_end;
    if (has_next) gload(k0 + kWBK);
    const char* sA = smem + buf * 2 * kWTileBytes;
    const char* sB = sA + kWTileBytes;
#pragma unroll
    for (int ks = 0; ks < kWBK / 16; ++ks) {
      bf16x8 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[i] = *(const bf16x8*)(sA + swz(wm * 64 + i * 32 + m_l, ks * 2 + h));
        b[i] = *(const bf16x8*)(sB + swz(wn * 64 + i * 32 + m_l, ks * 2 + h));
      }
      if (do_bias) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 8; ++j) bsum[i] += (float)a[i][j];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma_bf16(a[i], b[j], acc[i][j]);
    }
    if (has_next) sstore(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  // ---- epilogue: plain stores into this split's slab, reference layout
  float* slab = A.gslabs + (int64_t)blockIdx.y * A.slab_stride;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n_idx = tn * kWT + wn * 64 + j * 32 + m_l;      // accumulator column = lane & 31
    int col = -1;
    if (n_idx < J.n) {
      if (J.colmode == 0) col = J.col0 + n_idx;
      else {
        const int c = emb_col(n_idx >> 4, (n_idx >> 3) & 1, n_idx & 7);
        col = c < 0 ? -1 : J.col0 + c;
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int mrow = tm * kWT + wm * 64 + i * 32 + acc_row(r, h);
        if (col >= 0 && mrow < J.m) slab[J.w_off + (int64_t)mrow * J.ld + col] = acc[i][j][r];
      }
    }
  }
  if (do_bias) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float v = bsum[i] + __shfl_xor(bsum[i], 32, 64);
      const int mrow = tm * kWT + wm * 64 + i * 32 + m_l;
      if (h == 0 && mrow < J.m) slab[J.b_off + mrow] = v;
    }
  }
}

// Build the job table for NPP_Net (K>1) / NPP_Net_top1 (K==1).
static int build_jobs(int K, WArgs& A) {
  const NetDesc d = make_desc(K);
  int nj = 0, tile = 0;
  auto add = [&](int layer, int dz_row0, int src_row0, int n, int colmode, int col0, int bias_on) {
    WJob& j = A.jobs[nj++];
    j.dz_row0 = dz_row0; j.m = d.n_out[layer];
    j.src_row0 = src_row0; j.n = n;
    j.colmode = colmode; j.col0 = col0;
    j.ld = d.n_in[layer]; j.bias_on = bias_on;
    j.w_off = d.w_off[layer]; j.b_off = d.b_off[layer];
    j.tile0 = tile;
    j.tiles_n = (n + kWT - 1) / kWT;
    tile += ((j.m + kWT - 1) / kWT) * j.tiles_n;
  };
  const int emb0 = kActEmbRow0;
  add(L0, 0 * kW, emb0, kEmbSlots, 1, 0, 1);
  for (int l = L1; l <= L4; ++l) add(l, l * kW, (l - 1) * kW, kW, 0, 0, 1);
  add(L5, 5 * kW, emb0, kEmbSlots, 1, 0, 1);
  add(L5, 5 * kW, 4 * kW, kW, 0, kE, 0);
  add(L6, 6 * kW, 5 * kW, kW, 0, 0, 1);
  add(L7, 7 * kW, 6 * kW, kW, 0, 0, 1);
  add(LF1, kDzF1 * kW, 7 * kW, kW, 0, 0, 1);
  if (K > 1) {
    add(LS, kDzS * kW, kActF1 * kW, kW, 0, 0, 1);
    for (int p = 1; p < K; ++p) add(LS, kDzS * kW, emb0 + p * kEmbSlots, kEmbSlots, 1, kW + (p - 1) * kE, 0);
    add(LF2, kDzF2 * kW, kActAS * kW, kW, 0, 0, 1);
    add(LP, kDzP * kW, kActF1 * kW, kW, 0, 0, 1);
    add(LP, kDzP * kW, kActF2 * kW, kW, 0, kW, 0);
  } else {
    add(LP, kDzP * kW, kActF1 * kW, kW, 0, 0, 1);
  }
  add(LRGB, kDzRgbRow0, kActAP * kW, kW / 2, 0, 0, 1);
  A.njobs = nj;
  return tile;
}

}  // namespace npp

using namespace npp;

extern "C" int npp_mlp_wgrad(const void* d_dzT, const void* d_actT, int64_t Bp, int K, int width, int ksplit,
                             float* d_gslabs, void* stream) {
  if (K < 1 || K > NPP_MAX_K) { set_error("npp_mlp_wgrad: K=%d", K); return NPP_ERR_ARG; }
  if (width != NPP_WIDTH) { set_error("npp_mlp_wgrad: width %d unsupported (build is %d)", width, NPP_WIDTH); return NPP_ERR_UNSUPPORTED; }
  if (Bp <= 0 || Bp % kRowTile || ksplit < 1 || ksplit > 64) { set_error("npp_mlp_wgrad: bad Bp=%lld / ksplit=%d", (long long)Bp, ksplit); return NPP_ERR_ARG; }
  if (!d_dzT || !d_actT || !d_gslabs) { set_error("npp_mlp_wgrad: null pointer"); return NPP_ERR_ARG; }
  WArgs A{};
  A.dzT = (const __bf16*)d_dzT;
  A.actT = (const __bf16*)d_actT;
  A.Bp = Bp;
  A.gslabs = d_gslabs;
  A.slab_stride = make_desc(K).total_params;
  const int ntiles = build_jobs(K, A);
  const int64_t steps = Bp / kWBK;
  A.kchunk = (int)(((steps + ksplit - 1) / ksplit) * kWBK);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t ea = hipFuncSetAttribute((const void*)wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kSmemW);
    if (ea != hipSuccess) { set_error("npp_mlp_wgrad: smem attr: %s", hipGetErrorString(ea)); return NPP_ERR_LAUNCH; }
    attr_set = true;
  }
  hipLaunchKernelGGL(wgrad_kernel, dim3((unsigned)ntiles, (unsigned)ksplit), dim3(kWThreads), kSmemW,
                     (hipStream_t)stream, A);
  return check_launch("npp_mlp_wgrad");
}
