// npp_mlp_wgrad.hip -- K3b: weight / bias gradients of every layer in ONE grouped,
// split-K bf16 MFMA GEMM launch.
//
// dW_l[n][k] = sum_b dz_l[n][b] * a_{l-1}[k][b]   (autograd of F.linear, networks.py:56-95)
// db_l[n]    = sum_b dz_l[n][b]
// The contraction runs over the batch.  Both operands arrive as the bf16 fragments the
// fused forward / backward kernels hold in registers (one 16-byte unit = 8 features of one
// row; "W-format" arrays, npp_layout.h): a 128-feature operand tile of one 64-row workgroup
// tile is a contiguous 16-KiB chunk that is copied linearly into LDS, and the MFMA operand
// fragments (one feature, 8 consecutive rows per lane) are produced by the hardware
// transposing read ds_read_b64_tr_b16 -- conflict-free by construction of the line layout.
// A job = one (dz array, input array) pair; jobs are cut into 256x256 output tiles (the kernel is
// HBM-bound on re-reading the stash arrays: with 128x128 tiles every array was read twice, 1.0 GB
// per pass at c2; 256x256 tiles, 8 waves of 64x128, read each array once per job) and the
// batch is split over gridDim.y; every (tile, split) writes its partial sums with plain
// stores into slab `split` of the gradient buffer, in the reference's parameter layout
// ([out][in] row-major).  npp_adam_step adds the slabs: no atomics, bit-reproducible.
//
// Algorithmic work: 2 * sum_l n_out*n_in FLOP per batch row (embedding pad slots and the
// padding of the 3-row rgb job are not counted).
#include "npp_common.h"

namespace npp {

constexpr int kWT = 256;            // output tile (both dims)
constexpr int kWBK = 64;            // batch rows per main-loop step
constexpr int kWThreads = 512;      // 8 waves: 4 (m) x 2 (n), 64 x 128 outputs each
constexpr int kWPairs = kWT / 32;   // k-step pairs (32 features) per operand tile
constexpr int kWTileBytes = kWT * kWBK * 2;          // 32 KiB per operand tile
constexpr int kSmemW = 4 * kWTileBytes;              // A,B double buffered = 128 KiB
constexpr int kMaxJobs = 24;

struct WJob {
  int32_t a_ks0, a_nks, m;     // dz array: k-step offset inside dzF, k-steps, valid outputs
  int32_t b_ks0, b_nks, n;     // input array inside actF, k-steps, valid inputs / emb slots
  int32_t colmode, col0;       // 0: col = col0 + idx ; 1: col = col0 + emb_col(slot idx), pad slots skipped
  int32_t ld, bias_on;         // reference row stride (n_in) ; 1 = this job also produces db
  int32_t b_is_z, pad_;        // input array holds fp16 pre-activations: layer input = snake(z)
  int64_t w_off, b_off;        // float offsets in the parameter blob
  int32_t tile0, tiles_n;      // first global tile index, tiles along n
};

struct WArgs {
  const char* dzF;
  const char* actF;
  int64_t n_wg;                // 64-row workgroup tiles in the batch (Bp / 64)
  float* gslabs;
  int64_t slab_stride;
  int32_t njobs, wg_chunk;     // workgroup tiles per split
  WJob jobs[kMaxJobs];
};

__global__ __launch_bounds__(kWThreads, 2) void wgrad_kernel(WArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int m_l = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;     // wave tile: rows [64 wm, +64), cols [128 wn, +128)

  // locate the job of this tile: compile-time indices into the kernel-argument table so
  // it is read with scalar loads (a run-time index would force a scratch copy of it)
  WJob J = A.jobs[0];
#pragma unroll
  for (int j = 1; j < kMaxJobs; ++j)
    if (j < A.njobs && (int)blockIdx.x >= A.jobs[j].tile0) J = A.jobs[j];
  const int t_local = blockIdx.x - J.tile0;
  const int tm = t_local / J.tiles_n, tn = t_local - tm * J.tiles_n;
  // operand tiles: 8 k-step pairs (256 features); fewer are valid at the array's end
  const int a_pairs = min(kWPairs, (J.a_nks >> 1) - tm * kWPairs), b_pairs = min(kWPairs, (J.b_nks >> 1) - tn * kWPairs);
  const int a_bytes = a_pairs * 4096, b_bytes = b_pairs * 4096;
  const int64_t n_wg = A.n_wg;
  const int64_t wg_begin = (int64_t)blockIdx.y * A.wg_chunk;
  const int64_t wg_end = min(n_wg, wg_begin + (int64_t)A.wg_chunk);
  // byte address of (workgroup tile g, pair p) inside an array: ((g * nks/2 + p) * 2) * 2048
  const char* gA = A.dzF + wfmt_array_base(J.a_ks0, n_wg) + (int64_t)tm * kWPairs * 4096;
  const char* gB = A.actF + wfmt_array_base(J.b_ks0, n_wg) + (int64_t)tn * kWPairs * 4096;
  const int64_t a_stride = (int64_t)J.a_nks * 2048, b_stride = (int64_t)J.b_nks * 2048;

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  float bsum[2] = {0.0f, 0.0f};
  const bool do_bias = J.bias_on && tn == 0 && wn == 0;

  // staging: 2048 16-byte units per operand tile, 4 per thread, linear copy.  The stash arrays
  // stream from HBM (~2 us latency): global loads run TWO workgroup tiles ahead of the MFMAs
  // (two register sets, loop unrolled by two so they are named statically), the LDS image one.
  struct Stage { u32x4 a[4], b[4]; };
  auto gload = [&](Stage& st, int64_t g) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int off = (tid + kWThreads * i) * 16;
      const u32x4 z = {0u, 0u, 0u, 0u};
      st.a[i] = off < a_bytes ? *(const u32x4*)(gA + g * a_stride + off) : z;
      st.b[i] = off < b_bytes ? *(const u32x4*)(gB + g * b_stride + off) : z;
    }
  };
  const bool b_is_z = J.b_is_z != 0;
  auto sstore = [&](const Stage& st, int buf) {
    char* sA = smem + buf * 2 * kWTileBytes;
    char* sB = sA + kWTileBytes;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int off = (tid + kWThreads * i) * 16;
      *(u32x4*)(sA + off) = st.a[i];
      if (b_is_z) {                  // layer input = snake(z): once per element per workgroup tile
        const f16x8 z = __builtin_bit_cast(f16x8, st.b[i]);
        bf16x8 a;
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = (__bf16)snake_fast((float)z[j]);
        *(bf16x8*)(sB + off) = a;
      } else {
        *(u32x4*)(sB + off) = st.b[i];
      }
    }
  };
  // per-lane fragment offsets: feature tile (wm|wn)*2 + i, k-step t (4 per workgroup tile)
  int offA[2], offB[4];
#pragma unroll
  for (int i = 0; i < 2; ++i) offA[i] = wfrag_offset(wm * 2 + i, 0, lane);
#pragma unroll
  for (int j = 0; j < 4; ++j) offB[j] = wfrag_offset(wn * 4 + j, 0, lane);
  auto compute = [&](int buf) {
    const char* sA = smem + buf * 2 * kWTileBytes;
    const char* sB = sA + kWTileBytes;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      // wfrag_offset(tt, t, lane) - wfrag_offset(tt, 0, lane) = (t>>1)*2048 + (t&1)*1024
      const int dt = (t >> 1) * 2048 + (t & 1) * 1024;
      bf16x8 a[2], b[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = wfrag_read(sA, offA[i] + dt);
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = wfrag_read(sB, offB[j] + dt);
      if (do_bias) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 8; ++j) bsum[i] += (float)a[i][j];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma_bf16(a[i], b[j], acc[i][j]);
    }
  };

  // prologue: tile g0 -> LDS buffer 0, tile g0+1 -> register set s1
  Stage s0, s1;
  if (wg_begin < wg_end) {
    gload(s0, wg_begin);
    if (wg_begin + 1 < wg_end) gload(s1, wg_begin + 1);
    sstore(s0, 0);
  }
  wg_barrier();
  // steady state, two tiles per trip: on entry LDS[0] holds tile g, s1 holds tile g+1
  for (int64_t g = wg_begin; g < wg_end; g += 2) {
    if (g + 2 < wg_end) gload(s0, g + 2);
    compute(0);
    if (g + 1 < wg_end) sstore(s1, 1);
    wg_barrier();
    if (g + 1 >= wg_end) break;
    if (g + 3 < wg_end) gload(s1, g + 3);
    compute(1);
    if (g + 2 < wg_end) sstore(s0, 0);
    wg_barrier();
  }

  // ---- epilogue: plain stores into this split's slab, reference layout
  float* slab = A.gslabs + (int64_t)blockIdx.y * A.slab_stride;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n_idx = tn * kWT + wn * 128 + j * 32 + m_l;     // accumulator column = lane & 31
    int col = -1;
    if (n_idx < J.n) {
      if (J.colmode == 0) col = J.col0 + n_idx;
      else {
        // column c of k-step ks is element unperm_j(c) of lane-half unperm_hh(c) (perm16 order)
        const int c16 = n_idx & 15;
        const int c = emb_col(n_idx >> 4, unperm_hh(c16), unperm_j(c16));
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
  auto add = [&](int layer, int a_ks0, int a_nks, int b_ks0, int b_nks, int n, int colmode, int col0, int bias_on,
                 int b_is_z) {
    WJob& j = A.jobs[nj++];
    j.b_is_z = b_is_z;
    j.a_ks0 = a_ks0; j.a_nks = a_nks; j.m = d.n_out[layer];
    j.b_ks0 = b_ks0; j.b_nks = b_nks; j.n = n;
    j.colmode = colmode; j.col0 = col0;
    j.ld = d.n_in[layer]; j.bias_on = bias_on;
    j.w_off = d.w_off[layer]; j.b_off = d.b_off[layer];
    j.tile0 = tile;
    j.tiles_n = (n + kWT - 1) / kWT;
    tile += ((j.m + kWT - 1) / kWT) * j.tiles_n;
  };
  const int A16 = kKSAct;
  // source arrays 0..7 (L0..L7 outputs) and a_s, a_p hold z (fp16); f1 / f2 are linear (bf16)
  auto act = [&](int layer, int dz_idx, int src_idx, int col0, int bias_on) {
    add(layer, dz_idx * A16, A16, src_idx * A16, A16, kW, 0, col0, bias_on, src_idx != kActF1 && src_idx != kActF2);
  };
  auto emb = [&](int layer, int dz_idx, int p, int col0, int bias_on) {
    add(layer, dz_idx * A16, A16, kActKsEmb0 + p * kKSEmb, kKSEmb, kEmbSlots, 1, col0, bias_on, 0);
  };
  emb(L0, 0, 0, 0, 1);
  for (int l = L1; l <= L4; ++l) act(l, l, l - 1, 0, 1);
  emb(L5, 5, 0, 0, 1);
  act(L5, 5, 4, kE, 0);
  act(L6, 6, 5, 0, 1);
  act(L7, 7, 6, 0, 1);
  act(LF1, kDzF1, 7, 0, 1);
  if (K > 1) {
    act(LS, kDzS, kActF1, 0, 1);
    for (int p = 1; p < K; ++p) emb(LS, kDzS, p, kW + (p - 1) * kE, 0);
    act(LF2, kDzF2, kActAS, 0, 1);
    add(LP, kDzKsP, A16 / 2, kActF1 * A16, A16, kW, 0, 0, 1, 0);
    add(LP, kDzKsP, A16 / 2, kActF2 * A16, A16, kW, 0, kW, 0, 0);
  } else {
    add(LP, kDzKsP, A16 / 2, kActF1 * A16, A16, kW, 0, 0, 1, 0);
  }
  add(LRGB, kDzKsRgb, 2, kActKsAP, A16 / 2, kW / 2, 0, 0, 1, 1);
  A.njobs = nj;
  return tile;
}

}  // namespace npp

using namespace npp;

extern "C" int npp_mlp_wgrad_tiles(int K) {
  if (K < 1 || K > NPP_MAX_K) { set_error("npp_mlp_wgrad_tiles: K=%d", K); return NPP_ERR_ARG; }
  WArgs A{};
  return build_jobs(K, A);
}

extern "C" int npp_mlp_wgrad(const void* d_dzT, const void* d_actT, int64_t Bp, int K, int width, int ksplit,
                             float* d_gslabs, void* stream) {
  if (K < 1 || K > NPP_MAX_K) { set_error("npp_mlp_wgrad: K=%d", K); return NPP_ERR_ARG; }
  if (width != NPP_WIDTH) { set_error("npp_mlp_wgrad: width %d unsupported (build is %d)", width, NPP_WIDTH); return NPP_ERR_UNSUPPORTED; }
  if (Bp <= 0 || Bp % kRowTile || ksplit < 1 || ksplit > 64) { set_error("npp_mlp_wgrad: bad Bp=%lld / ksplit=%d", (long long)Bp, ksplit); return NPP_ERR_ARG; }
  if (!d_dzT || !d_actT || !d_gslabs) { set_error("npp_mlp_wgrad: null pointer"); return NPP_ERR_ARG; }
  WArgs A{};
  A.dzF = (const char*)d_dzT;
  A.actF = (const char*)d_actT;
  A.n_wg = Bp / kRowTile;
  A.gslabs = d_gslabs;
  A.slab_stride = make_desc(K).total_params;
  const int ntiles = build_jobs(K, A);
  A.wg_chunk = (int)((A.n_wg + ksplit - 1) / ksplit);
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
