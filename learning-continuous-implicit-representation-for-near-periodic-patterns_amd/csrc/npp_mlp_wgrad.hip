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
#ifndef NPP_WGRAD_NT_SLABS
#define NPP_WGRAD_NT_SLABS 0
#endif
#ifndef NPP_WGRAD_XCD
#define NPP_WGRAD_XCD 1
#endif
// Timing-only diagnostic builds (wrong results; never shipped): NPP_DIAG_WGRAD_SAMETILE (every load hits L2),
// NPP_DIAG_WGRAD_NOLOOP=n (prologue + n tiles + epilogue), NPP_DIAG_WGRAD_NOEPI (no slab stores) -- DESIGN.md section 4.

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
  int64_t dz_bytes, act_bytes;   // sizes of the two stash buffers (range of the buffer descriptors)
  int64_t n_wg;                // 64-row workgroup tiles in the batch (Bp / 64)
  float* gslabs;
  int64_t slab_stride;
  int32_t njobs, wg_chunk;     // workgroup tiles per split
  int32_t ntiles, ksplit;
  WJob jobs[kMaxJobs];
};

// One staged operand-tile pair (dz tile + input tile of one 64-row workgroup tile): 2048 16-byte units each, 4 per thread.
struct WStage { u32x4 a[4], b[4]; };
using rsrc_t = __amdgpu_buffer_rsrc_t;

// Main loop over the 64-row workgroup tiles [g0, g1) of this split.  One barrier per tile; per tile g:
//  * k-step t (of 4) of tile g is multiplied out of LDS buffer g & 1;
//  * piece t of tile g+1 goes registers -> LDS buffer (g+1) & 1 between the k-steps, so the snake(z) conversion and the
//    LDS stores issue in the shadow of the MFMAs instead of in a phase of their own, and the load of piece t of tile g+2
//    is re-issued into the SAME registers right behind it (ONE register set, a whole tile time of flight);
//  * one dword per 128-byte line of tile g+3 is requested (and ignored): the stash streams from HBM with ~2 us latency
//    under load, which this moves into L2 ahead of the real loads without holding registers for it.
// Loads are buffer loads: one descriptor per operand (base = this tile column of the array, range = to the end of the whole
// stash buffer), the tile's position is the SCALAR offset, the lane's 16 bytes the vector offset -- no address arithmetic in
// vector registers and no branch around a load, so every wait the compiler places is a counted one.  The short tiles at an
// array's end (a 224-slot embedding tile, the 128-wide dz_p, the 3-row rgb job) read whatever follows them instead of
// zeros: those operand rows only reach accumulator rows / columns the epilogue drops (mrow >= m, n_idx >= n).
template <bool ZB, bool BIAS>
__device__ __forceinline__ void wgrad_loop(f32x16 (&acc)[2][4], float (&bsum)[2], char* smem, const rsrc_t ra, const rsrc_t rb,
                                           uint32_t a_stride, uint32_t b_stride, int g0, int g1, int tid, bool pf_a,
                                           const int (&offA)[2], const int (&offB)[4]) {
  const int voff = tid * 16;
  WStage st;
  auto gload_piece = [&](int g, int i) {
    st.a[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, voff, (int)((uint32_t)g * a_stride) + 8192 * i, 0));
    st.b[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, voff, (int)((uint32_t)g * b_stride) + 8192 * i, 0));
  };
  // L2 prefetch of a tile pair: 2 x 256 lines of 128 B, one line per thread (waves 0..3 -> dz tile, 4..7 -> input tile)
  auto prefetch = [&](int g) -> uint32_t {
#ifdef NPP_DIAG_WGRAD_SAMETILE
    g = g0;
#endif
    const int line = (tid & 255) * 128;
    if (pf_a) return __builtin_amdgcn_raw_buffer_load_b32(ra, line, (int)((uint32_t)g * a_stride), 0);     // wave-uniform branch
    return __builtin_amdgcn_raw_buffer_load_b32(rb, line, (int)((uint32_t)g * b_stride), 0);
  };
  auto sstore_piece = [&](int buf, int i) {
    char* sA = smem + buf * 2 * kWTileBytes;
    char* sB = sA + kWTileBytes;
    const int off = voff + 8192 * i;
    *(u32x4*)(sA + off) = st.a[i];
    if (ZB) {                      // layer input = snake(z): once per element per workgroup tile
      const f16x8 z = __builtin_bit_cast(f16x8, st.b[i]);
      bf16x8 a;
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = (__bf16)snake_fast((float)z[j]);
      *(bf16x8*)(sB + off) = a;
    } else {
      *(u32x4*)(sB + off) = st.b[i];
    }
  };
  // multiply tile `buf`; between its k-steps hand piece t of the staged tile (g + 1) to the other buffer and re-issue its
  // load for tile g + 2
  auto compute = [&](int buf, int g, bool store, bool load) {
    const char* sA = smem + buf * 2 * kWTileBytes;
    const char* sB = sA + kWTileBytes;
#ifdef NPP_DIAG_WGRAD_SAMETILE
    g = g0 - 2;
#endif
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      // wfrag_offset(tt, t, lane) - wfrag_offset(tt, 0, lane) = (t>>1)*2048 + (t&1)*1024
      const int dt = (t >> 1) * 2048 + (t & 1) * 1024;
      bf16x8 a[2], b[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = wfrag_read(sA, offA[i] + dt);
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = wfrag_read(sB, offB[j] + dt);
      if (store) sstore_piece(buf ^ 1, t);
      if (load) gload_piece(g + 2, t);
      if (BIAS) {
        const bf16x2 ones = {(__bf16)1.0f, (__bf16)1.0f};
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 8; j += 2) {
            const bf16x2 pr = {a[i][j], a[i][j + 1]};
            bsum[i] = __builtin_amdgcn_fdot2_f32_bf16(pr, ones, bsum[i], false);
          }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma_bf16(a[i], b[j], acc[i][j]);
    }
  };

  const int n = g1 - g0;
  if (n <= 0) return;
  auto clampg = [&](int g) { return g < g1 ? g : g0; };     // past the split's end: re-load a valid tile nobody uses
  // prologue: tile g0 -> LDS buffer 0, tile g0+1 -> the register set, tiles g0+1, g0+2 requested into L2
#pragma unroll
  for (int i = 0; i < 4; ++i) gload_piece(g0, i);
  uint32_t pf = prefetch(clampg(g0 + 1));
#pragma unroll
  for (int i = 0; i < 4; ++i) sstore_piece(0, i);
#pragma unroll
  for (int i = 0; i < 4; ++i) gload_piece(clampg(g0 + 1), i);
  asm volatile("" :: "v"(pf));
  pf = prefetch(clampg(g0 + 2));
  wg_barrier();
  // steady state, two tiles per trip (static LDS buffer indices): on entry LDS[0] holds tile g, the registers tile g+1
  for (int g = g0; g < g1; g += 2) {
    compute(0, clampg(g + 2) - 2, g + 1 < g1, true);
    asm volatile("" :: "v"(pf));           // the prefetch issued a tile ago has completed (in-order returns behind the loads)
    pf = prefetch(clampg(g + 3));
    wg_barrier();
    if (g + 1 >= g1) break;
    compute(1, clampg(g + 3) - 2, g + 2 < g1, true);
    asm volatile("" :: "v"(pf));
    pf = prefetch(clampg(g + 4));
    wg_barrier();
  }
  asm volatile("" :: "v"(pf));
}

__global__ __launch_bounds__(kWThreads, 2) void wgrad_kernel(WArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int m_l = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;     // wave tile: rows [64 wm, +64), cols [128 wn, +128)

  // Work item (tile, split) of this workgroup.  Workgroups are dealt round-robin over the 8 XCDs (observed, speed only:
  // MI355X_MICROARCH.md "Workgroup dispatch"), so linear id i runs on XCD group i % 8; the items are numbered so that
  // every group owns a CONTIGUOUS range of them = all tiles of one batch split (+ part of the next): the tiles of a split
  // that read the same dz / input array (3 tiles share dz_5, 5 share dz_S, emb_0 feeds L0 and L5, f1 feeds S and P) then
  // stream it through ONE L2 at about the same time instead of each fetching it from HBM.
#if NPP_WGRAD_XCD
  const int n_items = (int)gridDim.x, xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
  const int q_ = n_items >> 3, r_ = n_items & 7;
  const int item = (xcd < r_ ? xcd * (q_ + 1) : r_ * (q_ + 1) + (xcd - r_) * q_) + slot;
#else
  const int item = (int)blockIdx.x;
#endif
  const int tile_id = item % A.ntiles, split_id = item / A.ntiles;
  // locate the job of this tile: compile-time indices into the kernel-argument table so
  // it is read with scalar loads (a run-time index would force a scratch copy of it)
  WJob J = A.jobs[0];
#pragma unroll
  for (int j = 1; j < kMaxJobs; ++j)
    if (j < A.njobs && tile_id >= A.jobs[j].tile0) J = A.jobs[j];
  const int t_local = tile_id - J.tile0;
  const int tm = t_local / J.tiles_n, tn = t_local - tm * J.tiles_n;
  // operand tiles: 8 k-step pairs (256 features); fewer are valid at the array's end
  const int64_t n_wg = A.n_wg;
  const int64_t wg_begin = (int64_t)split_id * A.wg_chunk;
  const int64_t wg_end = min(n_wg, wg_begin + (int64_t)A.wg_chunk);
  // byte address of (workgroup tile g, pair p) inside an array: ((g * nks/2 + p) * 2) * 2048
  const int64_t a_col = wfmt_array_base(J.a_ks0, n_wg) + (int64_t)tm * kWPairs * 4096;
  const int64_t b_col = wfmt_array_base(J.b_ks0, n_wg) + (int64_t)tn * kWPairs * 4096;
  const int64_t a_left = A.dz_bytes - a_col, b_left = A.act_bytes - b_col;
  const rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(A.dzF + a_col), 0, (int)(a_left > 0x7fffffffLL ? 0x7fffffffLL : a_left), 0x00020000);
  const rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(A.actF + b_col), 0, (int)(b_left > 0x7fffffffLL ? 0x7fffffffLL : b_left), 0x00020000);
  const uint32_t a_stride = (uint32_t)J.a_nks * 2048u, b_stride = (uint32_t)J.b_nks * 2048u;
  const bool pf_a = wave < 4;

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  float bsum[2] = {0.0f, 0.0f};
  const bool do_bias = J.bias_on && tn == 0 && wn == 0;          // wave-uniform
  const bool b_is_z = J.b_is_z != 0;                             // workgroup-uniform
  // per-lane fragment offsets: feature tile (wm|wn)*2 + i, k-step t (4 per workgroup tile)
  int offA[2], offB[4];
#pragma unroll
  for (int i = 0; i < 2; ++i) offA[i] = wfrag_offset(wm * 2 + i, 0, lane);
#pragma unroll
  for (int j = 0; j < 4; ++j) offB[j] = wfrag_offset(wn * 4 + j, 0, lane);

  // The main loop exists in four straight-line forms (input array holds z or ready activations; this wave sums db or not),
  // selected ONCE by uniform branches: no per-unit branches, so every wait the compiler places is a counted one.
#ifdef NPP_DIAG_WGRAD_NOLOOP
  const int g0 = (int)wg_begin, g1 = g0 + NPP_DIAG_WGRAD_NOLOOP;    // diagnostic: prologue + N tiles + epilogue only
#else
  const int g0 = (int)wg_begin, g1 = (int)wg_end;
#endif
  if (b_is_z) {
    if (do_bias) wgrad_loop<true, true>(acc, bsum, smem, ra, rb, a_stride, b_stride, g0, g1, tid, pf_a, offA, offB);
    else wgrad_loop<true, false>(acc, bsum, smem, ra, rb, a_stride, b_stride, g0, g1, tid, pf_a, offA, offB);
  } else {
    if (do_bias) wgrad_loop<false, true>(acc, bsum, smem, ra, rb, a_stride, b_stride, g0, g1, tid, pf_a, offA, offB);
    else wgrad_loop<false, false>(acc, bsum, smem, ra, rb, a_stride, b_stride, g0, g1, tid, pf_a, offA, offB);
  }

  // ---- epilogue: stores into this split's slab, reference layout
  float* slab = A.gslabs + (int64_t)split_id * A.slab_stride;
#ifndef NPP_DIAG_WGRAD_NOEPI
  if (J.colmode == 0 && ((J.ld | J.col0) & 1) == 0) {        // (every layer of this network has an even input width)
    // Plain-column jobs (17 of 21 tiles at K = 3): the accumulator holds one column per lane, i.e. 4-byte stores, 128 per wave
    // -- the tail was bound by store INSTRUCTIONS, not bytes (cdna_hip_programming.md T21).  Each wave transposes its 32 x 128
    // strips through its 16 KiB of the (now idle) operand buffers and stores float4s (two float2s where the reference rows
    // are only 8-byte aligned: ld or col0 not a multiple of 4): 16-32 store instructions per strip instead of 64.
    float* stg = (float*)(smem + wave * 16384);
    const bool a16 = ((J.ld | J.col0) & 3) == 0;            // (w_off and the slab stride are multiples of 4 floats)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) stg[acc_row(r, h) * 128 + 32 * j + m_l] = acc[i][j][r];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // same wave: the writes have landed before other lanes' elements are read
#pragma unroll
      for (int p = 0; p < 16; ++p) {
        const int row = 2 * p + h, c4 = 4 * m_l;
        const float4 v = *(const float4*)(stg + row * 128 + c4);
        const int mrow = tm * kWT + wm * 64 + i * 32 + row;
        const int n_idx = tn * kWT + wn * 128 + c4;
        if (mrow < J.m && n_idx < J.n) {                    // J.n is a multiple of 4 for these jobs
          float* dst = slab + J.w_off + (int64_t)mrow * J.ld + J.col0 + n_idx;
          typedef float f4v __attribute__((ext_vector_type(4)));
          typedef float f2v __attribute__((ext_vector_type(2)));
#if NPP_WGRAD_NT_SLABS
          if (a16) {
            __builtin_nontemporal_store((f4v){v.x, v.y, v.z, v.w}, (f4v*)dst);
          } else {
            __builtin_nontemporal_store((f2v){v.x, v.y}, (f2v*)dst);
            __builtin_nontemporal_store((f2v){v.z, v.w}, (f2v*)(dst + 2));
          }
#else
          if (a16) {
            *(f4v*)dst = (f4v){v.x, v.y, v.z, v.w};
          } else {
            *(f2v*)dst = (f2v){v.x, v.y};
            *(f2v*)(dst + 2) = (f2v){v.z, v.w};
          }
#endif
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // reads done before the next strip overwrites the staging area
    }
  } else
#endif
  {
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
#ifdef NPP_DIAG_WGRAD_NOEPI
        if (col >= 0 && mrow < J.m && acc[i][j][r] == 123.456f) slab[J.w_off + (int64_t)mrow * J.ld + col] = acc[i][j][r];
#else
        if (col >= 0 && mrow < J.m) slab[J.w_off + (int64_t)mrow * J.ld + col] = acc[i][j][r];
#endif
      }
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
  if (A.n_wg > 65536) { set_error("npp_mlp_wgrad: Bp=%lld too large (32-bit tile offsets: <= %d rows per call)", (long long)Bp, 65536 * kRowTile); return NPP_ERR_ARG; }
  A.dz_bytes = wfmt_array_base(kDzTotalKs, A.n_wg);
  A.act_bytes = wfmt_array_base(act_total_ks(K), A.n_wg);
  A.gslabs = d_gslabs;
  A.slab_stride = slab_stride_of(make_desc(K).total_params);
  const int ntiles = build_jobs(K, A);
  A.wg_chunk = (int)((A.n_wg + ksplit - 1) / ksplit);
  static SmemOnce once;
  if (!smem_attr(once, (const void*)wgrad_kernel, kSmemW)) { set_error("npp_mlp_wgrad: smem attribute"); return NPP_ERR_LAUNCH; }
  A.ntiles = ntiles; A.ksplit = ksplit;
  hipLaunchKernelGGL(wgrad_kernel, dim3((unsigned)(ntiles * ksplit)), dim3(kWThreads), kSmemW, (hipStream_t)stream, A);
  return check_launch("npp_mlp_wgrad");
}
