// npp_cx.hip -- K5 integer patch gather and K6 contextual-loss core (forward + d/dx).
//
// K5 replaces extract_glimpse(mode='nearest', padding 'zeros') as models/sampler.py:171-178,
// 284-291 calls it (utils/extract_glimpse.py:53-79): an integer crop [c - P/2, c + P/2) with
// zeros outside the image; the reference first tiles the whole image once per crop.
//
// K6 replaces contextual_loss(x, y, band_width, weight, 'cosine')
// (externel_lib/contextual_loss/functional.py:9-63, :127-163) and its autograd backward w.r.t.
// x, starting from the feature tensors (N, C, h, w) fp32 (NCHW, as the VGG trunk returns them):
//   mu_c = mean_{n,h,w} y ; xh = normalize_C(x - mu) ; yh = normalize_C(y - mu)
//   raw = xh^T yh (N, I, J) ; D = 1 - clamp(raw, 0, 1) ; Dt = D / (min_j D + 1e-5)
//   w = exp((1 - Dt)/h) ; cx = w / sum_j w ; cxn = mean_j max_i cx ; loss = mean_n -log(cxn + 1e-5)
// The similarity needs fp32 (D/(min D + 1e-5) amplifies operand rounding by up to 1e5), so the two
// matrix products run on v_mfma_f32_32x32x2_f32 (exact fp32, 157 TFLOP/s peak).  The (I x J) matrices
// are written once (D) and once more (cx, overwritten in place by d raw in the backward); the
// reference materialises 4-5 of them plus autograd copies.
//
// Backward (closed form, derivation in DESIGN.md section 7 / oracle cx_backward): the loss sees cx
// only at the per-column arg-max rows i*(j):  G_ij = g_n / J [i == i*(j)],  A_i = sum_j G_ij cx_ij,
//   dw = (G - A_i)/s_i ; dDt = -dw w / h ; dD = dDt/(m_i + 1e-5) - [j == argmin_j D_ij] sum_j dDt D/(m_i+1e-5)^2
//   draw = -dD on 0 < raw < 1 ; dxh = draw yh^T ; dx = (dxh - xh (xh . dxh)) / |x - mu|.
#include <stdlib.h>

#include "npp_common.h"
#include <type_traits>
#include "npp_trunk_layout.h"

namespace npp {

// ---------------------------------------------------------------- K5 patch gather
__global__ void patch_gather_kernel(const float* __restrict__ img, const float* __restrict__ mask, int H, int W,
                                    const int32_t* __restrict__ centres, int M, int P, float* __restrict__ out_rgb,
                                    float* __restrict__ out_mask) {
  const int m = blockIdx.y;
  const int cy = centres[2 * m], cx = centres[2 * m + 1];
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < P * P; idx += gridDim.x * blockDim.x) {
    const int py = idx / P, px = idx - py * P;
    const int y = cy - P / 2 + py, x = cx - P / 2 + px;
    const bool in = y >= 0 && y < H && x >= 0 && x < W;
    const int64_t src = (int64_t)y * W + x;
#pragma unroll
    for (int c = 0; c < 3; ++c) out_rgb[((int64_t)m * 3 + c) * P * P + idx] = in ? img[src * 3 + c] : 0.0f;
    if (out_mask) out_mask[(int64_t)m * P * P + idx] = in ? mask[src] : 0.0f;
  }
}

// ---------------------------------------------------------------- K6 contextual loss
struct CxWs {           // workspace carve (floats unless noted)
  float* mu;            // [groups][mu_stride]: mean over the GROUP's y samples (one group = one image of a stacked launch)
  float* ssx;           // [N*hw]  sum_c (x - mu)^2   (inverse norm = 1 / max(sqrt(.), 1e-12))
  float* ssy;           // [N*hw]
  unsigned* dmin;       // [N*hw]  row min of D as float bits (D >= 0: unsigned order == float order)
  float* s;             // [N*hw]  row sum of w
  unsigned* cmax;       // [N*hw]  column max of cx as float bits
  float* dot;           // [dot_slots][N*hw]  xh . dxh per position, one partial per channel tile (summed in fixed order)
  float* g;             // [N]     dL/dcxn / J
  float* lsum;          // [N]     the samples' loss terms (summed per group by the last block of cx_loss_kernel)
  unsigned* ticket;     // arrival counter of that reduction (cleared by cx_mean_kernel)
  float* inx;           // [N*hw]  1 / max(sqrt(ssx), 1e-12)  (written by cx_loss_kernel, read by the backward kernels)
  float* iny;           // [N*hw]
  float* D;             // [N*hw*hw]
  float* cx;            // [N*hw*hw]  cx, then d raw
  // sample groups (npp_cx_fwd_bwd_groups): group m = the nk samples from batch index iter[m].x0 on; iter == null: one group of N
  const StackIter* iter;
  int32_t M, mu_stride, N, dot_slots;
  int32_t min_in_rows;  // the row pass (cx_rows_fwd32_kernel) takes the row minima of D itself: cx_sim_kernel issues no atomicMin
};
// group of sample n and the group's sample count
__device__ __forceinline__ int cx_group(const CxWs& w, int n, int& ng) {
  if (!w.iter) { ng = w.N; return 0; }
  int m = 0;
  for (int j = 1; j < w.M; ++j)
    if (w.iter[j].nk > 0 && n >= w.iter[j].x0) m = j;
  ng = w.iter[m].nk;
  return m;
}
__device__ __forceinline__ const float* cx_mu(const CxWs& w, int n) {
  int ng;
  return w.mu + (int64_t)cx_group(w, n, ng) * w.mu_stride;
}

__host__ __device__ inline int64_t cx_ws_floats(int N, int C, int hw) {
  return (int64_t)NPP_MAX_STACK * (C + 16) + (7LL + C / 32) * N * hw + 3LL * N + 2LL * N * hw * hw + 160;
}

__host__ inline CxWs carve(float* base, int N, int C, int hw, const void* iter = nullptr, int M = 0) {
  CxWs w;
  float* p = base;
  w.iter = (const StackIter*)iter; w.M = M; w.N = N; w.mu_stride = (C + 15) / 16 * 16; w.dot_slots = C / 32; w.min_in_rows = 0;
  w.mu = p; p += (int64_t)(M > 0 ? M : 1) * w.mu_stride;
  const int64_t nh = (int64_t)N * hw;
  w.ssx = p; p += nh;
  w.ssy = p; p += nh;
  w.dmin = (unsigned*)p; p += nh;
  w.s = p; p += nh;
  w.cmax = (unsigned*)p; p += nh;
  w.dot = p; p += nh * w.dot_slots;
  w.g = p; p += (N + 15) / 16 * 16;
  w.lsum = p; p += (N + 15) / 16 * 16;
  w.ticket = (unsigned*)p; p += (N + 1 + 15) / 16 * 16;      // [0]: cx_loss_kernel, [1 + n]: sample n's row blocks
  w.inx = p; p += nh;
  w.iny = p; p += nh;
  w.D = p; p += nh * hw;
  w.cx = p;
  return w;
}

__device__ __forceinline__ float inv_norm(float ss) { return 1.0f / fmaxf(sqrtf(ss), 1e-12f); }   // F.normalize eps

// Workgroups are dispatched round-robin over the 8 XCDs, each with its own 4 MiB L2.  With the natural block order
// every XCD touched all samples (x + y = 7 MB at the loop's size) and its L2 thrashed: 75 % of wave time waiting on
// memory (SQ_WAIT_ANY).  Logical block = (bid % 8) * ceil(nb / 8) + bid / 8 gives each XCD one contiguous run of
// (sample, tile) pairs; the grid is rounded up to a multiple of 8 and surplus blocks exit.
__device__ __forceinline__ int xcd_block(int nb) {
  const int per = (nb + 7) >> 3;
  const int l = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  return l < nb ? l : -1;
}


// mu_c = mean over (n, pos) of y (functional.py:141), one workgroup per channel; the workgroups
// also clear the per-position accumulators of this call.
__global__ void cx_mean_kernel(const float* __restrict__ y, int N, int C, int hw, CxWs w) {
  __shared__ float red[4];
  const int c = blockIdx.x, grp = blockIdx.y;
  const int n0 = w.iter ? w.iter[grp].x0 : 0, ng = w.iter ? w.iter[grp].nk : N;
  float acc = 0.0f;
  for (int n = n0; n < n0 + ng; ++n)
    for (int p = threadIdx.x; p < hw; p += blockDim.x) acc += y[((int64_t)n * C + c) * hw + p];
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0 && ng > 0) w.mu[(int64_t)grp * w.mu_stride + c] = (red[0] + red[1] + red[2] + red[3]) / (float)((int64_t)ng * hw);
  if (grp) return;
  if (blockIdx.x == 0)
    for (int q = threadIdx.x; q <= N; q += blockDim.x) w.ticket[q] = 0u;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < (int64_t)N * hw; t += (int64_t)gridDim.x * blockDim.x) {
    w.dmin[t] = 0x7f800000u;   // +inf
    w.cmax[t] = 0u;
  }
}

// D = 1 - clamp(inx_i iny_j sum_c (x_ci - mu_c)(y_cj - mu_c), 0, 1), row minima by atomicMin.
// Workgroup = 64 x 64 output tile of one sample, 4 waves of 32 x 32; the channel axis is streamed
// through LDS in chunks of 32 ([32 c][64 pos] per operand, positions contiguous as in NCHW, so the
// fp32 MFMA operand reads are conflict-free), double buffered.
#ifndef NPP_CX_KC
#define NPP_CX_KC 32
#endif
constexpr int kCxKc = NPP_CX_KC;           // channels per staged chunk (32 or 64: measured the same, 23.6 / 23.3 us at the loop's size;
                                           // timing-only builds: without the MFMAs 20.6, without the D stores 23.5, without the row-minimum
                                           // atomics 21.8 -- ~15 us of the launch are none of these, tools/r4_cx_variants.sh)
constexpr int kCxNR = kCxKc / 16;          // rows of a chunk per thread
__global__ __launch_bounds__(256) void cx_sim_kernel(const float* __restrict__ x, const float* __restrict__ y, int N,
                                                     int C, int hw, CxWs w) {
  __shared__ __attribute__((aligned(16))) float sA[2][kCxKc][64];
  __shared__ __attribute__((aligned(16))) float sB[2][kCxKc][64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int tiles = (hw + 63) / 64;
  const int lb = xcd_block(N * tiles * tiles);          // whole block exits together: no barrier is skipped
  if (lb < 0) return;
  const int n = lb / (tiles * tiles);
  const int tij = lb - n * tiles * tiles;
  const int i0 = (tij / tiles) * 64, j0 = (tij % tiles) * 64;
  const int wi = wave >> 1, wj = wave & 1, l31 = lane & 31, kh = lane >> 5;
  const float* xn = x + (int64_t)n * C * hw;
  const float* yn = y + (int64_t)n * C * hw;
  const bool vec = (hw & 3) == 0;
  const float* mu = cx_mu(w, n);
  // Operand staging, round 4 (late): TWO chunks in flight in registers.  The first form loaded chunk c + 1, centred it and summed its
  // squares right away -- the first use of the loaded values sat in front of chunk c's MFMAs, so every chunk waited for its own load
  // (the features were just written by the trunk's last layer from other XCDs: ~1.5 us each, 8 chunks in a row) and nothing
  // overlapped.  Now the raw loads of chunk c + 2 are issued, chunk c is multiplied out of LDS, and only then chunk c + 1 (requested
  // a whole chunk earlier) is centred, squared and stored.  Same arithmetic in the same order: bit-identical.
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  float qa[4] = {0, 0, 0, 0}, qb[4] = {0, 0, 0, 0};
  auto body = [&](auto full_) {
  constexpr bool FULL = decltype(full_)::value;
  struct Raw { float4 a[kCxNR], b[kCxNR]; float m[kCxNR]; };      // (the channel means travel with the chunk: a load of their own in front of the
                                                      //  centring would be waited for in every chunk)
  Raw R[2];
  auto gissue = [&](int c0, Raw& q) {
#pragma unroll
    for (int r = 0; r < kCxNR; ++r) {
      const int idx = tid + 256 * r, row = idx >> 4, col = (idx & 15) * 4;
      const int c = c0 + row;
      if constexpr (FULL) {                              // whole tile inside the map, whole chunks: no predicate per element
        q.m[r] = mu[c];
        q.a[r] = *(const float4*)(xn + (int64_t)c * hw + i0 + col);
        q.b[r] = *(const float4*)(yn + (int64_t)c * hw + j0 + col);
        continue;
      }
      float va[4] = {0, 0, 0, 0}, vb[4] = {0, 0, 0, 0};
      q.m[r] = 0.0f;
      if (c < C) {
        const float m = mu[c];
        q.m[r] = m;
        if (vec && i0 + col + 3 < hw) { const float4 t = *(const float4*)(xn + (int64_t)c * hw + i0 + col); va[0] = t.x; va[1] = t.y; va[2] = t.z; va[3] = t.w; }
        else for (int e = 0; e < 4; ++e) if (i0 + col + e < hw) va[e] = xn[(int64_t)c * hw + i0 + col + e]; else va[e] = m;
        if (vec && j0 + col + 3 < hw) { const float4 t = *(const float4*)(yn + (int64_t)c * hw + j0 + col); vb[0] = t.x; vb[1] = t.y; vb[2] = t.z; vb[3] = t.w; }
        else for (int e = 0; e < 4; ++e) if (j0 + col + e < hw) vb[e] = yn[(int64_t)c * hw + j0 + col + e]; else vb[e] = m;
      }
      q.a[r] = make_float4(va[0], va[1], va[2], va[3]);
      q.b[r] = make_float4(vb[0], vb[1], vb[2], vb[3]);
    }
  };
  // Round 4: the squared norms sum_c (x - mu)^2, (y - mu)^2 of the tile's 64 + 64 positions are summed HERE, from the centred
  // operands every thread stages anyway (2 rows x 4 columns per chunk), instead of in a launch of their own (cx_sumsq_kernel:
  // 7 us of launch ramp for 0.2 us of arithmetic).  Every tile sums in the same order, so the tiles of a row agree bit for bit.
  // centre chunk c0 (in q), add its squares, store it as LDS buffer buf
  auto gfinish = [&](int c0, Raw& q, int buf) {
    // (the loaded values become visible HERE: without the empty asm the centring arithmetic -- pure VALU work, not ordered against a
    //  sched_barrier -- is hoisted above the MFMA loop together with the wait for its loads)
#pragma unroll
    for (int r = 0; r < kCxNR; ++r) {
      asm volatile("" : "+v"(q.a[r].x), "+v"(q.a[r].y), "+v"(q.a[r].z), "+v"(q.a[r].w));
      asm volatile("" : "+v"(q.b[r].x), "+v"(q.b[r].y), "+v"(q.b[r].z), "+v"(q.b[r].w), "+v"(q.m[r]));
    }
#pragma unroll
    for (int r = 0; r < kCxNR; ++r) {
      const int idx = tid + 256 * r, row = idx >> 4, col = (idx & 15) * 4;
      const float m = q.m[r];
      const float4 ra = make_float4(q.a[r].x - m, q.a[r].y - m, q.a[r].z - m, q.a[r].w - m);
      const float4 rb = make_float4(q.b[r].x - m, q.b[r].y - m, q.b[r].z - m, q.b[r].w - m);
      qa[0] = fmaf(ra.x, ra.x, qa[0]); qa[1] = fmaf(ra.y, ra.y, qa[1]); qa[2] = fmaf(ra.z, ra.z, qa[2]); qa[3] = fmaf(ra.w, ra.w, qa[3]);
      qb[0] = fmaf(rb.x, rb.x, qb[0]); qb[1] = fmaf(rb.y, rb.y, qb[1]); qb[2] = fmaf(rb.z, rb.z, qb[2]); qb[3] = fmaf(rb.w, rb.w, qb[3]);
      *(float4*)&sA[buf][row][col] = ra;
      *(float4*)&sB[buf][row][col] = rb;
    }
  };
  gissue(0, R[0]);
  if (kCxKc < C) gissue(kCxKc, R[1]);
  gfinish(0, R[0], 0);
  __syncthreads();
  // one chunk; PAR = (c0 / kCxKc) % 2 at compile time: chunk c0 sits in LDS buffer PAR, chunk c0 + 1 in register set PAR ^ 1, set PAR is free
  // one chunk; NEXT (compile time): another chunk follows.  Then BOTH the request for chunk c0 + 2 kc and the finish of chunk c0 + kc
  // are unconditional (the request is clamped: the chunk before the last asks for the last one again).  Found in the ISA, round 6:
  // with `if (c0 + 2 kc < C)` around the request, the wait in front of the LDS stores had to hold for the path WITHOUT new loads as
  // well -- the compiler emitted vmcnt(2..0), so every chunk also waited for the six loads it had just issued and "two chunks in
  // flight" was one; and loads whose only use sits under a condition are sunk into it.  Same arithmetic, same order.
  auto chunk = [&](int c0, auto par_, auto next_) {
    constexpr int PAR = decltype(par_)::value;
    constexpr bool NEXT = decltype(next_)::value;
    if constexpr (NEXT) gissue(max(0, min(c0 + 2 * kCxKc, (C - 1) / kCxKc * kCxKc)), R[PAR]);
    __builtin_amdgcn_sched_barrier(0);      // (requests first, the finish last: left alone the scheduler moves the centring of the next
#pragma unroll                              //  chunk -- and the wait for its loads -- up among the first MFMAs and the requests down)
    for (int ks = 0; ks < kCxKc / 2; ++ks) {
      const float a = sA[PAR][2 * ks + kh][wi * 32 + l31];
      const float b = sB[PAR][2 * ks + kh][wj * 32 + l31];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (NEXT) gfinish(c0 + kCxKc, R[PAR ^ 1], PAR ^ 1);
    __syncthreads();
  };
  using P0 = std::integral_constant<int, 0>; using P1 = std::integral_constant<int, 1>;
  int c0 = 0;
  for (; c0 + 2 * kCxKc < C; c0 += 2 * kCxKc) { chunk(c0, P0{}, std::true_type{}); chunk(c0 + kCxKc, P1{}, std::true_type{}); }
  if (c0 + kCxKc < C) { chunk(c0, P0{}, std::true_type{}); chunk(c0 + kCxKc, P1{}, std::false_type{}); }
  else chunk(c0, P0{}, std::false_type{});
  };
  if (vec && i0 + 64 <= hw && j0 + 64 <= hw && C % kCxKc == 0) body(std::true_type{});      // (block-uniform)
  else body(std::false_type{});
  // the 16 row classes' partials (thread tid holds rows tid >> 4 and 16 + (tid >> 4) of every chunk) meet in LDS, summed in class order
  float* ssl = &sA[0][0][0];                           // [2][16][64] floats: the operand buffers are idle now
  {
    const int cls = tid >> 4, col = (tid & 15) * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) { ssl[(0 * 16 + cls) * 64 + col + e] = qa[e]; ssl[(1 * 16 + cls) * 64 + col + e] = qb[e]; }
  }
  __syncthreads();
  float* ssn = &sB[0][0][0];                           // [2][64]: the tile's squared norms (x positions | y positions)
  if (tid < 128) {
    const int which = tid >> 6, col = tid & 63;
    float v = 0.0f;
#pragma unroll
    for (int q = 0; q < 16; ++q) v += ssl[(which * 16 + q) * 64 + col];
    ssn[which * 64 + col] = v;
    const int pos = (which ? j0 : i0) + col;
    if (pos < hw && (which ? i0 : j0) == 0) (which ? w.ssy : w.ssx)[(int64_t)n * hw + pos] = v;      // one tile column / row publishes them
  }
  __syncthreads();
  // accumulator: column (lane & 31) = j, register r = row acc_row(r, lane >> 5) = i.  The row
  // minimum is reduced over the 32 columns of the half-wave first: one atomic per row and wave
  // instead of one per element (the element-wise form was atomic-bound: 2 M atomics per call).
  const int j = j0 + wj * 32 + l31;
  const float sj = j < hw ? inv_norm(ssn[64 + wj * 32 + l31]) : 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = i0 + wi * 32 + acc_row(r, kh);
    float d = 2.0f;                                  // neutral for the minimum (D <= 1)
    if (i < hw && j < hw) {
      const float raw = acc[r] * inv_norm(ssn[wi * 32 + acc_row(r, kh)]) * sj;
      d = 1.0f - fminf(fmaxf(raw, 0.0f), 1.0f);
      w.D[((int64_t)n * hw + i) * hw + j] = d;
    }
    if (w.min_in_rows) continue;                       // (uniform) the block-parallel row pass reads the whole row anyway: 3 us of atomics saved
    float m = d;
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) m = fminf(m, __shfl_xor(m, off, 64));
    if (l31 == 0 && i < hw) atomicMin(&w.dmin[(int64_t)n * hw + i], __float_as_uint(m));
  }
}

// w = exp((1 - D/(dmin + 1e-5))/h), s = sum_j w, cx = w/s, column maxima.  One wave walks kCxRows
// consecutive rows of one sample and keeps the running column maxima of its lanes in registers:
// one atomicMax per column per kCxRows rows.
constexpr int kCxRows = 8;
constexpr int kCxMaxCols = 32;        // columns per lane held in registers: hw <= 64 * 32 = 2048
__global__ __launch_bounds__(256) void cx_rows_fwd_kernel(int N, int hw, float inv_h, CxWs w) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int groups = (hw + kCxRows - 1) / kCxRows;
  const int64_t gid = (int64_t)blockIdx.x * 4 + wave;
  if (gid >= (int64_t)N * groups) return;
  const int n = (int)(gid / groups), r0 = (int)(gid - (int64_t)n * groups) * kCxRows;
  float cmaxv[kCxMaxCols];
#pragma unroll
  for (int q = 0; q < kCxMaxCols; ++q) cmaxv[q] = 0.0f;
  for (int rr = 0; rr < kCxRows && r0 + rr < hw; ++rr) {
    const int64_t row = (int64_t)n * hw + r0 + rr;
    const float dm = __uint_as_float(w.dmin[row]) + 1e-5f;
    const float* Dr = w.D + row * hw;
    float* cr = w.cx + row * hw;
    float wv[kCxMaxCols];
    float s = 0.0f;
#pragma unroll
    for (int q = 0; q < kCxMaxCols; ++q) {
      const int j = lane + 64 * q;
      wv[q] = j < hw ? __expf((1.0f - Dr[j] / dm) * inv_h) : 0.0f;
      s += wv[q];
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) w.s[row] = s;
    const float inv = 1.0f / s;
#pragma unroll
    for (int q = 0; q < kCxMaxCols; ++q) {
      const int j = lane + 64 * q;
      if (j < hw) {
        const float c = wv[q] * inv;
        cr[j] = c;
        cmaxv[q] = fmaxf(cmaxv[q], c);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < kCxMaxCols; ++q) {
    const int j = lane + 64 * q;
    if (j < hw) atomicMax(&w.cmax[(int64_t)n * hw + j], __float_as_uint(cmaxv[q]));
  }
}

// The same row pass for hw > 64 * kCxMaxCols (whole-image crops of the proposal ranking, NPP_proposal/search.py:180-197): the
// row sums first (per lane over j = lane + 64 q ascending, then the same butterfly -- the summation order of the kernel
// above), then the columns in chunks of 64 * kCxMaxCols with the chunk's running maxima in registers.
__global__ __launch_bounds__(256) void cx_rows_fwd_big_kernel(int N, int hw, float inv_h, CxWs w) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int groups = (hw + kCxRows - 1) / kCxRows;
  const int64_t gid = (int64_t)blockIdx.x * 4 + wave;
  if (gid >= (int64_t)N * groups) return;
  const int n = (int)(gid / groups), r0 = (int)(gid - (int64_t)n * groups) * kCxRows;
  float inv[kCxRows], dmr[kCxRows];
#pragma unroll
  for (int rr = 0; rr < kCxRows; ++rr) {
    inv[rr] = 0.0f; dmr[rr] = 1.0f;
    if (r0 + rr < hw) {
      const int64_t row = (int64_t)n * hw + r0 + rr;
      const float dm = __uint_as_float(w.dmin[row]) + 1e-5f;
      const float* Dr = w.D + row * hw;
      float s = 0.0f;
      for (int j = lane; j < hw; j += 64) s += __expf((1.0f - Dr[j] / dm) * inv_h);
      for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
      if (lane == 0) w.s[row] = s;
      inv[rr] = 1.0f / s; dmr[rr] = dm;
    }
  }
  for (int c0 = 0; c0 < hw; c0 += 64 * kCxMaxCols) {
    float cmaxv[kCxMaxCols];
#pragma unroll
    for (int q = 0; q < kCxMaxCols; ++q) cmaxv[q] = 0.0f;
#pragma unroll
    for (int rr = 0; rr < kCxRows; ++rr) {
      if (r0 + rr >= hw) continue;
      const int64_t row = (int64_t)n * hw + r0 + rr;
      const float* Dr = w.D + row * hw;
      float* cr = w.cx + row * hw;
#pragma unroll
      for (int q = 0; q < kCxMaxCols; ++q) {
        const int j = c0 + lane + 64 * q;
        if (j < hw) {
          const float c = __expf((1.0f - Dr[j] / dmr[rr]) * inv_h) * inv[rr];
          cr[j] = c;
          cmaxv[q] = fmaxf(cmaxv[q], c);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < kCxMaxCols; ++q) {
      const int j = c0 + lane + 64 * q;
      if (j < hw) atomicMax(&w.cmax[(int64_t)n * hw + j], __float_as_uint(cmaxv[q]));
    }
  }
}

// ---- forward-only form for whole-image crops (hw > 64 * kCxMaxCols and no gradient wanted: the proposal ranking's score of a
// candidate, NPP_proposal/search.py:180-197, on 128 x 128 feature positions: 137 GFLOP and a 1 GiB distance matrix per score) --------
// cx_sim_kernel's 64 x 64 tiles ran that at 0.19 of the fp32 MFMA peak (4.5 ms: every 64 x 64 x 32 chunk is 16 KiB of operands for 16
// MFMAs per wave and one accumulator chain) and the 8-rows-per-wave row pass at 1.4 TB/s (2.3 ms: it also writes the cx matrix only
// the backward pass reads, and meets the column maxima with 33 M atomics).  Here: 128 x 128 tiles (four accumulator chains per wave,
// half the operand bytes per FLOP), the row sums as one streaming read, the column maxima as a second one over 64-column strips with
// the running maximum in a register -- the matrix is written once and read twice, nothing else is stored.
// Operands first (cx_prep_big_kernel): x and y centred, NORMALISED and re-tiled as [sample][tile of 128 positions][channel][128] --
// a tile's operand stream is then 128 KiB of consecutive bytes.  Read straight from the NCHW features every 16-channel chunk was 16
// rows of 512 B, 64 KiB apart: the load skeleton alone (MFMAs compiled out) took 1.85 ms of the first form's 2.5.
__global__ __launch_bounds__(256) void cx_prep_big_kernel(const float* __restrict__ x, const float* __restrict__ y, int N, int C, int hw,
                                                          CxWs w, float* __restrict__ xt, float* __restrict__ yt) {
  __shared__ float red[2][128];
  const int tiles = (hw + 127) / 128;
  int b = blockIdx.x;
  const int tile = b % tiles; b /= tiles;
  const int which = b & 1, n = b >> 1;
  if (n >= N) return;
  const float* src = (which ? y : x) + (int64_t)n * C * hw;
  float* dst = (which ? yt : xt) + ((int64_t)n * tiles + tile) * C * 128;
  const float* mu = cx_mu(w, n);
  const int pl = threadIdx.x & 127, half = threadIdx.x >> 7, pos = tile * 128 + pl;
  const bool live = pos < hw;
  float ss = 0.0f;
  for (int c = half; c < C; c += 2) {
    const float v = live ? src[(int64_t)c * hw + pos] - mu[c] : 0.0f;
    ss = fmaf(v, v, ss);
  }
  red[half][pl] = ss;
  __syncthreads();
  const float tot = red[0][pl] + red[1][pl];
  if (half == 0 && live) (which ? w.ssy : w.ssx)[(int64_t)n * hw + pos] = tot;
  const float inv = inv_norm(tot);
  for (int c = half; c < C; c += 2) dst[c * 128 + pl] = live ? (src[(int64_t)c * hw + pos] - mu[c]) * inv : 0.0f;
}

#ifndef NPP_CXBIG_KC
#define NPP_CXBIG_KC 16
#endif
constexpr int kBigKc = NPP_CXBIG_KC;
__global__ __launch_bounds__(256) void cx_sim_big_kernel(const float* __restrict__ xt, const float* __restrict__ yt, int N, int C, int hw,
                                                         CxWs w) {
  __shared__ __attribute__((aligned(16))) float sA[2][kBigKc][128];
  __shared__ __attribute__((aligned(16))) float sB[2][kBigKc][128];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // Tile order: an XCD's run of workgroups walks 8 x 8 SUPER-BLOCKS of tiles.  Row-major, the ~64 workgroups resident on an XCD held
  // one x tile and 64 different y tiles (8.3 MB of operands against 4 MB of L2): every y tile missed, and the kernel ran at the
  // L2-miss bandwidth (2.7 TB/s, 32 FLOP/B) whatever the MFMAs did.  A super-block needs 8 + 8 tiles = 2 MB.
  const int tiles = (hw + 127) / 128, t8 = (tiles + 7) / 8;
  const int lb = xcd_block(N * t8 * t8 * 64);
  if (lb < 0) return;
  const int n = lb / (t8 * t8 * 64);
  const int rem = lb - n * t8 * t8 * 64, sb = rem >> 6, win = rem & 63;
  const int ti = (sb / t8) * 8 + (win >> 3), tj = (sb % t8) * 8 + (win & 7), i0 = ti * 128, j0 = tj * 128;
  if (ti >= tiles || tj >= tiles) return;               // (whole block)
  const int wi = wave >> 1, wj = wave & 1, l31 = lane & 31, kh = lane >> 5;
  const float4* xa = (const float4*)(xt + ((int64_t)n * tiles + ti) * C * 128);
  const float4* yb = (const float4*)(yt + ((int64_t)n * tiles + tj) * C * 128);
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
  // chunk c0 = channels [c0, c0 + 16): 512 consecutive float4 per operand, two per thread.  Two chunks in flight in NAMED registers
  // (an array of structs handed to the staging lambdas by reference went to scratch memory, and every load was waited for where it
  // was issued: 2.1 ms with or without the MFMAs)
  static_assert(kBigKc == 16, "two float4 per thread and operand");
  float4 pa0, pa1, pb0, pb1, qa0, qa1, qb0, qb1;       // set P: even chunks, set Q: odd chunks
#define CXBIG_ISSUE(c0, A0, A1, B0, B1) { A0 = xa[(c0) * 32 + tid]; A1 = xa[(c0) * 32 + tid + 256]; B0 = yb[(c0) * 32 + tid]; B1 = yb[(c0) * 32 + tid + 256]; }
#define CXBIG_FINISH(buf, A0, A1, B0, B1) { float4* da = (float4*)&sA[buf][0][0]; float4* db = (float4*)&sB[buf][0][0]; \
    da[tid] = A0; da[tid + 256] = A1; db[tid] = B0; db[tid + 256] = B1; }
#ifdef NPP_CXBIG_NOMFMA          // (timing-only build)
#define CXBIG_MMA(PAR) _Pragma("unroll") for (int ks = 0; ks < 8; ++ks) { \
      const float a0 = sA[PAR][2 * ks + kh][wi * 64 + l31], a1 = sA[PAR][2 * ks + kh][wi * 64 + 32 + l31]; \
      const float b0 = sB[PAR][2 * ks + kh][wj * 64 + l31], b1 = sB[PAR][2 * ks + kh][wj * 64 + 32 + l31]; \
      acc[0][0][ks] += a0 * b0; acc[0][1][ks] += a0 * b1; acc[1][0][ks] += a1 * b0; acc[1][1][ks] += a1 * b1; }
#else
#define CXBIG_MMA(PAR) _Pragma("unroll") for (int ks = 0; ks < 8; ++ks) { \
      const float a0 = sA[PAR][2 * ks + kh][wi * 64 + l31], a1 = sA[PAR][2 * ks + kh][wi * 64 + 32 + l31]; \
      const float b0 = sB[PAR][2 * ks + kh][wj * 64 + l31], b1 = sB[PAR][2 * ks + kh][wj * 64 + 32 + l31]; \
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0); \
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0); \
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0); \
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0); }
#endif
  CXBIG_ISSUE(0, pa0, pa1, pb0, pb1);
  CXBIG_ISSUE(16, qa0, qa1, qb0, qb1);                  // (C >= 32)
  CXBIG_FINISH(0, pa0, pa1, pb0, pb1);
  __syncthreads();
  for (int c0 = 0; c0 < C; c0 += 32) {                  // (C % 32 == 0: chunk pairs)
    // even chunk c0 in LDS buffer 0, chunk c0 + 16 in flight in Q
    // (the requests are unconditional -- the last pair asks for the last chunks again: a conditional request makes the compiler wait
    //  for ALL outstanding loads in front of the LDS stores, see cx_sim_kernel)
    CXBIG_ISSUE(min(c0 + 32, C - 32), pa0, pa1, pb0, pb1);
    __builtin_amdgcn_sched_barrier(0);                  // (the scheduler sank the requests below the MFMAs, next to their use)
    CXBIG_MMA(0);
    __builtin_amdgcn_sched_barrier(0);
    CXBIG_FINISH(1, qa0, qa1, qb0, qb1);
    __syncthreads();
    CXBIG_ISSUE(min(c0 + 48, C - 16), qa0, qa1, qb0, qb1);
    __builtin_amdgcn_sched_barrier(0);
    CXBIG_MMA(1);
    __builtin_amdgcn_sched_barrier(0);
    CXBIG_FINISH(0, pa0, pa1, pb0, pb1);                // (unconditional too: loads used only under a condition get SUNK into it)
    __syncthreads();
  }
#undef CXBIG_ISSUE
#undef CXBIG_FINISH
#undef CXBIG_MMA
  // the distance matrix is stored TILED ([sample][ti][tj][128][128]: this tile is 64 KiB of consecutive bytes; in row-major form
  // every store instruction was two 128-byte pieces 64 KiB apart and the launch ran at the scattered-line write rate, 0.7 TB/s)
  float* Dt = w.D + (((int64_t)n * tiles + ti) * tiles + tj) * 16384;
#pragma unroll
  for (int bi = 0; bi < 2; ++bi)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = i0 + wi * 64 + bi * 32 + acc_row(r, kh);
      float m = 2.0f;
#pragma unroll
      for (int bj = 0; bj < 2; ++bj) {
        const int j = j0 + wj * 64 + bj * 32 + l31;
        if (i < hw && j < hw) {
          const float d = 1.0f - fminf(fmaxf(acc[bi][bj][r], 0.0f), 1.0f);
#ifdef NPP_CXBIG_NOSTORE         // (timing-only build)
          if (d == 12345.0f)
#endif
          Dt[(i - i0) * 128 + (j - j0)] = d;
          m = fminf(m, d);
        }
      }
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) m = fminf(m, __shfl_xor(m, off, 64));
#ifdef NPP_CXBIG_NOATOMIC        // (timing-only build)
      if (m == 12345.0f)
#endif
      if (l31 == 0 && i < hw) atomicMin(&w.dmin[(int64_t)n * hw + i], __float_as_uint(m));
    }
}

// s_i = sum_j exp((1 - D_ij / (dmin_i + 1e-5)) / h) over the tiled matrix: one workgroup per (sample, tile row, quarter of the tile
// columns); wave w owns rows w, w + 4, .. of the 128 (32 partial sums per lane), lane l the column pair 2 l: every load instruction is one
// 512-byte row of a tile.  The quarters' sums meet in the order of arrival-independent slots (s4), added by the column pass.
constexpr int kBigQ = 4;                    // column-tile ranges per tile row
__global__ __launch_bounds__(256) void cx_rowsum_big_kernel(int N, int hw, float inv_h, CxWs w, float* __restrict__ s4) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tiles = (hw + 127) / 128;
  int b = blockIdx.x;
  const int q = b % kBigQ; b /= kBigQ;
  const int ti = b % tiles, n = b / tiles;
  if (n >= N) return;
  float rdm[32], acc[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    const int i = ti * 128 + wave + 4 * k;
    rdm[k] = i < hw ? 1.0f / (__uint_as_float(w.dmin[(int64_t)n * hw + i]) + 1e-5f) : 0.0f;
    acc[k] = 0.0f;
  }
  const int tq = (tiles + kBigQ - 1) / kBigQ;
  for (int tj = q * tq; tj < min(tiles, (q + 1) * tq); ++tj) {
    const float* Dt = w.D + (((int64_t)n * tiles + ti) * tiles + tj) * 16384 + wave * 128 + 2 * lane;
    const int j = tj * 128 + 2 * lane;
    const bool v0 = j < hw, v1 = j + 1 < hw;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      // (entries beyond the map were never written: SELECTED out, not multiplied by 0 -- stale workspace bytes there can be anything,
      //  and 0 * exp(garbage) = NaN made the row's sum NaN, which the column pass's fmaxf then silently dropped)
      const float2 d = *(const float2*)(Dt + k * 512);
      acc[k] += (v0 ? __expf((1.0f - d.x * rdm[k]) * inv_h) : 0.0f) + (v1 ? __expf((1.0f - d.y * rdm[k]) * inv_h) : 0.0f);
    }
  }
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    float v = acc[k];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    const int i = ti * 128 + wave + 4 * k;
    if (lane == 0 && i < hw) s4[((int64_t)n * hw + i) * kBigQ + q] = v;
  }
}

// cmax_j = max_i exp(.) / s_i: one workgroup per (sample, tile column, range of tile rows); lane = column pair, the running maxima in
// registers; one atomicMax per column and range (cmax cleared by cx_mean_kernel).  s_i = the quarters' sums in quarter order.
constexpr int kBigR = 8;                    // tile-row ranges per tile column
__global__ __launch_bounds__(256) void cx_colmax_big_kernel(int N, int hw, float inv_h, CxWs w, const float* __restrict__ s4) {
  __shared__ float red[4][128];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tiles = (hw + 127) / 128;
  int b = blockIdx.x;
  const int rg = b % kBigR; b /= kBigR;
  const int tj = b % tiles, n = b / tiles;
  if (n >= N) return;
  const int tr = (tiles + kBigR - 1) / kBigR;
  float best0 = 0.0f, best1 = 0.0f;
  for (int ti = rg * tr; ti < min(tiles, (rg + 1) * tr); ++ti) {
    const float* Dt = w.D + (((int64_t)n * tiles + ti) * tiles + tj) * 16384 + wave * 128 + 2 * lane;
    // this wave's 32 rows: lane k < 32 fetches row k's scalars, handed round by shuffles
    const int irow = ti * 128 + wave + 4 * (lane & 31);
    float my_rdm = 0.0f, my_rs = 0.0f;
    if (irow < hw) {
      const int64_t row = (int64_t)n * hw + irow;
      my_rdm = 1.0f / (__uint_as_float(w.dmin[row]) + 1e-5f);
      const float* sq = s4 + row * kBigQ;
      my_rs = 1.0f / (((sq[0] + sq[1]) + sq[2]) + sq[3]);
    }
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      const float2 d = *(const float2*)(Dt + k * 512);
      const float rdm = __shfl(my_rdm, k, 64), rs = __shfl(my_rs, k, 64);        // (rows beyond the map: rs = 0, selected out)
      const float c0 = __expf((1.0f - d.x * rdm) * inv_h) * rs, c1 = __expf((1.0f - d.y * rdm) * inv_h) * rs;
      best0 = fmaxf(best0, rs > 0.0f ? c0 : 0.0f);
      best1 = fmaxf(best1, rs > 0.0f ? c1 : 0.0f);
    }
  }
  red[wave][2 * lane] = best0;
  red[wave][2 * lane + 1] = best1;
  __syncthreads();
  if (threadIdx.x < 128) {
    const int j = tj * 128 + threadIdx.x;
    const float v = fmaxf(fmaxf(red[0][threadIdx.x], red[1][threadIdx.x]), fmaxf(red[2][threadIdx.x], red[3][threadIdx.x]));
    if (j < hw) atomicMax(&w.cmax[(int64_t)n * hw + j], __float_as_uint(v));
  }
}

// per sample (one block each): cxn = mean_j cmax, l_n = -log(cxn [* weight] + 1e-5) [/ N_group], g = dL/dcxn / J, and the inverse
// norms of the sample's positions (once instead of once per use).  The block that arrives last adds each group's l_n in sample
// order to loss[group * loss_stride] (round 4: one float atomic per sample in arrival order before).
__global__ void cx_loss_kernel(int N, int hw, const float* __restrict__ weight, float scale, float* __restrict__ loss,
                               int loss_stride, CxWs w) {
  __shared__ float red[4];
  const int n = blockIdx.x;
  int ng;
  cx_group(w, n, ng);
  float acc = 0.0f;
  for (int j = threadIdx.x; j < hw; j += blockDim.x) {
    acc += __uint_as_float(w.cmax[(int64_t)n * hw + j]);
    w.inx[(int64_t)n * hw + j] = inv_norm(w.ssx[(int64_t)n * hw + j]);
    w.iny[(int64_t)n * hw + j] = inv_norm(w.ssy[(int64_t)n * hw + j]);
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float cxn = (red[0] + red[1] + red[2] + red[3]) / (float)hw;
    float l, dcxn;
    if (weight) {                       // functional.py:55-57: sum(-log(cx * w + 1e-5))
      const float wt = weight[n];
      l = -logf(cxn * wt + 1e-5f);
      dcxn = -wt / (cxn * wt + 1e-5f);
    } else {                            // mean over the group's samples
      l = -logf(cxn + 1e-5f) / (float)ng;
      dcxn = -1.0f / ((float)ng * (cxn + 1e-5f));
    }
    share_store(w.lsum + n, scale * l);
    w.g[n] = scale * dcxn / (float)hw;
  }
  if (!block_last_arriver(w.ticket, N)) return;
  const int groups = w.iter ? w.M : 1;
  if ((int)threadIdx.x < groups) {
    const int grp = threadIdx.x;
    const int n0 = w.iter ? w.iter[grp].x0 : 0, cnt = w.iter ? w.iter[grp].nk : N;
    float total = 0.0f;
    for (int q = n0; q < n0 + cnt; ++q) total += share_load(w.lsum + q);
    if (cnt > 0) atomicAdd(loss + (int64_t)grp * loss_stride, total);
  }
}

// one wave per row: cx row -> d raw row (in place).
__global__ __launch_bounds__(256) void cx_rows_bwd_kernel(int N, int hw, float inv_h, CxWs w, float* __restrict__ loss, int loss_stride) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (loss && blockIdx.x == 0 && (int)threadIdx.x < (w.iter ? w.M : 1)) {     // the groups' loss sums, in sample order (lsum: previous launch)
    const int grp = threadIdx.x;
    const int n0 = w.iter ? w.iter[grp].x0 : 0, cnt = w.iter ? w.iter[grp].nk : N;
    float total = 0.0f;
    for (int q = n0; q < n0 + cnt; ++q) total += w.lsum[q];
    if (cnt > 0) atomicAdd(loss + (int64_t)grp * loss_stride, total);
  }
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;
  if (row >= (int64_t)N * hw) return;
  const int n = (int)(row / hw);
  const float dmv = __uint_as_float(w.dmin[row]);
  const float dm = dmv + 1e-5f;
  const float s = w.s[row], g = w.g[n];
  const float* Dr = w.D + row * hw;
  float* cr = w.cx + row * hw;
  const unsigned* cm = w.cmax + (int64_t)n * hw;
  // A_i = sum_j G_ij cx_ij over the columns whose maximum sits in this row
  float A = 0.0f;
  for (int j = lane; j < hw; j += 64) {
    const float c = cr[j];
    if (__float_as_uint(c) == cm[j]) A += g * c;
  }
  for (int off = 32; off > 0; off >>= 1) A += __shfl_xor(A, off, 64);
  float corr = 0.0f;
  int jmin = 0x7fffffff;
  for (int j = lane; j < hw; j += 64) {
    const float c = cr[j], d = Dr[j];
    const float G = (__float_as_uint(c) == cm[j]) ? g : 0.0f;
    const float wv = c * s;
    const float dDt = -((G - A) / s) * wv * inv_h;
    corr += dDt * d;
    const float dD = dDt / dm;
    cr[j] = (d > 0.0f && d < 1.0f) ? -dD : 0.0f;       // clamp(raw, 0, 1) passes gradient inside (0,1)
    if (d == dmv) jmin = min(jmin, j);
  }
  for (int off = 32; off > 0; off >>= 1) {
    corr += __shfl_xor(corr, off, 64);
    jmin = min(jmin, __shfl_xor(jmin, off, 64));
  }
  if (jmin < hw && lane == (jmin & 63)) {     // the lane that stored cr[jmin]: same-lane program order
    const float d = Dr[jmin];
    if (d > 0.0f && d < 1.0f) cr[jmin] += corr / (dm * dm);   // -(-corr/dm^2): the min's own gradient path
  }
}

// dxh[c][i] = sum_j draw[i][j] * yh[c][j], yh = (y - mu_c) * iny[j].  Workgroup = 64 channels x 64
// positions, 4 waves of 32 x 32; both operands are contiguous along the contraction index j, so a
// chunk of 32 columns is staged TRANSPOSED into LDS ([32 j][64 rows + 1 pad]: the fp32 MFMA reads
// 32 consecutive rows of one j) and double buffered.  The epilogue also accumulates the per-position
// dot product xh . dxh that the normalisation backward needs.
__global__ __launch_bounds__(256) void cx_dx_kernel(const float* __restrict__ x, const float* __restrict__ y, int N, int C,
                                                    int hw, CxWs w, float* __restrict__ dxh) {
  __shared__ float sA[2][kCxKc][65];
  __shared__ float sB[2][kCxKc][65];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int itiles = (hw + 63) / 64, ctiles = (C + 63) / 64;
  const int n = blockIdx.x / (ctiles * itiles);
  const int rem = blockIdx.x - n * ctiles * itiles;
  const int c0 = (rem / itiles) * 64, i0 = (rem % itiles) * 64;
  const int wc = wave >> 1, wi = wave & 1, l31 = lane & 31, kh = lane >> 5;
  const bool vec = (hw & 3) == 0;
  const float* iny_ss = w.ssy + (int64_t)n * hw;
  const float* mup = cx_mu(w, n);
  float4 ra[2], rb[2];
  auto gload = [&](int j0) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int idx = tid + 256 * r, row = idx >> 3, col = (idx & 7) * 4;     // 64 rows x 8 float4
      const int j = j0 + col;
      float va[4] = {0, 0, 0, 0}, vb[4] = {0, 0, 0, 0};
      const int c = c0 + row, i = i0 + row;
      if (c < C) {
        const float m = mup[c];
        const float* yr = y + ((int64_t)n * C + c) * hw;
        if (vec && j + 3 < hw) { const float4 q = *(const float4*)(yr + j); va[0] = q.x - m; va[1] = q.y - m; va[2] = q.z - m; va[3] = q.w - m; }
        else for (int e = 0; e < 4; ++e) if (j + e < hw) va[e] = yr[j + e] - m;
        for (int e = 0; e < 4; ++e) va[e] = (j + e < hw) ? va[e] * inv_norm(iny_ss[j + e]) : 0.0f;
      }
      if (i < hw) {
        const float* dr = w.cx + ((int64_t)n * hw + i) * hw;
        if (vec && j + 3 < hw) { const float4 q = *(const float4*)(dr + j); vb[0] = q.x; vb[1] = q.y; vb[2] = q.z; vb[3] = q.w; }
        else for (int e = 0; e < 4; ++e) if (j + e < hw) vb[e] = dr[j + e];
      }
      ra[r] = make_float4(va[0], va[1], va[2], va[3]);
      rb[r] = make_float4(vb[0], vb[1], vb[2], vb[3]);
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int idx = tid + 256 * r, row = idx >> 3, col = (idx & 7) * 4;
      sA[buf][col + 0][row] = ra[r].x; sA[buf][col + 1][row] = ra[r].y; sA[buf][col + 2][row] = ra[r].z; sA[buf][col + 3][row] = ra[r].w;
      sB[buf][col + 0][row] = rb[r].x; sB[buf][col + 1][row] = rb[r].y; sB[buf][col + 2][row] = rb[r].z; sB[buf][col + 3][row] = rb[r].w;
    }
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  gload(0);
  sstore(0);
  __syncthreads();
  int buf = 0;
  for (int j0 = 0; j0 < hw; j0 += kCxKc) {
    const bool has_next = j0 + kCxKc < hw;
    if (has_next) gload(j0 + kCxKc);
#pragma unroll
    for (int ks = 0; ks < kCxKc / 2; ++ks) {
      const float a = sA[buf][2 * ks + kh][wc * 32 + l31];     // A[m = channel][k = j]
      const float b = sB[buf][2 * ks + kh][wi * 32 + l31];     // B[k = j][n = position]
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    if (has_next) sstore(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  // accumulator: column = position (lane & 31), rows = channels
  const int i = i0 + wi * 32 + l31;
  float part = 0.0f;
  if (i < hw) {
    const float inx = inv_norm(w.ssx[(int64_t)n * hw + i]);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = c0 + wc * 32 + acc_row(r, kh);
      if (c < C) {
        const int64_t o = ((int64_t)n * C + c) * hw + i;
        dxh[o] = acc[r];
        part = fmaf((x[o] - mup[c]) * inx, acc[r], part);
      }
    }
  }
  part += __shfl_xor(part, 32, 64);
  const int slot = c0 / 32 + wc;                       // one partial per 32-channel tile, summed in order by cx_dx_finish
  if (kh == 0 && i < hw && slot < w.dot_slots) w.dot[(int64_t)slot * N * hw + (int64_t)n * hw + i] = part;
}

// dx = (dxh - xh (xh . dxh)) * inx, xh = (x - mu) inx ; elementwise, in place on dxh
__global__ void cx_dx_finish_kernel(const float* __restrict__ x, int N, int C, int hw, CxWs w, float* __restrict__ dx, int n_dot) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)N * C * hw) return;
  const int p = (int)(t % hw);
  const int64_t nc = t / hw;
  const int c = (int)(nc % C), n = (int)(nc / C);
  const float inx = inv_norm(w.ssx[(int64_t)n * hw + p]);
  float dot = 0.0f;
  for (int q = 0; q < n_dot; ++q) dot += w.dot[(int64_t)q * N * hw + (int64_t)n * hw + p];     // fixed order
  dx[t] = (dx[t] - (x[t] - cx_mu(w, n)[c]) * inx * dot) * inx;
}

// The same finish written STRAIGHT into the trunk's flat bf16 gradient tensor (npp_trunk_layout.h), gated by the ReLU of the tapped
// layer: dz = dx * [y > 0] -- what cx_dx_finish_kernel + npp_trunk_grad_in did in two launches and an fp32 round trip
// (round 4).  One thread = one 16-byte unit (8 channels of one position): the position's dot product and inverse norm are
// formed once per unit.  Border / tail positions of the run are rewritten as zeros (the layout's invariant).
__global__ void cx_dx_finish_flat_kernel(const float* __restrict__ x, const float* __restrict__ dxh, int N, int C, int H, int W,
                                         CxWs w, const f16x8* __restrict__ yact, bf16x8* __restrict__ dz, int64_t nposp,
                                         int64_t npos_range, int n_dot) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int chunks = C / 8;
  if (t >= npos_range * chunks) return;
  const int c8 = (int)(t / npos_range);
  const int64_t p = t - (int64_t)c8 * npos_range;
  const int Wp = W + 2, S = (H + 2) * Wp, hw = H * W;
  const int n = (int)(p / S), r = (int)(p - (int64_t)n * S), yy = r / Wp, xx = r - yy * Wp;
  const int64_t u = (int64_t)c8 * nposp + kConvGuard + p;
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (__bf16)0.0f;
  if (n < N && yy >= 1 && yy <= H && xx >= 1 && xx <= W) {
    const int pos = (yy - 1) * W + (xx - 1);
    const int64_t np = (int64_t)n * hw + pos;
    const float inx = inv_norm(w.ssx[np]);
    float dot = 0.0f;
    for (int q = 0; q < n_dot; ++q) dot += w.dot[(int64_t)q * N * hw + np];          // fixed order
    const float* mu = cx_mu(w, n);
    const f16x8 m = yact[u];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = conv_chan(c8, j);
      const int64_t e = ((int64_t)n * C + c) * hw + pos;
      const float v = (dxh[e] - (x[e] - mu[c]) * inx * dot) * inx;
      o[j] = (__bf16)((float)m[j] > 0.0f ? v : 0.0f);
    }
  }
  dz[u] = o;
}

// ---- second-generation backward contraction and row pass (hw % 32 == 0) ---------------------------------------
// What was measured on the loop's size (6 x 576 x 576 x 256; rocprofv3 + SQ counters, tools/cx_probe.py):
//  * the row pass was serial (8 rows per wave, 36 us) -> one row per wave, 16 waves per block: 10 us;
//  * the backward contraction spent 10 k VALU instructions per wave on sqrt + div for the inverse norms of its operands
//    (35 x its MFMA count) -> inverse norms computed once (cx_loss_kernel), operands as float4 rows straight from L2
//    with 8 blocks in flight: 68 -> 31 us;
//  * feeding v_mfma_f32_32x32x2_f32 (ONE float per lane and operand) straight from global memory does NOT work for the
//    forward contraction, whose operands are contiguous along positions: it needs a dword load instruction per MFMA
//    and is bound by the texture addresser (48 - 74 us against 28 us for the LDS-tiled cx_sim_kernel, whatever the
//    prefetch depth), so that kernel stays LDS-tiled;
//  * blocks are dealt round-robin to the 8 XCDs (4 MiB L2 each); with the natural order every XCD touches all samples.

// The same contraction with its operands staged through LDS (round 4, late).  In cx_dx32_kernel every lane streams ITS OWN row of both
// operands: one load instruction touches 64 different 128-byte lines and uses 16 bytes of each, four waves per CU keep ~64 KiB of
// half-used lines alive in a 32-KiB L1, lines are evicted before their fourth use, and the four waves of a workgroup fetch the same 32
// rows of the contextual weights independently: 30 us for 7.7 us of dependent MFMAs per wave.  Here the workgroup stages, per block of
// 32 contraction columns, the weights' [32 positions][32 columns] block ONCE for its four waves and each wave's [32 channels][32 columns]
// block of centred, scaled features, with row-contiguous 128-byte global reads (8 threads per row), into k-major LDS tiles of stride 33
// (fragment reads conflict-free), double buffered through registers.  The MFMAs consume the columns in the SAME order as
// cx_dx32_kernel (k-step e of the 8-column group t = columns 8t + e | 8t + 4 + e): bit-identical results.
constexpr int kDxLd = 33;
__global__ __launch_bounds__(256) void cx_dx32s_kernel(const float* __restrict__ x, const float* __restrict__ y, int N, int C,
                                                       int hw, CxWs w, float* __restrict__ dxh) {
  __shared__ float sY[2][4][32][kDxLd];                 // [buffer][wave's channel tile][column k][channel]
  __shared__ float sD[2][32][kDxLd];                    // [buffer][column k][position]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
  const int t32 = hw / 32, ct = C / 32, cg = ct / 4 + (ct % 4 ? 1 : 0);    // 4 waves: 4 channel tiles of one position tile
  const int lb = xcd_block(N * t32 * cg);
  if (lb < 0) return;                                   // (whole block)
  const int n = lb / (t32 * cg);
  const int rem = lb - n * t32 * cg;
  const int ti = rem / cg, tc0 = (rem - ti * cg) * 4, tc = tc0 + wave;
  const int i0 = ti * 32, c0 = tc * 32;
  const float* mup = cx_mu(w, n);
  // staging role of this thread: row srow (a position of the weights block / a channel of every feature block), columns 4 scol .. + 3
  const int srow = tid >> 3, scol = tid & 7;
  const float4* dsrc = (const float4*)(w.cx + ((int64_t)n * hw + i0 + srow) * hw) + scol;
  const float4* ssrc = (const float4*)(w.iny + (int64_t)n * hw) + scol;
  const float4* ysrc[4];
  float mus[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = (tc0 + q) * 32 + srow;
    const bool ok = c < C;
    ysrc[q] = (const float4*)(y + ((int64_t)n * C + (ok ? c : 0)) * hw) + scol;
    mus[q] = ok ? mup[c] : 0.0f;
  }
  struct Raw { float4 d, s, yv[4]; };
  Raw R;
  auto gissue = [&](int jb) {                            // block jb = columns 32 jb .. 32 jb + 31 (float4 index 8 jb + scol)
    R.d = dsrc[8 * jb];
    R.s = ssrc[8 * jb];
#pragma unroll
    for (int q = 0; q < 4; ++q) R.yv[q] = ysrc[q][8 * jb];
  };
  auto sstore = [&](int buf) {
    const int k0 = 4 * scol;
    sD[buf][k0 + 0][srow] = R.d.x; sD[buf][k0 + 1][srow] = R.d.y; sD[buf][k0 + 2][srow] = R.d.z; sD[buf][k0 + 3][srow] = R.d.w;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float m = mus[q];
      sY[buf][q][k0 + 0][srow] = (R.yv[q].x - m) * R.s.x; sY[buf][q][k0 + 1][srow] = (R.yv[q].y - m) * R.s.y;
      sY[buf][q][k0 + 2][srow] = (R.yv[q].z - m) * R.s.z; sY[buf][q][k0 + 3][srow] = (R.yv[q].w - m) * R.s.w;
    }
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  const int nblk = hw / 32;
  gissue(0);
  sstore(0);
  __syncthreads();
  int buf = 0;
  for (int jb = 0; jb < nblk; ++jb) {
    const bool has_next = jb + 1 < nblk;
    if (has_next) gissue(jb + 1);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = 8 * t + 4 * kh + e;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sY[buf][wave][k][l31], sD[buf][k][l31], acc, 0, 0, 0);
      }
    if (has_next) sstore(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  if (tc >= ct) return;                                  // (after the last barrier)
  // accumulator: column = position (lane & 31), rows = channels
  const int i = i0 + l31;
  const float inx = w.inx[(int64_t)n * hw + i];
  float part = 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int c = c0 + acc_row(r, kh);
    const int64_t o = ((int64_t)n * C + c) * hw + i;
    dxh[o] = acc[r];
    part = fmaf((x[o] - mup[c]) * inx, acc[r], part);
  }
  part += __shfl_xor(part, 32, 64);
  if (kh == 0) w.dot[(int64_t)tc * N * hw + (int64_t)n * hw + i] = part;        // one partial per channel tile, summed by cx_dx_finish
}

// Row pass, block-parallel: 16 waves x 1 row = 16 consecutive rows of one sample per block (all rows of the matrix are
// in flight at once: the pass is one row's latency); the column maxima of the 16 waves meet in LDS -> one atomicMax
// per column and block.
constexpr int kCxRowsPerWave = 1;
constexpr int kCxRowWaves = 16;
__global__ __launch_bounds__(64 * kCxRowWaves) void cx_rows_fwd32_kernel(int N, int hw, float inv_h, CxWs w, float scale, int with_loss) {
  extern __shared__ float scm[];                         // [16 waves][hw]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int blocks_per_n = (hw + kCxRowWaves * kCxRowsPerWave - 1) / (kCxRowWaves * kCxRowsPerWave);
  const int n = blockIdx.x / blocks_per_n;
  const int r0 = (blockIdx.x - n * blocks_per_n) * kCxRowWaves * kCxRowsPerWave + wave * kCxRowsPerWave;
  float cmaxv[kCxMaxCols];
#pragma unroll
  for (int q = 0; q < kCxMaxCols; ++q) cmaxv[q] = 0.0f;
  const int ncol = (hw + 63) / 64;                       // <= kCxMaxCols
  for (int rr = 0; rr < kCxRowsPerWave; ++rr) {
    if (r0 + rr >= hw) break;
    const int64_t row = (int64_t)n * hw + r0 + rr;
    const float* Dr = w.D + row * hw;
    float* cr = w.cx + row * hw;
    float wv[kCxMaxCols];
    float s = 0.0f;
    float dm;
    if (w.min_in_rows) {                                 // (uniform) the row is in this wave's registers: its minimum is 6 lane exchanges
      float mn = 2.0f;                                   // (D <= 1)
#pragma unroll
      for (int q = 0; q < kCxMaxCols; ++q) {
        if (q < ncol) {
          const int j = lane + 64 * q;
          wv[q] = j < hw ? Dr[j] : 2.0f;
          mn = fminf(mn, wv[q]);
        }
      }
      for (int off = 32; off > 0; off >>= 1) mn = fminf(mn, __shfl_xor(mn, off, 64));
      if (lane == 0) w.dmin[row] = __float_as_uint(mn);  // the backward row pass reads it
      dm = mn + 1e-5f;
#pragma unroll
      for (int q = 0; q < kCxMaxCols; ++q) {
        if (q < ncol) {
          const int j = lane + 64 * q;
          wv[q] = j < hw ? __expf((1.0f - wv[q] / dm) * inv_h) : 0.0f;
          s += wv[q];
        }
      }
    } else {
      dm = __uint_as_float(w.dmin[row]) + 1e-5f;
#pragma unroll
      for (int q = 0; q < kCxMaxCols; ++q) {
        if (q < ncol) {
          const int j = lane + 64 * q;
          wv[q] = j < hw ? __expf((1.0f - Dr[j] / dm) * inv_h) : 0.0f;
          s += wv[q];
        }
      }
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) w.s[row] = s;
    const float inv = 1.0f / s;
#pragma unroll
    for (int q = 0; q < kCxMaxCols; ++q) {
      if (q < ncol) {
        const int j = lane + 64 * q;
        if (j < hw) {
          const float c = wv[q] * inv;
          cr[j] = c;
          cmaxv[q] = fmaxf(cmaxv[q], c);
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < kCxMaxCols; ++q) {
    if (q < ncol) {
      const int j = lane + 64 * q;
      if (j < hw) scm[wave * hw + j] = cmaxv[q];
    }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < hw; j += 64 * kCxRowWaves) {
    float m = scm[j];
#pragma unroll
    for (int v = 1; v < kCxRowWaves; ++v) m = fmaxf(m, scm[v * hw + j]);
    atomicMax(&w.cmax[(int64_t)n * hw + j], __float_as_uint(m));
  }
  if (!with_loss) return;
  // Round 4: what cx_loss_kernel did in a launch of its own (5 us of ramp), done by the block of the sample that finishes last:
  // the sample's column maxima are complete then -> cxn, its loss term, dL/dcxn, and the inverse norms of its positions.
  if (!block_last_arriver(w.ticket + 1 + n, blocks_per_n)) return;
  float acc = 0.0f;
  for (int j = threadIdx.x; j < hw; j += 64 * kCxRowWaves) {
    acc += __uint_as_float(__hip_atomic_load(&w.cmax[(int64_t)n * hw + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    w.inx[(int64_t)n * hw + j] = inv_norm(w.ssx[(int64_t)n * hw + j]);
    w.iny[(int64_t)n * hw + j] = inv_norm(w.ssy[(int64_t)n * hw + j]);
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  __syncthreads();
  if (lane == 0) scm[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.0f;
#pragma unroll
    for (int v = 0; v < kCxRowWaves; ++v) tot += scm[v];
    const float cxn = tot / (float)hw;
    int ng;
    cx_group(w, n, ng);
    w.lsum[n] = scale * (-logf(cxn + 1e-5f) / (float)ng);        // (mean over the group's samples; the weighted form keeps cx_loss_kernel)
    w.g[n] = scale * (-1.0f / ((float)ng * (cxn + 1e-5f))) / (float)hw;
  }
}

}  // namespace npp

using namespace npp;

extern "C" int npp_patch_gather(const float* d_img_hwc, const float* d_mask_hw, int H, int W, const int32_t* d_centres_yx,
                                int M, int P, float* d_out_rgb, float* d_out_mask, void* stream) {
  if (!d_img_hwc || !d_centres_yx || !d_out_rgb || H < 1 || W < 1 || M < 0 || P < 2 || (P & 1)) {
    set_error("npp_patch_gather: bad arguments (M=%d, P=%d must be even)", M, P);
    return NPP_ERR_ARG;
  }
  if (d_out_mask && !d_mask_hw) { set_error("npp_patch_gather: out_mask without mask"); return NPP_ERR_ARG; }
  if (M == 0) return NPP_OK;
  const int bx = (P * P + 255) / 256;
  hipLaunchKernelGGL(patch_gather_kernel, dim3(bx, M), dim3(256), 0, (hipStream_t)stream, d_img_hwc, d_mask_hw, H, W,
                     d_centres_yx, M, P, d_out_rgb, d_out_mask);
  return check_launch("npp_patch_gather");
}

extern "C" int64_t npp_cx_workspace_bytes(int N, int C, int hw) {
  if (N < 1 || C < 32 || (C % 32) || hw < 1) { set_error("npp_cx_workspace_bytes: need N>=1, C multiple of 32, hw>=1"); return -1; }
  return 4 * cx_ws_floats(N, C, hw);
}

struct CxFlatOut {        // optional: dL/dx * [y > 0] into the trunk's flat bf16 gradient tensor instead of fp32 d_dfx
  const void* yact;       // flat fp16 activations of the tapped layer (N_total images of H x W, C channels)
  void* dz;               // flat bf16 gradient tensor, same geometry
  int N_total, H, W;
};
// the value-only big form's arrays (tiled matrix, re-tiled operands, row-sum quarters) inside the 2 N hw^2 floats of D + cx
static bool big_fwd_fits(int N, int C, int hw) {
  const int64_t t = (hw + 127) / 128;
  return (int64_t)N * (t * t * 16384 + 2 * t * C * 128 + (int64_t)hw * kBigQ) <= 2LL * N * hw * hw;
}
static int cx_launch(const float* d_fx, const float* d_fy, int N, int C, int hw, float band_width, const float* d_weight,
                     float scale, float* d_loss, int loss_stride, float* d_dfx, void* d_workspace, int64_t workspace_bytes,
                     const void* d_iter, int M, void* stream, const char* who, const CxFlatOut* flat = nullptr) {
  if (!d_fx || !d_fy || !d_loss || !d_workspace || N < 1 || C < 32 || (C % 32) || hw < 1 || !(band_width > 0.0f) ||
      (int64_t)N * hw * hw > 0x7fffffffLL * 8 || M < 0 || M > NPP_MAX_STACK || (M > 0 && !d_iter)) {
    set_error("%s: bad arguments (N=%d C=%d hw=%d M=%d)", who, N, C, hw, M);
    return NPP_ERR_ARG;
  }
  if (workspace_bytes < 4 * cx_ws_floats(N, C, hw)) { set_error("%s: workspace too small", who); return NPP_ERR_ARG; }
  hipStream_t s = (hipStream_t)stream;
  CxWs w = carve((float*)d_workspace, N, C, hw, d_iter, M);
  const int groups = M > 0 ? M : 1;
  const int64_t nh = (int64_t)N * hw;
  const float inv_h = 1.0f / band_width;
  const bool big = hw > 64 * kCxMaxCols;     // whole-image crops: the generic kernels, column-chunked row pass
  const bool fast = !big && (hw % 32) == 0;  // LDS-free contractions + block-parallel row pass (all loop sizes: hw = (P/4)^2)
  w.min_in_rows = fast ? 1 : 0;
  hipLaunchKernelGGL(cx_mean_kernel, dim3(C, groups), dim3(256), 0, s, d_fy, N, C, hw, w);
  const int tiles = (hw + 63) / 64;
  const bool loss_in_rows = fast && d_dfx && !d_weight;      // the loss terms ride in the row pass, their group sums in the backward row pass
  const int t32 = hw / 32;
  if (fast) {
    hipLaunchKernelGGL(cx_sim_kernel, dim3((unsigned)(((int64_t)N * tiles * tiles + 7) / 8 * 8)), dim3(256), 0, s, d_fx, d_fy, N, C, hw, w);
    static SmemOnce once;
    const size_t smem = (size_t)kCxRowWaves * hw * sizeof(float);
    if (smem > 48 * 1024 && !smem_attr(once, (const void*)cx_rows_fwd32_kernel, kCxRowWaves * 64 * kCxMaxCols * 4)) {
      set_error("%s: smem attribute", who); return NPP_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(cx_rows_fwd32_kernel, dim3((unsigned)((int64_t)N * ((hw + kCxRowWaves - 1) / kCxRowWaves))), dim3(64 * kCxRowWaves), smem, s, N,
                       hw, inv_h, w, scale, (int)loss_in_rows);
  } else if (big && !d_dfx && big_fwd_fits(N, C, hw)) {
    // whole-image crop, value only (the ranking's score): re-tiled normalised operands, 128 x 128 tiles in super-block order, the
    // matrix stored tiled, written once and read twice.  Everything lives in the D + cx regions (2 N hw^2 floats).
    const int t128 = (hw + 127) / 128;
    const int64_t t8 = (t128 + 7) / 8;
    if ((int64_t)N * t8 * t8 * 64 > 0x7fffffffLL) { set_error("%s: grid too large", who); return NPP_ERR_ARG; }
    float* xt = w.D + (int64_t)N * t128 * t128 * 16384;
    float* yt = xt + (int64_t)N * t128 * C * 128;
    float* s4 = yt + (int64_t)N * t128 * C * 128;
    hipLaunchKernelGGL(cx_prep_big_kernel, dim3((unsigned)(2 * N * t128)), dim3(256), 0, s, d_fx, d_fy, N, C, hw, w, xt, yt);
    hipLaunchKernelGGL(cx_sim_big_kernel, dim3((unsigned)((int64_t)N * t8 * t8 * 64)), dim3(256), 0, s, xt, yt, N, C, hw, w);
    hipLaunchKernelGGL(cx_rowsum_big_kernel, dim3((unsigned)(N * t128 * kBigQ)), dim3(256), 0, s, N, hw, inv_h, w, s4);
    hipLaunchKernelGGL(cx_colmax_big_kernel, dim3((unsigned)(N * t128 * kBigR)), dim3(256), 0, s, N, hw, inv_h, w, s4);
  } else {
    hipLaunchKernelGGL(cx_sim_kernel, dim3((unsigned)(((int64_t)N * tiles * tiles + 7) / 8 * 8)), dim3(256), 0, s, d_fx, d_fy, N, C, hw, w);
    const int64_t row_groups = (int64_t)N * ((hw + kCxRows - 1) / kCxRows);
    if (big) hipLaunchKernelGGL(cx_rows_fwd_big_kernel, dim3((unsigned)((row_groups + 3) / 4)), dim3(256), 0, s, N, hw, inv_h, w);
    else hipLaunchKernelGGL(cx_rows_fwd_kernel, dim3((unsigned)((row_groups + 3) / 4)), dim3(256), 0, s, N, hw, inv_h, w);
  }
  if (!loss_in_rows) hipLaunchKernelGGL(cx_loss_kernel, dim3(N), dim3(256), 0, s, N, hw, d_weight, scale, d_loss, loss_stride, w);
  if (d_dfx) {
    hipLaunchKernelGGL(cx_rows_bwd_kernel, dim3((unsigned)((nh + 3) / 4)), dim3(256), 0, s, N, hw, inv_h, w, loss_in_rows ? d_loss : nullptr,
                       loss_stride);
    const int ctiles = (C + 63) / 64;
    if (fast) {
      const int64_t nb_dx = (int64_t)N * t32 * ((C / 32 + 3) / 4);
      // (measured and removed, profiles/r04_cx_dx_staged_ab.txt: the row-streaming form of this contraction -- 30.9 us against 20.1 --
      //  and a five-launch form with the finish fused behind eight-wave workgroups: +8 us on the chain)
      hipLaunchKernelGGL(cx_dx32s_kernel, dim3((unsigned)((nb_dx + 7) / 8 * 8)), dim3(256), 0, s, d_fx, d_fy, N, C, hw, w, d_dfx);
    }
    else
      hipLaunchKernelGGL(cx_dx_kernel, dim3((unsigned)((int64_t)N * ctiles * tiles)), dim3(256), 0, s, d_fx, d_fy, N, C, hw, w,
                         d_dfx);
    if (flat) {
      if (!flat->yact || !flat->dz || flat->H * flat->W != hw || N > flat->N_total || C % 16) {
        set_error("%s: bad flat-output description", who); return NPP_ERR_ARG;
      }
      const int64_t range = conv_npos_round(N, flat->H, flat->W), nt = range * (C / 8);
      hipLaunchKernelGGL(cx_dx_finish_flat_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, s, d_fx, d_dfx, N, C, flat->H, flat->W,
                         w, (const f16x8*)flat->yact, (bf16x8*)flat->dz, conv_nposp(flat->N_total, flat->H, flat->W), range, C / 32);
      return check_launch(who);
    }
    const int64_t ne = nh * C;
    hipLaunchKernelGGL(cx_dx_finish_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, s, d_fx, N, C, hw, w, d_dfx, C / 32);
  }
  return check_launch(who);
}

extern "C" int npp_cx_fwd_bwd(const float* d_fx, const float* d_fy, int N, int C, int hw, float band_width,
                              const float* d_weight, float scale, float* d_loss, float* d_dfx, void* d_workspace,
                              int64_t workspace_bytes, void* stream) {
  return cx_launch(d_fx, d_fy, N, C, hw, band_width, d_weight, scale, d_loss, 0, d_dfx, d_workspace, workspace_bytes, nullptr, 0,
                   stream, "npp_cx_fwd_bwd");
}

// The same over sample GROUPS (stacked launches: one group per image, M groups): group m = the iter[m].nk samples from index
// iter[m].x0 on, in BOTH tensors (d_fy = the real halves in the same order); the mean over y, the 1 / N of the loss and the
// loss accumulator (d_loss[m * loss_stride]) are per group; samples of different groups never meet.
extern "C" int npp_cx_fwd_bwd_groups(const float* d_fx, const float* d_fy, int N, int C, int hw, float band_width, float scale,
                                     float* d_loss, int loss_stride, float* d_dfx, const void* d_iter, int M, void* d_workspace,
                                     int64_t workspace_bytes, void* stream) {
  if (M < 1) { set_error("npp_cx_fwd_bwd_groups: M=%d", M); return NPP_ERR_ARG; }
  return cx_launch(d_fx, d_fy, N, C, hw, band_width, nullptr, scale, d_loss, loss_stride, d_dfx, d_workspace, workspace_bytes, d_iter,
                   M, stream, "npp_cx_fwd_bwd_groups");
}

// npp_cx_fwd_bwd / npp_cx_fwd_bwd_groups with the gradient delivered where the trunk's data-gradient pass reads it: the flat bf16
// tensor d_dz (geometry: N_total images of H x W, C channels; npp_trunk_act_bytes), already gated by the ReLU of the tapped layer
// (d_yact = that layer's flat fp16 output) -- replaces the last launch of the core + npp_trunk_grad_in.  d_scratch_dfx (N, C, H W)
// fp32 receives the un-normalised contraction (workspace of the call).  M = 0 / d_iter = NULL: one group of N samples.
extern "C" int npp_cx_fwd_bwd_flat(const float* d_fx, const float* d_fy, int N, int C, int H, int W, float band_width, float scale,
                                   float* d_loss, int loss_stride, float* d_scratch_dfx, const void* d_yact, void* d_dz, int N_total,
                                   const void* d_iter, int M, void* d_workspace, int64_t workspace_bytes, void* stream) {
  if (!d_scratch_dfx) { set_error("npp_cx_fwd_bwd_flat: null scratch"); return NPP_ERR_ARG; }
  const CxFlatOut flat{d_yact, d_dz, N_total, H, W};
  return cx_launch(d_fx, d_fy, N, C, H * W, band_width, nullptr, scale, d_loss, loss_stride, d_scratch_dfx, d_workspace, workspace_bytes,
                   d_iter, M, stream, "npp_cx_fwd_bwd_flat", &flat);
}
