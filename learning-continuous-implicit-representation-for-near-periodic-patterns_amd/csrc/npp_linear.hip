// npp_linear.hip -- generic dense layers in exact fp32 (v_mfma_f32_32x32x2_f32): what F.linear + SnakeActivation and
// their autograd do for topologies the fused chain kernels are not specialised for.  First user: the proposal-ranking
// fits of NPP_proposal/search.py:85-205 with NPP_Net_light (models/networks.py:176-263): 2048 rows x ~0.3 M parameters x
// 300 iterations per candidate -- a launch-bound problem for ONE candidate (128 workgroups per layer); the candidates of an
// image are independent and share their pixel rows, so the *_batched entry points carry all of them in one launch (the candidate
// is blockIdx.z): 9 x 2048 x 256 x 256 in 37 us = 0.41 of the fp32 MFMA peak, T(C) ~ 12 us + 2.9 us per candidate (MFMA floor 1.7).
//
// One GEMM kernel, C[m][n] = sum_k A(m,k) B(k,n) with arbitrary element strides, serves
//   forward      y  = act(x W^T + b)    A = x  (k contiguous), B = W read as (k,n) (k contiguous)
//   data grad    dx = dz W              A = dz (k contiguous), B = W read as (k=n_out, col) (col contiguous)
//   weight grad  dW = dz^T x            A = dz read as (m=n_out, k=row) (m contiguous), B = x (col contiguous)
// 64 x 64 output tile per workgroup (4 waves of 32 x 32), 32-wide k chunks staged through LDS (k-pairs interleaved, see
// kGemmLdp): the fp32 MFMA takes ONE float per lane and operand, so the operands cannot come straight from global memory
// (measured on the contextual-loss kernels: texture-addresser-bound).  The staging thread map follows each operand's
// contiguous index, 16 bytes per lane where rows are 16-byte aligned.  Measured with the phases switched off one at a time
// (tools/r3_gemm_probe.py history, 9 stacked problems): launch + epilogue 10 us, + staging 20, + MFMAs and their LDS reads 30,
// all 37 -- LDS traffic (stores ~190 + reads ~250-500 array cycles per workgroup-chunk) and the MFMA pipe (1024 cycles per
// wave-chunk) add up rather than overlap; a 64 x 64 tile per WAVE with the contraction split over the waves would halve the
// LDS reads at the price of a cross-wave reduction (not built: estimated 37 -> 31 us).
#include "npp_common.h"
#include "npp_light_layout.h"

namespace npp {

struct GemmArgs {
  const float* A; int64_t sam, sak;
  const float* B; int64_t sbk, sbn;
  float* C; int64_t ldc;
  float* Z; int64_t ldz;          // optional copy of the pre-activation (forward)
  const float* bias;              // optional, per column n
  float* rowsum;                  // optional: rowsum[m] += sum_k A(m,k) (the bias gradient of the weight-gradient form)
  int M, N, K, act, accumulate;   // act: 0 none, 1 snake (x + sin^2 x)
  int kchunk;                     // k range per blockIdx.z (split-K: partial sums meet in C by atomicAdd, C pre-zeroed)
  int nbatch;                     // > 1: independent problems in one launch, blockIdx.z = batch * splits + split; element strides
  int64_t sab, sbb, scb;          //      between the batches' A / B / C ...
  int64_t szb, sbiasb, srsb;      //      ... and Z / bias / rowsum
  int splits;                     // k ranges per problem (>= 1)
  const float* dact; int64_t lddact, sdactb; int dact_kind;   // optional: C = (A B) * act'(dact[m][n]) (kinds of act_bwd_kernel)
  int b_snake;                    // B operand is stored as pre-activations: snake() them on their way into LDS (fused-chain stashes)
  int vec_a, vec_b;               // set by the launcher: the operand's runs of 4 may be fetched as one 16-byte load
  // deterministic split (npp_light_wgrad_det): the splits of an output tile leave their partial tiles in `slab`
  // ([batch][tile][split][16][256] floats, then the bias partials [batch][tile row][split][64]); the workgroup that arrives LAST at the
  // tile's ticket adds them in split order and is the only writer of C / rowsum -- no float atomics, the same bits every run
  float* slab; float* rs_slab; unsigned* ticket;
};

// act'(.) : kind 1 snake from the stashed pre-activation z (1 + sin 2z); 2 sigmoid from its output y (y (1 - y)); 3 tanh from its
// output (1 - y^2); 4 relu from z ([z > 0])
__device__ __forceinline__ float act_deriv(float v, int kind) {
  return kind == 1 ? 1.0f + sinf(2.0f * v) : (kind == 2 ? v * (1.0f - v) : (kind == 3 ? 1.0f - v * v : (kind == 4 ? (v > 0.0f ? 1.0f : 0.0f) : 1.0f)));
}

// Operand tile in LDS: k-pairs interleaved, s[k >> 1][m][k & 1] with 66 pairs per row.  The fp32 MFMA takes A[m = lane & 31][k = lane >> 5]
// from each lane, so the 64 lanes of a fragment read are 64 consecutive floats (conflict-free); the staging stores are conflict-free
// in both thread maps below because 66 = 2 (mod 16).
constexpr int kGemmLdp = 66;
constexpr int kGemmTile = 16 * kGemmLdp * 2;          // floats of one 64 x 32 operand tile

// One operand's share of a chunk per thread: 8 floats = two runs of 4 along the operand's contiguous index.
//   KC  (k contiguous):  run r: m = (tid >> 3) + 32 r, k = 4 (tid & 7) .. + 3      -> two ds_write_b64 (two k-pairs of one m)
//   !KC (m contiguous):  run r: k = (tid >> 4) + 16 r, m = 4 (tid & 15) .. + 3     -> four ds_write_b32
// vec: the run is one 16-byte load (host checked alignment and that runs are all-valid or all-invalid); else four 4-byte loads.
// Out-of-range elements are fetched from a clamped address and zeroed when the registers are STORED to LDS (gemm_sstore), a chunk
// later: a select right behind the load would wait for it there and take the prefetch distance away.
template <bool KC>
__device__ __forceinline__ void gemm_gload(const float* __restrict__ P, int64_t s_m, int64_t s_k, int m0, int M, int k0, int kend, bool vec,
                                           int tid, float (&reg)[8]) {
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    int m, k;
    if (KC) { m = (tid >> 3) + 32 * r; k = 4 * (tid & 7); } else { k = (tid >> 4) + 16 * r; m = 4 * (tid & 15); }
    const int gm = m0 + m, gk = k0 + k;
    if (vec) {
      const f32x4 v = *(const f32x4*)(P + (int64_t)min(gm, M - (KC ? 1 : 4)) * s_m + (int64_t)min(gk, kend - (KC ? 4 : 1)) * s_k);
#pragma unroll
      for (int e = 0; e < 4; ++e) reg[4 * r + e] = v[e];
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int em = KC ? gm : gm + e, ek = KC ? gk + e : gk;
        reg[4 * r + e] = P[(int64_t)min(em, M - 1) * s_m + (int64_t)min(ek, kend - 1) * s_k];
      }
    }
  }
}
template <bool KC>
__device__ __forceinline__ void gemm_sstore(float* __restrict__ S, int tid, const float (&reg)[8], int m0, int M, int k0, int kend, bool snake = false) {
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    float v[4];
    if (KC) {
      const int m = (tid >> 3) + 32 * r, kp = 2 * (tid & 7);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x = snake ? snake_fast(reg[4 * r + e]) : reg[4 * r + e];
        v[e] = (m0 + m < M && k0 + 2 * kp + e < kend) ? x : 0.0f;
      }
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      *(f32x2*)(S + ((kp + 0) * kGemmLdp + m) * 2) = f32x2{v[0], v[1]};
      *(f32x2*)(S + ((kp + 1) * kGemmLdp + m) * 2) = f32x2{v[2], v[3]};
    } else {
      const int k = (tid >> 4) + 16 * r, m = 4 * (tid & 15);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x = snake ? snake_fast(reg[4 * r + e]) : reg[4 * r + e];
        S[((k >> 1) * kGemmLdp + m + e) * 2 + (k & 1)] = (m0 + m + e < M && k0 + k < kend) ? x : 0.0f;
      }
    }
  }
}

#ifndef NPP_GEMM_WAVES
#define NPP_GEMM_WAVES 5
#endif
template <bool A_KC, bool B_KC>
__device__ __forceinline__ void gemm32_body(GemmArgs g, const int bx, const int by, const int bzz) {
  // Occupancy is the point of the launch bound: the candidate-stacked layers of the ranking fit are 1152 workgroups; at 4 resident
  // per CU (1024 slots) the last 128 run as a second, 12 %-full round and the launch takes two workgroup times (measured 38 us against
  // a 15 us MFMA floor); 5 per CU hold them all.  One LDS buffer (17 KB) for the same reason.
  __shared__ __attribute__((aligned(16))) float sA[kGemmTile];
  __shared__ __attribute__((aligned(16))) float sB[kGemmTile];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
  const int m0 = by * 64, n0 = bx * 64, wm = wave >> 1, wn = wave & 1;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  const int bz = bzz / g.splits, sp = bzz - bz * g.splits;
  if (g.nbatch > 1) {
    g.A += bz * g.sab; g.B += bz * g.sbb; g.C += bz * g.scb;
    if (g.Z) g.Z += bz * g.szb;
    if (g.bias) g.bias += bz * g.sbiasb;
    if (g.rowsum) g.rowsum += bz * g.srsb;
    if (g.dact) g.dact += bz * g.sdactb;
  }
  const int kbeg = sp * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
  const bool split = g.splits > 1;
  const bool va = g.vec_a != 0, vb = g.vec_b != 0;
  const bool do_rowsum = g.rowsum && bx == 0 && tid < 64;
  float rs = 0.0f;
  // chunk c of 32 k: registers <- global while the MFMAs of chunk c - 1 run, LDS <- registers between two barriers.  Branch-free:
  // out-of-range elements read a clamped address and are zeroed on their way into LDS.
  float ra[8], rb[8];
  const int nchunk = (kend - kbeg + 31) / 32;
  if (nchunk > 0) {
    gemm_gload<A_KC>(g.A, g.sam, g.sak, m0, g.M, kbeg, kend, va, tid, ra);
    gemm_gload<B_KC>(g.B, g.sbn, g.sbk, n0, g.N, kbeg, kend, vb, tid, rb);
  }
  const float* __restrict__ a = sA + (wm * 32 + l31) * 2 + kh;
  const float* __restrict__ b = sB + (wn * 32 + l31) * 2 + kh;
  for (int c = 0; c < nchunk; ++c) {
    gemm_sstore<A_KC>(sA, tid, ra, m0, g.M, kbeg + 32 * c, kend);
    gemm_sstore<B_KC>(sB, tid, rb, n0, g.N, kbeg + 32 * c, kend, g.b_snake != 0);
    wg_barrier();
    if (c + 1 < nchunk) {
      gemm_gload<A_KC>(g.A, g.sam, g.sak, m0, g.M, kbeg + 32 * (c + 1), kend, va, tid, ra);
      gemm_gload<B_KC>(g.B, g.sbn, g.sbk, n0, g.N, kbeg + 32 * (c + 1), kend, vb, tid, rb);
    }
    if (do_rowsum) {
      const float* q = sA + tid * 2;
#pragma unroll
      for (int kp = 0; kp < 16; ++kp) rs += q[kp * kGemmLdp * 2] + q[kp * kGemmLdp * 2 + 1];
    }
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ks * kGemmLdp * 2], b[ks * kGemmLdp * 2], acc, 0, 0, 0);
    wg_barrier();
  }
  bool ordered = false;
  if (g.slab && split) {                                // (uniform over the launch's problem)
    const int gx = (g.N + 63) >> 6, gy = (g.M + 63) >> 6, tile = by * gx + bx;
    float* sl = g.slab + (((int64_t)bz * gx * gy + tile) * g.splits + sp) * 4096;
#pragma unroll
    for (int r = 0; r < 16; ++r) share_store(sl + r * 256 + tid, acc[r]);
    float* rsl = g.rs_slab + ((int64_t)bz * gy + by) * g.splits * 64;
    if (do_rowsum) share_store(rsl + sp * 64 + tid, rs);
    if (!g.ticket) return;                                // two-pass form: gemm_slab_reduce_kernel adds the ranges in a launch of its own
    if (!block_last_arriver(g.ticket + (int64_t)bz * gx * gy + tile, g.splits)) return;
    sl -= (int64_t)sp * 4096;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = 0.0f;
      for (int q = 0; q < g.splits; ++q) v += share_load(sl + (int64_t)q * 4096 + r * 256 + tid);
      acc[r] = v;
    }
    if (do_rowsum) {
      rs = 0.0f;
      for (int q = 0; q < g.splits; ++q) rs += share_load(rsl + q * 64 + tid);
    }
    ordered = true;
  }
  if (do_rowsum && m0 + tid < g.M) {
    if (ordered) g.rowsum[m0 + tid] += rs;              // one writer per element
    else atomicAdd(g.rowsum + m0 + tid, rs);
  }
  const int n = n0 + wn * 32 + l31;
  if (n >= g.N) return;
  const float bv = (g.bias && sp == 0) ? g.bias[n] : 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm * 32 + acc_row(r, kh);
    if (m >= g.M) continue;
    float v = acc[r] + bv;
    if (split && !ordered) { atomicAdd(g.C + (int64_t)m * g.ldc + n, v); continue; }      // linear outputs only (launcher guarantees)
    if (g.Z) g.Z[(int64_t)m * g.ldz + n] = v;
    if (g.dact) v *= act_deriv(g.dact[(int64_t)m * g.lddact + n], g.dact_kind);
    if (g.act == 1) { const float s = sinf(v); v = fmaf(s, s, v); }      // activations.py:29-35, a = 1
    else if (g.act == 2) v = fmaxf(v, 0.0f);                             // F.relu (networks.py:66-67, activation='relu')
    float* c = g.C + (int64_t)m * g.ldc + n;
    *c = g.accumulate ? *c + v : v;
  }
}

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256, NPP_GEMM_WAVES) void gemm32_kernel(GemmArgs g) {
  gemm32_body<A_KC, B_KC>(g, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
}

// Several problems of one operand form in ONE launch (a 1-D grid over all their workgroups): the seven weight gradients of the
// stacked NPP_Net_light candidates (npp_light_wgrad) -- each of them alone is a launch whose fixed ~12 us ramp is a third of its time.
constexpr int kGemmGroupMax = 8;
struct GemmGroup {
  GemmArgs g[kGemmGroupMax];
  int32_t first_wg[kGemmGroupMax + 1], gx[kGemmGroupMax], gy[kGemmGroupMax];
  int32_t n;
};
template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256, NPP_GEMM_WAVES) void gemm32_grouped_kernel(GemmGroup G) {
  int p = 0;
  for (int q = 1; q < G.n; ++q) if ((int)blockIdx.x >= G.first_wg[q]) p = q;
  const int local = (int)blockIdx.x - G.first_wg[p];
  const int bx = local % G.gx[p], rest = local / G.gx[p];
  gemm32_body<A_KC, B_KC>(G.g[p], bx, rest % G.gy[p], rest / G.gy[p]);
}

// Second pass of the two-pass ordered split (npp_gram_fwd_det): one block per (batch, output tile) adds the ranges' partial tiles in range
// order and writes C (no bias sums, no activation: plain linear outputs).  The ranges were written by the PREVIOUS launch: what makes
// them visible is the launch boundary, not an in-launch handshake -- the one-launch form (last arriver reads the others' tiles) gave
// results that moved in the last bits when the style branch of stacked images ran beside the contextual chain on a second stream.
__global__ __launch_bounds__(256) void gemm_slab_reduce_kernel(GemmArgs g, int gx, int gy) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, kh = lane >> 5, wm = wave >> 1, wn = wave & 1;
  const int tile = blockIdx.x % (gx * gy), bz = blockIdx.x / (gx * gy);
  const int by = tile / gx, bx = tile - by * gx;
  const float* sl = g.slab + ((int64_t)bz * gx * gy + tile) * g.splits * 4096;
  float* C = g.C + (g.nbatch > 1 ? bz * g.scb : 0);
  const int n = bx * 64 + wn * 32 + l31;
  if (n >= g.N) return;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = by * 64 + wm * 32 + acc_row(r, kh);
    if (m >= g.M) continue;
    float v = 0.0f;
    for (int q = 0; q < g.splits; ++q) v += sl[(int64_t)q * 4096 + r * 256 + tid];
    float* c = C + (int64_t)m * g.ldc + n;
    *c = g.accumulate ? *c + v : v;
  }
}

// dz = dy * act'(.) : act 1 snake from the stashed pre-activation z (1 + sin 2z); 2 sigmoid from its output y (y (1 - y));
// 3 tanh from its output (1 - y^2); 4 relu from z ([z > 0])
__global__ void act_bwd_kernel(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ zy, int64_t ldzy, int64_t B, int N,
                               int act, float* __restrict__ dz, int64_t lddz) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * N) return;
  const int64_t r = t / N;
  const int n = (int)(t - r * N);
  dz[r * lddz + n] = dy[r * lddy + n] * act_deriv(zy[r * ldzy + n], act);
}

// y = sigmoid(x) / tanh(x) elementwise (render's output squash, helpers.py:55-58)
__global__ void act_fwd_kernel(const float* __restrict__ x, int64_t n, int act, float* __restrict__ y) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const float v = x[t];
  y[t] = act == 2 ? 1.0f / (1.0f + expf(-v)) : (act == 3 ? tanhf(v) : v);
}

// LPIPS.forward(use_robust=False), one tap (lpips.py:99-101,110,117,130): sum over (n, pos) of
// sum_c lin_c (f0_c / (|f0| + eps) - f1_c / (|f1| + eps))^2, scaled by coef.  16 positions x 16 channel lanes per block.
// scratch (nullable): [gridDim.x partial sums | arrival counter], zero before the first launch that uses it -- the block that arrives
// last adds the partials in block order (bit-reproducible score); null: one float atomicAdd per block
__global__ __launch_bounds__(256) void lpips_plain_kernel(const float* __restrict__ f0, const float* __restrict__ f1, int N, int C, int hw,
                                                          const float* __restrict__ lin, float coef, float* __restrict__ out, float* scratch) {
  __shared__ float red[2][16][17];
  __shared__ float tot[4];
  const int pl = threadIdx.x & 15, cl = threadIdx.x >> 4;
  const int64_t npos = (int64_t)N * hw, ngroups = (npos + 15) / 16;
  float val = 0.0f;
  for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int64_t t = grp * 16 + pl;
    const bool live = t < npos;
    const int n = live ? (int)(t / hw) : 0, p = live ? (int)(t - (int64_t)n * hw) : 0;
    const float* a0 = f0 + (int64_t)n * C * hw + p;
    const float* a1 = f1 + (int64_t)n * C * hw + p;
    float s0 = 0.0f, s1 = 0.0f;
    if (live)
      for (int c = cl; c < C; c += 16) {
        const float u = a0[(int64_t)c * hw], v = a1[(int64_t)c * hw];
        s0 = fmaf(u, u, s0);
        s1 = fmaf(v, v, s1);
      }
    __syncthreads();
    red[0][cl][pl] = s0;
    red[1][cl][pl] = s1;
    __syncthreads();
    s0 = 0.0f; s1 = 0.0f;
#pragma unroll
    for (int q = 0; q < 16; ++q) { s0 += red[0][q][pl]; s1 += red[1][q][pl]; }
    const float i0 = 1.0f / (sqrtf(s0) + 1e-10f), i1 = 1.0f / (sqrtf(s1) + 1e-10f);
    if (live)
      for (int c = cl; c < C; c += 16) {
        const float d = a0[(int64_t)c * hw] * i0 - a1[(int64_t)c * hw] * i1;
        val = fmaf(lin[c] * d, d, val);
      }
  }
  for (int off = 32; off > 0; off >>= 1) val += __shfl_xor(val, off, 64);
  if ((threadIdx.x & 63) == 0) tot[threadIdx.x >> 6] = val;
  __syncthreads();
  if (!scratch) {
    if (threadIdx.x == 0) atomicAdd(out, coef * (tot[0] + tot[1] + tot[2] + tot[3]));
    return;
  }
  if (threadIdx.x == 0) share_store(scratch + blockIdx.x, coef * (tot[0] + tot[1] + tot[2] + tot[3]));
  if (block_last_arriver((unsigned*)(scratch + gridDim.x), (int)gridDim.x) && threadIdx.x == 0) {
    float s = 0.0f;
    for (unsigned b = 0; b < gridDim.x; ++b) s += share_load(scratch + b);
    *out += s;                                   // (the taps of a score are consecutive launches on one stream: one writer at a time)
  }
}

// Per-element adaptive robust NLL (robust_loss_pytorch/adaptive.py:183-204 with num_dims = D latents): one thread per
// element index j walks the N samples -- models/style_loss.py:60-69 applies it to the N x C^2 differences of two Gram
// matrices.  loss += sum_n coef_n sum_j nll(d[n][j]); dd[n][j] = coef_n dnll/dx; dlatent[j] / [D + j] += the latent gradients.
struct ElemCoef { float v[64]; };            // the N <= 64 per-sample factors, by value (kernel argument)
// (round 6) ONE launch: the difference a - b and the element's ChanParams are formed by the thread that uses them.  They were two
// launches of their own in front of this one (sub_kernel, elem_chan_kernel -> a ChanParams table in the workspace): three launches
// and two round trips through memory per Gram level for 2 N D subtractions and D table rows.  (It does NOT settle the run-to-run
// drift of the style latents' gradients on a side stream, which is what prompted it: DESIGN section 4.)
__global__ __launch_bounds__(256) void robust_elem_kernel(const float* __restrict__ a_in, const float* __restrict__ b_in, float* __restrict__ d_out,
                                                          int N, int D, const float* __restrict__ latents, const float* __restrict__ spline,
                                                          int n_knots, float x_scale, const ElemCoef coef_arg, float* __restrict__ loss,
                                                          float* __restrict__ dd, float* __restrict__ dlatent, float* __restrict__ part,
                                                          unsigned* __restrict__ ticket) {
  __shared__ float tot[4];
  __shared__ float coef_n[64];
  if (threadIdx.x < 64) coef_n[threadIdx.x] = coef_arg.v[threadIdx.x];     // (read from the kernel-argument segment)
  __syncthreads();
  const int j = blockIdx.x * 256 + threadIdx.x;
  float val = 0.0f;
  if (j < D) {
    const ChanParams P = chan_params(latents[j], latents[D + j], spline, n_knots, x_scale);
    float ga = 0.0f, gc = 0.0f;
    for (int n = 0; n < N; ++n) {
      const float x = a_in[(int64_t)n * D + j] - b_in[(int64_t)n * D + j], cf = coef_n[n];
      d_out[(int64_t)n * D + j] = x;
      const float xs = x / P.c, ssx = xs * xs;
      const float uu = ssx / P.beta + 1.0f, e = 0.5f * P.alpha, lnu = logf(uu);
      const float ue = expf(e * lnu), ue1 = ue / uu;
      val = fmaf(cf, (P.beta / P.alpha) * (ue - 1.0f) + P.logc_plus_logz, val);
      if (dd) {
        dd[(int64_t)n * D + j] = cf * (x / (P.c * P.c)) * ue1;
        ga = fmaf(cf, -(2.0f / (P.alpha * P.alpha)) * (ue - 1.0f) + (P.beta / P.alpha) * ue * (0.5f * lnu + e * ssx / (P.beta * P.beta * uu)) + P.dlogz, ga);
        gc = fmaf(cf, -(x * x) / (P.c * P.c * P.c) * ue1 + 1.0f / P.c, gc);
      }
    }
    if (dlatent) {
      dlatent[j] += ga * P.dalpha_dl;
      dlatent[D + j] += gc * P.dc_dl;
    }
  }
  for (int off = 32; off > 0; off >>= 1) val += __shfl_xor(val, off, 64);
  if ((threadIdx.x & 63) == 0) tot[threadIdx.x >> 6] = val;
  __syncthreads();
  // (round 6: the blocks' sums meet in block order -- one float atomicAdd per block in arrival order before; the style term was the
  //  last order-dependent float sum on a task's iteration path)
  if (threadIdx.x == 0) share_store(part + blockIdx.x, tot[0] + tot[1] + tot[2] + tot[3]);
  if (!block_last_arriver(ticket, gridDim.x)) return;
  if (threadIdx.x == 0) {
    float t = 0.0f;
    for (unsigned b = 0; b < gridDim.x; ++b) t += share_load(part + b);
    atomicAdd(loss, t);                           // ONE add per launch (the word is shared with launches on other streams: the
  }                                               //  contextual core's term of the same iteration)
}

// Split the contraction when the output is small and K long (weight gradients: 256 x 256 outputs over 2048 rows would
// be 16 workgroups looping 64 chunks each): partial sums by atomicAdd into a zeroed C.  Only for plain linear outputs
// written densely (ldc == N), which is what the weight-gradient form produces.  Batched launches (nbatch independent
// problems, blockIdx.z = batch * splits + split) count all the batches' tiles towards the fill target.
// Fills splits / kchunk / vec flags and returns the grid.
static dim3 gemm_prepare(GemmArgs& g, bool a_kc, bool b_kc, bool batch_split, int group_fill = 0) {
  const int nb = g.nbatch > 1 ? g.nbatch : 1;
  const int tiles = ((g.N + 63) / 64) * ((g.M + 63) / 64) * nb;
  // a batched launch may split when its partial sums have somewhere zeroed to meet: the caller accumulates, or the outputs of the
  // batches are back to back (one clear)
  const bool dense_out = g.scb == (int64_t)g.M * g.N;
  int splits = 1;
  if (g.act == 0 && !g.Z && !g.dact && g.ldc == g.N && tiles < 128 * nb && g.K >= 256 && (nb == 1 || (batch_split && (g.accumulate || dense_out)))) {
    static const int batch_fill = [] { const char* e = getenv("NPP_GEMM_BATCH_FILL"); return e ? atoi(e) : 640; }();
    const int fill = group_fill > 0 ? group_fill : (nb > 1 ? batch_fill : 512);
    splits = min(32, min((g.K + 63) / 64, max(1, (fill + tiles - 1) / tiles)));
  }
  g.kchunk = ((g.K + splits - 1) / splits + 31) / 32 * 32;
  splits = (g.K + g.kchunk - 1) / g.kchunk;
  g.splits = splits;
  // 16-byte loads along an operand's contiguous index: every run of 4 must be 16-byte aligned and whole (extent % 4 == 0)
  auto vec_ok = [&](const float* p, int64_t s_cont, int64_t s_other, int64_t s_batch, int extent) {
    return s_cont == 1 && (s_other % 4) == 0 && (extent % 4) == 0 && ((uintptr_t)p % 16) == 0 && (nb == 1 || (s_batch % 4) == 0);
  };
  g.vec_a = a_kc ? vec_ok(g.A, g.sak, g.sam, g.sab, g.K) : vec_ok(g.A, g.sam, g.sak, g.sab, g.M);
  g.vec_b = b_kc ? vec_ok(g.B, g.sbk, g.sbn, g.sbb, g.K) : vec_ok(g.B, g.sbn, g.sbk, g.sbb, g.N);
  return dim3((unsigned)((g.N + 63) / 64), (unsigned)((g.M + 63) / 64), (unsigned)(splits * nb));
}

// scratch of the ordered split of one (batched) problem: tickets, partial tiles, bias partials
static int64_t gemm_det_floats(const GemmArgs& g, const dim3& grid) {
  const int64_t nb = g.nbatch > 1 ? g.nbatch : 1, tiles = (int64_t)grid.x * grid.y;
  return nb * tiles + nb * tiles * g.splits * 4096 + nb * grid.y * g.splits * 64;
}
static int gemm_launch(GemmArgs g, bool a_kc, bool b_kc, hipStream_t s, bool batch_split = false, float* det_scratch = nullptr,
                       int64_t det_bytes = 0, bool two_pass = false) {
  const dim3 grid = gemm_prepare(g, a_kc, b_kc, batch_split);
  const int nb = g.nbatch > 1 ? g.nbatch : 1;
  if (det_scratch && g.splits > 1) {
    // the ranges of a tile meet in range order (gemm32_body): no float atomics, no clear of C; tickets zeroed once by the caller
    if (det_bytes < 4 * gemm_det_floats(g, grid)) { set_error("ordered-split scratch too small"); return NPP_ERR_ARG; }
    const int64_t tiles = (int64_t)grid.x * grid.y;
    g.ticket = two_pass ? nullptr : (unsigned*)det_scratch;
    g.slab = det_scratch + nb * tiles;
    g.rs_slab = g.slab + nb * tiles * g.splits * 4096;
    if (g.rowsum && !g.accumulate)                       // (the last arriver ADDS the bias sums)
      for (int b = 0; b < nb; ++b) (void)hipMemsetAsync(g.rowsum + b * g.srsb, 0, (size_t)g.M * sizeof(float), s);
  } else
  if (nb == 1) {
    if (g.splits > 1 && !g.accumulate) (void)hipMemsetAsync(g.C, 0, (size_t)g.M * g.N * sizeof(float), s);
    if (g.rowsum && !g.accumulate) (void)hipMemsetAsync(g.rowsum, 0, (size_t)g.M * sizeof(float), s);
  } else {
    if (g.splits > 1 && !g.accumulate) (void)hipMemsetAsync(g.C, 0, (size_t)nb * g.M * g.N * sizeof(float), s);      // dense_out
    if (g.rowsum && !g.accumulate)
      for (int b = 0; b < nb; ++b) (void)hipMemsetAsync(g.rowsum + b * g.srsb, 0, (size_t)g.M * sizeof(float), s);
  }
  if (a_kc && b_kc) hipLaunchKernelGGL((gemm32_kernel<true, true>), grid, dim3(256), 0, s, g);
  else if (a_kc) hipLaunchKernelGGL((gemm32_kernel<true, false>), grid, dim3(256), 0, s, g);
  else if (b_kc) hipLaunchKernelGGL((gemm32_kernel<false, true>), grid, dim3(256), 0, s, g);
  else hipLaunchKernelGGL((gemm32_kernel<false, false>), grid, dim3(256), 0, s, g);
  if (two_pass && g.slab && g.splits > 1 && !g.ticket)
    hipLaunchKernelGGL(gemm_slab_reduce_kernel, dim3((unsigned)(nb * grid.x * grid.y)), dim3(256), 0, s, g, (int)grid.x, (int)grid.y);
  return NPP_OK;
}

}  // namespace npp

using namespace npp;

static bool lin_dims_ok(int64_t B, int in, int out) { return B >= 1 && B < (1LL << 31) && in >= 1 && out >= 1; }

extern "C" int npp_linear_fwd(const float* d_x, int64_t ldx, const float* d_w, const float* d_b, int64_t B, int in, int out, int act,
                              float* d_y, int64_t ldy, float* d_z, int64_t ldz, void* stream) {
  if (!d_x || !d_w || !d_y || !lin_dims_ok(B, in, out) || ldx < in || ldy < out || (d_z && ldz < out) || act < 0 || act > 2) {
    set_error("npp_linear_fwd: bad argument (B=%lld in=%d out=%d act=%d)", (long long)B, in, out, act);
    return NPP_ERR_ARG;
  }
  GemmArgs g{};
  g.A = d_x; g.sam = ldx; g.sak = 1;
  g.B = d_w; g.sbk = 1; g.sbn = in;                  // B(k, n) = W[n][k]
  g.C = d_y; g.ldc = ldy; g.Z = d_z; g.ldz = ldz; g.bias = d_b;
  g.M = (int)B; g.N = out; g.K = in; g.act = act; g.accumulate = 0;
  gemm_launch(g, true, true, (hipStream_t)stream);
  return check_launch("npp_linear_fwd");
}

extern "C" int npp_linear_bwd_data(const float* d_dz, int64_t lddz, const float* d_w, int64_t B, int in, int out, float* d_dx,
                                   int64_t lddx, int in_used, int accumulate, void* stream) {
  if (!d_dz || !d_w || !d_dx || !lin_dims_ok(B, in, out) || lddz < out || in_used < 1 || in_used > in || lddx < in_used) {
    set_error("npp_linear_bwd_data: bad argument (B=%lld in=%d out=%d in_used=%d)", (long long)B, in, out, in_used);
    return NPP_ERR_ARG;
  }
  GemmArgs g{};
  g.A = d_dz; g.sam = lddz; g.sak = 1;               // A(m = row, k = n_out)
  g.B = d_w; g.sbk = in; g.sbn = 1;                  // B(k = n_out, col) = W[n_out][col]
  g.C = d_dx; g.ldc = lddx;
  g.M = (int)B; g.N = in_used; g.K = out; g.act = 0; g.accumulate = accumulate;
  gemm_launch(g, true, false, (hipStream_t)stream);
  return check_launch("npp_linear_bwd_data");
}

extern "C" int npp_linear_bwd_weight(const float* d_dz, int64_t lddz, const float* d_x, int64_t ldx, int64_t B, int in, int out,
                                     float* d_dw, float* d_db, int accumulate, void* stream) {
  if (!d_dz || !d_x || !d_dw || !lin_dims_ok(B, in, out) || lddz < out || ldx < in) {
    set_error("npp_linear_bwd_weight: bad argument (B=%lld in=%d out=%d)", (long long)B, in, out);
    return NPP_ERR_ARG;
  }
  GemmArgs g{};
  g.A = d_dz; g.sam = 1; g.sak = lddz;               // A(m = n_out, k = row) = dz[row][n_out]
  g.B = d_x; g.sbk = ldx; g.sbn = 1;                 // B(k = row, col) = x[row][col]
  g.C = d_dw; g.ldc = in;
  g.M = out; g.N = in; g.K = (int)B; g.act = 0; g.accumulate = accumulate;
  g.rowsum = d_db;                                   // db[n_out] = sum_rows dz[row][n_out] = row sums of A
  gemm_launch(g, false, false, (hipStream_t)stream);
  return check_launch("npp_linear_bwd_weight");
}

/* The three dense-layer forms over nbatch independent problems of one shape in ONE launch (the proposal-ranking candidates of one
 * image: same rows, each its own weights): element strides between the problems' arrays are the s*b arguments.  The weight-gradient
 * form splits its contraction like the single-problem form and therefore ACCUMULATES into d_dw / d_db (caller clears them). */
extern "C" int npp_linear_fwd_batched(const float* d_x, int64_t ldx, int64_t sxb, const float* d_w, int64_t swb, const float* d_b,
                                      int64_t sbb, int nbatch, int64_t B, int in, int out, int act, float* d_y, int64_t ldy, int64_t syb,
                                      float* d_z, int64_t ldz, int64_t szb, void* stream) {
  if (!d_x || !d_w || !d_y || nbatch < 1 || nbatch > 4096 || !lin_dims_ok(B, in, out) || ldx < in || ldy < out || (d_z && ldz < out) || act < 0 ||
      act > 2) {
    set_error("npp_linear_fwd_batched: bad argument (nbatch=%d B=%lld in=%d out=%d act=%d)", nbatch, (long long)B, in, out, act);
    return NPP_ERR_ARG;
  }
  GemmArgs g{};
  g.A = d_x; g.sam = ldx; g.sak = 1;
  g.B = d_w; g.sbk = 1; g.sbn = in;
  g.C = d_y; g.ldc = ldy; g.Z = d_z; g.ldz = ldz; g.bias = d_b;
  g.M = (int)B; g.N = out; g.K = in; g.act = act; g.accumulate = 0;
  g.nbatch = nbatch; g.sab = sxb; g.sbb = swb; g.scb = syb; g.szb = szb; g.sbiasb = sbb;
  gemm_launch(g, true, true, (hipStream_t)stream);
  return check_launch("npp_linear_fwd_batched");
}

extern "C" int npp_linear_bwd_data_batched(const float* d_dz, int64_t lddz, int64_t sdzb, const float* d_w, int64_t swb, int nbatch, int64_t B,
                                           int in, int out, float* d_dx, int64_t lddx, int64_t sdxb, int in_used, int accumulate,
                                           const float* d_zy, int64_t ldzy, int64_t szyb, int act, void* stream) {
  if (!d_dz || !d_w || !d_dx || nbatch < 1 || nbatch > 4096 || !lin_dims_ok(B, in, out) || lddz < out || in_used < 1 || in_used > in ||
      lddx < in_used || (d_zy && (ldzy < in_used || act < 1 || act > 4 || accumulate))) {
    set_error("npp_linear_bwd_data_batched: bad argument (nbatch=%d B=%lld in=%d out=%d in_used=%d)", nbatch, (long long)B, in, out, in_used);
    return NPP_ERR_ARG;
  }
  GemmArgs g{};
  g.A = d_dz; g.sam = lddz; g.sak = 1;
  g.B = d_w; g.sbk = in; g.sbn = 1;
  g.C = d_dx; g.ldc = lddx;
  g.M = (int)B; g.N = in_used; g.K = out; g.act = 0; g.accumulate = accumulate;
  g.nbatch = nbatch; g.sab = sdzb; g.sbb = swb; g.scb = sdxb;
  g.dact = d_zy; g.lddact = ldzy; g.sdactb = szyb; g.dact_kind = act;
  gemm_launch(g, true, false, (hipStream_t)stream);
  return check_launch("npp_linear_bwd_data_batched");
}

extern "C" int npp_linear_bwd_weight_batched(const float* d_dz, int64_t lddz, int64_t sdzb, const float* d_x, int64_t ldx, int64_t sxb,
                                             int nbatch, int64_t B, int in, int out, float* d_dw, int64_t sdwb, float* d_db, int64_t sdbb,
                                             void* stream) {
  if (!d_dz || !d_x || !d_dw || nbatch < 1 || nbatch > 4096 || !lin_dims_ok(B, in, out) || lddz < out || ldx < in) {
    set_error("npp_linear_bwd_weight_batched: bad argument (nbatch=%d B=%lld in=%d out=%d)", nbatch, (long long)B, in, out);
    return NPP_ERR_ARG;
  }
  GemmArgs g{};
  g.A = d_dz; g.sam = 1; g.sak = lddz;
  g.B = d_x; g.sbk = ldx; g.sbn = 1;
  g.C = d_dw; g.ldc = in;
  g.M = out; g.N = in; g.K = (int)B; g.act = 0; g.accumulate = 1;
  g.rowsum = d_db;
  g.nbatch = nbatch; g.sab = sdzb; g.sbb = sxb; g.scb = sdwb; g.srsb = sdbb;
  gemm_launch(g, false, false, (hipStream_t)stream, true);
  return check_launch("npp_linear_bwd_weight_batched");
}

extern "C" int npp_linear_bwd_weight_strided(const float* d_dz, int64_t dz_sr, int64_t dz_so, int64_t sdzb, const float* d_x, int64_t x_sr,
                                             int64_t x_si, int64_t sxb, int x_snake, int nbatch, int64_t B, int in, int out, float* d_dw,
                                             int64_t lddw, int64_t sdwb, float* d_db, int64_t sdbb, void* stream) {
  if (!d_dz || !d_x || !d_dw || nbatch < 1 || nbatch > 4096 || !lin_dims_ok(B, in, out) || lddw < in || dz_sr < 1 || dz_so < 1 || x_sr < 1 ||
      x_si < 1) {
    set_error("npp_linear_bwd_weight_strided: bad argument (nbatch=%d B=%lld in=%d out=%d)", nbatch, (long long)B, in, out);
    return NPP_ERR_ARG;
  }
  GemmArgs g{};
  g.A = d_dz; g.sam = dz_so; g.sak = dz_sr;          // A(m = n_out, k = row)
  g.B = d_x; g.sbk = x_sr; g.sbn = x_si;             // B(k = row, n = col)
  g.C = d_dw; g.ldc = lddw;
  g.M = out; g.N = in; g.K = (int)B; g.act = 0; g.accumulate = 1;
  g.rowsum = d_db; g.b_snake = x_snake;
  g.nbatch = nbatch; g.sab = sdzb; g.sbb = sxb; g.scb = sdwb; g.srsb = sdbb;
  gemm_launch(g, dz_sr == 1, x_sr == 1, (hipStream_t)stream, true);
  return check_launch("npp_linear_bwd_weight_strided");
}

/* The seven weight (and bias) gradients of C stacked NPP_Net_light candidates in ONE launch, over the feature-major stashes of the
 * fused chains (npp_light_fwd / npp_light_bwd): dW_l += d z_l^T x_l with x_0 = x_per, x_l = snake(z_{l-1}), x_f1 = snake(z_3),
 * x_pos = [f1 | x_pos], x_rgb = snake(z_p); gradients go to d_grad + c * grad_stride at the offsets npp_light_desc gives the
 * parameters (accumulated: clear the blob first). */
// the ordered-split form: ranges of 1024 rows whatever the number of candidates -- a candidate's gradient must not depend on how many
// others ride in the launch (the images of a rank searched together give the bits of the serial loop).  Measured at 8 candidates x 2048
// rows: 165 us unsplit (640 workgroups of 64 chunks, 2.5 per CU: every chunk waits for its own operand load) -> ~115 us with 2 ranges
// (5 per CU); splitting by the candidate count instead (16 ranges for one candidate) left the serial search where it was -- that loop
// is bound by the host's launches (tools/r6_search_together.py).  80 output tiles of 64 x 64 per candidate (4 + 3 x 16 + 16 + 10 + 2).
static int light_det_splits(int C, int64_t B) {
  (void)C;
  static const int rows = [] { const char* e = getenv("NPP_LIGHT_DET_ROWS"); return e ? atoi(e) : 1024; }();
  int64_t s = (B + rows - 1) / (rows > 32 ? rows : 32);
  return (int)(s < 1 ? 1 : (s > 32 ? 32 : s));
}
constexpr int kLightTiles = 96, kLightTileRows = 24;    // upper bounds of the 80 tiles / 19 tile rows (layer widths are parameters)
extern "C" int64_t npp_light_wgrad_det_scratch_bytes(int C, int64_t B) {
  if (C < 1 || B < 32) return 0;
  const int s = light_det_splits(C, B);
  return 4 * ((int64_t)C * kLightTiles /* tickets */ + (int64_t)C * kLightTiles * s * 4096 + (int64_t)C * kLightTileRows * s * 64);
}
static int light_wgrad_go(const npp_light_desc* L, const float* d_stash, const float* d_dstash, int C, int64_t B, float* d_grad,
                          int64_t grad_stride, float* d_scratch, int64_t scratch_bytes, void* stream) {
  if (!L || !d_stash || !d_dstash || !d_grad || C < 1 || C > 4096 || B < 32 || B % 32 || B * 512 >= 0x7fffffffLL) {
    set_error("npp_light_wgrad: bad argument (C=%d B=%lld)", C, (long long)B);
    return NPP_ERR_ARG;
  }
  const int det_s = d_scratch ? light_det_splits(C, B) : 0;
  if (d_scratch && scratch_bytes < npp_light_wgrad_det_scratch_bytes(C, B)) { set_error("npp_light_wgrad_det: scratch too small"); return NPP_ERR_ARG; }
  unsigned* tickets = (unsigned*)d_scratch;
  float* slabs = d_scratch ? d_scratch + (int64_t)C * kLightTiles : nullptr;
  float* rs_slabs = d_scratch ? slabs + (int64_t)C * kLightTiles * det_s * 4096 : nullptr;
  int tiles_used = 0, rows_used = 0;
  // npp_light_desc index -> (d z rows, x rows, x stored as pre-activation)
  const int dz_row[7] = {LD_Z0, LD_Z1, LD_Z2, LD_Z3, LD_ZP, LD_F1, LD_RAW};
  const int x_row[7] = {LS_XP, LS_Z0, LS_Z1, LS_Z2, LS_HP, LS_Z3, LS_ZP};
  const int x_snake[7] = {0, 1, 1, 1, 0, 1, 1};
  GemmGroup G{};
  int wg = 0;
  for (int i = 0; i < 7; ++i) {
    GemmArgs& g = G.g[i];
    const int out = L->n_out[i], in = L->ld[i];          // stored columns (zero-padded rows of the operand beyond n_in)
    if (in > (i == 4 ? kLHp : (i == 0 ? kLPer : kLW)) || L->ld[i] < L->n_in[i]) { set_error("npp_light_wgrad: layer %d ld %d", i, in); return NPP_ERR_ARG; }
    g.A = d_dstash + (int64_t)dz_row[i] * B; g.sam = B; g.sak = 1;       // A(m = n_out, k = row)
    g.B = d_stash + (int64_t)x_row[i] * B; g.sbk = 1; g.sbn = B;         // B(k = row, n = col)
    g.C = d_grad + L->w_off[i]; g.ldc = in;
    g.rowsum = d_grad + L->b_off[i];
    g.M = out; g.N = in; g.K = (int)B; g.accumulate = 1; g.b_snake = x_snake[i];
    g.nbatch = C; g.sab = (int64_t)LD_ROWS * B; g.sbb = (int64_t)LS_ROWS * B; g.scb = grad_stride; g.srsb = grad_stride;
    if (C == 1) g.nbatch = 1;
    // (one launch holds all seven problems: a problem needs only a few hundred workgroups of its own -- 2 ranges instead of 5 for the
    //  stacked 256-wide layers: 481 -> 469 us per iteration of 9 candidates; a single candidate: 8 ranges instead of 32 -- its 8.4 M
    //  atomic adds were half of the launch: 215 -> 184 us per iteration)
    static const int gfill = [] { const char* e = getenv("NPP_LIGHT_WGRAD_FILL"); return e ? atoi(e) : 160; }();
    static const int gfill1 = [] { const char* e = getenv("NPP_LIGHT_WGRAD_FILL1"); return e ? atoi(e) : 128; }();
    // npp_tune "light_det" (default 1): no split of the contraction -- every output element is the plain-store result of ONE
    // workgroup's fixed-order sum (split partial sums meet by float atomicAdd in arrival order): bit-reproducible candidate fits
    dim3 grid = gemm_prepare(g, true, true, true, C > 1 ? gfill : gfill1);
    if (d_scratch) {
      // npp_light_wgrad_det: the contraction IS split (det_s ranges of whole 32-row chunks), the ranges of a tile meet in the slab
      g.kchunk = (((g.K + det_s - 1) / det_s) + 31) / 32 * 32;
      g.splits = (g.K + g.kchunk - 1) / g.kchunk;
      const int gx = (int)grid.x, gy = (int)grid.y;
      if (tiles_used + gx * gy > kLightTiles || rows_used + gy > kLightTileRows) { set_error("npp_light_wgrad_det: layer %d too wide", i); return NPP_ERR_ARG; }
      // (batch c of problem i: tile slabs [c][tile][split], one region per problem)
      g.ticket = tickets + (int64_t)C * tiles_used;
      g.slab = slabs + (int64_t)C * tiles_used * det_s * 4096;
      g.rs_slab = rs_slabs + (int64_t)C * rows_used * det_s * 64;
      tiles_used += gx * gy; rows_used += gy;
      grid.z = (unsigned)(g.splits * (g.nbatch > 1 ? g.nbatch : 1));
    } else if (g.splits > 1 && __atomic_load_n(&g_tune.light_det, __ATOMIC_RELAXED)) {
      g.splits = 1; g.kchunk = (g.K + 31) / 32 * 32;
      grid.z = (unsigned)(g.nbatch > 1 ? g.nbatch : 1);
    }
    G.first_wg[i] = wg; G.gx[i] = (int)grid.x; G.gy[i] = (int)grid.y;
    wg += (int)(grid.x * grid.y * grid.z);
  }
  G.first_wg[7] = wg; G.n = 7;
  hipLaunchKernelGGL((gemm32_grouped_kernel<true, true>), dim3((unsigned)wg), dim3(256), 0, (hipStream_t)stream, G);
  return check_launch("npp_light_wgrad");
}
extern "C" int npp_light_wgrad(const npp_light_desc* L, const float* d_stash, const float* d_dstash, int C, int64_t B, float* d_grad,
                               int64_t grad_stride, void* stream) {
  return light_wgrad_go(L, d_stash, d_dstash, C, B, d_grad, grad_stride, nullptr, 0, stream);
}
/* npp_light_wgrad with the contraction split over enough workgroups to fill the chip AND bit-reproducible: the ranges of an output
 * tile leave their partial tiles in d_scratch and the workgroup that arrives last adds them in range order (no float atomics).
 * d_scratch: npp_light_wgrad_det_scratch_bytes(C, B) bytes, ZEROED once before its first use (the tickets reset themselves). */
extern "C" int npp_light_wgrad_det(const npp_light_desc* L, const float* d_stash, const float* d_dstash, int C, int64_t B, float* d_grad,
                                   int64_t grad_stride, float* d_scratch, int64_t scratch_bytes, void* stream) {
  if (!d_scratch) { set_error("npp_light_wgrad_det: null scratch"); return NPP_ERR_ARG; }
  return light_wgrad_go(L, d_stash, d_dstash, C, B, d_grad, grad_stride, d_scratch, scratch_bytes, stream);
}

extern "C" int npp_act_bwd(const float* d_dy, int64_t lddy, const float* d_zy, int64_t ldzy, int64_t B, int n, int act, float* d_dz,
                           int64_t lddz, void* stream) {
  if (!d_dy || !d_zy || !d_dz || B < 1 || n < 1 || lddy < n || ldzy < n || lddz < n || act < 0 || act > 4) {
    set_error("npp_act_bwd: bad argument");
    return NPP_ERR_ARG;
  }
  const int64_t t = B * n;
  hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_dy, lddy, d_zy, ldzy, B, n, act,
                     d_dz, lddz);
  return check_launch("npp_act_bwd");
}

extern "C" int npp_act_fwd(const float* d_x, int64_t n, int act, float* d_y, void* stream) {
  if (!d_x || !d_y || n < 1 || act < 0 || act > 3) { set_error("npp_act_fwd: bad argument"); return NPP_ERR_ARG; }
  hipLaunchKernelGGL(act_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_x, n, act, d_y);
  return check_launch("npp_act_fwd");
}

static int lpips_plain_go(const float* d_f0, const float* d_f1, int N, int C, int hw, const float* d_lin, float scale, float* d_out,
                          float* d_scratch, void* stream);
extern "C" int npp_lpips_plain_layer(const float* d_f0, const float* d_f1, int N, int C, int hw, const float* d_lin, float scale,
                                     float* d_out, void* stream) {
  return lpips_plain_go(d_f0, d_f1, N, C, hw, d_lin, scale, d_out, nullptr, stream);
}
// the same with the blocks' partial sums added in block order (bit-reproducible): d_scratch = NPP_LPIPS_PLAIN_SCRATCH_FLOATS floats,
// zeroed once by the caller (the launch re-arms it), not shared by launches that may run concurrently
extern "C" int npp_lpips_plain_layer_det(const float* d_f0, const float* d_f1, int N, int C, int hw, const float* d_lin, float scale,
                                         float* d_out, float* d_scratch, void* stream) {
  if (!d_scratch) { set_error("npp_lpips_plain_layer_det: null scratch"); return NPP_ERR_ARG; }
  return lpips_plain_go(d_f0, d_f1, N, C, hw, d_lin, scale, d_out, d_scratch, stream);
}
static int lpips_plain_go(const float* d_f0, const float* d_f1, int N, int C, int hw, const float* d_lin, float scale, float* d_out,
                          float* d_scratch, void* stream) {
  if (!d_f0 || !d_f1 || !d_lin || !d_out || N < 1 || C < 16 || (C % 16) || hw < 1) {
    set_error("npp_lpips_plain_layer: bad arguments (N=%d C=%d hw=%d)", N, C, hw);
    return NPP_ERR_ARG;
  }
  const int64_t groups = ((int64_t)N * hw + 15) / 16;
  hipLaunchKernelGGL(lpips_plain_kernel, dim3((unsigned)(groups < 256 ? groups : 256)), dim3(256), 0, (hipStream_t)stream, d_f0, d_f1, N, C,
                     hw, d_lin, scale / (float)hw, d_out, d_scratch);
  return check_launch("npp_lpips_plain_layer");
}

/* Gram matrices G[n] = F[n] F[n]^T of (N, C, hw) features (models/style_loss.py:55-58), and the backward
 * dF[n] = (dG[n] + dG[n]^T) F[n]. */
extern "C" int npp_gram_fwd(const float* d_f, int N, int C, int hw, float* d_g, void* stream) {
  if (!d_f || !d_g || N < 1 || C < 1 || hw < 1) { set_error("npp_gram_fwd: bad argument"); return NPP_ERR_ARG; }
  GemmArgs g{};
  g.A = d_f; g.sam = hw; g.sak = 1;                  // A(m = c, k = pos)
  g.B = d_f; g.sbk = 1; g.sbn = hw;                  // B(k = pos, n = c') = F[c'][pos]
  g.C = d_g; g.ldc = C;
  g.M = C; g.N = C; g.K = hw; g.nbatch = N; g.sab = g.sbb = (int64_t)C * hw; g.scb = (int64_t)C * C;
  // C x C outputs over hw = thousands of positions: without a split this is one workgroup per 64 x 64 tile walking the whole
  // contraction (measured at C = 64, hw = 25 600, N = 6: 6 workgroups, 235 us); split, the partial sums meet by atomicAdd
  gemm_launch(g, true, true, (hipStream_t)stream, true);
  return check_launch("npp_gram_fwd");
}
static GemmArgs gram_args(const float* d_f, int N, int C, int hw, float* d_g) {
  GemmArgs g{};
  g.A = d_f; g.sam = hw; g.sak = 1;
  g.B = d_f; g.sbk = 1; g.sbn = hw;
  g.C = d_g; g.ldc = C;
  g.M = C; g.N = C; g.K = hw; g.nbatch = N; g.sab = g.sbb = (int64_t)C * hw; g.scb = (int64_t)C * C;
  return g;
}
/* npp_gram_fwd with the split contraction's partial sums added in range order (bit-reproducible; npp_gram_fwd adds them with float
 * atomics in arrival order): the ranges leave their partial tiles in d_scratch, a second small launch adds them.
 * d_scratch: npp_gram_fwd_det_scratch_bytes(N, C, hw) bytes (no initial content required). */
extern "C" int64_t npp_gram_fwd_det_scratch_bytes(int N, int C, int hw) {
  if (N < 1 || C < 1 || hw < 1) return NPP_ERR_ARG;
  GemmArgs g = gram_args(nullptr, N, C, hw, nullptr);
  const dim3 grid = gemm_prepare(g, true, true, true);
  return 4 * gemm_det_floats(g, grid) + 64;
}
extern "C" int npp_gram_fwd_det(const float* d_f, int N, int C, int hw, float* d_g, float* d_scratch, int64_t scratch_bytes, void* stream) {
  if (!d_f || !d_g || !d_scratch || N < 1 || C < 1 || hw < 1) { set_error("npp_gram_fwd_det: bad argument"); return NPP_ERR_ARG; }
  const int rc = gemm_launch(gram_args(d_f, N, C, hw, d_g), true, true, (hipStream_t)stream, true, d_scratch, scratch_bytes, true);
  return rc != NPP_OK ? rc : check_launch("npp_gram_fwd_det");
}

extern "C" int npp_gram_bwd(const float* d_dg, const float* d_f, int N, int C, int hw, float* d_df, void* stream) {
  if (!d_dg || !d_f || !d_df || N < 1 || C < 1 || hw < 1) { set_error("npp_gram_bwd: bad argument"); return NPP_ERR_ARG; }
  GemmArgs g{};
  g.B = d_f; g.sbk = hw; g.sbn = 1;                  // B(k = c', n = pos)
  g.C = d_df; g.ldc = hw;
  g.M = C; g.N = hw; g.K = C; g.nbatch = N; g.sab = (int64_t)C * C; g.sbb = g.scb = (int64_t)C * hw;
  g.A = d_dg; g.sam = C; g.sak = 1;                  // dG F
  gemm_launch(g, true, false, (hipStream_t)stream);
  g.sam = 1; g.sak = C; g.accumulate = 1;            // + dG^T F
  if (N == 1) g.nbatch = 1;
  gemm_launch(g, false, false, (hipStream_t)stream);
  return check_launch("npp_gram_bwd");
}

/* diff = a - b, then the per-element adaptive robust NLL over (N, D) with D latent pairs [alpha(D) | scale(D)]:
 * d_loss[0] += sum_n coef_n sum_j nll ; d_ddiff (N, D) = coef_n dnll/dx ; d_dlatent [2 D] += latent gradients (both nullable
 * together).  coef_n: N per-sample factors (host array).  d_workspace: npp_robust_elem_workspace_bytes(D) bytes. */
extern "C" int npp_robust_elem(const float* d_a, const float* d_b, int N, int D, const float* d_latents, const float* d_spline,
                               int n_knots, float x_scale, const float* coef_n, float* d_loss, float* d_diff, float* d_ddiff,
                               float* d_dlatent, void* d_workspace, void* stream) {
  if (!d_a || !d_b || !d_latents || !d_spline || !coef_n || !d_loss || !d_diff || !d_workspace || N < 1 || N > 64 || D < 1 || n_knots < 2 ||
      ((d_ddiff == nullptr) != (d_dlatent == nullptr))) {
    set_error("npp_robust_elem: bad argument (N=%d D=%d; N <= 64)", N, D);
    return NPP_ERR_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  // workspace: [blocks] partial loss sums, then the arrival ticket (cleared here: the workspace needs no initial content)
  const unsigned blocks = (unsigned)((D + 255) / 256);
  float* d_part = (float*)d_workspace;
  unsigned* d_ticket = (unsigned*)(d_part + blocks);
  (void)hipMemsetAsync(d_ticket, 0, sizeof(unsigned), s);
  ElemCoef coef{};
  for (int q = 0; q < N; ++q) coef.v[q] = coef_n[q];
  hipLaunchKernelGGL(robust_elem_kernel, dim3(blocks), dim3(256), 0, s, d_a, d_b, d_diff, N, D, d_latents, d_spline, n_knots, x_scale, coef, d_loss, d_ddiff,
                     d_dlatent, d_part, d_ticket);
  return check_launch("npp_robust_elem");
}

extern "C" int64_t npp_robust_elem_workspace_bytes(int D) {
  return D < 1 ? NPP_ERR_ARG : (int64_t)((D + 255) / 256 + 16) * sizeof(float);
}
