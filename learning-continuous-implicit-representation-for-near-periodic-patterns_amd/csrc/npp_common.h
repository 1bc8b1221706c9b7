// npp_common.h -- device-side helpers shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "npp_layout.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

namespace npp {

// host-side error plumbing (npp_api.cpp)
void set_error(const char* fmt, ...);
int check_launch(const char* what);
// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE setting of a kernel: each launch site remembers which devices
// of this process already have it (bit d of the mask), so a second GPU driven from the same process is set up as well.
struct SmemOnce { unsigned long long done = 0; };

// Launch-time choices between kernel forms that compute the same result (npp_tune(); defaults = the measured-best forms).  Read
// with relaxed loads by the launchers; an environment variable of the upper-cased name prefixed NPP_ (NPP_CONV_WINK=0) sets the
// initial value, so that tools can A/B whole processes, and npp_tune() flips it inside one process (parity tests, probes).
struct Tunables {
  int conv_wink;      // 1: group-split window convolution on the channel-rich trunk layers; 0: never; 2: wherever feasible
  int conv_win;       // 1: window-staged convolution on conv1_1 / conv1_2 / conv2_1 forward; 0: never; 2: wherever feasible
  int conv_wstat;     // 1: weight-stationary block numbering where the pack outweighs the activations; 0: never
  int conv_pair;      // bit mask of the fused convolution pairs (conv a -> conv b -> pool) the trunks use: 1 = first block, 2 = second, 4 = the first block's data gradient, 8 = patch plumbing + pixel loss inside the first pair; 0: never
  int light_det;      // 1 (default): the proposal-ranking fits' fp32 weight-gradient launch does not split its contraction (no float atomics on the gradient path: bit-reproducible fits); 0: split-K by atomicAdd
  int stash8;         // 1 (default): the training stash that feeds the weight gradients is 8-bit (bf8 gradients with a per-tile power-of-two scale, fp8 layer inputs; npp_layout.h "W8-format") and npp_mlp_wgrad runs on v_mfma_scale_f32_32x32x64_f8f6f4; 0: the round-2..5 16-bit stash and bf16 weight-gradient launch.  Read by npp_mlp_fwd* / npp_mlp_bwd* / npp_mlp_wgrad* at launch: flip it only between complete iterations
};
extern Tunables g_tune;
bool smem_attr(SmemOnce& once, const void* fn, int bytes);

constexpr float kInv2Pi = 0.15915494309189535f;

// Embedder constants precomputed on the host from npp_embed_cfg (kernel argument).
// models/embedder.py:117-127: freq = period + offset, theta = deg2rad(angle).
struct EmbedDev {
  int32_t K, H, W, pad;
  float inv_w, inv_h;
  float cs[NPP_MAX_K][2], sn[NPP_MAX_K][2];            // cos/sin(theta_i)
  float per[NPP_MAX_K][2][NPP_N_OFF];                  // period_i + offset_o
  float freq[NPP_N_FREQ];                              // Fourier freq (rad per unit)
  float freq_rev[NPP_N_FREQ];                          // freq / 2pi (revolutions)
};

// sin / cos of an fp32 argument to ~1 ulp without libm's slow path: three-term Cody-Waite reduction by pi/2 (exact products
// for |x| up to ~1e5 rad; the embedder's arguments are |f v| < ~50) + the cephes single-precision minimax polynomials on
// [-pi/4, pi/4].  ~20 vector instructions, no branches: what lets the precise fp32 embedder run at the store rate instead of
// at libm's (5.0 vs 2.4 TB/s at 1024^2).
__device__ __forceinline__ float sincos_pi2(float x, bool want_cos) {
  const float jf = rintf(x * 0.636619772367581343f);           // x * 2 / pi
  float y = fmaf(-jf, 1.5703125f, x);
  y = fmaf(-jf, 4.837512969970703125e-4f, y);
  y = fmaf(-jf, 7.54978995489188e-8f, y);
  const int q = ((int)jf + (want_cos ? 1 : 0)) & 3;
  const float z = y * y;
  const float s = fmaf(y * z, fmaf(z, fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), y);
  const float c = fmaf(z * z, fmaf(z, fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f),
                       fmaf(z, -0.5f, 1.0f));
  const float r = (q & 1) ? c : s;
  return (q & 2) ? -r : r;
}

// ---- a1: one warped coordinate (models/embedder.py:110-133) -------------------
// i in [0,22): 0 -> x/W*2-1 ; 1..10 -> sin/cos pairs of orientation 0 over the 5
// offsets ; 11 -> y/H*2-1 ; 12..21 -> orientation 1.
template <bool PRECISE>
__device__ __forceinline__ float warp_value(const EmbedDev& e, int p, int i, float y, float x) {
  const int ori = i >= 11;
  const int ii = ori ? i - 11 : i;
  if (ii == 0) {
    // (x / res[1] - 0.5) * 2   (embedder.py:112-113)
    return ori ? (y / (float)e.H - 0.5f) * 2.0f : (x / (float)e.W - 0.5f) * 2.0f;
  }
  const int o = (ii - 1) >> 1;
  const bool is_cos = (ii - 1) & 1;
  const float per = e.per[p][ori][o];
  // y*cos + x*sin, rounded like the reference's two torch ops (no fma contraction)
  const float t = __fadd_rn(__fmul_rn(y, e.cs[p][ori]), __fmul_rn(x, e.sn[p][ori]));
  if (PRECISE) {
    float r = fmodf(t, per);                       // torch.remainder: sign of divisor
    if (r != 0.0f && ((per < 0.0f) != (r < 0.0f))) r += per;
    const float phi = ((r / per) * 2.0f) * 3.14159265358979323846f;
    return sincos_pi2(phi, is_cos);
  } else {
    const float q = floorf(t / per);
    float r = fmaf(-q, per, t);                    // exact remainder for |q| < 2^24
    r = r < 0.0f ? r + per : r;
    r = r >= per ? r - per : r;
    const float rev = r / per;                     // phase in revolutions, [0,1)
    return is_cos ? __builtin_amdgcn_cosf(rev) : __builtin_amdgcn_sinf(rev);
  }
}

// Both functions of one argument from ONE reduction (bitwise the values sincos_pi2(x, false) / (x, true) return): the
// precise embedder needs sin(f v) and cos(f v) of every (frequency, coordinate) pair.
__device__ __forceinline__ void sincos_pi2_both(float x, float& sn, float& cs) {
  const float jf = rintf(x * 0.636619772367581343f);
  float y = fmaf(-jf, 1.5703125f, x);
  y = fmaf(-jf, 4.837512969970703125e-4f, y);
  y = fmaf(-jf, 7.54978995489188e-8f, y);
  const int q = (int)jf;
  const float z = y * y;
  const float s = fmaf(y * z, fmaf(z, fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), y);
  const float c = fmaf(z * z, fmaf(z, fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f),
                       fmaf(z, -0.5f, 1.0f));
  const float rs = (q & 1) ? c : s;
  sn = (q & 2) ? -rs : rs;
  const int qc = q + 1;
  const float rc = (qc & 1) ? c : s;
  cs = (qc & 2) ? -rc : rc;
}

// ---- a6: snake and its derivative (models/activations.py:29-35, a = 1) -----------
__device__ __forceinline__ float snake_fast(float z) {
  const float s = __builtin_amdgcn_sinf(z * kInv2Pi);
  return fmaf(s, s, z);
}
__device__ __forceinline__ void snake_fast2(float z, float& a, float& da) {
  const float rev = z * kInv2Pi;
  const float s = __builtin_amdgcn_sinf(rev);
  const float c = __builtin_amdgcn_cosf(rev);
  a = fmaf(s, s, z);
  da = fmaf(2.0f * s, c, 1.0f);                    // 1 + sin(2z)
}

__device__ __forceinline__ bf16x8 pack_acc(const f32x16& acc, int s) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (__bf16)acc[8 * s + j];
  return r;
}

__device__ __forceinline__ f32x16 mfma_bf16(const bf16x8& a, const bf16x8& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float bf16_to_f32(__bf16 v) { return (float)v; }

// Streaming store of a 16-byte fragment to a stash array: written once, read once by a later
// kernel, so it should not displace the L2-resident weight packs (NPP_STASH_NT=0 to compare).
#ifndef NPP_STASH_NT
#define NPP_STASH_NT 1
#endif
__device__ __forceinline__ void stash_store(void* p, const bf16x8& v) {
#if NPP_STASH_NT
  __builtin_nontemporal_store(v, (bf16x8*)p);
#else
  *(bf16x8*)p = v;
#endif
}
// Pre-activation gradients (npp_mlp_bwd -> npp_mlp_wgrad, the next launch): NPP_DZ_NT=0 lets them allocate in the caches
#ifndef NPP_DZ_NT
#define NPP_DZ_NT NPP_STASH_NT
#endif
__device__ __forceinline__ void dz_store(void* p, const bf16x8& v) {
#if NPP_DZ_NT
  __builtin_nontemporal_store(v, (bf16x8*)p);
#else
  *(bf16x8*)p = v;
#endif
}
// ---- 8-bit stash helpers (npp_layout.h "W8-format") --------------------------------------------------------------------
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(8))) int i32x8;
// v_cvt_pk_fp8_f32 / v_cvt_pk_bf8_f32 turn a finite value beyond the format's range into NaN / infinity unless MODE.FP16_OVFL is
// set, in which case they saturate (0x7e = 448 / 0x7b = 57344; tools/micro/fp8_probe.hip, round 6).  Kernels that write the 8-bit
// stash set the bit once at entry (it also makes the fp16 z stash saturate instead of overflowing to infinity).
__device__ __forceinline__ void set_fp16_ovfl() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1"); }
// 8 consecutive accumulator registers (or any 8 floats) -> the 8 bytes of a W8 unit
__device__ __forceinline__ u32x2 pack8_fp8(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7) {
  int lo = __builtin_amdgcn_cvt_pk_fp8_f32(v0, v1, 0, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(v2, v3, lo, true);
  int hi = __builtin_amdgcn_cvt_pk_fp8_f32(v4, v5, 0, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(v6, v7, hi, true);
  return u32x2{(uint32_t)lo, (uint32_t)hi};
}
__device__ __forceinline__ u32x2 pack8_bf8(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7) {
  int lo = __builtin_amdgcn_cvt_pk_bf8_f32(v0, v1, 0, false);
  lo = __builtin_amdgcn_cvt_pk_bf8_f32(v2, v3, lo, true);
  int hi = __builtin_amdgcn_cvt_pk_bf8_f32(v4, v5, 0, false);
  hi = __builtin_amdgcn_cvt_pk_bf8_f32(v6, v7, hi, true);
  return u32x2{(uint32_t)lo, (uint32_t)hi};
}
__device__ __forceinline__ u32x2 pack8_bf8_acc(const f32x16& a, int s) {
  return pack8_bf8(a[8 * s], a[8 * s + 1], a[8 * s + 2], a[8 * s + 3], a[8 * s + 4], a[8 * s + 5], a[8 * s + 6], a[8 * s + 7]);
}
__device__ __forceinline__ u32x2 pack8_fp8_bf16(const bf16x8& f) {
  return pack8_fp8((float)f[0], (float)f[1], (float)f[2], (float)f[3], (float)f[4], (float)f[5], (float)f[6], (float)f[7]);
}
// snake'(z) = 1 + sin 2z in [0, 2] as one unsigned byte: u = round(127.5 * snake'(z)) (error <= 1 / 255, the size of the bf16
// rounding of the operand it multiplies); what the backward chain reads in stash8 mode instead of the fp16 pre-activation.
// v_cvt_pk_u8_f32 rounds to nearest even and saturates to [0, 255] (NaN -> 0): measured on gfx950, round 6.
constexpr float kSd8Scale = 127.5f, kSd8Inv = 1.0f / 127.5f;
__device__ __forceinline__ uint32_t pack4_u8(float v0, float v1, float v2, float v3) {
  uint32_t x = __builtin_amdgcn_cvt_pk_u8_f32(v0, 0u, 0u);
  x = __builtin_amdgcn_cvt_pk_u8_f32(v1, 1u, x);
  x = __builtin_amdgcn_cvt_pk_u8_f32(v2, 2u, x);
  return __builtin_amdgcn_cvt_pk_u8_f32(v3, 3u, x);
}
__device__ __forceinline__ float u8_byte_f32(uint32_t w, int j) {       // byte j of w as a float (v_cvt_f32_ubyteN)
  return (float)((w >> (8 * j)) & 255u);
}
__device__ __forceinline__ void stash8_store(void* p, const u32x2& v) {
#if NPP_STASH_NT
  __builtin_nontemporal_store(v, (u32x2*)p);
#else
  *(u32x2*)p = v;
#endif
}
// Pre-activations of the snake layers are stashed as fp16 (|z| is O(10); 11 significand bits):
// the backward chain derives snake'(z) = 1 + sin 2z from them and npp_mlp_wgrad derives the layer
// input snake(z) while staging -- ONE 16-bit array per layer instead of two.
__device__ __forceinline__ f16x8 pack_acc_f16(const f32x16& acc, int s) {
  f16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (_Float16)acc[8 * s + j];
  return r;
}
__device__ __forceinline__ void stash_store(void* p, const f16x8& v) {
#if NPP_STASH_NT
  __builtin_nontemporal_store(v, (f16x8*)p);
#else
  *(f16x8*)p = v;
#endif
}

// ---- K8: torch.optim.Adam single-tensor maths (helpers.py:164), shared by npp_loss_adam.hip and the fused Adam + re-pack
// launch of npp_api.hip.  tail (optional, handled by one extra block): a second small parameter group -- the adaptive-loss
// latents, whose gradient accumulator is consumed and cleared -- and an accumulator to clear for the next iteration, so
// that one launch replaces optimizer.step() over both groups plus zero_grad().
struct AdamTail {
  float *p, *m, *v, *g;
  int n;
  float* zero;
  int n_zero;
  // round 4: the pixel loss of the iteration left its per-block partial sums in `pl_part` (PixelLossArgs::scratch) instead of
  // adding them to g / the loss word by atomics in arrival order; this block -- the consumer of g, one launch boundary later --
  // sums them in block order: bit-reproducible and free (no ticket, no fence, no extra launch).  Nullable.
  float* pl_part;
  float* loss_cur;      // the iteration's pixel-loss accumulator (+= the partial losses), nullable
};
constexpr int kPixelLossScratch = 1024 * 8 + 8;       // [block][8] partials (7 used) + the block count in the last word
__device__ __forceinline__ void adam_tail_block(const AdamTail& tail, float step_size, float b1, float b2, float inv_sqrt_bc2,
                                                float eps) {
  const int t = threadIdx.x;
  int nb = 0;
  if (tail.pl_part) nb = (int)__float_as_uint(tail.pl_part[kPixelLossScratch - 1]);
  for (int i = t; i < tail.n; i += blockDim.x) {
    float gi = tail.g[i];
    if (i < 6)
      for (int b = 0; b < nb; ++b) gi += tail.pl_part[b * 8 + 1 + i];          // block order
    const float mi = b1 * tail.m[i] + (1.0f - b1) * gi;
    const float vi = b2 * tail.v[i] + (1.0f - b2) * gi * gi;
    tail.m[i] = mi;
    tail.v[i] = vi;
    tail.p[i] = tail.p[i] - step_size * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
    tail.g[i] = 0.0f;
  }
  if (nb > 0 && t == 64) {                               // (another wave than the latents': the two sums run side by side)
    float l = 0.0f;
    for (int b = 0; b < nb; ++b) l += tail.pl_part[b * 8];
    if (tail.loss_cur) tail.loss_cur[0] += l;
  }
  for (int i = t; i < tail.n_zero; i += blockDim.x) tail.zero[i] = 0.0f;
  __syncthreads();
  if (nb > 0 && t == 0) tail.pl_part[kPixelLossScratch - 1] = 0.0f;          // consumed
}
__device__ __forceinline__ float adam_update(float p, float& m, float& v, float g, float step_size, float b1, float b2,
                                             float inv_sqrt_bc2, float eps) {
  m = b1 * m + (1.0f - b1) * g;
  v = b2 * v + (1.0f - b2) * g * g;
  return p - step_size * (m / (sqrtf(v) * inv_sqrt_bc2 + eps));
}

// Workgroup barrier for LDS hand-offs that leaves global memory traffic in flight.
// __syncthreads() makes hipcc emit s_waitcnt vmcnt(0) first, which drains every outstanding
// stash store and weight prefetch at each of the ~30 barriers of the fused kernels (measured:
// 54 % of wave time in SQ_WAIT_ANY).  Only LDS operations (lgkmcnt) need to have completed
// before the other waves may read what this wave wrote; loads into registers are still
// waited for by the compiler at their first use.
__device__ __forceinline__ void wg_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// ---- W-format operand tiles in LDS (npp_layout.h wfmt_unit) ---------------------------
// A 16-KiB tile = 4 k-step pairs (128 features) x 2 batch tiles x 2 KiB of one 64-row
// workgroup tile, copied linearly from global memory.  wfrag_offset() is the byte offset of
// the first of the two ds_read_b64_tr_b16 that build, for this lane, the MFMA operand
// fragment "feature (lane & 31) of 32-feature tile tt, batch rows 16 t + 8 (lane >> 5) + 0..7"
// (t = 0..3 indexes the four 16-row k-steps of the 64 rows); the second read is 256 B on.
// Lane 4q+p of each 16-lane group addresses row q, feature quad p (cdna_hip_programming.md
// T10); within a 32-lane half the 32 addresses tile one 256-byte line: conflict-free.
__device__ __forceinline__ int wfrag_offset(int tt, int t, int lane) {
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3, h = g >> 1;
  return ((tt * 2 + (t >> 1)) * 8 + 4 * (t & 1) + 2 * h) * 256 + (g & 1) * 128 + (p & 1) * 64 + q * 16 + (p >> 1) * 8;
}
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
__device__ __forceinline__ bf16x8 wfrag_read(const char* tile, int off) {
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(tile + off));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(tile + off + 256));
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}

// ---- shared pieces of the fused MLP kernels (npp_mlp_fwd.hip, npp_mlp_bwd.hip) ------------
struct Lane {
  int tid, wave, lane, b, h;
  int n_wg;             // 64-row workgroup tiles of THIS image's batch (gridDim.x of a plain launch): the W-format array geometry
  int xslot, xcount;    // this workgroup's index among / the number of the image's workgroups on its XCD (weight-pack requests)
};

// ---- stacked launches (round 4): M independent images of one shape in every launch -----------------------------------------
// The reference loops its images serially (run_completion.sh:8-14); their fits are independent (own weights, Adam state, RNG),
// so M of them ride in one launch sequence: every buffer of the single-image path gets a leading image dimension with a fixed
// stride, and what differs per image and per iteration (patch source, k, Adam step) sits in a small device array the host uploads
// once per iteration (npp_stack_iter, include/npp_hip.h).  Work items are numbered so that an image's workgroups stay on 8 / M of
// the 8 XCDs (workgroups go round-robin over the XCDs by linear id -- observed, speed only): each XCD's 4-MiB L2 then keeps ONE
// image's 3.9-MB weight pack, as in the single-image launch, instead of M of them.
struct StackIter {                 // == npp_stack_iter
  int32_t active, k, comp, with_lp;
  int32_t x0, nk, same, pad1;      // x0: first image of this fit's prediction half in the stacked trunk batch; nk = n_p * k
  float step_size, inv_sqrt_bc2, pad2, pad3;
};
static_assert(sizeof(StackIter) == 48, "npp_stack_iter layout");
struct Stack {
  int32_t M;                       // 0: a plain single-image launch (blockIdx.x = work item)
  int32_t g;                       // XCDs per image (8 / M for M in {1, 2, 4, 8}), 0: images back to back in launch order
  int32_t n_items;                 // work items per image
  int32_t pad;
  const StackIter* iter;           // nullable: every image active
};
inline Stack make_stack(int M, int n_items, const void* d_iter) {
  Stack S{};
  S.M = M; S.n_items = n_items; S.iter = (const StackIter*)d_iter;
  S.g = (M == 1 || M == 2 || M == 4 || M == 8) ? 8 / M : 0;
  return S;
}
inline unsigned stack_grid(const Stack& S) {
  return S.g ? 8u * (unsigned)((S.n_items + S.g - 1) / S.g) : (unsigned)S.M * (unsigned)S.n_items;
}
// (image, item) of this workgroup; false: surplus workgroup of the rounded-up grid or an image that sits the iteration out
__device__ __forceinline__ bool stack_decode(const Stack& S, int& img, int& item, int& xslot, int& xcount) {
  const int b = (int)blockIdx.x;
  if (S.M == 0) { img = 0; item = b; xslot = b >> 3; xcount = ((int)gridDim.x + 7) >> 3; return true; }
  if (S.g) {
    const int xcd = b & 7, slot = b >> 3;
    img = xcd / S.g;
    item = slot * S.g + xcd % S.g;
    xslot = slot; xcount = (S.n_items + S.g - 1) / S.g;
  } else {
    img = b / S.n_items;
    item = b - img * S.n_items;
    xslot = item >> 3; xcount = (S.n_items + 7) >> 3;
  }
  if (item >= S.n_items) return false;
  return S.iter == nullptr || S.iter[img].active != 0;
}
__device__ __forceinline__ bf16x8 lds_frag(const char* region, int ks, int bt, int lane) {
  return *(const bf16x8*)(region + ((ks * kNB + bt) * 64 + lane) * 16);
}
__device__ __forceinline__ void lds_store_frag(char* region, int ks, int bt, int lane, const bf16x8& v) {
  *(bf16x8*)(region + ((ks * kNB + bt) * 64 + lane) * 16) = v;
}

// ---- weight stream: a rolling register ring of 4 k-steps --------------------------------
// With only Bp/64 workgroups in flight the kernel is latency-bound unless the L2 -> register
// weight stream runs ahead of the MFMAs.  Each wave keeps the weight fragments of the next
// 4 k-steps in a ring; right after the MFMAs of k-step ks have been issued, their slot is
// refilled with k-step ks+4 (16 MFMAs = 512+ cycles ahead).  The ring runs across part and
// layer boundaries (next_wp), so loads also fly under the epilogue and the barrier; START
// is the (compile-time) slot of this part's first k-step.
#ifndef NPP_RING_DEPTH
#define NPP_RING_DEPTH 4
#endif
constexpr int kRD = NPP_RING_DEPTH;     // k-steps of weight fragments in flight per wave
// The packed weights are addressed through ONE buffer descriptor (SGPRs) per kernel: a part of a
// layer is a wave-uniform offset into the pack (wptr_t, 16-byte units), the lane supplies only
// its constant 16-byte voffset, so a refill costs no vector ALU work at all (flat loads needed a
// 64-bit VALU add per k-step: ~1.2k of the ~9.6k VALU instructions a wave issued per tile).
using wrsrc_t = __amdgpu_buffer_rsrc_t;
using wptr_t = uint32_t;
constexpr wptr_t kNoW = 0xffffffffu;      // "no next part": the ring is not refilled past this part
__device__ __forceinline__ wrsrc_t make_wrsrc(const void* pack, int64_t units16) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(pack), 0, (int)(units16 * 16), 0x00020000);
}
template <int NTW>
struct WRing {
  bf16x8 w[kRD][NTW];
  wrsrc_t rsrc;
};

template <int NTW, int NT>
__device__ __forceinline__ void wslot_load(WRing<NTW>& r, int slot, wptr_t wp, int ks, int nt0, int lane) {
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
    const u32x4_t raw = __builtin_amdgcn_raw_buffer_load_b128(r.rsrc, lane * 16, (int)((wp + (uint32_t)((ks * NT + nt0 + nt) * 64)) * 16u), 0);
    r.w[slot][nt] = __builtin_bit_cast(bf16x8, raw);
  }
}
// fresh fill of the ring with k-steps 0..3 of wp (START = 0 for the consumer)
template <int NTW, int NT>
__device__ __forceinline__ void wring_fill(WRing<NTW>& r, wptr_t wp, int nt0, int lane) {
#pragma unroll
  for (int q = 0; q < kRD; ++q) wslot_load<NTW, NT>(r, q, wp, q, nt0, lane);
}

// Ring schedule positions [KS0, KS1) of a part with KSREAL real k-steps (weights at wp) padded
// to KSTOT (a multiple of 4) schedule positions, so that every part starts at ring slot 0:
// positions >= KSREAL issue no MFMA and load nothing, they only keep the refill cadence.
// Activation fragments of k-step ks sit at LDS k-step (ks - KS0 + ks_lds0) of `region`.
#ifndef NPP_SLICE_VALU_PER_GAP
#define NPP_SLICE_VALU_PER_GAP 4
#endif
struct NoHook {
  __device__ __forceinline__ void operator()(int) const {}
};
// `hook(ks - KS0)` is called once per schedule position, between the activation-fragment reads
// and the MFMAs: independent VALU / LDS work placed there (the embedding generator) issues in
// the shadow of the MFMAs of the same wave.
// SLICED: the hook emits a small, branch-free slice per position (a handful of VALU / LDS instructions); the scheduler is
// then told to place it INSIDE the position's MFMA run -- one MFMA, a share of the slice, one MFMA, ... -- instead of the
// blob-after-four-MFMAs it produces by itself (MI355X_MICROARCH.md: <= 5 single-issue instructions hide per MFMA gap).
template <int KS0, int KS1, int KSREAL, int KSTOT, int NTW, int NT, typename Hook = NoHook, bool SLICED = false>
__device__ __forceinline__ void mma_ring(f32x16 (&acc)[NTW][kNB], const char* region, int ks_lds0,
                                         wptr_t wp, wptr_t next_wp, int nt0,
                                         const Lane& L, WRing<NTW>& ring, const Hook& hook = Hook()) {
  static_assert(KSTOT % kRD == 0 && KSREAL <= KSTOT && KS1 <= KSTOT, "ring schedule");
  // activation fragments are read one k-step ahead of the MFMAs that use them (LDS latency
  // ~130+ cycles would otherwise serialise every k-step behind its own reads)
  constexpr int KSE = KS1 < KSREAL ? KS1 : KSREAL;       // real k-steps end
  bf16x8 xn[kNB];
  if (KS0 < KSE) {
#pragma unroll
    for (int bt = 0; bt < kNB; ++bt) xn[bt] = lds_frag(region, ks_lds0, bt, L.lane);
  }
#pragma unroll
  for (int ks = KS0; ks < KS1; ++ks) {
    const int slot = ks % kRD;
    if (ks < KSREAL) {
      bf16x8 x[kNB];
#pragma unroll
      for (int bt = 0; bt < kNB; ++bt) x[bt] = xn[bt];
      if (ks + 1 < KSE) {
#pragma unroll
        for (int bt = 0; bt < kNB; ++bt) xn[bt] = lds_frag(region, ks_lds0 + ks + 1 - KS0, bt, L.lane);
      }
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int bt = 0; bt < kNB; ++bt) acc[nt][bt] = mfma_bf16(ring.w[slot][nt], x[bt], acc[nt][bt]);
    }
    hook(ks - KS0);
    if (ks + kRD < KSREAL) wslot_load<NTW, NT>(ring, slot, wp, ks + kRD, nt0, L.lane);
    else if (ks + kRD >= KSTOT && next_wp != kNoW) wslot_load<NTW, NT>(ring, slot, next_wp, ks + kRD - KSTOT, nt0, L.lane);
    if (SLICED && ks < KSREAL) {
#pragma unroll
      for (int g = 0; g < NTW * kNB; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                        // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, NPP_SLICE_VALU_PER_GAP, 0);   // its share of the slice's VALU work
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                        // and of the LDS reads (slice + next fragments)
      }
    }
    asm volatile("" ::: "memory");   // pin the refill here: no hoisting of later loads
  }
}

// ---- adaptive robust loss: per-channel quantities derived from the latents --------------
// adaptive.py:146-181, distribution.py:90-114,143-169, cubic_spline.py:65-97 (see npp_loss_adam.hip)
struct ChanParams {   // per-channel quantities derived from the latents
  float alpha, c, beta, logc_plus_logz, dlogz, dalpha_dl, dc_dl;
};

__device__ inline ChanParams chan_params(float latent_alpha, float latent_scale, const float* spline,
                                         int n_knots, float x_scale) {
  ChanParams p;
  // adaptive.py:146-164 + util.py:64-72: alpha = sigmoid(l)*(hi-lo)+lo, lo=.001, hi=1.999
  const float sg = 1.0f / (1.0f + expf(-latent_alpha));
  p.alpha = sg * (1.999f - 0.001f) + 0.001f;
  p.dalpha_dl = sg * (1.0f - sg) * (1.999f - 0.001f);
  // adaptive.py:166-181 + util.py:86-95: c = (1-1e-5)*softplus(l + log(e-1)) + 1e-5
  const float xs = latent_scale + 0.54132485f;   // log(expm1(1))
  const float sp = xs > 20.0f ? xs : log1pf(expf(xs));
  p.c = (1.0f - 1e-5f) * sp + 1e-5f;
  p.dc_dl = (1.0f - 1e-5f) / (1.0f + expf(-xs));
  p.beta = fmaxf(1.1920929e-07f, fabsf(p.alpha - 2.0f));
  // distribution.py:90-114 partition_spline_curve (alpha < 4 branch)
  const float den = fabsf(p.alpha - 2.0f) + 0.25f;
  const float xc = (2.25f * p.alpha - 4.5f) / den + p.alpha + 2.0f;
  const float dxc = 0.5625f / (den * den) + 1.0f;
  // cubic_spline.py:65-97
  const float xq = xc * x_scale;
  const float* vals = spline;
  const float* tans = spline + n_knots;
  const int lo = (int)floorf(fminf(fmaxf(xq, 0.0f), (float)(n_knots - 2)));
  const float t = xq - (float)lo, t2 = t * t, t3 = t * t2;
  const float h01 = -2.0f * t3 + 3.0f * t2, h00 = 1.0f - h01, h11 = t3 - t2, h10 = h11 - t2 + t;
  const float v0 = vals[lo], v1 = vals[lo + 1], m0 = tans[lo], m1 = tans[lo + 1];
  float val = v0 * h00 + v1 * h01 + m0 * h10 + m1 * h11;
  const float dh01 = -6.0f * t2 + 6.0f * t, dh11 = 3.0f * t2 - 2.0f * t, dh10 = dh11 - 2.0f * t + 1.0f;
  float dval = (v1 - v0) * dh01 + m0 * dh10 + m1 * dh11;
  if (t < 0.0f) { val = tans[0] * t + vals[0]; dval = tans[0]; }
  else if (t > 1.0f) { val = tans[n_knots - 1] * (t - 1.0f) + vals[n_knots - 1]; dval = tans[n_knots - 1]; }
  p.logc_plus_logz = logf(p.c) + val;
  p.dlogz = dval * x_scale * dxc;
  return p;
}

// ---- a8: adaptive robust pixel loss, forward + gradients (mse_calculator.py:13-27, robust_loss_pytorch) -----------------
// One thread per row (3 channels), block-stride over `nb` blocks of 256 threads; block reduction via wave shuffles, then
// one atomicAdd per block per output (7 floats).  Shared by npp_pixel_loss and the fused patch-in launch.
struct PixelLossArgs {
  const float* pred; const float* gt; const float* mask; int64_t N;
  const float* latents; const float* spline; int n_knots; float x_scale, weight;
  float* loss_out; float* dpred; float* dlatent;
  float* scratch;      // nullable: kPixelLossScratch floats.  Given: the launch leaves its per-block partial sums there (loss_out /
                       // dlatent untouched) for the Adam launch of the iteration to add in block order (AdamTail::pl_part)
  float quad;          // 0: robust_loss_adaptive.  > 0: the non-adaptive switches of models/mse_calculator.py:19-23, both quadratic:
                       // loss = quad * mean(x^2) -- 'l2' (quad = 1) and 'robust_loss' = lossfun(x, alpha = 2, scale = 0.1) = 0.5 (x / 0.1)^2
                       // (quad = 50); latents / spline unused (may be null), no latent gradient
};
// Cross-block reduction in a FIXED order inside one launch: every block publishes its partials, the block that draws the last
// ticket sums them.  cdna_hip_programming.md Guideline 16, form R1 without fences: the (few, small) partials are written with
// relaxed agent-scope atomic stores (share_store: write-through, visible from every XCD once the storing wave's vmcnt has drained)
// and read back with relaxed agent-scope atomic loads (share_load: past this CU's L1 and this XCD's L2) -- an agent-scope release
// fence would write back the XCD's whole L2.  Call from every thread of every block after its share_store()s; true in the whole
// block that arrived last (which re-arms the counter).  The counter must be zero before the first launch that uses it.
// (Where the consumer of a sum is a LATER launch, leave the partials to it instead: AdamTail::pl_part.)
__device__ __forceinline__ void share_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float share_load(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool block_last_arriver(unsigned* counter, int nb) {
  __shared__ int s_last;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave: its partials have left for memory
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = t == (unsigned)nb - 1u;
    if (s_last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  return s_last != 0;
}
inline int pixel_loss_blocks(int64_t N) { const int64_t b = (N + 255) / 256; return (int)(b > 1024 ? 1024 : b); }
__device__ __forceinline__ void pixel_loss_body(const PixelLossArgs& a, int bid, int nb) {
  const float* __restrict__ pred = a.pred;
  const float* __restrict__ gt = a.gt;
  const float* __restrict__ mask = a.mask;
  float* __restrict__ dpred = a.dpred;
  const int64_t N = a.N;
  const float weight = a.weight;
  __shared__ ChanParams cp[3];
  __shared__ float red[4][7];
  const float quad = a.quad;
  if (threadIdx.x < 3) {
    if (quad > 0.0f) cp[threadIdx.x] = ChanParams{2.0f, 1.0f, 1.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    else cp[threadIdx.x] = chan_params(a.latents[threadIdx.x], a.latents[3 + threadIdx.x], a.spline, a.n_knots, a.x_scale);
  }
  __syncthreads();
  const float inv = 1.0f / (3.0f * (float)N);
  float acc[7] = {0, 0, 0, 0, 0, 0, 0};   // loss, dalpha[3], dc[3]
  for (int64_t r = (int64_t)bid * blockDim.x + threadIdx.x; r < N; r += (int64_t)nb * blockDim.x) {
    const float m = mask ? mask[r] : 1.0f;
    const float w = m + (1.0f - m) * 0.3f;          // mse_calculator.py:17
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const ChanParams p = cp[ch];
      const float d0 = pred[r * 3 + ch] - gt[r * 3 + ch];
      const float x = mask ? d0 * m + (1.0f - m) * d0 * 0.3f : d0;
      if (quad > 0.0f) {                            // (uniform branch) mse_calculator.py:19-23: 'l2' / 'robust_loss'
        acc[0] += quad * x * x;
        dpred[r * 3 + ch] = weight * inv * w * 2.0f * quad * x;
        continue;
      }
      const float xs = x / p.c, ssx = xs * xs;
      const float u = ssx / p.beta + 1.0f;
      const float e = 0.5f * p.alpha;
      const float lnu = logf(u);
      const float ue = expf(e * lnu);               // pow(u, e), u >= 1
      const float ue1 = ue / u;
      const float rho = (p.beta / p.alpha) * (ue - 1.0f);
      acc[0] += rho + p.logc_plus_logz;
      dpred[r * 3 + ch] = weight * inv * w * (x / (p.c * p.c)) * ue1;
      acc[1 + ch] += -(2.0f / (p.alpha * p.alpha)) * (ue - 1.0f) +
                     (p.beta / p.alpha) * ue * (0.5f * lnu + e * ssx / (p.beta * p.beta * u)) + p.dlogz;
      acc[4 + ch] += -(x * x) / (p.c * p.c * p.c) * ue1 + 1.0f / p.c;
    }
  }
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    float v = acc[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
  }
  __syncthreads();
  if (a.scratch) {
    // deterministic form: this block's seven sums, scaled as they enter the totals, for the Adam launch to add in block order
    if (threadIdx.x < 7) {
      const int k = threadIdx.x;
      const float v = red[0][k] + red[1][k] + red[2][k] + red[3][k];
      a.scratch[bid * 8 + k] = k == 0 ? weight * v * inv : k < 4 ? weight * inv * v * cp[k - 1].dalpha_dl : weight * inv * v * cp[k - 4].dc_dl;
    }
    if (bid == 0 && threadIdx.x == 8) a.scratch[kPixelLossScratch - 1] = __uint_as_float((unsigned)nb);
    return;
  }
  if (threadIdx.x < 7) {
    const int k = threadIdx.x;
    float v = red[0][k] + red[1][k] + red[2][k] + red[3][k];
    if (k == 0) atomicAdd(a.loss_out, weight * v * inv);      // the term as it enters the total (train.py:195-198: `loss = 0` under --no_pix_loss)
    else if (quad > 0.0f) {}                                  // (no latents in the quadratic forms)
    else if (k < 4) atomicAdd(a.dlatent + (k - 1), weight * inv * v * cp[k - 1].dalpha_dl);
    else atomicAdd(a.dlatent + 3 + (k - 4), weight * inv * v * cp[k - 4].dc_dl);
  }
}


}  // namespace npp
