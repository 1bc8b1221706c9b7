// npp_conv_pair.hip -- rows a11 / a13 (trunks), round 5: TWO convolution layers (+ the MaxPool2d(2,2) behind them) in ONE launch.
// The first two blocks of VGG19 (externel_lib/contextual_loss/modules/vgg.py:16-21) and VGG16
// (externel_lib/lpips/pretrained_networks.py:106-115) are  conv a -> ReLU -> conv b -> ReLU -> pool  on position-rich maps with few
// channels.  As launches of their own (npp_conv.hip) each of them is three phases that do not overlap -- every workgroup bursts its
// operands in (~11 B/clk per CU with the whole chip asking at once), multiplies for a few thousand cycles, bursts its outputs out
// (in-kernel stamps, profiles/r05_conv_phase_stamps.txt: conv1_1 spends 16 % of a wave's life in its MFMA loop, conv1_2 44 %) -- and
// the intermediate activation makes a round trip through the fabric (14.7 MB written + read at 12 x 96^2), as does the pool's input.
// Here a workgroup owns a TH x 16 tile of ONE image: it stages the (TH + 4) x 20 input window in LDS (conv a's weight fragments sit in
// registers: 18 of them for 3 -> 64 channels), computes conv a on the (TH + 2) x 18 halo-extended tile into LDS (fp16 after bias + ReLU; positions outside the image are the zero padding
// of conv b), streams conv b's weights through a double-buffered LDS stage per channel step, and finishes with bias + ReLU, the
// layer outputs the backward pass needs (only for the images that carry a gradient: n_keep), the optional fp32 tap and the 2 x 2
// max-pool by two lane exchanges (position tiles are 2 rows x 16 columns: a pool window = lanes b, b^1, b^16, b^17).
// Arithmetic: the same fp16-operand / fp32-accumulate MFMA chains in the same (channel step, tap) order as conv3x3_win_kernel, so
// conv a, conv b (where the separate launch is unsplit) and the pooled tensor are bit-identical to the three launches.
#include <stdlib.h>
#include <string.h>

#include "npp_common.h"
#include "npp_trunk_layout.h"
#ifdef NPP_DIAG
#include "npp_diag.h"          // tools/npp_diag.h: diagnostic builds only (in-kernel time stamps)
#else
#define NPP_DIAG_FIELD
#define NPP_DIAG_FILL(a) do { } while (0)
#define NPP_STAMP(a, k) do { } while (0)
#define NPP_STAMP_DRAIN() do { } while (0)
#endif

namespace npp {

struct PairArgs {
  const void* x;            // flat fp16 input, 16 CAS channels, geometry (N, H, W)
  const void* pack_a;       // forward pack of conv a [cot][ci_step][tap][64 lanes][8]
  const float* bias_a;
  const void* pack_b;
  const float* bias_b;
  void* y_a;                // flat fp16 relu(conv a) (nullable): written for images < n_keep (ReLU gates of the backward pass)
  void* y_b;                // flat fp16 relu(conv b) (nullable): images < n_keep (gates + the pool's arg-max source)
  void* y_pool;             // flat fp16 pooled output, geometry (N, H/2, W/2): all n_run images
  float* tap_b;             // optional fp32 (N, CC, H, W) copy of relu(conv b), all n_run images
  int32_t N, n_run, n_keep, H, W, Wp, S, tiles_x, tiles_y;
  int64_t nposp, pool_nposp;
  uint32_t x_bytes, pack_a_bytes, pack_b_bytes;
  // SRC form (npp_conv_pair_fwd_patch): the input window is COMPOSED here -- npp_trunk_patch_in's arithmetic (train.py:200-236) on the
  // prediction rows and the sampler's crops -- instead of read from x, and the launch's last nb_loss blocks are the adaptive pixel loss
  const float* s_pred; const float* s_fake; const float* s_fmask; const float* s_real; const float* s_rmask;
  float* s_zero;
  float s_sc[3], s_sh[3];
  int32_t s_n_p, s_k, s_comp, s_n_zero, nb_loss, n_tiles;
  PixelLossArgs pl;
  NPP_DIAG_FIELD
};

__device__ __forceinline__ f32x16 pmfma(const f16x8& a, const f16x8& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

template <int CAS, int CB, int CC, int TH>
struct PairGeom {
  static constexpr int MW = 18, MH = TH + 2, MN = MH * MW;            // conv a's output tile (halo of 1 for conv b)
  static constexpr int IW = 20, IH = TH + 4, INN = IH * IW;           // input tile (halo of 2)
  static constexpr int CHA = 2 * CAS, CHB = CB / 8, CBS = CB / 16, NCA = CB / 32, NCB = CC / 32;
  static constexpr int NPA = (MN + 31) / 32;                          // position tiles of conv a (32 consecutive tile positions)
  static constexpr int NPB = TH / 2;                                  // position tiles of conv b (2 rows x 16 columns)
  static constexpr bool WA_REGS = NCA * CAS * 9 <= 18;                // conv a's weight fragments live in registers (72 VGPRs at most)
  static constexpr int kMid = CHB * MN * 16, kIn = CHA * INN * 16, kWa = WA_REGS ? 0 : NCA * CAS * 9 * 1024, kWbStep = NCB * 9 * 1024;
  static constexpr int kR1 = (kIn + kWa) > 2 * kWbStep ? (kIn + kWa) : 2 * kWbStep;
  static constexpr int kLds = kMid + kR1;
  static constexpr int WC = NCB >= 4 ? 2 : 1, WPG = 4 / WC;           // conv b: wave grid (channel groups x position groups)
  static constexpr int CT = NCB / WC, PT = NPB / WPG;
  static_assert(NPB % WPG == 0 && NCB % WC == 0, "tile split");
};

template <int CAS, int CB, int CC, int TH, bool SRC = false>
__global__ __launch_bounds__(256, 2) void conv_pair_fwd_kernel(PairArgs a) {
  typedef PairGeom<CAS, CB, CC, TH> G;
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) char plds[];
  if (SRC && (int)blockIdx.x >= a.n_tiles) {                          // the iteration's adaptive pixel loss rides in this launch
    pixel_loss_body(a.pl, (int)blockIdx.x - a.n_tiles, a.nb_loss);
    return;
  }
  NPP_STAMP(a, 0);
  NPP_STAMP(a, 1);
  char* const lmid = plds;
  char* const lin = plds + G::kMid;
  char* const lwa = lin + G::kIn;
  char* const lwb = plds + G::kMid;                                  // (aliases lin / lwa: conv b's weight stages live after phase a)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = lane & 31, h = lane >> 5;
  // tile -> (image, tile row, tile column); consecutive blocks walk a tile row: neighbours (shared halo) run together
  const int T = blockIdx.x;
  const int per_img = a.tiles_x * a.tiles_y;
  const int n = T / per_img, tr = T - n * per_img, ty = tr / a.tiles_x, tx = tr - ty * a.tiles_x;
  const int y0 = ty * TH, x0 = tx * 16;
  const bool keep = n < a.n_keep;
  const wrsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  const wrsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.pack_a), 0, (int)a.pack_a_bytes, 0x00020000);
  const wrsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.pack_b), 0, (int)a.pack_b_bytes, 0x00020000);

  // ---- phase 0: input window -> LDS; conv a's weights -> registers (or LDS when there are many); conv b's first step -> registers
  constexpr int NI = (G::CHA * G::INN + 255) / 256, NWA = (G::NCA * CAS * 576 + 255) / 256, NWB = (G::NCB * 576 + 255) / 256;
  f16x8 Areg[G::WA_REGS ? G::NCA * CAS * 9 : 1];
  {
    u32x4_t ri[NI], rw[G::WA_REGS ? 1 : NWA];
    if constexpr (SRC) {
      // unit u < INN: position q of the window, chunk 0 = [c0 c1 c2 0 ...] (chunk 1 is zeros); the value of trunk_patch_in_kernel
      static_assert(!SRC || CAS == 1, "a composed input is a 3-channel image");
      if (blockIdx.x == 0 && tid < a.s_n_zero) a.s_zero[tid] = 0.0f;
      const int nk = a.s_n_p * a.s_k;
      const int64_t pp = (int64_t)a.H * a.W;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int u = tid + 256 * i;
        const int chunk = u / G::INN, q = u - chunk * G::INN, r = q / G::IW, c = q - r * G::IW;
        const int iy = y0 - 2 + r, ix = x0 - 2 + c;
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (_Float16)0.0f;
        if (chunk == 0 && u < G::INN && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) {
          const int64_t qq = (int64_t)iy * a.W + ix;
          const int pk = n < nk ? n : n - nk;
          const float rm = a.s_rmask[(int64_t)pk * pp + qq];
          float v[3];
          if (n < nk) {                                                                 // prediction half
            const int pi = pk / a.s_k;
            const float fm = a.s_comp ? a.s_fmask[(int64_t)pi * pp + qq] : 0.0f;
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) {
              const float pv = a.s_pred[((int64_t)pi * pp + qq) * 3 + cc];
              const float w = a.s_comp ? a.s_fake[((int64_t)pi * 3 + cc) * pp + qq] * fm + pv * (1.0f - fm) : pv;   // train.py:230-231
              v[cc] = w * rm;                                                                                       // :232-233
            }
          } else {                                                                      // real half, :235-236
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) v[cc] = a.s_real[((int64_t)pk * 3 + cc) * pp + qq] * rm;
          }
#pragma unroll
          for (int cc = 0; cc < 3; ++cc) o[cc] = trunk_in_f16(v[cc], a.s_sc[cc], a.s_sh[cc]);
        }
        ri[i] = __builtin_bit_cast(u32x4_t, o);
      }
    } else {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int u = tid + 256 * i;
      const int chunk = u / G::INN, q = u - chunk * G::INN, r = q / G::IW, c = q - r * G::IW;
      const int iy = y0 - 2 + r, ix = x0 - 2 + c;                     // image coordinates (0-based interior)
      const bool ok = u < G::CHA * G::INN && iy >= -1 && iy <= a.H && ix >= -1 && ix <= a.W;   // (the border ring is zero in memory)
      const int64_t pos = (int64_t)n * a.S + (int64_t)(iy + 1) * a.Wp + (ix + 1);
      const int off = ok ? (int)(((int64_t)chunk * a.nposp + kConvGuard + pos) * 16) : -1;
      ri[i] = __builtin_amdgcn_raw_buffer_load_b128(rX, off < 0 ? 0 : off, 0, 0);
      if (off < 0) ri[i] = u32x4_t{0u, 0u, 0u, 0u};
    }
    }
    if constexpr (G::WA_REGS) {
#pragma unroll
      for (int f = 0; f < G::NCA * CAS * 9; ++f)
        Areg[f] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rA, lane * 16, f * 1024, 0));
    } else {
#pragma unroll
      for (int i = 0; i < NWA; ++i) {
        const int u = tid + 256 * i;
        rw[i] = __builtin_amdgcn_raw_buffer_load_b128(rA, u < G::NCA * CAS * 576 ? u * 16 : 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < NI; ++i)
      if (tid + 256 * i < G::CHA * G::INN) *(u32x4_t*)(lin + (tid + 256 * i) * 16) = ri[i];
    if constexpr (!G::WA_REGS) {
#pragma unroll
      for (int i = 0; i < NWA; ++i)
        if (tid + 256 * i < G::NCA * CAS * 576) *(u32x4_t*)(lwa + (tid + 256 * i) * 16) = rw[i];
    }
  }
  u32x4_t rwb[NWB];
  auto gload_wb = [&](int ci) {                                       // unit u = (cot, tap, lane) of channel step ci
#pragma unroll
    for (int i = 0; i < NWB; ++i) {
      const int u = tid + 256 * i, cot = u / 576, r = u - cot * 576;
      rwb[i] = __builtin_amdgcn_raw_buffer_load_b128(rB, u < G::NCB * 576 ? ((cot * G::CBS + ci) * 9) * 1024 + r * 16 : 0, 0, 0);
    }
  };
  auto sstore_wb = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NWB; ++i)
      if (tid + 256 * i < G::NCB * 576) *(u32x4_t*)(lwb + buf * G::kWbStep + (tid + 256 * i) * 16) = rwb[i];
  };
  gload_wb(0);
  float bias_ar[G::NCA][16];
#pragma unroll
  for (int ct = 0; ct < G::NCA; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) bias_ar[ct][r] = a.bias_a[32 * ct + acc_row(r, h)];
  __syncthreads();
  NPP_STAMP(a, 2);

  // ---- phase a: conv a on the halo-extended tile -> LDS (fp16), and -> y_a for the tile's own positions --------------------
  for (int p = wave; p < G::NPA; p += 4) {
    const int m = 32 * p + b, mc_ = m < G::MN ? m : G::MN - 1;
    const int mr = mc_ / G::MW, mc = mc_ - mr * G::MW;
    const char* bI = lin + (h * G::INN + mr * G::IW + mc) * 16;
    f32x16 acc[G::NCA];
#pragma unroll
    for (int ct = 0; ct < G::NCA; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ct][r] = 0.0f;
    // (the window unit of k-step ks + 1 is requested before the MFMAs of k-step ks: one LDS read per NCA MFMAs, latency hidden)
    f16x8 B[2];
    auto bread = [&](int set, int ks) {
      const int ci = ks / 9, tap = ks - 9 * ci;
      B[set] = *(const f16x8*)(bI + (2 * ci * G::INN + (tap / 3) * G::IW + (tap % 3)) * 16);
    };
    bread(0, 0);
#pragma unroll
    for (int ks = 0; ks < CAS * 9; ++ks) {
      if (ks + 1 < CAS * 9) bread((ks + 1) & 1, ks + 1);
      const int ci = ks / 9, tap = ks - 9 * ci;
#pragma unroll
      for (int ct = 0; ct < G::NCA; ++ct) {
        f16x8 A;
        if constexpr (G::WA_REGS) A = Areg[(ct * CAS + ci) * 9 + tap];
        else A = *(const f16x8*)(lwa + (((ct * CAS + ci) * 9 + tap) * 64 + lane) * 16);
        acc[ct] = pmfma(A, B[ks & 1], acc[ct]);
      }
    }
    const int iy = y0 - 1 + mr, ix = x0 - 1 + mc;
    const bool inside = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;   // outside the image: conv b's zero padding
    const bool own = keep && a.y_a && inside && mr >= 1 && mr <= TH && mc >= 1 && mc <= 16;
    const int64_t pos = (int64_t)n * a.S + (int64_t)(iy + 1) * a.Wp + (ix + 1);
#pragma unroll
    for (int ct = 0; ct < G::NCA; ++ct)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int chunk = 4 * ct + 2 * s + h;
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float v = fminf(fmaxf(acc[ct][8 * s + j] + bias_ar[ct][8 * s + j], 0.0f), 65504.0f);
          o[j] = (_Float16)(inside ? v : 0.0f);
        }
        if (m < G::MN) *(f16x8*)(lmid + (chunk * G::MN + m) * 16) = o;
        if (own && m < G::MN) ((f16x8*)a.y_a)[(int64_t)chunk * a.nposp + kConvGuard + pos] = o;
      }
  }
  __syncthreads();                                                     // the mid tile is complete; input window / conv a weights are dead
  NPP_STAMP(a, 3);

  // ---- phase b: conv b, weights streamed per channel step through two LDS stages -----------------------------------------
  sstore_wb(0);
  if (G::CBS > 1) gload_wb(1);
  const int wc = wave % G::WC, wp = wave / G::WC;                     // my channel group / position group
  float bias_br[G::CT][16];
#pragma unroll
  for (int ct = 0; ct < G::CT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) bias_br[ct][r] = a.bias_b[32 * (wc * G::CT + ct) + acc_row(r, h)];
  int bbase[G::PT];
#pragma unroll
  for (int pt = 0; pt < G::PT; ++pt) bbase[pt] = (2 * (wp * G::PT + pt) + (b >> 4)) * G::MW + (b & 15);
  f32x16 acc[G::CT][G::PT];
#pragma unroll
  for (int ct = 0; ct < G::CT; ++ct)
#pragma unroll
    for (int pt = 0; pt < G::PT; ++pt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ct][pt][r] = 0.0f;
  __syncthreads();
  int buf = 0;
  for (int ci = 0; ci < G::CBS; ++ci) {
    const char* bA = lwb + buf * G::kWbStep + ((wc * G::CT) * 9 * 64 + lane) * 16;
    const char* bM = lmid + ((2 * ci + h) * G::MN) * 16;
    f16x8 A[2][G::CT], B[2][G::PT];
    auto lread = [&](int set, int tap) {
#pragma unroll
      for (int ct = 0; ct < G::CT; ++ct) A[set][ct] = *(const f16x8*)(bA + ((ct * 9 + tap) * 64) * 16);
#pragma unroll
      for (int pt = 0; pt < G::PT; ++pt) B[set][pt] = *(const f16x8*)(bM + (bbase[pt] + (tap / 3) * G::MW + (tap % 3)) * 16);
    };
    lread(0, 0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (tap + 1 < 9) lread((tap + 1) & 1, tap + 1);
#pragma unroll
      for (int ct = 0; ct < G::CT; ++ct)
#pragma unroll
        for (int pt = 0; pt < G::PT; ++pt) acc[ct][pt] = pmfma(A[tap & 1][ct], B[tap & 1][pt], acc[ct][pt]);
    }
    if (ci + 1 < G::CBS) {
      sstore_wb(buf ^ 1);                                              // (registers hold step ci + 1; that stage was read in step ci - 1)
      if (ci + 2 < G::CBS) gload_wb(ci + 2);
    }
    __syncthreads();
    buf ^= 1;
  }
  NPP_STAMP(a, 4);

  // ---- epilogue: bias + ReLU, y_b / tap for the images that need them, 2 x 2 max-pool by lane exchanges ----------------------
  const int Ho = a.H >> 1, Wo = a.W >> 1;
#pragma unroll
  for (int ct = 0; ct < G::CT; ++ct)
#pragma unroll
    for (int pt = 0; pt < G::PT; ++pt) {
      const int row = 2 * (wp * G::PT + pt) + (b >> 4), col = b & 15;
      const int iy = y0 + row, ix = x0 + col;
      const bool inside = iy < a.H && ix < a.W;
      const int64_t pos = (int64_t)n * a.S + (int64_t)(iy + 1) * a.Wp + (ix + 1);
      const int cot = wc * G::CT + ct;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int chunk = 4 * cot + 2 * s + h;
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float v = fminf(fmaxf(acc[ct][pt][8 * s + j] + bias_br[ct][8 * s + j], 0.0f), 65504.0f);
          o[j] = (_Float16)(inside ? v : 0.0f);
          if (a.tap_b && inside) {
            const int co = 32 * cot + acc_row(8 * s + j, h);
            a.tap_b[(((int64_t)n * CC + co) * a.H + iy) * a.W + ix] = v;
          }
        }
        if (keep && a.y_b && inside) ((f16x8*)a.y_b)[(int64_t)chunk * a.nposp + kConvGuard + pos] = o;
        // values are post-ReLU fp16 (>= 0; positions outside the image hold 0): the maximum of the rounded values is the rounded
        // maximum -- what maxpool2_fwd_kernel stores
        typedef int i32x4_t __attribute__((ext_vector_type(4)));
        i32x4_t w = __builtin_bit_cast(i32x4_t, o);
#pragma unroll
        for (int step = 0; step < 2; ++step) {
          i32x4_t t;
#pragma unroll
          for (int e = 0; e < 4; ++e) t[e] = __shfl_xor(w[e], step == 0 ? 1 : 16, 64);
          const f16x8 mine8 = __builtin_bit_cast(f16x8, w), other8 = __builtin_bit_cast(f16x8, t);
          w = __builtin_bit_cast(i32x4_t, __builtin_elementwise_max(mine8, other8));      // v_pk_max_f16 (values >= 0, no NaNs)
        }
        if (inside && (b & 17) == 0 && (iy >> 1) < Ho && (ix >> 1) < Wo) {
          const int64_t up = (int64_t)chunk * a.pool_nposp + kConvGuard + (int64_t)n * (Ho + 2) * (Wo + 2) +
                             (int64_t)((iy >> 1) + 1) * (Wo + 2) + ((ix >> 1) + 1);
          ((f16x8*)a.y_pool)[up] = __builtin_bit_cast(f16x8, w);
        }
      }
    }
  NPP_STAMP(a, 5);
  NPP_STAMP_DRAIN();
  NPP_STAMP(a, 6);
}

// ---- the second block: 64 -> 128 -> 128 channels (+ pool2).  conv a's weights alone are 144 KB: BOTH layers' weights stream through
// two 36-KB LDS stages, one per channel step, as ONE sequence of CAS + 8 stages (stage t in buffer t & 1; registers hold stage t + 1
// while stage t is multiplied); the intermediate tile is 10 x 18 x 128 channels = 45 KB, the input window 12 x 20 x 64 = 30 KB: 147 KB,
// one workgroup of EIGHT waves per CU (two per SIMD), an 8 x 16 output tile each: 216 workgroups at 12 x 48^2.
// Wave roles: conv a -- output-channel tile w & 3, position tiles {g, g + 2, g + 4} (g = w >> 2) of the six 32-position tiles of the
// halo-extended tile; conv b -- output-channel tile w & 3, position tiles {2 g, 2 g + 1} of the four (2 rows x 16 columns each).
template <int CAS>
__global__ __launch_bounds__(512) void conv_pair8_fwd_kernel(PairArgs a) {
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  constexpr int CB = 128, CC = 128, TH = 8;
  constexpr int MW = 18, MH = TH + 2, MN = MH * MW, IW = 20, IH = TH + 4, INN = IH * IW;
  constexpr int CHA = 2 * CAS, CHB = CB / 8, CBS = CB / 16, NCB = CC / 32;
  constexpr int kMid = CHB * MN * 16, kIn = CHA * INN * 16, kW = 4 * 9 * 1024;
  static_assert(CB / 32 == 4 && NCB == 4 && (MN + 31) / 32 == 6 && TH / 2 == 4, "wave roles are written for this shape");
  extern __shared__ __attribute__((aligned(16))) char plds[];
  NPP_STAMP(a, 0);
  NPP_STAMP(a, 1);
  char* const lmid = plds;
  char* const lin = plds + kMid;
  char* const lw = plds + kMid + kIn;                                // two weight stages
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = lane & 31, h = lane >> 5;
  const int ct = wave & 3, g = wave >> 2;
  const int T = blockIdx.x;
  const int per_img = a.tiles_x * a.tiles_y;
  const int n = T / per_img, tr = T - n * per_img, ty = tr / a.tiles_x, tx = tr - ty * a.tiles_x;
  const int y0 = ty * TH, x0 = tx * 16;
  const bool keep = n < a.n_keep;
  const wrsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  const wrsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.pack_a), 0, (int)a.pack_a_bytes, 0x00020000);
  const wrsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.pack_b), 0, (int)a.pack_b_bytes, 0x00020000);
  // the weight stream: stage t < CAS = conv a's channel step t, else conv b's step t - CAS; unit u = (cot, tap, lane)
  constexpr int NWU = (4 * 576 + 511) / 512;
  u32x4_t rw[NWU];
  auto gload = [&](int t) {
#pragma unroll
    for (int i = 0; i < NWU; ++i) {
      const int u = tid + 512 * i, cot = u / 576, r = u - cot * 576;
      const bool ok = u < 4 * 576;
      if (t < CAS) rw[i] = __builtin_amdgcn_raw_buffer_load_b128(rA, ok ? ((cot * CAS + t) * 9) * 1024 + r * 16 : 0, 0, 0);
      else rw[i] = __builtin_amdgcn_raw_buffer_load_b128(rB, ok ? ((cot * CBS + (t - CAS)) * 9) * 1024 + r * 16 : 0, 0, 0);
    }
  };
  auto sstore = [&](int t) {
#pragma unroll
    for (int i = 0; i < NWU; ++i)
      if (tid + 512 * i < 4 * 576) *(u32x4_t*)(lw + (t & 1) * kW + (tid + 512 * i) * 16) = rw[i];
  };
  // ---- phase 0: input window -> LDS, weight stage 0 -> LDS, stage 1 -> registers ----------------------------------------------
  {
    constexpr int NI = (CHA * INN + 511) / 512;
    u32x4_t ri[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int u = tid + 512 * i;
      const int chunk = u / INN, q = u - chunk * INN, r = q / IW, c = q - r * IW;
      const int iy = y0 - 2 + r, ix = x0 - 2 + c;
      const bool ok = u < CHA * INN && iy >= -1 && iy <= a.H && ix >= -1 && ix <= a.W;
      const int64_t pos = (int64_t)n * a.S + (int64_t)(iy + 1) * a.Wp + (ix + 1);
      const int off = ok ? (int)(((int64_t)chunk * a.nposp + kConvGuard + pos) * 16) : -1;
      ri[i] = __builtin_amdgcn_raw_buffer_load_b128(rX, off < 0 ? 0 : off, 0, 0);
      if (off < 0) ri[i] = u32x4_t{0u, 0u, 0u, 0u};
    }
    gload(0);
#pragma unroll
    for (int i = 0; i < NI; ++i)
      if (tid + 512 * i < CHA * INN) *(u32x4_t*)(lin + (tid + 512 * i) * 16) = ri[i];
    sstore(0);
    gload(1);
  }
  float bias_ar[16], bias_br[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    bias_ar[r] = a.bias_a[32 * ct + acc_row(r, h)];
    bias_br[r] = a.bias_b[32 * ct + acc_row(r, h)];
  }
  __syncthreads();
  NPP_STAMP(a, 2);

  // ---- phase a: conv a on the 10 x 18 halo-extended tile -> LDS (fp16), -> y_a for the tile's own positions ----------------------
  int mj[3], ibase[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    mj[j] = 32 * (g + 2 * j) + b;
    const int mc_ = mj[j] < MN ? mj[j] : MN - 1;
    ibase[j] = (mc_ / MW) * IW + (mc_ % MW);
  }
  {
    f32x16 acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    for (int ci = 0; ci < CAS; ++ci) {
      const char* bA = lw + (ci & 1) * kW + ((ct * 9) * 64 + lane) * 16;
      const char* bI = lin + ((2 * ci + h) * INN) * 16;
      f16x8 A[2], B[2][3];
      auto lread = [&](int set, int tap) {
        A[set] = *(const f16x8*)(bA + tap * 1024);
#pragma unroll
        for (int j = 0; j < 3; ++j) B[set][j] = *(const f16x8*)(bI + (ibase[j] + (tap / 3) * IW + (tap % 3)) * 16);
      };
      lread(0, 0);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        if (tap + 1 < 9) lread((tap + 1) & 1, tap + 1);
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[j] = pmfma(A[tap & 1], B[tap & 1][j], acc[j]);
      }
      sstore(ci + 1);                                               // (registers hold stage ci + 1; its buffer was read in stage ci - 1)
      gload(ci + 2);                                                // (CAS + 8 >= ci + 3 stages: always a valid stage)
      __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int m = mj[j], mc_ = m < MN ? m : MN - 1, mr = mc_ / MW, mc = mc_ - mr * MW;
      const int iy = y0 - 1 + mr, ix = x0 - 1 + mc;
      const bool inside = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      const bool own = keep && a.y_a && inside && mr >= 1 && mr <= TH && mc >= 1 && mc <= 16 && m < MN;
      const int64_t pos = (int64_t)n * a.S + (int64_t)(iy + 1) * a.Wp + (ix + 1);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int chunk = 4 * ct + 2 * s + h;
        f16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float v = fminf(fmaxf(acc[j][8 * s + k] + bias_ar[8 * s + k], 0.0f), 65504.0f);
          o[k] = (_Float16)(inside ? v : 0.0f);
        }
        if (m < MN) *(f16x8*)(lmid + (chunk * MN + m) * 16) = o;
        if (own) ((f16x8*)a.y_a)[(int64_t)chunk * a.nposp + kConvGuard + pos] = o;
      }
    }
  }
  __syncthreads();                                                   // the intermediate tile is complete (weight stage CAS is in LDS already)
  NPP_STAMP(a, 3);

  // ---- phase b: conv b, weight stages CAS .. CAS + 7 ----------------------------------------------------------------------------
  int bbase[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) bbase[j] = (2 * (2 * g + j) + (b >> 4)) * MW + (b & 15);
  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
  for (int ci = 0; ci < CBS; ++ci) {
    const int t = CAS + ci;
    const char* bA = lw + (t & 1) * kW + ((ct * 9) * 64 + lane) * 16;
    const char* bM = lmid + ((2 * ci + h) * MN) * 16;
    f16x8 A[2], B[2][2];
    auto lread = [&](int set, int tap) {
      A[set] = *(const f16x8*)(bA + tap * 1024);
#pragma unroll
      for (int j = 0; j < 2; ++j) B[set][j] = *(const f16x8*)(bM + (bbase[j] + (tap / 3) * MW + (tap % 3)) * 16);
    };
    lread(0, 0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (tap + 1 < 9) lread((tap + 1) & 1, tap + 1);
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j] = pmfma(A[tap & 1], B[tap & 1][j], acc[j]);
    }
    if (ci + 1 < CBS) {
      sstore(t + 1);
      if (ci + 2 < CBS) gload(t + 2);
    }
    __syncthreads();
  }
  NPP_STAMP(a, 4);

  // ---- epilogue: bias + ReLU, y_b / tap, 2 x 2 max-pool by lane exchanges (as conv_pair_fwd_kernel) -------------------------------
  const int Ho = a.H >> 1, Wo = a.W >> 1;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = 2 * (2 * g + j) + (b >> 4), col = b & 15;
    const int iy = y0 + row, ix = x0 + col;
    const bool inside = iy < a.H && ix < a.W;
    const int64_t pos = (int64_t)n * a.S + (int64_t)(iy + 1) * a.Wp + (ix + 1);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int chunk = 4 * ct + 2 * s + h;
      f16x8 o;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float v = fminf(fmaxf(acc[j][8 * s + k] + bias_br[8 * s + k], 0.0f), 65504.0f);
        o[k] = (_Float16)(inside ? v : 0.0f);
        if (a.tap_b && inside) {
          const int co = 32 * ct + acc_row(8 * s + k, h);
          a.tap_b[(((int64_t)n * CC + co) * a.H + iy) * a.W + ix] = v;
        }
      }
      if (keep && a.y_b && inside) ((f16x8*)a.y_b)[(int64_t)chunk * a.nposp + kConvGuard + pos] = o;
      typedef int i32x4_t __attribute__((ext_vector_type(4)));
      i32x4_t w = __builtin_bit_cast(i32x4_t, o);
#pragma unroll
      for (int step = 0; step < 2; ++step) {
        i32x4_t t2;
#pragma unroll
        for (int e = 0; e < 4; ++e) t2[e] = __shfl_xor(w[e], step == 0 ? 1 : 16, 64);
        w = __builtin_bit_cast(i32x4_t, __builtin_elementwise_max(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, t2)));
      }
      if (inside && (b & 17) == 0 && (iy >> 1) < Ho && (ix >> 1) < Wo) {
        const int64_t up = (int64_t)chunk * a.pool_nposp + kConvGuard + (int64_t)n * (Ho + 2) * (Wo + 2) +
                           (int64_t)((iy >> 1) + 1) * (Wo + 2) + ((ix >> 1) + 1);
        ((f16x8*)a.y_pool)[up] = __builtin_bit_cast(f16x8, w);
      }
    }
  }
  NPP_STAMP(a, 5);
  NPP_STAMP_DRAIN();
  NPP_STAMP(a, 6);
}

// ---- the data gradient of the first block in one launch: dL/d(pre-activation of conv b) -> conv b's data gradient -> ReLU gate of
// conv a -> conv a's data gradient -> dL/dimage (fp32, times the input scale): what HipTrunk._backward runs as npp_conv3x3 mode 1
// (mask = relu(conv a)) followed by mode 2 with the fp32 tap.  bf16 operands like those launches, same (channel step, tap) order.
// Eight waves own a 16 x 16 tile of one image: the 20 x 20 x 64-channel gradient window in LDS, both (flipped, transposed) weight
// streams through two 18-KB LDS stages as ONE sequence of stages, the gated intermediate gradient (18 x 18 x 64) in LDS.
// Wave roles: first contraction -- input-channel tile w & 1, position tiles {g, g + 4, g + 8} (g = w >> 1) of the eleven; second --
// position tile w (2 rows x 16 columns), the one 32-row tile that holds the 3 image channels.
struct PairDgradArgs {
  const void* dz;           // flat bf16 dL/d(pre-activation of conv b), CM channels, geometry (N, H, W)
  const void* pack_b;       // data-gradient pack of conv b  [cit (CM / 32)][co_step (CM / 16)][tap][64][8]
  const void* gate;         // flat fp16 relu(conv a): the ReLU gate of the intermediate gradient
  const void* pack_a;       // data-gradient pack of conv a  [1][co_step (CM / 16)][tap][64][8]
  float* dimg;              // fp32 (N, 3, H, W)
  float scale[4];
  int32_t N, n_run, H, W, Wp, S, tiles_x, tiles_y;
  int64_t nposp;
  uint32_t dz_bytes, pack_a_bytes, pack_b_bytes;
  NPP_DIAG_FIELD
};

__device__ __forceinline__ f32x16 pmfma(const bf16x8& a, const bf16x8& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__global__ __launch_bounds__(512) void conv_pair_dgrad_kernel(PairDgradArgs a) {
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  constexpr int CM = 64, TH = 16;
  constexpr int MW = 18, MH = TH + 2, MN = MH * MW, IW = 20, IH = TH + 4, INN = IH * IW;
  constexpr int CH = CM / 8, KS = CM / 16, NPA = (MN + 31) / 32;      // 8 chunks, 4 channel steps per contraction, 11 position tiles
  constexpr int kMid = CH * MN * 16, kIn = CH * INN * 16, kW = 2 * 9 * 1024;
  extern __shared__ __attribute__((aligned(16))) char plds[];
  NPP_STAMP(a, 0);
  NPP_STAMP(a, 1);
  char* const lmid = plds;
  char* const lin = plds + kMid;
  char* const lw = plds + kMid + kIn;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = lane & 31, h = lane >> 5;
  const int ct = wave & 1, g = wave >> 1;
  const int T = blockIdx.x;
  const int per_img = a.tiles_x * a.tiles_y;
  const int n = T / per_img, tr = T - n * per_img, ty = tr / a.tiles_x, tx = tr - ty * a.tiles_x;
  const int y0 = ty * TH, x0 = tx * 16;
  const wrsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.dz), 0, (int)a.dz_bytes, 0x00020000);
  const wrsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.pack_b), 0, (int)a.pack_b_bytes, 0x00020000);
  const wrsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.pack_a), 0, (int)a.pack_a_bytes, 0x00020000);
  // weight stream: stage t < KS = conv b's data-gradient step t (two 32-channel tiles), else conv a's step t - KS (one tile)
  constexpr int NWU = (2 * 576 + 511) / 512;
  u32x4_t rw[NWU];
  auto gload = [&](int t) {
#pragma unroll
    for (int i = 0; i < NWU; ++i) {
      const int u = tid + 512 * i, cot = u / 576, r = u - cot * 576;
      if (t < KS) rw[i] = __builtin_amdgcn_raw_buffer_load_b128(rB, u < 2 * 576 ? ((cot * KS + t) * 9) * 1024 + r * 16 : 0, 0, 0);
      else rw[i] = __builtin_amdgcn_raw_buffer_load_b128(rA, u < 576 ? ((t - KS) * 9) * 1024 + r * 16 : 0, 0, 0);
    }
  };
  auto sstore = [&](int t) {
#pragma unroll
    for (int i = 0; i < NWU; ++i)
      if (tid + 512 * i < (t < KS ? 2 : 1) * 576) *(u32x4_t*)(lw + (t & 1) * kW + (tid + 512 * i) * 16) = rw[i];
  };
  {
    constexpr int NI = (CH * INN + 511) / 512;
    u32x4_t ri[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int u = tid + 512 * i;
      const int chunk = u / INN, q = u - chunk * INN, r = q / IW, c = q - r * IW;
      const int iy = y0 - 2 + r, ix = x0 - 2 + c;
      const bool ok = u < CH * INN && iy >= -1 && iy <= a.H && ix >= -1 && ix <= a.W;
      const int64_t pos = (int64_t)n * a.S + (int64_t)(iy + 1) * a.Wp + (ix + 1);
      const int off = ok ? (int)(((int64_t)chunk * a.nposp + kConvGuard + pos) * 16) : -1;
      ri[i] = __builtin_amdgcn_raw_buffer_load_b128(rX, off < 0 ? 0 : off, 0, 0);
      if (off < 0) ri[i] = u32x4_t{0u, 0u, 0u, 0u};
    }
    gload(0);
#pragma unroll
    for (int i = 0; i < NI; ++i)
      if (tid + 512 * i < CH * INN) *(u32x4_t*)(lin + (tid + 512 * i) * 16) = ri[i];
    sstore(0);
    gload(1);
  }
  // first contraction: my position tiles, and the ReLU-gate units of their positions (requested now, used in the epilogue)
  int mj[3], ibase[3];
  bool insj[3];
  f16x8 gate[3][2];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int p = g + 4 * j;
    mj[j] = p < NPA ? 32 * p + b : MN;                                // (group 3 has two tiles)
    const int mc_ = mj[j] < MN ? mj[j] : MN - 1, mr = mc_ / MW, mc = mc_ - mr * MW;
    ibase[j] = mr * IW + mc;
    const int iy = y0 - 1 + mr, ix = x0 - 1 + mc;
    insj[j] = mj[j] < MN && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
    const int64_t pos = (int64_t)n * a.S + (int64_t)(iy + 1) * a.Wp + (ix + 1);
#pragma unroll
    for (int s = 0; s < 2; ++s)
      if (insj[j]) gate[j][s] = ((const f16x8*)a.gate)[(int64_t)(4 * ct + 2 * s + h) * a.nposp + kConvGuard + pos];
  }
  __syncthreads();
  NPP_STAMP(a, 2);
  {
    f32x16 acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    for (int ci = 0; ci < KS; ++ci) {
      const char* bA = lw + (ci & 1) * kW + ((ct * 9) * 64 + lane) * 16;
      const char* bI = lin + ((2 * ci + h) * INN) * 16;
      bf16x8 A[2], B[2][3];
      auto lread = [&](int set, int tap) {
        A[set] = *(const bf16x8*)(bA + tap * 1024);
#pragma unroll
        for (int j = 0; j < 3; ++j) B[set][j] = *(const bf16x8*)(bI + (ibase[j] + (tap / 3) * IW + (tap % 3)) * 16);
      };
      lread(0, 0);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        if (tap + 1 < 9) lread((tap + 1) & 1, tap + 1);
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[j] = pmfma(A[tap & 1], B[tap & 1][j], acc[j]);
      }
      sstore(ci + 1);
      gload(ci + 2);                                                // (2 KS >= ci + 3 stages)
      __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (__bf16)((insj[j] && (float)gate[j][s][k] > 0.0f) ? acc[j][8 * s + k] : 0.0f);
        if (mj[j] < MN) *(bf16x8*)(lmid + ((4 * ct + 2 * s + h) * MN + mj[j]) * 16) = o;
      }
  }
  __syncthreads();
  NPP_STAMP(a, 3);
  // second contraction: position tile `wave`, the image-channel tile
  const int bbase = (2 * wave + (b >> 4)) * MW + (b & 15);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  for (int ci = 0; ci < KS; ++ci) {
    const int t = KS + ci;
    const char* bA = lw + (t & 1) * kW + lane * 16;
    const char* bM = lmid + ((2 * ci + h) * MN) * 16;
    bf16x8 A[2], B[2];
    auto lread = [&](int set, int tap) {
      A[set] = *(const bf16x8*)(bA + tap * 1024);
      B[set] = *(const bf16x8*)(bM + (bbase + (tap / 3) * MW + (tap % 3)) * 16);
    };
    lread(0, 0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (tap + 1 < 9) lread((tap + 1) & 1, tap + 1);
      acc = pmfma(A[tap & 1], B[tap & 1], acc);
    }
    if (ci + 1 < KS) {
      sstore(t + 1);
      if (ci + 2 < KS) gload(t + 2);
    }
    __syncthreads();
  }
  NPP_STAMP(a, 4);
  {
    const int row = 2 * wave + (b >> 4), col = b & 15;
    const int iy = y0 + row, ix = x0 + col;
    if (h == 0 && iy < a.H && ix < a.W) {                             // rows 0..2 of the accumulator tile = registers 0..2 of lane half 0
#pragma unroll
      for (int c = 0; c < 3; ++c) a.dimg[(((int64_t)n * 3 + c) * a.H + iy) * a.W + ix] = acc[c] * a.scale[c];
    }
  }
  NPP_STAMP(a, 5);
  NPP_STAMP_DRAIN();
  NPP_STAMP(a, 6);
}

// (Round 5 also built the SECOND block's data gradient + pool1's backward as one launch -- eight waves per 8 x 16 tile, ONE 36-KB weight
// stage beside a 60-KB gradient window and a 45-KB intermediate tile, two barriers per stage: 108 workgroups at six 48 x 48 images =
// 42 % of the CUs; measured SLOWER, data gradient 82.3 -> 85.2 us, and no gain at 4 / 8 stacked images: removed,
// profiles/r05_conv_phase_stamps.txt.)

}  // namespace npp

using namespace npp;

#ifdef NPP_DIAG
namespace npp { extern unsigned long long* g_diag_stamps; extern long long g_diag_n; }
#endif

template <int CAS, int CB, int CC, int TH>
static int pair_launch(PairArgs& a, hipStream_t s) {
  typedef PairGeom<CAS, CB, CC, TH> G;
  static_assert(G::kLds <= 80 * 1024, "two workgroups per CU");
  a.tiles_x = (a.W + 15) / 16;
  a.tiles_y = (a.H + TH - 1) / TH;
  static SmemOnce once;
  if (!smem_attr(once, (const void*)conv_pair_fwd_kernel<CAS, CB, CC, TH>, G::kLds)) { set_error("npp_conv_pair_fwd: smem attribute"); return NPP_ERR_LAUNCH; }
  hipLaunchKernelGGL((conv_pair_fwd_kernel<CAS, CB, CC, TH>), dim3((unsigned)(a.n_run * a.tiles_x * a.tiles_y)), dim3(256), G::kLds, s, a);
  return check_launch("npp_conv_pair_fwd");
}

// Shapes the fused pair is built for: (Cin, Cmid, Cout) = (16, 64, 64) and (64, 128, 128).  H, W even (the pool), any size.
extern "C" int npp_conv_pair_fwd_ok(int H, int W, int Cin, int Cmid, int Cout) {
  if (H < 2 || W < 2 || (H & 1) || (W & 1)) return 0;
  return ((Cin == 16 && Cmid == 64 && Cout == 64) || (Cin == 64 && Cmid == 128 && Cout == 128)) ? 1 : 0;
}

extern "C" int npp_conv_pair_fwd(const void* d_x, int N_total, int n_run, int n_keep, int H, int W, int Cin, int Cmid, int Cout,
                                 const void* d_pack_a, const float* d_bias_a, const void* d_pack_b, const float* d_bias_b,
                                 void* d_y_a, void* d_y_b, void* d_y_pool, float* d_tap_b, void* stream) {
  if (!npp_conv_pair_fwd_ok(H, W, Cin, Cmid, Cout)) {
    set_error("npp_conv_pair_fwd: not a shape of the fused pair (H=%d W=%d %d -> %d -> %d; npp_conv_pair_fwd_ok)", H, W, Cin, Cmid, Cout);
    return NPP_ERR_ARG;
  }
  if (N_total < 1 || n_run < 1 || n_run > N_total || n_keep < 0 || W + 3 > kConvGuard || !d_x || !d_pack_a || !d_bias_a || !d_pack_b ||
      !d_bias_b || !d_y_pool || (n_keep > 0 && (!d_y_a || !d_y_b))) {
    set_error("npp_conv_pair_fwd: bad argument (N=%d n_run=%d n_keep=%d)", N_total, n_run, n_keep);
    return NPP_ERR_ARG;
  }
  if (conv_nposp(N_total, H, W) * 16 * 64 > 0x7fffffffLL) { set_error("npp_conv_pair_fwd: tensor too large for one launch"); return NPP_ERR_ARG; }
  PairArgs a{};
  a.x = d_x; a.pack_a = d_pack_a; a.bias_a = d_bias_a; a.pack_b = d_pack_b; a.bias_b = d_bias_b;
  a.y_a = d_y_a; a.y_b = d_y_b; a.y_pool = d_y_pool; a.tap_b = d_tap_b;
  a.N = N_total; a.n_run = n_run; a.n_keep = n_keep > n_run ? n_run : n_keep; a.H = H; a.W = W; a.Wp = W + 2; a.S = (H + 2) * (W + 2);
  a.nposp = conv_nposp(N_total, H, W);
  a.pool_nposp = conv_nposp(N_total, H / 2, W / 2);
  a.x_bytes = (uint32_t)((int64_t)(Cin / 8) * a.nposp * 16);
  a.pack_a_bytes = (uint32_t)((int64_t)((Cmid + 31) / 32) * (Cin / 16) * 9 * 1024);
  a.pack_b_bytes = (uint32_t)((int64_t)((Cout + 31) / 32) * (Cmid / 16) * 9 * 1024);
#ifdef NPP_DIAG
  a.stamps = npp::g_diag_stamps; a.stamps_n = npp::g_diag_n;
#endif
  if (Cin == 16) return pair_launch<1, 64, 64, 16>(a, (hipStream_t)stream);
  // the second block: 8 x 16 tiles, eight waves, both weight streams through LDS
  constexpr int kLds8 = 16 * 180 * 16 + 8 * 240 * 16 + 2 * 4 * 9 * 1024;
  a.tiles_x = (W + 15) / 16;
  a.tiles_y = (H + 7) / 8;
  static SmemOnce once8;
  if (!smem_attr(once8, (const void*)conv_pair8_fwd_kernel<4>, kLds8)) { set_error("npp_conv_pair_fwd: smem attribute"); return NPP_ERR_LAUNCH; }
  hipLaunchKernelGGL((conv_pair8_fwd_kernel<4>), dim3((unsigned)(a.n_run * a.tiles_x * a.tiles_y)), dim3(512), kLds8, (hipStream_t)stream, a);
  return check_launch("npp_conv_pair_fwd");
}


// Data gradient of the first block (H, W: any size; 3 -> Cmid -> Cmid forward channels with Cmid = 64).
extern "C" int npp_conv_pair_dgrad_ok(int H, int W, int Cmid) { return (H >= 1 && W >= 1 && Cmid == 64) ? 1 : 0; }

extern "C" int npp_conv_pair_dgrad(const void* d_dz_b, int N_total, int n_run, int H, int W, int Cmid, const void* d_pack_b_bwd,
                                   const void* d_y_a, const void* d_pack_a_bwd, float* d_dimg, const float scale[3], void* stream) {
  if (!npp_conv_pair_dgrad_ok(H, W, Cmid)) { set_error("npp_conv_pair_dgrad: not a shape of the fused pair (Cmid=%d)", Cmid); return NPP_ERR_ARG; }
  if (N_total < 1 || n_run < 1 || n_run > N_total || W + 3 > kConvGuard || !d_dz_b || !d_pack_b_bwd || !d_y_a || !d_pack_a_bwd || !d_dimg || !scale) {
    set_error("npp_conv_pair_dgrad: bad argument (N=%d n_run=%d)", N_total, n_run);
    return NPP_ERR_ARG;
  }
  if (conv_nposp(N_total, H, W) * 16 * 64 > 0x7fffffffLL) { set_error("npp_conv_pair_dgrad: tensor too large for one launch"); return NPP_ERR_ARG; }
  PairDgradArgs a{};
  a.dz = d_dz_b; a.pack_b = d_pack_b_bwd; a.gate = d_y_a; a.pack_a = d_pack_a_bwd; a.dimg = d_dimg;
  for (int i = 0; i < 3; ++i) a.scale[i] = scale[i];
  a.N = N_total; a.n_run = n_run; a.H = H; a.W = W; a.Wp = W + 2; a.S = (H + 2) * (W + 2);
  a.nposp = conv_nposp(N_total, H, W);
  a.dz_bytes = (uint32_t)((int64_t)(Cmid / 8) * a.nposp * 16);
  a.pack_b_bytes = (uint32_t)((int64_t)(Cmid / 32) * (Cmid / 16) * 9 * 1024);
  a.pack_a_bytes = (uint32_t)((int64_t)1 * (Cmid / 16) * 9 * 1024);
  a.tiles_x = (W + 15) / 16; a.tiles_y = (H + 15) / 16;
#ifdef NPP_DIAG
  a.stamps = npp::g_diag_stamps; a.stamps_n = npp::g_diag_n;
#endif
  constexpr int kLds = 8 * 324 * 16 + 8 * 400 * 16 + 2 * 2 * 9 * 1024;
  static SmemOnce once;
  if (!smem_attr(once, (const void*)conv_pair_dgrad_kernel, kLds)) { set_error("npp_conv_pair_dgrad: smem attribute"); return NPP_ERR_LAUNCH; }
  hipLaunchKernelGGL(conv_pair_dgrad_kernel, dim3((unsigned)(n_run * a.tiles_x * a.tiles_y)), dim3(512), kLds, (hipStream_t)stream, a);
  return check_launch("npp_conv_pair_dgrad");
}


// npp_trunk_patch_in_loss + npp_conv_pair_fwd in ONE launch (first block only): the 2 n_p k patches of P x P are composed inside the
// pair's input staging from the prediction rows and the sampler's crops (train.py:200-236, normalised with scale / shift) -- the
// flat input tensor is never written -- and the launch's last blocks are the adaptive pixel loss.  For iterations that need no fp32
// copy of the batch (no LPIPS / style branch).  Results: bit-identical to the two launches.
extern "C" int npp_conv_pair_fwd_patch(const float* d_pred_rows, const float* d_fake, const float* d_fmask, const float* d_real,
                                       const float* d_rmask, int n_p, int k, int P, int comp, const float scale[3], const float shift[3],
                                       float* d_zero, int n_zero, const npp_pixel_loss_args* loss, int n_keep, int Cmid, int Cout,
                                       const void* d_pack_a, const float* d_bias_a, const void* d_pack_b, const float* d_bias_b,
                                       void* d_y_a, void* d_y_b, void* d_y_pool, float* d_tap_b, void* stream) {
  const char* who = "npp_conv_pair_fwd_patch";
  if (n_p < 1 || k < 1 || n_zero < 0 || n_zero > 256 || !(Cmid == 64 && Cout == 64) || P < 2 || (P & 1)) {
    set_error("%s: bad n_p=%d k=%d n_zero=%d P=%d or not the first block (%d -> %d)", who, n_p, k, n_zero, P, Cmid, Cout);
    return NPP_ERR_ARG;
  }
  const int N = 2 * n_p * k;
  if (!d_pred_rows || !d_real || !d_rmask || !scale || !shift || (comp && (!d_fake || !d_fmask)) || (n_zero && !d_zero) || !d_pack_a ||
      !d_bias_a || !d_pack_b || !d_bias_b || !d_y_pool || n_keep < 0 || (n_keep > 0 && (!d_y_a || !d_y_b)) || P + 3 > kConvGuard) {
    set_error("%s: null pointer / bad size", who);
    return NPP_ERR_ARG;
  }
  if (loss && (loss->N <= 0 || !loss->pred || !loss->gt || !loss->loss || !loss->dpred || loss->quad < 0.0f ||
               (loss->quad == 0.0f && (!loss->latents || !loss->spline || !loss->dlatent || loss->n_knots < 2)))) {
    set_error("%s: bad pixel-loss arguments (N=%lld)", who, (long long)loss->N);
    return NPP_ERR_ARG;
  }
  if (conv_nposp(N, P, P) * 16 * 64 > 0x7fffffffLL) { set_error("%s: tensor too large for one launch", who); return NPP_ERR_ARG; }
  PairArgs a{};
  a.pack_a = d_pack_a; a.bias_a = d_bias_a; a.pack_b = d_pack_b; a.bias_b = d_bias_b;
  a.y_a = d_y_a; a.y_b = d_y_b; a.y_pool = d_y_pool; a.tap_b = d_tap_b;
  a.N = N; a.n_run = N; a.n_keep = n_keep > N ? N : n_keep; a.H = P; a.W = P; a.Wp = P + 2; a.S = (P + 2) * (P + 2);
  a.nposp = conv_nposp(N, P, P);
  a.pool_nposp = conv_nposp(N, P / 2, P / 2);
  a.x = d_pack_a; a.x_bytes = 16;                                    // (unused in the SRC form: a valid descriptor)
  a.pack_a_bytes = (uint32_t)(2 * 1 * 9 * 1024);
  a.pack_b_bytes = (uint32_t)(2 * 4 * 9 * 1024);
  a.s_pred = d_pred_rows; a.s_fake = d_fake; a.s_fmask = d_fmask; a.s_real = d_real; a.s_rmask = d_rmask; a.s_zero = d_zero;
  for (int i = 0; i < 3; ++i) { a.s_sc[i] = scale[i]; a.s_sh[i] = shift[i]; }
  a.s_n_p = n_p; a.s_k = k; a.s_comp = comp; a.s_n_zero = n_zero;
  a.tiles_x = (P + 15) / 16; a.tiles_y = (P + 15) / 16;
  a.n_tiles = N * a.tiles_x * a.tiles_y;
  a.nb_loss = loss ? pixel_loss_blocks(loss->N) : 0;
  if (loss) a.pl = PixelLossArgs{loss->pred, loss->gt, loss->mask, loss->N, loss->latents, loss->spline, loss->n_knots, loss->x_scale,
                                 loss->weight, loss->loss, loss->dpred, loss->dlatent, loss->scratch, loss->quad};
#ifdef NPP_DIAG
  a.stamps = npp::g_diag_stamps; a.stamps_n = npp::g_diag_n;
#endif
  typedef PairGeom<1, 64, 64, 16> G;
  static SmemOnce once;
  if (!smem_attr(once, (const void*)conv_pair_fwd_kernel<1, 64, 64, 16, true>, G::kLds)) { set_error("%s: smem attribute", who); return NPP_ERR_LAUNCH; }
  hipLaunchKernelGGL((conv_pair_fwd_kernel<1, 64, 64, 16, true>), dim3((unsigned)(a.n_tiles + a.nb_loss)), dim3(256), G::kLds, (hipStream_t)stream, a);
  return check_launch(who);
}
