// npp_api.hip -- host-side glue of libnpp_hip.so: error plumbing, the parameter table,
// weight packing (device kernel + host twin), and the MFMA lane-map self test.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <vector>

#include "npp_common.h"

namespace npp {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return NPP_ERR_LAUNCH;
  }
  return NPP_OK;
}

bool smem_attr(SmemOnce& once, const void* fn, int bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  const unsigned long long bit = 1ull << (dev & 63);
  if (__atomic_load_n(&once.done, __ATOMIC_RELAXED) & bit) return true;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
  __atomic_fetch_or(&once.done, bit, __ATOMIC_RELAXED);
  return true;
}

static const char* kNames[kNumLayers] = {
    "periodic_linears.0", "periodic_linears.1", "periodic_linears.2", "periodic_linears.3",
    "periodic_linears.4", "periodic_linears.5", "periodic_linears.6", "periodic_linears.7",
    "feature_linear1",    "scale_linears.0",    "feature_linear2",    "pos_linears.0",
    "rgb_linear"};
static char g_name_buf[kNumLayers * 2][40];

// ---- weight packing --------------------------------------------------------------
// forward pack element (l, ks, nt, lane, j) = W_l[nt*32 + (lane&31)][fwd_col(...)]
// backward pack element (v, ns, kt, lane, j) = W_layer[16 ns + perm16(h, j)][col0 + kt*32 + (lane&31)]
NPP_HD float fwd_pack_value(const float* P, const NetDesc& d, int l, int ks, int nt, int lane, int j) {
  const int col = fwd_col(d.K, l, ks, lane >> 5, j);
  if (col < 0) return 0.0f;
  const int row = nt * 32 + (lane & 31);
  return P[d.w_off[l] + (int64_t)row * d.n_in[l] + col];
}
NPP_HD float bwd_pack_value(const float* P, const NetDesc& d, const BwdDesc& b, int v, int ns, int kt,
                            int lane, int j) {
  const int l = b.layer[v];
  const int n = 16 * ns + perm16(lane >> 5, j);
  const int col = b.col0[v] + kt * 32 + (lane & 31);
  return P[d.w_off[l] + (int64_t)n * d.n_in[l] + col];
}

// Decode a forward-pack 16-byte unit index into (l, ks, nt, lane).
NPP_HD bool fwd_unit(const NetDesc& d, int64_t u, int& l, int& ks, int& nt, int& lane) {
  for (l = 0; l < kNumLayers; ++l) {
    if (!d.present[l] || d.nt_f[l] == 0) continue;
    const int64_t n = (int64_t)d.ks_f[l] * d.nt_f[l] * 64;
    if (u >= d.wf_off[l] && u < d.wf_off[l] + n) {
      int64_t r = u - d.wf_off[l];
      lane = (int)(r & 63);
      r >>= 6;
      nt = (int)(r % d.nt_f[l]);
      ks = (int)(r / d.nt_f[l]);
      return true;
    }
  }
  return false;
}
NPP_HD bool bwd_unit(const BwdDesc& b, int64_t u, int& v, int& ns, int& kt, int& lane) {
  for (v = 0; v < kNumBwd; ++v) {
    if (!b.present[v]) continue;
    const int64_t n = (int64_t)b.ns[v] * kNT * 64;
    if (u >= b.off16[v] && u < b.off16[v] + n) {
      int64_t r = u - b.off16[v];
      lane = (int)(r & 63);
      r >>= 6;
      kt = (int)(r % kNT);
      ns = (int)(r / kNT);
      return true;
    }
  }
  return false;
}

__global__ void pack_weights_kernel(const float* __restrict__ P, bf16x8* __restrict__ wf,
                                    bf16x8* __restrict__ wb, NetDesc d, BwdDesc b, int64_t nf,
                                    int64_t nb) {
  const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (u < nf) {
    int l, ks, nt, lane;
    if (fwd_unit(d, u, l, ks, nt, lane)) {
      bf16x8 r;
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] = (__bf16)fwd_pack_value(P, d, l, ks, nt, lane, j);
      wf[u] = r;
    }
  } else if (u < nf + nb) {
    const int64_t ub = u - nf;
    int v, ns, kt, lane;
    if (bwd_unit(b, ub, v, ns, kt, lane)) {
      bf16x8 r;
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] = (__bf16)bwd_pack_value(P, d, b, v, ns, kt, lane, j);
      wb[ub] = r;
    }
  }
}

// ---- K8 + re-pack in ONE launch (round 3): optimizer.step() over the blob in parameter order (the split-K reduction of the
// weight gradient included, like npp_adam_step) and, for every updated weight, its bf16 image scattered into the forward
// and the backward MFMA pack through the inverse maps of npp_layout.h (fwd_pack_pos / bwd_pack_pos): 2-byte stores, 8 per
// thread, into packs that stay L2-resident -- against a second launch that gathered the whole blob again with row-strided
// 4-byte loads (12 us at c2).  The packs' padding elements are written once by npp_pack_weights and never again.
struct AdamPackArgs {
  float *p, *m, *v;
  const float* g;
  int64_t n;
  int32_t n_slabs;
  int64_t slab_stride;
  float step_size, b1, b2, inv_sqrt_bc2, eps;
  AdamTail tail;
  __bf16 *wf, *wb;
  uint32_t magic[kNumLayers];      // floor(2^32 / n_in) + 1: r / n_in = umulhi(r, magic) for r < 2^20
  // stacked launch (npp_adam_step_net_pack_stack): blockIdx.y = image; blobs / slabs / packs / latents at + y * stride, the
  // step's bias corrections from iter[y] (an image that skipped iterations has its own step count and learning rate)
  const StackIter* iter;
  int64_t blob_stride, slab_img_stride, wf_stride, wb_stride;      // floats, floats, bf16 elements, bf16 elements
  int32_t lat_stride, zero_stride;                                 // floats
  int64_t pl_stride;
};

__global__ __launch_bounds__(256) void adam_pack_kernel(AdamPackArgs a_in, NetDesc d_arg, BwdDesc b_arg) {
  AdamPackArgs a = a_in;
  if (a.iter) {
    const int y = (int)blockIdx.y;
    const StackIter it = a.iter[y];
    if (!it.active) return;
    a.p += y * a.blob_stride; a.m += y * a.blob_stride; a.v += y * a.blob_stride;
    a.g += y * a.slab_img_stride;
    a.wf += y * a.wf_stride; a.wb += y * a.wb_stride;
    a.step_size = it.step_size; a.inv_sqrt_bc2 = it.inv_sqrt_bc2;
    a.tail.p += y * a.lat_stride; a.tail.m += y * a.lat_stride; a.tail.v += y * a.lat_stride; a.tail.g += y * a.lat_stride;
    a.tail.zero += y * a.zero_stride;
    if (a.tail.pl_part) a.tail.pl_part += y * a.pl_stride;
    if (a.tail.loss_cur) a.tail.loss_cur += y * a.zero_stride;
  }
  // the descriptors are indexed with per-lane layer numbers: LDS copies (filled with compile-time indices, so that the
  // by-value kernel arguments never need a scratch copy)
  __shared__ NetDesc d;
  __shared__ BwdDesc b;
  __shared__ uint32_t magic[kNumLayers];
  if (threadIdx.x == 0) {
    const uint32_t* s1 = (const uint32_t*)&d_arg;
    uint32_t* t1 = (uint32_t*)&d;
#pragma unroll
    for (int i = 0; i < (int)(sizeof(NetDesc) / 4); ++i) t1[i] = s1[i];
    const uint32_t* s2 = (const uint32_t*)&b_arg;
    uint32_t* t2 = (uint32_t*)&b;
#pragma unroll
    for (int i = 0; i < (int)(sizeof(BwdDesc) / 4); ++i) t2[i] = s2[i];
#pragma unroll
    for (int i = 0; i < kNumLayers; ++i) magic[i] = a.magic[i];
  }
  __syncthreads();
  if ((int64_t)blockIdx.x * blockDim.x * 4 >= a.n) {      // the extra block: latents + accumulators
    adam_tail_block(a.tail, a.step_size, a.b1, a.b2, a.inv_sqrt_bc2, a.eps);
    return;
  }
  typedef float vec_t __attribute__((ext_vector_type(4)));
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= a.n) return;
  const int cnt = (int)(a.n - i < 4 ? a.n - i : 4);       // 4, or the last n % 4 parameters (the count is odd)
  float pn[4], mi[4], vi[4], gi[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (cnt == 4) {
    auto ld = [&](int sl) { return *(const vec_t*)(a.g + (int64_t)sl * a.slab_stride + i); };
    vec_t gs = (vec_t)(0.0f);
    int sl = 0;
    for (; sl + 4 <= a.n_slabs; sl += 4) {              // same summation order as adam_kernel<4>: bit-identical
      const vec_t a0 = ld(sl), a1 = ld(sl + 1), a2 = ld(sl + 2), a3 = ld(sl + 3);
      gs += a0; gs += a1; gs += a2; gs += a3;
    }
    for (; sl < a.n_slabs; ++sl) gs += ld(sl);
    const vec_t m4 = *(const vec_t*)(a.m + i), v4 = *(const vec_t*)(a.v + i), p4 = *(const vec_t*)(a.p + i);
#pragma unroll
    for (int e = 0; e < 4; ++e) { gi[e] = gs[e]; mi[e] = m4[e]; vi[e] = v4[e]; pn[e] = p4[e]; }
  } else {
    for (int e = 0; e < cnt; ++e) {
      float ge = 0.0f;
      for (int sl = 0; sl < a.n_slabs; ++sl) ge += a.g[(int64_t)sl * a.slab_stride + i + e];
      gi[e] = ge; mi[e] = a.m[i + e]; vi[e] = a.v[i + e]; pn[e] = a.p[i + e];
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (e < cnt) pn[e] = adam_update(pn[e], mi[e], vi[e], gi[e], a.step_size, a.b1, a.b2, a.inv_sqrt_bc2, a.eps);
  if (cnt == 4) {
    *(vec_t*)(a.m + i) = (vec_t){mi[0], mi[1], mi[2], mi[3]};
    *(vec_t*)(a.v + i) = (vec_t){vi[0], vi[1], vi[2], vi[3]};
    *(vec_t*)(a.p + i) = (vec_t){pn[0], pn[1], pn[2], pn[3]};
  } else {
    for (int e = 0; e < cnt; ++e) { a.m[i + e] = mi[e]; a.v[i + e] = vi[e]; a.p[i + e] = pn[e]; }
  }
  // ---- scatter into the packs: locate (layer, row, column) of the first element, then walk
  int l = 0;
#pragma unroll
  for (int q = 1; q < kNumLayers; ++q)
    if (d.present[q] && i >= d.w_off[q]) l = q;
  int64_t r = i - d.w_off[l];
  int n = (int)__umulhi((uint32_t)r, magic[l]), k = (int)r - n * d.n_in[l];
  // four consecutive columns of one weight row that start at a multiple of 4: in the forward pack they are elements j0 .. j0 + 3 of ONE
  // 16-byte unit (npp_layout.h: a non-embedding column c sits at element 4 (c16 >> 3) + (c16 & 3) of lane half (c16 >> 2) & 1), so they
  // leave as one 8-byte store instead of four 2-byte ones; the transposed pack keeps its 2-byte stores (its unit runs along n)
  bool fwd4 = false;
  if (cnt == 4 && n < d.n_out[l] && (k & 3) == 0 && k + 3 < d.n_in[l]) {
    const int64_t p0 = fwd_pack_pos(d, l, n, k), p3 = fwd_pack_pos(d, l, n, k + 3);
    if (p0 >= 0 && p3 == p0 + 3 && (p0 & 3) == 0) {
      typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
      const bf16x4_t w4 = {(__bf16)pn[0], (__bf16)pn[1], (__bf16)pn[2], (__bf16)pn[3]};
      *(bf16x4_t*)(a.wf + p0) = w4;
      fwd4 = true;
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (e < cnt) {
      if (n < d.n_out[l]) {                               // a weight (rows beyond n_out = the layer's bias vector: not packed)
        const __bf16 w = (__bf16)pn[e];
        if (!fwd4) {
          const int64_t pf = fwd_pack_pos(d, l, n, k);
          if (pf >= 0) a.wf[pf] = w;
        }
        const int64_t pb = bwd_pack_pos(b, l, n, k);
        if (pb >= 0) a.wb[pb] = w;
      }
      // next parameter: next column; past the row's end the next row; past the layer's end (weights + bias) the next layer
      if (++k == d.n_in[l]) { k = 0; ++n; }
      if (n >= d.n_out[l] && i + e + 1 >= d.b_off[l] + d.n_out[l]) {
        do { ++l; } while (l < kNumLayers && !d.present[l]);
        if (l >= kNumLayers) break;
        n = 0; k = 0;
      }
    }
  }
}

static uint16_t f32_to_bf16_host(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN
  u += 0x7FFF + ((u >> 16) & 1);
  return (uint16_t)(u >> 16);
}

// ---- MFMA lane-map self test -------------------------------------------------------
// out[0..1023]   : D = A(32x16) * B(16x32) stored row-major via the documented C/D map
// out[1024..2047]: Y = A2(32x32) * X, X = D converted to bf16 and fed back as the B
//                  operand of two k-steps (accumulator-as-operand chain), A2 packed with
//                  perm16 k order.
__global__ void selftest_kernel(float* out) {
  const int lane = threadIdx.x & 63;
  const int m = lane & 31, h = lane >> 5;
  bf16x8 a, b;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = 8 * h + j;                       // documented A/B operand map
    a[j] = (__bf16)(float)(((m * 3 + k * 5) % 5) - 2);   // A[m][k]
    b[j] = (__bf16)(float)(((k * 7 + m * 2) % 4) - 1);   // B[k][n=m]
  }
  f32x16 acc = {};
  acc = mfma_bf16(a, b, acc);
#pragma unroll
  for (int r = 0; r < 16; ++r) out[acc_row(r, h) * 32 + m] = acc[r];
  f32x16 y = {};
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    bf16x8 x = pack_acc(acc, s), a2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 16 * s + perm16(h, j);         // row of X this element multiplies
      a2[j] = (__bf16)(float)(((m * 5 + k * 3) % 7) - 3);  // A2[m][k]
    }
    y = mfma_bf16(a2, x, y);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) out[1024 + acc_row(r, h) * 32 + m] = y[r];

  // part 3: a 16-KiB W-format operand tile (128 features x 64 rows) holding
  // v(feature, row) = (7 feature + 3 row) % 251 - 125, written with the producer-side
  // addressing (wfmt_unit + perm16) and read back with the consumer-side transposed reads;
  // out[2048 + ((tt*4 + t)*64 + lane)*8 + j] must equal v(32 tt + (lane&31), 16 t + 8 (lane>>5) + j)
  __shared__ __attribute__((aligned(16))) char tile[16384];
  for (int u = lane; u < 1024; u += 64) {            // unit = (ks 0..7, bt, b, hh)
    const int hh = u & 1, b = (u >> 1) & 31, bt = (u >> 6) & 1, ks = u >> 7;
    bf16x8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int feat = 16 * ks + perm16(hh, j), row = bt * 32 + b;
      f[j] = (__bf16)(float)(((7 * feat + 3 * row) % 251) - 125);
    }
    *(bf16x8*)(tile + wfmt_unit(8, 0, ks, bt, b, hh)) = f;
  }
  __syncthreads();
  for (int tt = 0; tt < 4; ++tt)
    for (int t = 0; t < 4; ++t) {
      const bf16x8 f = wfrag_read(tile, wfrag_offset(tt, t, lane));
#pragma unroll
      for (int j = 0; j < 8; ++j) out[2048 + ((tt * 4 + t) * 64 + lane) * 8 + j] = (float)f[j];
    }
}

static int tune_env(const char* name, int dflt) {
  const char* e = getenv(name);
  return e && *e ? atoi(e) : dflt;
}
Tunables g_tune = {tune_env("NPP_CONV_WINK", 1), tune_env("NPP_CONV_WIN", 1), tune_env("NPP_CONV_WSTAT", 1), tune_env("NPP_CONV_PAIR", 15),
                   tune_env("NPP_LIGHT_DET", 1), tune_env("NPP_STASH8", 1)};

}  // namespace npp

using namespace npp;

extern "C" {

int npp_version(void) { return 100; }

int npp_tune(const char* key, int value) {
  struct { const char* k; int* v; } tab[] = {{"conv_wink", &g_tune.conv_wink}, {"conv_win", &g_tune.conv_win},
                                             {"conv_wstat", &g_tune.conv_wstat}, {"conv_pair", &g_tune.conv_pair},
                                             {"stash8", &g_tune.stash8}, {"light_det", &g_tune.light_det}};
  if (key)
    for (auto& t : tab)
      if (!strcmp(key, t.k)) {
        const int old = __atomic_load_n(t.v, __ATOMIC_RELAXED);
        if (value >= 0) __atomic_store_n(t.v, value, __ATOMIC_RELAXED);
        return old;
      }
  set_error("npp_tune: unknown key '%s'", key ? key : "(null)");
  return NPP_ERR_ARG;
}
const char* npp_last_error_string(void) { return g_err; }

int npp_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

static int check_kw(int K, int width) {
  if (K < 1 || K > NPP_MAX_K) { set_error("K=%d outside [1,%d]", K, NPP_MAX_K); return NPP_ERR_ARG; }
  if (width != NPP_WIDTH) {
    set_error("width=%d: this build is specialised for width %d", width, NPP_WIDTH);
    return NPP_ERR_UNSUPPORTED;
  }
  return NPP_OK;
}

int npp_param_layout(int K, int width, const char** names, int64_t* offsets, int32_t* rows,
                     int32_t* cols, int64_t* total) {
  int rc = check_kw(K, width);
  if (rc) return rc;
  const NetDesc d = make_desc(K);
  int n = 0;
  for (int l = 0; l < kNumLayers; ++l) {
    if (!d.present[l]) continue;
    for (int wb = 0; wb < 2; ++wb) {
      snprintf(g_name_buf[n], sizeof(g_name_buf[n]), "%s.%s", kNames[l], wb ? "bias" : "weight");
      if (names) names[n] = g_name_buf[n];
      if (offsets) offsets[n] = wb ? d.b_off[l] : d.w_off[l];
      if (rows) rows[n] = d.n_out[l];
      if (cols) cols[n] = wb ? 1 : d.n_in[l];
      ++n;
    }
  }
  if (total) *total = d.total_params;
  return n;
}

int64_t npp_pack_bytes(int K, int width, int which) {
  if (check_kw(K, width)) return -1;
  return 16 * (which == 0 ? make_desc(K).wf_total16 : bwd_total16(K));
}

int npp_pack_weights(const float* d_params, void* d_wf, void* d_wb, int K, int width, void* stream) {
  int rc = check_kw(K, width);
  if (rc) return rc;
  if (!d_params || !d_wf || !d_wb) { set_error("npp_pack_weights: null pointer"); return NPP_ERR_ARG; }
  const NetDesc d = make_desc(K);
  const BwdDesc b = make_bwd_desc(K);
  const int64_t nf = d.wf_total16, nb = bwd_total16(K);
  const int threads = 256;
  const int64_t blocks = (nf + nb + threads - 1) / threads;
  hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)blocks), dim3(threads), 0, (hipStream_t)stream,
                     d_params, (bf16x8*)d_wf, (bf16x8*)d_wb, d, b, nf, nb);
  return check_launch("npp_pack_weights");
}

int npp_pack_weights_host(const float* params, void* wf, void* wb, int K, int width) {
  int rc = check_kw(K, width);
  if (rc) return rc;
  const NetDesc d = make_desc(K);
  const BwdDesc b = make_bwd_desc(K);
  uint16_t* f = (uint16_t*)wf;
  uint16_t* g = (uint16_t*)wb;
  for (int64_t u = 0; u < d.wf_total16; ++u) {
    int l, ks, nt, lane;
    if (!fwd_unit(d, u, l, ks, nt, lane)) { set_error("fwd unit %lld unmapped", (long long)u); return NPP_ERR_ARG; }
    for (int j = 0; j < 8; ++j) f[u * 8 + j] = f32_to_bf16_host(fwd_pack_value(params, d, l, ks, nt, lane, j));
  }
  const int64_t nb = bwd_total16(K);
  for (int64_t u = 0; u < nb; ++u) {
    int v, ns, kt, lane;
    if (!bwd_unit(b, u, v, ns, kt, lane)) { set_error("bwd unit %lld unmapped", (long long)u); return NPP_ERR_ARG; }
    for (int j = 0; j < 8; ++j) g[u * 8 + j] = f32_to_bf16_host(bwd_pack_value(params, d, b, v, ns, kt, lane, j));
  }
  return NPP_OK;
}

int npp_pack_scatter_host(const float* params, void* wf, void* wb, int K, int width) {
  // host twin of the scatter half of npp_adam_step_net_pack: every weight of the blob through fwd_pack_pos / bwd_pack_pos
  // into zero-filled packs -- must reproduce npp_pack_weights_host bit for bit (tests/test_cabi_cpu.py)
  int rc = check_kw(K, width);
  if (rc) return rc;
  const NetDesc d = make_desc(K);
  const BwdDesc b = make_bwd_desc(K);
  uint16_t* f = (uint16_t*)wf;
  uint16_t* g = (uint16_t*)wb;
  memset(f, 0, (size_t)d.wf_total16 * 16);
  memset(g, 0, (size_t)bwd_total16(K) * 16);
  for (int l = 0; l < kNumLayers; ++l) {
    if (!d.present[l]) continue;
    for (int n = 0; n < d.n_out[l]; ++n)
      for (int k = 0; k < d.n_in[l]; ++k) {
        const uint16_t w = f32_to_bf16_host(params[d.w_off[l] + (int64_t)n * d.n_in[l] + k]);
        const int64_t pf = fwd_pack_pos(d, l, n, k), pb = bwd_pack_pos(b, l, n, k);
        if (pf >= 0) {
          if (pf >= d.wf_total16 * 8) { set_error("fwd_pack_pos out of range (l=%d n=%d k=%d)", l, n, k); return NPP_ERR_ARG; }
          f[pf] = w;
        }
        if (pb >= 0) {
          if (pb >= bwd_total16(K) * 8) { set_error("bwd_pack_pos out of range (l=%d n=%d k=%d)", l, n, k); return NPP_ERR_ARG; }
          g[pb] = w;
        }
      }
  }
  return NPP_OK;
}

int npp_adam_step_net_pack(float* d_p, float* d_m, float* d_v, const float* d_gslabs, int64_t n, int n_slabs,
                           int64_t slab_stride, float* d_lat, float* d_lat_m, float* d_lat_v, float* d_dlat, int n_lat,
                           float* d_zero, int n_zero, float lr, float beta1, float beta2, float eps, int step, int K,
                           int width, void* d_wf, void* d_wb, float* d_pl_partials, float* d_loss_cur, void* stream) {
  int rc = check_kw(K, width);
  if (rc) return rc;
  const NetDesc d = make_desc(K);
  if (n != d.total_params || !d_p || !d_m || !d_v || !d_gslabs || !d_wf || !d_wb || n_slabs < 1 || step < 1 || n_lat < 0 ||
      n_zero < 0 || (n_lat > 0 && (!d_lat || !d_lat_m || !d_lat_v || !d_dlat)) || (n_zero > 0 && !d_zero) || slab_stride % 4 ||
      (((uintptr_t)d_p | (uintptr_t)d_m | (uintptr_t)d_v | (uintptr_t)d_gslabs) & 15)) {
    set_error("npp_adam_step_net_pack: bad arguments (n=%lld, expected %lld parameters; 16-byte aligned blobs, slab stride %% 4 == 0)",
              (long long)n, (long long)d.total_params);
    return NPP_ERR_ARG;
  }
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  AdamPackArgs a{};
  a.p = d_p; a.m = d_m; a.v = d_v; a.g = d_gslabs; a.n = n; a.n_slabs = n_slabs; a.slab_stride = slab_stride;
  a.step_size = (float)((double)lr / bc1); a.b1 = beta1; a.b2 = beta2; a.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2)); a.eps = eps;
  a.tail = AdamTail{d_lat, d_lat_m, d_lat_v, d_dlat, n_lat, d_zero, n_zero, d_pl_partials, d_loss_cur};
  a.wf = (__bf16*)d_wf; a.wb = (__bf16*)d_wb;
  for (int l = 0; l < kNumLayers; ++l) a.magic[l] = d.present[l] ? (uint32_t)((1ull << 32) / (uint64_t)d.n_in[l]) + 1u : 0u;
  const int64_t threads = (n + 3) / 4;
  hipLaunchKernelGGL(adam_pack_kernel, dim3((unsigned)((threads + 255) / 256 + 1)), dim3(256), 0, (hipStream_t)stream, a, d,
                     make_bwd_desc(K));
  return check_launch("npp_adam_step_net_pack");
}

int npp_adam_step_net_pack_stack(float* d_p, float* d_m, float* d_v, int64_t blob_stride, const float* d_gslabs, int64_t n,
                                 int n_slabs, int64_t slab_stride, int64_t slab_img_stride, float* d_lat, float* d_lat_m,
                                 float* d_lat_v, float* d_dlat, int n_lat, int lat_stride, float* d_zero, int n_zero,
                                 int zero_stride, float beta1, float beta2, float eps, int M, int K, int width, void* d_wf,
                                 int64_t wf_stride_bytes, void* d_wb, int64_t wb_stride_bytes, float* d_pl_partials,
                                 int64_t pl_stride, float* d_loss_cur, const void* d_iter, void* stream) {
  int rc = check_kw(K, width);
  if (rc) return rc;
  const NetDesc d = make_desc(K);
  if (n != d.total_params || !d_p || !d_m || !d_v || !d_gslabs || !d_wf || !d_wb || !d_iter || n_slabs < 1 || n_lat < 0 || n_zero < 0 ||
      M < 1 || M > NPP_MAX_STACK || (n_lat > 0 && (!d_lat || !d_lat_m || !d_lat_v || !d_dlat)) || (n_zero > 0 && !d_zero) ||
      slab_stride % 4 || blob_stride % 4 || slab_img_stride % 4 || blob_stride < n || wf_stride_bytes % 16 || wb_stride_bytes % 16 ||
      (((uintptr_t)d_p | (uintptr_t)d_m | (uintptr_t)d_v | (uintptr_t)d_gslabs) & 15)) {
    set_error("npp_adam_step_net_pack_stack: bad arguments (n=%lld, expected %lld parameters; 16-byte aligned blobs and strides)",
              (long long)n, (long long)d.total_params);
    return NPP_ERR_ARG;
  }
  AdamPackArgs a{};
  a.p = d_p; a.m = d_m; a.v = d_v; a.g = d_gslabs; a.n = n; a.n_slabs = n_slabs; a.slab_stride = slab_stride;
  a.b1 = beta1; a.b2 = beta2; a.eps = eps;
  a.tail = AdamTail{d_lat, d_lat_m, d_lat_v, d_dlat, n_lat, d_zero, n_zero, d_pl_partials, d_loss_cur};
  a.pl_stride = pl_stride;
  a.wf = (__bf16*)d_wf; a.wb = (__bf16*)d_wb;
  a.iter = (const StackIter*)d_iter;
  a.blob_stride = blob_stride; a.slab_img_stride = slab_img_stride; a.wf_stride = wf_stride_bytes / 2; a.wb_stride = wb_stride_bytes / 2;
  a.lat_stride = lat_stride; a.zero_stride = zero_stride;
  for (int l = 0; l < kNumLayers; ++l) a.magic[l] = d.present[l] ? (uint32_t)((1ull << 32) / (uint64_t)d.n_in[l]) + 1u : 0u;
  const int64_t threads = (n + 3) / 4;
  hipLaunchKernelGGL(adam_pack_kernel, dim3((unsigned)((threads + 255) / 256 + 1), (unsigned)M), dim3(256), 0, (hipStream_t)stream, a, d,
                     make_bwd_desc(K));
  return check_launch("npp_adam_step_net_pack_stack");
}

int npp_train_workspace(int K, int width, int64_t Bp, int ksplit, int64_t sizes[4]) {
  int rc = check_kw(K, width);
  if (rc) return rc;
  if (Bp <= 0 || Bp % kRowTile || ksplit < 1 || !sizes) { set_error("npp_train_workspace: bad Bp/ksplit"); return NPP_ERR_ARG; }
  sizes[0] = 0;   // (the separate snake-derivative stash is gone: npp_mlp_bwd reads z from actF)
  // actF: the 16-bit W-format arrays, then (stash8 mode) the 8-bit W8 arrays of the same k-step table; dzF: sized for the 16-bit
  // form, which also holds the 8-bit form + its per-tile scale words -- one allocation serves both settings of npp_tune("stash8")
  sizes[1] = (int64_t)act_total_ks(K) * (Bp / kRowTile) * (2048 + 1024);
  sizes[2] = (int64_t)kDzTotalKs * (Bp / kRowTile) * 2048;
  sizes[3] = (int64_t)ksplit * slab_stride_of(make_desc(K).total_params) * 4;      // ksplit slabs, see slab_stride_of
  return NPP_OK;
}

int npp_selftest_mfma(void* d_scratch, void* stream) {
  if (!d_scratch) { set_error("npp_selftest_mfma: null scratch"); return NPP_ERR_ARG; }
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, s, (float*)d_scratch);
  int rc = check_launch("npp_selftest_mfma");
  if (rc) return rc;
  std::vector<float> out(2048 + 8192);
  hipError_t e = hipMemcpyAsync(out.data(), d_scratch, (2048 + 8192) * sizeof(float), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) { set_error("selftest copy: %s", hipGetErrorString(e)); return NPP_ERR_LAUNCH; }
  // host model with exact integers
  int A[32][16], B[16][32], D[32][32], A2[32][32];
  for (int m = 0; m < 32; ++m)
    for (int k = 0; k < 16; ++k) { A[m][k] = ((m * 3 + k * 5) % 5) - 2; B[k][m] = ((k * 7 + m * 2) % 4) - 1; }
  for (int m = 0; m < 32; ++m)
    for (int k = 0; k < 32; ++k) A2[m][k] = ((m * 5 + k * 3) % 7) - 3;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      int acc = 0;
      for (int k = 0; k < 16; ++k) acc += A[i][k] * B[k][j];
      D[i][j] = acc;
    }
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      if (out[i * 32 + j] != (float)D[i][j]) {
        set_error("selftest: plain MFMA lane map mismatch at (%d,%d): got %g want %d", i, j, out[i * 32 + j], D[i][j]);
        return NPP_ERR_SELFTEST;
      }
      int acc = 0;
      for (int k = 0; k < 32; ++k) acc += A2[i][k] * D[k][j];
      if (out[1024 + i * 32 + j] != (float)acc) {
        set_error("selftest: accumulator-as-operand chain mismatch at (%d,%d): got %g want %d", i, j,
                  out[1024 + i * 32 + j], acc);
        return NPP_ERR_SELFTEST;
      }
    }
  for (int tt = 0; tt < 4; ++tt)
    for (int t = 0; t < 4; ++t)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) {
          const int feat = 32 * tt + (lane & 31), row = 16 * t + 8 * (lane >> 5) + j;
          const float want = (float)(((7 * feat + 3 * row) % 251) - 125);
          const float got = out[2048 + ((tt * 4 + t) * 64 + lane) * 8 + j];
          if (got != want) {
            set_error("selftest: W-format transposed read mismatch tt=%d t=%d lane=%d j=%d: got %g want %g", tt, t, lane, j, got, want);
            return NPP_ERR_SELFTEST;
          }
        }
  return NPP_OK;
}

}  // extern "C"
