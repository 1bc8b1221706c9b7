// npp_segment.hip -- SURVEY.md 8 rows f3 / f4 front ends: the pieces around the AlexNet convolutions of the segmentation task's
// LPIPS(alex, spatial) criterion (NPP_segmentation/train.py:361-372 via externel_lib/lpips/lpips.py:92-133 with spatial = True,
// use_robust = False) and of the proposal search's conv1 features (NPP_proposal/feature_searching.py:20-24).  The convolutions
// themselves are im2col rows x npp_linear_fwd; round 5 formed the rows, the pooling, the per-pixel head and the upsampling with
// torch calls (F.unfold, F.max_pool2d, elementwise chains, F.interpolate): they are these four kernels now.  All HBM-bound gathers;
// activations stay position-major ([n][y][x][c]: what the GEMM writes) between the layers, so no transposition pass exists.
#include "npp_common.h"

namespace npp {

// rows (n, oy, ox) x columns (c, ky, kx) -- torch.nn.functional.unfold's column order, which is the order the (Cout, Cin k k)
// weight matrix contracts over.  nhwc: the source is position-major (the previous layer's GEMM output), else (N, C, H, W).
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ x, int N, int C, int H, int W, int k, int stride, int pad,
                                                     int ho, int wo, int nhwc, float* __restrict__ cols) {
  const int K = C * k * k;
  const int64_t total = (int64_t)N * ho * wo * K;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int col = (int)(t % K);
    const int64_t row = t / K;
    const int ox = (int)(row % wo), oy = (int)((row / wo) % ho), n = (int)(row / ((int64_t)wo * ho));
    const int kx = col % k, ky = (col / k) % k, c = col / (k * k);
    const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
    float v = 0.0f;
    if (iy >= 0 && iy < H && ix >= 0 && ix < W)
      v = nhwc ? x[(((int64_t)n * H + iy) * W + ix) * C + c] : x[(((int64_t)n * C + c) * H + iy) * W + ix];
    cols[t] = v;
  }
}

// max over k x k windows, stride s, no padding, floor output size (nn.MaxPool2d(3, 2) of torchvision's AlexNet), position-major.
__global__ __launch_bounds__(256) void maxpool_nhwc_kernel(const float* __restrict__ x, int N, int H, int W, int C, int k, int stride, int ho,
                                                           int wo, float* __restrict__ y) {
  const int64_t total = (int64_t)N * ho * wo * C;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int c = (int)(t % C);
    const int64_t p = t / C;
    const int ox = (int)(p % wo), oy = (int)((p / wo) % ho), n = (int)(p / ((int64_t)wo * ho));
    float m = -INFINITY;
    for (int dy = 0; dy < k; ++dy)
      for (int dx = 0; dx < k; ++dx) {
        const float v = x[(((int64_t)n * H + oy * stride + dy) * W + ox * stride + dx) * C + c];
        m = (v > m || v != v) ? v : m;                 // (NaN propagates, as torch's max_pool2d does)
      }
    y[t] = m;
  }
}

// lpips.py:99-110 at one tap, spatial form: d(n, p) = sum_c lin_c (a_c / (|a| + 1e-10) - b_c / (|b| + 1e-10))^2 over position-major
// features; one wave per position (C = 64 .. 384 channels), lanes stride the channels, two shuffle reductions.
__global__ __launch_bounds__(256) void lpips_spatial_kernel(const float* __restrict__ f0, const float* __restrict__ f1, int64_t P, int C,
                                                            const float* __restrict__ lin, float* __restrict__ d) {
  const int lane = threadIdx.x & 63;
  for (int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); p < P; p += (int64_t)gridDim.x * 4) {
    const float* a = f0 + p * C;
    const float* b = f1 + p * C;
    float sa = 0.0f, sb = 0.0f;
    for (int c = lane; c < C; c += 64) { sa = fmaf(a[c], a[c], sa); sb = fmaf(b[c], b[c], sb); }
    for (int off = 32; off > 0; off >>= 1) { sa += __shfl_xor(sa, off, 64); sb += __shfl_xor(sb, off, 64); }
    const float na = sqrtf(sa) + 1e-10f, nb = sqrtf(sb) + 1e-10f;         // lpips/__init__.py:42-44
    float s = 0.0f;
    for (int c = lane; c < C; c += 64) { const float e = a[c] / na - b[c] / nb; s = fmaf(lin[c] * e, e, s); }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) d[p] = s;
  }
}

// F.interpolate(mode = 'bilinear', align_corners = False) of (N, h, w) maps to (N, H, W): source index max(0, (o + 0.5) h / H - 0.5),
// the +1 neighbour clamped at the border (ATen UpSampleBilinear2d.cu: area_pixel_compute_source_index).  accumulate: out += value
// (lpips.py:125-127 sums the taps' maps).
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ x, int N, int h, int w, int H, int W, int accumulate,
                                                              float* __restrict__ y) {
  const float rh = (float)h / (float)H, rw = (float)w / (float)W;
  const int64_t total = (int64_t)N * H * W;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int ox = (int)(t % W), oy = (int)((t / W) % H), n = (int)(t / ((int64_t)W * H));
    const float sy = fmaxf(rh * ((float)oy + 0.5f) - 0.5f, 0.0f), sx = fmaxf(rw * ((float)ox + 0.5f) - 0.5f, 0.0f);
    const int y0 = (int)sy, x0 = (int)sx;
    const int yp = y0 < h - 1 ? 1 : 0, xp = x0 < w - 1 ? 1 : 0;
    const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.0f - ly, hx = 1.0f - lx;
    const float* s = x + (int64_t)n * h * w;
    const float v = hy * (hx * s[y0 * w + x0] + lx * s[y0 * w + x0 + xp]) + ly * (hx * s[(y0 + yp) * w + x0] + lx * s[(y0 + yp) * w + x0 + xp]);
    y[t] = accumulate ? y[t] + v : v;
  }
}

static unsigned grid_for(int64_t total, int per_block) { return (unsigned)std::min<int64_t>((total + per_block - 1) / per_block, 8192); }

}  // namespace npp

using namespace npp;

extern "C" int npp_im2col(const float* d_x, int N, int C, int H, int W, int k, int stride, int pad, int nhwc, float* d_cols, void* stream) {
  if (!d_x || !d_cols || N < 1 || C < 1 || H < 1 || W < 1 || k < 1 || stride < 1 || pad < 0 || H + 2 * pad < k || W + 2 * pad < k) {
    set_error("npp_im2col: bad argument (N=%d C=%d H=%d W=%d k=%d stride=%d pad=%d)", N, C, H, W, k, stride, pad);
    return NPP_ERR_ARG;
  }
  const int ho = (H + 2 * pad - k) / stride + 1, wo = (W + 2 * pad - k) / stride + 1;
  const int64_t total = (int64_t)N * ho * wo * C * k * k;
  hipLaunchKernelGGL(im2col_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, d_x, N, C, H, W, k, stride, pad, ho, wo, nhwc ? 1 : 0,
                     d_cols);
  return check_launch("npp_im2col");
}

extern "C" int npp_maxpool_nhwc(const float* d_x, int N, int H, int W, int C, int k, int stride, float* d_y, void* stream) {
  if (!d_x || !d_y || N < 1 || C < 1 || k < 1 || stride < 1 || H < k || W < k) {
    set_error("npp_maxpool_nhwc: bad argument (N=%d H=%d W=%d C=%d k=%d stride=%d)", N, H, W, C, k, stride);
    return NPP_ERR_ARG;
  }
  const int ho = (H - k) / stride + 1, wo = (W - k) / stride + 1;
  hipLaunchKernelGGL(maxpool_nhwc_kernel, dim3(grid_for((int64_t)N * ho * wo * C, 256)), dim3(256), 0, (hipStream_t)stream, d_x, N, H, W, C, k, stride,
                     ho, wo, d_y);
  return check_launch("npp_maxpool_nhwc");
}

extern "C" int npp_lpips_spatial_layer(const float* d_f0, const float* d_f1, int64_t P, int C, const float* d_lin, float* d_map, void* stream) {
  if (!d_f0 || !d_f1 || !d_lin || !d_map || P < 1 || C < 1) {
    set_error("npp_lpips_spatial_layer: bad argument (P=%lld C=%d)", (long long)P, C);
    return NPP_ERR_ARG;
  }
  hipLaunchKernelGGL(lpips_spatial_kernel, dim3(grid_for(P, 4)), dim3(256), 0, (hipStream_t)stream, d_f0, d_f1, P, C, d_lin, d_map);
  return check_launch("npp_lpips_spatial_layer");
}

extern "C" int npp_resize_bilinear(const float* d_x, int N, int h, int w, int H, int W, int accumulate, float* d_y, void* stream) {
  if (!d_x || !d_y || N < 1 || h < 1 || w < 1 || H < 1 || W < 1) {
    set_error("npp_resize_bilinear: bad argument (N=%d %dx%d -> %dx%d)", N, h, w, H, W);
    return NPP_ERR_ARG;
  }
  hipLaunchKernelGGL(resize_bilinear_kernel, dim3(grid_for((int64_t)N * H * W, 256)), dim3(256), 0, (hipStream_t)stream, d_x, N, h, w, H, W,
                     accumulate ? 1 : 0, d_y);
  return check_launch("npp_resize_bilinear");
}
