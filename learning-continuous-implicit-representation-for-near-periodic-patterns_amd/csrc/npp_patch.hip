// npp_patch.hip -- row a10: the patch plumbing between the MLP prediction and the patch losses
// (NPP_completion/train.py:200-236) and its backward, one launch each.
//
// The reference reshapes / tiles the predicted patch rows to (n_p*k, 3, P, P), in 'val' mode composites
// them with the known pixels of the fake patch (train.py:230-231, use_comp), multiplies prediction and
// real patches by the REAL patch mask (:232-236) and hands both to contextualLoss / percepLoss; autograd
// later folds the k tiled copies back onto the n_p*P*P prediction rows.  About twenty small torch kernels
// each way; here: compose_fwd writes the concatenated (2*n_p*k, 3, P, P) batch [x | y] the trunks take,
// compose_bwd turns dL/dx into dL/dpred rows.
#include "npp_common.h"

namespace npp {

// pred rows: (n_p*P*P, 3) row-major, row = (p*P + y)*P + x  (train.py:178-181 coordinate order)
// fake (n_p,3,P,P), fmask (n_p,1,P,P): the fake patch and its known-pixel mask (untiled)
// real (n_p*k,3,P,P), rmask (n_p*k,1,P,P)
__global__ void patch_compose_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ fake,
                                         const float* __restrict__ fmask, const float* __restrict__ real,
                                         const float* __restrict__ rmask, int n_p, int k, int P, int comp,
                                         float* __restrict__ xy) {
  const int64_t pp = (int64_t)P * P;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)n_p * k * pp) return;
  const int pk = (int)(t / pp);
  const int64_t q = t - (int64_t)pk * pp;
  const int p = pk / k;
  const float rm = rmask[(int64_t)pk * pp + q];
  const float fm = comp ? fmask[(int64_t)p * pp + q] : 0.0f;
  float* x = xy + (int64_t)pk * 3 * pp + q;
  float* y = xy + ((int64_t)n_p * k + pk) * 3 * pp + q;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float pv = pred[((int64_t)p * pp + q) * 3 + c];
    const float v = comp ? fake[((int64_t)p * 3 + c) * pp + q] * fm + pv * (1.0f - fm) : pv;    // train.py:230-231
    x[c * pp] = v * rm;                                                                          // :232-233
    y[c * pp] = real[((int64_t)pk * 3 + c) * pp + q] * rm;                                        // :235-236
  }
}

// dpred[row][c] = sum_kk (dx_a + dx_b)[p*k+kk][c][q] * rmask[p*k+kk][q] * (comp ? 1 - fmask[p][q] : 1)
__global__ void patch_compose_bwd_kernel(const float* __restrict__ dxa, const float* __restrict__ dxb,
                                         const float* __restrict__ fmask, const float* __restrict__ rmask, int n_p, int k,
                                         int P, int comp, float* __restrict__ dpred) {
  const int64_t pp = (int64_t)P * P;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)n_p * pp) return;
  const int p = (int)(t / pp);
  const int64_t q = t - (int64_t)p * pp;
  const float g = comp ? 1.0f - fmask[t] : 1.0f;
  float acc[3] = {0.0f, 0.0f, 0.0f};
  for (int kk = 0; kk < k; ++kk) {
    const int64_t pk = (int64_t)p * k + kk;
    const float rm = rmask[pk * pp + q] * g;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float d = dxa[(pk * 3 + c) * pp + q];
      if (dxb) d += dxb[(pk * 3 + c) * pp + q];
      acc[c] = fmaf(d, rm, acc[c]);
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) dpred[t * 3 + c] = acc[c];
}

// The per-iteration input rows of the loop (train.py:166-181) in one launch: the N_rand pixel rows chosen by
// np.random.choice (indices into the known-pixel table i_train, :172-174) followed by the n_p * P * P rows of the fake
// patches (window [c - P/2, c + P/2) around each centre, row-major, sampler.py:269-279), zero rows up to Bp; plus the
// ground-truth colours (and the pixel-weight mask of the remapping variant) of the pixel rows.
__global__ void batch_assemble_kernel(const int32_t* __restrict__ i_train, const int64_t* __restrict__ pix, int64_t n_pix,
                                      const int32_t* __restrict__ cen, int n_p, int P, int64_t bp,
                                      const float* __restrict__ img, const float* __restrict__ pmask_img, int W,
                                      int32_t* __restrict__ coords, float* __restrict__ gt, float* __restrict__ pmask) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= bp) return;
  int32_t r = 0, c = 0;
  if (t < n_pix) {
    const int64_t i = pix[t];
    r = i_train[2 * i];
    c = i_train[2 * i + 1];
    const float* px = img + ((int64_t)r * W + c) * 3;
    gt[t * 3 + 0] = px[0];
    gt[t * 3 + 1] = px[1];
    gt[t * 3 + 2] = px[2];
    if (pmask) pmask[t] = pmask_img[(int64_t)r * W + c];
  } else if (t < n_pix + (int64_t)n_p * P * P) {
    const int64_t q = t - n_pix;
    const int p = (int)(q / ((int64_t)P * P));
    const int rem = (int)(q - (int64_t)p * P * P);
    r = cen[2 * p] - P / 2 + rem / P;
    c = cen[2 * p + 1] - P / 2 + rem % P;
  }
  coords[2 * t] = r;
  coords[2 * t + 1] = c;
}

}  // namespace npp

using namespace npp;

extern "C" int npp_batch_assemble(const int32_t* d_i_train, int64_t n_train, const int64_t* d_pix, int64_t n_pix,
                                  const int32_t* d_cen, int n_p, int P, int64_t Bp, const float* d_img_hwc,
                                  const float* d_pmask_hw, int H, int W, int32_t* d_coords, float* d_gt, float* d_pmask,
                                  void* stream) {
  if (!d_i_train || !d_pix || !d_img_hwc || !d_coords || !d_gt || n_pix < 0 || n_train < 1 || n_p < 0 || P < 0 ||
      (n_p > 0 && !d_cen) || Bp < n_pix + (int64_t)n_p * P * P || (d_pmask && !d_pmask_hw) || H < 1 || W < 1) {
    set_error("npp_batch_assemble: bad argument (n_pix=%lld n_p=%d P=%d Bp=%lld)", (long long)n_pix, n_p, P, (long long)Bp);
    return NPP_ERR_ARG;
  }
  hipLaunchKernelGGL(batch_assemble_kernel, dim3((unsigned)((Bp + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_i_train, d_pix,
                     n_pix, d_cen, n_p, P, Bp, d_img_hwc, d_pmask_hw, W, d_coords, d_gt, d_pmask);
  return check_launch("npp_batch_assemble");
}

extern "C" int npp_patch_compose_fwd(const float* d_pred_rows, const float* d_fake, const float* d_fmask, const float* d_real,
                                     const float* d_rmask, int n_p, int k, int P, int comp, float* d_xy, void* stream) {
  if (!d_pred_rows || !d_real || !d_rmask || !d_xy || (comp && (!d_fake || !d_fmask)) || n_p < 1 || k < 1 || P < 1) {
    set_error("npp_patch_compose_fwd: bad argument (n_p=%d k=%d P=%d comp=%d)", n_p, k, P, comp);
    return NPP_ERR_ARG;
  }
  const int64_t n = (int64_t)n_p * k * P * P;
  hipLaunchKernelGGL(patch_compose_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_pred_rows,
                     d_fake, d_fmask, d_real, d_rmask, n_p, k, P, comp, d_xy);
  return check_launch("npp_patch_compose_fwd");
}

extern "C" int npp_patch_compose_bwd(const float* d_dx_a, const float* d_dx_b, const float* d_fmask, const float* d_rmask,
                                     int n_p, int k, int P, int comp, float* d_dpred_rows, void* stream) {
  if (!d_dx_a || !d_rmask || !d_dpred_rows || (comp && !d_fmask) || n_p < 1 || k < 1 || P < 1) {
    set_error("npp_patch_compose_bwd: bad argument (n_p=%d k=%d P=%d comp=%d)", n_p, k, P, comp);
    return NPP_ERR_ARG;
  }
  const int64_t n = (int64_t)n_p * P * P;
  hipLaunchKernelGGL(patch_compose_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_dx_a, d_dx_b,
                     d_fmask, d_rmask, n_p, k, P, comp, d_dpred_rows);
  return check_launch("npp_patch_compose_bwd");
}
