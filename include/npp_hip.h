/*
 * npp_hip.h -- C ABI of libnpp_hip.so: the MI355X (gfx950) implementation of the
 * per-image NPP-Net optimisation path.
 *
 * The reference (ArmastusChen/Learning-Continuous-Implicit-Representation-for-Near-
 * Periodic-Patterns) has no FFI or operator registry for this path: it is plain
 * Python calling PyTorch.  Each entry point below therefore replaces a Python call
 * site of the reference (cited as file:line relative to the reference root) and is
 * what a ctypes binding on the reference side would bind.
 *
 * Conventions
 *   - every function returns 0 (NPP_OK) or a negative npp_status; nothing throws;
 *     npp_last_error_string() describes the last failure on the calling thread;
 *   - every pointer named d_* is a DEVICE pointer owned by the caller; the library
 *     allocates no caller-visible memory and keeps no state between calls;
 *   - `stream` is a hipStream_t passed as void*; calls are asynchronous on it and
 *     re-entrant across streams; nothing here synchronises the device;
 *   - coordinates are (row=y, col=x) int32 pairs;
 *   - parameters live in ONE fp32 blob in the reference's own tensor layout
 *     (torch nn.Linear: weight [out][in] row-major, then bias), tensors in the order
 *     npp_param_layout() reports, so it maps 1:1 onto the reference's state_dict.
 */
#ifndef NPP_HIP_H
#define NPP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NPP_MAX_K 5         /* proposals (options/arg_config.py:73 p_topk, BASELINE c5) */
#define NPP_N_OFF 5         /* options/arg_config.py:20 freq_offsets */
#define NPP_N_FREQ 10       /* options/arg_config.py:27 multires */
#define NPP_E 462           /* embedding width per proposal: 22 * (1 + 2*10) */
#ifndef NPP_WIDTH           /* MLP width this build of the fused chain is specialised for: 256 (BASELINE c2; libnpp_hip.so) */
#define NPP_WIDTH 256       /* or 512 (the reference's default --netwidth, options/arg_config.py:57; libnpp_hip_w512.so)  */
#endif
#define NPP_MAX_TENSORS 32
#define NPP_MAX_STACK 16      /* images per stacked launch (npp_*_stack) */
#ifndef NPP_ROW_TILE
#define NPP_ROW_TILE 64     /* rows per workgroup of the fused MLP kernels */
#endif

typedef enum {
  NPP_OK = 0,
  NPP_ERR_ARG = -1,         /* bad argument (null pointer, size, unsupported K/width) */
  NPP_ERR_LAUNCH = -2,      /* hipLaunchKernel / runtime error, see error string */
  NPP_ERR_UNSUPPORTED = -3,
  NPP_ERR_SELFTEST = -4
} npp_status;

/* models/embedder.py:60-90 get_embedder(...) arguments + :26 the Fourier freqs
 * (random in the reference; an explicit input here). */
typedef struct {
  int32_t K;                          /* number of periodicity proposals */
  int32_t H, W;                       /* res = (H, W), embedder.py:112-113 */
  float angles_deg[NPP_MAX_K][2];     /* selected_angles[k] (degrees) */
  float periods[NPP_MAX_K][2];        /* selected_periods[k] */
  float offsets[NPP_N_OFF];           /* freq_offsets, default 0,-1,1,.5,-.5 */
  float freqs[NPP_N_FREQ];            /* Embedder freq_bands (embedder.py:26) */
} npp_embed_cfg;

/* ---- meta ---------------------------------------------------------------- */
int npp_version(void);
const char* npp_last_error_string(void);
int npp_device_count(void);
/* Launch-time choice between kernel FORMS that compute the same result (the reference has no counterpart: torch / cuDNN pick
 * their algorithms internally).  Keys: "conv_wink" (group-split window convolution: 0 never, 1 where measured best, 2 wherever
 * feasible), "conv_win" (window-staged convolution, same values), "conv_wstat" (weight-stationary block numbering, 0 / 1),
 * "conv_pair" (fused convolution pairs: bit 0 the first VGG block, bit 1 the second, bit 2 the first block's data gradient, bit 3 the loop's patch plumbing + pixel loss inside the first pair; 0 never, default 15),
 * "light_det" (1 default / 0: the proposal-ranking fits' fp32 weight-gradient launch without / with split-K float atomics),
 * "stash8" (round 6; 1 default / 0): the FORMAT of the training stash between npp_mlp_fwd*, npp_mlp_bwd* and npp_mlp_wgrad* -- 1: 8-bit
 * (csrc/npp_layout.h "W8-format": per snake layer its output as fp8 e4m3 and snake'(z) as an unsigned byte, f1 / f2 / embedding slots
 * as fp8, the pre-activation gradients as bf8 e5m2 scaled per 64-row tile by a power of two derived from max |dL/draw| of the tile;
 * the weight gradients are contracted by v_mfma_scale_f32_32x32x64_f8f6f4, fp32 accumulate); 0: the 16-bit stash of rounds 2-5 (fp16
 * z, bf16 gradients, bf16 MFMA).  Unlike the other keys this one changes operand ROUNDING (not the algorithm): forward values are
 * identical, weight gradients differ by the quantisation noise of bf8 x fp8 products (tests/test_gpu_parity.py states both
 * tolerances; every reference trajectory golden holds within 0.1 dB with either).  The three launches of an iteration must see the
 * same value: flip it between complete iterations only.  value < 0 only reads.  Returns the previous value, NPP_ERR_ARG for an unknown
 * key.  Initial values come from the environment (NPP_CONV_WINK=...), defaults are the measured-best forms. */
int npp_tune(const char* key, int value);

/* ---- parameters ---------------------------------------------------------- */
/* Tensor table of the fp32 parameter blob for NPP_Net (K>1, models/networks.py:40-49)
 * or NPP_Net_top1 (K==1, :128-140).  names[i] points at static strings with the
 * reference's state_dict names; offsets/rows/cols are in floats.  Returns the number
 * of tensors (>0) or a negative status.  *total = floats in the blob. */
int npp_param_layout(int K, int width, const char** names, int64_t* offsets,
                     int32_t* rows, int32_t* cols, int64_t* total);

/* Bytes of the bf16 MFMA-fragment-ordered weight packs the fused kernels read.
 * which = 0: forward pack, 1: backward (dgrad, transposed) pack. */
int64_t npp_pack_bytes(int K, int width, int which);

/* Re-pack the fp32 blob into both bf16 packs (run after every optimiser step). */
int npp_pack_weights(const float* d_params, void* d_wf, void* d_wb, int K, int width,
                     void* stream);
/* Same mapping evaluated on the HOST (no GPU needed): used by the CPU tests to check
 * the fragment maps against a NumPy model of the MFMA.  Buffers are host memory. */
int npp_pack_weights_host(const float* params, void* wf, void* wb, int K, int width);

/* ---- exact-fp32 fused forward (BASELINE config c4: "fp32"): the same function as npp_mlp_fwd on v_mfma_f32_32x32x2_f32
 * (f32 operands and accumulation, 157 TFLOP/s peak; no bf16 rounding anywhere) -- the reference's own arithmetic type
 * (models/networks.py:56-95 via F.linear, models/embedder.py:11-56,102-148).  Inference only (train.py:270-331).
 * npp_pack32_bytes / npp_pack_weights32: the fp32 A-operand pack of the blob ([layer][group of 4 k-steps][tile][lane][4]).
 * out_act: 0 raw / 1 sigmoid / 2 tanh (models/helpers.py:55-58). */
int64_t npp_pack32_bytes(int K, int width);
int npp_pack_weights32(const float* d_params, void* d_w32, int K, int width, void* stream);
int npp_mlp_fwd32(const int32_t* d_coords_yx, int64_t Bp, const npp_embed_cfg* cfg, int width, const void* d_w32,
                  const float* d_params, float* d_out, int out_act, void* stream);

/* ---- a1+a2+a4: embedder -------------------------------------------------- */
/* Replaces Embedder_periodic.embed + Embedder.embed + cat (models/embedder.py:140-148,
 * :51-56; NPP_completion/train.py:93-105): coords (N,2) -> (N, K*462), row-major,
 * reference column order.  out_dtype 0 = fp32, 1 = bf16.  precise != 0 uses full-
 * range sinf/cosf (fp32 parity path), 0 uses the hardware v_sin/v_cos. */
int npp_embed_fwd(const int32_t* d_coords_yx, int64_t N, const npp_embed_cfg* cfg,
                  void* d_out, int out_dtype, int precise, void* stream);
/* The 22-vector stage alone (embedder.py:140-148): (N,2) -> (N, K*22) fp32. */
int npp_warp_fwd(const int32_t* d_coords_yx, int64_t N, const npp_embed_cfg* cfg,
                 float* d_out, void* stream);

/* ---- a5+a6+a7: fused coordinate MLP -------------------------------------- */
/* Workspace sizes (bytes) for a padded batch of Bp rows (multiple of NPP_ROW_TILE):
 *  sizes[0] 0 (reserved)
 *  sizes[1] actF   forward stash: the 16-bit W-format region (csrc/npp_layout.h: fp16 pre-activations of the snake layers, bf16
 *                  f1 / f2, bf16 embedding slots) followed by the 8-bit W8-format region of the same k-step table; with
 *                  npp_tune("stash8") = 1 the first region holds the snake'(z) bytes (at W8 size) and the second the fp8 arrays,
 *                  with 0 only the first is used -- one allocation serves both settings
 *  sizes[2] dzF    pre-activation gradients: bf16 fragments in W-format, or (stash8) bf8 units in W8-format + one int32 scale
 *                  word per 64-row tile behind them (fits the same allocation)
 *  sizes[3] grad slabs (ksplit * S * 4 bytes, S = the parameter count rounded up to a multiple of 4 floats: the slab stride) */
int npp_train_workspace(int K, int width, int64_t Bp, int ksplit, int64_t sizes[4]);

/* Replaces render() -> run_network -> NPP_Net.forward -> sigmoid
 * (models/helpers.py:41-62, models/networks.py:56-95 / :145-173) INCLUDING the
 * embedding lookup it is fed with (train.py:166-181): coords (Bp,2) -> pred (Bp,3).
 * d_wf: forward pack; d_params: fp32 blob (biases + rgb_linear are read from it).
 * d_actF may be NULL (inference / full-image render, train.py:277-309); when given, the
 * kernel also writes the stash npp_mlp_bwd / npp_mlp_wgrad need.
 * Bp must be a multiple of NPP_ROW_TILE (pad with any valid coordinate). */
int npp_mlp_fwd(const int32_t* d_coords_yx, int64_t Bp, const npp_embed_cfg* cfg,
                int width, const void* d_wf, const float* d_params, float* d_pred,
                void* d_actF, void* stream);
/* The same with render()'s output nonlinearity as an argument (models/helpers.py:55-60): out_act 1 = sigmoid (npp_mlp_fwd), 2 = tanh
 * (--normalize_type 2: images scaled to [-1, 1], loaders/loaders.py:56,111), 0 = the raw network output. */
int npp_mlp_fwd_act(const int32_t* d_coords_yx, int64_t Bp, const npp_embed_cfg* cfg, int width, const void* d_wf,
                    const float* d_params, float* d_pred, void* d_actT, int out_act, void* stream);

/* Backward of the same (what loss.backward() does through networks.py:56-95):
 * d_dpred (Bp,3) = dL/dpred (rows >= the real batch must be 0).  Reads actF, writes dzF. */
int npp_mlp_bwd(const float* d_dpred, const float* d_pred, int64_t Bp, int K, int width,
                const void* d_wb, const float* d_params, const void* d_actF,
                void* d_dzF, void* stream);

/* The same with the patch rows' dL/dpred formed inside the launch from the patch losses' image gradients (the sum
 * npp_patch_compose_bwd computes: NPP_completion/train.py:200-236 backwards) -- rows [row0, row0 + n_p*P*P) of d_dpred are
 * WRITTEN (so the buffer ends up as loss.backward() would leave it), all other rows are read as in npp_mlp_bwd. */
typedef struct {
  const float* dx_a;      /* (n_p*k, 3, P, P) dL/dx of the contextual branch */
  const float* dx_b;      /* same shape, second branch (LPIPS / style) or NULL */
  const float* fmask;     /* (n_p, P, P) */
  const float* rmask;     /* (n_p*k, P, P) */
  int64_t row0;           /* first patch row of the batch (= N_rand) */
  int32_t n_p, k, P, comp;
} npp_patch_grad;
int npp_mlp_bwd_patch(float* d_dpred, const float* d_pred, int64_t Bp, int K, int width,
                      const void* d_wb, const float* d_params, const void* d_actF,
                      void* d_dzF, const npp_patch_grad* patch, void* stream);
/* npp_mlp_bwd_patch behind out_act (see npp_mlp_fwd_act). */
int npp_mlp_bwd_patch_act(float* d_dpred, const float* d_pred, int64_t Bp, int K, int width, const void* d_wb,
                          const float* d_params, const void* d_actT, void* d_dzT, const npp_patch_grad* pg, int out_act,
                          void* stream);

/* Compatibility forms at the reference's module boundary.  NPP_Net(...).forward(None, x_periodic)
 * (models/networks.py:56-95, NPP_Net_top1 :134-173) receives a MATERIALISED embedding:
 * d_emb (Bp, ld >= K*462) fp32 row-major in the reference's column order (any embedder may have
 * produced it).  out_act selects what render() applies to the network output (models/helpers.py:55-60):
 * 0 = none (the module's own return value), 1 = sigmoid, 2 = tanh.  The backward of that forward
 * takes the same out_act.  These read 5.5 KB per row from HBM that npp_mlp_fwd never materialises. */
int npp_mlp_fwd_emb(const float* d_emb, int64_t ld, int64_t Bp, int K, int width, const void* d_wf,
                    const float* d_params, float* d_out, void* d_actF, int out_act, void* stream);
int npp_mlp_bwd_act(const float* d_dout, const float* d_out, int64_t Bp, int K, int width,
                    const void* d_wb, const float* d_params, const void* d_actF, void* d_dzF,
                    int out_act, void* stream);

/* Weight/bias gradients: ksplit partial slabs in the parameter-blob layout
 * (slab s at d_gslabs + s * S floats, S = sizes[3] / (4 * ksplit) of npp_train_workspace: 16-byte aligned slabs);
 * npp_adam_step sums them (pass S as its slab_stride). */
int npp_mlp_wgrad(const void* d_dzF, const void* d_actF, int64_t Bp, int K, int width,
                  int ksplit, float* d_gslabs, void* stream);

/* Output tiles (256 x 256) of the grouped weight-gradient launch for K proposals: the launch has
 * tiles x ksplit workgroups of one per CU, so ksplit = CUs / tiles fills the chip in one round
 * (K = 3: 21 tiles -> 12 on 256 CUs; K = 5: 25 -> 10; K = 1: 14 -> 18). */
int npp_mlp_wgrad_tiles(int K);

/* d_grad[n] (+)= sum of the slabs: the parameter gradient as one blob, for optimisers other than
 * npp_adam_step (the reference hands model.parameters() to torch.optim.Adam, helpers.py:164). */
int npp_grad_reduce(const float* d_gslabs, int n_slabs, int64_t slab_stride, int64_t n,
                    float* d_grad, int accumulate, void* stream);

/* ---- a2 stand-alone: Embedder.embed (models/embedder.py:11-56) on any (N,d) fp32 input:
 * out (N, d*(2*n_freq + include_input)) = [x | sin(f0 x) | cos(f0 x) | ...]; freqs is a HOST array. */
int npp_fourier_fwd(const float* d_x, int64_t N, int d, const float* freqs, int n_freq,
                    int include_input, float* d_out, void* stream);

/* ---- a8: adaptive robust pixel loss -------------------------------------- */
/* Replaces img2mse(pred, gt, 'robust_loss_adaptive', adaptive_pix, mask)
 * (models/mse_calculator.py:13-27 -> robust_loss_pytorch/adaptive.py:183-204) and its
 * backward.  pred/gt (N,3), mask (N,1) or NULL.  latents: [alpha(3) | scale(3)].
 * spline: values[n_knots] then tangents[n_knots] fp32 (distribution.py:129-141).
 * Outputs: d_loss[0] += mean nll (caller zeroes), d_dpred (N,3) = weight * dL/dpred,
 * d_dlatent[6] += weight * dL/dlatents (caller zeroes). */
int npp_pixel_loss(const float* d_pred, const float* d_gt, const float* d_mask, int64_t N,
                   const float* d_latents, const float* d_spline, int n_knots,
                   float x_scale, float weight, float* d_loss, float* d_dpred,
                   float* d_dlatent, void* stream);

/* ---- a14: Adam ------------------------------------------------------------ */
/* torch.optim.Adam step (models/helpers.py:164; NPP_completion/train.py:253-254) over
 * n floats: g = sum of n_slabs slabs (slab stride = slab_stride floats).  `step` is
 * the 1-based step count used for bias correction. */
int npp_adam_step(float* d_p, float* d_m, float* d_v, const float* d_gslabs, int64_t n,
                  int n_slabs, int64_t slab_stride, float lr, float beta1, float beta2,
                  float eps, int step, void* stream);

/* One launch for optimizer.step() over the network blob AND the adaptive-loss latents
 * (create_npp_net puts both groups into one torch.optim.Adam, models/helpers.py:144-164), which
 * also performs the next optimizer.zero_grad() (train.py:192) of the small accumulators: the
 * latent gradient d_dlat[n_lat] is cleared after use and d_zero[n_zero] is cleared. */
int npp_adam_step_net(float* d_p, float* d_m, float* d_v, const float* d_gslabs, int64_t n,
                      int n_slabs, int64_t slab_stride, float* d_lat, float* d_lat_m,
                      float* d_lat_v, float* d_dlat, int n_lat, float* d_zero, int n_zero,
                      float lr, float beta1, float beta2, float eps, int step, void* stream);

/* npp_adam_step_net + npp_pack_weights in ONE launch: every updated weight is also written, as bf16, into the forward and
 * backward MFMA packs (which npp_pack_weights must have filled once: their padding elements are not rewritten).  Replaces
 * optimizer.step() (models/helpers.py:164; NPP_completion/train.py:253) followed by the re-pack the fused kernels need.
 * npp_pack_scatter_host: host twin of the scatter (inverse pack maps), for tests. */
int npp_adam_step_net_pack(float* d_p, float* d_m, float* d_v, const float* d_gslabs, int64_t n,
                           int n_slabs, int64_t slab_stride, float* d_lat, float* d_lat_m,
                           float* d_lat_v, float* d_dlat, int n_lat, float* d_zero, int n_zero,
                           float lr, float beta1, float beta2, float eps, int step, int K, int width,
                           void* d_wf, void* d_wb, float* d_pl_partials, float* d_loss_cur, void* stream);
/* (d_pl_partials, nullable: the scratch of the iteration's pixel-loss launch, npp_pixel_loss_args.scratch -- its per-block sums
 *  are added, in block order, to the latent gradients before the latents' step and to d_loss_cur[0], the iteration's pixel-loss
 *  accumulator.) */
int npp_pack_scatter_host(const float* params, void* wf, void* wb, int K, int width);

/* Same step with step_size = lr / (1 - b1^t) and 1 / sqrt(1 - b2^t) read from device memory
 * (d_hp[0], d_hp[1]): lets a captured HIP graph of one optimisation iteration be replayed. */
int npp_adam_step_dev(float* d_p, float* d_m, float* d_v, const float* d_gslabs, int64_t n,
                      int n_slabs, int64_t slab_stride, float beta1, float beta2, float eps,
                      const float* d_hp, void* stream);

/* ---- a9/a10: patch crops ---------------------------------------------------- */
/* Replaces extract_glimpse(..., mode='nearest', padding_mode='zeros', normalized=False,
 * centered=False) as models/sampler.py:171-178,284-291 calls it (utils/extract_glimpse.py:
 * 53-79): integer crops [c - P/2, c + P/2) of the (H,W,3) image and (H,W) mask at M centres
 * (row, col), zeros outside.  out_rgb (M,3,P,P), out_mask (M,1,P,P) or NULL. */
int npp_patch_gather(const float* d_img_hwc, const float* d_mask_hw, int H, int W,
                     const int32_t* d_centres_yx, int M, int P, float* d_out_rgb,
                     float* d_out_mask, void* stream);

/* The input rows of one loop iteration (NPP_completion/train.py:166-181) assembled in one launch: rows [0, n_pix) =
 * i_train[pix[t]] (the np.random.choice of :172-174), rows [n_pix, n_pix + n_p P^2) = the pixel coordinates of the fake
 * patches around centres `cen` (models/sampler.py:269-279), zero rows up to Bp; gt = img[row] and (optional, remapping)
 * pmask = pmask_img[row] of the pixel rows.  coords int32 (Bp, 2) (row, col); gt (n_pix, 3). */
int npp_batch_assemble(const int32_t* d_i_train, int64_t n_train, const int64_t* d_pix, int64_t n_pix,
                       const int32_t* d_cen, int n_p, int P, int64_t Bp, const float* d_img_hwc,
                       const float* d_pmask_hw, int H, int W, int32_t* d_coords, float* d_gt, float* d_pmask,
                       void* stream);

/* ---- a10: patch plumbing (NPP_completion/train.py:200-236) and its backward ------ */
/* d_pred_rows (n_p*P*P, 3): the predicted patch rows (row = (p*P + y)*P + x, train.py:178-181);
 * d_fake (n_p,3,P,P) / d_fmask (n_p,1,P,P): fake patch and its known mask (untiled; the reference
 * tiles them k times); d_real (n_p*k,3,P,P) / d_rmask (n_p*k,1,P,P).  comp != 0: 'val' compositing
 * (train.py:230-231).  d_xy (2*n_p*k, 3, P, P) = [x | y]: x = (comp ? fake*fmask + pred*(1-fmask)
 * : pred) * rmask, y = real * rmask -- the inputs of contextualLoss (and of percepLoss in 'same'
 * mode).  The backward sums dL/dx (two sources: d_dx_b nullable) over the k copies into
 * d_dpred_rows (n_p*P*P, 3) (overwritten). */
int npp_patch_compose_fwd(const float* d_pred_rows, const float* d_fake, const float* d_fmask,
                          const float* d_real, const float* d_rmask, int n_p, int k, int P,
                          int comp, float* d_xy, void* stream);
int npp_patch_compose_bwd(const float* d_dx_a, const float* d_dx_b, const float* d_fmask,
                          const float* d_rmask, int n_p, int k, int P, int comp,
                          float* d_dpred_rows, void* stream);

/* ---- a12: contextual loss core ------------------------------------------------ */
/* Replaces contextual_loss(x, y, band_width, weight, 'cosine') and its backward w.r.t. x
 * (externel_lib/contextual_loss/functional.py:9-63,127-163) on feature tensors (N,C,h*w) fp32
 * NCHW (C a multiple of 32).  d_loss[0] += scale * loss ; d_dfx (N,C,hw) = scale * dL/dx or
 * NULL for forward only.  d_weight: per-sample weights (functional.py:55-57) or NULL.  Any hw: the loop's
 * patches give hw <= 1600; the whole-image crops of the proposal ranking (NPP_proposal/search.py:180-197)
 * go through a column-chunked row pass (workspace = 2 N hw^2 floats + O(N hw)). */
int64_t npp_cx_workspace_bytes(int N, int C, int hw);
int npp_cx_fwd_bwd(const float* d_fx, const float* d_fy, int N, int C, int hw, float band_width,
                   const float* d_weight, float scale, float* d_loss, float* d_dfx,
                   void* d_workspace, int64_t workspace_bytes, void* stream);

/* ---- a13: LPIPS head (adaptive-robust variant), one VGG16 tap per call --------- */
/* Replaces, for tap kk, lpips.py:99-121 + :130 (normalize_tensor, per-channel robust NLL of
 * the difference, lin 1x1 conv, spatial mean) and its backward.  feats (N,C,hw) fp32 NCHW;
 * d_latents [alpha(C) | scale(C)]; d_loss[0] += scale * mean_n(...); d_df0 (N,C,hw) and
 * d_dlatent [2C] (accumulated) may both be NULL for forward only.  C in {16, 32, 64, 128, 192,
 * 256, 384, 512}.  d_workspace: npp_lpips_workspace_bytes(C) bytes, ZEROED once by the caller and owned by one
 * stream -- the latent gradients and the loss are then summed over the launch's blocks as fixed-point integers
 * (order-independent, bit-reproducible); NULL: float atomics in arrival order.
 * d_latents == NULL: the PLAIN head, LPIPS.forward(use_robust=False) (lpips.py:108-109: diffs = (feats0 - feats1)^2) -- the in-loop
 * form under --use_adaptive_perceptual_loss off (train.py:241-251) -- with its gradient d_df0 (d_spline, d_dlatent NULL). */
int64_t npp_lpips_workspace_bytes(int C);
/* All taps of LPIPS.forward in ONE launch (lpips.py:99-133: the heads are independent, `val` is their sum): taps[i] = the per-tap
 * arguments of npp_lpips_layer (host array, n_taps <= 5); a tap's workspace must not be shared with another tap of the call. */
typedef struct {
  const float* f0; const float* f1; int32_t C, hw; const float* lin; const float* latents;
  float* df0; float* dlatent; void* workspace;
  /* optional: the gradient w.r.t. feats0 as bf16 in the trunk's flat layout (a zero-initialised npp_trunk_act buffer of geometry
   * (N_total, C, H, W), H * W == hw) INSTEAD of df0 (then NULL): what npp_trunk_grad_in(df0, NULL, ...) would produce in a launch of
   * its own -- the tap gradient npp_maxpool2_bwd / npp_conv3x3_dgrad_pool add in. */
  void* dflat; int32_t N_total, H, W;
  const void* yact;   /* optional with dflat: the tapped layer's own flat fp16 activation; the gradient is gated by [yact > 0]
                       * (= npp_trunk_grad_in(df0, yact, ...): dL/d(pre-activation) of the top tap's layer) */
} npp_lpips_tap;
int npp_lpips_layers(int n_taps, const npp_lpips_tap* taps, int N, const float* d_spline, int n_knots, float x_scale, float scale,
                     float* d_loss, void* stream);
int npp_lpips_layer(const float* d_f0, const float* d_f1, int N, int C, int hw,
                    const float* d_lin, const float* d_latents, const float* d_spline,
                    int n_knots, float x_scale, float scale, float* d_loss, float* d_df0,
                    float* d_dlatent, void* d_workspace, void* stream);

/* ---- a11 / a13 trunks: frozen VGG19[0:18] / VGG16 stacks ----------------------- */
/* Replace the torchvision/cuDNN convolution stacks the two patch losses run their inputs
 * through (externel_lib/contextual_loss/modules/vgg.py:16-21,30-36: features[0:18] up to
 * relu3_4; externel_lib/lpips/pretrained_networks.py:96-134: five VGG16 slices), forward and
 * data gradient (the weights are frozen: vgg.py:26-28, pretrained_networks.py:116-118).
 *
 * Between layers tensors stay in a "flat padded" 16-bit layout (csrc/npp_conv.hip): for a
 * logical (N, C, H, W) tensor, [C/8][npp_trunk_nposp(N,H,W)][8] with a zero one-pixel border
 * around every image, so conv2d(padding=1) needs no boundary handling.  Forward activations and
 * forward weight packs are fp16 (saturating; 11 significand bits = the TF32 class cuDNN runs the
 * reference's trunks in), gradient tensors and gradient packs are bf16; accumulation is fp32.  Buffers must be
 * ZERO-INITIALISED by the caller once (kernels rewrite only the position range they compute and
 * keep borders zero).  N_total fixes a buffer's geometry; n_run <= N_total restricts a backward
 * launch to the leading images (the ones that need a gradient).  Limits: W <= 1021 (the guard band of
 * the flat layout), 512 * npp_trunk_nposp * 16 bytes < 2 GiB per tensor; NPP_ERR_ARG beyond them. */
int64_t npp_trunk_nposp(int N, int H, int W);                /* 16-byte units per channel chunk */
int64_t npp_trunk_act_bytes(int N, int C, int H, int W);     /* bytes of a flat tensor (C padded to 16) */

/* Pack a torch Conv2d weight (Cout, Cin, 3, 3) fp32 into bf16 MFMA A-operand fragments:
 * which = 0 forward pack, 1 data-gradient pack (transposed, taps flipped).  in_natural != 0:
 * the layer's input channels are in natural order (the image layer, Cin = 3 padded to 16);
 * otherwise they are in the accumulator order the previous conv layer stored them in. */
int64_t npp_conv_pack_bytes(int Cin, int Cout, int which);
int npp_conv_pack(const float* d_w, int Cin, int Cout, int in_natural, void* d_pack_fwd,
                  void* d_pack_bwd, void* stream);

/* (N,3,H,W) fp32 image -> flat C=16 tensor of x*scale[c] + shift[c] (the reference's input
 * normalisations: contextual.py:56-61 (x-mean)/std; lpips.py:96-98,136-143 scaling layer). */
int npp_trunk_image_in(const float* d_img_nchw, int N, int H, int W, const float scale[3],
                       const float shift[3], void* d_x0, void* stream);

/* npp_patch_compose_fwd + npp_trunk_image_in in ONE launch (NPP_completion/train.py:200-236 followed by
 * contextual.py:56-61): the batch [x | y] of the patch plumbing is written straight into the flat
 * C=16 trunk input d_x0 (2*n_p*k images of P x P, value*scale[c] + shift[c]); d_xy (nullable) also
 * receives it as fp32 (2*n_p*k,3,P,P) for the other consumers of the iteration (LPIPS / style
 * trunks); d_zero[0..n_zero) (n_zero <= 256, nullable) is set to 0 -- the iteration's patch-loss
 * accumulator.  Bit-identical to the two separate calls.  which = 0: both halves (d_x0 holds
 * 2*n_p*k images); 1: only the prediction half x, 2: only the real half y (d_x0 holds n_p*k
 * images; d_xy, if given, is still the full [x | y] tensor and receives that half; the inputs
 * of the other half may be NULL) -- the two halves can then run through the trunk on two streams. */
int npp_trunk_patch_in(const float* d_pred_rows, const float* d_fake, const float* d_fmask,
                       const float* d_real, const float* d_rmask, int n_p, int k, int P, int comp,
                       const float scale[3], const float shift[3], void* d_x0, float* d_xy,
                       float* d_zero, int n_zero, int which, void* stream);
/* The same launch also computing npp_pixel_loss (the arguments of that entry point, in a struct): the two consumers of
 * the forward launch's prediction (train.py:195 and :200-236) are independent, so they share one launch. */
#define NPP_PIXEL_LOSS_SCRATCH_FLOATS (1024 * 8 + 8)
typedef struct {
  const float* pred; const float* gt; const float* mask; int64_t N;
  const float* latents; const float* spline; int32_t n_knots; float x_scale, weight;
  float* loss; float* dpred; float* dlatent;
  /* nullable.  NPP_PIXEL_LOSS_SCRATCH_FLOATS floats, ZEROED once by the caller and then owned by ONE stream.  Given: the launch
   * leaves its per-block partial sums there and does NOT touch loss / dlatent; the Adam launch of the iteration
   * (npp_adam_step_net_pack(_stack), d_pl_partials = this buffer) adds them in block order -- bit-reproducible, and the
   * reduction costs no ticket, fence or launch.  NULL: float atomics onto loss / dlatent in arrival order. */
  float* scratch;
  /* 0: robust_loss_adaptive (the reference's default).  > 0: the non-adaptive switches of models/mse_calculator.py:19-23, both of the
   * form quad * mean(x^2): --loss_type l2 (quad = 1) and robust_loss = lossfun(x, alpha = 2, scale = 0.1) (quad = 50); latents, spline
   * and dlatent are then unused (may be NULL). */
  float quad;
} npp_pixel_loss_args;
int npp_trunk_patch_in_loss(const float* d_pred_rows, const float* d_fake, const float* d_fmask,
                            const float* d_real, const float* d_rmask, int n_p, int k, int P, int comp,
                            const float scale[3], const float shift[3], void* d_x0, float* d_xy,
                            float* d_zero, int n_zero, int which, const npp_pixel_loss_args* loss,
                            void* stream);

/* One 3x3 / pad 1 convolution launch on flat tensors (Cin, Cout multiples of 16, <= 512):
 *  mode 0  y = relu(conv(x, w) + bias)                    nn.Conv2d + nn.ReLU
 *  mode 1  y = conv_transpose(x) * [mask > 0]             dL/d(pre-activation) of the layer below:
 *                                                         x = dL/d(pre-act) of this layer, d_pack its
 *                                                         gradient pack, d_mask the layer-below output
 *  mode 2  y = conv_transpose(x)                          the same without a ReLU gate (into a pooled
 *                                                         tensor or the image)
 * d_tap (nullable): additionally writes the result as plain fp32 (n, Ctap, H, W) for channels
 * < Ctap (times tap_scale[c], Ctap <= 4, when tap_scale is given): feature taps for
 * npp_cx_fwd_bwd / npp_lpips_layer, and the image gradient.  d_y may be NULL if d_tap is set. */
int npp_conv3x3(const void* d_x, int N_total, int n_run, int H, int W, int Cin, int Cout,
                const void* d_pack, const float* d_bias, int mode, const void* d_mask, void* d_y,
                float* d_tap, int Ctap, const float* tap_scale, void* stream);
/* The same, additionally requesting the weight pack of the launch that follows into L2 (d_next_pack may be NULL). */
int npp_conv3x3_pf(const void* d_x, int N_total, int n_run, int H, int W, int Cin, int Cout,
                const void* d_pack, const float* d_bias, int mode, const void* d_mask, void* d_y,
                float* d_tap, int Ctap, const float* tap_scale, const void* d_next_pack, int64_t next_pack_bytes,
                   void* stream);

/* nn.MaxPool2d(2,2) on flat tensors, and its backward fused with the ReLU gate of the pooled
 * layer: dz = (route(dy) + addend) * [x > 0]; the gradient goes to the first maximum of each
 * window in scan order (torch semantics); d_addend (nullable) is a tap gradient on x. */
int npp_maxpool2_fwd(const void* d_x, int N, int H, int W, int C, void* d_y, void* stream);
int npp_maxpool2_bwd(const void* d_dy, const void* d_x, const void* d_addend, int N_total,
                     int n_run, int H, int W, int C, void* d_dz, void* stream);
/* npp_conv3x3 mode 2 INTO a pooled tensor followed by npp_maxpool2_bwd, as ONE launch (the pool's backward, the pre-pool
 * ReLU gate and the optional tap gradient ride in the convolution's epilogue; results bit-identical to the two launches).
 * H, W: the pooled geometry (= this convolution's); d_xpre (fp16 activation of the pre-pool layer), d_addend (nullable) and
 * d_dz: flat tensors of geometry (N_total, Cout, 2H, 2W); d_dz's border must be zero (npp_trunk_act buffers are) and is not
 * written.  Replaces the autograd nodes of nn.MaxPool2d + nn.ReLU in the reference's trunks
 * (externel_lib/contextual_loss/modules/vgg.py:20-27, externel_lib/lpips/pretrained_networks.py:96-130). */
/* npp_conv3x3 mode 0 followed by npp_maxpool2_fwd as ONE launch: d_y = relu(conv(x) + bias) (flat fp16, + the optional fp32
 * tap as in npp_conv3x3) and d_ypool = maxpool2x2(d_y), geometry (N_total, Cout, H/2, W/2); H, W even.  Only the n_run leading
 * images are computed / pooled; the borders of both outputs must be zero (npp_trunk_act buffers are) and are not written.
 * Bit-identical to the two launches. */
int npp_conv3x3_pool(const void* d_x, int N_total, int n_run, int H, int W, int Cin, int Cout,
                     const void* d_pack, const float* d_bias, void* d_y, void* d_ypool, float* d_tap, int Ctap,
                     const float* tap_scale, const void* d_next_pack, int64_t next_pack_bytes, void* stream);
/* TWO forward layers and the MaxPool2d(2,2) behind them in ONE launch (round 5): relu(conv a) -> relu(conv b) -> pool, the first
 * blocks of VGG19 / VGG16 (contextual_loss/modules/vgg.py:16-21, lpips/pretrained_networks.py:106-115: features[0:5], [5:10]).  A
 * workgroup owns a 16 x 16 tile of one image; the intermediate activation lives in LDS.  d_y_a / d_y_b (flat fp16, geometry
 * (N_total, Cmid | Cout, H, W)) are written for the first n_keep images only -- what the data-gradient pass reads (ReLU gates, the
 * pool's arg-max source); d_y_pool (geometry (N_total, Cout, H/2, W/2)) and the optional fp32 tap d_tap_b (n_run, Cout, H, W) for
 * all n_run images.  Borders of the outputs are not touched (they are zero from npp_trunk_alloc-style zero initialisation).
 * Bit-identical to npp_conv3x3 x 2 + npp_maxpool2_fwd.  npp_conv_pair_fwd_ok() != 0 says whether a shape is built. */
int npp_conv_pair_fwd_ok(int H, int W, int Cin, int Cmid, int Cout);
int npp_conv_pair_fwd(const void* d_x, int N_total, int n_run, int n_keep, int H, int W, int Cin, int Cmid, int Cout,
                      const void* d_pack_a, const float* d_bias_a, const void* d_pack_b, const float* d_bias_b,
                      void* d_y_a, void* d_y_b, void* d_y_pool, float* d_tap_b, void* stream);
/* npp_trunk_patch_in_loss + npp_conv_pair_fwd in ONE launch (first block; iterations that need no fp32 copy of the batch): the
 * 2 n_p k patches are composed inside the pair's input staging from the prediction rows and the sampler's crops
 * (NPP_completion/train.py:200-236; x * scale + shift), the flat trunk input is never written, and the launch's last blocks are the
 * adaptive pixel loss (`loss` as in npp_trunk_patch_in_loss; NULL: none).  d_zero / n_zero: accumulators cleared on the way.
 * Outputs as npp_conv_pair_fwd with N_total = n_run = 2 n_p k, H = W = P.  Bit-identical to the two launches. */
int npp_conv_pair_fwd_patch(const float* d_pred_rows, const float* d_fake, const float* d_fmask, const float* d_real,
                            const float* d_rmask, int n_p, int k, int P, int comp, const float scale[3], const float shift[3],
                            float* d_zero, int n_zero, const npp_pixel_loss_args* loss, int n_keep, int Cmid, int Cout,
                            const void* d_pack_a, const float* d_bias_a, const void* d_pack_b, const float* d_bias_b,
                            void* d_y_a, void* d_y_b, void* d_y_pool, float* d_tap_b, void* stream);
/* The data gradient of the first block in ONE launch: d_dz_b = dL/d(pre-activation of conv b) (flat bf16, Cmid channels) ->
 * conv b's data gradient -> ReLU gate of conv a (d_y_a: flat fp16 relu(conv a)) -> conv a's data gradient -> d_dimg fp32
 * (n_run, 3, H, W) times scale[c] -- npp_conv3x3 mode 1 (mask = d_y_a) + mode 2 with the fp32 tap, the gated intermediate gradient
 * in LDS (contextual_loss/modules/vgg.py:16-21 / lpips/pretrained_networks.py:106 under autograd).  Packs: the data-gradient packs
 * of npp_conv_pack.  Same arithmetic up to the fp32 summation order of the split-K forms it replaces. */
int npp_conv_pair_dgrad_ok(int H, int W, int Cmid);
int npp_conv_pair_dgrad(const void* d_dz_b, int N_total, int n_run, int H, int W, int Cmid, const void* d_pack_b_bwd,
                        const void* d_y_a, const void* d_pack_a_bwd, float* d_dimg, const float scale[3], void* stream);
int npp_conv3x3_dgrad_pool(const void* d_x, int N_total, int n_run, int H, int W, int Cin, int Cout,
                           const void* d_pack, const void* d_xpre, const void* d_addend, void* d_dz,
                           const void* d_next_pack, int64_t next_pack_bytes, void* stream);

/* Tap gradient (n_run, C, H, W) fp32 -> flat (bf16, or fp16 when as_f16), gated by [y > 0]
 * when the fp16 activation tensor d_y is given; accumulate != 0 adds it onto the gradient already in d_dz
 * (a tap on a tensor that also feeds the next layer).  Also the generic fp32 -> flat importer. */
int npp_trunk_grad_in(const float* d_df_nchw, const void* d_y, int N_total, int n_run, int C,
                      int H, int W, void* d_dz, int as_f16, int accumulate, void* stream);
/* the same, also requesting the weight pack of the data-gradient launch that follows into L2 (like npp_conv3x3_pf) */
int npp_trunk_grad_in_pf(const float* d_df_nchw, const void* d_y, int N_total, int n_run, int C,
                         int H, int W, void* d_dz, int as_f16, int accumulate, const void* d_next_pack,
                         int64_t next_pack_bytes, void* stream);
/* flat (fp16 activations when is_f16, else bf16 gradients) -> (n_run, C, H, W) fp32 */
int npp_trunk_export(const void* d_act, int N_total, int n_run, int C, int H, int W,
                     float* d_out_nchw, int is_f16, void* stream);

/* ---- stacked launches: M independent images of one shape in every launch (round 4) ---------------------------
 * The reference fits its images one after the other (run_completion.sh:8-14: one `python train.py` per directory);
 * the fits are independent -- own weights, own Adam state, own random stream -- so M of them can ride
 * in ONE launch sequence: BASELINE config c3 at fewer than 8 GPUs (8 / 4 / 2 images per GPU), and the cure for the
 * ~5 us every dependent launch of the single-image loop costs.  Every per-image buffer of the entry points above gets a
 * leading image dimension with a fixed stride; what differs per image AND per iteration sits in a device array of M
 * npp_stack_iter the caller uploads once per iteration.  All images share K, width, the row count Bp, n_p, P and the
 * known-pixel row count; an image whose sampler found no valid real patch (train.py:160-161) has active = 0 and is
 * skipped by every launch of that iteration (its Adam step count and LR clock do not advance).
 * Work items are numbered so that an image's workgroups stay on 8 / M of the 8 XCDs (M in {1, 2, 4, 8}): every L2 then
 * keeps one image's weight pack, as in the single-image launch. */
typedef struct {
  int32_t active;        /* 0: the image sits this iteration out */
  int32_t k;             /* real patches per fake patch (models/sampler.py:297-354), 1 on 'same' iterations */
  int32_t comp;          /* 'val' compositing (train.py:230-231) */
  int32_t with_lp;       /* 'same' iteration with the LPIPS term (train.py:241-251) */
  int32_t x0;            /* first image of this fit's prediction half in the stacked trunk batch [x_0..x_{M-1} | y_0..y_{M-1}] */
  int32_t nk;            /* n_p * k (0 when inactive) */
  int32_t same;          /* real patches := fake patches (sampler.py:338) */
  int32_t pad1;
  float step_size;       /* Adam: lr / (1 - beta1^t) of THIS image's step t */
  float inv_sqrt_bc2;    /* 1 / sqrt(1 - beta2^t) */
  float pad2, pad3;
} npp_stack_iter;

/* The embedder constants the fused kernels derive from npp_embed_cfg, as a device-memory blob (one per image of a stack):
 * npp_embed_dev_bytes() bytes each, built on the host by npp_embed_dev_build and uploaded by the caller once per fit. */
int npp_embed_dev_bytes(void);
int npp_embed_dev_build(const npp_embed_cfg* cfg, void* host_out);
/* npp_mlp_fwd over M images: coords (M, Bp, 2), pred (M, Bp, 3); image m's forward pack / parameter blob / stash at
 * + m * stride (bytes / floats / bytes).  d_iter nullable (all active). */
int npp_mlp_fwd_stack(const int32_t* d_coords_yx, int64_t Bp, const void* d_embed_dev, int M, int K, int width,
                      const void* d_wf, int64_t wf_stride_bytes, const float* d_params, int64_t params_stride,
                      float* d_pred, void* d_actF, int64_t act_stride_bytes, const void* d_iter, void* stream);
/* npp_trunk_patch_in_loss over M images.  d_crops (M, n_p + n_p*kmax, 3, P, P): per image [fake | real] as ONE
 * npp_patch_gather writes them; d_cmasks likewise (M, n_p + n_p*kmax, P, P).  The trunk batch is laid out
 * [x_0 .. x_{M-1} | y_0 .. y_{M-1}] (X = sum of the nk images each) inside a flat tensor of FIXED geometry N_total
 * (>= 2 X: 2 M n_p kmax), so the trunk launches run with n_run = 2 X (forward) / X (data gradient).  d_xy (M, 2 n_p kmax,
 * 3, P, P), nullable: fp32 [x | y] of the images with with_lp set.  d_zero: M patch-loss accumulators, cleared.
 * loss: the pixel-loss arguments of image 0; image m at pred / dpred + m Bp 3, gt + m gt_stride, latents / dlatent +
 * m lat_stride, loss + m loss_stride, scratch + m scratch_stride; loss->mask (nullable: per-pixel loss weights of the remapping
 * task, NPP_remapping/train.py:186-190) is (M, loss->N) contiguous. */
int npp_trunk_patch_in_loss_stack(const float* d_pred, int64_t Bp, int64_t row0, const float* d_crops, int64_t crop_stride,
                                  const float* d_cmasks, int64_t cmask_stride, int M, int n_p, int P, int X, int N_total,
                                  const float scale[3], const float shift[3], void* d_x0, float* d_xy, int64_t xy_stride,
                                  float* d_zero, const void* d_iter, const npp_pixel_loss_args* loss, int64_t gt_stride,
                                  int lat_stride, int loss_stride, int64_t scratch_stride, void* stream);
/* npp_cx_fwd_bwd over sample groups (one per image): see csrc/npp_cx.hip. */
int npp_cx_fwd_bwd_groups(const float* d_fx, const float* d_fy, int N, int C, int hw, float band_width, float scale,
                          float* d_loss, int loss_stride, float* d_dfx, const void* d_iter, int M, void* d_workspace,
                          int64_t workspace_bytes, void* stream);
/* npp_cx_fwd_bwd(_groups) delivering dL/dx where the trunk's data-gradient pass reads it: the flat bf16 tensor d_dz (geometry
 * N_total x C x H x W, npp_trunk_act_bytes), gated by the ReLU of the tapped layer (d_yact: its flat fp16 output) -- one launch
 * instead of the core's last one + npp_trunk_grad_in.  d_scratch_dfx (N, C, H*W) fp32: workspace.  M = 0: one group. */
int npp_cx_fwd_bwd_flat(const float* d_fx, const float* d_fy, int N, int C, int H, int W, float band_width, float scale,
                        float* d_loss, int loss_stride, float* d_scratch_dfx, const void* d_yact, void* d_dz, int N_total,
                        const void* d_iter, int M, void* d_workspace, int64_t workspace_bytes, void* stream);
/* npp_mlp_bwd_patch over M images: d_dx_a = dL/d(trunk batch) (only the leading X images are read: image m's at x0),
 * d_dx_b (M, 2 n_p kmax, 3, P, P) nullable = the LPIPS branch's gradient of the images with with_lp set. */
int npp_mlp_bwd_patch_stack(float* d_dpred, const float* d_pred, int64_t Bp, int M, int K, int width, const void* d_wb,
                            int64_t wb_stride_bytes, const float* d_params, int64_t params_stride, const void* d_actF,
                            int64_t act_stride_bytes, void* d_dzF, int64_t dz_stride_bytes, const float* d_dx_a,
                            const float* d_dx_b, int64_t dxb_stride, const float* d_cmasks, int64_t cmask_stride,
                            int64_t row0, int n_p, int P, const void* d_iter, void* stream);
/* npp_mlp_wgrad over M images: image m's ksplit slabs at d_gslabs + m * slab_img_stride floats. */
int npp_mlp_wgrad_stack(const void* d_dzF, int64_t dz_stride_bytes, const void* d_actF, int64_t act_stride_bytes,
                        int64_t Bp, int M, int K, int width, int ksplit, float* d_gslabs, int64_t slab_img_stride,
                        const void* d_iter, void* stream);
/* npp_adam_step_net_pack over M images (step size / bias correction per image from d_iter). */
int npp_adam_step_net_pack_stack(float* d_p, float* d_m, float* d_v, int64_t blob_stride, const float* d_gslabs, int64_t n,
                                 int n_slabs, int64_t slab_stride, int64_t slab_img_stride, float* d_lat, float* d_lat_m,
                                 float* d_lat_v, float* d_dlat, int n_lat, int lat_stride, float* d_zero, int n_zero,
                                 int zero_stride, float beta1, float beta2, float eps, int M, int K, int width, void* d_wf,
                                 int64_t wf_stride_bytes, void* d_wb, int64_t wb_stride_bytes, float* d_pl_partials,
                                 int64_t pl_stride, float* d_loss_cur, const void* d_iter, void* stream);

/* ---- generic dense layers (exact fp32), SURVEY.md 8 f1 -------------------------- */
/* What F.linear + SnakeActivation (models/activations.py:29-35) and their autograd do for topologies the fused chain
 * kernels are not specialised for -- first user: NPP_Net_light of the proposal-ranking fits (models/networks.py:176-263,
 * NPP_proposal/search.py:85-205).  Row-major tensors with explicit leading dimensions (so a layer can write into / read
 * from a column block of a concatenated buffer, networks.py:247 torch.cat); w (out, in) like nn.Linear.
 *  fwd:        y = act(x w^T + b), act 0 none / 1 snake / 2 relu; d_z (nullable) receives the pre-activation
 *  bwd_data:   dx[:, :in_used] (+)= dz w[:, :in_used]
 *  bwd_weight: dw (+)= dz^T x ; db (+)= column sums of dz (d_db nullable)
 *  act_bwd:    dz = dy * act'(.), act 1 snake from z, 2 sigmoid from its output, 3 tanh from its output, 4 relu from z
 *  act_fwd:    y = sigmoid (2) / tanh (3) of x, elementwise (render, models/helpers.py:55-58) */
int npp_linear_fwd(const float* d_x, int64_t ldx, const float* d_w, const float* d_b, int64_t B, int in,
                   int out, int act, float* d_y, int64_t ldy, float* d_z, int64_t ldz, void* stream);
int npp_linear_bwd_data(const float* d_dz, int64_t lddz, const float* d_w, int64_t B, int in, int out,
                        float* d_dx, int64_t lddx, int in_used, int accumulate, void* stream);
int npp_linear_bwd_weight(const float* d_dz, int64_t lddz, const float* d_x, int64_t ldx, int64_t B, int in,
                          int out, float* d_dw, float* d_db, int accumulate, void* stream);
/* The same three forms over nbatch independent problems of ONE shape in one launch -- the proposal-ranking candidates of one
 * image (NPP_proposal/search.py:85-147: each candidate its own NPP_Net_light, all on the same pixel rows): s*b are the element
 * strides between the problems' arrays.  The weight-gradient form ACCUMULATES into d_dw / d_db (split contraction; caller clears).
 * The data-gradient form optionally applies the PREVIOUS layer's activation derivative on the way out (d_zy non-null: d_dx =
 * (d_dz w) * act'(d_zy), act / d_zy as npp_act_bwd takes them; not with accumulate): one launch instead of two per hidden layer. */
int npp_linear_fwd_batched(const float* d_x, int64_t ldx, int64_t sxb, const float* d_w, int64_t swb, const float* d_b,
                           int64_t sbb, int nbatch, int64_t B, int in, int out, int act, float* d_y, int64_t ldy,
                           int64_t syb, float* d_z, int64_t ldz, int64_t szb, void* stream);
int npp_linear_bwd_data_batched(const float* d_dz, int64_t lddz, int64_t sdzb, const float* d_w, int64_t swb, int nbatch,
                                int64_t B, int in, int out, float* d_dx, int64_t lddx, int64_t sdxb, int in_used,
                                int accumulate, const float* d_zy, int64_t ldzy, int64_t szyb, int act, void* stream);
int npp_linear_bwd_weight_batched(const float* d_dz, int64_t lddz, int64_t sdzb, const float* d_x, int64_t ldx, int64_t sxb,
                                  int nbatch, int64_t B, int in, int out, float* d_dw, int64_t sdwb, float* d_db,
                                  int64_t sdbb, void* stream);
/* Weight gradient of nbatch stacked dense layers with EXPLICIT operand strides: dw[b][o][i] += sum_r dz(b, r, o) x(b, r, i),
 * db[b][o] += sum_r dz(b, r, o), where dz(b, r, o) = d_dz[b * sdzb + r * dz_sr + o * dz_so] and x(b, r, i) = d_x[b * sxb + r * x_sr +
 * i * x_si]: row-major operands (s?r = ld, s?o / s?i = 1: the form above) or feature-major ones (s?r = 1: the stashes of the fused
 * chains below).  x_snake != 0: d_x holds PRE-activations and the layer input is snake(d_x) = x + sin^2 x, formed while the operand is
 * staged (the fused chains stash z only).  Accumulates (caller clears). */
int npp_linear_bwd_weight_strided(const float* d_dz, int64_t dz_sr, int64_t dz_so, int64_t sdzb, const float* d_x, int64_t x_sr,
                                  int64_t x_si, int64_t sxb, int x_snake, int nbatch, int64_t B, int in, int out, float* d_dw,
                                  int64_t lddw, int64_t sdwb, float* d_db, int64_t sdbb, void* stream);

/* ---- f1 fused: NPP_Net_light(D = 4, W = 256, snake) forward and data-gradient chains, all candidates of an image in one launch
 * each, exact fp32 (models/networks.py:176-263 with len(freq_scales) == 1; NPP_proposal/search.py:113-147).  The parameter blob of
 * candidate c starts at d_params + c * params_stride; npp_light_desc says where its seven layers live (index 0-3 periodic_linears,
 * 4 pos_linears.0, 5 feature_linear1, 6 rgb_linear; weight [n_out][ld] with ld >= n_in, bias [n_out]).
 * npp_light_pack: the MFMA-ordered copies of the weights the chains stream (forward and transposed), npp_light_pack_floats()
 * floats per candidate; rebuild after every optimiser step.
 * npp_light_fwd: x_per (C, n_src, 20), x_pos (n_src, 42) shared; row r of the batch is table row d_idx[r] (the pixel rows drawn for
 * this iteration, search.py:113-116; d_idx null: r itself, n_src = B) -> d_pred (C, B, 3) = sigmoid(raw) and the FEATURE-major stash
 * (C, npp_light_stash_rows(), B): pre-activations z_0 .. z_3, [f1 | x_pos | 0] (304 rows), z_p -- row offsets by
 * npp_light_stash_row(0..6) = z0 z1 z2 z3 hp zp xper^T (the activations are snake(z): recomputed by their consumers).
 * npp_light_bwd: d_dpred (C, B, 3) = dL/dpred -- or, with d_gt (B, 3) non-null, the adaptive robust pixel loss itself folded in
 * (npp_pixel_loss_batched's arithmetic: d_latents (C, 6), loss words d_loss (C) and latent gradients d_dlatent (C, 6) accumulated,
 * d_dpred unused) -> d_draw (C, B, 3) = dL/draw and the gradient stash (C, npp_light_dstash_rows(), B):
 * d z_0 .. d z_3, d f1, d z_p, d raw^T (npp_light_dstash_row(0..6)).  The weight gradients: npp_light_wgrad (or
 * npp_linear_bwd_weight_strided layer by layer) over the two stashes.  B a multiple of 32 (workgroups of 64 or 32 rows, whichever loads the CUs more evenly; NPP_LIGHT_ROWS forces one). */
typedef struct {
  int64_t w_off[7], b_off[7];
  int32_t n_out[7], n_in[7], ld[7];
} npp_light_desc;
int64_t npp_light_pack_floats(void);
int64_t npp_light_stash_rows(void);
int64_t npp_light_dstash_rows(void);
int npp_light_stash_row(int which);
int npp_light_dstash_row(int which);
int npp_light_pack(const npp_light_desc* L, const float* d_params, int64_t params_stride, int C, float* d_pack, int64_t pack_stride,
                   void* stream);
int npp_light_fwd(const npp_light_desc* L, const float* d_params, int64_t params_stride, const float* d_pack, int64_t pack_stride,
                  const float* d_x_per, const float* d_x_pos, const int64_t* d_idx, int64_t n_src, int C, int64_t B, float* d_stash,
                  float* d_pred, void* stream);
/* optimizer.step() (Adam over the stacked blobs and the candidates' six latents each, npp_adam_step_net's arithmetic), zero_grad()
 * (d_grad, d_dlat and the C loss words d_zero cleared) and the re-pack of every updated weight (npp_light_pack's layout) in ONE launch.
 * n: live floats per candidate (<= stride). */
int npp_light_adam_pack(const npp_light_desc* L, float* d_params, float* d_m, float* d_v, float* d_grad, int64_t stride, int64_t n, int C,
                        float* d_pack, int64_t pack_stride, float* d_lat, float* d_lat_m, float* d_lat_v, float* d_dlat, float* d_zero,
                        float lr, float beta1, float beta2, float eps, int step, void* stream);
/* The seven weight / bias gradients of all candidates in one launch over the two stashes: into d_grad + c * grad_stride at the
 * parameters' own offsets (accumulated; clear first). */
int npp_light_wgrad(const npp_light_desc* L, const float* d_stash, const float* d_dstash, int C, int64_t B, float* d_grad,
                    int64_t grad_stride, void* stream);
int npp_light_bwd(const npp_light_desc* L, const float* d_params, int64_t params_stride, const float* d_pack, int64_t pack_stride,
                  const float* d_stash, const float* d_pred, const float* d_dpred, const float* d_gt, const float* d_latents,
                  const float* d_spline, int n_knots, float x_scale, float* d_loss, float* d_dlatent, int C, int64_t B,
                  float* d_draw, float* d_dstash, void* stream);
/* Bit-reproducible forms (round 6; the reference's candidate fits are not -- cuBLAS / atomics -- but rankings with near-equal scores
 * should not depend on arrival order): npp_light_bwd_det = npp_light_bwd with the pixel loss folded in (d_gt required) whose per-block
 * loss / latent-gradient sums go to d_part (C, npp_light_part_blocks(C, B), 8) by plain stores; npp_light_adam_pack_det adds them in
 * block order (latent gradients = d_dlat + the blocks' sums; d_loss_cur[c] += the blocks' loss terms, nullable).  With
 * npp_tune("light_det") = 1 (default) npp_light_wgrad does not split its contraction (no float atomicAdd): the whole candidate fit
 * has no order-dependent float sum left.  NPP_proposal/search.py:113-147. */
int npp_light_part_blocks(int C, int64_t B);
/* npp_light_wgrad with its contraction split over enough workgroups to fill the chip (a single candidate is 80 output tiles: 80
 * workgroups walking 2048 rows each when unsplit) AND bit-reproducible: the row ranges of an output tile leave their partial tiles in
 * d_scratch, the workgroup that arrives last adds them in range order and is the tile's only writer (no float atomicAdd).
 * d_scratch: npp_light_wgrad_det_scratch_bytes(C, B) bytes, ZEROED once before the first use (the arrival tickets reset themselves). */
int64_t npp_light_wgrad_det_scratch_bytes(int C, int64_t B);
int npp_light_wgrad_det(const npp_light_desc* L, const float* d_stash, const float* d_dstash, int C, int64_t B, float* d_grad,
                        int64_t grad_stride, float* d_scratch, int64_t scratch_bytes, void* stream);
/* "multi" forms: candidate c = ONE IMAGE's fit (the searches of several images of a rank advanced in one launch sequence: candidate k of
 * every image together, where run_completion.sh / search.py:85-215 walk images and candidates one after the other): per candidate its own
 * positional table d_x_pos (C, n_src, 42), periodic table d_x_per (C, n_src, 20) as before, pixel rows d_idx (C, B) into its own tables
 * (shorter tables padded to n_src rows) and targets d_gt (C, B, 3). */
int npp_light_fwd_multi(const npp_light_desc* L, const float* d_params, int64_t params_stride, const float* d_pack, int64_t pack_stride,
                        const float* d_x_per, const float* d_x_pos, const int64_t* d_idx, int64_t n_src, int C, int64_t B, float* d_stash,
                        float* d_pred, void* stream);
int npp_light_bwd_det_multi(const npp_light_desc* L, const float* d_params, int64_t params_stride, const float* d_pack, int64_t pack_stride,
                            const float* d_stash, const float* d_pred, const float* d_gt, const float* d_latents, const float* d_spline,
                            int n_knots, float x_scale, float* d_part, int C, int64_t B, float* d_draw, float* d_dstash, void* stream);
int npp_light_bwd_det(const npp_light_desc* L, const float* d_params, int64_t params_stride, const float* d_pack, int64_t pack_stride,
                      const float* d_stash, const float* d_pred, const float* d_gt, const float* d_latents, const float* d_spline,
                      int n_knots, float x_scale, float* d_part, int C, int64_t B, float* d_draw, float* d_dstash, void* stream);
int npp_light_adam_pack_det(const npp_light_desc* L, float* d_params, float* d_m, float* d_v, float* d_grad, int64_t stride, int64_t n,
                            int C, float* d_pack, int64_t pack_stride, float* d_lat, float* d_lat_m, float* d_lat_v, float* d_dlat,
                            float* d_zero, float lr, float beta1, float beta2, float eps, int step, const float* d_part, int n_part,
                            float* d_loss_cur, void* stream);
/* ---- f1 on the 16-bit matrix pipe (csrc/npp_light16.hip): the same chains with bf16 operands, fp32 accumulation and fp32 master
 * weights -- the numeric contract of the main loop's coordinate MLP (npp_mlp_fwd / npp_mlp_bwd / npp_mlp_wgrad), models/networks.py:176-263,
 * NPP_proposal/search.py:113-147.  B a multiple of 64.  Per candidate: a bf16 pack of npp_light16_pack_bytes() bytes (forward and
 * transposed weights), a forward stash of npp_light16_stash_bytes(B, 0) bytes (W-format fragment arrays: fp16 pre-activations z_0..z_3,
 * z_p, bf16 [f1 | x_pos], x_per) and a gradient stash of npp_light16_stash_bytes(B, 1) bytes (bf16 d z_0..d z_3, d f1, d z_p, d raw);
 * the strides (bytes, multiples of 16) place candidate c at + c * stride.
 * npp_light16_fwd: as npp_light_fwd.  npp_light16_bwd: as npp_light_bwd (d_gt non-null folds the adaptive robust pixel loss in).
 * npp_light16_wgrad: all seven weight / bias gradients of the C <= NPP_MAX_STACK candidates in ONE launch of the main loop's grouped
 * split-K kernel: candidate c's ksplit partial sums land, by plain stores, in d_gslabs + c * slab_cand_stride + s * slab_stride
 * (s < ksplit) at the parameters' own offsets (npp_light_desc; leading dimensions and offsets multiples of 4, pos_linears.0 stored 300
 * wide).  npp_light16_adam_pack: optimizer.step() over the stacked fp32 blobs with gradient = the slabs summed in slab order
 * (bit-reproducible), the six latents per candidate (d_dlat consumed and cleared, loss words d_zero cleared) and the re-pack of every
 * updated weight into the bf16 packs, one launch. */
int64_t npp_light16_pack_bytes(void);
int64_t npp_light16_stash_bytes(int64_t B, int which);
int npp_light16_pack(const npp_light_desc* L, const float* d_params, int64_t params_stride, int C, void* d_pack, int64_t pack_stride_bytes,
                     void* stream);
int npp_light16_fwd(const npp_light_desc* L, const float* d_params, int64_t params_stride, const void* d_pack, int64_t pack_stride_bytes,
                    const float* d_x_per, const float* d_x_pos, const int64_t* d_idx, int64_t n_src, int C, int64_t B, void* d_actF,
                    int64_t act_stride_bytes, float* d_pred, void* stream);
int npp_light16_bwd(const npp_light_desc* L, const float* d_params, int64_t params_stride, const void* d_pack, int64_t pack_stride_bytes,
                    const void* d_actF, int64_t act_stride_bytes, const float* d_pred, const float* d_dpred, const float* d_gt,
                    const float* d_latents, const float* d_spline, int n_knots, float x_scale, float* d_loss, float* d_dlatent, int C,
                    int64_t B, void* d_dzF, int64_t dz_stride_bytes, void* stream);
int npp_light16_wgrad(const npp_light_desc* L, const void* d_actF, int64_t act_stride_bytes, const void* d_dzF, int64_t dz_stride_bytes,
                      int C, int64_t B, int ksplit, float* d_gslabs, int64_t slab_stride, int64_t slab_cand_stride, void* stream);
int npp_light16_adam_pack(const npp_light_desc* L, float* d_params, float* d_m, float* d_v, int64_t stride, int64_t n, int C,
                          const float* d_gslabs, int n_slabs, int64_t slab_stride, int64_t slab_cand_stride, void* d_pack,
                          int64_t pack_stride_bytes, float* d_lat, float* d_lat_m, float* d_lat_v, float* d_dlat, float* d_zero, float lr,
                          float beta1, float beta2, float eps, int step, void* stream);
/* (round 6) the 16-bit candidate fits, bit-reproducible and across images -- what npp_light_bwd_det / npp_light_adam_pack_det /
 * npp_light_fwd_multi are to the fp32 chains.  npp_light16_bwd_det: the pixel loss folded in (d_gt required), every block's seven loss /
 * latent-gradient sums to d_part (C, B / 64, 8) by plain stores; gt_cs = elements per candidate of d_gt (0: one (B, 3) target shared;
 * 3 B: candidate c = one image's fit with its own targets).  npp_light16_adam_pack_det adds them in block order (latent gradients =
 * d_dlat + the sums; d_loss_cur[c] += the loss terms, nullable).  npp_light16_fwd_multi: per candidate its own positional table
 * d_x_pos (C, n_src, 42) and pixel rows d_idx (C, B).  NPP_proposal/search.py:85-215. */
int npp_light16_fwd_multi(const npp_light_desc* L, const float* d_params, int64_t params_stride, const void* d_pack,
                          int64_t pack_stride_bytes, const float* d_x_per, const float* d_x_pos, const int64_t* d_idx, int64_t n_src,
                          int C, int64_t B, void* d_actF, int64_t act_stride_bytes, float* d_pred, void* stream);
int npp_light16_bwd_det(const npp_light_desc* L, const float* d_params, int64_t params_stride, const void* d_pack,
                        int64_t pack_stride_bytes, const void* d_actF, int64_t act_stride_bytes, const float* d_pred,
                        const float* d_gt, int64_t gt_cs, const float* d_latents, const float* d_spline, int n_knots, float x_scale,
                        float* d_part, int C, int64_t B, void* d_dzF, int64_t dz_stride_bytes, void* stream);
int npp_light16_adam_pack_det(const npp_light_desc* L, float* d_params, float* d_m, float* d_v, int64_t stride, int64_t n, int C,
                              const float* d_gslabs, int n_slabs, int64_t slab_stride, int64_t slab_cand_stride, void* d_pack,
                              int64_t pack_stride_bytes, float* d_lat, float* d_lat_m, float* d_lat_v, float* d_dlat, float* d_zero,
                              float lr, float beta1, float beta2, float eps, int step, const float* d_part, int n_part,
                              float* d_loss_cur, void* stream);
/* img2mse with --loss_type l2 (coef = 1) or robust_loss (coef = 50: lossfun(diff, alpha = 2, scale = 0.1) = 0.5 (diff / 0.1)^2),
 * models/mse_calculator.py:13-27, for nbatch problems: d_loss[b] += weight * coef * mean(x^2), d_dpred = its gradient; the (N) mask
 * (nullable) is shared, the targets too when gt_stride == 0. */
int npp_pixel_loss_quad(const float* d_pred, const float* d_gt, int64_t gt_stride, const float* d_mask, int64_t N, int nbatch, float coef,
                        float weight, float* d_loss, float* d_dpred, void* stream);
/* npp_pixel_loss over nbatch problems: d_pred / d_dpred (nbatch, N, 3), d_latents / d_dlatent (nbatch, 6), d_loss (nbatch);
 * the targets d_gt (N, 3) are shared when gt_stride == 0, else problem b reads d_gt + b * gt_stride. */
int npp_pixel_loss_batched(const float* d_pred, const float* d_gt, int64_t gt_stride, int64_t N, int nbatch,
                           const float* d_latents, const float* d_spline, int n_knots, float x_scale, float weight,
                           float* d_loss, float* d_dpred, float* d_dlatent, void* stream);
int npp_act_bwd(const float* d_dy, int64_t lddy, const float* d_zy, int64_t ldzy, int64_t B, int n, int act,
                float* d_dz, int64_t lddz, void* stream);
int npp_act_fwd(const float* d_x, int64_t n, int act, float* d_y, void* stream);
/* LPIPS.forward(use_robust=False), one VGG16 tap, forward only (externel_lib/lpips/lpips.py:99-101,110,117,130; the
 * candidate score of NPP_proposal/search.py:193): d_out[0] += scale * sum_n mean_pos sum_c lin_c (f0n - f1n)^2. */
int npp_lpips_plain_layer(const float* d_f0, const float* d_f1, int N, int C, int hw, const float* d_lin,
                          float scale, float* d_out, void* stream);
/* The same with the workgroups' partial sums added in block order instead of by float atomics (the candidate SCORES that rank the
 * proposals, NPP_proposal/search.py:193, are then bit-reproducible): d_scratch = NPP_LPIPS_PLAIN_SCRATCH_FLOATS floats, zeroed once by
 * the caller (each launch re-arms it); launches that may run concurrently need scratches of their own. */
#define NPP_LPIPS_PLAIN_SCRATCH_FLOATS 264
int npp_lpips_plain_layer_det(const float* d_f0, const float* d_f1, int N, int C, int hw, const float* d_lin, float scale, float* d_out,
                              float* d_scratch, void* stream);

/* ---- remapping variant: Gram-matrix style loss (models/style_loss.py:37-74) ----------------- */
/* G[n] = F[n] F[n]^T for (N, C, hw) features and its backward dF[n] = (dG[n] + dG[n]^T) F[n]; the per-element adaptive
 * robust NLL of a - b over (N, D) with D latent pairs (AdaptiveLossFunction(num_dims = C^2), style_loss.py:23-27,60-69):
 * d_loss[0] += sum_n coef_n[n] sum_j nll(a - b)[n][j]; d_diff (N, D) scratch; d_ddiff / d_dlatent (nullable together):
 * coef_n dnll/dx and the accumulated latent gradients.  coef_n is a HOST array of N <= 64 factors. */
int npp_gram_fwd(const float* d_f, int N, int C, int hw, float* d_g, void* stream);
/* (round 6) the same with the split contraction's partial sums added in range order: bit-reproducible where npp_gram_fwd adds them with
 * float atomics in arrival order.  Two launches: the ranges leave their partial tiles in d_scratch, a small second launch adds them.
 * d_scratch: npp_gram_fwd_det_scratch_bytes(N, C, hw) bytes, no initial content required, not shared by launches that may run
 * concurrently. */
int64_t npp_gram_fwd_det_scratch_bytes(int N, int C, int hw);
int npp_gram_fwd_det(const float* d_f, int N, int C, int hw, float* d_g, float* d_scratch, int64_t scratch_bytes, void* stream);
int npp_gram_bwd(const float* d_dg, const float* d_f, int N, int C, int hw, float* d_df, void* stream);
/* (round 6) ONE launch: the per-channel constants, the difference a - b (written to d_diff) and the loss are all formed inside it
 * -- nothing is handed from one launch to the next through the workspace, which now holds only the blocks' loss partials and their
 * arrival ticket (the launcher clears the ticket itself; no initial content required; not shared by concurrent launches). */
int64_t npp_robust_elem_workspace_bytes(int D);
int npp_robust_elem(const float* d_a, const float* d_b, int N, int D, const float* d_latents,
                    const float* d_spline, int n_knots, float x_scale, const float* coef_n, float* d_loss,
                    float* d_diff, float* d_ddiff, float* d_dlatent, void* d_workspace, void* stream);

/* ---- SURVEY.md 8 f3 / f4 front ends: around the AlexNet convolutions ------------------------ */
/* (round 6) The segmentation task's LPIPS(alex, spatial = True) criterion (NPP_segmentation/train.py:361-372; externel_lib/lpips/
 * lpips.py:92-133, pretrained_networks.py alexnet slices) and the proposal search's conv1 features (NPP_proposal/feature_searching.py:
 * 20-24) run their convolutions as im2col rows x npp_linear_fwd; these are the pieces around the products (torch calls before).
 * Activations between the layers are POSITION-MAJOR (n, y, x, c) -- what npp_linear_fwd writes for rows (n, y, x).
 * npp_im2col: rows (n, oy, ox), columns (c, ky, kx) (torch.nn.functional.unfold's order = the (Cout, Cin k k) weight matrix's), zeros
 *   outside the image; nhwc != 0: d_x is position-major, else (N, C, H, W).  d_cols: N ho wo x C k k floats, ho = (H + 2 pad - k) / stride + 1.
 * npp_maxpool_nhwc: nn.MaxPool2d(k, stride) (no padding, floor size) on a position-major tensor; NaN propagates.
 * npp_lpips_spatial_layer: one tap of LPIPS.forward(spatial = True, use_robust = False) before the upsampling: d_map[p] = sum_c lin_c
 *   (a_c / (|a| + 1e-10) - b_c / (|b| + 1e-10))^2 for P positions of C channels (lpips.py:99-110, lpips/__init__.py:42-44).
 * npp_resize_bilinear: F.interpolate(size = (H, W), mode = 'bilinear', align_corners = False) of N maps (h, w) (lpips.py:20-22);
 *   accumulate != 0 adds into d_y (lpips.py:125-127 sums the taps). */
int npp_im2col(const float* d_x, int N, int C, int H, int W, int k, int stride, int pad, int nhwc, float* d_cols, void* stream);
int npp_maxpool_nhwc(const float* d_x, int N, int H, int W, int C, int k, int stride, float* d_y, void* stream);
int npp_lpips_spatial_layer(const float* d_f0, const float* d_f1, int64_t P, int C, const float* d_lin, float* d_map, void* stream);
int npp_resize_bilinear(const float* d_x, int N, int h, int w, int H, int W, int accumulate, float* d_y, void* stream);

/* ---- brute-force displacement search ------------------------------------- */
/* compute_loss of NPP_proposal/feature_searching.py:208-264: act (C, h, w) fp32 feature map whose LAST channel is excluded
 * from the sum (:247,250), mask (h, w) 1 = known, shifts (n, 2) int32 (dx, dy).  losses[s] = sum over positions of
 * mask * shifted mask * sum_c f(shifted act, act), f = -a*b when edge_searching else (a-b)^2; zero outside the map. */
int npp_shift_search(const float* d_act_chw, const float* d_mask_hw, int C, int h, int w,
                     const int32_t* d_shifts_xy, int n, int edge_searching, float* d_losses, void* stream);

/* ---- host side: the reference's NumPy random stream, GIL-free ----------------------- */
/* numpy.random.RandomState(seed) restated bit for bit for the three draws of an iteration (models/sampler.py:260,324;
 * NPP_completion/train.py:172): MT19937 with init_genrand seeding, uniform() from the 53-bit double, and
 * choice(n, size, replace=False) == permutation(n)[:size] (full Fisher-Yates with the legacy masked-rejection
 * random_interval).  Host memory only, no GPU; one handle per stream, not thread-safe per handle. */
void* npp_rng_create(uint32_t seed);
void npp_rng_destroy(void* rng);
int npp_rng_seed(void* rng, uint32_t seed);
int npp_rng_get_state(void* rng, uint32_t* key624, int32_t* pos);      /* == RandomState.get_state()[1:3] */
int npp_rng_set_state(void* rng, const uint32_t* key624, int32_t pos);
double npp_rng_uniform(void* rng, double lo, double hi);
int npp_rng_choice_noreplace(void* rng, int64_t n, int64_t size, int64_t* scratch_n, int64_t* out_size);

/* ---- GridPatchSampler's host-side draw (models/sampler.py:242-354: patch source :324-331, fake centres :260, lattice
 * candidates + unknown-pixel filter + k nearest :148-214) in one GIL-free host call, consuming `rng` like the reference
 * consumes np.random.  sat: (H+1) x (W+1) summed-area table of the known mask; pools: (row, col) int32 pairs in the
 * reference's np.nonzero order; shifts_dydx: the two lattice shifts as (dy, dx) (sampler.py:35). */
void* npp_sampler_create(const int64_t* sat, int H, int W, const int32_t* pool_train, int64_t n_train,
                         const int32_t* pool_val, int64_t n_val, const double* shifts_dydx);
void npp_sampler_destroy(void* sampler);
int npp_sampler_set_patch(void* sampler, int patch_size, int n_samples, int64_t* pool_train_n, int64_t* pool_val_n);
int npp_sampler_draw(void* sampler, void* rng, int topk, double invalid_ratio, int32_t* source, int32_t* k,
                     int32_t* cen_n2, double* real_cen_ntopk2, float* weights_ntopk);

/* ---- diagnostics ---------------------------------------------------------- */
/* Checks the MFMA operand / accumulator lane maps this library relies on (incl. the
 * accumulator-as-next-operand chain) with exact integer data.  d_scratch >= 1 MiB. */
int npp_selftest_mfma(void* d_scratch, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NPP_HIP_H */
