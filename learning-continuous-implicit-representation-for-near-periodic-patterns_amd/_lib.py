"""ctypes binding of libnpp_hip.so (include/npp_hip.h).  Fails loudly; never falls back."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NPP_LIB_PATH") or os.path.join(HERE, "libnpp_hip.so")   # override: A/B builds
# The fused MLP chain is specialised at compile time for one network width (include/npp_hip.h NPP_WIDTH): the same sources
# built with -DNPP_WIDTH=512 give the library for the reference's default --netwidth (options/arg_config.py:57).
LIB_PATHS = {256: LIB_PATH, 512: os.environ.get("NPP_LIB_PATH_W512") or os.path.join(HERE, "libnpp_hip_w512.so")}
FUSED_WIDTHS = tuple(sorted(LIB_PATHS))

NPP_MAX_K, NPP_N_OFF, NPP_N_FREQ, NPP_E, NPP_WIDTH, NPP_ROW_TILE = 5, 5, 10, 462, 256, 64


class NppError(RuntimeError):
    pass


class PixelLossArgs(C.Structure):
    """npp_pixel_loss_args (include/npp_hip.h)."""
    _fields_ = [("pred", C.c_void_p), ("gt", C.c_void_p), ("mask", C.c_void_p), ("N", C.c_int64), ("latents", C.c_void_p),
                ("spline", C.c_void_p), ("n_knots", C.c_int32), ("x_scale", C.c_float), ("weight", C.c_float), ("loss", C.c_void_p),
                ("dpred", C.c_void_p), ("dlatent", C.c_void_p), ("scratch", C.c_void_p), ("quad", C.c_float)]


class LpipsTap(C.Structure):
    """npp_lpips_tap (include/npp_hip.h): the per-tap arguments of npp_lpips_layers."""
    _fields_ = [("f0", C.c_void_p), ("f1", C.c_void_p), ("C", C.c_int32), ("hw", C.c_int32), ("lin", C.c_void_p), ("latents", C.c_void_p),
                ("df0", C.c_void_p), ("dlatent", C.c_void_p), ("workspace", C.c_void_p),
                ("dflat", C.c_void_p), ("N_total", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("yact", C.c_void_p)]


class PatchGrad(C.Structure):
    """npp_patch_grad (include/npp_hip.h)."""
    _fields_ = [("dx_a", C.c_void_p), ("dx_b", C.c_void_p), ("fmask", C.c_void_p), ("rmask", C.c_void_p),
                ("row0", C.c_int64), ("n_p", C.c_int32), ("k", C.c_int32), ("P", C.c_int32), ("comp", C.c_int32)]


class StackIter(C.Structure):
    """npp_stack_iter (include/npp_hip.h): what differs per image and per iteration in a stacked launch."""
    _fields_ = [("active", C.c_int32), ("k", C.c_int32), ("comp", C.c_int32), ("with_lp", C.c_int32), ("x0", C.c_int32),
                ("nk", C.c_int32), ("same", C.c_int32), ("pad1", C.c_int32), ("step_size", C.c_float), ("inv_sqrt_bc2", C.c_float),
                ("pad2", C.c_float), ("pad3", C.c_float)]


class LightDesc(C.Structure):
    """npp_light_desc (include/npp_hip.h): where the seven layers of an NPP_Net_light live in its parameter blob."""
    _fields_ = [("w_off", C.c_int64 * 7), ("b_off", C.c_int64 * 7), ("n_out", C.c_int32 * 7), ("n_in", C.c_int32 * 7), ("ld", C.c_int32 * 7)]


class EmbedCfg(C.Structure):
    """npp_embed_cfg (include/npp_hip.h) == get_embedder(...) arguments, models/embedder.py:60-90."""
    _fields_ = [("K", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("angles_deg", (C.c_float * 2) * NPP_MAX_K),
                ("periods", (C.c_float * 2) * NPP_MAX_K),
                ("offsets", C.c_float * NPP_N_OFF),
                ("freqs", C.c_float * NPP_N_FREQ)]

    @classmethod
    def make(cls, angles_deg, periods, freqs, res, offsets=(0.0, -1.0, 1.0, 0.5, -0.5)):
        import numpy as np
        a = np.asarray(angles_deg, dtype=np.float32).reshape(-1, 2)
        p = np.asarray(periods, dtype=np.float32).reshape(-1, 2)
        f = np.asarray(freqs, dtype=np.float32).reshape(-1)
        if a.shape != p.shape or not (1 <= a.shape[0] <= NPP_MAX_K):
            raise ValueError(f"need 1..{NPP_MAX_K} proposals with 2 angles and 2 periods each")
        if f.shape[0] != NPP_N_FREQ or len(offsets) != NPP_N_OFF:
            raise ValueError("this build is specialised for multires=10 and 5 freq_offsets")
        c = cls()
        c.K, c.H, c.W = a.shape[0], int(res[0]), int(res[1])
        for k in range(a.shape[0]):
            for o in range(2):
                c.angles_deg[k][o] = float(a[k, o])
                c.periods[k][o] = float(p[k, o])
        for j in range(NPP_N_OFF):
            c.offsets[j] = float(offsets[j])
        for j in range(NPP_N_FREQ):
            c.freqs[j] = float(f[j])
        return c


_vp, _i64, _i32, _f32 = C.c_void_p, C.c_int64, C.c_int, C.c_float
_cfgp = C.POINTER(EmbedCfg)

# every symbol include/npp_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "npp_version": (_i32, []),
    "npp_last_error_string": (C.c_char_p, []),
    "npp_device_count": (_i32, []),
    "npp_tune": (_i32, [C.c_char_p, _i32]),
    "npp_param_layout": (_i32, [_i32, _i32, C.POINTER(C.c_char_p), C.POINTER(_i64), C.POINTER(C.c_int32),
                                C.POINTER(C.c_int32), C.POINTER(_i64)]),
    "npp_pack_bytes": (_i64, [_i32, _i32, _i32]),
    "npp_pack_weights": (_i32, [_vp, _vp, _vp, _i32, _i32, _vp]),
    "npp_pack_weights_host": (_i32, [_vp, _vp, _vp, _i32, _i32]),
    "npp_pack32_bytes": (_i64, [_i32, _i32]),
    "npp_pack_weights32": (_i32, [_vp, _vp, _i32, _i32, _vp]),
    "npp_mlp_fwd32": (_i32, [_vp, _i64, _cfgp, _i32, _vp, _vp, _vp, _i32, _vp]),
    "npp_embed_fwd": (_i32, [_vp, _i64, _cfgp, _vp, _i32, _i32, _vp]),
    "npp_warp_fwd": (_i32, [_vp, _i64, _cfgp, _vp, _vp]),
    "npp_train_workspace": (_i32, [_i32, _i32, _i64, _i32, C.POINTER(_i64)]),
    "npp_mlp_fwd": (_i32, [_vp, _i64, _cfgp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "npp_mlp_fwd_act": (_i32, [_vp, _i64, _cfgp, _i32, _vp, _vp, _vp, _vp, _i32, _vp]),
    "npp_mlp_bwd": (_i32, [_vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "npp_mlp_wgrad": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp]),
    "npp_mlp_wgrad_tiles": (_i32, [_i32]),
    "npp_pixel_loss": (_i32, [_vp, _vp, _vp, _i64, _vp, _vp, _i32, _f32, _f32, _vp, _vp, _vp, _vp]),
    "npp_adam_step": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i64, _f32, _f32, _f32, _f32, _i32, _vp]),
    "npp_mlp_fwd_emb": (_i32, [_vp, _i64, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _vp]),
    "npp_mlp_bwd_act": (_i32, [_vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _vp]),
    "npp_mlp_bwd_patch": (_i32, [_vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "npp_mlp_bwd_patch_act": (_i32, [_vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "npp_grad_reduce": (_i32, [_vp, _i32, _i64, _i64, _vp, _i32, _vp]),
    "npp_fourier_fwd": (_i32, [_vp, _i64, _i32, C.POINTER(C.c_float), _i32, _i32, _vp, _vp]),
    "npp_adam_step_net": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i64, _vp, _vp, _vp, _vp, _i32, _vp, _i32,
                                 _f32, _f32, _f32, _f32, _i32, _vp]),
    "npp_adam_step_net_pack": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i64, _vp, _vp, _vp, _vp, _i32, _vp, _i32,
                                      _f32, _f32, _f32, _f32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "npp_pack_scatter_host": (_i32, [_vp, _vp, _vp, _i32, _i32]),
    "npp_adam_step_dev": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i64, _f32, _f32, _f32, _vp, _vp]),
    "npp_patch_gather": (_i32, [_vp, _vp, _i32, _i32, _vp, _i32, _i32, _vp, _vp, _vp]),
    "npp_batch_assemble": (_i32, [_vp, _i64, _vp, _i64, _vp, _i32, _i32, _i64, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp]),
    "npp_cx_workspace_bytes": (_i64, [_i32, _i32, _i32]),
    "npp_cx_fwd_bwd": (_i32, [_vp, _vp, _i32, _i32, _i32, _f32, _vp, _f32, _vp, _vp, _vp, _i64, _vp]),
    "npp_lpips_workspace_bytes": (_i64, [_i32]),
    "npp_lpips_layer": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _f32, _f32, _vp, _vp, _vp, _vp, _vp]),
    "npp_embed_dev_bytes": (_i32, []),
    "npp_embed_dev_build": (_i32, [_cfgp, _vp]),
    "npp_mlp_fwd_stack": (_i32, [_vp, _i64, _vp, _i32, _i32, _i32, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _vp]),
    "npp_trunk_patch_in_loss_stack": (_i32, [_vp, _i64, _i64, _vp, _i64, _vp, _i64, _i32, _i32, _i32, _i32, _i32, C.POINTER(C.c_float),
                                             C.POINTER(C.c_float), _vp, _vp, _i64, _vp, _vp, C.POINTER(PixelLossArgs), _i64, _i32, _i32, _i64, _vp]),
    "npp_cx_fwd_bwd_groups": (_i32, [_vp, _vp, _i32, _i32, _i32, _f32, _f32, _vp, _i32, _vp, _vp, _i32, _vp, _i64, _vp]),
    "npp_cx_fwd_bwd_flat": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _f32, _f32, _vp, _i32, _vp, _vp, _vp, _i32, _vp, _i32, _vp, _i64, _vp]),
    "npp_mlp_bwd_patch_stack": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp,
                                       _i64, _i64, _i32, _i32, _vp, _vp]),
    "npp_mlp_wgrad_stack": (_i32, [_vp, _i64, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _vp, _i64, _vp, _vp]),
    "npp_adam_step_net_pack_stack": (_i32, [_vp, _vp, _vp, _i64, _vp, _i64, _i32, _i64, _i64, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _i32,
                                            _i32, _f32, _f32, _f32, _i32, _i32, _i32, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp]),
    "npp_selftest_mfma": (_i32, [_vp, _vp]),
    "npp_shift_search": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp, _i32, _i32, _vp, _vp]),
    "npp_rng_create": (_vp, [C.c_uint32]),
    "npp_rng_destroy": (None, [_vp]),
    "npp_rng_seed": (_i32, [_vp, C.c_uint32]),
    "npp_rng_get_state": (_i32, [_vp, _vp, C.POINTER(C.c_int32)]),
    "npp_rng_set_state": (_i32, [_vp, _vp, C.c_int32]),
    "npp_rng_uniform": (C.c_double, [_vp, C.c_double, C.c_double]),
    "npp_rng_choice_noreplace": (_i32, [_vp, _i64, _i64, _vp, _vp]),
    "npp_sampler_create": (_vp, [_vp, _i32, _i32, _vp, _i64, _vp, _i64, _vp]),
    "npp_sampler_destroy": (None, [_vp]),
    "npp_sampler_set_patch": (_i32, [_vp, _i32, _i32, C.POINTER(_i64), C.POINTER(_i64)]),
    "npp_sampler_draw": (_i32, [_vp, _vp, _i32, C.c_double, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _vp, _vp, _vp]),
    "npp_linear_fwd": (_i32, [_vp, _i64, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _i64, _vp, _i64, _vp]),
    "npp_linear_bwd_data": (_i32, [_vp, _i64, _vp, _i64, _i32, _i32, _vp, _i64, _i32, _i32, _vp]),
    "npp_linear_bwd_weight": (_i32, [_vp, _i64, _vp, _i64, _i64, _i32, _i32, _vp, _vp, _i32, _vp]),
    "npp_linear_fwd_batched": (_i32, [_vp, _i64, _i64, _vp, _i64, _vp, _i64, _i32, _i64, _i32, _i32, _i32, _vp, _i64, _i64, _vp, _i64, _i64, _vp]),
    "npp_linear_bwd_data_batched": (_i32, [_vp, _i64, _i64, _vp, _i64, _i32, _i64, _i32, _i32, _vp, _i64, _i64, _i32, _i32, _vp, _i64, _i64, _i32, _vp]),
    "npp_linear_bwd_weight_batched": (_i32, [_vp, _i64, _i64, _vp, _i64, _i64, _i32, _i64, _i32, _i32, _vp, _i64, _vp, _i64, _vp]),
    "npp_linear_bwd_weight_strided": (_i32, [_vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _i32, _i32, _i64, _i32, _i32, _vp, _i64, _i64, _vp, _i64, _vp]),
    "npp_light_pack_floats": (_i64, []),
    "npp_light_stash_rows": (_i64, []),
    "npp_light_dstash_rows": (_i64, []),
    "npp_light_stash_row": (_i32, [_i32]),
    "npp_light_dstash_row": (_i32, [_i32]),
    "npp_light_pack": (_i32, [C.POINTER(LightDesc), _vp, _i64, _i32, _vp, _i64, _vp]),
    "npp_light_fwd": (_i32, [C.POINTER(LightDesc), _vp, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _i32, _i64, _vp, _vp, _vp]),
    "npp_light_adam_pack": (_i32, [C.POINTER(LightDesc), _vp, _vp, _vp, _vp, _i64, _i64, _i32, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _f32, _i32, _vp]),
    "npp_light_wgrad": (_i32, [C.POINTER(LightDesc), _vp, _vp, _i32, _i64, _vp, _i64, _vp]),
    "npp_lpips_layers": (_i32, [_i32, C.POINTER(LpipsTap), _i32, _vp, _i32, _f32, _f32, _vp, _vp]),
    "npp_pixel_loss_quad": (_i32, [_vp, _vp, _i64, _vp, _i64, _i32, _f32, _f32, _vp, _vp, _vp]),
    "npp_light16_pack_bytes": (_i64, []),
    "npp_light16_stash_bytes": (_i64, [_i64, _i32]),
    "npp_light16_pack": (_i32, [C.POINTER(LightDesc), _vp, _i64, _i32, _vp, _i64, _vp]),
    "npp_light16_fwd": (_i32, [C.POINTER(LightDesc), _vp, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _i32, _i64, _vp, _i64, _vp, _vp]),
    "npp_light16_bwd": (_i32, [C.POINTER(LightDesc), _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _f32, _vp, _vp, _i32, _i64, _vp, _i64, _vp]),
    "npp_light16_wgrad": (_i32, [C.POINTER(LightDesc), _vp, _i64, _vp, _i64, _i32, _i64, _i32, _vp, _i64, _i64, _vp]),
    "npp_light16_adam_pack": (_i32, [C.POINTER(LightDesc), _vp, _vp, _vp, _i64, _i64, _i32, _vp, _i32, _i64, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _f32, _i32, _vp]),
    "npp_light16_fwd_multi": (_i32, [C.POINTER(LightDesc), _vp, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _i32, _i64, _vp, _i64, _vp, _vp]),
    "npp_light16_bwd_det": (_i32, [C.POINTER(LightDesc), _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _i32, _f32, _vp, _i32, _i64, _vp, _i64, _vp]),
    "npp_light16_adam_pack_det": (_i32, [C.POINTER(LightDesc), _vp, _vp, _vp, _i64, _i64, _i32, _vp, _i32, _i64, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _f32, _i32, _vp, _i32, _vp, _vp]),
    "npp_light_part_blocks": (_i32, [_i32, _i64]),
    "npp_light_wgrad_det_scratch_bytes": (_i64, [_i32, _i64]),
    "npp_light_wgrad_det": (_i32, [C.POINTER(LightDesc), _vp, _vp, _i32, _i64, _vp, _i64, _vp, _i64, _vp]),
    "npp_light_fwd_multi": (_i32, [C.POINTER(LightDesc), _vp, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _i32, _i64, _vp, _vp, _vp]),
    "npp_light_bwd_det_multi": (_i32, [C.POINTER(LightDesc), _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _f32, _vp, _i32, _i64, _vp, _vp, _vp]),
    "npp_light_bwd_det": (_i32, [C.POINTER(LightDesc), _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _f32, _vp, _i32, _i64, _vp, _vp, _vp]),
    "npp_light_adam_pack_det": (_i32, [C.POINTER(LightDesc), _vp, _vp, _vp, _vp, _i64, _i64, _i32, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _f32,
                                       _i32, _vp, _i32, _vp, _vp]),
    "npp_light_bwd": (_i32, [C.POINTER(LightDesc), _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _f32, _vp, _vp, _i32, _i64, _vp, _vp, _vp]),
    "npp_pixel_loss_batched": (_i32, [_vp, _vp, _i64, _i64, _i32, _vp, _vp, _i32, _f32, _f32, _vp, _vp, _vp, _vp]),
    "npp_act_bwd": (_i32, [_vp, _i64, _vp, _i64, _i64, _i32, _i32, _vp, _i64, _vp]),
    "npp_act_fwd": (_i32, [_vp, _i64, _i32, _vp, _vp]),
    "npp_lpips_plain_layer": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp, _f32, _vp, _vp]),
    "npp_lpips_plain_layer_det": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp, _f32, _vp, _vp, _vp]),
    "npp_patch_compose_fwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "npp_patch_compose_bwd": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "npp_trunk_nposp": (_i64, [_i32, _i32, _i32]),
    "npp_trunk_act_bytes": (_i64, [_i32, _i32, _i32, _i32]),
    "npp_conv_pack_bytes": (_i64, [_i32, _i32, _i32]),
    "npp_conv_pack": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "npp_trunk_image_in": (_i32, [_vp, _i32, _i32, _i32, C.POINTER(C.c_float), C.POINTER(C.c_float), _vp, _vp]),
    "npp_trunk_patch_in": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, C.POINTER(C.c_float), C.POINTER(C.c_float),
                           _vp, _vp, _vp, _i32, _i32, _vp]),
    "npp_trunk_patch_in_loss": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, C.POINTER(C.c_float), C.POINTER(C.c_float),
                           _vp, _vp, _vp, _i32, _i32, _vp, _vp]),
    "npp_conv3x3": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _i32,
                           C.POINTER(C.c_float), _vp]),
    "npp_conv3x3_pf": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _i32,
                           C.POINTER(C.c_float), _vp, _i64, _vp]),
    "npp_conv3x3_pool": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i32, C.POINTER(C.c_float), _vp,
                         _i64, _vp]),
    "npp_conv_pair_fwd_ok": (_i32, [_i32, _i32, _i32, _i32, _i32]),
    "npp_conv_pair_fwd": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "npp_conv_pair_fwd_patch": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                 _vp, _i32, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "npp_conv_pair_dgrad_ok": (_i32, [_i32, _i32, _i32]),
    "npp_conv_pair_dgrad": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, C.POINTER(C.c_float), _vp]),
    "npp_conv3x3_dgrad_pool": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "npp_maxpool2_fwd": (_i32, [_vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "npp_maxpool2_bwd": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "npp_trunk_grad_in": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _vp]),
    "npp_trunk_grad_in_pf": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _vp, _i64, _vp]),
    "npp_gram_fwd": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp]),
    "npp_gram_fwd_det_scratch_bytes": (_i64, [_i32, _i32, _i32]),
    "npp_gram_fwd_det": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _i64, _vp]),
    "npp_gram_bwd": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "npp_robust_elem_workspace_bytes": (_i64, [_i32]),
    "npp_robust_elem": (_i32, [_vp, _vp, _i32, _i32, _vp, _vp, _i32, _f32, C.POINTER(C.c_float), _vp, _vp, _vp, _vp, _vp, _vp]),
    "npp_trunk_export": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp]),
    "npp_im2col": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "npp_maxpool_nhwc": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "npp_lpips_spatial_layer": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp, _vp]),
    "npp_resize_bilinear": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
}

_LIBS = {}


def lib(width=NPP_WIDTH):
    """Load the library built for `width` (libnpp_hip.so: 256, libnpp_hip_w512.so: 512) once and type every entry point.
    Raises if the library or any declared symbol is missing -- the product has no other implementation to fall back to."""
    L = _LIBS.get(width)
    if L is not None:
        return L
    path = LIB_PATHS.get(width)
    if path is None:
        raise NppError(f"no fused library for width {width} (built: {FUSED_WIDTHS}); other widths run through npp_amd.dense")
    if not os.path.exists(path):
        raise NppError(f"{path} not found: run `python __graft_entry__.py` (hipcc --offload-arch=gfx950) first; "
                       "there is no CPU fallback")
    # PyTorch-ROCm ships its own HIP runtime (torch/lib/libamdhip64.so).  It must be in the process BEFORE this library
    # is loaded, so that the library's dependency resolves to the same runtime instance torch allocates memory and creates
    # streams with; loaded the other way round, the system runtime comes in first and the library's launches fail with
    # "no ROCm-capable device is detected" (seen with build() followed by smoke() in one process).
    import torch  # noqa: F401
    try:
        L = C.CDLL(path)
    except OSError as e:
        raise NppError(f"cannot load {path}: {e}") from e
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(L, name)
        except AttributeError as e:
            raise NppError(f"{path} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _LIBS[width] = L
    return L


def check(rc, what, width=NPP_WIDTH):
    if rc is not None and rc < 0:
        msg = lib(width).npp_last_error_string()
        raise NppError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
    return rc


def param_layout(K, width=NPP_WIDTH):
    """[(name, offset, rows, cols)], total floats -- the reference's state_dict order."""
    L = lib(width)
    names = (C.c_char_p * 32)()
    offs = (_i64 * 32)()
    rows = (C.c_int32 * 32)()
    cols = (C.c_int32 * 32)()
    total = _i64(0)
    n = check(L.npp_param_layout(K, width, names, offs, rows, cols, C.byref(total)), "npp_param_layout", width)
    out = []
    for i in range(n):
        out.append((names[i].decode(), int(offs[i]), int(rows[i]), int(cols[i])))
    return out, int(total.value)
