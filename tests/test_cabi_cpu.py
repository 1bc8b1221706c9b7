"""CPU-side checks of the C ABI: the library loads, exports every symbol the header
declares, reports the reference's parameter table, and its weight-fragment packing
(evaluated by the HOST twin of the device packer) reproduces W @ x under a NumPy model
of the documented v_mfma_f32_32x32x16_bf16 lane maps.  No GPU calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle
import npp_amd
from npp_amd._lib import SYMBOLS, param_layout, check

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "npp_hip.h")).read()
    declared = set(re.findall(r"\b(npp_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(SYMBOLS), declared ^ set(SYMBOLS)
    for width in npp_amd.FUSED_WIDTHS:                   # libnpp_hip.so (W = 256) and libnpp_hip_w512.so: same C ABI
        L = npp_amd.lib(width)
        for name in declared:
            assert hasattr(L, name)
        assert L.npp_version() >= 100


@pytest.mark.parametrize("K,W", [(1, 256), (3, 256), (5, 256), (3, 512), (1, 512)])
def test_param_layout_matches_reference_state_dict(K, W):
    lay, total = param_layout(K, W)
    shapes = oracle.param_shapes(K, W=W)
    assert [n for n, *_ in lay] == list(shapes)          # same tensors, same order
    off = 0
    for name, o, rows, cols in lay:
        assert o == off
        shp = shapes[name]
        assert rows == shp[0] and cols == (shp[1] if len(shp) == 2 else 1)
        off += int(np.prod(shp))
    assert off == total
    if K == 3 and W == 256:
        assert total == 1197572 - 257                     # SURVEY.md 8a5 minus the unused alpha_linear
    if K == 3 and W == 512:
        assert total == 3836932 - 513                     # the reference's default width


def test_bad_arguments_are_reported_not_crashed():
    L = npp_amd.lib()
    assert L.npp_pack_bytes(9, 256, 0) < 0
    assert L.npp_pack_bytes(3, 512, 0) < 0               # each library serves the width it was compiled for
    assert b"width" in L.npp_last_error_string()
    L5 = npp_amd.lib(512)
    assert L5.npp_pack_bytes(3, 256, 0) < 0 and b"width" in L5.npp_last_error_string()
    assert L5.npp_pack_bytes(3, 512, 0) > 4 * L.npp_pack_bytes(3, 256, 0) // 2
    # the folded-launch entry points validate on the host before anything is launched
    assert L.npp_mlp_bwd_patch(None, None, 64, 3, 256, None, None, None, None, None, None) < 0
    assert b"patch" in L.npp_last_error_string()
    three = (C.c_float * 3)(1, 1, 1)
    assert L.npp_trunk_patch_in_loss(None, None, None, None, None, 2, 3, 64, 0, three, three, None, None, None, 0, 0, None, None) < 0
    sizes = (C.c_int64 * 4)()
    assert L.npp_train_workspace(3, 256, 100, 4, sizes) < 0   # Bp not a multiple of 64
    assert L.npp_train_workspace(3, 256, 128, 4, sizes) == 0
    assert sizes[3] == 4 * 1197316 * 4                       # 4 slabs of the parameter count rounded up to 4 floats (16-byte aligned slabs)


def test_stash8_switch_and_its_host_side_checks():
    """npp_tune("stash8") (round 6: the 8-bit training stash, default on) reads / writes / rejects like the other keys; the forward
    stash of npp_train_workspace holds both formats; the 8-bit weight-gradient launch refuses a split whose per-tile scale words do
    not fit its LDS -- on the host, before anything is launched -- and the deterministic f1 entry points check their scratch."""
    L = npp_amd.lib()
    assert L.npp_tune(b"stash8", -1) == 1 and L.npp_tune(b"light_det", -1) == 1
    assert L.npp_tune(b"stash8", 0) == 1 and L.npp_tune(b"stash8", -1) == 0 and L.npp_tune(b"stash8", 1) == 0
    assert L.npp_tune(b"no_such_key", 1) < 0
    sizes = (C.c_int64 * 4)()
    assert L.npp_train_workspace(3, 256, 64, 1, sizes) == 0
    n_ks = 11 * 16 + 8 + 3 * 30                                        # k-steps of the forward arrays (csrc/npp_layout.h act_total_ks)
    assert sizes[1] == n_ks * (2048 + 1024) and sizes[2] == (11 * 16 + 8 + 2) * 2048
    fake = C.c_void_p(64)                                             # never dereferenced: validation comes first
    assert L.npp_mlp_wgrad(fake, fake, 64 * 65536, 3, 256, 1, fake, None) < 0
    assert b"raise ksplit" in L.npp_last_error_string()
    assert L.npp_lpips_plain_layer_det(fake, fake, 1, 64, 100, fake, 1.0, fake, None, None) < 0
    assert L.npp_light_part_blocks(9, 100) < 0 and L.npp_light_part_blocks(9, 2048) in (32, 64) and L.npp_light_part_blocks(64, 2048) == 32


def test_light_chain_entry_points_validate_on_the_host():
    """The fused NPP_Net_light entry points refuse topologies / shapes they are not built for before anything is launched (no GPU
    needed to see it), and their layout queries are consistent."""
    from npp_amd._lib import LightDesc
    L = npp_amd.lib()
    # forward pack: 32 (20 real) x 256, four 256 x 256, 304 (298 real) x 128; transposed pack: 128 x 256 and four 256 x 256
    assert L.npp_light_pack_floats() == (32 * 256 + 4 * 65536 + 304 * 128) + (128 * 256 + 4 * 65536)
    rows = [L.npp_light_stash_row(i) for i in range(8)]
    assert rows[0] == 0 and rows == sorted(rows) and rows[7] == L.npp_light_stash_rows() and L.npp_light_stash_row(8) < 0
    drows = [L.npp_light_dstash_row(i) for i in range(8)]
    assert drows == sorted(drows) and drows[7] == L.npp_light_dstash_rows()
    d = LightDesc()
    n_out, n_in = [256, 256, 256, 256, 128, 256, 3], [20, 256, 256, 256, 298, 256, 128]
    for i in range(7):
        d.n_out[i], d.n_in[i], d.ld[i] = n_out[i], n_in[i], (n_in[i] + 3) // 4 * 4
    fake = C.c_void_p(64)                                             # never dereferenced: validation comes first
    assert L.npp_light_fwd(C.byref(d), fake, 0, fake, 0, fake, fake, None, 100, 1, 100, fake, fake, None) < 0      # B not a multiple of 32
    assert b"multiple of 32" in L.npp_last_error_string()
    d.n_out[1] = 128                                                   # a width the chains are not built for
    assert L.npp_light_pack(C.byref(d), fake, 0, 1, fake, L.npp_light_pack_floats(), None) < 0
    assert b"fuses NPP_Net_light(D=4, W=256)" in L.npp_last_error_string()
    d.n_out[1] = 256
    assert L.npp_light_bwd(C.byref(d), fake, 0, fake, 0, fake, fake, None, None, None, None, 0, 0.0, None, None, 1, 64, fake, fake, None) < 0
    assert b"d_dpred" in L.npp_last_error_string()                     # neither d_dpred nor the folded loss's arguments
    assert L.npp_light_wgrad(C.byref(d), fake, fake, 0, 64, fake, 0, None) < 0
    assert L.npp_linear_bwd_weight_strided(fake, 0, 1, 0, fake, 1, 1, 0, 0, 1, 64, 4, 4, fake, 4, 0, None, 0, None) < 0


def test_light16_entry_points_validate_on_the_host():
    """The 16-bit candidate chains (csrc/npp_light16.hip): size queries against the documented layout, and the argument checks that come
    before any launch -- batch a multiple of 64, strides that hold one candidate's arrays, at most NPP_MAX_STACK candidates in the
    weight-gradient launch, leading dimensions a multiple of 4 there, the plain LPIPS head without a latent gradient."""
    from npp_amd._lib import LightDesc
    L = npp_amd.lib()
    fwd_units = 2 * 8 * 64 + 4 * 16 * 8 * 64 + 20 * 4 * 64          # [k-steps][neuron tiles][64 lanes] per layer: x_per, 4 x 256 wide, pos
    bwd_units = 8 * 8 * 64 + 4 * 16 * 8 * 64                          # transposed: pos (128 neurons), feature1 + three periodic layers
    assert L.npp_light16_pack_bytes() == 16 * (fwd_units + bwd_units)
    assert L.npp_light16_stash_bytes(2048, 0) == 94 * (2048 // 64) * 2048      # z_0..z_3 (64 k-steps) + [f1 | x_pos] (20) + z_p (8) + x_per (2)
    assert L.npp_light16_stash_bytes(2048, 1) == 90 * (2048 // 64) * 2048      # dz_0..dz_3 (64) + d f1 (16) + d z_p (8) + d raw (2)
    assert L.npp_light16_stash_bytes(100, 0) < 0 and L.npp_light16_stash_bytes(64, 2) < 0
    d = LightDesc()
    n_out, n_in = [256, 256, 256, 256, 128, 256, 3], [20, 256, 256, 256, 298, 256, 128]
    off = 0
    for i in range(7):
        d.n_out[i], d.n_in[i], d.ld[i] = n_out[i], n_in[i], (n_in[i] + 3) // 4 * 4
        d.w_off[i] = off
        off += n_out[i] * d.ld[i]
        d.b_off[i] = off
        off += n_out[i]
    n_pad = (off + 3) // 4 * 4
    fake = C.c_void_p(64)
    pb, ab, db = L.npp_light16_pack_bytes(), L.npp_light16_stash_bytes(64, 0), L.npp_light16_stash_bytes(64, 1)
    assert L.npp_light16_fwd(C.byref(d), fake, 0, fake, pb, fake, fake, None, 96, 1, 96, fake, ab, fake, None) < 0           # B = 96
    assert b"multiple of 64" in L.npp_last_error_string()
    assert L.npp_light16_fwd(C.byref(d), fake, 0, fake, pb, fake, fake, None, 64, 1, 64, fake, ab - 16, fake, None) < 0      # stash stride too small
    assert L.npp_light16_bwd(C.byref(d), fake, 0, fake, pb, fake, ab, fake, None, None, None, None, 0, 0.0, None, None, 1, 64, fake, db, None) < 0
    assert b"d_dpred" in L.npp_last_error_string()
    assert L.npp_light16_wgrad(C.byref(d), fake, ab, fake, db, 17, 64, 1, fake, n_pad, n_pad, None) < 0                     # 17 candidates
    assert b"<= 16" in L.npp_last_error_string()
    d.ld[4] = 298                                                                                                           # pos_linears.0 stored 298 wide
    assert L.npp_light16_wgrad(C.byref(d), fake, ab, fake, db, 2, 64, 1, fake, n_pad, n_pad, None) < 0
    d.ld[4] = 300
    assert L.npp_light16_adam_pack(C.byref(d), fake, fake, fake, n_pad, off, 1, fake, 0, n_pad, n_pad, fake, pb, fake, fake, fake, fake, None,
                                   5e-4, 0.9, 0.999, 1e-8, 1, None) < 0                                                   # no slabs
    assert L.npp_pixel_loss_quad(fake, fake, 0, None, 64, 1, 0.0, 1.0, fake, fake, None) < 0                                # coef must be > 0
    assert L.npp_lpips_layer(fake, fake, 1, 64, 16, fake, None, None, 0, 0.0, 1.0, fake, fake, fake, None, None) < 0        # plain head: no dlatent
    assert b"plain head" in L.npp_last_error_string()
    assert L.npp_mlp_fwd_act(fake, 64, None, 256, fake, fake, fake, None, 1, None) < 0                                     # null embedder config


# ---- NumPy model of the MFMA fragment maps (cdna_hip_programming.md section 3) ----
def perm16(h, j):
    return 8 * (j >> 2) + 4 * h + (j & 3)


def emb_col(ks, h, j):
    if ks < 28:
        t = 8 * ks + j
        if t >= 220:
            return -1
        return (1 + 2 * (t // 22) + h) * 22 + t % 22
    if ks == 28:
        return 8 * h + j
    return 16 + j if (h == 0 and j < 6) else -1


def bf16_bits_to_f32(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32)


def mfma_tile(a_frag, b_frag):
    """a_frag, b_frag: (64, 8) per-lane operand fragments -> D[32][32] with
    D[m][n] = sum over (h, j) of A[lane m + 32h][j] * B[lane n + 32h][j]."""
    A = a_frag.reshape(2, 32, 8)
    B = b_frag.reshape(2, 32, 8)
    return np.einsum("hmj,hnj->mn", A.astype(np.float64), B.astype(np.float64))


def pack_host(P_flat, K):
    L = npp_amd.lib()
    nf, nb = L.npp_pack_bytes(K, 256, 0), L.npp_pack_bytes(K, 256, 1)
    wf = np.zeros(nf // 2, np.uint16)
    wb = np.zeros(nb // 2, np.uint16)
    check(L.npp_pack_weights_host(P_flat.ctypes.data, wf.ctypes.data, wb.ctypes.data, K, 256), "pack_host")
    return bf16_bits_to_f32(wf).reshape(-1, 64, 8), bf16_bits_to_f32(wb).reshape(-1, 64, 8)


def flat_params(P, K):
    lay, total = param_layout(K)
    flat = np.zeros(total, np.float32)
    for name, o, rows, cols in lay:
        flat[o:o + rows * cols] = P[name].reshape(-1)
    return flat, {n: o for n, o, *_ in lay}


def act_frags(X):
    """X: (feat, 32) activation tile set -> fragments [ks][64][8] in accumulator-chain order."""
    ks_n = X.shape[0] // 16
    out = np.zeros((ks_n, 64, 8), np.float32)
    for ks in range(ks_n):
        for h in range(2):
            for j in range(8):
                out[ks, 32 * h:32 * h + 32, j] = X[16 * ks + perm16(h, j), :]
    return out


def emb_frags(E):
    """E: (462, 32) one proposal's embedding (reference column order) -> [30][64][8]."""
    out = np.zeros((30, 64, 8), np.float32)
    for ks in range(30):
        for h in range(2):
            for j in range(8):
                c = emb_col(ks, h, j)
                if c >= 0:
                    out[ks, 32 * h:32 * h + 32, j] = E[c, :]
    return out


def test_emb_slot_order_is_a_bijection_onto_the_462_columns():
    cols = [emb_col(ks, h, j) for ks in range(30) for h in range(2) for j in range(8)]
    real = [c for c in cols if c >= 0]
    assert sorted(real) == list(range(462)) and len(cols) == 480


@pytest.mark.parametrize("K", [3, 1])
def test_forward_pack_reproduces_linear_layers(K):
    rng = np.random.RandomState(0)
    P = oracle.init_params(K, seed=3)
    P = {k: oracle.bf16_round(v) for k, v in P.items()}       # exact in bf16: pack is lossless
    flat, _ = flat_params(P, K)
    wf, _ = pack_host(flat, K)
    W = 256
    unit = 0
    # layer order and k-step structure of the forward pack (npp_layout.h make_desc)
    layers = [("periodic_linears.0", ["emb0"])] + [(f"periodic_linears.{i}", ["act"]) for i in range(1, 5)]
    layers += [("periodic_linears.5", ["emb0", "act"]), ("periodic_linears.6", ["act"]),
               ("periodic_linears.7", ["act"]), ("feature_linear1", ["act"])]
    if K > 1:
        layers += [("scale_linears.0", ["act"] + [f"emb{p}" for p in range(1, K)]), ("feature_linear2", ["act"]),
                   ("pos_linears.0", ["act", "act"])]
    else:
        layers += [("pos_linears.0", ["act"])]
    for name, parts in layers:
        Wm = P[name + ".weight"]
        nt_n = Wm.shape[0] // 32
        # a random input in the REFERENCE column order, plus its fragments in kernel order
        x_ref, frags = [], []
        for part in parts:
            if part.startswith("emb"):
                E = rng.randn(462, 32).astype(np.float32)
                frags.append(emb_frags(E))
                x_ref.append(E)
            else:
                X = rng.randn(W, 32).astype(np.float32)
                frags.append(act_frags(X))
                x_ref.append(X)
        # reference concat orders: L5 [emb, h] (networks.py:71), S [f1, aux] (:76), P [f1, f2] (:85)
        x_ref = np.concatenate(x_ref, 0)
        frags = np.concatenate(frags, 0)
        assert x_ref.shape[0] == Wm.shape[1], name
        ks_n = frags.shape[0]
        out = np.zeros((Wm.shape[0], 32))
        for ks in range(ks_n):
            for nt in range(nt_n):
                out[nt * 32:(nt + 1) * 32] += mfma_tile(wf[unit + ks * nt_n + nt], frags[ks])
        unit += ks_n * nt_n
        np.testing.assert_allclose(out, Wm.astype(np.float64) @ x_ref.astype(np.float64), rtol=1e-9, atol=1e-9,
                                   err_msg=name)
    assert unit == wf.shape[0]


@pytest.mark.parametrize("K", [3, 1])
def test_backward_pack_reproduces_transposed_products(K):
    rng = np.random.RandomState(1)
    P = {k: oracle.bf16_round(v) for k, v in oracle.init_params(K, seed=4).items()}
    flat, _ = flat_params(P, K)
    _, wb = pack_host(flat, K)
    E = 462
    # virtual dgrad layers in kernel order (npp_layout.h BwdLayer): (weight, first input column)
    v = [("pos_linears.0", 0)]
    if K > 1:
        v += [("pos_linears.0", 256), ("feature_linear2", 0), ("scale_linears.0", 0)]
    v += [("feature_linear1", 0), ("periodic_linears.7", 0), ("periodic_linears.6", 0), ("periodic_linears.5", E),
          ("periodic_linears.4", 0), ("periodic_linears.3", 0), ("periodic_linears.2", 0), ("periodic_linears.1", 0)]
    unit = 0
    for name, col0 in v:
        Wm = P[name + ".weight"]
        n_out = Wm.shape[0]
        dz = rng.randn(n_out, 32).astype(np.float32)
        frags = act_frags(dz)                       # contraction index = the layer's output neurons
        ns_n = n_out // 16
        out = np.zeros((256, 32))
        for ns in range(ns_n):
            for kt in range(8):
                out[kt * 32:(kt + 1) * 32] += mfma_tile(wb[unit + ns * 8 + kt], frags[ns])
        unit += ns_n * 8
        ref = Wm[:, col0:col0 + 256].astype(np.float64).T @ dz.astype(np.float64)
        np.testing.assert_allclose(out, ref, rtol=1e-9, atol=1e-9, err_msg=f"{name}@{col0}")
    assert unit == wb.shape[0]


@pytest.mark.parametrize("K,width", [(1, 256), (3, 256), (5, 256), (3, 512)])
def test_pack_scatter_inverse_maps_reproduce_the_packers(K, width):
    """npp_adam_step_net_pack scatters every updated weight into the two bf16 packs through the inverse maps of
    csrc/npp_layout.h (fwd_pack_pos / bwd_pack_pos); their host twin must rebuild exactly what npp_pack_weights_host builds
    (every real element hit once, padding left zero), at both compiled widths."""
    L = npp_amd.lib(width)
    _, total = param_layout(K, width)
    rng = np.random.RandomState(K * 7 + width)
    flat = (rng.randn(total) * 0.3).astype(np.float32)
    nf, nb = L.npp_pack_bytes(K, width, 0), L.npp_pack_bytes(K, width, 1)
    a_f, a_b = np.zeros(nf // 2, np.uint16), np.zeros(nb // 2, np.uint16)
    s_f, s_b = np.full(nf // 2, 0x7fc0, np.uint16), np.full(nb // 2, 0x7fc0, np.uint16)
    check(L.npp_pack_weights_host(flat.ctypes.data, a_f.ctypes.data, a_b.ctypes.data, K, width), "pack_host")
    check(L.npp_pack_scatter_host(flat.ctypes.data, s_f.ctypes.data, s_b.ctypes.data, K, width), "pack_scatter_host")
    assert np.array_equal(a_f, s_f)
    assert np.array_equal(a_b, s_b)


def test_pool_fold_entry_points_validate_on_the_host():
    """The pools folded into their neighbouring convolutions (csrc/npp_conv.hip) and the fused layer pair (csrc/npp_conv_pair.hip): the
    shape query and the argument checks that come before any launch."""
    from npp_amd._lib import LpipsTap
    L = npp_amd.lib()
    fake = C.c_void_p(64)
    # the fused layer pair (csrc/npp_conv_pair.hip): the shape query, and the argument checks in front of the launch
    assert L.npp_conv_pair_fwd_ok(96, 96, 16, 64, 64) == 1
    assert L.npp_conv_pair_fwd_ok(95, 96, 16, 64, 64) == 0                  # the pool needs even sizes
    assert L.npp_conv_pair_fwd_ok(48, 48, 64, 128, 128) == 1 and L.npp_conv_pair_fwd_ok(24, 24, 128, 256, 256) == 0
    assert L.npp_conv_pair_fwd(fake, 4, 4, 2, 24, 24, 128, 256, 256, fake, fake, fake, fake, fake, fake, fake, None, None) < 0
    assert b"npp_conv_pair_fwd_ok" in L.npp_last_error_string()
    assert L.npp_conv_pair_fwd(fake, 4, 5, 2, 96, 96, 16, 64, 64, fake, fake, fake, fake, fake, fake, fake, None, None) < 0   # n_run > N_total
    assert L.npp_conv_pair_fwd(fake, 4, 4, 2, 96, 96, 16, 64, 64, fake, fake, fake, fake, None, fake, fake, None, None) < 0   # kept images need y_a
    assert 0 <= L.npp_tune(b"conv_pair", -1) <= 15 and L.npp_tune(b"no_such_key", 0) < 0
    assert b"unknown key" in L.npp_last_error_string()
    assert L.npp_conv3x3_pool(fake, 2, 2, 47, 48, 64, 64, fake, fake, fake, fake, None, 0, None, None, 0, None) < 0            # odd H
    assert b"even" in L.npp_last_error_string()
    assert L.npp_conv3x3_pool(fake, 2, 2, 48, 48, 64, 64, fake, fake, fake, None, None, 0, None, None, 0, None) < 0            # no pooled output
    assert L.npp_conv3x3_dgrad_pool(fake, 2, 2, 24, 24, 128, 64, fake, None, None, fake, None, 0, None) < 0                    # no pre-pool tensor
    assert b"pre-pool" in L.npp_last_error_string()
    assert L.npp_conv3x3_dgrad_pool(fake, 2, 3, 24, 24, 128, 64, fake, fake, None, fake, None, 0, None) < 0                    # n_run > N_total
    # npp_lpips_tap.dflat takes the place of df0 and needs the tap's geometry
    t = (LpipsTap * 1)()
    t[0] = LpipsTap(64, 64, 64, 16, 64, None, 64, None, None, 64, 2, 4, 4, None)                 # df0 AND dflat
    assert L.npp_lpips_layers(1, t, 2, None, 0, 0.0, 1.0, fake, None) < 0
    assert b"flat gradient" in L.npp_last_error_string()
    t[0] = LpipsTap(64, 64, 64, 16, 64, None, None, None, None, 64, 2, 4, 5, None)               # H * W != hw
    assert L.npp_lpips_layers(1, t, 2, None, 0, 0.0, 1.0, fake, None) < 0
    t[0] = LpipsTap(64, 64, 64, 16, 64, None, None, None, None, None, 2, 4, 4, 64)               # yact without dflat
    assert L.npp_lpips_layers(1, t, 2, None, 0, 0.0, 1.0, fake, None) < 0
