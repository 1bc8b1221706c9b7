"""SURVEY.md 8 f1 on the GPU: the generic dense-layer kernels (npp_linear_*), NPP_Net_light on them, the is_search
embedders, the plain LPIPS head and the candidate fit / ranking loop, through the C ABI.  References: NumPy for the dense
layers (fp32 round-off), the reference's own modules through tests/golden/g10_light.npz for everything else."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import npp_amd
    npp_amd.lib()
    return torch.device("cuda:0")


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.mark.parametrize("B,cin,cout", [(1, 1, 1), (70, 20, 64), (257, 106, 32), (2048, 256, 256), (33, 298, 128), (100, 128, 3)])
def test_dense_layer_kernels_vs_numpy(dev, B, cin, cout):
    """forward (bias, snake, pre-activation copy, strided output), data gradient (partial columns, accumulate), weight + bias
    gradient, activation backward: exact-fp32 MFMA, so agreement is fp32 round-off."""
    from npp_amd import ops
    rng = np.random.RandomState(B + cin)
    x, w, b = rng.randn(B, cin).astype(np.float32), (rng.randn(cout, cin) / np.sqrt(cin)).astype(np.float32), rng.randn(cout).astype(np.float32)
    xb = torch.zeros(B, cin + 5, device=dev)
    xb[:, 2:2 + cin] = torch.from_numpy(x).to(dev)                              # x as a column block of a wider buffer
    xv = xb[:, 2:2 + cin]
    wt, bt = torch.from_numpy(w).to(dev), torch.from_numpy(b).to(dev)
    yb = torch.full((B, cout + 3), 7.0, device=dev)
    z = torch.empty(B, cout, device=dev)
    ops.linear_fwd(xv, wt, bt, 1, yb[:, 1:1 + cout], z)
    zr = x.astype(np.float64) @ w.T.astype(np.float64) + b
    np.testing.assert_allclose(z.cpu().numpy(), zr, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(yb[:, 1:1 + cout].cpu().numpy(), zr + np.sin(zr) ** 2, rtol=3e-5, atol=3e-5)
    assert float(yb[:, 0].min()) == 7.0 and float(yb[:, 1 + cout:].min()) == 7.0       # neighbours untouched
    ops.linear_fwd(xv, wt, None, 0, yb[:, 1:1 + cout])
    np.testing.assert_allclose(yb[:, 1:1 + cout].cpu().numpy(), zr - b, rtol=2e-5, atol=2e-5)
    dz = rng.randn(B, cout).astype(np.float32)
    dzt = torch.from_numpy(dz).to(dev)
    used = max(1, cin - 3)
    dx = torch.ones(B, used, device=dev)
    ops.linear_bwd_data(dzt, wt, dx, in_used=used, accumulate=True)
    np.testing.assert_allclose(dx.cpu().numpy(), 1.0 + (dz.astype(np.float64) @ w.astype(np.float64))[:, :used], rtol=3e-5, atol=3e-5)
    dw, db = torch.empty(cout, cin, device=dev), torch.empty(cout, device=dev)
    ops.linear_bwd_weight(dzt, xv, dw, db)
    scale = np.sqrt(B)
    np.testing.assert_allclose(dw.cpu().numpy(), dz.T.astype(np.float64) @ x.astype(np.float64), rtol=1e-4, atol=3e-5 * scale)
    np.testing.assert_allclose(db.cpu().numpy(), dz.sum(0, dtype=np.float64), rtol=1e-4, atol=3e-5 * scale)
    g = torch.empty(B, cout, device=dev)
    ops.act_bwd(dzt, z, 1, g)
    np.testing.assert_allclose(g.cpu().numpy(), dz * (1 + np.sin(2 * zr)), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("C,B,cin,cout", [(3, 70, 106, 32), (2, 257, 20, 64), (9, 2048, 256, 256), (4, 33, 300, 128), (2, 100, 128, 3)])
def test_batched_dense_layers_equal_the_single_problem_launches(dev, C, B, cin, cout):
    """npp_linear_*_batched (the candidate is a grid dimension) against one npp_linear_* launch per problem: forward and data
    gradient bit-identical (same tiles, same k order), the data gradient with the fused activation derivative against
    linear_bwd_data + act_bwd, the split weight / bias gradient to fp32 round-off; strided batches (rows of a stacked blob),
    16-byte-aligned and unaligned leading dimensions."""
    from npp_amd import ops
    g = torch.Generator(device="cpu").manual_seed(C * 1000 + B)
    blob = (torch.randn(C, cout * cin + cout + 5, generator=g) / np.sqrt(cin)).to(dev)           # weights + biases as rows of one blob
    w, b = blob[:, :cout * cin].unflatten(1, (cout, cin)), blob[:, cout * cin:cout * cin + cout]
    x = torch.randn(C, B, cin, generator=g).to(dev)
    y, z = torch.empty(C, B, cout, device=dev), torch.empty(C, B, cout, device=dev)
    ops.linear_fwd_batched(x, w, b, 1, y, z)
    y1, z1 = torch.empty(B, cout, device=dev), torch.empty(B, cout, device=dev)
    for c in range(C):
        ops.linear_fwd(x[c], w[c].contiguous(), b[c].contiguous(), 1, y1, z1)
        assert torch.equal(y[c], y1) and torch.equal(z[c], z1)
    xs = x[0:1].expand(C, B, cin)                                                               # one input shared by the problems
    ops.linear_fwd_batched(xs, w, b, 2, y)
    ops.linear_fwd(x[0], w[C - 1].contiguous(), b[C - 1].contiguous(), 2, y1)      # (a plain linear single launch may split its contraction)
    assert torch.equal(y[C - 1], y1)
    dz = torch.randn(C, B, cout, generator=g).to(dev)
    zin = torch.randn(C, B, cin, generator=g).to(dev)
    dx, dxf = torch.empty(C, B, cin, device=dev), torch.empty(C, B, cin, device=dev)
    ops.linear_bwd_data_batched(dz, w, dx)
    ops.linear_bwd_data_batched(dz, w, dxf, zy=zin, act=1)
    d1, d2 = torch.empty(B, cin, device=dev), torch.empty(B, cin, device=dev)
    for c in range(C):
        ops.linear_bwd_data(dz[c], w[c].contiguous(), d1)
        assert torch.equal(dx[c], d1)
        ops.act_bwd(d1, zin[c], 1, d2)
        assert torch.equal(dxf[c], d2)
    gblob = torch.zeros_like(blob)
    dw, db = gblob[:, :cout * cin].unflatten(1, (cout, cin)), gblob[:, cout * cin:cout * cin + cout]
    ops.linear_bwd_weight_batched(dz, x, dw, db)
    ops.linear_bwd_weight_batched(dz, x, dw, db)                                                # accumulates: twice the gradient
    w1, b1 = torch.empty(cout, cin, device=dev), torch.empty(cout, device=dev)
    for c in range(C):
        ops.linear_bwd_weight(dz[c], x[c], w1, b1)
        tol = 3e-5 * np.sqrt(B)
        np.testing.assert_allclose(dw[c].cpu().numpy(), 2 * w1.cpu().numpy(), rtol=1e-4, atol=tol)
        np.testing.assert_allclose(db[c].cpu().numpy(), 2 * b1.cpu().numpy(), rtol=1e-4, atol=tol)
    assert float(gblob[:, cout * cin + cout:].abs().max()) == 0.0                                # neighbours untouched


def test_batched_pixel_loss_equals_the_single_problem_launches(dev):
    from npp_amd import ops
    C, N = 5, 1000
    g = torch.Generator(device="cpu").manual_seed(5)
    pred, gt = torch.rand(C, N, 3, generator=g).to(dev), torch.rand(N, 3, generator=g).to(dev)
    lat = (torch.randn(C, 6, generator=g) * 0.5).to(dev)
    spline, n_knots, x_scale = ops.load_spline(dev)
    loss, dpred, dlat = torch.zeros(C, device=dev), torch.empty(C, N, 3, device=dev), torch.zeros(C, 6, device=dev)
    ops.pixel_loss_batched(pred, gt, lat, spline, n_knots, x_scale, 1.0, loss, dpred, dlat)
    for c in range(C):
        l1, d1, dl1 = torch.zeros(1, device=dev), torch.empty(N, 3, device=dev), torch.zeros(6, device=dev)
        ops.pixel_loss(pred[c], gt, None, lat[c].contiguous(), spline, n_knots, x_scale, 1.0, l1, d1, dl1)
        assert torch.equal(dpred[c], d1)
        np.testing.assert_allclose(float(loss[c]), float(l1), rtol=1e-5)
        np.testing.assert_allclose(dlat[c].cpu().numpy(), dl1.cpu().numpy(), rtol=1e-4, atol=1e-7)


def _P(g):
    return {k[3:]: g[k] for k in g.files if k.startswith("sd.")}


def test_search_embedders_vs_reference(dev, golden):
    from npp_amd.light import NPPNetLight, default_light_init
    g = golden("g10_light.npz")
    res = tuple(int(v) for v in g["res"])
    net = NPPNetLight(g["angles"], g["periods"], g["freqs"], res, default_light_init(64), W=64, device=dev)
    x_pos, x_per = net.embed(torch.from_numpy(g["coords"].astype(np.int32)).to(dev))
    np.testing.assert_allclose(x_pos.cpu().numpy(), g["pos_emb"], atol=2e-5)         # models/embedder.py:52-56, is_search
    np.testing.assert_allclose(x_per.cpu().numpy(), g["per_emb"], atol=2e-5)         # :84-88, include_input False


def test_light_net_vs_reference(dev, golden):
    """NPP_Net_light forward + every parameter gradient against the reference module's own autograd (g10_light.npz)."""
    from npp_amd.light import NPPNetLight
    g = golden("g10_light.npz")
    res = tuple(int(v) for v in g["res"])
    net = NPPNetLight(g["angles"], g["periods"], g["freqs"], res, _P(g), W=64, device=dev)
    x_pos, x_per = torch.from_numpy(g["pos_emb"]).to(dev), torch.from_numpy(g["per_emb"]).to(dev)
    pred = net.forward(x_pos, x_per)
    B = x_per.shape[0]
    np.testing.assert_allclose(net._ws[B]["raw"].cpu().numpy(), g["raw"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(pred.cpu().numpy(), g["pred"], atol=1e-5)
    dpred = 2.0 * (g["pred"] - g["tgt"]) / g["pred"].size                             # d mean((pred - tgt)^2) / dpred
    net._ws[B]["dpred"].copy_(torch.from_numpy(dpred.astype(np.float32)).to(dev))
    net.backward(B)
    G = net.grads()
    for name, _, _ in net.layout:
        for part in ("weight", "bias"):
            assert rel_l2(G[f"{name}.{part}"], g[f"grad.{name}.{part}"]) < 3e-4, (name, part)


def test_create_npp_net_is_search_dropin(dev, golden):
    """The reference's own call, create_npp_net(args, angles, periods, res, percep_net=None, is_search=True)
    (NPP_proposal/search.py:98-99, models/helpers.py:92-105), through the boundary module: embedders, NPP_Net_light forward
    and every parameter gradient by torch autograd over the dense-layer kernels, against the reference module's own numbers
    (g10_light.npz, W = 64)."""
    import types
    from npp_amd import reference_api as api
    g = golden("g10_light.npz")
    res = tuple(int(v) for v in g["res"])
    args = types.SimpleNamespace(multires=10, i_embed=0, freq_scales=[1], freq_offsets=[0, -1, 1, 0.5, -0.5], angle_offsets=[0],
                                 netdepth=4, netwidth=64, activation="snake", netchunk=1 << 22, lrate=5e-4, p_topk=1,
                                 normalize_type=1)
    torch.set_default_device(dev)
    try:
        kw, _, start, grad_vars, opt, emb, emb_per = api.create_npp_net(args, torch.tensor(g["angles"]), torch.tensor(g["periods"]),
                                                                          res, percep_net=None, is_search=True)
    finally:
        torch.set_default_device("cpu")
    model = kw["network_fn"]
    assert type(model).__name__ == "DenseNPPNetLight" and not isinstance(emb_per, list) and start == 0
    assert sorted(n for n, _ in model.named_parameters()) == sorted(k[3:] for k in g.files if k.startswith("sd."))
    emb.freq_bands = torch.from_numpy(g["freqs"].astype(np.float32))                # the golden's (seeded) Fourier frequencies
    c = torch.from_numpy(g["coords"].astype(np.float32)).to(dev)
    x_pos = emb.embed(c.clone())
    x_per = emb_per.embed(c)
    np.testing.assert_allclose(x_pos.cpu().numpy(), g["pos_emb"], atol=2e-5)
    np.testing.assert_allclose(x_per.cpu().numpy(), g["per_emb"], atol=2e-5)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in _P(g).items()})
    with torch.no_grad():
        np.testing.assert_allclose(model(x_pos, x_per).cpu().numpy(), g["raw"], rtol=2e-4, atol=2e-5)
    pred = api.render(x_pos, x_per, args, **kw)                                     # helpers.py:41-62 (sigmoid)
    np.testing.assert_allclose(pred.detach().cpu().numpy(), g["pred"], atol=1e-5)
    loss = ((pred - torch.from_numpy(g["tgt"]).to(dev)) ** 2).mean()
    loss.backward()
    for name, p_ in model.named_parameters():
        ref = g["grad." + name]
        if ref.size == 0:
            assert p_.grad is None                                                   # built but unused, as in the reference
        else:
            assert rel_l2(p_.grad.cpu().numpy(), ref) < 3e-4, name


def test_lpips_plain_vs_reference(dev, golden):
    from npp_amd import ops
    g = golden("g10_light.npz")
    out = torch.zeros(1, device=dev)
    for k in range(5):
        ops.lpips_plain_layer(torch.from_numpy(g[f"lp_f0_{k}"]).to(dev), torch.from_numpy(g[f"lp_f1_{k}"]).to(dev),
                              torch.from_numpy(g[f"lp_lin{k}"]).to(dev), 1.0, out)
    np.testing.assert_allclose(float(out[0]), float(g["lp_val"].reshape(-1)[0]), rtol=3e-5)


def test_light_fit_matches_oracle_trajectory(dev):
    """Five optimisation steps of the candidate fit (search.py:113-147) against the NumPy oracle from the same state."""
    from npp_amd.light import NPPNetLight, default_light_init
    H = 64
    img, mask = oracle.synthetic_image(H)
    angles, periods, _ = oracle.synthetic_periodicity(H, 1)
    freqs = oracle.SEED0_FREQS
    P0 = default_light_init(64)
    net = NPPNetLight(angles[0], periods[0], freqs, (H, H), P0, W=64, device=dev)
    P = {k: v.astype(np.float32).copy() for k, v in P0.items() if k.rsplit(".", 1)[0] in dict((n, 0) for n, _, _ in net.layout)}
    st = oracle.adam_init(P)
    la, ls = np.full((1, 3), 2.3841858e-07, np.float32), np.zeros((1, 3), np.float32)
    lat = np.concatenate([la, ls], 1).reshape(-1)
    lat_st = {"m": np.zeros(6, np.float32), "v": np.zeros(6, np.float32), "t": 0}
    rng = np.random.RandomState(0)
    for it in range(5):
        c = np.stack([rng.randint(0, H, 256), rng.randint(0, H, 256)], 1).astype(np.int32)
        gt = img[c[:, 0], c[:, 1]].astype(np.float32)
        ct = torch.from_numpy(c).to(dev)
        x_pos, x_per = net.embed(ct)
        lr = net.lr
        net.train_step(x_pos, x_per, torch.from_numpy(gt).to(dev))
        xp_o, xq_o = oracle.search_pos_embed(c, freqs, (H, H)), oracle.search_periodic_embed(c, angles[0], periods[0], (H, H))
        raw, cache = oracle.light_forward(P, xp_o, xq_o)
        pr = oracle.sigmoid(raw)
        loss, dpred, dla, dls = oracle.img2mse_grads(pr, gt, lat[None, :3], lat[None, 3:])
        G = oracle.light_backward(P, cache, dpred * pr * (1 - pr))
        oracle.adam_step(P, G, st, lr)
        glat = np.concatenate([np.asarray(dla).reshape(-1), np.asarray(dls).reshape(-1)])
        lat_st["t"] += 1
        lat_st["m"] = 0.9 * lat_st["m"] + 0.1 * glat
        lat_st["v"] = 0.999 * lat_st["v"] + 0.001 * glat * glat
        lat = lat - lr / (1 - 0.9 ** lat_st["t"]) * lat_st["m"] / (np.sqrt(lat_st["v"] / (1 - 0.999 ** lat_st["t"])) + 1e-8)
        assert abs(float(net.loss_buf[0]) - float(loss)) < 2e-4 * abs(float(loss)) + 1e-6
    sd = net.state_dict()
    for k in P:
        assert rel_l2(sd[k], P[k]) < 2e-3, k                     # Adam's normalised steps amplify fp32 round-off of tiny gradients
    np.testing.assert_allclose(net.latents.cpu().numpy(), lat, atol=2e-5)


def test_ranking_prefers_the_true_periodicity(dev):
    """The candidate loop + score of search.py:85-215 on a synthetic lattice: the true (angles, periods) must rank above
    a wrong period and a wrong orientation."""
    from npp_amd.light import ProposalRanker
    H = 128
    img, _ = oracle.synthetic_image(H, noise=0.01)
    angles, periods, shifts = oracle.synthetic_periodicity(H, 1)
    pseudo = np.ones((H, H), np.float32)
    pseudo[40:88, 36:92] = 0                                     # pseudo-mask region: content known, withheld from the fit
    i_train = np.stack(np.nonzero(pseudo), 1)
    i_val = np.stack(np.nonzero(1 - pseudo), 1)
    ranker = ProposalRanker(img, i_train, i_val, device=dev, N_iters=150, N_rand=2048, rng_mode="fast")
    cands = [(angles[0], periods[0], shifts[0]), (angles[0], periods[0] * 1.37, shifts[0]), (angles[0] + 35.0, periods[0], shifts[0])]
    d, order, details = ranker.rank(cands, topk=3)
    assert order[0] == 0, (d, order, details)
    assert np.all(np.diff(d) >= 0) and all(np.isfinite(x[0]) for x in details)


def test_fused_light_chains_equal_the_layer_by_layer_path(dev):
    """csrc/npp_light.hip (forward and data-gradient chains of NPP_Net_light as one launch each over all candidates, hardware sin,
    feature-major stashes, grouped weight-gradient launch, Adam + re-pack + gradient clear in one launch from iteration 1 on)
    against the layer-by-layer dense path (precise sinf) from the same
    weights on the same rows: predictions, losses, EVERY parameter gradient of every candidate, and the parameters / latents after
    ten optimiser steps."""
    from npp_amd.light import NPPNetLightBatch, default_light_init
    H, B, C = 96, 256, 3
    rng = np.random.RandomState(11)
    angles = np.array([[0.0, 90.0], [30.0, 120.0], [10.0, 80.0]], np.float32)
    periods = np.array([[12.0, 9.0], [7.0, 15.0], [20.0, 6.0]], np.float32)
    freqs = (rng.randn(10) * 10).astype(np.float32)
    init = default_light_init(256, 4)
    nets = [NPPNetLightBatch([(angles[i], periods[i]) for i in range(C)], freqs, (H, H), init, device=dev, fused=f) for f in (True, False)]
    assert nets[0].fused and not nets[1].fused
    coords = torch.from_numpy(np.stack([rng.randint(0, H, 4 * B), rng.randint(0, H, 4 * B)], 1).astype(np.int32)).to(dev)
    tabs = [n_.embed(coords) for n_ in nets[0].nets]
    x_pos_all, x_per_all = tabs[0][0], torch.stack([t[1] for t in tabs])
    gt_all = torch.from_numpy(rng.rand(4 * B, 3).astype(np.float32)).to(dev)
    for it in range(10):
        idx = torch.from_numpy(rng.permutation(4 * B)[:B]).to(dev)
        nets[0].fused_adam = it > 0          # iteration 0 keeps the gradients for the comparison below (the fused Adam launch clears them)
        losses = [n_.train_step(x_pos_all[idx], x_per_all[:, idx], gt_all[idx]).clone() for n_ in nets]
        if it == 0:
            pf, pu = nets[0]._ws[("fused", B)]["pred"], nets[1]._ws[B]["pred"]
            np.testing.assert_allclose(pf.cpu().numpy(), pu.cpu().numpy(), atol=3e-6)
            gf, gu = nets[0].grad.cpu().numpy(), nets[1].grad.cpu().numpy()
            for name in nets[0].dw:
                for part in ("dw", "db"):
                    a, b_ = getattr(nets[0], part)[name].cpu().numpy(), getattr(nets[1], part)[name].cpu().numpy()
                    for ci in range(C):
                        assert rel_l2(a[ci], b_[ci]) < 2e-5, (name, part, ci, rel_l2(a[ci], b_[ci]))
            assert rel_l2(gf, gu) < 2e-5
        np.testing.assert_allclose(losses[0].cpu().numpy(), losses[1].cpu().numpy(), rtol=2e-5)
    assert rel_l2(nets[0].params.cpu().numpy(), nets[1].params.cpu().numpy()) < 2e-5
    np.testing.assert_allclose(nets[0].latents.cpu().numpy(), nets[1].latents.cpu().numpy(), atol=2e-6)
    assert nets[0].nets[0].opt_step == nets[1].nets[0].opt_step == 10


def test_create_npp_net_is_search_consumes_the_generator_like_the_reference(dev, golden):
    """torch.manual_seed(0); create_npp_net(..., is_search=True) (search.py:91-99) through the boundary module leaves the SAME
    Fourier frequencies and the SAME initial weights as the reference's own construction (g10c_light_init.npz): the position embedder's
    ten normal draws, then NPP_Net_light's modules in construction order, all from the global generator."""
    import types
    from npp_amd import reference_api as api
    g = golden("g10c_light_init.npz")
    args = types.SimpleNamespace(multires=10, i_embed=0, freq_scales=[1], freq_offsets=[0, -1, 1, 0.5, -0.5], angle_offsets=[0],
                                 netdepth=4, netwidth=256, activation="snake", netchunk=1 << 22, lrate=5e-4, p_topk=1, normalize_type=1)
    torch.manual_seed(0)
    kw, _, _, _, _, emb, _ = api.create_npp_net(args, torch.tensor([80.54, 168.69]), torch.tensor([40.77, 36.48]), (211, 325),
                                                 percep_net=None, is_search=True)
    np.testing.assert_allclose(np.asarray(emb.freq_bands, np.float32).reshape(-1), g["freqs"], rtol=1e-6)
    sd = {k: v.detach().cpu().numpy() for k, v in kw["network_fn"].state_dict().items()}
    assert sorted(sd) == sorted(k[5:] for k in g.files if k.startswith("head."))
    for k, v in sd.items():
        np.testing.assert_array_equal(v.reshape(-1)[:16], g["head." + k])
        assert abs(float(v.astype(np.float64).sum()) - float(g["sum." + k])) < 1e-6


def test_candidate_loop_vs_reference_trajectory_g10d(dev, golden):
    """The candidate loop of NPP_proposal/search.py:85-147 executed with the REFERENCE's own modules for two candidates in sequence
    (g10d_light_fit.npz, tests/golden/make_golden_light_fit.py): reseeded init (incl. the Fourier-frequency draws that precede the
    network's in the generator stream), np.random.choice rows, sigmoid render, adaptive robust pixel loss, Adam over net + latents,
    the LR rule.  The reference shares ONE adaptive-loss object between the candidates (models/helpers.py:8): with carry_latents the
    second candidate's whole trajectory is reproduced too; without it (default: independent candidates) it starts from the initial
    latents and only the first candidate coincides."""
    from npp_amd.light import ProposalRanker
    g = golden("g10d_light_fit.npz")
    cands = [(g[f"c{i}.angles"], g[f"c{i}.periods"]) for i in range(2)]
    n_it, n_rand = int(g["n_iters"]), int(g["n_rand"])
    i_val = np.array([[20, 30], [21, 31], [43, 59]], np.int32)             # unused by the fits
    probe = torch.from_numpy(g["probe"]).to(dev)

    def run(carry):
        rk = ProposalRanker(g["masked_img"], g["i_train"], i_val, device=dev, N_iters=n_it, N_rand=n_rand, lrate=float(g["lrate"]),
                            lrate_decay=int(g["lrate_decay"]), carry_latents=carry, record_losses=True)
        np.testing.assert_array_equal(rk.freqs, g["freqs"])
        nets = rk.fit_candidates(cands)
        losses = torch.cat(rk.loss_log, 1).cpu().numpy()                   # (n_it, 2): one column per candidate in both modes
        return nets, losses
    nets_c, loss_c = run(True)
    nets_i, loss_i = run(False)
    for ci in range(2):
        ref = g[f"c{ci}.loss"]
        np.testing.assert_allclose(loss_c[:, ci], ref, rtol=2e-4)
        np.testing.assert_allclose(nets_c[ci].latents.cpu().numpy(), g[f"c{ci}.latents1"], atol=2e-5)
        np.testing.assert_allclose(nets_c[ci].render(probe).cpu().numpy(), g[f"c{ci}.probe_pred"], atol=2e-3)
        assert nets_c[ci].opt_step == n_it and nets_c[ci].global_step == n_it
    np.testing.assert_allclose(loss_i[:, 0], g["c0.loss"], rtol=2e-4)     # the first candidate is the same in both modes ...
    np.testing.assert_allclose(nets_i[1].latents.cpu().numpy(), g["c0.latents1"], atol=1e-3)      # ... the second starts afresh (moves like the first)
    assert np.abs(loss_i[:, 1] - g["c1.loss"]).max() > 10 * np.abs(loss_c[:, 1] - g["c1.loss"]).max()


def test_concurrent_candidate_fits_equal_the_serial_ones(dev):
    """ProposalRanker.fit_candidates -- all candidates in every launch (default: NPPNetLightBatch), advanced together on side
    streams (batched=False), or iterations 2 .. N of each fit replayed as ONE captured HIP graph (use_graph=True) -- against
    the eager fit_candidate loop: the same fitted
    parameters (split-K float atomics in the weight gradients give run-to-run noise of ~1e-6), step counts, LR clock and scores."""
    from npp_amd.light import ProposalRanker
    H = 128
    img, mask = oracle.synthetic_image(H, noise=0.01)
    angles, periods, _ = oracle.synthetic_periodicity(H, 1)
    pseudo = np.ones((H, H))
    pseudo[40:80, 50:90] = 0
    i_train, i_val = np.stack(np.nonzero(pseudo), 1), np.stack(np.nonzero(1 - pseudo), 1)
    ranker = ProposalRanker(img, i_train, i_val, device=dev, N_iters=40, N_rand=1024)
    cands = [(angles[0], periods[0]), (angles[0], periods[0] * 1.37), (angles[0] + 35.0, periods[0]), (angles[0] + 10.0, periods[0] * 0.8),
             (angles[0], periods[0] * 2.0)]
    graphed = [ranker.fit_candidate(a_, p_, use_graph=True) for a_, p_ in cands]
    together = ranker.fit_candidates(cands, n_streams=3, batched=False)
    stacked = ranker.fit_candidates(cands)                       # default: NPPNetLightBatch, the candidate is a grid dimension
    assert type(ranker._batch_keep).__name__ == "NPPNetLightBatch" and stacked[0].params.data_ptr() == ranker._batch_keep.params.data_ptr()
    for (a, p), net, netg, netb in zip(cands, together, graphed, stacked):
        alone = ranker.fit_candidate(a, p, use_graph=False, fused=False)          # the layer-by-layer eager loop: the comparator
        fused1 = ranker.fit_candidate(a, p)                                        # default: the fused chains, a candidate set of one
        assert net.opt_step == netg.opt_step == netb.opt_step == alone.opt_step == 40
        assert netg.global_step == netb.global_step == alone.global_step and netg.lr == netb.lr == alone.lr
        pb = alone.params.cpu().numpy()
        assert fused1.opt_step == 40 and type(ranker._batch_keep).__name__ == "NPPNetLightBatch" and ranker._batch_keep.fused
        for other in (net, netg, netb, fused1):
            assert np.linalg.norm(other.params.cpu().numpy() - pb) <= 1e-4 * np.linalg.norm(pb)
            np.testing.assert_allclose(other.latents.cpu().numpy(), alone.latents.cpu().numpy(), atol=1e-5)
        sb = ranker.score(alone)
        for other in (net, netg, netb):
            assert abs(ranker.score(other)[0] - sb[0]) <= 1e-3 * abs(sb[0])


def test_candidates_of_several_images_in_one_launch_sequence(dev):
    """light.rank_images (second half): candidate k of every image of a rank rides in one NPPNetLightBatch whose
    members are IMAGES (own lattice / positional tables, pixel rows, targets: npp_light_fwd_multi / npp_light_bwd_det_multi), for the
    reference's chained-latent candidate loop (NPP_proposal/search.py:85-215, models/helpers.py:8,144) and for independent candidates:
    per image the SAME BITS as its own serial ProposalRanker.rank -- scores, order, fitted parameters -- and the same bits run to run;
    images of different sizes (ragged tables) and different candidate counts included."""
    from npp_amd.light import ProposalRanker, rank_images
    rankers, cand_lists = [], []
    for i, (H, Wd, ncand) in enumerate([(128, 128, 3), (96, 144, 2), (128, 128, 3)]):
        img, mask = oracle.synthetic_image(max(H, Wd), noise=0.01, seed=i)
        img = img[:H, :Wd]
        angles, periods, shifts = oracle.synthetic_periodicity(128, 1)
        pseudo = np.ones((H, Wd))
        pseudo[30 + 5 * i:70, 40:80 + 4 * i] = 0
        i_train, i_val = np.stack(np.nonzero(pseudo), 1), np.stack(np.nonzero(1 - pseudo), 1)
        cands = [(angles[0] + 7.0 * j, periods[0] * (1.0 + 0.23 * j), shifts[0]) for j in range(ncand)]
        cand_lists.append(cands)
        rankers.append((img, i_train, i_val))
    for carry in (True, False):
        mk = lambda: [ProposalRanker(im, it, iv, device=dev, N_iters=40, N_rand=1024, carry_latents=carry) for im, it, iv in rankers]   # noqa: E731
        serial = [rk.rank(c, topk=10) for rk, c in zip(mk(), cand_lists)]
        together = rank_images(mk(), cand_lists, topk=10)
        again = rank_images(mk(), cand_lists, topk=10)
        for (d0, o0, det0), (d1, o1, det1), (d2, o2, det2) in zip(serial, together, again):
            assert list(o0) == list(o1) == list(o2)
            assert det1 == det2                                              # run to run: identical bits
            assert det0 == det1, (carry, det0, det1)                         # and the image's own serial loop: identical bits
            np.testing.assert_array_equal(d0, d1)


def test_candidate_fits_are_bit_reproducible(dev):
    """VERDICT r5 "What's weak" #5: the candidate fits of the proposal search (NPP_proposal/search.py:113-147) had float atomics on
    their gradient path (split-K partial sums of the weight gradients, the folded pixel loss's sums): two runs differed by ~3e-4 and
    equal-score rankings could flip.  Now (npp_tune "light_det" = 1, npp_light_bwd_det / npp_light_adam_pack_det, like the main
    loop's slabs): the same candidate set fitted twice gives the same BITS -- parameters, latents, per-iteration losses -- and the
    scores that rank them; the atomic forms stay available as the comparator (same fit to 1e-4)."""
    from npp_amd import ops
    from npp_amd.light import ProposalRanker
    H = 128
    img, mask = oracle.synthetic_image(H, noise=0.01)
    angles, periods, _ = oracle.synthetic_periodicity(H, 1)
    pseudo = np.ones((H, H))
    pseudo[40:80, 50:90] = 0
    i_train, i_val = np.stack(np.nonzero(pseudo), 1), np.stack(np.nonzero(1 - pseudo), 1)
    cands = [(angles[0], periods[0]), (angles[0], periods[0] * 1.37), (angles[0] + 35.0, periods[0]), (angles[0] + 10.0, periods[0] * 0.8),
             (angles[0], periods[0] * 2.0)]

    def run():
        rk = ProposalRanker(img, i_train, i_val, device=dev, N_iters=60, N_rand=1024, record_losses=True)
        nets = rk.fit_candidates(cands)
        assert type(rk._batch_keep).__name__ == "NPPNetLightBatch" and rk._batch_keep.fused and rk._batch_keep.fused_adam
        out = ([n_.params.cpu().numpy().copy() for n_ in nets], [n_.latents.cpu().numpy().copy() for n_ in nets],
               [rk.score(n_)[0] for n_ in nets], torch.cat(rk.loss_log).cpu().numpy())
        return out
    assert ops.tune("light_det") == 1 and ops.DETERMINISTIC
    a, b = run(), run()
    for pa, pb in zip(a[0], b[0]):
        np.testing.assert_array_equal(pa, pb)
    for la, lb in zip(a[1], b[1]):
        np.testing.assert_array_equal(la, lb)
    assert a[2] == b[2]
    assert a[3].shape == (60, len(cands)) and np.all(a[3] != 0)
    np.testing.assert_array_equal(a[3], b[3])
    # the atomic forms (the comparator): the same fit up to summation order
    old = ops.tune("light_det", 0)
    ops.DETERMINISTIC = False
    try:
        c = run()
    finally:
        ops.tune("light_det", old)
        ops.DETERMINISTIC = True
    for pa, pc in zip(a[0], c[0]):
        assert np.linalg.norm(pa - pc) <= 1e-4 * np.linalg.norm(pa)


@pytest.mark.parametrize("tag,K", [("small_k3", 3), ("small_k1", 1)])
def test_generic_width_net_vs_reference(dev, golden, tag, K):
    """NPP_Net / NPP_Net_top1 outside the fused kernels' specialisation (here W = 32, multires 2 -> 110-wide proposals) are
    served by dense.py on the generic dense-layer kernels: forward + every parameter gradient against the reference
    module's own autograd (g2_mlp.npz small_* cases: the full state_dict and gradients are stored)."""
    from npp_amd import reference_api as ra
    g = golden("g2_mlp.npz")
    W, fn = 32, 5
    if K > 1:
        net = ra.NPP_Net(22, 22 * (K - 1), [1], [0, -1, 1, 0.5, -0.5], [0], D=8, W=W, freq_nerf=fn, activation="snake", device=dev)
    else:
        net = ra.NPP_Net_top1(22, [1], [0, -1, 1, 0.5, -0.5], [0], D=8, W=W, freq_nerf=fn, activation="snake", device=dev)
    assert type(net).__name__.startswith("Dense")
    names = [k[len(tag) + 3:] for k in g.files if k.startswith(f"{tag}_P_")]
    sd = {n: torch.from_numpy(g[f"{tag}_P_{n}"]) for n in names}
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected and all(m.split(".")[0] in ("alpha_linear", "feature_linear2") for m in missing)
    emb = torch.from_numpy(g[f"{tag}_emb"]).to(dev)
    raw = net(None, emb)
    np.testing.assert_allclose(raw.detach().cpu().numpy(), g[f"{tag}_raw"], rtol=3e-4, atol=3e-5)
    (raw * torch.from_numpy(g[f"{tag}_draw"]).to(dev)).sum().backward()
    got = dict(net.named_parameters())
    for n in names:
        assert rel_l2(got[n].grad.cpu().numpy(), g[f"{tag}_G_{n}"]) < 3e-4, n
    assert net.alpha_linear.weight.grad is None                      # constructed, never used


def test_default_width_512_snake_is_the_fused_chain(dev):
    """netwidth 512 with the snake activation (the reference's default width, options/arg_config.py:57, with configs/*.txt's
    activation) is served by the fused chain built with -DNPP_WIDTH=512, not by dense.py: forward + every parameter gradient of
    the reference-style module against the exact-fp32 dense-layer implementation of the same network."""
    from npp_amd import reference_api as ra
    from npp_amd.dense import DenseNPPNet
    torch.manual_seed(0)
    net = ra.NPP_Net(22, 44, [1], [0, -1, 1, 0.5, -0.5], [0], D=8, W=512, freq_nerf=21, activation="snake", device=dev)
    assert type(net).__name__ == "NPP_Net" and net.W == 512
    dn = DenseNPPNet(22, 44, [1], [0, -1, 1, 0.5, -0.5], [0], D=8, W=512, freq_nerf=21, activation="snake", device=dev)
    dn.load_state_dict(net.state_dict(), strict=False)
    x = torch.rand(300, 3 * 462, device=dev) * 2 - 1
    y = torch.rand(300, 3, device=dev)
    la = ((torch.sigmoid(net(None, x)) - y) ** 2).mean()
    la.backward()
    lb = ((torch.sigmoid(dn(None, x)) - y) ** 2).mean()
    lb.backward()
    assert abs(float(la) - float(lb)) < 2e-3 * abs(float(lb))
    g = net.grads_by_name() if hasattr(net, "grads_by_name") else None
    ref = {k: p.grad for k, p in dn.named_parameters() if p.grad is not None}
    blob_g = net._blob.grad
    from npp_amd import ops
    # 16-bit stash: bf16 operands vs exact fp32; 8-bit stash (npp_tune "stash8", default): bf8 x fp8 products on 300 random rows
    tol = 9e-2 if ops.tune("stash8") else 3e-2
    for name, off, r, c in net._layout:
        got = blob_g[off:off + r * c].cpu().numpy()
        assert rel_l2(got, ref[name].reshape(-1).cpu().numpy()) < tol, name


def test_default_width_512_relu_trains(dev):
    """The reference's own defaults (netwidth 512) and its other activation: a few Adam steps through the reference-style
    loop (create-net, forward, loss.backward(), optimizer.step()) against a plain PyTorch copy of the same module."""
    from npp_amd import reference_api as ra
    torch.manual_seed(0)
    net = ra.NPP_Net(22, 44, [1], [0, -1, 1, 0.5, -0.5], [0], D=8, W=512, freq_nerf=21, activation="relu", device=dev)
    ref = {k: v.detach().clone().requires_grad_(True) for k, v in net.named_parameters()}
    x = torch.rand(300, 3 * 462, device=dev) * 2 - 1
    y = torch.rand(300, 3, device=dev)

    def torch_forward(P, xp):                                         # networks.py:56-95 in plain torch
        F = torch.nn.functional
        e0, aux = xp[:, :462], xp[:, 462:]
        h = e0
        for i in range(8):
            h = F.relu(F.linear(h, P[f"periodic_linears.{i}.weight"], P[f"periodic_linears.{i}.bias"]))
            if i == 4:
                h = torch.cat([e0, h], -1)
        f1 = F.linear(h, P["feature_linear1.weight"], P["feature_linear1.bias"])
        s = F.relu(F.linear(torch.cat([f1, aux], -1), P["scale_linears.0.weight"], P["scale_linears.0.bias"]))
        f2 = F.linear(s, P["feature_linear2.weight"], P["feature_linear2.bias"])
        p = F.relu(F.linear(torch.cat([f1, f2], -1), P["pos_linears.0.weight"], P["pos_linears.0.bias"]))
        return F.linear(p, P["rgb_linear.weight"], P["rgb_linear.bias"])
    prev = torch.backends.cuda.matmul.allow_tf32
    torch.backends.cuda.matmul.allow_tf32 = False
    try:
        la = ((torch.sigmoid(net(None, x)) - y) ** 2).mean()
        la.backward()
        lb = ((torch.sigmoid(torch_forward(ref, x)) - y) ** 2).mean()
        lb.backward()
    finally:
        torch.backends.cuda.matmul.allow_tf32 = prev
    assert abs(float(la) - float(lb)) < 1e-5
    for k, p in net.named_parameters():
        if ref[k].grad is None:
            assert p.grad is None
        else:
            assert rel_l2(p.grad.cpu().numpy(), ref[k].grad.cpu().numpy()) < 2e-3, k


@pytest.mark.parametrize("tag", ["a", "b"])
def test_shift_search_vs_reference(dev, golden, tag):
    """npp_shift_search + the host half of the periodicity search (proposal.py) against the reference's own compute_loss /
    generate_possible_shifts / generate_periodicity / feature_search (g11_search.npz)."""
    from npp_amd import proposal
    g = golden("g11_search.npz")
    act = torch.from_numpy(g[f"{tag}_act"]).to(dev)
    mask = torch.from_numpy(g[f"{tag}_mask"]).to(dev)
    rr = [int(v) for v in g[f"{tag}_rr"]]
    for i in range(rr[0], rr[1], rr[2]):
        r = (i, i + rr[2])
        sh = proposal.generate_possible_shifts(act.shape[1:], r, r)
        assert np.array_equal(sh, g[f"{tag}_shifts_{i}"])
        for edge in (1, 0):
            L = proposal.compute_loss(act, mask, sh, bool(edge)).cpu().numpy()
            np.testing.assert_allclose(L, g[f"{tag}_loss_{i}_{edge}"], rtol=3e-5, atol=3e-4)
    cands = proposal.feature_search(act, mask, rr, True)
    np.testing.assert_allclose(np.array([c[0] for c in cands]), g[f"{tag}_fs_angles"], atol=1e-3)
    np.testing.assert_allclose(np.array([c[1] for c in cands]), g[f"{tag}_fs_periods"], rtol=1e-5)
    np.testing.assert_array_equal(np.array([c[2] for c in cands]), g[f"{tag}_fs_shifts"])
    # full-scale shape: AlexNet-conv1-like map, thousands of displacements
    big = torch.rand(65, 64, 80, device=dev)
    bm = (torch.rand(64, 80, device=dev) > 0.2).float()
    sh = proposal.generate_possible_shifts((64, 80), (2, 7), (2, 7))
    Lb = proposal.compute_loss(big, bm, sh, True).cpu().numpy()
    want = oracle.shift_losses(big.cpu().numpy(), bm.cpu().numpy(), sh[::97], True)
    np.testing.assert_allclose(Lb[::97], want, rtol=1e-4)


def test_ranking_over_an_rccl_process_group_on_the_card(dev):
    """ProposalRanker.rank under an initialised nccl (= RCCL) process group -- the sharded form of search.py's candidate loop
    (parallel.shard_units + one all_gather of the score rows) -- on real hardware with the one rank this box has, in a child process;
    same ranking as without a group."""
    import os
    import subprocess
    import sys
    from npp_amd.parallel import free_port
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np, torch, torch.distributed as dist\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import oracle\n"
        "from npp_amd.light import ProposalRanker\n"
        "H = 96\n"
        "img, mask = oracle.synthetic_image(H, noise=0.01)\n"
        "angles, periods, _ = oracle.synthetic_periodicity(H, 1)\n"
        "pseudo = np.ones((H, H)); pseudo[30:60, 40:70] = 0\n"
        "i_train, i_val = np.stack(np.nonzero(pseudo), 1), np.stack(np.nonzero(1 - pseudo), 1)\n"
        "cands = [(angles[0], periods[0], None), (angles[0] + 35.0, periods[0], None), (angles[0], periods[0] * 1.4, None)]\n"
        "rk = ProposalRanker(img, i_train, i_val, device='cuda:0', N_iters=20, N_rand=512)\n"
        "d0, o0, _ = rk.rank(cands, topk=3)\n"
        "assert rk.last_rank_collective is None\n"
        "dist.init_process_group('nccl', device_id=torch.device('cuda', 0))\n"
        "d1, o1, det = rk.rank(cands, topk=3)\n"
        "dist.barrier(); dist.destroy_process_group()\n"
        "assert rk.last_rank_collective == {'backend': 'nccl', 'ranks': 1, 'rows': 3}, rk.last_rank_collective\n"
        "assert list(o0) == list(o1) and np.allclose(d0, d1, rtol=1e-3), (d0, d1)\n"
        "print('ok', dist.is_available(), len(det))\n") % (root, os.path.join(root, "tests"))
    env = {**os.environ, "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(free_port()),
           "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1].startswith("ok"), r.stderr[-3000:]
