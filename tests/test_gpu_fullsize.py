"""BASELINE.json's FULL sizes (c2: 512 x 512, K = 3; c4: 1024 x 1024; c5: K = 5), where the NumPy oracle would take
minutes for the whole grid: size-independent properties of the path (every row is computed independently of its
tile neighbours => permutation / chunking invariance is bit-exact; sin^2 + cos^2 = 1 over the whole table; the raw
columns are the warped coordinates), the oracle on a row sample of the full grid, and the fused bf16 chain against the
library's second, independent implementation of the same network (exact-fp32 dense-layer kernels) on EVERY pixel."""
import numpy as np
import pytest

import oracle
from comparators import step_from_autograd  # noqa: E402  (tests/comparators.py)

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import npp_amd
    npp_amd.lib()
    return torch.device("cuda:0")


def _grid(H, W, dev):
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.int32), torch.arange(W, dtype=torch.int32), indexing="ij")
    return torch.stack([yy.reshape(-1), xx.reshape(-1)], 1).contiguous().to(dev)


def _net(dev, K, H, seed=0, width=256):
    from npp_amd.model import NPPNet
    angles, periods, _ = oracle.synthetic_periodicity(H, K)
    P = oracle.init_params(K, W=width, seed=seed)
    return NPPNet(angles, periods, oracle.SEED0_FREQS, (H, H), params=P, device=dev, ksplit=3, width=width), P, angles, periods


@pytest.mark.parametrize("K,H", [(3, 512), (5, 512), (3, 1024)])
def test_full_grid_render_is_row_independent_and_matches_oracle_sample(dev, K, H):
    net, P, angles, periods = _net(dev, K, H)
    grid = _grid(H, H, dev)
    n = grid.shape[0]
    full = net.render(grid)
    assert full.shape == (n, 3) and bool(torch.isfinite(full).all())
    # (1) chunking: three ragged chunks (none a multiple of the 64-row tile) == one launch, bit for bit
    cuts = [0, 100_003, n // 2 + 17, n]
    parts = torch.cat([net.render(grid[a:b].contiguous()) for a, b in zip(cuts[:-1], cuts[1:])], 0)
    assert torch.equal(parts, full)
    # (2) permutation: a row's result does not depend on which rows share its tile
    perm = torch.randperm(n, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    assert torch.equal(net.render(grid[perm].contiguous()), full[perm])
    # (3) the oracle on a sample of the full grid (corners included)
    rng = np.random.RandomState(K + H)
    idx = np.concatenate([[0, H - 1, n - H, n - 1], rng.randint(0, n, 1020)])
    c = grid[torch.from_numpy(idx).to(dev)].cpu().numpy()
    emb = oracle.embed(c, angles, periods, oracle.SEED0_FREQS, (H, H))
    raw_b, _ = oracle.mlp_forward(P, emb, K, emulate_bf16=True)
    raw_f, _ = oracle.mlp_forward(P, emb, K)
    got = full[torch.from_numpy(idx).to(dev)].cpu().numpy()
    assert np.abs(got - oracle.sigmoid(raw_b)).max() < 4e-3          # same operand rounding
    assert np.abs(got - oracle.sigmoid(raw_f)).max() < 2e-2          # plain fp32 maths


@pytest.mark.parametrize("K,W", [(3, 256), (1, 256), (3, 512), (1, 512)])
def test_c2_full_grid_fused_chain_vs_exact_fp32_dense_path(dev, K, W):
    """Two independent implementations of NPP_Net inside the library on all 262 144 pixels of the c2 grid: the fused bf16
    chain (coordinates in, embedding generated in registers) and dense.py (materialised fp32 embedding from the stand-alone
    embedder, one exact-fp32 MFMA GEMM per layer).  Budget: the 0.1 dB PSNR tolerance of BASELINE.json corresponds to
    ~1 % relative error of the residual; the two renders agree to 84 dB."""
    from npp_amd import ops, EmbedCfg
    from npp_amd.dense import DenseNPPNet, DenseNPPNetTop1
    H = 512
    net, P, angles, periods = _net(dev, K, H, width=W)       # W = 512: the reference's default --netwidth (libnpp_hip_w512.so)
    grid = _grid(H, H, dev)
    fused = net.render(grid)
    cfg = EmbedCfg.make(angles, periods, oracle.SEED0_FREQS, (H, H))
    emb = ops.embed_fwd(grid, cfg, torch.float32, precise=True)
    assert emb.shape == (H * H, K * 462)
    if K > 1:
        dn = DenseNPPNet(22, 22 * (K - 1), [1], [0, -1, 1, 0.5, -0.5], [0], D=8, W=W, freq_nerf=21, activation="snake", device=dev)
    else:
        dn = DenseNPPNetTop1(22, [1], [0, -1, 1, 0.5, -0.5], [0], D=8, W=W, freq_nerf=21, activation="snake", device=dev)
    missing, unexpected = dn.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()}, strict=False)
    assert not unexpected and all(m.split(".")[0] in ("alpha_linear", "feature_linear2") for m in missing), (missing, unexpected)
    with torch.no_grad():
        dense = torch.sigmoid(dn(None, emb))
    d = (fused - dense).abs()
    mse = float((d.double() ** 2).mean())
    psnr = -10.0 * np.log10(mse)
    print(f"K={K} W={W}: fused vs exact-fp32 dense path over {H * H} pixels: max |d| {float(d.max()):.2e}, PSNR {psnr:.1f} dB")
    assert float(d.max()) < 5e-3 and psnr > 65.0                     # measured: 3.8e-4 / 84 dB (K = 3), 1.6e-4 / 90 dB (K = 1)


def test_c4_full_grid_embedder_properties(dev):
    """Config c4 (1024 x 1024, K = 3, fp32): the whole 1 048 576 x 1386 table (5.8 GB)."""
    from npp_amd import ops, EmbedCfg
    H, K = 1024, 3
    angles, periods, _ = oracle.synthetic_periodicity(H, K)
    cfg = EmbedCfg.make(angles, periods, oracle.SEED0_FREQS, (H, H))
    grid = _grid(H, H, dev)
    n = grid.shape[0]
    for precise, tol, unit_tol in ((True, 5e-4, 2e-6), (False, 5e-4, 6e-5)):
        emb = ops.embed_fwd(grid, cfg, torch.float32, precise=precise)
        assert emb.shape == (n, K * 462)
        t = emb.view(n, K, 21, 22)                                   # [raw 22 | (sin, cos) x 10 frequencies] per proposal
        # raw block == the stand-alone warp (same kernel family, separate entry point)
        warp = ops.warp_fwd(grid, cfg).view(n, K, 22)
        assert float((t[:, :, 0, :] - warp).abs().max()) < 2e-6
        # normalised coordinates: column 0 = x / W * 2 - 1, column 11 = y / H * 2 - 1 (embedder.py:112-113)
        x = grid[:, 1].float() / H * 2 - 1
        y = grid[:, 0].float() / H * 2 - 1
        assert float((t[:, 0, 0, 0] - x).abs().max()) < 1e-6 and float((t[:, 0, 0, 11] - y).abs().max()) < 1e-6
        # sin^2 + cos^2 = 1 for every one of the 3 x 10 x 22 pairs of every pixel
        s, c = t[:, :, 1::2, :], t[:, :, 2::2, :]
        worst = 0.0
        for a in range(0, n, 1 << 18):                               # bounded temporaries
            worst = max(worst, float((s[a:a + (1 << 18)] ** 2 + c[a:a + (1 << 18)] ** 2 - 1).abs().max()))
        assert worst < unit_tol, worst
        # the warp's own pairs (columns 1..10 and 12..21 of the raw block are sin / cos of the lattice phase)
        for base in (1, 12):
            ps, pc = t[:, :, 0, base:base + 10:2], t[:, :, 0, base + 1:base + 10:2]
            assert float((ps ** 2 + pc ** 2 - 1).abs().max()) < 2e-5
        # the oracle on a sample of rows
        rng = np.random.RandomState(7)
        idx = np.concatenate([[0, H - 1, n - H, n - 1], rng.randint(0, n, 508)])
        ref = oracle.embed(grid[torch.from_numpy(idx).to(dev)].cpu().numpy(), angles, periods, oracle.SEED0_FREQS, (H, H))
        np.testing.assert_allclose(emb[torch.from_numpy(idx).to(dev)].cpu().numpy(), ref, atol=tol)
        del emb, t, warp, s, c
    # bf16 table (the HBM-light variant): row-permutation invariance, bit-exact
    perm = torch.randperm(n, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    b = ops.embed_fwd(grid, cfg, torch.bfloat16, precise=False)
    assert torch.equal(ops.embed_fwd(grid[perm].contiguous(), cfg, torch.bfloat16, precise=False), b[perm])


def test_c2_training_step_is_invariant_to_row_order(dev):
    """Linearity / order-independence of one full-size c2 training step (26 624 rows): permuting the batch rows changes
    neither the loss nor (beyond fp32 summation order) any parameter gradient."""
    K, H = 3, 512
    rng = np.random.RandomState(5)
    n = 26_624
    c = np.stack([rng.randint(0, H, n), rng.randint(0, H, n)], 1).astype(np.int32)
    gt = rng.rand(n, 3).astype(np.float32)
    perm = rng.permutation(n)
    out = []
    for order in (np.arange(n), perm):
        net, *_ = _net(dev, K, H)
        net.zero_grad()
        net.forward_train(torch.from_numpy(c[order]).to(dev))
        net.workspace(n)["dpred"].zero_()
        net.pixel_loss(n, n, torch.from_numpy(gt[order]).to(dev))
        net.backward(n)
        torch.cuda.synchronize()
        G = net.grads()
        out.append((float(net.loss_buf.item()), np.concatenate([np.asarray(G[k], np.float64).ravel() for k in sorted(G)])))
    (l0, g0), (l1, g1) = out
    assert abs(l0 - l1) <= 1e-5 * max(1.0, abs(l0))
    rel = float(np.linalg.norm(g0 - g1) / np.linalg.norm(g0))
    print(f"row-permuted full-size step: gradient rel-L2 difference {rel:.2e}")
    assert rel < 2e-4                                                # every rounding is per row; only the fp32 K-sum order moves


def test_c5_full_size_iteration_set(dev):
    """BASELINE.json configs[4] at FULL size: 512 x 512 image, top-5 proposals (NPP_Net with a 4-proposal scale layer, 2310
    inputs), contextual loss every iteration + LPIPS on 'same' iterations, P = 96, 8192 + 2 * 96^2 rows per iteration.
    The oracle would need minutes per iteration here, so the checks are the size-independent ones: for one iteration of
    EVERY patch source, the explicit kernel sequence (step_from) and the autograd restatement of train.py:200-251 over the same
    kernels (step_from_autograd) produce the same patch loss, the same dL/dpred on the patch rows and the same parameters from
    the same state; the pixel-loss part of dL/dpred is bit-identical; the MLP part is invariant to the order of the pixel rows
    (every row is computed independently); everything stays finite and the fit converges."""
    from npp_amd.fit import CompletionFit
    H, K = 512, 5
    img, mask = oracle.synthetic_image(H)
    angles, periods, shifts = oracle.synthetic_periodicity(H, K)

    def make():
        return CompletionFit(img, mask, angles, periods, oracle.SEED0_FREQS, oracle.init_params(K, seed=0), device=dev,
                             N_rand=8192, seed=3, shifts=shifts)

    def rel(a, b):
        return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
    src = make()
    assert src.patch_size == 96 and src.patch_num == 2 and src.net.K == 5
    by_source = {}
    for _ in range(80):
        b = src.sample_batch()
        if b is not None:
            by_source.setdefault(b["source"], b)
        if len(by_source) == 3:
            break
    assert set(by_source) == {"val", "train", "same"}
    for source, batch in by_source.items():
        assert batch["n"] == 8192 + 2 * 96 * 96
        a, b = make(), make()
        a.step_from(batch)
        step_from_autograd(b, batch)
        n_pix, n, bp = batch["n_pix"], batch["n"], batch["bp"]
        da, db = a.net.workspace(bp)["dpred"].cpu().numpy(), b.net.workspace(bp)["dpred"].cpu().numpy()
        assert np.isfinite(da).all() and np.abs(db[n_pix:n]).max() > 0
        np.testing.assert_array_equal(da[:n_pix], db[:n_pix])
        assert rel(da[n_pix:n], db[n_pix:n]) < 6e-3, source
        la, lb = float(a.last_patch_loss[0]), float(b.last_patch_loss[0])
        assert abs(la - lb) < 1e-5 * abs(lb) + 1e-9
        assert rel(a.net.params.cpu().numpy(), b.net.params.cpu().numpy()) < 2e-4
        if source == "same":
            for x_, y_ in zip(a.percepLoss.latents, b.percepLoss.latents):
                assert rel(x_.cpu().numpy(), y_.cpu().numpy()) < 1e-3
        # row-order invariance of the MLP half at this size: permute the PIXEL rows of the batch (coordinates + colours move
        # together through the gather), same loss, same gradients up to the summation order of the split-K partials
        c = make()
        perm = torch.randperm(n_pix, device=dev, generator=torch.Generator(device=dev).manual_seed(7))
        b2 = dict(batch)
        b2["coords"] = torch.cat([batch["coords"][:n_pix][perm], batch["coords"][n_pix:]], 0).contiguous()
        b2["gt"] = batch["gt"][perm].contiguous()
        assert batch.get("pmask") is None                      # completion: gt_mask = ones (train.py:176)
        c.step_from(b2)
        assert abs(float(c.net.loss_buf[0]) - float(a.net.loss_buf[0])) < 2e-6 * abs(float(a.net.loss_buf[0])) + 1e-9
        ga, gc = a.net.grads(), c.net.grads()
        for name in ga:
            # (not bit-equal: the contextual-loss kernels reduce with float atomics, so dL/dpred of the patch rows differs in
            # the last bits between any two runs; the MLP half alone is order-invariant to 4e-7, test above)
            assert rel(gc[name], ga[name]) < 5e-4, (source, name)
    fit = make()
    p0 = fit.psnr()
    for _ in range(40):
        fit.step_full()
    assert bool(torch.isfinite(fit.net.params).all()) and fit.psnr() > max(p0 + 8.0, 25.0)


@pytest.mark.parametrize("K,H,W", [(1, 256, 256), (3, 512, 256), (5, 512, 256), (3, 256, 512), (3, 1024, 256)])    # last: config c4 itself
def test_fused_fp32_render_matches_fp32_oracle(dev, K, H, W):
    """npp_mlp_fwd32 (BASELINE config c4's arithmetic: fused chain on v_mfma_f32_32x32x2_f32, f32 operands and accumulation)
    against the NumPy oracle in plain fp32 -- no bf16 emulation on either side -- on a row sample of the full grid, and
    against the bf16 chain on every pixel; chunking / permutation invariance is bit-exact."""
    net, P, angles, periods = _net(dev, K, H, width=W)
    grid = _grid(H, H, dev)
    n = grid.shape[0]
    full = net.render_fp32(grid)
    assert full.shape == (n, 3) and bool(torch.isfinite(full).all())
    rng = np.random.RandomState(K * 7 + H)
    idx = np.concatenate([[0, H - 1, n - H, n - 1], rng.randint(0, n, 1020)])
    c = grid[torch.from_numpy(idx).to(dev)].cpu().numpy()
    emb = oracle.embed(c, angles, periods, oracle.SEED0_FREQS, (H, H))
    raw, _ = oracle.mlp_forward(P, emb, K)
    got = full[torch.from_numpy(idx).to(dev)].cpu().numpy()
    err = np.abs(got - oracle.sigmoid(raw)).max()
    assert err < 5e-5, err                                            # fp32 round-off + hardware sin (1e-6 per call) through 10 layers
    assert np.abs(full.cpu().numpy() - net.render(grid).cpu().numpy()).max() < 2e-2      # the bf16 chain, every pixel
    perm = torch.randperm(n, device=dev, generator=torch.Generator(device=dev).manual_seed(2))
    assert torch.equal(net.render_fp32(grid[perm].contiguous()), full[perm])
    cuts = [0, n // 4 + 1, n // 2 + 33, n]
    assert torch.equal(torch.cat([net.render_fp32(grid[a:b].contiguous()) for a, b in zip(cuts[:-1], cuts[1:])], 0), full)


def test_c4_remapping_loop_at_1024sq(dev):
    """BASELINE.json configs[3]'s TASK at its full size (1024 x 1024 remapping image, K = 3: whole image = 1 048 576 training
    pixels, clear-region sampler mask, 0.3-weighted blurry pixels, contextual + Gram style loss, P = 160): five complete
    iterations stay finite and train every latent group, and for one batch the explicit kernel sequence equals the autograd
    restatement of NPP_remapping/train.py:198-262 over the same kernels (the reference trajectory of this loop is pinned at
    256^2 by g8r; the oracle would need minutes per iteration here)."""
    from npp_amd.fit import CompletionFit
    H, K = 1024, 3
    img, _ = oracle.synthetic_image(H, seed=7)
    angles, periods, shifts = oracle.synthetic_periodicity(H, K)
    clear = np.ones((H, H, 1), np.float32)
    clear[H // 3:H // 2] = 0.0

    def make():
        return CompletionFit(img, np.ones((H, H, 1), np.float32), angles, periods, oracle.SEED0_FREQS, oracle.init_params(K, seed=0),
                             device=dev, N_rand=8192, seed=0, shifts=shifts, task="remapping", clear_mask=clear, rng_mode="fast",
                             contextual_weight=0.01, style_weight=1.0, use_perceptual_loss=False)
    fit = make()
    assert fit.patch_size == 160 and fit.i_train.shape[0] == H * H and fit.i_val.shape[0] == int(clear.sum())
    lat0 = [l.clone() for l in fit.style.latents]
    p0 = fit.net.params.clone()
    done = 0
    while done < 5:
        done += bool(fit.step_full())
        assert bool(torch.isfinite(fit.last_patch_loss).all())
    assert bool(torch.isfinite(fit.net.params).all()) and float((fit.net.params - p0).abs().max()) > 1e-4
    assert all(bool(torch.isfinite(l).all()) and float((a - l).abs().max()) > 0 for a, l in zip(lat0, fit.style.latents))
    pred = fit.render_image()
    assert pred.shape == (H, H, 3) and bool(torch.isfinite(pred).all())
    # explicit launches == the pixel part of the autograd restatement on the pixel rows, bit for bit (same kernel, same inputs)
    a, b = make(), make()
    batch = None
    while batch is None:
        batch = a.sample_batch()
    a.step_from(batch)
    n_pix, bp = batch["n_pix"], batch["bp"]
    b.net.zero_grad()
    b.net.forward_train(batch["coords"])
    b.net.workspace(bp)["dpred"].zero_()
    b.net.pixel_loss(bp, n_pix, batch["gt"], mask=batch.get("pmask"), weight=1.0)
    assert batch.get("pmask") is not None                         # remapping: gt_mask = clear_mask (train.py:203)
    da, db = a.net.workspace(bp)["dpred"][:n_pix], b.net.workspace(bp)["dpred"][:n_pix]
    # (the gradient rows are written once each: bit-identical.  The loss word: fit `a` ran the iteration's form -- per-block partials
    #  summed in block order by the Adam launch, no atomics --, `b` the STAND-ALONE npp_pixel_loss call, which adds its block sums
    #  with atomicAdd and is not on the iteration's path: the two sums agree to round-off)
    assert torch.equal(da, db) and abs(float(a.net.loss_buf[0]) - float(b.net.loss_buf[0])) <= 1e-6 * abs(float(b.net.loss_buf[0]))


def test_c3_eight_images_on_one_gpu(dev):
    """BASELINE config c3's workload -- eight independent 512 x 512 completion images, top-3 proposals -- on the ONE GPU a box has:
    what a node with fewer than eight GPUs runs per GPU (8 / 4 / 2 images), as one stacked launch sequence (npp_amd.stack).  Every
    image has its own noise field, initial weights and sampler stream; after 60 complete iterations each one has converged like
    the single-image fit of config c2 (28 dB on the known pixels after 40 iterations there), their random streams stayed apart,
    and the first image equals its stand-alone fit (same split-K) to the tolerance of tests/test_gpu_stack.py."""
    from npp_amd.fit import CompletionFit
    from npp_amd.stack import StackedFit
    H, K, M = 512, 3, 8
    angles, periods, shifts = oracle.synthetic_periodicity(H, K)

    def fit(i, ks=None):
        img, mask = oracle.synthetic_image(H, seed=i)
        return CompletionFit(img, mask, angles, periods, oracle.SEED0_FREQS, oracle.init_params(K, seed=i), device=dev, N_rand=8192,
                             shifts=shifts, seed=100 + i, ksplit=ks)
    st = StackedFit([fit(i) for i in range(M)])
    assert st.Bp == 26624 and st.P == 96
    seqs = [[] for _ in range(M)]
    for _ in range(60):
        assert st.step_full() == M
        for i in range(M):
            seqs[i].append(st.last_sources[i])
    torch.cuda.synchronize()
    assert bool(torch.isfinite(st.params).all())
    assert len({tuple(s) for s in seqs}) == M                               # eight different sampler streams
    ps = st.psnr("known")
    print("PSNR on the known pixels after 60 stacked iterations:", [round(p, 2) for p in ps])
    assert min(ps) > 27.5, ps
    alone = fit(0, st.ksplit)
    for _ in range(60):
        alone.step_full()
    a, b = alone.net.params.cpu().numpy(), st.fits[0].net.params.cpu().numpy()
    e = float(np.linalg.norm(a - b) / np.linalg.norm(a))
    print(f"image 0 stacked vs alone after 60 iterations: rel-L2 {e:.2e}, PSNR {alone.psnr('known'):.2f} vs {ps[0]:.2f}")
    assert e < 3e-3 and abs(alone.psnr("known") - ps[0]) < 0.1
