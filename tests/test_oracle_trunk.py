"""The trunk oracle (oracle/npp_patch_oracle.py: conv3x3 / ReLU / MaxPool2d stacks of VGG19[0:18] and
VGG16, forward and data gradient) against torch's own F.conv2d / F.max_pool2d + autograd on CPU -- the
third-party ops the reference's trunks are made of (contextual_loss/modules/vgg.py:16-36,
lpips/pretrained_networks.py:96-134).  Weights are arbitrary: the pretrained ones are not available
offline, the structure is what is pinned."""
import numpy as np
import pytest

import oracle

torch = pytest.importorskip("torch")
F = torch.nn.functional


def _weights(cfg, rng, scale=1.0):
    ws, cin = [], 3
    for v in cfg:
        if v == "M":
            continue
        ws.append(((rng.randn(v, cin, 3, 3) * scale * np.sqrt(2.0 / (9 * cin))).astype(np.float32),
                   (rng.randn(v) * 0.05).astype(np.float32)))
        cin = v
    return ws


def _torch_trunk(x, cfg, ws, taps):
    outs, idx, wi = [], 0, 0
    for v in cfg:
        if v == "M":
            x = F.max_pool2d(x, 2, 2)
            idx += 1
        else:
            w, b = ws[wi]
            wi += 1
            x = F.relu(F.conv2d(x, torch.from_numpy(w), torch.from_numpy(b), padding=1))
            idx += 2
            if idx - 1 in taps:
                outs.append(x)
    return outs


@pytest.mark.parametrize("name", ["vgg19_cx", "vgg16_lpips"])
def test_trunk_oracle_matches_torch(name):
    cfg, taps = ((oracle.VGG19_CX_CFG, oracle.VGG19_CX_TAPS) if name == "vgg19_cx"
                 else (oracle.VGG16_LPIPS_CFG, oracle.VGG16_LPIPS_TAPS))
    rng = np.random.RandomState(3)
    # narrow copy of the stack (channels / 8) so the CPU test stays fast; same layer sequence
    cfg = [v if v == "M" else max(4, v // 8) for v in cfg]
    ws = _weights(cfg, rng)
    x = rng.rand(2, 3, 32, 32).astype(np.float32)
    outs, cache = oracle.trunk_forward(x, cfg, ws, taps)
    xt = torch.from_numpy(x).requires_grad_(True)
    ref = _torch_trunk(xt, cfg, ws, taps)
    assert len(outs) == len(ref) == len(taps)
    gs = []
    for o, r in zip(outs, ref):
        assert o.shape == tuple(r.shape)
        np.testing.assert_allclose(o, r.detach().numpy(), rtol=2e-4, atol=2e-5)
        gs.append(rng.randn(*o.shape).astype(np.float32))
    loss = sum((r * torch.from_numpy(g)).sum() for r, g in zip(ref, gs))
    loss.backward()
    dx = oracle.trunk_backward(cfg, cache, taps, gs)
    ref_dx = xt.grad.numpy()
    assert np.linalg.norm(dx - ref_dx) / np.linalg.norm(ref_dx) < 1e-4


def test_maxpool_first_max_ties():
    """torch routes the gradient of a tied window to the first maximum in scan order; ReLU outputs
    (and bf16 storage) make ties common."""
    x = np.array([[[[1, 1, 0, 2], [1, 0, 2, 2], [0, 0, 3, 1], [0, 0, 1, 3]]]], np.float32)
    y, arg = oracle.maxpool2(x)
    dy = np.arange(1, 5, dtype=np.float32).reshape(1, 1, 2, 2)
    dx = oracle.maxpool2_bwd(dy, arg, x.shape)
    xt = torch.from_numpy(x).requires_grad_(True)
    yt = F.max_pool2d(xt, 2, 2)
    (yt * torch.from_numpy(dy)).sum().backward()
    np.testing.assert_array_equal(y, yt.detach().numpy())
    np.testing.assert_array_equal(dx, xt.grad.numpy())


def test_gemm_form_matches_direct_form():
    """conv3x3_gemm / conv3x3_dgrad_gemm (the im2col + SGEMM form bench.py's cpu_baseline times) == the direct form."""
    rng = np.random.RandomState(0)
    x = rng.randn(2, 5, 9, 7).astype(np.float32)
    w = rng.randn(6, 5, 3, 3).astype(np.float32)
    b = rng.randn(6).astype(np.float32)
    np.testing.assert_allclose(oracle.conv3x3_gemm(x, w, b), oracle.conv3x3(x, w, b), rtol=1e-4, atol=1e-5)
    dz = rng.randn(2, 6, 9, 7).astype(np.float32)
    np.testing.assert_allclose(oracle.conv3x3_dgrad_gemm(dz, w), oracle.conv3x3_dgrad(dz, w), rtol=1e-4, atol=1e-5)
