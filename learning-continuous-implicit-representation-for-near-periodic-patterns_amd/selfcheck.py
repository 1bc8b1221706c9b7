"""__graft_entry__.smoke(): one small invocation of the whole hot path on cuda:0
(fused forward with stash -> adaptive pixel loss -> backward chain -> grouped wgrad ->
Adam), checked against the oracle.  The oracle is imported here only as the checker."""
import numpy as np
import torch


def smoke():
    import oracle
    from . import ops
    from .model import NPPNet
    dev = torch.device("cuda:0")
    ops.selftest(dev)
    K, H, n = 3, 64, 128
    angles, periods, _ = oracle.synthetic_periodicity(256, K)
    P = oracle.init_params(K, seed=0)
    net = NPPNet(angles, periods, oracle.SEED0_FREQS, (H, H), params=P, device=dev, ksplit=2)
    rng = np.random.RandomState(0)
    c = np.stack([rng.randint(0, H, n), rng.randint(0, H, n)], 1).astype(np.int32)
    gt = rng.rand(n, 3).astype(np.float32)
    net.zero_grad()
    pred = net.forward_train(torch.from_numpy(c).to(dev))
    net.workspace(n)["dpred"].zero_()
    net.pixel_loss(n, n, torch.from_numpy(gt).to(dev))
    net.backward(n)
    G = net.grads()
    net.optimizer_step(n)
    torch.cuda.synchronize()
    emb = oracle.embed(c, angles, periods, oracle.SEED0_FREQS, (H, H))
    raw, cache = oracle.mlp_forward(P, emb, K, emulate_bf16=True)
    pr = oracle.sigmoid(raw)
    err = float(np.abs(pred.cpu().numpy() - pr).max())
    assert err < 5e-3, f"fused forward differs from the oracle: {err}"
    la = net.latents[:3].cpu().numpy()[None] * 0 + 2.3841858e-07
    loss, dpred, _, _ = oracle.img2mse_grads(pr, gt, la, np.zeros((1, 3), np.float32))
    Gref = oracle.mlp_backward(P, cache, dpred * pr * (1 - pr), emulate_bf16=True)
    worst = max(np.linalg.norm(G[k] - Gref[k]) / max(np.linalg.norm(Gref[k]), 1e-30) for k in Gref)
    assert worst < 5e-2, f"gradients differ from the oracle: rel L2 {worst}"
    print(f"smoke ok: |pred-oracle|max={err:.2e}, worst grad rel-L2={worst:.2e}, loss={float(loss):.5f}")
