"""Round 6: would the pixel rows' backward + weight-gradient launches hide under the patch-loss chain?  Stream A: the contextual
branch of one iteration (VGG19 forward on 12 patches of 96^2, contextual core, data gradients: ContextualLoss.fused).  Stream B: the
fused backward chain + weight gradients of an 8192-row batch (the pixel rows of config c2).  Wall time of A alone, B alone, A and B
started together on two streams (HIP events on each stream; the later end counts)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from npp_amd import synthetic as syn  # noqa: E402
from npp_amd.losses import ContextualLoss  # noqa: E402
from npp_amd.model import NPPNet  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
H, K = 512, 3
angles, periods, shifts = syn.synthetic_periodicity(H, K)
net = NPPNet(angles, periods, syn.SEED0_FREQS, (H, H), syn.init_params(K, seed=0), device=dev)
ROWS = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
coords = torch.stack([torch.randint(0, H, (ROWS,)), torch.randint(0, H, (ROWS,))], 1).to(torch.int32).to(dev)
net.forward_train(coords)
ws = net.workspace(ROWS)
ws["dpred"].normal_(0, 1e-3)
cx = ContextualLoss(use_vgg=True, device=dev)
xy = torch.rand(12, 3, 96, 96, device=dev)
loss_buf = torch.zeros(1, device=dev)
sA, sB = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def A():
    cx.fused(xy, 6, 1e-3, loss_buf)


def B():
    net.backward(ROWS)


def timed(fa, fb, reps=100):
    """reps launch sequences per stream, enqueued without waiting (the host stays ahead of the device); wall per repetition."""
    for _ in range(5):
        if fa:
            with torch.cuda.stream(sA):
                fa()
        if fb:
            with torch.cuda.stream(sB):
                fb()
    torch.cuda.synchronize()
    e0, ea, eb = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record(torch.cuda.current_stream())
    sA.wait_event(e0)
    sB.wait_event(e0)
    for _ in range(reps):
        if fa:
            with torch.cuda.stream(sA):
                fa()
        if fb:
            with torch.cuda.stream(sB):
                fb()
    ea.record(sA)
    eb.record(sB)
    torch.cuda.synchronize()
    return 1e3 * max(e0.elapsed_time(ea), e0.elapsed_time(eb)) / reps


ta, tb, tab = timed(A, None), timed(None, B), timed(A, B)
print(f"rows on stream B: {ROWS}")
print(f"A alone (patch-loss chain)                {ta:7.1f} us")
print(f"B alone (backward chain + weight grads)   {tb:7.1f} us")
print(f"A and B together, two streams             {tab:7.1f} us   (serial {ta + tb:.1f}; hidden {ta + tb - tab:.1f} us = {100 * (ta + tb - tab) / tb:.0f} % of B)")
