"""Round 6: cost of EXTRA backward + weight-gradient work (an 8192-row batch of a second network) inside the complete iteration --
(a) appended on the main stream, (b) on a side stream forked after the MLP forward and joined before the iteration's own backward
launch, i.e. beside the patch-loss chain.  (a) - base = the work's device time in the loop; (b) - base = what is left of it when it may
hide under the trunk / contextual launches (fork + join included).  Device-only loop over a pre-drawn batch pool, like bench.py."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from npp_amd import synthetic as syn  # noqa: E402
from npp_amd.fit import CompletionFit  # noqa: E402
from npp_amd.model import NPPNet  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
H, K = 512, 3
img, mask = syn.synthetic_image(H)
angles, periods, shifts = syn.synthetic_periodicity(H, K)
fit = CompletionFit(img, mask, angles, periods, syn.SEED0_FREQS, syn.init_params(K, seed=0), device=dev, N_rand=8192, shifts=shifts,
                    seed=0, rng_mode="fast")
pool = []
while len(pool) < 40:
    b = fit.sample_batch()
    if b is not None and b["source"] != "same":
        pool.append(b)
ROWS = 8192
net2 = NPPNet(angles, periods, syn.SEED0_FREQS, (H, H), syn.init_params(K, seed=1), device=dev)
coords = torch.stack([torch.randint(0, H, (ROWS,)), torch.randint(0, H, (ROWS,))], 1).to(torch.int32).to(dev)
net2.forward_train(coords)
net2.workspace(ROWS)["dpred"].normal_(0, 1e-3)
side = torch.cuda.Stream(dev)
mode = ["base"]
orig_fused = fit.contextualLoss.fused


def fused(*a, **k):
    main = torch.cuda.current_stream(dev)
    if mode[0] == "side":
        side.wait_stream(main)
        with torch.cuda.stream(side):
            net2.backward(ROWS)
    r = orig_fused(*a, **k)
    if mode[0] == "side":
        main.wait_stream(side)
    elif mode[0] == "main":
        net2.backward(ROWS)
    return r


fit.contextualLoss.fused = fused


def run(m, iters=400):
    mode[0] = m
    for i in range(40):
        fit.step_from(pool[i % len(pool)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(iters):
        fit.step_from(pool[i % len(pool)])
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / iters


res = {m: [] for m in ("base", "main", "side")}
for rep in range(3):
    for m in res:
        res[m].append(run(m))
for m, v in res.items():
    print(f"{m:5s}: " + "  ".join(f"{x:7.1f}" for x in v) + f"   us per iteration (median {sorted(v)[1]:.1f})")
b, a_, s_ = (sorted(res[m])[1] for m in ("base", "main", "side"))
print(f"extra work on the main stream +{a_ - b:.1f} us; beside the patch-loss chain +{s_ - b:.1f} us -> {a_ - s_:.1f} us of {a_ - b:.1f} hidden")
