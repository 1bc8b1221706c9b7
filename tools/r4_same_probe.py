"""Round-4 probe: iterations of ONE patch source (default 'same': the LPIPS branch runs beside the contextual chain) for a kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d out -- python3 tools/r4_same_probe.py [same|val|train] [iters]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from npp_amd import synthetic as syn                 # noqa: E402
from npp_amd.fit import CompletionFit                # noqa: E402

src = sys.argv[1] if len(sys.argv) > 1 else "same"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
H, K = 512, 3
angles, periods, shifts = syn.synthetic_periodicity(H, K)
img, mask = syn.synthetic_image(H, seed=0)
f = CompletionFit(img, mask, angles, periods, syn.SEED0_FREQS, syn.init_params(K, seed=0), device=dev, N_rand=8192, shifts=shifts, seed=0,
                  rng_mode="fast", use_perceptual_loss=os.environ.get("R4_NO_LPIPS", "0") == "0",
                  use_contextual_loss=os.environ.get("R4_NO_CX", "0") == "0")
pool = []
while len(pool) < 8:
    b = f.sample_batch()
    if b is not None and b["source"] == src:
        pool.append(b)
for i in range(10):
    f.step_from(pool[i % len(pool)])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(iters):
    f.step_from(pool[i % len(pool)])
torch.cuda.synchronize()
per = (time.perf_counter() - t0) / iters * 1e3
# host enqueue time alone: iterations issued after a drain, timed up to the return of step_from (nothing waits on the device)
enq = []
for i in range(20):
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    f.step_from(pool[i % len(pool)])
    enq.append((time.perf_counter() - t1) * 1e3)
torch.cuda.synchronize()
enq.sort()
print(f"{src}: {per:.4f} ms per iteration; host enqueue alone {enq[len(enq) // 2]:.4f} ms (median of 20)", flush=True)
