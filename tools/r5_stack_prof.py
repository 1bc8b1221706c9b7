"""Round 5: the stacked loop (npp_amd.stack.StackedFit, M images per launch sequence) as a profiling target: pre-drawn batch sets,
device-only iterations -- run it under `rocprofv3 --kernel-trace --stats` (tools/r5_stack_prof.sh) for per-kernel durations, or
with --pmc for FETCH_SIZE / WRITE_SIZE / SQ counters.  usage: r5_stack_prof.py [M] [iterations] [batch_lpips 0/1]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from npp_amd import synthetic as syn                 # noqa: E402
from npp_amd.fit import CompletionFit                # noqa: E402
from npp_amd.stack import StackedFit                 # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
H, K = 512, 3
angles, periods, shifts = syn.synthetic_periodicity(H, K)
if os.environ.get("R5_PERIOD_SCALE"):                 # e.g. 1.33: patch size 128 instead of 96 (loaders.py:133-134)
    sc_ = float(os.environ["R5_PERIOD_SCALE"])
    periods = [[float(v) * sc_ for v in p_] for p_ in periods]
    shifts = [[[float(v) * sc_ for v in s_] for s_ in sh_] for sh_ in shifts]
fits = []
for i in range(M):
    img, mask = syn.synthetic_image(H, seed=i)
    fits.append(CompletionFit(img, mask, angles, periods, syn.SEED0_FREQS, syn.init_params(K, seed=i), device=dev, N_rand=8192,
                              shifts=shifts, seed=i, rng_mode=os.environ.get("R5_RNG", "fast"), use_perceptual_loss=os.environ.get("R5_NO_LPIPS", "0") != "1"))   # R5_NO_LPIPS=1: the floor without the branch
st = StackedFit(fits, ksplit=int(os.environ["R5_KSPLIT"]) if os.environ.get("R5_KSPLIT") else None)
if len(sys.argv) > 3:
    st.batch_lpips = bool(int(sys.argv[3]))       # 0: one LPIPS branch per 'same' image (the comparator)
for _ in range(10):
    st.step_full()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(ITERS):
    st.step_full()                                    # (the next draw rides on the sampler stream under the iteration)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / ITERS
rows = fits[0].N_rand + fits[0].patch_num * fits[0].patch_size ** 2
print(f"stacked M = {M}: {dt * 1e3:.4f} ms per stacked iteration (incl. sampling), {M * rows / dt / 1e6:.2f} M rows/s, ksplit {st.ksplit}, LPIPS of several 'same' images in one pass: {st.batch_lpips}")
st.close()
