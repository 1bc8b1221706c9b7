import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from npp_amd import io as nio, synthetic as syn, train
tmp = tempfile.mkdtemp()
S, K = 512, 3
a, p, sh = syn.synthetic_periodicity(S, K)
dirs = []
for i in range(8):
    im, mk = syn.synthetic_image(S, seed=10 + i)
    dirs.append(nio.write_detected_dir(os.path.join(tmp, "det", f"img{i}"), im, mk, np.ones_like(mk), a, p, sh))
T = {"prepare": 0.0, "after": 0.0, "finish": 0.0}
def wrap(name, key):
    f = getattr(train, name)
    def g(*a_, **k_):
        t0 = time.perf_counter()
        try:
            return f(*a_, **k_)
        finally:
            T[key] += time.perf_counter() - t0
    setattr(train, name, g)
wrap("_prepare", "prepare"); wrap("_after_iteration", "after"); wrap("_finish", "finish")
t0 = time.perf_counter()
fits = train.main_stacked([["--datadir", d, "--basedir", os.path.join(tmp, "res"), "--p_topk", "3", "--random-trunks", "--netwidth", "256"] for d in dirs])
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print("total", tot, T, "loop", tot - sum(T.values()))
