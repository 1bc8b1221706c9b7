import os, sys, time, json, tempfile
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from npp_amd import io as nio, synthetic as syn, run
S, M = 512, 8
tmp = tempfile.mkdtemp()
srcs = []
for i in range(M):
    im, mk = syn.synthetic_image(S, seed=10 + i)
    srcs.append(nio.write_detected_dir(os.path.join(tmp, "input", f"img{i}"), im, mk, np.ones_like(mk), [[0, 0]], [[1, 1]], [[[1, 0], [0, 1]]]))
flags = ["--device", "cuda:0", "--random-trunks"]
res = {}
for thr in (1, 8, 8, 1):
    det = os.path.join(tmp, f"det_{thr}_{len(res)}")
    t0 = time.perf_counter()
    errs = run.search_all(srcs, det, flags, threads=thr)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert all(e is None for e in errs), errs
    out = []
    for i in range(M):
        with open(os.path.join(det, f"img{i}", "config.odgt")) as f:
            out.append(json.loads(f.readline()))
    res[(thr, len(res))] = out
    print(f"threads {thr}: {dt:.2f} s for {M} images", flush=True)
keys = list(res)
ref = res[keys[0]]
for k in keys[1:]:
    worst = 0.0
    same_rank = True
    for a, b in zip(ref, res[k]):
        same_rank &= (a["selected_angles"] == b["selected_angles"]) and (a["selected_periods"] == b["selected_periods"])
        worst = max(worst, max(abs(x - y) / abs(x) for x, y in zip(a["distances"], b["distances"])))
    print(k, "same ranking (angles, periods of all candidates in order):", same_rank, " max rel. difference of the distances:", f"{worst:.2e}")
