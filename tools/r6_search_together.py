"""Round 6: the periodicity search of 8 images of 512^2 at the default iteration counts -- one after the other, on 8
host threads (round 5), and candidate k of every image in one launch sequence (search.main_multi / light.rank_images): wall time and
whether every form writes the same config.odgt (run from the repo root on the GPU box)."""
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from npp_amd import io as nio, synthetic as syn, run  # noqa: E402

S, M = 512, 8
tmp = tempfile.mkdtemp()
srcs = []
for i in range(M):
    im, mk = syn.synthetic_image(S, seed=10 + i)
    srcs.append(nio.write_detected_dir(os.path.join(tmp, "input", f"img{i}"), im, mk, np.ones_like(mk), [[0, 0]], [[1, 1]], [[[1, 0], [0, 1]]]))
flags = ["--device", "cuda:0", "--random-trunks"] + sys.argv[1:]          # e.g. --precision bf16
res = {}
for tag, kw in (("together", {}), ("serial", dict(threads=1, together=False)), ("threads8", dict(threads=8, together=False)), ("together", {}), ("threads8", dict(threads=8, together=False))):
    det = os.path.join(tmp, f"det_{tag}_{len(res)}")
    t0 = time.perf_counter()
    errs = run.search_all(srcs, det, flags, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert all(e is None for e in errs), errs
    out = [json.loads(open(os.path.join(det, f"img{i}", "config.odgt")).readline()) for i in range(M)]
    res[(tag, len(res))] = out
    print(f"{tag:9s}: {dt:.2f} s for {M} images of {S}^2 ({len(out[0]['distances'])} ranked of the candidates found)", flush=True)
keys = list(res)
ref = res[keys[1]]
for k in keys:
    same = all(a["selected_angles"] == b["selected_angles"] and a["selected_periods"] == b["selected_periods"] and a["distances"] == b["distances"]
               for a, b in zip(ref, res[k]))
    print(k, "identical to the serial loop's config.odgt (candidates, order, distances bit for bit):", same)
