"""Whole workflow at the reference's default iteration counts on 8 synthetic 512 x 512 images: where the wall time goes."""
import os, sys, subprocess, tempfile, time
import numpy as np
sys.path.insert(0, os.getcwd())
from npp_amd import io as nio, synthetic as syn
tmp = tempfile.mkdtemp()
S = 512
inp = os.path.join(tmp, "data", "completion", "input")
for i in range(8):
    im, mk = syn.synthetic_image(S, seed=10 + i)
    nio.write_detected_dir(os.path.join(inp, f"img{i}"), im, mk, np.ones_like(mk), [[0, 0]], [[1, 1]], [[[1, 0], [0, 1]]])
t0 = time.time()
r = subprocess.run([sys.executable, "-m", "npp_amd.run", "--task", "completion", "--input_path", inp, "--detected_path", os.path.join(tmp, "data", "completion", "detected"),
                    "--basedir", os.path.join(tmp, "res"), "--random-trunks", "--stack", "8", "--train-args", "--netwidth 256"], capture_output=True, text=True)
print("rc", r.returncode, "wall", time.time() - t0)
print("\n".join(l for l in r.stdout.splitlines() if "[stack]" in l or "[run" in l or "search" in l.lower()))
print(r.stderr[-2000:] if r.returncode else "")
