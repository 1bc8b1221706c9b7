"""End-to-end soak of the command lines on a GPU box: train at both widths and the periodicity search on a synthetic image."""
import os, sys, subprocess, tempfile, numpy as np
sys.path.insert(0, os.getcwd())
from npp_amd import io as nio, synthetic as syn
tmp = tempfile.mkdtemp()
H, K = 256, 3
img, mask = syn.synthetic_image(H)
a, p, sh = syn.synthetic_periodicity(H, K)
d = nio.write_detected_dir(os.path.join(tmp, "detected", "lat"), img, mask, np.ones_like(mask), a, p, sh)
for W in (256, 512):
    r = subprocess.run([sys.executable, "-m", "npp_amd.train", "--datadir", d, "--basedir", os.path.join(tmp, "res"), "--expname", f"w{W}", "--p_topk", "3",
                        "--N_iters", "201", "--i_testset", "200", "--i_print", "100", "--netwidth", str(W), "--random-trunks"], capture_output=True, text=True)
    print("train W", W, "rc", r.returncode, r.stdout.strip().splitlines()[-2:] if r.stdout else r.stderr[-300:])
    assert r.returncode == 0
out = [f for _, _, fs in os.walk(os.path.join(tmp, "res")) for f in fs]
print(len(out), "files written;", sorted(set(out))[:6])
src = os.path.join(tmp, "input", "lat")
nio.write_detected_dir(src, img, mask, np.ones_like(mask), [[0, 0]], [[1, 1]], [[[1, 0], [0, 1]]])
r = subprocess.run([sys.executable, "-m", "npp_amd.search", "--datadir", src, "--outdir", os.path.join(tmp, "det2"), "--N_iters", "60", "--search_range", "2", "9", "3",
                    "--topk_detection", "3", "--random-trunks"], capture_output=True, text=True)
print("search rc", r.returncode, (r.stdout.strip().splitlines() or [""])[-1], r.stderr[-1500:] if r.returncode else "")
assert r.returncode == 0
