"""End-to-end soak of the command lines on a GPU box, on a non-square synthetic image of the size of the reference's samples
(300 x 340): train (completion at both widths, remapping, segmentation) and the periodicity search."""
import os, sys, subprocess, tempfile
import numpy as np
sys.path.insert(0, os.getcwd())
from npp_amd import io as nio, synthetic as syn

tmp = tempfile.mkdtemp()
S, K, H, W = 384, 3, 300, 340
img, mask = syn.synthetic_image(S)
img, mask = img[:H, :W], mask[:H, :W]
a, p, sh = syn.synthetic_periodicity(S, K)
d = nio.write_detected_dir(os.path.join(tmp, "detected", "lat"), img, mask, np.ones_like(mask), a, p, sh)
yy, xx = np.mgrid[:H, :W]
blob = ((yy - 150) ** 2 + (xx - 200) ** 2 < 40 ** 2).astype(np.float64)
nio.imsave(os.path.join(d, "non_period_mask.png"), np.repeat(blob[..., None], 3, 2))
nio.imsave(os.path.join(d, "period_mask.png"), np.repeat(1.0 - blob[..., None], 3, 2))


def run(what, argv):
    r = subprocess.run([sys.executable, "-m"] + argv, capture_output=True, text=True)
    print(what, "rc", r.returncode, (r.stdout.strip().splitlines() or [""])[-2:], r.stderr[-1500:] if r.returncode else "", flush=True)
    assert r.returncode == 0


common = ["npp_amd.train", "--datadir", d, "--basedir", os.path.join(tmp, "res"), "--p_topk", "3", "--N_iters", "201", "--i_testset", "200",
          "--i_print", "100", "--random-trunks"]
for Wn in (256, 512):
    run(f"completion W={Wn}", common + ["--expname", f"w{Wn}", "--netwidth", str(Wn)])
run("remapping", common + ["--task", "remapping"])
run("segmentation", common + ["--task", "segmentation"])
out = [f for _, _, fs in os.walk(os.path.join(tmp, "res")) for f in fs]
print(len(out), "files written")
src = os.path.join(tmp, "input", "lat")
nio.write_detected_dir(src, img, mask, np.ones_like(mask), [[0, 0]], [[1, 1]], [[[1, 0], [0, 1]]])
run("search", ["npp_amd.search", "--datadir", src, "--outdir", os.path.join(tmp, "det2"), "--N_iters", "60", "--search_range", "2", "9", "3",
               "--topk_detection", "3", "--random-trunks"])
run("search alexnet", ["npp_amd.search", "--datadir", src, "--outdir", os.path.join(tmp, "det3"), "--N_iters", "60", "--search_range", "2", "9", "3",
                       "--topk_detection", "3", "--random-trunks", "--gray_only"])
# the whole shell workflow, three images of one rank fitted together (round 5: python -m npp_amd.run --stack)
inp = os.path.join(tmp, "data", "completion", "input")
for i in range(3):
    im_i, mk_i = syn.synthetic_image(S, seed=10 + i)
    nio.write_detected_dir(os.path.join(inp, f"img{i}"), im_i[:H, :W], mk_i[:H, :W], np.ones_like(mk_i[:H, :W]), [[0, 0]], [[1, 1]], [[[1, 0], [0, 1]]])
run("run --stack 8", ["npp_amd.run", "--task", "completion", "--input_path", inp, "--detected_path", os.path.join(tmp, "data", "completion", "detected"),
                      "--basedir", os.path.join(tmp, "res_run"), "--random-trunks", "--stack", "8",
                      "--search-args", "--N_iters 60 --search_range 2 9 3 --topk_detection 3",
                      "--train-args", "--N_iters 121 --i_testset 120 --i_print 60 --netwidth 256"])
out = [f for _, _, fs in os.walk(os.path.join(tmp, "res_run")) for f in fs]
print(len(out), "files written by run --stack")
assert len(out) == 18
# the remapping task through the stacked driver (style term + per-pixel loss weights per image): two images of one patch size
det = os.path.join(tmp, "data", "completion", "detected")
dirs = sorted(os.path.join(det, n) for n in os.listdir(det))[:2]
code = ("import sys; sys.path.insert(0, %r); from npp_amd import train; "
        "fits = train.main_stacked([['--task', 'remapping', '--datadir', d, '--basedir', %r, '--p_topk', '3', '--N_iters', '81', '--i_testset', '80', "
        "'--i_print', '40', '--random-trunks'] for d in %r]); "
        "assert train.main_stacked.last_error is None and all(f is not None and f.has_style for f in fits); print('stacked remapping ok')"
        % (os.getcwd(), os.path.join(tmp, "res_remap"), dirs))
r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
print("remapping, stacked", "rc", r.returncode, (r.stdout.strip().splitlines() or [""])[-3:], r.stderr[-1500:] if r.returncode else "", flush=True)
assert r.returncode == 0 and "[stack] 2 images per launch sequence" in r.stdout
