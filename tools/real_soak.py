"""The reference's run_completion.sh / run_remapping.sh / run_segmentation.sh workflow (periodicity search, then the fit) on sample
directories in the reference's input format:  python tools/real_soak.py <root>  with <root>/{completion,remapping,segmentation}/input/<name>/
{gt_img,masked_img,unknown_mask,valid_mask}.png.  Pretrained trunks are not in this image: --random-trunks (the ranking / patch losses run
their real code on fixed-seed random filters).  Prints wall-clock per stage; outputs under <root>/out."""
import os, sys, subprocess, time, glob
root = os.path.abspath(sys.argv[1])
out = os.path.join(root, "out")
extra = sys.argv[2:]


def run(what, argv):
    t0 = time.time()
    r = subprocess.run([sys.executable, "-m"] + argv, capture_output=True, text=True)
    dt = time.time() - t0
    tail = [l for l in r.stdout.strip().splitlines() if l.startswith(("[EVAL]", "[search]", "[TRAIN]"))][-2:]
    print(f"{what}: rc {r.returncode}, {dt:.1f} s wall", tail, r.stderr[-2000:] if r.returncode else "", flush=True)
    return r.returncode


rc = 0
for task in ("completion", "remapping", "segmentation"):
    for src in sorted(glob.glob(os.path.join(root, task, "input", "*"))):
        name = os.path.basename(src)
        det = os.path.join(out, task, "detected")
        rc |= run(f"{task}/{name} search", ["npp_amd.search", "--datadir", src, "--outdir", det, "--random-trunks"] + extra)
        if task == "segmentation":                      # the imsegm initial segmentation is an input of this build: a stand-in
            import numpy as np
            sys.path.insert(0, os.getcwd())
            from npp_amd import io as nio
            g = nio._imread_gray(os.path.join(src, "valid_mask.png"))[..., 0]
            yy, xx = np.mgrid[:g.shape[0], :g.shape[1]]
            blob = ((yy - g.shape[0] // 2) ** 2 + (xx - g.shape[1] // 2) ** 2 < (min(g.shape) // 6) ** 2).astype(np.float64)
            nio.imsave(os.path.join(det, name, "non_period_mask.png"), np.repeat(blob[..., None], 3, 2))
            nio.imsave(os.path.join(det, name, "period_mask.png"), np.repeat(1.0 - blob[..., None], 3, 2))
        rc |= run(f"{task}/{name} train", ["npp_amd.train", "--datadir", os.path.join(det, name), "--basedir", os.path.join(out, "results"), "--p_topk", "3",
                                            "--random-trunks"] + (["--task", task] if task != "completion" else []) + extra)
sys.exit(rc)
