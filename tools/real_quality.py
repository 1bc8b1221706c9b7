"""Completion of one sample directory (search, then the default fit) with the result pictures copied to gpurun_out/vis:
python tools/real_quality.py <root> <name>   (random trunks: pretrained checkpoints are not in this image)."""
import os, sys, subprocess, shutil
root, name = sys.argv[1], sys.argv[2]
det = f"{root}/out/completion/detected"
shutil.rmtree(f"{root}/out", ignore_errors=True)
for mod, argv in (("npp_amd.search", ["--datadir", f"{root}/completion/input/{name}", "--outdir", det]),
                  ("npp_amd.train", ["--datadir", f"{det}/{name}", "--basedir", f"{root}/out/results", "--p_topk", "3"])):
    r = subprocess.run([sys.executable, "-m", mod] + argv + ["--random-trunks"], capture_output=True, text=True)
    print([l for l in r.stdout.splitlines() if l.startswith(("[EVAL]", "[search]"))][-1][:200], r.stderr[-500:] if r.returncode else "")
os.makedirs("gpurun_out/vis", exist_ok=True)
for f in ("pred_rgb_img.png", "pred_rgb_img_comp.png", "input_rgb_img.png"):
    shutil.copy(f"{root}/out/results/completion_top3/{name}/testset_002000/{f}", f"gpurun_out/vis/{name[:15]}_{f}")
shutil.copy(f"{det}/{name}/reg_img_0.png", f"gpurun_out/vis/{name[:15]}_reg_img_0.png")
