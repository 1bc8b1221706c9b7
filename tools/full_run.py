"""Full-length completion fit (2001 iterations, reference schedule incl. the patch-size decay at i = 2000) through the driver."""
import os, sys, time, tempfile, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from npp_amd import io as nio, train
H, K = 512, 3
img, mask = oracle.synthetic_image(H)
a, p, s = oracle.synthetic_periodicity(H, K)
root = tempfile.mkdtemp()
d = nio.write_detected_dir(os.path.join(root, "detected", "syn512"), img, mask, np.ones_like(mask), a, p, s)
for mode in ("fast", "reference"):
    t0 = time.time()
    fit = train.main(["--datadir", d, "--basedir", os.path.join(root, "results_" + mode), "--p_topk", "3", "--N_iters", "2001",
                      "--i_testset", "1000", "--i_print", "1000", "--rng_mode", mode])
    print(f"{mode}: 2000 iterations in {time.time() - t0:.1f}s, PSNR known {fit.psnr('known'):.2f} unknown {fit.psnr('unknown'):.2f}, "
          f"skipped {fit.skipped}, patch_size now {fit.patch_size} x{fit.patch_num}, global_step {fit.net.global_step}")
    fit.close()
