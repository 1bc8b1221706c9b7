"""Stage times of the periodicity search (npp_amd.search) on one sample directory:  python tools/search_prof.py <dir>"""
import os, sys, time, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from npp_amd import io as nio, search, proposal
from npp_amd.light import ProposalRanker
src = sys.argv[1]
args = search.parse(["--datadir", src, "--random-trunks"])
torch.cuda.set_device(0)
masked_img, img, mask, valid = [f(os.path.join(src, n)) for f, n in ((nio._imread_rgb, "masked_img.png"), (nio._imread_rgb, "gt_img.png"), (nio._imread_gray, "unknown_mask.png"), (nio._imread_gray, "valid_mask.png"))]
m2, v2 = mask[..., 0], valid[..., 0]
warnings.simplefilter("ignore")
for rep in range(3):
    sync = torch.cuda.synchronize
    t0 = time.time()
    ang, per, sh = proposal.search_periodicity_by_feat(np.uint8(masked_img * 255), np.uint8(v2 * m2), repeat_range=tuple(args.search_range), edge_searching=True, gray_only=True, device="cuda:0")
    sync(); t1 = time.time()
    _, i_train, i_val = search.pseudo_mask_split(m2, v2); t2 = time.time()
    ranker = ProposalRanker(masked_img.astype(np.float32), i_train, i_val, device="cuda:0"); sync(); t3 = time.time()
    t4 = time.time()                                    # (the pixel draws now stream under the fits: no separate stage)
    cands = [(ang[i], per[i]) for i in range(min(9, len(ang)))]
    nets = ranker.fit_candidates(cands); sync(); t5 = time.time()
    sc = [ranker.score(n) for n in nets]; sync(); t6 = time.time()
    print(f"rep {rep}: {masked_img.shape[:2]} frontend+displacement {t1 - t0:.3f} s, pseudo mask {t2 - t1:.3f}, ranker init (trunks) {t3 - t2:.3f}, "
          f"{len(cands)} fits incl. pixel draws {t5 - t4:.3f}, {len(cands)} scores {t6 - t5:.3f}; total {t6 - t0:.3f}", flush=True)
