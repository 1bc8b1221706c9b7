import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
from npp_amd import io as nio, search, proposal
from npp_amd.light import ProposalRanker
src = sys.argv[1]
args = search.parse(["--datadir", src, "--random-trunks"])
torch.cuda.set_device(0)
masked_img, img, mask, valid = [f(os.path.join(src, n)) for f, n in ((nio._imread_rgb, "masked_img.png"), (nio._imread_rgb, "gt_img.png"), (nio._imread_gray, "unknown_mask.png"), (nio._imread_gray, "valid_mask.png"))]
m2, v2 = mask[..., 0], valid[..., 0]
for rep in range(2):
    t0 = time.time()
    ang, per, sh = proposal.search_periodicity_by_feat(np.uint8(masked_img * 255), np.uint8(v2 * m2), repeat_range=tuple(args.search_range), edge_searching=True, gray_only=True, device="cuda:0")
    torch.cuda.synchronize(); t1 = time.time()
    _, i_train, i_val = search.pseudo_mask_split(m2, v2); t2 = time.time()
    import warnings; warnings.simplefilter("ignore")
    ranker = ProposalRanker(masked_img.astype(np.float32), i_train, i_val, device="cuda:0"); torch.cuda.synchronize(); t3 = time.time()
    net = ranker.fit_candidate(ang[0], per[0]); torch.cuda.synchronize(); t4 = time.time()
    s = ranker.score(net); torch.cuda.synchronize(); t5 = time.time()
    print(f"rep {rep}: frontend+displacement {t1-t0:.2f} s, pseudo mask {t2-t1:.2f}, ranker init {t3-t2:.2f}, one fit {t4-t3:.3f}, one score {t5-t4:.3f}; {len(ang)} candidates", flush=True)
