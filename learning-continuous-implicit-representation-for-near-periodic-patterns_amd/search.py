"""`python -m npp_amd.search --datadir data/completion/input/<name> --outdir data/completion/detected`: the periodicity
proposal stage, NPP_proposal/search.py:28-280 + loaders/loaders.py:9-66 for one image directory (masked_img.png, gt_img.png,
unknown_mask.png, valid_mask.png):

  feature map (proposal.im2act: gray image, or AlexNet conv1 + gray with --alexnet) -> Canny edge masking (cvlite) ->
  brute-force displacement search per repeat-range group (npp_shift_search) -> pseudo mask around the points furthest from
  the unknown region (utils/miscs.py:53-96) -> one 300-iteration NPP_Net_light fit per candidate, scored by
  30 * LPIPS + 1 * CX on the pseudo-mask region (light.ProposalRanker) -> config.odgt with the top-k candidates.

Flags keep the reference's names (options/arg_config.py:105-145).  Two of them are `store_false` switches there, so the
reference's DEFAULT run is gray_only = True, edge_searching = True -- no AlexNet involved; passing `--gray_only` turns the
AlexNet-conv1 features ON (then --alexnet <torchvision alexnet state_dict> is needed, the checkpoint the reference downloads is
not in its tree).  Under torch.distributed the candidate fits are sharded over the ranks (ProposalRanker.rank)."""
import argparse
import json
import os

import numpy as np
import torch


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--datadir", required=True)
    ap.add_argument("--outdir", default="data/completion/detected")
    ap.add_argument("--netdepth", type=int, default=4)
    ap.add_argument("--netwidth", type=int, default=256)
    ap.add_argument("--N_rand", type=int, default=32 * 32 * 2)
    ap.add_argument("--N_iters", type=int, default=300)
    ap.add_argument("--lrate", type=float, default=5e-4)
    ap.add_argument("--lrate_decay", type=int, default=500)
    ap.add_argument("--gray_only", action="store_false", help="(store_false like the reference) pass it to ADD the AlexNet conv1 features")
    ap.add_argument("--edge_searching", action="store_false", help="(store_false like the reference) pass it to search WITHOUT Canny edge masking")
    ap.add_argument("--topk_detection", type=int, default=10)
    ap.add_argument("--search_range", type=int, nargs=3, default=(1, 10, 1))
    ap.add_argument("--contextual_weight", type=float, default=1.0)
    ap.add_argument("--perceptual_weight", type=float, default=30.0)
    ap.add_argument("--alexnet", default=None, help="torchvision alexnet state_dict (.pth): conv1 of the feature extractor")
    ap.add_argument("--vgg19", default=None, help="torchvision vgg19 state_dict (.pth): contextual-loss trunk of the ranking")
    ap.add_argument("--vgg16", default=None, help="torchvision vgg16 state_dict (.pth): LPIPS trunk of the ranking")
    ap.add_argument("--lpips_lin", default=None, help="lpips weights/v0.1/vgg.pth")
    ap.add_argument("--random-trunks", action="store_true", help="fixed-seed random VGG / AlexNet weights (synthetic runs only)")
    ap.add_argument("--rng_mode", default="reference", choices=["reference", "fast"])
    ap.add_argument("--independent_candidates", "--fast", action="store_true",
                    help="every candidate starts from the INITIAL adaptive pixel-loss latents: the candidates ride in one launch sequence "
                         "(2-3x faster) and shard over ranks.  Default (one rank): the reference's behaviour -- ONE module-level adaptive_pix "
                         "(models/helpers.py:8-9,144) trained through the candidates in order, each starting from the latents the previous "
                         "one left.  Under torch.distributed with more than one rank the independent mode is always used")
    ap.add_argument("--carry_adaptive_latents", action="store_true", help="(the default since round 5; accepted for compatibility)")
    ap.add_argument("--loss_type", default="robust_loss_adaptive", choices=["robust_loss_adaptive", "l2", "robust_loss"],
                    help="options/arg_config.py:34 (models/mse_calculator.py:19-23): the pixel loss of the candidate fits")
    ap.add_argument("--precision", default=None, choices=["fp32", "bf16"], help="arithmetic of the candidate fits (default fp32)")
    ap.add_argument("--device", default="cuda:0")
    return ap.parse_args(argv)


def _carry(args):
    """The reference's shared-latent chaining (models/helpers.py:8-9,144; NPP_proposal/search.py:85-205) unless the candidates are
    asked to be independent or are sharded over ranks (a chain cannot be sharded)."""
    import torch.distributed as dist
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    return not (getattr(args, "independent_candidates", False) or multi)


def find_mask_centroid(mask, topk=3, threshold_ratio=0.3):
    """utils/miscs.py:53-96: the top-k pixels furthest (Euclidean distance transform) from the unknown region and the image
    border... of the mask, at least threshold_ratio * min(H, W) apart.  -> ([[h, w]], [distance])."""
    import scipy.ndimage as ndimage
    m = np.asarray(mask)
    dis = ndimage.distance_transform_edt(m).reshape(-1)
    order = np.argsort(-dis)
    W = m.shape[1]
    thr = min(m.shape[0], m.shape[1]) * threshold_ratio
    cents, sel = [], []
    for idx in order:
        h, w = int(idx // W), int(idx % W)
        if all(np.sqrt((c[0] - h) ** 2 + (c[1] - w) ** 2) >= thr for c in cents):
            cents.append([h, w])
            sel.append(float(dis[idx]))
        if len(cents) == topk:
            break
    return cents, sel


def pseudo_mask_split(mask, valid_mask):
    """loaders/loaders.py:34-54: square holes of half-width dist / sqrt(2) / 1.2 around the centroids; the known pixels inside
    them are the evaluation ('val') region of the candidate fits, the rest of the known pixels train them."""
    m = np.asarray(mask, np.float64).reshape(mask.shape[0], mask.shape[1], 1)
    v = np.asarray(valid_mask, np.float64).reshape(m.shape)
    cents, dist = find_mask_centroid((m * v)[..., 0])
    pseudo = np.ones_like(m)
    for (h, w), d in zip(cents, dist):
        hw = int(d / np.sqrt(2) / 1.2)
        pseudo[max(h - hw, 0):h + hw, max(w - hw, 0):w + hw, :] = 0          # (negative starts would wrap in the reference's slice)
    i_train = np.stack(np.nonzero(pseudo * m * v)[:2], 1)
    i_val = np.stack(np.nonzero((1 - pseudo) * m * v)[:2], 1)
    return pseudo, i_train, i_val


def prepare_image(masked_img, mask, valid_mask, args, conv1=None, trunks=None):
    """The per-image front half of search_image: displacement search -> candidates, pseudo mask -> ranker.  -> (candidates, ranker)."""
    from . import proposal
    from .light import ProposalRanker
    m2 = np.asarray(mask, np.float64).reshape(masked_img.shape[:2])
    v2 = np.asarray(valid_mask, np.float64).reshape(masked_img.shape[:2])
    angles, periods, shifts = proposal.search_periodicity_by_feat(
        np.uint8(np.asarray(masked_img) * 255), np.uint8(v2 * m2), repeat_range=tuple(args.search_range),
        edge_searching=args.edge_searching, gray_only=args.gray_only, conv1=conv1, device=args.device)
    if not angles:
        raise RuntimeError("periodicity search: no displacement pair passed the angle test in any repeat-range group")
    _, i_train, i_val = pseudo_mask_split(m2, v2)
    t = trunks or {}
    args.carry_effective = _carry(args)
    ranker = ProposalRanker(masked_img, i_train, i_val, device=args.device, N_iters=args.N_iters, N_rand=args.N_rand, W=args.netwidth,
                            D=args.netdepth, lrate=args.lrate, lrate_decay=args.lrate_decay, perceptual_weight=args.perceptual_weight,
                            contextual_weight=args.contextual_weight, vgg19_state_dict=t.get("vgg19"), vgg16_state_dict=t.get("vgg16"),
                            lpips_lin_weights=t.get("lin"), rng_mode=args.rng_mode, carry_latents=args.carry_effective,
                            loss_type=getattr(args, "loss_type", "robust_loss_adaptive"), precision=getattr(args, "precision", None))
    return list(zip(angles, periods, shifts)), ranker


def _ranked(cands, dist, order, details):
    angles, periods, shifts = zip(*cands)
    return {"angles": [np.asarray(angles[i], np.float64).tolist() for i in order],
            "periods": [np.asarray(periods[i], np.float64).tolist() for i in order],
            "shifts": [[list(map(float, s)) for s in shifts[i]] for i in order],
            "distances": [float(d) for d in dist], "n_candidates": len(cands), "details": details}


def search_image(masked_img, mask, valid_mask, args, conv1=None, trunks=None):
    """masked_img (H,W,3) in [0,1]; mask / valid_mask (H,W[,1]) 1 = known / valid.  -> dict with the ranked candidates."""
    cands, ranker = prepare_image(masked_img, mask, valid_mask, args, conv1, trunks)
    dist, order, details = ranker.rank(cands, topk=args.topk_detection)
    return _ranked(cands, dist, order, details)


def _load(args):
    """Flags -> (output directory, the four input images, feature extractor, trunk weights); SystemExit when the output exists."""
    from . import io as nio
    from . import weights
    weights.resolve(args, ["vgg19", "vgg16"] + ([] if args.gray_only else ["alexnet"]), args.random_trunks)
    torch.cuda.set_device(torch.device(args.device))
    name = os.path.basename(os.path.normpath(args.datadir))
    out = os.path.join(args.outdir, name)
    if os.path.exists(out):
        raise SystemExit("Searching: file exists, exit!!")                           # search.py:42-44
    rd = lambda f: nio._imread_rgb(os.path.join(args.datadir, f))                   # noqa: E731  ([0, 1], loaders/loaders.py:17-25)
    rg = lambda f: nio._imread_gray(os.path.join(args.datadir, f))                  # noqa: E731
    masked_img, img, mask, valid = rd("masked_img.png"), rd("gt_img.png"), rg("unknown_mask.png"), rg("valid_mask.png")
    load = weights.load_state_dict                                                  # (one read per file and process)
    lin = None if (args.random_trunks and args.lpips_lin is None and args.vgg16 is None) else weights.lpips_lin("vgg", args.lpips_lin)
    conv1 = None
    if not args.gray_only:
        from .proposal import AlexConv1
        conv1 = AlexConv1(load(args.alexnet), device=args.device, allow_random=args.random_trunks)
    return out, (masked_img, img, mask, valid), conv1, {"vgg19": load(args.vgg19), "vgg16": load(args.vgg16), "lin": lin}


def _write(args, out, imgs, res):
    from . import io as nio
    masked_img, img, mask, valid = imgs
    nio.write_detected_dir(out, img, mask, valid, res["angles"], res["periods"], res["shifts"], res["distances"], masked_img=masked_img,
                           draw=True)
    with open(os.path.join(out, "config.odgt")) as f:
        odgt = json.loads(f.readline())
    odgt.update(search_range=list(args.search_range), epoch=args.N_iters)           # search.py:236-237
    # (not a reference field) how the candidates' adaptive-loss latents were handled: the reference trains ONE module-level
    # adaptive_pix through all candidates in order (models/helpers.py:8,144), so its distances depend on the candidate order; the
    # single-rank default here does the same; --independent_candidates / more than one rank start every candidate from the initial
    # latents.  Recorded so that rankings are compared like with like.
    odgt.update(carry_adaptive_latents=bool(getattr(args, "carry_effective", _carry(args))))
    with open(os.path.join(out, "config.odgt"), "w") as f:
        json.dump(odgt, f)
        f.write("\n")
    print(f"[search] {res['n_candidates']} candidates ranked; best periods {res['periods'][0]} angles {res['angles'][0]} "
          f"distance {res['distances'][0]:.4f} -> {out}/config.odgt")


def main(argv=None):
    args = parse(argv)
    out, imgs, conv1, trunks = _load(args)
    res = search_image(imgs[0].astype(np.float32), imgs[2], imgs[3], args, conv1, trunks)
    _write(args, out, imgs, res)
    return 0


def main_multi(argvs):
    """Several image directories of one rank searched TOGETHER (light.rank_images: candidate k of every image in one launch
    sequence).  Per image the result is that of main(); a failure -- an existing output, an unreadable input, no displacement found
    -- is recorded for that image only.  -> list of None / the exception (SystemExit for "file exists") per argv."""
    from .light import rank_images
    n = len(argvs)
    errors, prepared = [None] * n, [None] * n
    def prep(i):
        # the images' front halves side by side: PNG decoding, the displacement search (host NumPy + a few small launches) and the
        # ranker's tables are 30 ms per image of mostly host time -- each on its own thread and stream, like run.search_all's thread form
        try:
            args = parse(argvs[i])
            st = torch.cuda.Stream(torch.device(args.device)) if torch.cuda.is_available() else None
            with (torch.cuda.stream(st) if st is not None else contextlib.nullcontext()):
                out, imgs, conv1, trunks = _load(args)
                cands, ranker = prepare_image(imgs[0].astype(np.float32), imgs[2], imgs[3], args, conv1, trunks)
            if st is not None:
                st.synchronize()
            prepared[i] = (args, out, imgs, cands, ranker)
        except (Exception, SystemExit) as e:                                         # noqa: B014
            errors[i] = e
    import contextlib
    from concurrent.futures import ThreadPoolExecutor
    if n > 1:
        with ThreadPoolExecutor(min(8, n), thread_name_prefix="npp-prep") as pool:
            list(pool.map(prep, range(n)))
    else:
        prep(0)
    live = [i for i in range(n) if prepared[i] is not None]
    # images whose fits share hyper-parameters ride together; anything else falls back to its own loop inside rank_images' check
    keyf = lambda pr: (pr[0].N_iters, pr[0].netwidth, pr[0].netdepth, pr[0].lrate, pr[0].lrate_decay, pr[0].loss_type, pr[4].carry_latents,  # noqa: E731
                       pr[0].precision, pr[0].topk_detection, str(pr[4].device))
    groups = {}
    for i in live:
        groups.setdefault(keyf(prepared[i]), []).append(i)
    for key, members in groups.items():
        try:
            ranked = rank_images([prepared[i][4] for i in members], [prepared[i][3] for i in members], topk=key[8])
        except Exception as e:                                                       # the group's launches failed: every member is affected
            for i in members:
                errors[i] = e
            continue
        def put(job):                                                                # (PNG encoding releases the GIL: the images' 13 files each
            i, (dist, order, details) = job                                          #  are written side by side -- 0.8 s -> 0.15 s for 8 images)
            try:
                args, out, imgs, cands, _ = prepared[i]
                _write(args, out, imgs, _ranked(cands, dist, order, details))
            except Exception as e:
                errors[i] = e
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(min(8, len(members))) as pool:
            list(pool.map(put, zip(members, ranked)))
    return errors


if __name__ == "__main__":
    raise SystemExit(main())
