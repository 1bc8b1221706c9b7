"""Completion driver: what `python NPP_completion/train.py --datadir D --basedir B --p_topk K` does (run_completion.sh:13),
on the HIP path.

    python -m npp_amd.train --datadir data/completion/detected/<name> --basedir ./results --p_topk 3

Reads config.odgt + PNGs (npp_amd.io), builds the net with torch's default nn.Linear init and Gaussian Fourier
frequencies drawn from torch's global generator like the reference (models/embedder.py:26, models/networks.py:40-49;
seed with --seed), runs N_iters iterations of the complete loop body (CompletionFit.step_full) and writes
results/<expname>_top<K>/<name>/testset_<iter>/*.png every --i_testset iterations (train.py:270-328).
Flags keep the reference's names and defaults (options/arg_config.py:10-36,55-100), --netwidth included: the default is the
reference's 512 (the second fused build, libnpp_hip_w512.so); BASELINE.json's configurations are --netwidth 256.
"""
import argparse
import os
import time

import numpy as np
import torch


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--task", default="completion", choices=["completion", "remapping", "segmentation"],
                    help="completion: NPP_completion/train.py; remapping: NPP_remapping/train.py (whole image trained, blur-detected clear "
                         "mask as sampler / pixel-weight mask, Gram style loss; its defaults: contextual_weight 0.01, style_weight 1, no LPIPS); "
                         "segmentation: NPP_segmentation/train.py (fit on the masked-blurred image with the initial periodic region as "
                         "the known mask, constant LR, contextual_weight 0.005, no LPIPS, 601 iterations, then the L1 + LPIPS(alex) "
                         "criteria; needs period_mask.png / non_period_mask.png beside config.odgt)")
    ap.add_argument("--alexnet", default=None, help="torchvision alexnet state_dict (.pth) for the segmentation criterion")
    ap.add_argument("--lpips_alex_lin", default=None, help="lpips weights/v0.1/alex.pth (segmentation criterion)")
    ap.add_argument("--l1_thresh", type=float, default=0.15)
    ap.add_argument("--lpips_thresh", type=float, default=0.3)
    ap.add_argument("--lpips_layers", type=int, default=1)
    ap.add_argument("--blur_thresh", type=float, default=50)
    ap.add_argument("--contextual_weight", type=float, default=None)
    ap.add_argument("--style_weight", type=float, default=1.0)
    ap.add_argument("--datadir", required=True)
    ap.add_argument("--basedir", default="./results")
    ap.add_argument("--expname", default="completion")
    ap.add_argument("--p_topk", type=int, default=3)
    ap.add_argument("--N_iters", type=int, default=None, help="default 2001 (completion, arg_config.py) / 2801 (remapping, NPP_remapping)")
    ap.add_argument("--N_rand", type=int, default=8192)
    ap.add_argument("--lrate", type=float, default=5e-4)
    ap.add_argument("--lrate_decay", type=int, default=500)
    ap.add_argument("--netwidth", type=int, default=512, help="512: the reference's default (arg_config.py:57); 256: the BASELINE.json configurations")
    ap.add_argument("--netdepth", type=int, default=8)
    ap.add_argument("--activation", default="snake")
    ap.add_argument("--loss_type", default="robust_loss_adaptive")
    ap.add_argument("--normalize_type", type=int, default=1)
    # the reference's ablation switches (arg_config.py:78-92), with its store_false / store_true quirks kept
    ap.add_argument("--no_reg_sampling", action="store_true", help="real patches drawn at random instead of along the lattice (sampler.py:219-228)")
    ap.add_argument("--use_comp", action="store_false", help="(store_false) pass it to NOT paste the known region into 'val' patches")
    ap.add_argument("--use_perceptual_loss", action="store_true",
                    help="toggles the task's default: completion has LPIPS on 'same' iterations ON (store_false there, :84), "
                         "segmentation / remapping have it OFF (store_true, :190,274)")
    ap.add_argument("--perceptual_weight", type=float, default=1e-3)
    ap.add_argument("--use_contextual_loss", action="store_false", help="(store_false) pass it to drop the contextual term")
    ap.add_argument("--use_adaptive_perceptual_loss", action="store_false",
                    help="(store_false, as in arg_config.py:78) given: LPIPS(use_robust=False) in the loop instead of the adaptive-robust head")
    ap.add_argument("--use_patch_weight", action="store_true", help="1/d lattice weights on the patch terms (train.py:224-250)")
    ap.add_argument("--no_pix_loss", action="store_true", help="no pixel loss (train.py:197-198)")
    ap.add_argument("--patch_num", type=int, default=2)
    ap.add_argument("--num_real_patch_per_sample", type=int, default=3)
    ap.add_argument("--invalid_ratio", type=float, default=0.3)
    ap.add_argument("--patch_size_decay", type=int, default=2000)
    ap.add_argument("--i_testset", type=int, default=None, help="default 500 (completion) / 400 (remapping)")
    ap.add_argument("--i_print", type=int, default=500)
    ap.add_argument("--invalid_as_unknown", action="store_true")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--rng_mode", default="reference", choices=["reference", "numpy", "fast"],
                    help="reference: NumPy's stream from the native generator; numpy: NumPy itself; fast: O(size) draws, other stream")
    ap.add_argument("--prefetch", type=int, default=8, help="iterations of sampler draws prepared ahead on a producer thread")
    ap.add_argument("--vgg19", default=None, help="torchvision vgg19 state_dict (.pth) for the contextual loss trunk")
    ap.add_argument("--vgg16", default=None, help="torchvision vgg16 state_dict (.pth) for the LPIPS trunk")
    ap.add_argument("--lpips_lin", default=None, help="lpips weights/v0.1/vgg.pth (the five 1x1 lin layers)")
    ap.add_argument("--random-trunks", action="store_true",
                    help="run WITHOUT pretrained VGG19 / VGG16 / LPIPS-lin weights (fixed-seed random trunks: synthetic and bench "
                         "runs only -- the losses then differ from the reference's)")
    ap.add_argument("--device", default="cuda:0")
    return ap.parse_args(argv)


def default_linear_init(layout, n_params, seed):
    """torch nn.Linear default init per tensor of the blob (kaiming_uniform(a=sqrt 5) weight, uniform bias), drawn
    tensor by tensor in state_dict order from a seeded CPU generator."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    fan_in = {}
    for name, off, rows, cols in layout:
        if name.endswith("weight"):
            fan_in[name[:-7]] = cols
            bound = 1.0 / np.sqrt(cols)                       # kaiming_uniform with a = sqrt(5): sqrt(6 / ((1 + 5) fan_in))
            sd[name] = ((torch.rand(rows, cols, generator=g) * 2 - 1) * bound).numpy()
        else:
            bound = 1.0 / np.sqrt(fan_in[name[:-5]])
            sd[name] = ((torch.rand(rows * cols, generator=g) * 2 - 1) * bound).numpy()
    return sd


class _Job:
    """One image's fit between _prepare() and _finish(): arguments, loaded data, the CompletionFit, its output directory and the
    PNG writer thread."""
    pass


def _prepare(argv=None, stacked=False):
    """Everything `main` does before the loop (NPP_completion/train.py:28-131): flags, data, output directory, network init, the
    fit object.  -> _Job, or None when the result directory already exists (train.py:42-44).  stacked: the fit will ride in a
    StackedFit (which draws for every image itself: no producer thread)."""
    plan = _plan(argv)
    return None if plan is None else _build(plan, stacked)


def _plan(argv=None):
    """The side-effect-free half of _prepare: flags, the loaded detection, the output directory's NAME.  Creates nothing and touches
    no GPU memory, so a directory run can plan every image, group them by batch shape, and create an image's result directory only
    when its group starts fitting.  -> _Job without a fit, or None when the result directory already exists (train.py:42-44)."""
    args = parse(argv)
    if args.N_iters is None:
        args.N_iters = {"remapping": 2801, "segmentation": 601}.get(args.task, 2001)     # arg_config.py:96,202,289
    if args.i_testset is None:
        args.i_testset = {"remapping": 400, "segmentation": 600}.get(args.task, 500)
    if args.netwidth not in (256, 512):
        raise SystemExit("the fused chain is built for --netwidth 256 (BASELINE.json) and 512 (the reference's default)")
    refused = [n for n, bad in (("--netdepth != 8", args.netdepth != 8), ("--activation != snake", args.activation != "snake"),
                                ("--loss_type not in robust_loss_adaptive / l2 / robust_loss", args.loss_type not in ("robust_loss_adaptive", "l2", "robust_loss")),
                                ("--normalize_type not in 1 / 2", args.normalize_type not in (1, 2)),
                                ) if bad]
    if refused:
        raise SystemExit(f"{refused}: ablation switches of options/arg_config.py that the fused loop is not built for (D = 8, snake, sigmoid / tanh output); other widths / depths / activations run through reference_api.NPP_Net (dense.py)")
    seg_task = args.task == "segmentation"
    from . import weights
    names = ["vgg19"] + ([] if seg_task else ["vgg16"]) + (["alexnet"] if seg_task else [])    # remapping: VGG16 is the style trunk
    weights.resolve(args, names, args.random_trunks)
    torch.cuda.set_device(torch.device(args.device))
    from . import io as nio
    from ._lib import param_layout
    from .fit import CompletionFit
    remap = args.task == "remapping"
    seg = args.task == "segmentation"
    if remap:
        d = nio.load_npp_remapping(args.datadir, args.p_topk, args.blur_thresh)
        d["mask"], d["masked_img"] = d["clear_mask"], d["img"]
    elif seg:
        d = nio.load_npp_segmentation(args.datadir, args.p_topk)
        d["mask"], d["masked_img"] = d["period_mask"], d["blur_img"]
    else:
        # --normalize_type 2: tanh output; the reference rescales only the evaluation image (loaders.py:111), training stays on
        # masked_img in [0, 1] (train.py:173) -- reproduced, not fixed
        d = nio.load_npp_completion(args.datadir, args.p_topk, args.invalid_as_unknown, normalize_type=args.normalize_type)
    name = os.path.basename(os.path.normpath(args.datadir))
    expname = args.expname if not ((remap or seg) and args.expname == "completion") else args.task
    outroot = os.path.join(args.basedir, f"{expname}_top{args.p_topk}", name)
    if os.path.exists(outroot):                                                             # train.py:42-44: results are never overwritten
        print(f"{args.task.capitalize()}: file exists, exit!!")
        return None
    plan = _Job()
    plan.args, plan.d, plan.outroot, plan.name, plan.remap, plan.seg, plan.fit = args, d, outroot, name, remap, seg, None
    return plan


def _build(plan, stacked=False):
    """The other half: the result directory (from here on a failure removes it again) and the CompletionFit."""
    from . import weights
    from . import io as nio
    from ._lib import param_layout
    from .fit import CompletionFit
    args, d, outroot, name, remap, seg = plan.args, plan.d, plan.outroot, plan.name, plan.remap, plan.seg
    os.makedirs(outroot, exist_ok=True)
    print("Loaded NPP", d["img"].shape, args.datadir)
    print("selected_angles: " + str(np.asarray(d["angles"]).tolist()))
    print("selected_periods: " + str(np.asarray(d["periods"]).tolist()))
    K = len(d["angles"])
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    freqs = (torch.normal(mean=0.0, std=1.0, size=(10, 1)) * 10).reshape(-1).numpy()       # embedder.py:26
    layout, n_params = param_layout(K, args.netwidth)
    params = default_linear_init(layout, n_params, args.seed)

    load = weights.load_state_dict                           # (one read per file and process: shared dicts -> shared packed trunk weights)
    lin = None if (args.random_trunks and args.lpips_lin is None and args.vgg16 is None) else weights.lpips_lin("vgg", args.lpips_lin)
    try:
        fit = CompletionFit(d["img"], d["mask"], d["angles"], d["periods"], freqs, params, device=args.device, N_rand=args.N_rand,
                            seed=args.seed, lrate=args.lrate, lrate_decay=args.lrate_decay, valid_mask=d["valid_mask"],
                            shifts=d["shifts"], patch_size=d["patch_size"], patch_num=args.patch_num,
                            num_real_patch_per_sample=args.num_real_patch_per_sample, invalid_ratio=args.invalid_ratio,
                            patch_size_decay=args.patch_size_decay, vgg19_state_dict=load(args.vgg19),
                            # --vgg16: the LPIPS trunk (completion) or the STYLE trunk of the remapping task (models/style_loss.py:11,
                            # VGG16FeatureExtractor; LPIPS is off there, NPP_remapping/train.py:253-261)
                            vgg16_state_dict=None if remap else load(args.vgg16),
                            vgg16_style_state_dict=load(args.vgg16) if remap else None,
                            lpips_lin_weights=lin, rng_mode=args.rng_mode, prefetch=0 if stacked else args.prefetch,
                            task=args.task, clear_mask=d["clear_mask"] if remap else None,
                            masked_img=None if remap else d.get("masked_img"),
                            contextual_weight=args.contextual_weight if args.contextual_weight is not None else (0.01 if remap else (0.005 if seg else 1e-3)),
                            style_weight=args.style_weight if remap else None,
                            use_perceptual_loss=(not (remap or seg)) != args.use_perceptual_loss, perceptual_weight=args.perceptual_weight,
                            use_comp=args.use_comp, no_reg_sampling=args.no_reg_sampling, use_patch_weight=args.use_patch_weight,
                            no_pix_loss=args.no_pix_loss, use_contextual_loss=args.use_contextual_loss, width=args.netwidth,
                            loss_type=args.loss_type, use_adaptive_perceptual_loss=args.use_adaptive_perceptual_loss,
                            normalize_type=args.normalize_type)
    except BaseException:
        import shutil
        shutil.rmtree(outroot, ignore_errors=True)          # (nothing was fitted: a re-run must not take the directory for a result)
        raise
    # the six PNGs of a test set take ~150 ms to encode (zlib, GIL released): written behind the loop, joined before returning
    from concurrent.futures import ThreadPoolExecutor
    job = plan
    job.args, job.fit, job.d, job.outroot, job.seg, job.load, job.weights, job.nio = args, fit, d, outroot, seg, load, weights, nio
    job.writer, job.pending, job.t0, job.name = ThreadPoolExecutor(1), [], time.time(), name
    return job


def _finish(job, failed):
    """The end of a fit, successful or not: producer thread stopped, queued PNG writers awaited; a fit that died mid-loop must not
    leave an output directory behind that a re-run would take for a finished one (`file exists, exit!!`) -- but an interrupted one
    (Ctrl-C) keeps the test sets it has already written.  Raises a writer's error when the loop itself succeeded."""
    write_error = None
    job.fit.close()                                         # the sampler's producer thread
    # first the writers: a queued dump_testset re-creates its directory (os.makedirs(..., exist_ok=True)), so nothing is
    # removed while one may still run
    for p in job.pending:
        if not p.cancel():
            try:
                p.result()
            except Exception as e:
                write_error = write_error or e
                print(f"[WARN] writing a test set failed: {e}")
    job.writer.shutdown(wait=True)
    if (failed is not None and not isinstance(failed, KeyboardInterrupt)) or write_error is not None:
        import shutil
        shutil.rmtree(job.outroot, ignore_errors=True)
    if write_error is not None and failed is None:          # the loop itself succeeded: a lost output is still a failed run
        raise write_error


def main(argv=None):
    job = _prepare(argv)
    if job is None:
        return None
    failed = None
    try:
        for i in range(1, job.args.N_iters):                                                # trange(start = 1, N_iters)
            job.fit.step_full()
            _after_iteration(job, i)
    except BaseException as e:
        failed = e
        raise
    finally:
        _finish(job, failed)
    return job.fit


def stack_key(job):
    """Fits that can share one launch sequence (stack.StackedFit): same network shape, batch shape, loss switches and schedule."""
    a, f = job.args, job.fit
    if f.patch_sampler is None or f.use_patch_weight or not f.use_contextual_loss or f.net.out_act != 1:
        return None
    return (a.task, f.style is not None, f.style_w, f.pixel_mask is not None, f.net.K, f.net.width, f.N_rand, f.patch_size, f.patch_num, f.topk, f.net.quad, f.pix_w, f.use_comp, f.cx_w, f.lp_w,
            f.lp_robust, f.use_perceptual_loss, a.N_iters, a.i_testset, a.i_print, a.patch_size_decay, str(f.device))


def plan_key(plan):
    """stack_key's refinement that needs no fit: every flag that shapes the loop + what the detection fixes (K, patch size).  Plans of
    one key build fits of one stack_key (checked again on the built fits)."""
    a, d = plan.args, plan.d
    flags = tuple(sorted((k, str(v)) for k, v in vars(a).items() if k not in ("datadir", "expname", "seed")))
    return (flags, len(d["angles"]), d["patch_size"], tuple(np.asarray(d["img"]).shape))


class FitResult:
    """What a directory run keeps of a finished fit.  The fit itself -- weights, stashes, the trunks' activation buffers, captured
    LPIPS graphs -- is released with its group: device memory no longer grows with the number of images of a rank."""

    def __init__(self, job):
        f = job.fit
        self.name, self.outroot = job.name, job.outroot
        self.patch_size, self.patch_num, self.skipped = f.patch_size, f.patch_num, f.skipped
        self.psnr_known, self.psnr_unknown = f.psnr("known"), f.psnr("unknown")
        self.opt_step = f.net.opt_step
        self.has_style, self.style_lat_step = f.style is not None, (f.style.lat_step if f.style is not None else None)
        self.has_pixel_mask = f.pixel_mask is not None

    def psnr(self, region="known"):
        return self.psnr_known if region == "known" else self.psnr_unknown

    def close(self):
        pass


def main_stacked(argvs, max_stack=8):
    """Several images' fits in ONE launch sequence per group (stack.StackedFit): what `python -m npp_amd.run` does when a rank has
    more than one image (run_completion.sh:8-14 loops them serially; the fits are independent -- own weights, Adam state, random
    stream -- so the image becomes a grid dimension: 1.4 x the rows per second of the serial loop at 8 images per GPU).  Images group by
    batch shape (plan_key / stack_key: detected periods give patch sizes 64 .. 160, loaders.py:133-134); a group of one runs the
    plain loop.  Every image is PLANNED first (flags, detection: no directory, no device memory); an image's result directory is
    created and its fit built when its group starts, a failure -- of the plan, of the build, of the loop -- is recorded for the
    images it concerns only, and a group's fits are released before the next group is built.
    -> list of FitResult (None for skipped / failed outputs), in the order of argvs; main_stacked.last_error holds the first failure,
    main_stacked.errors the failure per argv (None where there was none)."""
    from .stack import StackedFit
    n = len(argvs)
    plans, errors, results = [None] * n, [None] * n, [None] * n
    for idx, a in enumerate(argvs):
        try:
            plans[idx] = _plan(a)
        except (Exception, SystemExit) as e:                                                # noqa: B014  (one image's bad flag / detection)
            errors[idx] = e
            print(f"[stack] {a}: not fitted ({type(e).__name__}: {e})")
    groups = {}
    for idx, plan in enumerate(plans):
        if plan is not None:
            groups.setdefault(plan_key(plan), []).append(idx)
    for key, members in groups.items():
        for c0 in range(0, len(members), max_stack):
            chunk_idx = members[c0:c0 + max_stack]
            jobs = []                                                                       # (idx, job) of the images whose fit was built
            for idx in chunk_idx:
                try:
                    jobs.append((idx, _build(plans[idx], stacked=len(chunk_idx) > 1)))
                except Exception as e:
                    import shutil
                    shutil.rmtree(plans[idx].outroot, ignore_errors=True)                   # (it did not exist when the image was planned)
                    errors[idx] = e
                    print(f"[stack] {plans[idx].name}: not fitted ({type(e).__name__}: {e})")
            if not jobs:
                continue
            chunk = [j for _, j in jobs]
            failed, st = None, None
            try:
                keys = {stack_key(j) for j in chunk}
                if len(chunk) == 1 or len(keys) != 1 or None in keys:
                    for job in chunk:
                        for i in range(1, job.args.N_iters):
                            job.fit.step_full()
                            _after_iteration(job, i)
                else:
                    st = StackedFit([j.fit for j in chunk])
                    print(f"[stack] {len(chunk)} images per launch sequence: {[j.name for j in chunk]} (patch size {st.P}, wgrad split-K {st.ksplit})")
                    for i in range(1, chunk[0].args.N_iters):
                        if st.shape_change_due():                                           # patch-size decay (train.py:137-141): new batch shape
                            st = st.restacked()
                        st.step_full()
                        for job in chunk:
                            _after_iteration(job, i)
            except BaseException as e:                                                      # noqa: B902
                failed = e
                if not isinstance(e, KeyboardInterrupt):
                    import traceback
                    traceback.print_exc()
                    print(f"[stack] group {[j.name for j in chunk]} FAILED ({type(e).__name__}: {e})")
            finally:
                if st is not None:
                    st.close()                                                              # its draw threads; the fits' own close() follows
                for idx, job in jobs:
                    try:
                        if failed is None:
                            results[idx] = FitResult(job)
                        _finish(job, failed)
                    except Exception as e:                                                  # a lost output of one image
                        errors[idx] = errors[idx] or e
                        results[idx] = None
                    if failed is not None:
                        errors[idx] = errors[idx] or failed
                        results[idx] = None
                    job.fit = None                                                          # release the group's device memory ...
                del st, chunk, jobs
                if torch.cuda.is_available():
                    torch.cuda.empty_cache()                                                # ... before the next group allocates its own
            if isinstance(failed, KeyboardInterrupt):
                raise failed
    main_stacked.errors = errors
    main_stacked.last_error = next((e for e in errors if e is not None), None)
    return results


def _after_iteration(job, i):
    """What follows optimizer.step() in the reference's loop body at iteration i: the periodic evaluation dump (train.py:270-331;
    segmentation: NPP_segmentation/train.py:337-406) and the progress line."""
    args, fit, d, outroot, seg, load, weights, nio, writer, pending, t0 = (job.args, job.fit, job.d, job.outroot, job.seg, job.load,
                                                                           job.weights, job.nio, job.writer, job.pending, job.t0)
    if i % args.i_testset == 0:
        pred = fit.render_image().cpu().numpy()
        os.makedirs(os.path.join(outroot, f"testset_{i:06d}"), exist_ok=True)
        pending.append(writer.submit(nio.dump_testset, os.path.join(outroot, f"testset_{i:06d}"), pred, d["img"], d["masked_img"], d["mask"],
                                     d["valid_mask"]))
        print(f"[EVAL] iter {i}: PSNR known {fit.psnr('known'):.2f} dB, unknown {fit.psnr('unknown'):.2f} dB")
        if seg:                                                                         # NPP_segmentation/train.py:337-406
            from . import segment
            alex = segment.AlexFeatures(load(args.alexnet), device=args.device)
            lins = weights.lpips_lin("alex", args.lpips_alex_lin)
            r = segment.segmentation_eval(pred * d["valid_mask"], d["blur_img"], d["valid_mask"], d["non_period_mask"], alex, lins,
                                          args.l1_thresh, args.lpips_thresh, args.lpips_layers)
            tdir = os.path.join(outroot, f"testset_{i:06d}")
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt                                             # single-channel plt.imsave: viridis, autoscaled (:356-389)
            plt.imsave(os.path.join(tdir, "l1_diff_img.png"), r["l1_img"])
            plt.imsave(os.path.join(tdir, "l1_img_mask.png"), ~r["l1_mask"])
            for j, m in enumerate(r["lpips_maps"]):
                plt.imsave(os.path.join(tdir, f"lpips_diff_img_{j}.png"), m)
                plt.imsave(os.path.join(tdir, f"lpips_img_mask_{j}.png"), ~(m < args.lpips_thresh))
            nio.imsave(os.path.join(tdir, "non_period_mask_final.png"), np.repeat(r["non_period_mask_final"].astype(np.float64), 3, 2))
            m = r["non_period_mask_final"].astype(np.float64)
            vis = d["img"] * 0.7 + 0.3 * (np.array([0.0, 1.0, 0.0]) * m + d["img"] * (1 - m))    # :396-404
            nio.imsave(os.path.join(tdir, "segment.png"), vis * d["valid_mask"])
            fit.segmentation = r
    if i % args.i_print == 0:
        print(f"[TRAIN] Iter: {i} Loss: {float(fit.net.loss_buf[0]):.6f} Patch Loss: {float(fit.last_patch_loss[0]):.6f} "
              f"({(time.time() - t0) / i * 1e3:.2f} ms/iter, skipped {fit.skipped})")


if __name__ == "__main__":
    main()
