"""`python -m npp_amd.run --task completion|segmentation|remapping`: the reference's run_completion.sh / run_segmentation.sh /
run_remapping.sh -- for every directory under data/<task>/input: the periodicity search (NPP_proposal/search.py ->
data/<task>/detected/<name>), then the task's fit (NPP_<task>/train.py -> results/<task>_top<k>/<name>) -- as ONE process per GPU that
keeps the library, the packed loss trunks' code and the CUDA context across images instead of two interpreter starts per image.
Launched under torch.distributed.run (one rank per GPU; or `--gpus N`, which starts the ranks itself) the image directories are
sharded over the ranks (parallel.shard_units): independent images, no data-path collective, no process group.
An image whose search / fit output already exists is skipped like in the reference (search.py:42-44, train.py:42-44); an image that
fails is reported and the loop goes on (the shell loop's behaviour); exit status 1 if any failed."""
import argparse
import glob
import os
import shlex
import sys
import time
import traceback


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--task", default="completion", choices=["completion", "segmentation", "remapping"])
    ap.add_argument("--input_path", default=None, help="default data/<task>/input")
    ap.add_argument("--detected_path", default=None, help="default data/<task>/detected")
    ap.add_argument("--basedir", default="./results")
    ap.add_argument("--p_topk", type=int, default=3)
    ap.add_argument("--random-trunks", action="store_true", help="passed to both stages (synthetic runs)")
    ap.add_argument("--search-args", default="", help="extra flags for npp_amd.search, one quoted string")
    ap.add_argument("--train-args", default="", help="extra flags for npp_amd.train, one quoted string")
    ap.add_argument("--gpus", type=int, default=None, help="start this many ranks (one per GPU) unless already under torch.distributed.run")
    ap.add_argument("--stack", type=int, default=8,
                    help="images of a rank fitted TOGETHER, one launch sequence for all of them (npp_amd.stack: the image is a grid "
                         "dimension; ~1.4 x the rows per second at 8).  1: one image after the other like the shell loop.  Images are "
                         "searched first, then grouped by batch shape (patch size) and fitted")
    ap.add_argument("--search-threads", type=int, default=None,
                    help="images of a rank SEARCHED side by side, each on its own host thread and stream (a candidate fit is a chain of "
                         "small launches that leaves the chip idle).  1: one after the other.  Default (flag absent): candidate k of every "
                         "image of the rank in ONE launch sequence (search.main_multi)")
    return ap.parse_args(argv)


def my_share(items):
    """The image directories of this rank (RANK / WORLD_SIZE of torch.distributed.run; everything when not launched by it)."""
    from .parallel import shard_units
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    return [items[i] for i in shard_units(len(items), rank, world)], rank, world


def main(argv=None, search_main=None, train_main=None):
    args = parse(argv)
    if args.gpus and args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        from .parallel import launch_ranks                      # fresh child processes: this one has not touched the GPU
        me = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "run_ranks.py")
        return launch_ranks(me, args.gpus, list(sys.argv[1:] if argv is None else argv))
    inp = args.input_path or os.path.join("data", args.task, "input")
    det = args.detected_path or os.path.join("data", args.task, "detected")
    dirs = sorted(p for p in glob.glob(os.path.join(inp, "*")) if os.path.isdir(p))
    mine, rank, world = my_share(dirs)
    device = f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}"
    user_search = search_main                                    # None: the package's own search (its multi-image form is then available)
    if search_main is None:
        from .search import main as search_main
    common = ["--device", device] + (["--random-trunks"] if args.random_trunks else [])
    failed, t_all = [], time.time()
    if args.stack > 1 and len(mine) > 1 and train_main is None:
        return _main_stacked(args, mine, det, common, rank, world, user_search)
    if train_main is None:
        from .train import main as train_main
    for src in mine:
        name = os.path.basename(os.path.normpath(src))
        t0 = time.time()
        try:
            try:
                search_main(["--datadir", src, "--outdir", det] + common + shlex.split(args.search_args))
            except SystemExit as e:                                # "Searching: file exists, exit!!": the detected directory is reused
                if "exists" not in str(e):
                    raise
                print(e)
            t1 = time.time()
            fit = train_main(["--datadir", os.path.join(det, name), "--basedir", args.basedir, "--p_topk", str(args.p_topk)] + common
                             + (["--task", args.task] if args.task != "completion" else []) + shlex.split(args.train_args))
            if fit is not None and hasattr(fit, "close"):
                fit.close()
            print(f"[run rank {rank}/{world}] {args.task}/{name}: search {t1 - t0:.1f} s, fit {time.time() - t1:.1f} s", flush=True)
        except (Exception, SystemExit) as e:                       # noqa: B014  -- report, keep going like the shell loop
            traceback.print_exc()
            print(f"[run rank {rank}/{world}] {args.task}/{name}: FAILED ({type(e).__name__}: {e})", flush=True)
            failed.append(name)
    print(f"[run rank {rank}/{world}] {len(mine) - len(failed)} of {len(mine)} images done in {time.time() - t_all:.1f} s"
          + (f"; failed: {failed}" if failed else ""), flush=True)
    return 1 if failed else 0


def search_all(srcs, det, flags, search_main=None, threads=8, together=True):
    """The periodicity search of every directory in srcs (-> det/<name>), `threads` images side by side: each on its own host thread
    and HIP stream.  The search of one image is nine candidate fits in a row (the reference chains them through one set of
    adaptive-loss latents, models/helpers.py:8,144), each 300 iterations of four small launches -- a dependent chain that leaves the
    chip idle; images are independent, so their chains interleave on the device.  Per image the results are those of the serial
    loop (own random streams, own score trunks over the process's shared packed weights; the process-wide torch generator is borrowed
    under ops.RNG_LOCK).  -> list of None / the exception per directory ("file exists" is not an error: the directory is reused)."""
    if search_main is None and together and len(srcs) > 1:
        # round 6: candidate k of EVERY image in one launch sequence (search.main_multi / light.rank_images) -- the images' chains of
        # nine candidate fits become one chain of nine stacked fits; per image the bits of its own serial loop
        from .search import main_multi
        out = []
        for e in main_multi([["--datadir", s_, "--outdir", det] + list(flags) for s_ in srcs]):
            if isinstance(e, SystemExit) and "exists" in str(e):   # "Searching: file exists, exit!!": the detected directory is reused
                print(e)
                e = None
            elif e is not None:
                traceback.print_exception(type(e), e, e.__traceback__)
            out.append(e)
        return out
    if search_main is None:
        from .search import main as search_main

    def one(src):
        try:
            try:
                search_main(["--datadir", src, "--outdir", det] + list(flags))
            except SystemExit as e:                                # "Searching: file exists, exit!!": the detected directory is reused
                if "exists" not in str(e):
                    raise
                print(e)
            return None
        except (Exception, SystemExit) as e:                       # noqa: B014
            traceback.print_exc()
            return e
    if threads <= 1 or len(srcs) <= 1:
        return [one(s_) for s_ in srcs]
    import torch
    from concurrent.futures import ThreadPoolExecutor

    def on_stream(src):
        if not torch.cuda.is_available():
            return one(src)
        dev = next((f for i, f in enumerate(flags) if i and flags[i - 1] == "--device"), None)
        dev = torch.device(dev) if dev else torch.device("cuda", torch.cuda.current_device())
        torch.cuda.set_device(dev)
        st = torch.cuda.Stream(dev)
        with torch.cuda.stream(st):
            r = one(src)
        st.synchronize()
        return r
    with ThreadPoolExecutor(min(threads, len(srcs)), thread_name_prefix="npp-search") as pool:
        return list(pool.map(on_stream, srcs))


def _main_stacked(args, mine, det, common, rank, world, search_main):
    """Search every image of this rank, then fit them together (train.main_stacked: groups of up to --stack images of one batch
    shape per launch sequence)."""
    from .train import main_stacked
    failed, t_all, argvs, names = [], time.time(), [], []
    n_thr = args.search_threads if args.search_threads else min(8, max(1, args.stack))
    # (--search-threads given: the round-5 form, one host thread + stream per image; default: the images' candidate fits stacked)
    errors = search_all(mine, det, common + shlex.split(args.search_args), search_main, threads=n_thr, together=args.search_threads is None)
    for src, err in zip(mine, errors):
        name = os.path.basename(os.path.normpath(src))
        if err is not None:
            print(f"[run rank {rank}/{world}] {args.task}/{name}: search FAILED ({type(err).__name__}: {err})", flush=True)
            failed.append(name)
            continue
        argvs.append(["--datadir", os.path.join(det, name), "--basedir", args.basedir, "--p_topk", str(args.p_topk)] + common
                     + (["--task", args.task] if args.task != "completion" else []) + shlex.split(args.train_args))
        names.append(name)
    t1 = time.time()
    def has_testset(n):
        """A finished result: the directory holds at least one written test set (an empty directory is what an aborted earlier run
        left behind -- train.py:42-44 would skip it forever: it counts as failed and is named, not as done)."""
        root = os.path.join(args.basedir, f"{args.task}_top{args.p_topk}", n)
        return os.path.isdir(root) and any(e.startswith("testset_") and os.listdir(os.path.join(root, e)) for e in os.listdir(root)
                                           if os.path.isdir(os.path.join(root, e)))
    fits = main_stacked(argvs, max_stack=args.stack) if argvs else []
    for n, f in zip(names, fits):
        if not has_testset(n):                                   # whether fitted now (f) or found in place (f is None)
            failed.append(n)
    print(f"[run rank {rank}/{world}] {args.task}: {len(names)} images searched in {t1 - t_all:.1f} s, fitted {args.stack}-stacked in "
          f"{time.time() - t1:.1f} s; {len(mine) - len(failed)} of {len(mine)} done" + (f"; failed: {failed}" if failed else ""), flush=True)
    return 1 if failed else 0


if __name__ == "__main__":
    raise SystemExit(main())
