"""Round 6: where the wall time of the 8-image periodicity search goes (search.main_multi's phases, synchronised between them)."""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from npp_amd import io as nio, synthetic as syn, search, light  # noqa: E402

S, M = 512, 8
tmp = tempfile.mkdtemp()
srcs = []
for i in range(M):
    im, mk = syn.synthetic_image(S, seed=10 + i)
    srcs.append(nio.write_detected_dir(os.path.join(tmp, "input", f"img{i}"), im, mk, np.ones_like(mk), [[0, 0]], [[1, 1]], [[[1, 0], [0, 1]]]))


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


for rep in range(2):
    det = os.path.join(tmp, f"det{rep}")
    T = {}
    prepared = []
    for s in srcs:
        t0 = sync()
        args = search.parse(["--datadir", s, "--outdir", det, "--device", "cuda:0", "--random-trunks"])
        out, imgs, conv1, trunks = search._load(args)
        t1 = sync()
        from npp_amd import proposal
        m2 = np.asarray(imgs[2], np.float64).reshape(imgs[0].shape[:2])
        v2 = np.asarray(imgs[3], np.float64).reshape(imgs[0].shape[:2])
        proposal.search_periodicity_by_feat(np.uint8(np.asarray(imgs[0].astype(np.float32)) * 255), np.uint8(v2 * m2), repeat_range=tuple(args.search_range),
                                            edge_searching=args.edge_searching, gray_only=args.gray_only, conv1=conv1, device=args.device)
        t2 = sync()
        cands, ranker = search.prepare_image(imgs[0].astype(np.float32), imgs[2], imgs[3], args, conv1, trunks)
        t3 = sync()
        T["load"] = T.get("load", 0) + t1 - t0
        T["displacement search"] = T.get("displacement search", 0) + t2 - t1
        T["ranker build"] = T.get("ranker build", 0) + (t3 - t2) - (t2 - t1)
        prepared.append((args, out, imgs, cands, ranker))
    # rank_images with its inner phases timed through monkeypatching
    t_step = [0.0]
    t_score = [0.0]
    orig_step = light.NPPNetLightBatch.train_step
    orig_score = light.ProposalRanker.score

    def score(self, net):
        t0 = sync()
        r = orig_score(self, net)
        t_score[0] += sync() - t0
        return r
    light.ProposalRanker.score = score
    t_init, t_embed, t_draw = [0.0], [0.0], [0.0]
    orig_init, orig_embed, orig_draws = light.NPPNetLightBatch.__init__, light.NPPNetLight.embed, light.ProposalRanker._pixel_draws

    def init(self, *a, **k):
        t0 = sync()
        orig_init(self, *a, **k)
        t_init[0] += sync() - t0

    def embed(self, *a, **k):
        t0 = sync()
        r = orig_embed(self, *a, **k)
        t_embed[0] += sync() - t0
        return r

    def pdraws(self):
        t0 = sync()
        r = orig_draws(self)
        t_draw[0] += sync() - t0
        return r
    light.NPPNetLightBatch.__init__, light.NPPNetLight.embed, light.ProposalRanker._pixel_draws = init, embed, pdraws
    t0 = sync()
    ranked = light.rank_images([p[4] for p in prepared], [p[3] for p in prepared], topk=10)
    t1 = sync()
    light.ProposalRanker.score = orig_score
    light.NPPNetLightBatch.__init__, light.NPPNetLight.embed, light.ProposalRanker._pixel_draws = orig_init, orig_embed, orig_draws
    T["rank_images total"] = t1 - t0
    T["  of which score()"] = t_score[0]
    T["  of which batch construction"] = t_init[0]
    T["  of which lattice / positional tables"] = t_embed[0]
    T["  of which pixel draws (host RNG + upload)"] = t_draw[0]
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(8) as pool:
        list(pool.map(lambda pr: search._write(pr[0][0], pr[0][1], pr[0][2], search._ranked(pr[0][3], *pr[1])), zip(prepared, ranked)))
    T["write (png drawings + odgt)"] = sync() - t1
    print(f"--- pass {rep}: {M} images, {len(prepared[0][3])} candidates each")
    for k, v in T.items():
        print(f"{k:32s} {v * 1e3:8.1f} ms")
    print(f"{'sum':32s} {(T['load'] + T['displacement search'] + T['ranker build'] + T['rank_images total'] + T['write (png drawings + odgt)']) * 1e3:8.1f} ms")
