#!/usr/bin/env python3
"""g10d_light_fit.npz: the candidate loop of NPP_proposal/search.py:85-147 executed with the REFERENCE's own modules (build container
only) for TWO candidates in sequence on a small image: per candidate torch.manual_seed(0) / np.random.seed(0), create_npp_net's
construction order (position embedder, is_search periodic embedder, NPP_Net_light(D = 4, W = 256), torch.optim.Adam over the model AND
the adaptive pixel loss), N iterations of np.random.choice rows -> render (sigmoid) -> img2mse(robust_loss_adaptive) -> step -> LR rule.
`adaptive_pix` is ONE module-level object in the reference (models/helpers.py:8), so the second candidate starts from the latents the
first one left (with fresh Adam moments): stored are each candidate's loss per iteration, its latents before and after, and its
rendering of 48 probe coordinates.
    python tests/golden/make_golden_light_fit.py
"""
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from make_golden import import_reference, OUT, FREQ_SCALES, FREQ_OFFSETS, ANGLE_OFFSETS  # noqa: E402

H, W_IMG, N_ITERS, N_RAND, LRATE, LRATE_DECAY = 72, 96, 30, 256, 5e-4, 500


def main():
    R = import_reference()
    emb, nets, msec, adaptive = R["emb"], R["nets"], R["msec"], R["adaptive"]
    import oracle
    img, mask = oracle.synthetic_image(96, noise=0.01)
    img, mask = img[:H, :W_IMG], mask[:H, :W_IMG]
    known = np.ones((H, W_IMG))
    known[20:44, 30:60] = 0                                               # the hole
    masked = (img * known[..., None]).astype(np.float32)
    i_train_np = np.stack(np.nonzero(known), 1)
    masked_img = torch.from_numpy(masked)[None]
    i_train = torch.Tensor(i_train_np)
    cands = [(np.array([0.0, 90.0], np.float32), np.array([12.0, 12.0], np.float32)),
             (np.array([30.0, 120.0], np.float32), np.array([9.0, 15.0], np.float32))]
    rng = np.random.RandomState(5)
    probe = np.stack([rng.randint(0, H, 48), rng.randint(0, W_IMG, 48)], 1)
    adaptive_pix = adaptive.AdaptiveLossFunction(num_dims=3, float_dtype=np.float32, device="cpu")     # helpers.py:8 (module level: shared)
    args = types.SimpleNamespace(normalize_type=1)
    out = {"masked_img": masked, "i_train": i_train_np.astype(np.int32), "probe": probe.astype(np.int32),
           "n_iters": np.int64(N_ITERS), "n_rand": np.int64(N_RAND), "lrate": np.float64(LRATE), "lrate_decay": np.int64(LRATE_DECAY)}
    for ci, (angles, periods) in enumerate(cands):
        torch.manual_seed(0)                                              # search.py:91-92
        np.random.seed(0)
        embedder, freq_nerf = emb.get_embedder(10, 0, (H, W_IMG), is_search=True)              # helpers.py:84
        ep, in_ch_p = emb.get_embedder(10, 0, (H, W_IMG), selected_angles=torch.Tensor(angles), selected_periods=torch.Tensor(periods),
                                       freq_scales=FREQ_SCALES, freq_offsets=FREQ_OFFSETS, angle_offsets=ANGLE_OFFSETS, is_search=True)
        model = nets.NPP_Net_light(D=4, W=256, input_ch=int(freq_nerf), input_ch_periodic=int(in_ch_p), freq_scales=FREQ_SCALES,
                                   freq_offsets=FREQ_OFFSETS, angle_offsets=ANGLE_OFFSETS, output_ch=3, skips=[4], activation="snake")
        grad_vars = list(model.parameters()) + list(adaptive_pix.parameters())                   # helpers.py:144
        optimizer = torch.optim.Adam(params=grad_vars, lr=LRATE, betas=(0.9, 0.999))             # helpers.py:164
        out[f"c{ci}.angles"], out[f"c{ci}.periods"] = angles, periods
        out[f"c{ci}.latents0"] = np.concatenate([p.detach().numpy().reshape(-1) for p in adaptive_pix.parameters()])
        out["freqs"] = np.array([float(fn.__defaults__[1]) for fn in embedder.embed_fns[1::2]], np.float32)
        i_train_emb = embedder.embed(i_train.clone())                      # search.py:104-108
        i_train_emb_periodic = ep.embed(i_train)
        global_step, losses = 0, []
        for i in range(1, N_ITERS + 1):                                    # search.py:111-147
            select_inds = np.random.choice(i_train.shape[0], size=[N_RAND], replace=False)
            select_coords = i_train[select_inds].long()
            gt_rgb = masked_img[0, select_coords[:, 0], select_coords[:, 1], :]
            pred_rgb = torch.sigmoid(model(i_train_emb[select_inds], i_train_emb_periodic[select_inds]))    # helpers.py:41-62
            optimizer.zero_grad()
            loss = msec.img2mse(pred_rgb, gt_rgb, "robust_loss_adaptive", adaptive_pix, None)
            loss.backward()
            optimizer.step()
            new_lrate = LRATE * (0.1 ** (global_step / (LRATE_DECAY * 100)))
            for pg in optimizer.param_groups:
                pg["lr"] = new_lrate
            losses.append(float(loss))
            global_step += 1
        out[f"c{ci}.loss"] = np.array(losses, np.float64)
        out[f"c{ci}.latents1"] = np.concatenate([p.detach().numpy().reshape(-1) for p in adaptive_pix.parameters()])
        with torch.no_grad():
            pc = torch.Tensor(probe.astype(np.float32))
            out[f"c{ci}.probe_pred"] = torch.sigmoid(model(embedder.embed(pc.clone()), ep.embed(pc))).numpy()
        print(f"candidate {ci}: loss {losses[0]:.5f} -> {losses[-1]:.5f}; latents {out[f'c{ci}.latents0']} -> {out[f'c{ci}.latents1']}")
    np.savez(os.path.join(OUT, "g10d_light_fit.npz"), **out)


if __name__ == "__main__":
    main()
