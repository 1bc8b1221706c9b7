"""GridPatchSampler -- host-side mirror of models/sampler.py:8-354 (periodicity-guided mode).

Same constructor, same methods, same 8-tuple from sample_patches(), same NumPy RNG call order
(np.random.uniform, then np.random.choice), so it drops into the loop of
NPP_completion/train.py:152-157.  What changed is how the work is done:

* the reference crops EVERY lattice candidate (up to 400 per fake patch) by tiling the whole
  image once per candidate and running grid_sample, only to count unknown pixels
  (sampler.py:171-181); here the count comes from a summed-area table of the known mask
  (4 look-ups per candidate, zero padding counted as unknown), on the host;
* only the patches that are returned are cropped, by the HIP gather kernel
  (npp_patch_gather == extract_glimpse with nearest / zeros padding).

The random (no_reg_sampling=True) mode of the reference (sampler.py:219-228) is not built.
"""
import numpy as np
import torch

from . import ops


class GridPatchSampler:
    def __init__(self, img, mask, N_samples, patch_size, height, width, pool_train, pool_val, selected_shifts,
                 no_reg_sampling=False, rng=None, fast_rng=None):
        """img (1,H,W,3), mask (1,H,W,1) tensors (sampler.py:29); pools (n,2) (row, col);
        selected_shifts[0] = [(dx, dy), (dx, dy)] (flipped to (dy, dx) at sampler.py:35).
        rng: a np.random.RandomState, default the global np.random like the reference.
        fast_rng: optional np.random.Generator; when given, the without-replacement draw of the fake-patch centres
        uses Generator.choice (O(size)) instead of np.random.choice(replace=False), which permutes the WHOLE pool
        (0.7 ms for the 100 k-pixel train pool).  Same distribution, NOT the reference's random stream."""
        if no_reg_sampling:
            raise NotImplementedError("random patch sampling (no_reg_sampling) is outside the built path")
        self.rng = rng if rng is not None else np.random
        self.fast_rng = fast_rng
        self.height, self.width = int(height), int(width)
        self.img = img[0].contiguous().float()                       # (H,W,3) on the device
        self.mask = mask[0, ..., 0].contiguous().float()             # (H,W)
        self.device = self.img.device
        s = selected_shifts[0]
        self.selected_shifts = [np.array([sh[1], sh[0]], np.float64) for sh in s]
        known = (self.mask.detach().cpu().numpy() >= 0.5).astype(np.int64)
        self.sat = np.zeros((self.height + 1, self.width + 1), np.int64)     # summed-area table of known pixels
        self.sat[1:, 1:] = known.cumsum(0).cumsum(1)
        self.reset_patchsize(img, mask, patch_size, N_samples)
        self.reset_pool(pool_train, pool_val)

    def reset_patchsize(self, img, mask, patch_size, N_samples, ratio=0.0):
        self.N_samples = int(N_samples)
        self.patch_size_h_half = self.patch_size_w_half = int(patch_size) // 2
        self.max_shifting_ind = 10
        a, b = np.meshgrid(np.arange(-10, 10), np.arange(-10, 10), indexing="ij")      # sampler.py:90-93
        self._a, self._b = a.reshape(-1), b.reshape(-1)
        self.permute_distance = (np.abs(a) + np.abs(b)).reshape(-1).astype(np.float64)

    def reset_pool(self, pool_train, pool_val):
        def valid(pool):
            pool = pool.detach().cpu().numpy() if isinstance(pool, torch.Tensor) else np.asarray(pool)
            h, w = self.patch_size_h_half, self.patch_size_w_half
            ok = (pool[:, 0] > h) & (pool[:, 0] < self.height - (h + 1)) & (pool[:, 1] > w) & (pool[:, 1] < self.width - (w + 1))
            return pool[ok].astype(np.int64)
        self.pool_train, self.pool_val = valid(pool_train), valid(pool_val)

    # ---- helpers ---------------------------------------------------------------------
    def _unknown_count(self, cen):
        """#pixels with mask < 0.5 in the P x P window of each centre, out-of-image = unknown."""
        P = 2 * self.patch_size_h_half
        c = np.rint(cen).astype(np.int64)
        y0 = np.clip(c[:, 0] - P // 2, 0, self.height); y1 = np.clip(c[:, 0] + P // 2, 0, self.height)
        x0 = np.clip(c[:, 1] - P // 2, 0, self.width); x1 = np.clip(c[:, 1] + P // 2, 0, self.width)
        known = self.sat[y1, x1] - self.sat[y0, x1] - self.sat[y1, x0] + self.sat[y0, x0]
        return P * P - known

    def _gather(self, cen):
        c = ops.h2d(np.rint(cen).astype(np.int32), self.device)
        return ops.patch_gather(self.img, self.mask, c, 2 * self.patch_size_h_half)

    # ---- reference API ---------------------------------------------------------------
    def sample_patch_fake(self, mode):
        pool = self.pool_train if mode == "train" else self.pool_val
        if self.fast_rng is not None:
            sel = self.fast_rng.choice(pool.shape[0], size=self.N_samples, replace=False)
        else:
            sel = self.rng.choice(pool.shape[0], size=[self.N_samples], replace=False)
        cen = pool[sel]
        h = self.patch_size_h_half
        yy = cen[:, 0, None, None] + np.arange(-h, h)[None, :, None]
        xx = cen[:, 1, None, None] + np.arange(-h, h)[None, None, :]
        grids = np.stack(np.broadcast_arrays(yy, xx), -1)                                   # (n,P,P,2)
        patch, pmask = self._gather(cen)
        return patch, pmask, ops.h2d(grids.astype(np.int64), self.device), cen

    def sample_patch_real(self, centres, topk=5, invalid_ratio=0.3):
        P = 2 * self.patch_size_h_half
        chosen, weights = [], []
        topk_min = topk
        for i in range(self.N_samples):
            cand = centres[i][None].astype(np.float64) + self._a[:, None] * self.selected_shifts[0][None] \
                + self._b[:, None] * self.selected_shifts[1][None]
            ok = (cand[:, 0] > 0) & (cand[:, 0] < self.height - 1) & (cand[:, 1] > 0) & (cand[:, 1] < self.width - 1)
            cand, dist = cand[ok], self.permute_distance[ok]
            good = ~(self._unknown_count(cand) > P * P * invalid_ratio)                      # sampler.py:181
            cand, dist = cand[good], dist[good].copy()
            dist[dist == 0] = 10000                                                          # :197
            if min(len(dist) - 1, topk) < topk_min:
                topk_min = min(len(dist) - 1, topk)
                if topk_min <= 0:
                    return None, None, None, 0
            # torch.topk(largest=False): ties are backend-defined; here: stable order
            order = np.argsort(dist, kind="stable")[:topk_min]
            inv = 1.0 / dist[order]
            weights.append((inv / inv.sum()).astype(np.float32))
            chosen.append(cand[order])
        if topk_min < topk:
            weights = [w[:topk_min] for w in weights]
            chosen = [c[:topk_min] for c in chosen]
        rgb, m = self._gather(np.concatenate(chosen, 0))
        n, k = self.N_samples, topk_min
        self._raw_real = (rgb, m)                                                            # contiguous (n*k,3,P,P), (n*k,1,P,P)
        rgb = rgb.reshape(n, k, 3, P, P).permute(0, 1, 3, 4, 2)                              # (n,k,P,P,3)
        m = m.reshape(n, k, 1, P, P).permute(0, 1, 3, 4, 2)
        return rgb, m, ops.h2d(np.concatenate(weights), self.device), topk_min

    def sample_patches(self, topk, invalid_ratio):
        """-> (real_patch (n,k,P,P,3), real_mask (n,k,P,P,1), fake_patch (n,k,3,P,P), fake_mask (n,k,1,P,P),
        fake_coords (n,P,P,2), patch_source, k, weight)   (sampler.py:297-354)"""
        prob = self.rng.uniform(0, 1)
        if prob < 0.5:
            source = "val"
        elif 0.5 < prob < 0.8:
            source = "train"
        else:
            source = "same"
        fake, fmask, coords, cen = self.sample_patch_fake("val" if source == "val" else "train")
        if source == "same":
            real, rmask = fake.permute(0, 2, 3, 1)[:, None].clone(), fmask.permute(0, 2, 3, 1)[:, None].clone()
            self._raw_real = (fake, fmask)
            k = 1
            weight = torch.ones(self.N_samples, dtype=torch.float32, device=self.device)
        else:
            real, rmask, weight, k = self.sample_patch_real(cen, topk=topk, invalid_ratio=invalid_ratio)
        if k == 0:
            return None, None, None, None, None, None, 0, None
        # contiguous channel-first crops for the fused plumbing kernels (npp_patch_compose_*): the untiled fake
        # patch / mask and the k real patches per fake patch
        self.last_raw = dict(fake=fake, fmask=fmask, real=self._raw_real[0], rmask=self._raw_real[1])
        fake = fake[:, None].tile([1, k, 1, 1, 1])
        fmask = fmask[:, None].tile([1, k, 1, 1, 1])
        self.last_centres = cen
        return real, rmask, fake, fmask, coords, source, k, weight
