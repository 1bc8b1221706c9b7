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

The random mode (no_reg_sampling=True, sampler.py:66-85,219-228: real patches drawn uniformly from the stride-P/10
windows without unknown pixels) is served from the same table.  With the library's own random stream the whole host half
runs in native code (npp_sampler_draw).
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
        self.no_reg_sampling = bool(no_reg_sampling)
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
        self._native = None
        self._raw_pools = None
        self.reset_patchsize(img, mask, patch_size, N_samples)
        self.reset_pool(pool_train, pool_val)
        self._grid_cache = {}

    def reset_patchsize(self, img, mask, patch_size, N_samples, ratio=0.0):
        self.N_samples = int(N_samples)
        self.patch_size_h_half = self.patch_size_w_half = int(patch_size) // 2
        self.max_shifting_ind = 10
        a, b = np.meshgrid(np.arange(-10, 10), np.arange(-10, 10), indexing="ij")      # sampler.py:90-93
        self._a, self._b = a.reshape(-1), b.reshape(-1)
        self.permute_distance = (np.abs(a) + np.abs(b)).reshape(-1).astype(np.float64)
        if self.no_reg_sampling:
            # sampler.py:66-85: every P x P window at stride P // 10 (unfold order: rows outer, columns inner) WITHOUT an
            # unknown pixel (ratio = 0) is a real-patch candidate of the random mode; kept as window centres
            P, st = 2 * self.patch_size_h_half, max(int(patch_size) // 10, 1)
            ys = np.arange(0, self.height - P + 1, st)
            xs = np.arange(0, self.width - P + 1, st)
            yy, xx = np.meshgrid(ys, xs, indexing="ij")
            known = self.sat[yy + P, xx + P] - self.sat[yy, xx + P] - self.sat[yy + P, xx] + self.sat[yy, xx]
            ok = (P * P - known) <= 0
            self.random_centres = np.stack([yy[ok] + P // 2, xx[ok] + P // 2], 1).astype(np.int64)

    def reset_pool(self, pool_train, pool_val):
        def valid(pool):
            pool = pool.detach().cpu().numpy() if isinstance(pool, torch.Tensor) else np.asarray(pool)
            h, w = self.patch_size_h_half, self.patch_size_w_half
            ok = (pool[:, 0] > h) & (pool[:, 0] < self.height - (h + 1)) & (pool[:, 1] > w) & (pool[:, 1] < self.width - (w + 1))
            return pool[ok].astype(np.int64)
        self.pool_train, self.pool_val = valid(pool_train), valid(pool_val)
        self._setup_native(pool_train, pool_val)

    def _setup_native(self, pool_train, pool_val):
        """The host half in native code (csrc/npp_host_rng.hip: npp_sampler_*) when the random stream is the library's own
        MT19937 (host_rng.NativeRandomState): one GIL-free call per iteration instead of ~40 NumPy calls."""
        from .host_rng import NativeRandomState
        if not isinstance(self.rng, NativeRandomState) or self.fast_rng is not None or self.no_reg_sampling:
            return
        import ctypes as C
        from ._lib import lib, check
        L = lib()
        if self._native is None:
            def raw(pool):
                pool = pool.detach().cpu().numpy() if isinstance(pool, torch.Tensor) else np.asarray(pool)
                return np.ascontiguousarray(pool, np.int32)
            pt, pv = raw(pool_train), raw(pool_val)
            sh = np.ascontiguousarray(np.stack(self.selected_shifts), np.float64)
            h = L.npp_sampler_create(self.sat.ctypes.data_as(C.c_void_p), self.height, self.width, pt.ctypes.data_as(C.c_void_p), pt.shape[0],
                                     pv.ctypes.data_as(C.c_void_p), pv.shape[0], sh.ctypes.data_as(C.c_void_p))
            if not h:
                raise MemoryError("npp_sampler_create")
            self._native = C.c_void_p(h)
        nt, nv = C.c_int64(0), C.c_int64(0)
        check(L.npp_sampler_set_patch(self._native, 2 * self.patch_size_h_half, self.N_samples, C.byref(nt), C.byref(nv)),
              "npp_sampler_set_patch")
        assert (nt.value, nv.value) == (self.pool_train.shape[0], self.pool_val.shape[0])

    def __del__(self):
        try:
            if getattr(self, "_native", None):
                from ._lib import lib
                lib().npp_sampler_destroy(self._native)
                self._native = None
        except Exception:
            pass

    def _draw_native(self, topk, invalid_ratio):
        import ctypes as C
        from ._lib import lib, check
        n = self.N_samples
        src, k = C.c_int32(0), C.c_int32(0)
        cen = np.empty((n, 2), np.int32)
        real = np.empty((n, topk, 2), np.float64)
        w = np.empty((n, topk), np.float32)
        check(lib().npp_sampler_draw(self._native, self.rng._h, int(topk), float(invalid_ratio), C.byref(src), C.byref(k),
                                     cen.ctypes.data_as(C.c_void_p), real.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p)),
              "npp_sampler_draw")
        source, k = ("val", "train", "same")[src.value], int(k.value)
        d = dict(source=source, cen=cen.astype(np.int64), P=2 * self.patch_size_h_half, n=n, k=k)
        if source == "same":
            d.update(real_cen=None, weights=np.ones(n, np.float32))
        elif k == 0:
            d.update(real_cen=None, weights=None)
        else:
            d.update(real_cen=np.ascontiguousarray(real[:, :k]).reshape(n * k, 2), weights=np.ascontiguousarray(w[:, :k]).reshape(-1))
        return d

    # ---- helpers ---------------------------------------------------------------------
    def _unknown_count(self, cen):
        """#pixels with mask < 0.5 in the P x P window of each centre, out-of-image = unknown."""
        P = 2 * self.patch_size_h_half
        c = np.rint(cen).astype(np.int64)
        y0 = np.clip(c[:, 0] - P // 2, 0, self.height); y1 = np.clip(c[:, 0] + P // 2, 0, self.height)
        x0 = np.clip(c[:, 1] - P // 2, 0, self.width); x1 = np.clip(c[:, 1] + P // 2, 0, self.width)
        known = self.sat[y1, x1] - self.sat[y0, x1] - self.sat[y1, x0] + self.sat[y0, x0]
        return P * P - known

    def _gather(self, cen, P=None):
        c = ops.h2d(np.rint(cen).astype(np.int32), self.device)
        return ops.patch_gather(self.img, self.mask, c, P if P is not None else 2 * self.patch_size_h_half)

    # ---- host half: everything that consumes random numbers or walks the lattice (NumPy / native RNG only, no GPU) ----
    def draw(self, topk, invalid_ratio):
        """The host-side decisions of one sample_patches() call, in the reference's RNG order (sampler.py:324 uniform,
        :260 choice): patch source, fake-patch centres, and per fake patch the k best lattice candidates with their 1/d
        weights.  Returns a dict of NumPy arrays (k == 0: no valid real patch -> the iteration is skipped, train.py:160-161).
        Touches no device memory, so it can run ahead of the training loop on another thread."""
        if self._native is not None:
            return self._draw_native(topk, invalid_ratio)
        prob = self.rng.uniform(0, 1)
        if prob < 0.5:
            source = "val"
        elif 0.5 < prob < 0.8:
            source = "train"
        else:
            source = "same"
        pool = self.pool_val if source == "val" else self.pool_train
        if self.fast_rng is not None:
            sel = self.fast_rng.choice(pool.shape[0], size=self.N_samples, replace=False)
        else:
            sel = self.rng.choice(pool.shape[0], size=[self.N_samples], replace=False)
        cen = pool[sel]
        h = self.patch_size_h_half
        d = dict(source=source, cen=cen, P=2 * h, n=self.N_samples)
        if source == "same":
            d.update(k=1, real_cen=None, weights=np.ones(self.N_samples, np.float32))
            return d
        P = 2 * h
        if self.no_reg_sampling:
            # sampler.py:219-228: N_samples * topk windows, uniformly without replacement, no weights, k = topk
            if self.random_centres.shape[0] < self.N_samples * topk:
                d.update(k=0, real_cen=None, weights=None)
                return d
            rs = self.rng.choice(self.random_centres.shape[0], size=[self.N_samples * topk], replace=False)
            d.update(k=topk, real_cen=self.random_centres[rs].astype(np.float64), weights=None)
            return d
        chosen, weights = [], []
        topk_min = topk
        for i in range(self.N_samples):
            cand = cen[i][None].astype(np.float64) + self._a[:, None] * self.selected_shifts[0][None] \
                + self._b[:, None] * self.selected_shifts[1][None]
            ok = (cand[:, 0] > 0) & (cand[:, 0] < self.height - 1) & (cand[:, 1] > 0) & (cand[:, 1] < self.width - 1)
            cand, dist = cand[ok], self.permute_distance[ok]
            good = ~(self._unknown_count(cand) > P * P * invalid_ratio)                      # sampler.py:181
            cand, dist = cand[good], dist[good].copy()
            dist[dist == 0] = 10000                                                          # :197
            if min(len(dist) - 1, topk) < topk_min:
                topk_min = min(len(dist) - 1, topk)
                if topk_min <= 0:
                    d.update(k=0, real_cen=None, weights=None)
                    return d
            # torch.topk(largest=False): ties are backend-defined; here: stable order
            order = np.argsort(dist, kind="stable")[:topk_min]
            inv = 1.0 / dist[order]
            weights.append((inv / inv.sum()).astype(np.float32))
            chosen.append(cand[order])
        if topk_min < topk:
            weights = [w[:topk_min] for w in weights]
            chosen = [c[:topk_min] for c in chosen]
        d.update(k=topk_min, real_cen=np.concatenate(chosen, 0), weights=np.concatenate(weights))
        return d

    # ---- device half: crops of exactly the patches that are returned (npp_patch_gather) ----------------------------
    @staticmethod
    def centres_i32(d):
        """All patch centres of a draw (fake, then real) as the int32 (row, col) array the gather kernel takes."""
        return np.rint(d["cen"] if d["real_cen"] is None else np.concatenate([d["cen"], d["real_cen"]], 0)).astype(np.int32)

    def materialise(self, d, want_coords=True, want_tuple=True, cen_dev=None, crops_out=None):
        """-> the reference's 8-tuple (sampler.py:297-354) from a draw(); also sets self.last_raw (contiguous crops for
        the fused plumbing kernels).  want_tuple=False: only last_raw, source, k and the weights are produced (entries 0..3 None)."""
        if d["k"] == 0:
            return None, None, None, None, None, None, 0, None
        n, k, P = d["n"], d["k"], d["P"]
        # ONE host -> device transfer of all centres (fake, then real) and ONE gather launch for all crops
        # (cen_dev: the caller uploaded centres_i32(d) itself, together with its other per-iteration indices)
        c_dev = cen_dev if cen_dev is not None else ops.h2d(self.centres_i32(d), self.device)
        self.last_cen_dev = c_dev[:n]
        rgb_all, m_all = ops.patch_gather(self.img, self.mask, c_dev, P, out=crops_out)     # crops_out: a stacked fit's buffers
        fake, fmask = rgb_all[:n], m_all[:n]
        # fake_coords (n,P,P,2) of the 8-tuple (sampler.py:269-279); the fused loop builds its input rows from the centres
        # (npp_batch_assemble) and skips it
        coords = self._coords(self.last_cen_dev, P) if want_coords else None
        real = rmask = None
        if d["source"] == "same":
            if want_tuple:
                real, rmask = fake.permute(0, 2, 3, 1)[:, None].clone(), fmask.permute(0, 2, 3, 1)[:, None].clone()
            raw_real = (fake, fmask)
        else:
            rgb, m = rgb_all[n:], m_all[n:]
            raw_real = (rgb, m)                                                              # contiguous (n*k,3,P,P), (n*k,1,P,P)
            if want_tuple:
                real = rgb.reshape(n, k, 3, P, P).permute(0, 1, 3, 4, 2)                     # (n,k,P,P,3)
                rmask = m.reshape(n, k, 1, P, P).permute(0, 1, 3, 4, 2)
        weight = None if (d["weights"] is None or not want_tuple) else ops.h2d(d["weights"], self.device)   # random mode: none (:228)
        self.last_raw = dict(fake=fake, fmask=fmask, real=raw_real[0], rmask=raw_real[1])
        fake_t = fake[:, None].tile([1, k, 1, 1, 1]) if want_tuple else None
        fmask_t = fmask[:, None].tile([1, k, 1, 1, 1]) if want_tuple else None
        self.last_centres = d["cen"]
        return real, rmask, fake_t, fmask_t, coords, d["source"], k, weight

    def _coords(self, cen_dev, P):
        """Pixel coordinates (row, col) of the fake patches, built on the device from the centres: window [c - P/2, c + P/2)."""
        base = self._grid_cache.get(P)
        if base is None:
            r = torch.arange(-(P // 2), P // 2, dtype=torch.int64, device=self.device)
            base = self._grid_cache[P] = torch.stack(torch.meshgrid(r, r, indexing="ij"), -1)            # (P,P,2)
        return cen_dev.to(torch.int64)[:, None, None, :] + base[None]

    # ---- reference API ---------------------------------------------------------------
    def sample_patches(self, topk, invalid_ratio):
        """-> (real_patch (n,k,P,P,3), real_mask (n,k,P,P,1), fake_patch (n,k,3,P,P), fake_mask (n,k,1,P,P),
        fake_coords (n,P,P,2), patch_source, k, weight)   (sampler.py:297-354)"""
        return self.materialise(self.draw(topk, invalid_ratio))
