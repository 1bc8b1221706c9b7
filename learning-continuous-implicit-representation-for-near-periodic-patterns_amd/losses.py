"""Patch losses of the loop: ContextualLoss (externel_lib/contextual_loss/modules/contextual.py:
9-68) and LPIPS(net='vgg') (externel_lib/lpips/lpips.py:27-133), same call signatures.

From the feature tensors onward everything runs in libnpp_hip.so (npp_cx_fwd_bwd,
npp_lpips_layer); they are wired into torch.autograd with two small Functions so that the
modules remain drop-ins for the reference's (`loss.backward()` keeps working).

The VGG19[0:18] / VGG16 trunks are frozen third-party convolution stacks whose pretrained
torchvision weights are not available offline: they are built here layer by
layer (no torchvision import) and take either a torchvision-format state_dict supplied by the
user or a fixed-seed random init.  They run in libnpp_hip.so as well (HipTrunk: npp_conv3x3 /
npp_maxpool2_* on flat padded bf16 tensors, forward and data gradient); `_Trunk` is the same
stack through torch.nn (fp32, MIOpen) and is kept only as the comparator of the GPU tests.
"""
import numpy as np
import os

import torch
import torch.nn as nn

from . import ops
from ._lib import lib

_VGG19 = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256]                       # features[0:18] -> relu3_4
_VGG16 = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512]


def _make_features(cfg):
    layers, cin = [], 3
    for v in cfg:
        if v == "M":
            layers.append(nn.MaxPool2d(2, 2))
        else:
            layers += [nn.Conv2d(cin, v, 3, padding=1), nn.ReLU(inplace=False)]
            cin = v
    return nn.Sequential(*layers)


_warned_random = set()


def _warn_random_trunk(cfg, seed):
    """No pretrained weights were supplied: the trunk keeps its fixed-seed random init.  That is what the synthetic bench and
    the parity tests use (the reference's torchvision weights are not available offline); for a real fit it
    changes the contextual / LPIPS / style losses, so say so once per trunk shape."""
    key = (tuple(cfg), seed)
    if key not in _warned_random:
        _warned_random.add(key)
        import warnings
        warnings.warn(f"npp_amd.losses: VGG trunk {len([v for v in cfg if v != 'M'])} convs built with fixed-seed RANDOM weights "
                      f"(seed {seed}): pass the torchvision state_dict (vgg_state_dict=...) to reproduce the reference's losses",
                      stacklevel=3)


def _trunk_keys(state_dict, features):
    """A torchvision VGG state_dict restricted to this trunk's layers, as keys of `features` ('<idx>.weight' / '<idx>.bias').
    Accepts the model's form ('features.<idx>.*', classifier keys ignored) and the form saved from `vgg.features` ('<idx>.*').
    Raises if a convolution of the trunk is missing or has another shape -- never leaves a layer at its random init."""
    want = features.state_dict()
    out, missing = {}, []
    for k, ref in want.items():
        v = state_dict.get("features." + k, state_dict.get(k))
        if v is None:
            missing.append(k)
        elif tuple(v.shape) != tuple(ref.shape):
            raise ValueError(f"VGG state_dict: '{k}' has shape {tuple(v.shape)}, this trunk needs {tuple(ref.shape)}")
        else:
            out[k] = v
    if missing:
        raise KeyError(f"VGG state_dict lacks {missing[:4]}{'...' if len(missing) > 4 else ''} (expected torchvision keys "
                       f"'features.<idx>.weight/bias' or '<idx>.weight/bias')")
    return out


class _Trunk(nn.Module):
    """The trunk's LAYERS and weights (torchvision `features` construction order, the user's state dict or the fixed-seed init):
    what HipTrunk packs for the HIP kernels.  It has no forward of its own here -- the torch fp32 forward that the tests and the
    golden generators compare against lives in tests/comparators.py (TorchTrunk)."""

    def __init__(self, cfg, taps, state_dict=None, seed=1234):
        super().__init__()
        with ops.RNG_LOCK:
            g = torch.random.get_rng_state()
            torch.manual_seed(seed)
            self.features = _make_features(cfg)
            torch.random.set_rng_state(g)
        if state_dict is not None:
            self.features.load_state_dict(_trunk_keys(state_dict, self.features), strict=True)
        else:
            _warn_random_trunk(cfg, seed)
        self.taps = taps
        for p in self.parameters():
            p.requires_grad = False     # vgg.py:26-28, pretrained_networks.py:116-118


class _HipTrunkFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, trunk, n_grad, scale, shift):
        taps = trunk._forward(x.contiguous(), scale, shift, n_keep=n_grad)
        ctx.trunk, ctx.n_grad, ctx.scale, ctx.gen, ctx.shape = trunk, n_grad, scale, trunk._gen, tuple(x.shape)
        return tuple(taps)

    @staticmethod
    def backward(ctx, *gt):
        t = ctx.trunk
        if t._gen != ctx.gen:
            raise RuntimeError("HipTrunk: backward after a newer forward of the same trunk (its activation buffers are "
                               "reused between calls; run forward -> backward pairwise)")
        n = ctx.n_grad
        g = [None if (v is None or n == 0) else v[:n].contiguous() for v in gt]
        return t._backward(g, n, ctx.scale, ctx.shape), None, None, None, None


class HipTrunk:
    """conv3x3+ReLU / MaxPool2d(2,2) stack (torchvision `features` indexing) on the HIP kernels of
    csrc/npp_conv.hip.  __call__(x (N,3,H,W), n_grad, scale, shift) -> list of fp32 (N,C,h,w) taps of
    relu outputs; the first n_grad images carry a gradient back to x, the rest are constants (the
    reference runs those under torch.no_grad(): contextual.py:63-64, frozen weights vgg.py:26-28)."""

    # The weights are frozen (vgg.py:26-28): every trunk of one (layer list, weights, device) shares ONE set of device weights and MFMA
    # packs -- the fits of a directory run each build three trunks (8 images: 24 random initialisations / state-dict copies + packs).
    # key -> (layers, state_dict): the state dict is kept alive so that its id stays unique.
    _packs, _packs_lock = {}, __import__("threading").Lock()

    def __init__(self, cfg, taps, state_dict=None, seed=1234, device="cuda"):
        self.device = torch.device(device)
        self.taps = tuple(taps)
        key = (tuple(cfg), ("weights", id(state_dict)) if state_dict is not None else ("seed", int(seed)), str(self.device))
        with HipTrunk._packs_lock:
            hit = HipTrunk._packs.get(key)
            if hit is not None and hit[1] is not state_dict:
                hit = None
            if hit is None:
                ref = _Trunk(cfg, taps, state_dict, seed)          # same layer construction / init as the comparator
                layers = []                                        # ("conv", feat_idx_of_relu, cin, cout, w, b, pf, pb) | ("pool",)
                for i, m in enumerate(ref.features):
                    if isinstance(m, nn.Conv2d):
                        w = m.weight.detach().to(self.device, torch.float32).contiguous()
                        b = m.bias.detach().to(self.device, torch.float32).contiguous()
                        pf, pb = ops.conv_pack(w, in_natural=(len(layers) == 0))
                        layers.append(dict(kind="conv", relu_idx=i + 1, cin=w.shape[1], cout=w.shape[0], w=w, b=b, pf=pf, pb=pb))
                    elif isinstance(m, nn.MaxPool2d):
                        layers.append(dict(kind="pool", idx=i))
                # The uploads and npp_conv_pack launches above ran on THIS thread's current stream; other threads (run.search_all:
                # one stream per host thread) take the entry from the cache and launch on theirs with no event between the two.
                # Publish only what has completed (once per weight set and process).
                torch.cuda.current_stream(self.device).synchronize()
                if len(HipTrunk._packs) >= 12:
                    HipTrunk._packs.pop(next(iter(HipTrunk._packs)))          # evict the oldest entry only (instances keep their own references)
                hit = HipTrunk._packs[key] = (layers, state_dict)
        self.layers = [dict(L) for L in hit[0]]                    # (own dicts: the per-instance tap flags below; the tensors are shared)
        for j, L in enumerate(self.layers):                         # gradient taps are supported on the top layer and before pools
            if L["kind"] == "conv" and L["relu_idx"] in self.taps:
                nxt = self.layers[j + 1]["kind"] if j + 1 < len(self.layers) else None
                L["tap_ok"] = nxt in (None, "pool")
        self._buf, self._gen = {}, 0
        self.final_next_pack = None
        self.prefetch_next = True      # next layer's weights requested into L2 (npp_conv3x3_pf)
        # comparator switches of tests/test_gpu_trunk.py (bit-identical forms): the max-pool backward in the data-gradient launch above
        # it, the max-pool forward in the epilogue of the layer below it (layers with >= fold_pool_fwd_min_cin input channels: measured)
        self.fold_pool_bwd, self.fold_pool_fwd, self.fold_pool_fwd_min_cin = True, True, 128
        # conv a -> conv b -> pool of the first blocks as ONE launch (ops.conv_pair_fwd; round 5): the intermediate activation
        # stays in LDS, the layer outputs are stored only for the images that carry a gradient
        # (bit 0: the first block, 3 -> 64 -> 64; bit 1: the second, 64 -> 128 -> 128; bit 2: the first block's data gradient;
        #  bit 3: the loop's patch plumbing + pixel loss composed inside the first block's launch)
        self.fuse_pairs = int(ops.tune("conv_pair"))
        self._n_keep = None

    def _pb_below(self, j):
        """The backward pack of the next convolution layer below layer j (what the data-gradient pass runs next)."""
        if not self.prefetch_next:
            return None
        return next((self.layers[i]["pb"] for i in range(j - 1, -1, -1) if self.layers[i]["kind"] == "conv"), None)

    def _flat(self, tag, N, C, H, W):
        key = (tag, N, C, H, W)
        t = self._buf.get(key)
        if t is None:
            t = self._buf[key] = ops.trunk_alloc(N, C, H, W, self.device)
        return t

    def __call__(self, x, n_grad=0, scale=(1.0, 1.0, 1.0), shift=(0.0, 0.0, 0.0)):
        need = n_grad > 0 and x.requires_grad
        return list(_HipTrunkFunction.apply(x, self, n_grad if need else 0, tuple(scale), tuple(shift)))

    def input_buffer(self, N, H, W):
        """The flat C=16 input tensor of an (N,3,H,W) batch, for producers that write it directly (ops.trunk_patch_in)."""
        return self._flat("x0", N, 16, H, W)

    def can_compose_input(self, H, W):
        """Whether _forward(x0_src=...) can compose the patch batch inside the first block's fused launch (ops.conv_pair_fwd_patch)."""
        Ls = self.layers
        return bool((int(self.fuse_pairs) & 8) and (int(self.fuse_pairs) & 1) and len(Ls) >= 3 and Ls[0]["kind"] == "conv"
                    and Ls[1]["kind"] == "conv" and Ls[2]["kind"] == "pool" and Ls[0]["relu_idx"] not in self.taps
                    and Ls[2]["idx"] not in self.taps and H % 2 == 0 and W % 2 == 0 and (Ls[0]["cout"], Ls[1]["cout"]) == (64, 64))

    def _forward(self, x, scale, shift, x0_ready=False, n_run=None, n_keep=None, x0_src=None):
        """x: the (N,3,H,W) batch, or just its shape when the flat input was already written (x0_ready).  n_run: only the leading
        n_run images are computed (N fixes the buffers' geometry; rows >= n_run of the returned taps are undefined).  n_keep: the
        leading images a _backward() may follow for (None: all): a fused layer pair stores its layer outputs only for those."""
        N, _, H, W = x if x0_ready else x.shape
        nr = N if n_run is None else int(n_run)
        self._n_keep = nr if n_keep is None else min(int(n_keep), nr)
        self._gen += 1
        self._geom = []
        skip = 0
        if x0_src is not None and not (x0_ready and nr == N and self.can_compose_input(H, W)):
            raise ValueError("HipTrunk._forward(x0_src=...): the first block is not run as a fused pair here (can_compose_input)")
        cur = self._flat("x0", N, 16, H, W)
        if not x0_ready:
            ops.trunk_image_in(x, scale, shift, cur)
        c, outs, pooled = 16, [], None
        for j, L in enumerate(self.layers):
            if skip:                                                  # layers j-1 .. of a fused pair: already run
                skip -= 1
                continue
            Lb = self.layers[j + 1] if j + 2 < len(self.layers) else None
            if ((int(self.fuse_pairs) & (1 if c == 16 else 2)) and L["kind"] == "conv" and Lb is not None and Lb["kind"] == "conv" and self.layers[j + 2]["kind"] == "pool"
                    and L["relu_idx"] not in self.taps and self.layers[j + 2]["idx"] not in self.taps
                    and ops.conv_pair_fwd_ok(H, W, c, L["cout"], Lb["cout"])):
                ya = self._flat(("a", j), N, L["cout"], H, W)
                yb = self._flat(("a", j + 1), N, Lb["cout"], H, W)
                yp = self._flat(("a", j + 2), N, Lb["cout"], H // 2, W // 2)
                tap = None
                if Lb["relu_idx"] in self.taps:
                    tap = torch.empty((N, Lb["cout"], H, W), dtype=torch.float32, device=self.device)
                    outs.append(tap)
                if x0_src is not None and j == 0:
                    # the flat input was NOT written: the pair composes the patches itself (and runs the pixel loss in its last blocks)
                    ops.conv_pair_fwd_patch(*x0_src["patch"], scale, shift, x0_src.get("zero"), x0_src.get("loss"), self._n_keep,
                                            L["cout"], Lb["cout"], L["pf"], L["b"], Lb["pf"], Lb["b"], ya, yb, yp, tap)
                else:
                    ops.conv_pair_fwd(cur, N, nr, self._n_keep, H, W, c, L["cout"], Lb["cout"], L["pf"], L["b"], Lb["pf"], Lb["b"],
                                      ya, yb, yp, tap)
                self._geom += [(ya, L["cout"], H, W), (yb, Lb["cout"], H, W), (yp, Lb["cout"], H // 2, W // 2)]
                c, H, W, cur, pooled, skip = Lb["cout"], H // 2, W // 2, yp, None, 2
                continue
            if L["kind"] == "conv":
                y = self._flat(("a", j), N, L["cout"], H, W)
                tap = None
                if L["relu_idx"] in self.taps:
                    tap = torch.empty((N, L["cout"], H, W), dtype=torch.float32, device=self.device)
                    outs.append(tap)
                nxt = next((M["pf"] for M in self.layers[j + 1:] if M["kind"] == "conv"), None) if self.prefetch_next else None
                pooled = None
                if (self.fold_pool_fwd and j + 1 < len(self.layers) and self.layers[j + 1]["kind"] == "pool" and H % 2 == 0
                        and W % 2 == 0 and c >= self.fold_pool_fwd_min_cin):
                    # the pool that follows rides in this launch's epilogue (ops.conv3x3_pool): no maxpool2_fwd launch.  Measured in
                    # the c2 iteration (profiles/r04_pool_fold_ab.txt): conv2_2 18.3 + pool 4.8 -> 20.8 us; conv1_2, whose plain
                    # launch is the window-staged kernel (no two-row tiles there): 16.7 + 5.3 -> 22.6 us, hence c >= 128 only
                    pooled = self._flat(("a", j + 1), N, L["cout"], H // 2, W // 2)
                    ops.conv3x3_pool(cur, N, nr, H, W, c, L["cout"], L["pf"], L["b"], y, pooled, tap,
                                     L["cout"] if tap is not None else 0, next_pack=nxt)
                else:
                    ops.conv3x3(cur, N, nr, H, W, c, L["cout"], L["pf"], L["b"], 0, None, y, tap, L["cout"] if tap is not None else 0,
                                next_pack=nxt)
                c = L["cout"]
            else:
                y = self._flat(("a", j), N, c, H // 2, W // 2)
                if pooled is not None:
                    pass                                              # written by the launch of the layer below
                else:
                    ops.maxpool2_fwd(cur, N, H, W, c, y)
                pooled = None
                H, W = H // 2, W // 2
                if L["idx"] in self.taps:                             # a tap on a pooled tensor (models/style_loss.py:12-14)
                    outs.append(ops.trunk_export(y, N, N, c, H, W, is_f16=True))
            self._geom.append((y, c, H, W))
            cur = y
        return outs

    def _tap_addend(self, g, N, n, c, H, W):
        """A tap gradient on a pre-pool layer as the flat bf16 tensor the pool's backward adds in: g is either that tensor already
        (uint8 storage from _flat("tapadd", ...): written by ops.lpips_layers itself) or the fp32 (n, c, H, W) gradient."""
        if g.dtype == torch.uint8:
            return g
        add = self._flat("tapadd", N, c, H, W)
        ops.trunk_grad_in(g, None, N, n, c, H, W, add)
        return add

    def _backward(self, gtaps, n, scale, xshape, zero_rest=True, top_writer=None):
        """dL/dx for the first n images from the tap gradients.  zero_rest=False leaves images >= n of the
        returned tensor uninitialised (callers that only read [:n]).  top_writer(y, N, n, c, H, W, dz): the producer of the TOP
        tap's gradient writes the flat gated tensor dz itself (ops.cx_fwd_bwd_flat) instead of handing over an fp32 tensor for
        npp_trunk_grad_in; its entry in gtaps is then just a non-None placeholder."""
        N = xshape[0]
        dimg = (torch.zeros if zero_rest else torch.empty)(xshape, dtype=torch.float32, device=self.device)
        if n == 0:
            return dimg
        if self._n_keep is not None and n > self._n_keep:
            raise RuntimeError(f"HipTrunk: backward for {n} images after a forward that kept the layer outputs of {self._n_keep}")
        tap_of = {}
        k = 0
        for j, L in enumerate(self.layers):
            tapped = (L["kind"] == "conv" and L["relu_idx"] in self.taps) or (L["kind"] == "pool" and L["idx"] in self.taps)
            if tapped:
                if gtaps[k] is not None:
                    if L["kind"] == "conv" and not L.get("tap_ok"):
                        raise NotImplementedError("HipTrunk: gradient taps must sit on the top layer, on a pool, or right before a pool")
                    tap_of[j] = gtaps[k]
                k += 1
        if not tap_of:
            return dimg
        flip = [0]

        def gbuf(C, H, W):
            flip[0] ^= 1
            return self._flat(("g", flip[0]), N, C, H, W)

        # Reverse walk.  State: either dz = dL/d(pre-activation) of conv layer j (ReLU gate applied), or gp = dL/d(output) of
        # pool layer j.  Layers above the highest tapped one receive no gradient.
        j = max(tap_of)
        y, c, H, W = self._geom[j]
        cur = gbuf(c, H, W)
        if self.layers[j]["kind"] == "conv":
            if tap_of[j].dtype == torch.uint8 if isinstance(tap_of[j], torch.Tensor) else False:
                cur = tap_of[j]                                   # already dL/d(pre-activation), flat and gated (ops.lpips_layers)
            elif top_writer is not None:
                top_writer(y, N, n, c, H, W, cur)
            else:
                ops.trunk_grad_in(tap_of[j], y, N, n, c, H, W, cur,     # dz_j = dL/dtap * [y > 0]
                                  next_pack=self.layers[j]["pb"] if self.prefetch_next else None)
            state = "dz"
        else:
            ops.trunk_grad_in(tap_of[j], None, N, n, c, H, W, cur)
            state = "gp"
        while True:
            L = self.layers[j]
            if state == "gp":                                        # pool j: route into the conv layer below, gate, add its tap
                yp, cp, Hp, Wp = self._geom[j - 1]
                add = None
                if (j - 1) in tap_of:
                    add = self._tap_addend(tap_of[j - 1], N, n, cp, Hp, Wp)
                dzp = gbuf(cp, Hp, Wp)
                ops.maxpool2_bwd(cur, yp, add, N, n, Hp, Wp, cp, dzp)
                cur, j, H, W, state = dzp, j - 1, Hp, Wp, "dz"
                continue
            if j == 0:                                               # dz of the image layer -> dL/dimage (fp32, times input scale)
                # (final_next_pack: what the launch AFTER this pass streams first -- the loop sets it to the MLP's backward pack)
                ops.conv3x3(cur, N, n, H, W, L["cout"], 16, L["pb"], None, 2, None, None, dimg, 3, scale,
                            next_pack=self.final_next_pack if self.prefetch_next else None)
                return dimg
            prev = self.layers[j - 1]
            if (j == 1 and prev["kind"] == "conv" and (int(self.fuse_pairs) & 4) and 0 not in tap_of and prev["cout"] == L["cout"]
                    and lib().npp_conv_pair_dgrad_ok(H, W, L["cout"])):
                # layers 1 and 0 in ONE launch: conv b's data gradient, conv a's ReLU gate, conv a's data gradient -> dL/dimage
                ops.conv_pair_dgrad(cur, N, n, H, W, L["cout"], L["pb"], self._geom[0][0], prev["pb"], dimg, scale)
                return dimg
            if prev["kind"] == "conv":
                if (j - 1) in tap_of:
                    raise NotImplementedError("HipTrunk: gradient tap below a conv layer")
                yp, cp, _, _ = self._geom[j - 1]
                dzp = gbuf(cp, H, W)
                ops.conv3x3(cur, N, n, H, W, L["cout"], cp, L["pb"], None, 1, yp, dzp, next_pack=self._pb_below(j))
                cur, j = dzp, j - 1
            elif self.fold_pool_bwd and (j - 1) not in tap_of and j >= 2 and self.layers[j - 2]["kind"] == "conv":
                # pool at j-1, conv at j-2: the data gradient, the pool's backward, layer j-2's ReLU gate and its tap gradient in
                # ONE launch (ops.conv3x3_dgrad_pool) -- no pooled gradient tensor, no maxpool2_bwd launch
                yp, cp, Hp, Wp = self._geom[j - 2]
                add = None
                if (j - 2) in tap_of:
                    add = self._tap_addend(tap_of[j - 2], N, n, cp, Hp, Wp)
                dzp = gbuf(cp, Hp, Wp)
                ops.conv3x3_dgrad_pool(cur, N, n, H, W, L["cout"], cp, L["pb"], yp, add, dzp, next_pack=self._pb_below(j))
                cur, j, H, W, state = dzp, j - 2, Hp, Wp, "dz"
            else:                                                    # pool at j-1: ungated gradient w.r.t. the pooled tensor
                _, cp, _, _ = self._geom[j - 1]
                g = gbuf(cp, H, W)
                ops.conv3x3(cur, N, n, H, W, L["cout"], cp, L["pb"], None, 2, None, g, next_pack=self._pb_below(j))
                if (j - 1) in tap_of:
                    ops.trunk_grad_in(tap_of[j - 1], None, N, n, cp, H, W, g, accumulate=True)
                cur, j, state = g, j - 1, "gp"


class _CXFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fx, fy, band_width, weight):
        loss, dfx = ops.cx_fwd_bwd(fx.contiguous(), fy.contiguous(), band_width, weight, 1.0, None, fx.requires_grad)
        ctx.save_for_backward(dfx if dfx is not None else torch.empty(0))
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dfx,) = ctx.saved_tensors
        return dfx * g, None, None, None


def contextual_loss(x, y, band_width=0.5, weight=None, loss_type="cosine"):
    """contextual_loss/functional.py:9-63 on feature tensors."""
    assert x.size() == y.size(), "input tensor must have the same size."
    assert loss_type == "cosine", "only the cosine distance is on the built path"
    return _CXFunction.apply(x, y, float(band_width), weight)


_CX_FLAT = True     # False: the separate cx_dx_finish + npp_trunk_grad_in launches (comparator of tests/test_gpu_parity.py)


class ContextualLoss(nn.Module):
    _MEAN, _STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)        # contextual.py:41-46

    def __init__(self, band_width=0.5, loss_type="cosine", use_vgg=False, vgg_layer="relu3_4", vgg_state_dict=None, device="cuda"):
        super().__init__()
        assert band_width > 0, "band_width parameter must be positive."
        assert loss_type == "cosine" and vgg_layer == "relu3_4"
        self.band_width = band_width
        if use_vgg:
            self.hip_trunk = HipTrunk(_VGG19, taps=(17,), state_dict=vgg_state_dict, device=device)

    def forward(self, x, y, weight=None):
        if hasattr(self, "hip_trunk"):
            assert x.shape[1] == 3 and y.shape[1] == 3, "VGG model takes 3 chennel images."
            n = x.shape[0]                                   # (x - mean) / std is folded into the image-in kernel
            f = self.hip_trunk(torch.cat([x, y.detach()], 0), n_grad=n, scale=[1.0 / s for s in self._STD],
                               shift=[-m / s for m, s in zip(self._MEAN, self._STD)])[0]
            x, y = f[:n], f[n:].detach()
        return contextual_loss(x, y, self.band_width, weight)

    def input_norm(self):
        """(scale, shift) of the trunk's input normalisation (x - mean) / std, contextual.py:56-61."""
        return [1.0 / s for s in self._STD], [-m / s for m, s in zip(self._MEAN, self._STD)]

    def fused(self, xy, n, scale, loss_buf, weight=None, x0_ready=False, x0_src=None):
        """Explicit forward + backward of `scale * self(xy[:n], xy[n:])` without autograd (the loop's path):
        accumulates the loss into loss_buf[0] and returns dL/dxy (only [:n] is defined).  x0_ready: xy is only the
        SHAPE of the batch, whose normalised flat form was already written into hip_trunk.input_buffer()."""
        t = self.hip_trunk
        sc, sh = self.input_norm()
        f = t._forward(xy, sc, sh, x0_ready, n_keep=n, x0_src=x0_src)[0]
        shape = tuple(xy) if x0_ready else tuple(xy.shape)
        if weight is None and _CX_FLAT:
            # the core's last launch writes the trunk's flat gradient tensor itself (no fp32 dL/dfeatures, no npp_trunk_grad_in)
            def top(y, N, nn, c, H, W, dz):
                ops.cx_fwd_bwd_flat(f[:n], f[n:], y, dz, N, self.band_width, scale, loss_buf)
            return t._backward([True], n, sc, shape, zero_rest=False, top_writer=top)
        _, dfx = ops.cx_fwd_bwd(f[:n], f[n:], self.band_width, weight, scale, loss_buf, True)
        return t._backward([dfx], n, sc, shape, zero_rest=False)


class _LPIPSLayerFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, f0, f1, owner, kk, robust=True):
        loss = torch.zeros(1, dtype=torch.float32, device=f0.device)
        need = f0.requires_grad
        df0 = torch.empty_like(f0) if need else None
        dlat = torch.zeros_like(owner.latents[kk]) if (need and robust) else None
        ops.lpips_layer(f0.contiguous(), f1.contiguous(), owner.lins[kk], owner.latents[kk] if robust else None, owner.spline, owner.n_knots,
                        owner.x_scale, 1.0, loss, df0, dlat)
        ctx.owner, ctx.kk, ctx.robust = owner, kk, robust
        ctx.save_for_backward(df0 if need else torch.empty(0), dlat if dlat is not None else torch.empty(0))
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        df0, dlat = ctx.saved_tensors
        if ctx.robust:
            ctx.owner.dlatents[ctx.kk].add_(dlat * g)       # the reference's list of AdaptiveLossFunction params
            ctx.owner.touched = True
        return df0 * g, None, None, None, None


def _latent_blobs(inits, dev):
    """The adaptive-loss latents of several taps / levels as views of ONE blob each for the values, their gradients and the two Adam
    moments: optimizer.step() and zero_grad() over them are one launch each instead of one per tap (found in the kernel trace of the
    'same'-source iterations: 5 Adam launches + 5 fills of ~4.8 us for 2944 latents)."""
    sizes = [int(t.numel()) for t in inits]
    lat = torch.cat([t.reshape(-1).to(torch.float32) for t in inits]).to(dev)
    blobs = (lat, torch.zeros_like(lat), torch.zeros_like(lat), torch.zeros_like(lat))
    offs = np.concatenate([[0], np.cumsum(sizes)])
    views = tuple([b[int(offs[i]):int(offs[i + 1])] for i in range(len(sizes))] for b in blobs)
    return blobs, views


class LPIPS(nn.Module):
    """LPIPS(net='vgg') with the reference's adaptive-robust head.  forward(in0, in1, use_robust,
    normalize) returns the batch MEAN as a scalar (the caller's torch.mean, train.py:249, is folded in)."""
    chns = [64, 128, 256, 512, 512]

    _SHIFT, _SCALE = (-.030, -.088, -.188), (.458, .448, .450)          # lpips.py:136-143 ScalingLayer

    def __init__(self, net="vgg", lin_weights=None, vgg_state_dict=None, device="cuda"):
        super().__init__()
        assert net in ("vgg", "vgg16")
        dev = torch.device(device)
        self.hip_trunk = HipTrunk(_VGG16, taps=(3, 8, 15, 22, 29), state_dict=vgg_state_dict, seed=4321, device=dev)
        self.register_buffer("shift", torch.tensor(self._SHIFT)[None, :, None, None])
        self.register_buffer("scale", torch.tensor(self._SCALE)[None, :, None, None])
        if lin_weights is None:          # weights/v0.1/vgg.pth is not redistributed here: fixed-seed non-negative stand-ins
            rng = np.random.RandomState(7)
            lin_weights = [np.abs(rng.randn(c)).astype(np.float32) * 0.05 for c in self.chns]
        self.lins = [torch.as_tensor(np.asarray(w, np.float32).reshape(-1)).to(dev) for w in lin_weights]
        # AdaptiveLossFunction(num_dims=chn) per tap (lpips.py:57-61): [latent_alpha(C) | latent_scale(C)]
        (self._lat, self._dlat, self._lat_m, self._lat_v), (self.latents, self.dlatents, self.lat_m, self.lat_v) = _latent_blobs(
            [torch.cat([torch.full((c,), 2.3841858e-07), torch.zeros(c)]) for c in self.chns], dev)
        self.lat_step = 0
        self.touched = False
        # comparator switches of tests/test_gpu_parity.py (all bit-identical; the defaults are the measured-best forms): one launch per
        # head; tap gradients as fp32 tensors through npp_trunk_grad_in instead of written flat by the heads launch
        self.grouped_heads, self.flat_tap_grads, self.flat_top_tap = True, True, True
        self.spline, self.n_knots, self.x_scale = ops.load_spline(dev)
        self.to(dev)

    def forward(self, in0, in1, use_robust=True, retPerLayer=False, normalize=False):
        """use_robust=False: the plain head (lpips.py:108-109) with its gradient -- the loop under --use_adaptive_perceptual_loss off."""
        assert not retPerLayer, "retPerLayer is not on the built path"
        # 2x-1 (lpips.py:96-98) and the scaling layer are folded into the image-in kernel
        a = 2.0 if normalize else 1.0
        n = in0.shape[0]
        f = self.hip_trunk(torch.cat([in0, in1.detach()], 0), n_grad=n, scale=[a / s for s in self._SCALE],
                           shift=[((-1.0 if normalize else 0.0) - sh) / s for sh, s in zip(self._SHIFT, self._SCALE)])
        outs0, outs1 = [t[:n] for t in f], [t[n:].detach() for t in f]
        val = 0
        for kk in range(5):
            val = val + _LPIPSLayerFunction.apply(outs0[kk], outs1[kk], self, kk, bool(use_robust))
        return val

    def fused(self, xy, n, scale, loss_buf, normalize=True, use_robust=True):
        """Explicit forward + backward of `scale * self(xy[:n], xy[n:], use_robust, normalize)` (batch mean)
        without autograd: accumulates into loss_buf[0] and self.dlatents, returns dL/dxy ([:n] defined)."""
        a = 2.0 if normalize else 1.0
        sc = [a / s for s in self._SCALE]
        sh = [((-1.0 if normalize else 0.0) - b) / s for b, s in zip(self._SHIFT, self._SCALE)]
        t = self.hip_trunk
        dfs = [None] * 5

        def head(kk, f):
            df0 = torch.empty((n,) + tuple(f.shape[1:]), dtype=torch.float32, device=f.device)
            ops.lpips_layer(f[:n], f[n:], self.lins[kk], self.latents[kk] if use_robust else None, self.spline, self.n_knots, self.x_scale, scale,
                            loss_buf, df0, self.dlatents[kk])
            dfs[kk] = df0
        # (Measured and dropped, round 4: the five heads -- 15-25 us each, 100 us in a row behind the trunk -- on a helper stream beside
        # the deeper layers of the forward pass: the 'same' iteration went 0.791 -> 0.811 ms; the branch is not what the device waits for.)
        feats = t._forward(xy, sc, sh, n_keep=n)
        if self.grouped_heads:                                  # the five heads in ONE launch (they are independent: 100 us in a row before)
            N = xy.shape[0]
            # the taps right before a pool hand their gradient over as the flat bf16 tensor the backward pass adds in (no fp32
            # tensor, no npp_trunk_grad_in launch each); the top tap's goes through the ReLU gate of its own layer as before
            flat = [self.flat_tap_grads and (kk < len(feats) - 1 or self.flat_top_tap) for kk in range(len(feats))]
            ytop = t._geom[-1][0]                                # the top tap's gradient passes its own layer's ReLU gate on the way
            dfl = [(t._flat("tapadd", N, f.shape[1], f.shape[2], f.shape[3]), N, ytop if kk == len(feats) - 1 else None) if fl else None
                   for kk, (f, fl) in enumerate(zip(feats, flat))]
            dfs = [None if fl else torch.empty((n,) + tuple(f.shape[1:]), dtype=torch.float32, device=f.device) for f, fl in zip(feats, flat)]
            ops.lpips_layers([f[:n] for f in feats], [f[n:] for f in feats], self.lins, self.latents if use_robust else None, self.spline,
                             self.n_knots, self.x_scale, scale, loss_buf, dfs, self.dlatents, dflats=dfl)
            dfs = [d[0] if d is not None else g for d, g in zip(dfl, dfs)]
        else:
            for kk, f in enumerate(feats):
                head(kk, f)
        self.touched = self.touched or bool(use_robust)        # (the plain head gives the latents no gradient: Adam skips them)
        return t._backward(dfs, n, sc, tuple(xy.shape), zero_rest=False)

    def fused_groups(self, xy, XL, groups, scale, normalize=True, use_robust=True):
        """fused() for several independent fits in ONE pass of the (frozen, shared) trunk -- the 'same' images of a stacked iteration
        (stack.StackedFit): xy (N, 3, H, W) holds the prediction halves of all groups in [0, XL) and the real halves in [XL, 2 XL)
        (N fixes the buffers' geometry); groups = [(o, n, lp, loss_buf)]: samples [o, o + n) of both halves belong to the LPIPS object
        lp -- its latents, its latent gradients, its own batch mean (scale / n) accumulated into loss_buf[0].  The heads run one launch
        per group, each writing its samples' rows of the shared flat tap-gradient tensors.  Returns dL/dxy ([:XL] defined)."""
        if not (self.grouped_heads and self.flat_tap_grads and self.flat_top_tap):
            raise RuntimeError("LPIPS.fused_groups: the grouped heads with flat tap gradients only")
        a = 2.0 if normalize else 1.0
        sc = [a / s for s in self._SCALE]
        sh = [((-1.0 if normalize else 0.0) - b) / s for b, s in zip(self._SHIFT, self._SCALE)]
        t = self.hip_trunk
        N = xy.shape[0]
        feats = t._forward(xy, sc, sh, n_run=2 * XL, n_keep=XL)
        ytop = t._geom[-1][0]
        flats = [t._flat("tapadd", N, f.shape[1], f.shape[2], f.shape[3]) for f in feats]

        def from_image(buf, o, H, W):                           # the flat tensor as image o's launch sees it (csrc/npp_trunk_layout.h:
            return buf[o * (H + 2) * (W + 2) * 16:]             # image n starts n (H + 2)(W + 2) 16-byte units into every channel row)
        for o, n, lp, loss_buf in groups:
            dfl = [(from_image(fl, o, f.shape[2], f.shape[3]), N,
                    from_image(ytop, o, f.shape[2], f.shape[3]) if kk == len(feats) - 1 else None) for kk, (f, fl) in enumerate(zip(feats, flats))]
            ops.lpips_layers([f[o:o + n] for f in feats], [f[XL + o:XL + o + n] for f in feats], lp.lins, lp.latents if use_robust else None,
                             lp.spline, lp.n_knots, lp.x_scale, scale, loss_buf, [None] * len(feats), lp.dlatents, dflats=dfl)
            lp.touched = lp.touched or bool(use_robust)
        return t._backward(flats, XL, sc, tuple(xy.shape), zero_rest=False)

    @torch.no_grad()
    def plain(self, in0, in1, normalize=False):
        """LPIPS.forward(in0, in1, use_robust=False) (lpips.py:92-133), forward only: the candidate score of
        NPP_proposal/search.py:193.  Returns a (1,) tensor: the SUM over the batch of the per-image distances
        (search.py evaluates one image pair)."""
        a = 2.0 if normalize else 1.0
        sc = [a / s for s in self._SCALE]
        sh = [((-1.0 if normalize else 0.0) - b) / s for b, s in zip(self._SHIFT, self._SCALE)]
        n = in0.shape[0]
        feats = self.hip_trunk._forward(torch.cat([in0, in1], 0).contiguous(), sc, sh)
        out = torch.zeros(1, dtype=torch.float32, device=in0.device)
        ws = None
        if ops.DETERMINISTIC:                           # fixed-order sums: the score that ranks the proposals is bit-reproducible
            key = torch.cuda.current_stream(in0.device).cuda_stream      # (one scratch per stream: search threads score side by side)
            ws = self._plain_ws.get(key) if hasattr(self, "_plain_ws") else None
            if ws is None:
                if not hasattr(self, "_plain_ws"):
                    self._plain_ws = {}
                ws = self._plain_ws[key] = torch.zeros(len(feats), ops.LPIPS_PLAIN_SCRATCH, dtype=torch.float32, device=in0.device)
        for kk, f in enumerate(feats):
            ops.lpips_plain_layer(f[:n], f[n:], self.lins[kk], 1.0, out, None if ws is None else ws[kk])
        return out

    def zero_latent_grads(self):
        self._dlat.zero_()                              # one fill: the per-tap vectors are views of one blob (_latent_blobs)
        self.touched = False

    def adam_step(self, lr):
        """Adam over the robust latents; only called when they received a gradient this iteration
        (torch skips parameters whose grad is None: their step count does not advance).  One launch over the blob of all taps."""
        self.lat_step += 1
        ops.adam_step(self._lat, self._lat_m, self._lat_v, self._dlat, 1, self._lat.numel(), lr, self.lat_step)


_VGG16_STYLE = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M"]          # vgg16.features[:17]: enc_1 | enc_2 | enc_3


class StyleLoss:
    """models/style_loss.py VGG16FeatureExtractor.style_loss with use_adaptive=True (the remapping task's extra patch loss,
    NPP_remapping/train.py:253-261): VGG16 features[:5], [5:10], [10:17] (the three POOLED tensors; inputs are used as
    they are, no mean / std normalisation) -> Gram matrices -> per-element adaptive robust NLL of their difference with
    num_dims = C^2 latents per level -> / (c w h) -> mean (or per-sample mean times `weight`, summed).
    Trunk: HipTrunk (weights unpinned like the other trunks); everything after it: npp_gram_* / npp_robust_elem."""
    chns = [64, 128, 256]

    def __init__(self, vgg_state_dict=None, device="cuda", seed=777):
        self.device = torch.device(device)
        self.hip_trunk = HipTrunk(_VGG16_STYLE, taps=(4, 9, 16), state_dict=vgg_state_dict, seed=seed, device=self.device)
        # AdaptiveLossFunction(num_dims = chn ** 2) per level (style_loss.py:23-27): [latent_alpha(D) | latent_scale(D)]
        (self._lat, self._dlat, self._lat_m, self._lat_v), (self.latents, self.dlatents, self.lat_m, self.lat_v) = _latent_blobs(
            [torch.cat([torch.full((c * c,), 2.3841858e-07), torch.zeros(c * c)]) for c in self.chns], self.device)
        self.lat_step = 0
        self.spline, self.n_knots, self.x_scale = ops.load_spline(self.device)

    def head(self, feats, n, scale, loss_buf, weight=None, want_grad=True):
        """From the three pooled feature tensors (2n, C, h, w) = [A | B]: accumulates scale * style_loss into loss_buf[0] and
        the latent gradients into self.dlatents; returns [dL/dA_i] (n, C, h, w)."""
        dfs = []
        for i, f in enumerate(feats):
            C_, h, w = f.shape[1:]
            fa, fb = f[:n].contiguous(), f[n:].contiguous()
            ga, gb = ops.gram_fwd(fa), ops.gram_fwd(fb)
            D = C_ * C_
            denom = float(C_ * h * w)
            if weight is None:
                coef = [scale / (n * D * denom)] * n                       # torch.mean over (N, C^2) of nll / (c w h)
            else:
                coef = [scale * float(wv) / (D * denom) for wv in weight]  # style_loss.py:66-69
            dd = ops.robust_elem(ga.reshape(n, D), gb.reshape(n, D), self.latents[i], self.spline, self.n_knots, self.x_scale, coef,
                                 loss_buf, want_grad, self.dlatents[i])
            dfs.append(ops.gram_bwd(dd.reshape(n, C_, C_), fa) if want_grad else None)
        return dfs

    def fused(self, xy, n, scale, loss_buf, weight=None):
        """Explicit forward + backward of scale * style_loss(xy[:n], xy[n:], weight): returns dL/dxy ([:n] defined)."""
        ones, zeros = (1.0, 1.0, 1.0), (0.0, 0.0, 0.0)
        t = self.hip_trunk
        feats = t._forward(xy, ones, zeros, n_keep=n)
        dfs = self.head(feats, n, scale, loss_buf, weight)
        return t._backward(dfs, n, ones, tuple(xy.shape), zero_rest=False)

    def zero_latent_grads(self):
        self._dlat.zero_()

    def adam_step(self, lr):
        self.lat_step += 1
        ops.adam_step(self._lat, self._lat_m, self._lat_v, self._dlat, 1, self._lat.numel(), lr, self.lat_step)
