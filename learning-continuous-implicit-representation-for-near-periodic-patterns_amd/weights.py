"""Where the pretrained weights of the loss trunks come from.  The reference gets them implicitly: torchvision's `pretrained=True`
(VGG19: externel_lib/contextual_loss/modules/vgg.py:22; VGG16: externel_lib/lpips/pretrained_networks.py:99; AlexNet: :60) downloads
into the torch hub cache, the proposal stage reads ./alexnet-owt-4df8aa71.pth (README "How to Run", models/model_def.py:20), and the
LPIPS 'lin' layers are files in its tree (lpips.py:60-75).  Here nothing is downloaded: an explicit --flag path wins, else the same
places are searched (working directory, $TORCH_HOME/hub/checkpoints, ~/.cache/torch/hub/checkpoints); the lin layers ship with
the package (resources/lpips_lin_v0_1.npz, tools/make_lpips_resource.py)."""
import glob
import os

import numpy as np

_PATTERNS = {"vgg19": ("vgg19-*.pth",), "vgg16": ("vgg16-*.pth",), "alexnet": ("alexnet-owt-*.pth", "alexnet-*.pth")}


def search_dirs():
    home = os.environ.get("TORCH_HOME") or os.path.join(os.environ.get("XDG_CACHE_HOME") or os.path.expanduser("~/.cache"), "torch")
    return [os.getcwd(), os.path.join(home, "hub", "checkpoints"), os.path.join(home, "checkpoints")]


def find_checkpoint(name, explicit=None):
    """Path of the torchvision state_dict for `name` ('vgg19' | 'vgg16' | 'alexnet'), or None."""
    if explicit is not None:
        if not os.path.isfile(explicit):
            raise FileNotFoundError(f"{explicit}: no such {name} checkpoint")
        return explicit
    for d in search_dirs():
        for pat in _PATTERNS[name]:
            hits = sorted(glob.glob(os.path.join(d, pat)))
            if hits:
                return hits[0]
    return None


_STATE_DICTS = {}      # (path, mtime) -> state dict
_STATE_DICTS_LOCK = __import__("threading").Lock()      # run.search_all loads from several host threads


def load_state_dict(path):
    """torch.load(path) once per file and process: the images of a directory run share the dicts, and with them the packed trunk
    weights (losses.HipTrunk keys its device weights / MFMA packs by the state dict's identity).  None -> None."""
    if path is None:
        return None
    import torch
    key = (os.path.abspath(path), os.path.getmtime(path))
    with _STATE_DICTS_LOCK:                                # one torch.load per file: two threads missing together must end up with ONE dict
        sd = _STATE_DICTS.get(key)                         # (losses.HipTrunk keys its packs by the dict's identity)
        if sd is None:
            if len(_STATE_DICTS) >= 8:
                _STATE_DICTS.pop(next(iter(_STATE_DICTS)))
            sd = _STATE_DICTS[key] = torch.load(path, map_location="cpu")
    return sd


def lpips_lin(net="vgg", path=None):
    """The five lin-layer weight vectors [(C,)] of LPIPS v0.1: from a user's lpips weights file (`lin{i}.model.1.weight`), else the
    packaged copy."""
    if path is not None:
        import torch
        sd = torch.load(path, map_location="cpu")
        return [sd[f"lin{i}.model.1.weight"].reshape(-1).numpy() for i in range(5)]
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "resources", "lpips_lin_v0_1.npz"))
    return [z[f"{net}_lin{i}"] for i in range(5)]


def resolve(args, names, random_ok):
    """Fill args.<name> (names from 'vgg19', 'vgg16', 'alexnet') with discovered checkpoint paths; SystemExit naming what is
    missing and where it was looked for unless random_ok (--random-trunks)."""
    lacking = []
    for n in names:
        p = find_checkpoint(n, getattr(args, n, None))
        setattr(args, n, p)
        if p is None:
            lacking.append(n)
    if lacking and not random_ok:
        raise SystemExit(f"missing pretrained torchvision weights {lacking}: pass --{' / --'.join(lacking)} <state_dict.pth>, or put "
                         f"{[_PATTERNS[n][0] for n in lacking]} into one of {search_dirs()} (where torchvision's pretrained=True leaves "
                         f"them), or pass --random-trunks to run on fixed-seed random trunks (synthetic / bench runs)")
    return lacking
