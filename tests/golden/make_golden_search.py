#!/usr/bin/env python3
"""Golden G11: the brute-force displacement search of NPP_proposal/feature_searching.py --
generate_possible_shifts (:267-277), compute_loss (:208-264), generate_periodicity (:118-156) with its helpers
find_second_shift_by_angle / shifts2angle / shifts2period / angle_diff -- on synthetic feature maps.

The module itself cannot be imported here (cv2, skimage, torchvision at import time), so the generator EXECUTES the
definitions of exactly these pure-torch functions from the reference file (ast-extracted at run time, nothing is copied into
the repository), plus gen_batches / calc_batch_size from utils/ops.py.      python tests/golden/make_golden_search.py
"""
import ast
import math
import os

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
WANT = {"NPP_proposal/feature_searching.py": ["feature_search", "generate_periodicity", "compute_loss", "generate_possible_shifts",
                                              "find_second_shift_by_angle", "shifts2angle", "shifts2period", "vector_norm", "angle_diff"],
        "utils/ops.py": ["gen_batches", "calc_batch_size"]}


def load_functions():
    ns = {"torch": torch, "math": math, "np": np}
    for rel, names in WANT.items():
        tree = ast.parse(open(os.path.join(REF, rel)).read())
        for node in tree.body:
            if isinstance(node, ast.FunctionDef) and node.name in names:
                exec(compile(ast.Module([node], []), os.path.join(REF, rel), "exec"), ns)
    return ns


def main():
    F = load_functions()
    g = torch.Generator().manual_seed(3)
    out = {}
    for tag, (C, h, w), rr in (("a", (7, 30, 44), (3, 6, 1)), ("b", (5, 41, 33), (2, 8, 3))):
        # a lattice-like map so that the search has a meaningful optimum, plus noise; last channel = the extra (gray / edge)
        yy, xx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
        base = torch.stack([torch.cos(2 * math.pi * (xx * (0.13 + 0.01 * c) + yy * 0.03 * c)) * torch.cos(2 * math.pi * (yy * 0.17 - xx * 0.02))
                            for c in range(C)]).float()
        act = torch.relu(base + 0.1 * torch.randn(C, h, w, generator=g))
        mask = torch.ones(h, w)
        mask[h // 3:h // 3 + h // 4, w // 4:w // 4 + w // 3] = 0                   # unknown block
        out[f"{tag}_act"], out[f"{tag}_mask"], out[f"{tag}_rr"] = act.numpy(), mask.numpy(), np.array(rr)
        for i in range(rr[0], rr[1], rr[2]):
            rx = ry = (i, i + rr[2])
            shifts = F["generate_possible_shifts"]((h, w), rx, ry, "cpu")
            out[f"{tag}_shifts_{i}"] = shifts.numpy()
            for edge in (True, False):
                losses = F["compute_loss"](act, mask, shifts, rx, edge_searching=edge)
                out[f"{tag}_loss_{i}_{int(edge)}"] = losses.numpy()
                per = F["generate_periodicity"](losses, shifts)
                if per[0] is not None:
                    out[f"{tag}_angles_{i}_{int(edge)}"] = np.array([float(a) for a in per[0]])
                    out[f"{tag}_periods_{i}_{int(edge)}"] = np.array([float(p) for p in per[1]])
                    out[f"{tag}_sel_{i}_{int(edge)}"] = np.stack([s.numpy() for s in per[2]])
        ang, per, sh = F["feature_search"](act, mask, repeat_range=rr, edge_searching=True)
        out[f"{tag}_fs_angles"] = np.array([[float(a) for a in x] for x in ang])
        out[f"{tag}_fs_periods"] = np.array([[float(p) for p in x] for x in per])
        out[f"{tag}_fs_shifts"] = np.array([[s.numpy() for s in x] for x in sh])
    np.savez_compressed(os.path.join(OUT, "g11_search.npz"), **out)
    print({k: v.shape for k, v in out.items() if "fs_" in k or "shifts_" in k})


if __name__ == "__main__":
    main()
