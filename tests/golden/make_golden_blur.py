#!/usr/bin/env python3
"""Golden G13: NPP_remapping/blur_detection.py:14-60 get_blur_map on a small synthetic image.  cv2 is not installed: the
function definition is ast-executed from the reference file with a stand-in `cv2` whose cvtColor is OpenCV's documented
14-bit fixed-point RGB2GRAY, and `np.float` (removed from NumPy) mapped to float.      python tests/golden/make_golden_blur.py"""
import ast
import os
import types

import numpy as np
import scipy.ndimage as ndimage

REF, OUT = "/root/reference", os.path.dirname(os.path.abspath(__file__))


def main():
    cv2 = types.SimpleNamespace(COLOR_RGB2GRAY=7,
                                cvtColor=lambda img, code: ((img.astype(np.int64) * np.array([4899, 9617, 1868])).sum(-1) + 8192 >> 14).astype(np.uint8))
    npx = types.ModuleType("np")
    npx.__dict__.update(np.__dict__)
    npx.float = float
    ns = {"cv2": cv2, "np": npx, "ndimage": ndimage}
    tree = ast.parse(open(os.path.join(REF, "NPP_remapping/blur_detection.py")).read())
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == "get_blur_map":
            exec(compile(ast.Module([node], []), "blur_detection.py", "exec"), ns)
    rng = np.random.RandomState(0)
    H, W = 72, 88
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    sharp = 0.5 + 0.5 * np.sin(xx * 0.9) * np.cos(yy * 0.7)
    soft = 0.5 + 0.5 * np.sin(xx * 0.12) * np.cos(yy * 0.1)
    t = (xx > W // 2)[..., None]
    img = np.where(t, np.stack([soft, soft * 0.9, soft * 0.8], -1), np.stack([sharp, sharp * 0.8, 1 - sharp], -1))
    img = np.uint8(np.clip(img + 0.02 * rng.randn(H, W, 3), 0, 1) * 255)
    bm, clear = ns["get_blur_map"](img, thresh=50)
    np.savez_compressed(os.path.join(OUT, "g13_blur.npz"), img=img, blur_map=bm, clear=clear)
    print(bm.shape, clear.mean() / 255)


if __name__ == "__main__":
    main()
