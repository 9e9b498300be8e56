#!/usr/bin/env python3
"""Writes tests/golden/img_interp.npz: inputs (from seeds, not stored) -> outputs of the img_interp oracle, i.e. of
Pillow's resampler driven exactly as /root/reference/utils.py:242-269 drives it.  Run in the build container:
    python tests/golden/make_golden_interp.py
Outputs are stored as uint8 (they are k/255 by construction), so the fixture is a few KB."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import fill, img_interp  # noqa: E402

CASES = [  # tag, shape, scale, interpolation, seed
    ("bicubic_x2_rgb", (2, 3, 17, 23), 2, "bicubic", 1),
    ("bicubic_x4_rgb", (2, 3, 16, 16), 4, "bicubic", 2),
    ("bicubic_x3_l", (1, 1, 11, 13), 3, "bicubic", 3),
    ("bicubic_x1p5", (1, 3, 10, 14), 1.5, "bicubic", 4),
    ("bicubic_down_x0p5", (1, 3, 24, 20), 0.5, "bicubic", 5),
    ("bilinear_x2", (2, 3, 9, 9), 2, "bilinear", 6),
    ("bilinear_x3", (1, 1, 7, 12), 3, "bilinear", 7),
    ("nearest_x2", (1, 3, 8, 8), 2, "nearest", 8),
    ("nearest_x3", (1, 3, 7, 10), 3, "nearest", 9),
]

if __name__ == "__main__":
    out = {}
    for tag, shape, scale, interp, seed in CASES:
        x = fill.rand(shape, seed)
        y = img_interp.img_interp(x, scale, interp)
        q = np.rint(y.numpy() * 255.0).astype(np.uint8)
        assert np.array_equal((q.astype(np.float32) / np.float32(255.0)), y.numpy())
        out[tag] = q
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "img_interp.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
