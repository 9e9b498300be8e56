#!/usr/bin/env python3
"""tests/golden/dataset_r2.npz: (lr, hr, bc) items of oracle/dataset_pil.py — the Pillow restatement of the reference's
TrainDatasetFromFolder / TestDatasetFromFolder (the reference's own classes need torchvision, which this image lacks:
see the oracle's header) — on the synthetic PNGs of tests/test_data_pipeline.py::make_images, for fixed `random` seeds.
Pins the oracle (and the Pillow build behind it) over time; the GPU tests compare the device pipeline with the oracle live.
Run:  python tests/golden/make_golden_dataset.py      (CPU only)"""
import os
import random
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import dataset_pil as O  # noqa: E402
import test_data_pipeline as T      # noqa: E402

CASES = [(0, 3, dict()), (2, 11, dict()), (4, 13, dict(random_scale=False, crop_size=32)), (1, 14, dict(is_gray=True)),
         (3, 15, dict(crop_size=24, scale_factor=2))]


def main():
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        folder = T.make_images(os.path.join(tmp, "train"), T.SIZES)
        for k, (index, seed, cfg) in enumerate(CASES):
            kw = dict(crop_size=32, scale_factor=4)
            kw.update(cfg)
            ds = O.TrainDatasetFromFolder([folder], **kw)
            random.seed(seed)
            for name, t in zip(("lr", "hr", "bc"), ds[index]):
                out["train%d.%s" % (k, name)] = (t.numpy() * 255 + 0.5).astype(np.uint8)   # items are uint8 / 255 exactly
        ts = O.TestDatasetFromFolder(folder, scale_factor=4)
        for name, t in zip(("lr", "hr", "bc"), ts[0]):
            out["test0.%s" % name] = (t.numpy() * 255 + 0.5).astype(np.uint8)
    path = os.path.join(HERE, "dataset_r2.npz")
    np.savez_compressed(path, **out)
    print("wrote %s (%.3f MB, %d arrays)" % (path, os.path.getsize(path) / 1e6, len(out)))


if __name__ == "__main__":
    main()
