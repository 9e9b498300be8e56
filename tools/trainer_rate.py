#!/usr/bin/env python3
"""Iterations/s of the reference-surface trainer (MODEL(args).train(), main.py flags) fed from an image folder through
the GPU input pipeline (synthetic PNGs): the step replayed as a hipGraph (default) vs launched kernel by kernel from
Python (--eager).  The first epoch decodes and uploads; later epochs run on images resident in HBM.
   python tools/trainer_rate.py [batch] [n_images] [epochs]"""
import os, sys, time, tempfile, io, contextlib
import numpy as np, torch
from PIL import Image
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import main as cli
from pytorch_super_resolution_model_collection_amd.sr_trainers import TRAINERS
batch = sys.argv[1] if len(sys.argv) > 1 else "16"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
epochs = int(sys.argv[3]) if len(sys.argv) > 3 else 4
with tempfile.TemporaryDirectory() as root:
    d = os.path.join(root, "DIV2K", "DIV2K_train_LR_bicubic", "X4")
    os.makedirs(d)
    rng = np.random.RandomState(0)
    for i in range(n):
        Image.fromarray(rng.randint(0, 256, (256, 256, 3), dtype=np.uint8)).save(os.path.join(d, "%04d.png" % i))
    for mode in ("graph", "eager"):
        args = cli.parse_args(["--model_name", "EDSR", "--num_epochs", "1", "--save_epochs", "100", "--batch_size", batch,
                               "--crop_size", "128", "--data_dir", root, "--num_threads", "8",
                               "--save_dir", os.path.join(root, "out_" + mode)] + (["--eager"] if mode == "eager" else []))
        t = TRAINERS["EDSR"](args)
        loader = t.load_dataset(t.train_dataset, is_train=True)
        assert loader is not None
        t.num_epochs = 1
        with contextlib.redirect_stdout(io.StringIO()):
            t.train(loader)                      # epoch 1: decode + upload, eager first step, capture
            t.num_epochs = epochs
            torch.cuda.synchronize(); t0 = time.perf_counter()
            t.train(loader)                      # (re-initialises the model; the images stay resident)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        iters = epochs * len(loader)
        print("EDSR trainer, batch %s, %d resident images: %s step %.3f ms / iteration = %.0f patches/s (%d iterations incl. "
              "model set-up)" % (batch, n, mode, dt / iters * 1e3, int(batch) * iters / dt, iters))
