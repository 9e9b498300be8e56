#!/usr/bin/env python3
"""Iterations/s of the reference-surface trainer (MODEL(args).train(), main.py flags) on synthetic patches: the step
replayed as a hipGraph (default) vs launched kernel by kernel from Python (--eager).
   python tools/trainer_rate.py [MODEL] [batch] [steps_per_epoch]"""
import os, sys, time, tempfile, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import main as cli
from pytorch_super_resolution_model_collection_amd.sr_trainers import TRAINERS
name = sys.argv[1] if len(sys.argv) > 1 else "EDSR"
batch = sys.argv[2] if len(sys.argv) > 2 else "16"
steps = sys.argv[3] if len(sys.argv) > 3 else "300"
crop = {"EDSR": "128", "VDSR": "41", "SRCNN": "64", "ESPCN": "64", "FSRCNN": "64"}.get(name, "64")
for mode in ("graph", "eager"):
    with tempfile.TemporaryDirectory() as d:
        args = cli.parse_args(["--model_name", name, "--num_epochs", "2", "--save_epochs", "100", "--batch_size", batch,
                               "--steps_per_epoch", steps, "--crop_size", crop, "--save_dir", d] + (["--eager"] if mode == "eager" else []))
        t = TRAINERS[name](args)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        t.train()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("%s batch %s: %s step: %.3f ms / iteration (%d iterations incl. set-up)" % (name, batch, mode, dt / (2 * int(steps)) * 1e3, 2 * int(steps)))
