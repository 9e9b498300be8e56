#!/usr/bin/env python3
"""SRGAN through the trainer surface (main.py flags) with and without --prune_dead_grads: same loss history."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import main as cli
from pytorch_super_resolution_model_collection_amd.sr_trainers import TRAINERS
import tempfile
hist = {}
for flag in ([], ["--prune_dead_grads"]):
    d = tempfile.mkdtemp()
    args = cli.parse_args(["--model_name", "SRGAN", "--num_epochs", "2", "--save_epochs", "10", "--batch_size", "2",
                           "--synthetic", "--steps_per_epoch", "4", "--lr", "1e-4", "--crop_size", "32", "--epoch_pretrain", "1",
                           "--save_dir", d] + flag)
    torch.manual_seed(0)
    hist[bool(flag)] = TRAINERS["SRGAN"](args).train()
print(hist)
for (d0, g0), (d1, g1) in zip(hist[True], hist[False]):
    assert abs(d0 - d1) <= 1e-3 * abs(d1) + 1e-6 and abs(g0 - g1) <= 1e-3 * abs(g1) + 1e-6
print("trainer ok")
