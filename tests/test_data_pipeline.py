"""The image-folder input pipeline (/root/reference/dataset.py:22-149, data.py:33-65).
CPU: the product draws Python's `random` exactly like the oracle restatement of the reference (same patch for the same
seed).  GPU: every item — LR, HR and bicubic images — is BIT-EQUAL to the oracle (Pillow does the pixel work there):
the device-side 8-bit resampler, crop / rotation / flip gather, ToTensor and the ToPILImage round trip."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import dataset_pil as O


def _pkg():
    import __graft_entry__
    __graft_entry__.build()
    import pytorch_super_resolution_model_collection_amd as pkg
    return pkg


def make_images(folder, sizes, seed=5):
    """Synthetic PNGs: smooth gradients + noise + a few hard edges (so that resampling overshoot / clipping occurs)."""
    from PIL import Image
    os.makedirs(folder, exist_ok=True)
    rs = np.random.RandomState(seed)
    for i, (w, h) in enumerate(sizes):
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.stack([(xx * 255.0 / max(w - 1, 1)), (yy * 255.0 / max(h - 1, 1)), ((xx + yy) % 64) * 4.0], -1)
        img += rs.normal(0, 25, img.shape)
        img[h // 3:h // 3 + 5, :, :] = 255
        img[:, w // 2:w // 2 + 3, :] = 0
        Image.fromarray(np.clip(img, 0, 255).astype(np.uint8), "RGB").save(os.path.join(folder, "img_%02d.png" % i))
    with open(os.path.join(folder, "notes.txt"), "w") as f:   # non-image files are skipped (dataset.py:9-10)
        f.write("x")
    return folder


SIZES = [(97, 61), (64, 64), (130, 150), (33, 48), (256, 40)]
CONFIGS = [dict(), dict(random_scale=False, crop_size=32), dict(rotate=False), dict(fliplr=False, fliptb=False),
           dict(crop_size=50, scale_factor=4), dict(crop_size=48, scale_factor=2), dict(random_scale=False, crop_size=24,
                                                                                        rotate=False, fliplr=False, fliptb=False)]


@pytest.mark.parametrize("cfg", CONFIGS, ids=[str(i) for i in range(len(CONFIGS))])
def test_random_draws_follow_the_reference_order(tmp_path, cfg):
    """Same seed -> same consumption of Python's `random` as the oracle's __getitem__ (hence the same patch)."""
    pkg = _pkg()
    folder = make_images(str(tmp_path / "train"), SIZES)
    kw = dict(crop_size=32, scale_factor=4)
    kw.update(cfg)
    ora = O.TrainDatasetFromFolder([folder], **kw)
    ds = pkg.data.TrainDatasetFromFolder([folder], **kw)
    assert ds.image_filenames == ora.image_filenames and len(ds) == len(SIZES)
    for index, (w, h) in enumerate(SIZES):
        if not kw.get("random_scale", True) and min(w, h) < kw["crop_size"]:
            continue
        for seed in (0, 1, 7, 123):
            random.seed(seed)
            ora[index]
            want = random.getstate()
            random.seed(seed)
            ds.draw(w, h)
            assert random.getstate() == want, (index, seed)
    assert pkg.data.calculate_valid_crop_size(50, 4) == 48 and pkg.data.is_image_file("a.jpeg") and not pkg.data.is_image_file("a.gif")


def test_oracle_items_match_committed_fixture(tmp_path):
    """oracle/dataset_pil.py (Pillow) against tests/golden/dataset_r2.npz: the oracle — and the Pillow build under it —
    still produce the items they produced when the fixture was written."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden_dataset as G
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dataset_r2.npz"))
    folder = make_images(str(tmp_path / "train"), SIZES)
    for k, (index, seed, cfg) in enumerate(G.CASES):
        kw = dict(crop_size=32, scale_factor=4)
        kw.update(cfg)
        random.seed(seed)
        for name, t in zip(("lr", "hr", "bc"), O.TrainDatasetFromFolder([folder], **kw)[index]):
            assert torch.equal(t, torch.from_numpy(fx["train%d.%s" % (k, name)]).float().div(255)), (k, name)
    for name, t in zip(("lr", "hr", "bc"), O.TestDatasetFromFolder(folder, scale_factor=4)[0]):
        assert torch.equal(t, torch.from_numpy(fx["test0.%s" % name]).float().div(255)), name


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", CONFIGS + [dict(is_gray=True), dict(is_gray=True, random_scale=False, crop_size=32)],
                         ids=[str(i) for i in range(len(CONFIGS) + 2)])
def test_train_items_bit_equal_to_oracle(gpu, tmp_path, cfg):
    pkg = _pkg()
    folder = make_images(str(tmp_path / "train"), SIZES)
    kw = dict(crop_size=32, scale_factor=4)
    kw.update(cfg)
    ora = O.TrainDatasetFromFolder([folder], **kw)
    ds = pkg.data.TrainDatasetFromFolder([folder], device=gpu, **kw)
    n = 0
    for index, (w, h) in enumerate(SIZES):
        if not kw.get("random_scale", True) and min(w, h) < kw["crop_size"]:
            continue
        for seed in (3, 11, 12, 13, 14, 15):     # rotations 1..3 and all flip combinations occur
            random.seed(seed)
            want = ora[index]
            random.seed(seed)
            got = ds[index]
            for name, a, b in zip(("lr", "hr", "bc"), got, want):
                assert a.is_cuda and a.dtype == torch.float32 and tuple(a.shape) == tuple(b.shape), name
                assert torch.equal(a.cpu(), b), (name, index, seed, float((a.cpu() - b).abs().max()))
            n += 1
    assert n >= 18


@pytest.mark.gpu
@pytest.mark.parametrize("is_gray", [False, True])
def test_test_items_bit_equal_to_oracle(gpu, tmp_path, is_gray):
    pkg = _pkg()
    folder = make_images(str(tmp_path / "Set5"), SIZES)
    for sf in (2, 3, 4):
        ora = O.TestDatasetFromFolder(folder, is_gray=is_gray, scale_factor=sf)
        ds = pkg.data.TestDatasetFromFolder(folder, is_gray=is_gray, scale_factor=sf, device=gpu)
        for i in range(len(ora)):
            for a, b in zip(ds[i], ora[i]):
                assert torch.equal(a.cpu(), b)


@pytest.mark.gpu
def test_patch_loader_batches_and_trains(gpu, tmp_path):
    """PatchLoader: shuffled batches == the same items drawn one by one, decode prefetch included; and an EDSR trainer
    consumes it (get_training_set layout of data.py:33-51)."""
    pkg = _pkg()
    root = str(tmp_path / "Data")
    make_images(os.path.join(root, "DIV2K", "DIV2K_train_LR_bicubic", "X4"), SIZES + [(80, 90), (70, 70)])
    ds = pkg.data.get_training_set(root, ["DIV2K"], 32, 4, device=gpu)
    loader = pkg.data.PatchLoader(ds, batch_size=3, shuffle=True, num_threads=2, seed=9)
    assert len(loader) == 3
    order = torch.randperm(len(ds), generator=torch.Generator().manual_seed(9)).tolist()
    ora = O.TrainDatasetFromFolder(ds.image_filenames and [os.path.dirname(ds.image_filenames[0])], crop_size=32, scale_factor=4)
    random.seed(21)
    batches = list(loader)
    random.seed(21)
    want = [ora[i] for i in order]
    assert [b[0].shape[0] for b in batches] == [3, 3, 1]
    k = 0
    for lr, hr, bc in batches:
        assert lr.shape[1:] == (3, 8, 8) and hr.shape[1:] == (3, 32, 32) and bc.shape == hr.shape
        for j in range(lr.shape[0]):
            for a, b in zip((lr[j], hr[j], bc[j]), want[k]):
                assert torch.equal(a.cpu(), b)
            k += 1
    # second epoch: every image is resident in HBM now (no decode, no PCIe copy) -- same pixels, same draws, same batches
    assert all(ds.resident(i) is not None for i in range(len(ds))) and ds._resident_used == sum(
        int(np.prod(ds.resident(i).shape)) for i in range(len(ds)))
    loader2 = pkg.data.PatchLoader(ds, batch_size=3, shuffle=True, num_threads=2, seed=9)
    random.seed(21)
    again = list(loader2)
    for (a0, a1, a2), (b0, b1, b2) in zip(batches, again):
        assert torch.equal(a0, b0) and torch.equal(a1, b1) and torch.equal(a2, b2)
    # a byte budget of zero keeps nothing resident and gives the same batches
    ds0 = pkg.data.get_training_set(root, ["DIV2K"], 32, 4, device=gpu)
    ds0.resident_bytes = 0
    random.seed(21)
    cold = list(pkg.data.PatchLoader(ds0, batch_size=3, shuffle=True, num_threads=2, seed=9))
    assert not ds0._resident
    for (a0, a1, a2), (b0, b1, b2) in zip(batches, cold):
        assert torch.equal(a0, b0) and torch.equal(a1, b1) and torch.equal(a2, b2)
    # a trainer fed from the folder (2 epochs over 7 images)
    import main as cli
    from pytorch_super_resolution_model_collection_amd.sr_trainers import TRAINERS
    args = cli.parse_args(["--model_name", "EDSR", "--num_epochs", "2", "--save_epochs", "5", "--batch_size", "3",
                           "--crop_size", "32", "--lr", "1e-4", "--data_dir", root, "--save_dir", str(tmp_path / "out")])
    t = TRAINERS["EDSR"](args)
    hist = t.train()
    assert len(hist) == 2 and all(np.isfinite(hist))
    assert t.data_source == "folder"
    # test(): the reference's test folders (data.py:54-65) when they exist — per-image device PSNR, HR target = item[1]
    make_images(os.path.join(root, "Set5"), [(64, 48), (40, 40)])
    psnr = t.test()
    assert len(psnr) == 2 and all(np.isfinite(psnr)) and list(t.test_psnr) == ["Set5"]
