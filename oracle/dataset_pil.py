"""CPU ORACLE of the reference's image-folder datasets (/root/reference/dataset.py:22-149) — TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED BY THE REFERENCE: dataset.py does `from torchvision.transforms import *` and torchvision is not installed
in this image (the reference pins no version; `transforms.Scale` places it at torchvision 0.2), so the reference's own
classes cannot be imported here to generate vectors.  This module restates them on the real third-party library that
does the pixel work — Pillow, which IS installed — with the torchvision 0.2 transforms written out from their published
behaviour:
  Scale((w, h), interpolation)  ->  img.resize((w, h), interpolation)
  RandomCrop(size)              ->  if img.size == (size, size): img, else i = random.randint(0, h - th),
                                    j = random.randint(0, w - tw), img.crop((j, i, j + tw, i + th))   (0.2.0 get_params order)
  RandomHorizontalFlip()        ->  if random.random() < 0.5: img.transpose(FLIP_LEFT_RIGHT)
  ToTensor()                    ->  HWC uint8 -> CHW float32 .div(255)
  ToPILImage()                  ->  pic.mul(255).byte() -> HWC uint8
The calls to Python's `random` happen in the reference's order, so `random.seed(s); ds[i]` selects the same patch.
Pinned against Pillow itself (the resampling, rotations and flips are executed by Pillow); the committed fixture
tests/golden/dataset_r2.npz was generated from THIS module (tests/golden/make_golden_dataset.py)."""
import random
from os import listdir
from os.path import join

import numpy as np
import torch
from PIL import Image


def is_image_file(filename):  # dataset.py:9-10
    return any(filename.endswith(extension) for extension in [".png", ".jpg", ".jpeg", ".bmp"])


def load_img(filepath):  # dataset.py:13-15
    return Image.open(filepath).convert('RGB')


def calculate_valid_crop_size(crop_size, scale_factor):  # dataset.py:18-19
    return crop_size - (crop_size % scale_factor)


def to_tensor(img):
    a = np.array(img, dtype=np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).float().div(255)


def to_pil(t):
    a = t.mul(255).byte().numpy().transpose(1, 2, 0)
    return Image.fromarray(a[:, :, 0], mode="L") if a.shape[2] == 1 else Image.fromarray(a, mode="RGB")


def _triple(img, hr_w, hr_h, lr_w, lr_h):
    hr_img = to_tensor(img.resize((hr_w, hr_h), Image.BICUBIC))               # dataset.py:90-91
    lr_img = to_tensor(img.resize((lr_w, lr_h), Image.BICUBIC))               # :94-95
    bc_img = to_tensor(to_pil(lr_img).resize((hr_w, hr_h), Image.BICUBIC))    # :98-99
    return lr_img, hr_img, bc_img


class TrainDatasetFromFolder(object):  # dataset.py:22-104
    def __init__(self, image_dirs, is_gray=False, random_scale=True, crop_size=128, rotate=True, fliplr=True,
                 fliptb=True, scale_factor=4):
        self.image_filenames = []
        for image_dir in image_dirs:
            self.image_filenames.extend(join(image_dir, x) for x in sorted(listdir(image_dir)) if is_image_file(x))
        self.is_gray, self.random_scale, self.crop_size = is_gray, random_scale, crop_size
        self.rotate, self.fliplr, self.fliptb, self.scale_factor = rotate, fliplr, fliptb, scale_factor

    def __getitem__(self, index):
        img = load_img(self.image_filenames[index])
        self.crop_size = calculate_valid_crop_size(self.crop_size, self.scale_factor)
        hr_img_w = hr_img_h = self.crop_size
        lr_img_w, lr_img_h = hr_img_w // self.scale_factor, hr_img_h // self.scale_factor
        if self.random_scale:                                                  # :51-63
            eps = 1e-3
            ratio = random.randint(5, 10) * 0.1
            if hr_img_w * ratio < self.crop_size:
                ratio = self.crop_size / hr_img_w + eps
            if hr_img_h * ratio < self.crop_size:
                ratio = self.crop_size / hr_img_h + eps
            img = img.resize((int(hr_img_w * ratio), int(hr_img_h * ratio)), Image.BICUBIC)
        w, h = img.size                                                        # RandomCrop, :65-67
        th = tw = self.crop_size
        if not (w == tw and h == th):
            i = random.randint(0, h - th)
            j = random.randint(0, w - tw)
            img = img.crop((j, i, j + tw, i + th))
        if self.rotate:                                                        # :70-72
            rv = random.randint(1, 3)
            img = img.rotate(90 * rv, expand=True)
        if self.fliplr:                                                        # :75-77
            if random.random() < 0.5:
                img = img.transpose(Image.FLIP_LEFT_RIGHT)
        if self.fliptb:                                                        # :80-82
            if random.random() < 0.5:
                img = img.transpose(Image.FLIP_TOP_BOTTOM)
        if self.is_gray:                                                       # :85-86
            img = img.convert('YCbCr')
        return _triple(img, hr_img_w, hr_img_h, lr_img_w, lr_img_h)

    def __len__(self):
        return len(self.image_filenames)


class TestDatasetFromFolder(object):  # dataset.py:106-149
    def __init__(self, image_dir, is_gray=False, scale_factor=4):
        self.image_filenames = [join(image_dir, x) for x in sorted(listdir(image_dir)) if is_image_file(x)]
        self.is_gray, self.scale_factor = is_gray, scale_factor

    def __getitem__(self, index):
        img = load_img(self.image_filenames[index])
        w, h = img.size
        hr_img_w = calculate_valid_crop_size(w, self.scale_factor)
        hr_img_h = calculate_valid_crop_size(h, self.scale_factor)
        if self.is_gray:
            img = img.convert('YCbCr')
        return _triple(img, hr_img_w, hr_img_h, hr_img_w // self.scale_factor, hr_img_h // self.scale_factor)

    def __len__(self):
        return len(self.image_filenames)
