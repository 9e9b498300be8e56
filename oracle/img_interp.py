"""CPU oracle for utils.img_interp (/root/reference/utils.py:242-269).  TEST INFRASTRUCTURE ONLY.

The reference resizes image batches through torchvision transforms around Pillow:

    transforms.Compose([ToPILImage(), Scale((w, h), interpolation), ToTensor()])          (utils.py:255-259, 265-268)

torchvision is not installed in this image (and the reference pins no version), so the three
transforms are restated from their published behaviour for float CHW tensors:
  * ToPILImage:  pic.mul(255).byte() -> HWC uint8 -> Image.fromarray (mode 'RGB' for 3 channels, 'L' for 1);
  * Scale/Resize((w, h), interpolation):  img.resize((w, h), interpolation)  — Pillow, which IS installed
    (12.2 here): the resampling itself is executed by the real third-party library the reference uses;
  * ToTensor:  HWC uint8 -> CHW float32, .div(255).
Pinned by tests/golden/img_interp.npz (written by tests/golden/make_golden_interp.py from this module, i.e. from
Pillow's output) and, on the GPU box, re-checked against Pillow live."""
import numpy as np
import torch
from PIL import Image

_PIL = {"bicubic": Image.BICUBIC, "bilinear": Image.BILINEAR, "nearest": Image.NEAREST}


def _one(img, tw, th, interpolation):
    """img: float CHW tensor in [0,1] -> float CHW tensor (utils.py:265-268)."""
    arr = img.mul(255).byte().numpy()                     # ToPILImage: truncation
    hwc = np.transpose(arr, (1, 2, 0))
    pil = Image.fromarray(hwc[:, :, 0], mode="L") if hwc.shape[2] == 1 else Image.fromarray(hwc, mode="RGB")
    pil = pil.resize((tw, th), _PIL[interpolation])       # Scale((w, h), interpolation)
    out = np.array(pil, dtype=np.uint8)
    if out.ndim == 2:
        out = out[:, :, None]
    return torch.from_numpy(np.ascontiguousarray(np.transpose(out, (2, 0, 1)))).float().div(255)   # ToTensor


def img_interp(imgs, scale_factor, interpolation="bicubic"):
    """utils.py:242-269: 4-D [B,C,H,W] or 3-D [C,H,W]; target size int(H*s) x int(W*s)."""
    size = list(imgs.shape)
    if len(size) == 4:
        th, tw = int(size[2] * scale_factor), int(size[3] * scale_factor)
        out = torch.empty(size[0], size[1], th, tw, dtype=torch.float32)
        for i, img in enumerate(imgs):
            out[i] = _one(img, tw, th, interpolation)
        return out
    th, tw = int(size[1] * scale_factor), int(size[2] * scale_factor)
    return _one(imgs, tw, th, interpolation)
