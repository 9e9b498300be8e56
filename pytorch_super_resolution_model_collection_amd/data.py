"""The reference's image-folder input pipeline (dataset.py:22-149, data.py:33-65) re-designed for the MI355X:

  host   : file listing, PNG/JPEG decode (Pillow, in a thread pool — decode releases the GIL), nothing else;
  PCIe   : the decoded uint8 image goes through a pinned staging buffer and an asynchronous H2D copy on the loader's own
           HIP stream (1 byte per sample instead of the 4-byte floats the reference's DataLoader hands over);
  device : every transform of `TrainDatasetFromFolder.__getitem__` — random bicubic re-scale (Pillow's 8-bit two-pass
           resampler, bit-exact: csrc/resize_pil.hip), RandomCrop, 90-degree rotations, flips, the HR / LR bicubic
           resizes, ToTensor and the ToPILImage -> bicubic -> ToTensor round trip of the "bicubic" image — as a handful of
           kernels per patch plus three batched resizes per minibatch.

Same classes / functions / argument names as the reference (`TrainDatasetFromFolder`, `TestDatasetFromFolder`,
`get_training_set`, `get_test_set`), same `(lr_img, hr_img, bc_img)` items, and the SAME calls to Python's `random` in the
same order, so `random.seed(s); ds[i]` picks the patch the reference would (parity: oracle/dataset_pil.py, bit-exact).
Reference quirk kept: with random_scale the whole image is first resized to crop x crop — the
"random crop" is then a no-op.  `is_gray` (convert('YCbCr'), dataset.py:85-86) is done by Pillow on the host on the
augmented 8-bit patch (one small D2H/H2D per patch, gray mode only): its integer colour matrix is Pillow's own.
Not provided: the BSDS300 HTTP download of data.py:9-30 (no network code in this package).
"""
import ctypes
import os
import random
from concurrent.futures import ThreadPoolExecutor
from os import listdir
from os.path import join

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr

BICUBIC = 3  # PIL.Image.BICUBIC / srk_interp


def is_image_file(filename):
    """dataset.py:9-10"""
    return any(filename.endswith(extension) for extension in [".png", ".jpg", ".jpeg", ".bmp"])


def load_img(filepath):
    """dataset.py:13-15 — host decode to an RGB uint8 array [H, W, 3]."""
    from PIL import Image
    return np.asarray(Image.open(filepath).convert('RGB'), dtype=np.uint8)


def calculate_valid_crop_size(crop_size, scale_factor):
    """dataset.py:18-19"""
    return crop_size - (crop_size % scale_factor)


class _Null(object):
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_NULL = _Null()


def _stream_ptr(stream):
    return ctypes.c_void_p(stream.cuda_stream)


class _Device(object):
    """Device-side transforms on 8-bit images (thin wrappers over the C ABI), all on one HIP stream."""

    def __init__(self, device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the input pipeline's transforms run on the GPU (got device %s); there is no CPU fallback"
                               % (device,))
        self.lib = _lib.load()
        self.stream = torch.cuda.Stream(device=self.device)
        self._sp = _stream_ptr(self.stream)
        self._ws_bytes = {}
        self._slab_buf, self._slab_off, self._depth = None, 0, 0

    # Every helper below runs on the loader stream.  PatchLoader enters the stream ONCE per batch (`batch()`); a
    # torch.cuda.stream context per call costs ~10 us of host time, three times per patch.
    def _on_stream(self):
        return _NULL if self._depth else torch.cuda.stream(self.stream)

    def batch(self):
        """Context of one batch: the loader stream is current, workspaces come from the start of the slab again."""
        dev = self

        class _Ctx(object):
            def __enter__(self_):
                self_.cm = torch.cuda.stream(dev.stream)
                self_.cm.__enter__()
                dev._depth += 1
                dev._slab_off = 0

            def __exit__(self_, *a):
                dev._depth -= 1
                return self_.cm.__exit__(*a)

        return _Ctx()

    def _slab(self, nbytes):
        nbytes = (nbytes + 255) & ~255
        if self._slab_buf is None or self._slab_off + nbytes > self._slab_buf.numel():
            # (out of room in the middle of a batch: a fresh slab; the old one is released to the loader stream's pool,
            #  whose reuse is ordered behind the kernels still reading it)
            self._slab_buf = torch.empty(max(64 << 20, 4 * nbytes), dtype=torch.uint8, device=self.device)
            self._slab_off = 0
        v = self._slab_buf[self._slab_off:self._slab_off + nbytes]
        self._slab_off += nbytes
        if not self._depth:
            self._slab_off = 0   # outside a batch context every call may reuse the slab (stream order protects it)
        return v

    # Pinned staging buffers are a ring allocated once and reused (pinning a fresh buffer per image is a driver call
    # of several milliseconds that also serialises against the device: 163 patches/s with it, see tools/loader_bench.py)
    RING = 64

    def upload(self, hwc):
        """HWC uint8 numpy -> device uint8 tensor through pinned memory, async on the loader stream."""
        if not hasattr(self, "_ring"):
            self._ring, self._ring_ev, self._ring_k = [None] * self.RING, [None] * self.RING, 0
        i = self._ring_k
        self._ring_k = (i + 1) % self.RING
        n = int(hwc.size)
        if self._ring_ev[i] is not None:
            self._ring_ev[i].synchronize()   # the copy that last used this slot (RING uploads ago) is done
        if self._ring[i] is None or self._ring[i].numel() < n:
            self._ring[i] = torch.empty(max(n, 1 << 20), dtype=torch.uint8).pin_memory()
        pinned = self._ring[i][:n].view(hwc.shape)
        np.copyto(pinned.numpy(), hwc)
        with self._on_stream():
            d = torch.empty(hwc.shape, dtype=torch.uint8, device=self.device)
            d.copy_(pinned, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self._ring_ev[i] = ev
        return d

    def resize(self, x, strides, planes, h, w, oh, ow, out_float):
        """Image.resize((ow, oh), BICUBIC) of `planes` 8-bit planes addressed by element strides (plane, row, pixel)."""
        lib = self.lib
        with self._on_stream():
            y = torch.empty((planes, oh, ow), dtype=torch.float32 if out_float else torch.uint8, device=self.device)
            key = (planes, h, w, oh, ow)
            nbytes = self._ws_bytes.get(key)
            if nbytes is None:
                nbytes = self._ws_bytes[key] = max(int(lib.srk_img_resize_u8_workspace_bytes(planes, h, w, oh, ow, BICUBIC)), 256)
            # one workspace per call (the calls of a batch are in flight together on the loader stream), drawn from a
            # slab that is re-used batch after batch
            ws = self._slab(nbytes)
            check(lib.srk_img_resize_u8(ptr(x), strides[0], strides[1], strides[2], ptr(y), int(out_float), planes, h, w,
                                        oh, ow, BICUBIC, ptr(ws), nbytes, self._sp), "srk_img_resize_u8")
        return y

    def augment(self, x, strides, c, h, w, crop, rot, fliplr, fliptb, out=None):
        """crop (x0, y0, cw, ch) -> rot x 90 deg ccw -> flips; planar uint8 [c, oh, ow]."""
        x0, y0, cw, ch = crop
        oh, ow = (cw, ch) if rot & 1 else (ch, cw)
        with self._on_stream():
            y = out if out is not None else torch.empty((c, oh, ow), dtype=torch.uint8, device=self.device)
            check(self.lib.srk_patch_augment_u8(ptr(x), strides[0], strides[1], strides[2], ptr(y), c, h, w, x0, y0, cw, ch,
                                                rot, int(fliplr), int(fliptb), self._sp), "srk_patch_augment_u8")
        return y

    def patch(self, img, c, h, w, scale, crop, rot, fliplr, fliptb, out=None):
        """Decoded interleaved image on the device -> augmented planar 8-bit patch (srk_patch_from_image_u8)."""
        x0, y0, cw, ch = crop
        oh, ow = (cw, ch) if rot & 1 else (ch, cw)
        sh, sw = (scale[1], scale[0]) if scale is not None else (0, 0)
        lib = self.lib
        with self._on_stream():
            y = out if out is not None else torch.empty((c, oh, ow), dtype=torch.uint8, device=self.device)
            key = ("patch", c, h, w, sh, sw)
            nbytes = self._ws_bytes.get(key)
            if nbytes is None:
                nbytes = self._ws_bytes[key] = int(lib.srk_patch_from_image_u8_workspace_bytes(c, h, w, sh, sw))
            ws = self._slab(nbytes)
            check(lib.srk_patch_from_image_u8(ptr(img), c, h, w, sh, sw, x0, y0, cw, ch, rot, int(fliplr), int(fliptb),
                                              ptr(y), ptr(ws), nbytes, self._sp), "srk_patch_from_image_u8")
        return y

    def bicubic_of_lr(self, lr, oh, ow):
        """ToPILImage -> Scale -> ToTensor on the float LR batch (dataset.py:98-99) = utils.img_interp's kernel."""
        lib = self.lib
        n, c, h, w = lr.shape
        with self._on_stream():
            y = torch.empty((n, c, oh, ow), dtype=torch.float32, device=self.device)
            nbytes = int(lib.srk_img_interp_workspace_bytes(n, c, h, w, oh, ow, BICUBIC))
            ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=self.device)
            check(lib.srk_img_interp(ptr(lr), ptr(y), n, c, h, w, oh, ow, BICUBIC, ptr(ws), ws.numel(),
                                     self._sp), "srk_img_interp")
        return y

    def hand_over(self, *tensors):
        """Make the consumer's current stream wait for the loader stream; tensors become usable there."""
        cur = torch.cuda.current_stream(self.device)
        cur.wait_stream(self.stream)
        for t in tensors:
            t.record_stream(cur)
        return tensors


_HWC = lambda h, w, c: (1, w * c, c)       # element strides (plane, row, pixel) of an interleaved image
_CHW = lambda h, w: (h * w, w, 1)


class TrainDatasetFromFolder(object):
    """dataset.py:22-104 with device-side transforms.  `__getitem__` returns (lr_img, hr_img, bc_img) CUDA tensors."""

    def __init__(self, image_dirs, is_gray=False, random_scale=True, crop_size=128, rotate=True, fliplr=True,
                 fliptb=True, scale_factor=4, device="cuda"):
        self.image_filenames = []
        for image_dir in image_dirs:
            self.image_filenames.extend(join(image_dir, x) for x in sorted(listdir(image_dir)) if is_image_file(x))
        self.is_gray, self.random_scale, self.crop_size = is_gray, random_scale, crop_size
        self.rotate, self.fliplr, self.fliptb, self.scale_factor = rotate, fliplr, fliptb, scale_factor
        self._device, self._dev = device, None
        # Decoded images stay RESIDENT IN HBM as 8-bit tensors once they have been uploaded (DIV2K's 800 LR training images
        # are ~0.4 GB of 288): from the second epoch on a patch costs no PNG decode, no PCIe copy and one kernel call.  The
        # transforms still start from the same 8-bit pixels, so nothing changes numerically.  `resident_bytes` bounds it.
        self.resident_bytes = int(float(os.environ.get("SRK_DATA_RESIDENT_GB", "16")) * (1 << 30))
        self._resident, self._resident_used = {}, 0

    @property
    def dev(self):
        if self._dev is None:   # (created on first use: listing a folder and drawing parameters need no GPU)
            self._dev = _Device(self._device)
        return self._dev

    def __len__(self):
        return len(self.image_filenames)

    # -- the reference's random draws, in its order (dataset.py:51-82) -----------------------------------------
    def draw(self, w, h):
        """Consumes Python's `random` exactly like the reference's __getitem__ for a w x h image.  Returns
        (scale_wh or None, (x0, y0), rot_k, fliplr, fliptb)."""
        self.crop_size = calculate_valid_crop_size(self.crop_size, self.scale_factor)
        crop = self.crop_size
        scale = None
        if self.random_scale:
            eps = 1e-3
            ratio = random.randint(5, 10) * 0.1
            if crop * ratio < crop:
                ratio = crop / crop + eps
            if crop * ratio < crop:       # (second test of the reference, on the height: same numbers)
                ratio = crop / crop + eps
            scale = (int(crop * ratio), int(crop * ratio))
            w, h = scale
        x0 = y0 = 0
        if not (w == crop and h == crop):   # torchvision RandomCrop: no draw when the sizes already match
            if w < crop or h < crop:
                raise ValueError("image %dx%d is smaller than the crop size %d" % (w, h, crop))
            y0 = random.randint(0, h - crop)
            x0 = random.randint(0, w - crop)
        rot = random.randint(1, 3) if self.rotate else 0
        fl = (random.random() < 0.5) if self.fliplr else False
        ft = (random.random() < 0.5) if self.fliptb else False
        return scale, (x0, y0), rot, fl, ft

    def patch_u8(self, hwc, params, out=None):
        """Decoded image (numpy HWC uint8) + draws -> augmented 8-bit patch [3, crop, crop] on the device."""
        dev, crop = self.dev, self.crop_size
        scale, (x0, y0), rot, fl, ft = params
        if isinstance(hwc, torch.Tensor):    # resident image (see __init__)
            img = hwc
            h, w, c = img.shape
        else:
            h, w, c = hwc.shape
            img = dev.upload(hwc)
        if not self.is_gray:   # rescale + crop / rotation / flips in ONE call, straight into the batch buffer
            return dev.patch(img, c, h, w, scale, (x0, y0, crop, crop), rot, fl, ft, out)
        strides = _HWC(h, w, c)
        if scale is not None:
            img = dev.resize(img, strides, c, h, w, scale[1], scale[0], out_float=False)
            h, w = scale[1], scale[0]
            strides = _CHW(h, w)
        patch = dev.augment(img, strides, c, h, w, (x0, y0, crop, crop), rot, fl, ft, out=out)
        if self.is_gray:   # Pillow's integer RGB -> YCbCr matrix on the host (see module docstring)
            from PIL import Image
            with torch.cuda.stream(dev.stream):
                host = patch.permute(1, 2, 0).contiguous().cpu().numpy()
            ycc = np.asarray(Image.fromarray(host, "RGB").convert("YCbCr"), dtype=np.uint8)
            with torch.cuda.stream(dev.stream):
                patch.copy_(dev.upload(ycc).permute(2, 0, 1))
        return patch

    def resident(self, index):
        """The image's 8-bit device tensor if it is resident, else None."""
        return self._resident.get(index)

    def keep_resident(self, index, hwc):
        """Upload a decoded image and keep it in HBM (within the byte budget); returns what patch_u8 should consume."""
        n = int(hwc.size)
        if self._resident_used + n > self.resident_bytes:
            return hwc
        img = self.dev.upload(hwc)
        self._resident[index] = img
        self._resident_used += n
        return img

    def finish(self, patches):
        """[B, 3, crop, crop] 8-bit patches -> (lr, hr, bc) float batches: the three resizes of dataset.py:89-99."""
        dev, crop, sf = self.dev, self.crop_size, self.scale_factor
        b, c = patches.shape[0], patches.shape[1]
        lr_sz = crop // sf
        st = _CHW(crop, crop)
        hr = dev.resize(patches, st, b * c, crop, crop, crop, crop, out_float=True).view(b, c, crop, crop)
        lr = dev.resize(patches, st, b * c, crop, crop, lr_sz, lr_sz, out_float=True).view(b, c, lr_sz, lr_sz)
        bc = dev.bicubic_of_lr(lr, crop, crop)
        return lr, hr, bc

    def __getitem__(self, index):
        hwc = self.resident(index)
        if hwc is None:
            hwc = self.keep_resident(index, load_img(self.image_filenames[index]))
        params = self.draw(hwc.shape[1], hwc.shape[0])
        patch = self.patch_u8(hwc, params)
        lr, hr, bc = self.finish(patch.unsqueeze(0))
        return self.dev.hand_over(lr[0], hr[0], bc[0])


class TestDatasetFromFolder(object):
    """dataset.py:106-149 with device-side resizes."""

    def __init__(self, image_dir, is_gray=False, scale_factor=4, device="cuda"):
        self.image_filenames = [join(image_dir, x) for x in sorted(listdir(image_dir)) if is_image_file(x)]
        self.is_gray, self.scale_factor = is_gray, scale_factor
        self._device, self._dev = device, None

    @property
    def dev(self):
        if self._dev is None:
            self._dev = _Device(self._device)
        return self._dev

    def __len__(self):
        return len(self.image_filenames)

    def __getitem__(self, index):
        from PIL import Image
        dev, sf = self.dev, self.scale_factor
        if self.is_gray:
            hwc = np.asarray(Image.open(self.image_filenames[index]).convert('RGB').convert('YCbCr'), dtype=np.uint8)
        else:
            hwc = load_img(self.image_filenames[index])
        h, w, c = hwc.shape
        hr_w, hr_h = calculate_valid_crop_size(w, sf), calculate_valid_crop_size(h, sf)
        lr_w, lr_h = hr_w // sf, hr_h // sf
        img = dev.upload(hwc)
        st = _HWC(h, w, c)
        hr = dev.resize(img, st, c, h, w, hr_h, hr_w, out_float=True)
        lr = dev.resize(img, st, c, h, w, lr_h, lr_w, out_float=True)
        bc = dev.bicubic_of_lr(lr.unsqueeze(0), hr_h, hr_w)[0]
        return dev.hand_over(lr, hr, bc)


def get_training_set(data_dir, datasets, crop_size, scale_factor, is_gray=False, device="cuda"):
    """data.py:33-51 (directory layout and augmentation switches; 'bsds300' must already be on disk)."""
    train_dir = []
    for dataset in datasets:
        if dataset == 'bsds300':
            train_dir.append(join(data_dir, "BSDS300/images", "train"))
        elif dataset == 'DIV2K':
            train_dir.append(join(data_dir, dataset, 'DIV2K_train_LR_bicubic/X4'))
        else:
            train_dir.append(join(data_dir, dataset))
    return TrainDatasetFromFolder(train_dir, is_gray=is_gray, random_scale=True, crop_size=crop_size, rotate=True,
                                  fliplr=True, fliptb=True, scale_factor=scale_factor, device=device)


def get_test_set(data_dir, dataset, scale_factor, is_gray=False, device="cuda"):
    """data.py:54-65"""
    if dataset == 'bsds300':
        test_dir = join(data_dir, "BSDS300/images", "test")
    elif dataset == 'DIV2K':
        test_dir = join(data_dir, dataset, 'DIV2K_test_LR_bicubic/X4')
    else:
        test_dir = join(data_dir, dataset)
    return TestDatasetFromFolder(test_dir, is_gray=is_gray, scale_factor=scale_factor, device=device)


class PatchLoader(object):
    """DataLoader(dataset=train_set, num_workers=num_threads, batch_size=B, shuffle=True) of the reference
    (edsr.py:75-77), MI355X-first: `num_threads` host threads only DECODE (the next batch is decoded while the current
    one trains), pixels cross PCIe as uint8 through pinned memory, all transforms run on the loader's HIP stream, and a
    batch costs 2-3 small launches per patch plus three batched resizes.  Yields (lr, hr, bc) CUDA batches."""

    def __init__(self, dataset, batch_size, shuffle=True, num_threads=4, drop_last=False, seed=None, background=True,
                 rank=0, world=1):
        self.ds, self.batch_size, self.shuffle, self.drop_last = dataset, int(batch_size), shuffle, drop_last
        # data parallel (world > 1): every rank draws the SAME permutation per epoch (so `seed` must be common; default
        # 0) and takes its own strided 1/world of it, wrapped around so that all ranks see the same number of batches
        self.rank, self.world = int(rank), max(1, int(world))
        if self.world > 1 and seed is None:
            seed = 0
        # background: batches are produced by a loader thread up to two ahead of the consumer (its host work -- draws,
        # one call per patch, three batched resizes -- overlaps with the consumer launching / replaying the train step)
        self.background = bool(background) and os.environ.get("SRK_LOADER_THREAD", "1") != "0"
        self.pool = ThreadPoolExecutor(max_workers=max(1, int(num_threads)))
        self.gen = torch.Generator()
        if seed is not None:
            self.gen.manual_seed(int(seed))

    def _shard_len(self):
        return -(-len(self.ds) // self.world)      # ceil: short ranks wrap around (torch's DistributedSampler rule)

    def __len__(self):
        n = self._shard_len()
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def _order(self):
        """This rank's image indices of one epoch."""
        n = len(self.ds)
        order = torch.randperm(n, generator=self.gen).tolist() if self.shuffle else list(range(n))
        if self.world > 1:
            per = self._shard_len()
            order = (order + order[:per * self.world - n])[self.rank:per * self.world:self.world]
        return order

    def _batches(self):
        order = self._order()
        n = len(order)
        for i in range(0, n, self.batch_size):
            idx = order[i:i + self.batch_size]
            if len(idx) < self.batch_size and self.drop_last:
                return
            yield idx

    def _decode(self, idx):
        res = getattr(self.ds, "resident", None)
        return [(i, None if (res is not None and res(i) is not None) else self.pool.submit(load_img, self.ds.image_filenames[i]))
                for i in idx]

    def __iter__(self):
        ds = self.ds
        if not (self.background and isinstance(ds, TrainDatasetFromFolder)):
            for tensors, ev in self._produce():
                yield ds.dev.hand_over(*tensors) if ev is not None else tensors
            return
        import queue
        import threading
        q, stop, END = queue.Queue(maxsize=2), threading.Event(), object()

        def put(item):
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.05)
                    return True
                except queue.Full:
                    pass
            return False

        def work():
            try:
                for item in self._produce():
                    if not put(item):
                        return
                put(END)
            except BaseException as e:   # noqa: BLE001 -- re-raised in the consumer
                put(e)

        th = threading.Thread(target=work, name="srk-loader", daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is END:
                    break
                if isinstance(item, BaseException):
                    raise item
                tensors, ev = item
                cur = torch.cuda.current_stream(ds.dev.device)
                cur.wait_event(ev)            # the consumer's stream waits for THIS batch only, not for the loader stream
                for t in tensors:
                    t.record_stream(cur)
                yield tensors
        finally:
            stop.set()

    def _produce(self):
        """Generator of (batch tensors, event recorded on the loader stream behind them); event None = nothing to wait
        for (test items are handed over one by one)."""
        ds = self.ds
        batches = list(self._batches())
        # decode ahead: enough batches in flight to keep every decode thread busy while the current batch is uploaded
        # and transformed (one batch ahead leaves 32 threads half idle at batch 16: 2.6 k -> patches/s decode-bound)
        depth = max(1, -(-self.pool._max_workers // max(1, self.batch_size))) + 1
        from collections import deque
        pending = deque(self._decode(b) for b in batches[:depth])
        for k, idx in enumerate(batches):
            images = []
            for i, f in pending.popleft():
                if f is None:
                    images.append(ds.resident(i))
                elif isinstance(ds, TrainDatasetFromFolder):
                    with ds.dev.batch():
                        images.append(ds.keep_resident(i, f.result()))
                else:
                    images.append(f.result())
            if k + depth < len(batches):
                pending.append(self._decode(batches[k + depth]))   # overlaps with the work below
            if isinstance(ds, TrainDatasetFromFolder):
                crop = calculate_valid_crop_size(ds.crop_size, ds.scale_factor)
                with ds.dev.batch():
                    patches = torch.empty((len(idx), 3, crop, crop), dtype=torch.uint8, device=ds.dev.device)
                    for j, hwc in enumerate(images):
                        ds.patch_u8(hwc, ds.draw(hwc.shape[1], hwc.shape[0]), out=patches[j])
                    out = ds.finish(patches)
                    ev = torch.cuda.Event()
                    ev.record(ds.dev.stream)
                yield out, ev
            else:   # test images differ in size: one item per batch entry, stacked only when they agree
                items = [ds[i] for i in idx]
                yield tuple(torch.stack(t) if len(set(x.shape for x in t)) == 1 else list(t) for t in zip(*items)), None
