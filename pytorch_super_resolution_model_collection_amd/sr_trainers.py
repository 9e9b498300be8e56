"""Trainer objects with the reference's surface — MODEL(args).train() / test() / test_single(fn) /
save_model(epoch) / load_model() — for SRCNN, ESPCN, FSRCNN, VDSR, EDSR, LapSRN and SRGAN
(srcnn.py:32-281, espcn.py:32-281, fsrcnn.py:58-307, vdsr.py:39-301, edsr.py:48-351,
lapsrn.py:88-349, srgan.py:93-528), on the MI355X hot path.

Kept from the reference: per-model hyper-parameters hard-coded in train() (e.g. EDSR base_filter 64 /
16 residuals, edsr.py:87; VDSR momentum 0.9 / wd 1e-4 / clip 0.4, vdsr.py:86-90,149), the epoch-wise
LR decay rules, the checkpoint file names and that checkpoints are weights-only state_dict pickles.
Not kept (out of scope): TF1 logging, PNG/plot side effects and the per-iteration host
sync (`loss.data[0]`).  Training data comes from a `loader` argument (any iterable of tensor tuples in
the reference's (lr, hr, bicubic) order), else from the reference's image folders under `data_dir`
(data.PatchLoader: decode on host threads, transforms on the GPU) when they exist, else from seeded
synthetic patches of the configured crop size.
"""
import os

import torch

from . import dp as dpmod
from . import models, ops, optim, trainers, utils


def synthetic_loader(kind, args, steps, device, seed=1234):
    """Seeded random (input, target...) batches with the shapes each reference train loop feeds."""
    g = torch.Generator().manual_seed(seed)
    b, c, r = args.batch_size, args.num_channels, args.scale_factor
    lr = args.crop_size // r
    hr = lr * r  # the reference crops HR patches to a multiple of the scale (dataset.py calculate_valid_crop_size)
    for _ in range(steps):
        # every reference dataset yields (LR input, HR target) image batches in [0,1] (dataset.py:51-99); the
        # per-model pre-steps (bicubic up/down-sampling, shaving) run on the device in _Trainer.prepare
        if kind == "fsrcnn":            # output r(H-5)+4 = target shaved by 2r (fsrcnn.py:143-150)
            yield torch.rand(b, c, lr, lr, generator=g).to(device), torch.rand(b, c, r * (lr - 5) + 4, r * (lr - 5) + 4, generator=g).to(device)
        elif kind == "espcn":           # net output r(H-8) (the reference's own target is inconsistent, App. B-2)
            yield torch.rand(b, c, lr, lr, generator=g).to(device), torch.rand(b, c, r * (lr - 8), r * (lr - 8), generator=g).to(device)
        else:                           # srcnn / vdsr / lapsrn / edsr / srgan: (lr, hr)
            yield torch.rand(b, c, lr, lr, generator=g).to(device), torch.rand(b, c, hr, hr, generator=g).to(device)


# Epoch-wise learning-rate decay of each reference trainer: kind -> (every N epochs, divide by).  Applied at the top of
# epoch e when (e + 1) % N == 0, to every param group of every optimizer of the trainer.
#   vdsr.py:127-129  /10 every 20     edsr.py:131-133  /2 every 40     lapsrn.py:173-175  /10 every 100
#   srgan.py:239-244 /10 every 20 (G and D)            srcnn.py / espcn.py / fsrcnn.py: no decay
LR_DECAY = {"vdsr": (20, 10.0), "edsr": (40, 2.0), "lapsrn": (100, 10.0), "srgan": (20, 10.0)}


def apply_lr_decay(kind, epoch, *optimizers):
    """The reference's `if (epoch+1) % N == 0: param_group['lr'] /= F` (see LR_DECAY).  Returns True if it decayed."""
    rule = LR_DECAY.get(kind)
    if rule is None or (epoch + 1) % rule[0] != 0:
        return False
    for opt in optimizers:
        for g in opt.param_groups:
            g['lr'] = g['lr'] / rule[1]
    return True


class _Trainer(object):
    kind = None

    def __init__(self, args):
        # the reference copies args field by field (edsr.py:49-65)
        for k in ("model_name", "train_dataset", "test_dataset", "crop_size", "num_threads", "num_channels",
                  "scale_factor", "num_epochs", "save_epochs", "batch_size", "test_batch_size", "lr", "data_dir",
                  "save_dir", "gpu_mode"):
            setattr(self, k, getattr(args, k, None))
        self.args = args
        self.steps_per_epoch = getattr(args, "steps_per_epoch", 8)
        if not torch.cuda.is_available():
            raise RuntimeError("the MI355X hot path needs a GPU (gpu_mode=False has no CPU fallback; see oracle/)")
        self.rank, self.world, self.local = dpmod.init_from_env()
        torch.cuda.set_device(self.local)
        self.device = torch.device("cuda", self.local)
        self.model = None

    # -- overridables --------------------------------------------------------------------------
    def build_model(self):
        raise NotImplementedError

    def lr_decay(self, epoch, opt):
        apply_lr_decay(self.kind, epoch, opt)

    def prepare(self, inp, target):
        """(input, target) of the data loader -> the tensors the train step consumes, on the device:
          SRCNN   y = img_interp(input, r) (bicubic), x = shave(target, 8)               (srcnn.py:116-125)
          VDSR    y = img_interp(input, r), x = target                                    (vdsr.py:133-142)
          LapSRN  y = input, x_coarse = img_interp(target, 1/r*2), x_finer = target       (lapsrn.py:179-188)
        utils.img_interp is the bit-exact GPU form of the reference's per-image PIL loop."""
        if self.kind == "srcnn":
            return utils.img_interp(inp, self.scale_factor), utils.shave(target, 8).contiguous()
        if self.kind == "vdsr":
            return utils.img_interp(inp, self.scale_factor), target
        if self.kind == "lapsrn":
            return inp, utils.img_interp(target, 1 / self.scale_factor * 2), target
        return inp, target

    def load_dataset(self, dataset, is_train=True):
        """edsr.py:67-84 / srgan.py:110-129: the image-folder loaders of data.py behind the reference's directory layout
        (data_dir/<dataset>/..., DIV2K_train_LR_bicubic/X4 for DIV2K), as data.PatchLoader (decode on host threads,
        uint8 over PCIe through pinned memory, every transform on the GPU).  Under data parallelism every rank draws the
        same per-epoch permutation and takes its own 1/world of it (PatchLoader rank / world).
        A missing or empty TRAINING folder raises, as the reference's DataLoader would -- unless the run asked for
        seeded synthetic patches (`--synthetic`), in which case None is returned.  Missing TEST folders return None:
        test() evaluates the test sets that exist (the reference's list names three)."""
        from . import data
        is_gray = self.num_channels == 1
        synthetic = bool(getattr(self.args, "synthetic", False))
        if synthetic:
            return None
        try:
            if is_train:
                ds = data.get_training_set(self.data_dir, dataset, self.crop_size, self.scale_factor, is_gray=is_gray,
                                           device=self.device)
                bs, shuffle = self.batch_size, True
            else:
                ds = data.get_test_set(self.data_dir, dataset, self.scale_factor, is_gray=is_gray, device=self.device)
                bs, shuffle = self.test_batch_size or 1, False
        except (OSError, TypeError) as e:
            if not is_train:
                return None
            raise FileNotFoundError("training set %r not found under --data_dir %r (%s: %s); pass --synthetic to train on "
                                    "seeded random patches instead" % (dataset, self.data_dir, type(e).__name__, e))
        if len(ds) == 0:
            if not is_train:
                return None
            raise FileNotFoundError("training set %r under --data_dir %r holds no images; pass --synthetic to train on "
                                    "seeded random patches instead" % (dataset, self.data_dir))
        if is_train:   # DP: disjoint 1/world shards of one common permutation per epoch
            return data.PatchLoader(ds, bs, shuffle=shuffle, num_threads=self.num_threads or 4, seed=1234,
                                    rank=self.rank, world=self.world)
        return data.PatchLoader(ds, bs, shuffle=shuffle, num_threads=self.num_threads or 4)

    def _announce_data(self):
        if self.rank == 0:
            print("training data: %s%s" % (self.data_source, "" if self.data_source != "folder" else
                                           " %s under %s" % (self.train_dataset, self.data_dir)))

    def _channels(self, *tensors):
        """num_channels == 1: only the Y channel is super-resolved (edsr.py:139-142: hr[:, 0].unsqueeze(1))."""
        if self.num_channels == 1:
            return tuple(t[:, :1].contiguous() if t.shape[1] != 1 else t for t in tensors)
        return tensors

    # -- reference surface ------------------------------------------------------------------------
    def train(self, loader=None, log_every=0):
        self.model = self.build_model()
        self.model.weight_init()
        self.model.to(self.device).train()
        utils.print_network(self.model) if self.rank == 0 else None
        self.flat, self.optimizer, self.dp, step = trainers.build(self.kind, self.model, self.lr,
                                                                  use_dp=self.world > 1)
        avg_loss = []
        self.data_source = "loader"
        if loader is None:
            loader = self.load_dataset(self.train_dataset, is_train=True)
            self.data_source = "folder" if loader is not None else "synthetic"
        self._announce_data()
        trainers.quiesce_gc()    # no full cyclic collection (~80 ms) inside a step from here on
        for epoch in range(self.num_epochs):
            self.lr_decay(epoch, self.optimizer)
            batches = loader if loader is not None else synthetic_loader(self.kind, self.args, self.steps_per_epoch,
                                                                         self.device, 1234 + epoch * self.world + self.rank)
            total, n = torch.zeros((), device=self.device), 0
            for batch in batches:
                inp, target = self._channels(*[t.to(self.device, non_blocking=True) for t in batch][:2])
                out = self._step(step, self.prepare(inp, target))
                loss = sum(out) if isinstance(out, tuple) else out
                total += loss.detach()   # device-side accumulation: no host sync inside the loop
                n += 1
            avg_loss.append(float(total) / max(n, 1))   # one sync per epoch
            if self.rank == 0:
                print('Epoch: [%2d] avg loss: %.8f' % (epoch + 1, avg_loss[-1]))
                if (epoch + 1) % self.save_epochs == 0:
                    self.save_model(epoch + 1)
        self._close_graph()
        if self.rank == 0:
            self.save_model(epoch=None)
        return avg_loss

    # -- the train step as a hipGraph ------------------------------------------------------------------------------
    # The reference's loop launches ~110 (EDSR) small kernels per iteration; at its default batch sizes the step is a
    # few hundred microseconds of GPU work behind 1.5+ ms of host launches.  The FIRST batch of a shape runs eagerly (a
    # real training step, and every lazy initialisation happens outside a capture), the next one is captured
    # (trainers.GraphedStep: zero_grad + filter packing + forward + loss + backward + optimizer as one graph, split at
    # the gradient exchange under data parallelism) and replayed from then on.  A batch of another shape (the ragged
    # last one) runs eagerly; a learning-rate decay needs nothing (the rate is a device scalar).  --eager turns it off.
    _GRAPH_LOSS = {"edsr": (ops.l1_loss, None), "vdsr": (ops.mse_loss, 0.4), "srcnn": (ops.mse_loss, None),
                   "fsrcnn": (ops.mse_loss, None), "espcn": (ops.mse_loss, None)}

    def _step(self, eager_step, tensors):
        auto = getattr(self, "_auto", None)
        if auto is None or auto.eager is not eager_step:
            spec = self._GRAPH_LOSS.get(self.kind)

            single = self.dp is None or not self.dp.active

            def make(ts):
                if spec is None:    # steps with their own loss structure (LapSRN's two Charbonnier terms): the eager step
                    return trainers.GraphedFn(eager_step, ts, warmup=0, flats=[self.flat])   # function itself, one GPU
                return trainers.GraphedStep(self.model, self.optimizer, spec[0], ts, dp=self.dp, clip=spec[1], warmup=0)

            auto = self._auto = trainers.AutoGraph(eager_step, make,
                                                   enabled=(spec is not None or single) and not getattr(self.args, "eager", False))
        return auto(*tensors)

    @property
    def _graph(self):
        auto = getattr(self, "_auto", None)
        return None if auto is None else auto.graph

    def _close_graph(self):
        auto = getattr(self, "_auto", None)
        if auto is not None:
            auto.close()

    def _net_input(self, x):
        """SRCNN / VDSR feed the bicubic-upsampled image to the net (srcnn.py:145, vdsr.py:160)."""
        return utils.img_interp(x, self.scale_factor) if self.kind in ("srcnn", "vdsr") else x

    def _infer(self, x):
        self.model.eval()
        with torch.no_grad():
            return self.model(x.to(self.device))

    def test(self, loader=None):
        """Evaluation loop (espcn.py:173-215, edsr.py:196-250): forward + PSNR per image (computed on the device), over
        `loader`, else over every folder of `test_dataset` that exists under `data_dir` (data.get_test_set), else over
        seeded synthetic pairs.  Returns the list of PSNRs; `self.test_psnr` holds the per-dataset averages."""
        if self.model is None:
            self.model = self.build_model().to(self.device)
            self.load_model()
        sources = []
        if loader is not None:
            sources.append(("loader", loader))
        else:
            for name in (self.test_dataset or []):
                ld = self.load_dataset(name, is_train=False) if isinstance(name, str) and len(name) > 1 else None
                if ld is not None:
                    sources.append((name, ld))
            if not sources:
                sources.append(("synthetic", synthetic_loader(self.kind, self.args, 2, self.device, 4321)))
        psnrs, self.test_psnr = [], {}
        for name, batches in sources:
            mine = []
            for batch in batches:
                items = [batch] if torch.is_tensor(batch[0]) else list(zip(*batch))   # ragged test images come as lists
                for item in items:
                    lr_img, hr_img = self._channels(*[t if t.dim() == 4 else t.unsqueeze(0) for t in item[:2]])
                    out = self._infer(self._net_input(lr_img.to(self.device)))
                    out = out[-1] if isinstance(out, tuple) else out
                    tgt = hr_img.to(self.device)
                    if self.kind == "srcnn":     # srcnn.py:193-199: border pixels excluded
                        tgt = utils.shave(tgt, 8)
                    if out.shape == tgt.shape:
                        mine.append(utils.PSNR(out, tgt))   # 0-dim device tensors: nothing syncs inside the loop
            vals = [float(v) for v in torch.stack(mine).cpu()] if mine else []
            if vals:
                self.test_psnr[name] = sum(vals) / len(vals)
            psnrs += vals
        return psnrs

    def test_single(self, img):
        """Super-resolve one [C,H,W] (or [1,C,H,W]) tensor (the reference reads an image file with PIL)."""
        if self.model is None:
            self.model = self.build_model().to(self.device)
            self.load_model()
        x = img if img.dim() == 4 else img.unsqueeze(0)
        out = self._infer(self._net_input(x.to(self.device)))
        return (out[-1] if isinstance(out, tuple) else out).cpu()

    def _ckpt_name(self, epoch):
        # srcnn.py:260-269 style for the simple trainers, edsr.py:324-337 style (ch/batch/epoch/lr) for EDSR / SRGAN
        model_dir = os.path.join(self.save_dir, 'model')
        os.makedirs(model_dir, exist_ok=True)
        if self.kind in ("edsr",):
            return model_dir + '/' + self.model_name + '_param_ch%d_batch%d_epoch%d_lr%.g.pkl' % (
                self.num_channels, self.batch_size, self.num_epochs if epoch is None else epoch, self.lr)
        if epoch is not None:
            return model_dir + '/' + self.model_name + '_param_epoch_%d.pkl' % epoch
        return model_dir + '/' + self.model_name + '_param.pkl'

    def save_model(self, epoch=None):
        sd = {k: v.detach().cpu().clone() for k, v in self.model.state_dict().items()}
        torch.save(sd, self._ckpt_name(epoch))
        print('Trained model is saved.')

    def load_model(self):
        name = self._ckpt_name(None)
        if os.path.exists(name):
            self.model.load_state_dict(torch.load(name))
            print('Trained model is loaded.')
            return True
        print('No model exists to load.')
        return False


class SRCNN(_Trainer):
    kind = "srcnn"

    def build_model(self):
        return models.SRCNNNet(self.num_channels, 64)   # srcnn.py:73


class ESPCN(_Trainer):
    kind = "espcn"

    def build_model(self):
        return models.ESPCNNet(self.num_channels, 64, self.scale_factor)   # espcn.py:73


class FSRCNN(_Trainer):
    kind = "fsrcnn"

    def build_model(self):
        return models.FSRCNNNet(self.num_channels, self.scale_factor, 56, 12, 4)   # fsrcnn.py:99


class VDSR(_Trainer):
    kind = "vdsr"

    def build_model(self):
        return models.VDSRNet(self.num_channels, 64, 18)   # vdsr.py:80



class EDSR(_Trainer):
    kind = "edsr"

    def build_model(self):
        return models.EDSRNet(self.num_channels, 64, 16)   # edsr.py:87



class LapSRN(_Trainer):
    kind = "lapsrn"

    def build_model(self):
        return models.LapSRNNet(self.num_channels, 64, 10)   # lapsrn.py:129



class SRGAN(_Trainer):
    """srgan.py:93-528: generator pre-training with MSE, then the adversarial loop."""
    kind = "srgan"

    def build_model(self):
        self.G = models.SRGANGenerator(self.num_channels, 64, 16)                 # srgan.py:136
        self.D = models.SRGANDiscriminator(self.num_channels, 64, self.crop_size)  # srgan.py:137
        return self.G

    def train(self, loader=None, pretrain_epochs=None, log_every=0):
        self.model = self.build_model()
        self.G.weight_init(mean=0.0, std=0.02)
        self.D.weight_init(mean=0.0, std=0.02)
        self.G.to(self.device).train()
        self.D.to(self.device).train()
        g_flat, d_flat = optim.FlatParams(self.G), optim.FlatParams(self.D)
        g_opt = optim.make_optimizer("srgan_g", g_flat, self.lr)
        d_opt = optim.make_optimizer("srgan_d", d_flat, self.lr)
        g_dp = d_dp = None
        if self.world > 1:
            g_dp, d_dp = dpmod.DataParallel(g_flat), dpmod.DataParallel(d_flat)
            g_dp.broadcast_params()
            d_dp.broadcast_params()
            if getattr(self.args, "sync_bn", False):   # statistics of the global batch (SURVEY.md 8e caveat)
                trainers.sync_batchnorm(self.G)
                trainers.sync_batchnorm(self.D)
        norm = lambda t: utils.norm(t, vgg=True)   # srgan.py:193-194,257-258

        self.data_source = "loader"
        if loader is None:
            loader = self.load_dataset(self.train_dataset, is_train=True)
            self.data_source = "folder" if loader is not None else "synthetic"
        self._announce_data()
        trainers.quiesce_gc()    # no full cyclic collection (~80 ms) inside a step from here on

        def batches(seed):   # loaders yield (lr, hr) or the reference's (lr, hr, bicubic) tuples (dataset.py:101)
            for batch in (loader or synthetic_loader("srgan", self.args, self.steps_per_epoch, self.device, seed)):
                lr_img, hr_img = self._channels(*batch[:2])
                yield norm(lr_img.to(self.device, non_blocking=True)), norm(hr_img.to(self.device, non_blocking=True))

        # generator pre-training (srgan.py:179-219): 50 epochs of MSE unless a pre-trained generator checkpoint loads
        self.epoch_pretrain = int(getattr(self.args, "epoch_pretrain", 50)) if pretrain_epochs is None else pretrain_epochs
        if self.load_model(is_pretrain=True):
            g_flat.mark_changed()      # parameters changed behind the optimizer's back: re-pack filters
        else:
            eager = bool(getattr(self.args, "eager", False))
            pre_step = trainers.AutoGraph(
                trainers.mse_step(self.G, g_opt, g_dp),
                lambda ts: trainers.GraphedStep(self.G, g_opt, ops.mse_loss, ts, dp=g_dp, warmup=0), enabled=not eager)
            for epoch in range(self.epoch_pretrain):
                for y_, x_ in batches(77 + epoch):
                    pre_step(y_, x_)
            pre_step.close()
            if self.rank == 0:
                self.save_model(is_pretrain=True)
        # the adversarial step (two models, two optimizers) as one hipGraph; data parallel: graphs split at the two exchanges
        eager_step = trainers.srgan_step(self.G, self.D, g_opt, d_opt, g_dp, d_dp, lazy_pack=True,
                                         prune_dead_grads=bool(getattr(self.args, "prune_dead_grads", False)))

        def make_graph(ts):
            if g_dp is not None and g_dp.active:
                return trainers.GraphedSegments(trainers.srgan_segments(self.G, self.D, g_opt, d_opt, g_dp, d_dp, lazy_pack=True), ts,
                                                warmup=0)
            return trainers.GraphedFn(eager_step, ts, warmup=0, flats=[g_flat, d_flat])

        step = trainers.AutoGraph(eager_step, make_graph, enabled=not bool(getattr(self.args, "eager", False)))
        hist = []
        for epoch in range(self.num_epochs):
            apply_lr_decay("srgan", epoch, g_opt, d_opt)   # srgan.py:239-244: both learning rates /10 every 20 epochs
            d_tot, g_tot, n = torch.zeros((), device=self.device), torch.zeros((), device=self.device), 0
            for y_, x_ in batches(1234 + epoch):
                d_loss, g_loss = step(y_, x_)
                d_tot += d_loss.detach()
                g_tot += g_loss.detach()
                n += 1
            hist.append((float(d_tot) / max(n, 1), float(g_tot) / max(n, 1)))
            if self.rank == 0:
                print('Epoch: [%2d] D_loss: %.8f G_loss: %.8f' % ((epoch + 1,) + hist[-1]))
                if (epoch + 1) % self.save_epochs == 0:
                    self.save_model(epoch + 1)
        step.close()
        if self.rank == 0:
            self.save_model(epoch=None)
        return hist

    def _names(self, epoch):
        model_dir = os.path.join(self.save_dir, 'model')
        os.makedirs(model_dir, exist_ok=True)
        tail = '_param_ch%d_batch%d_epoch%d_lr%.g.pkl' % (self.num_channels, self.batch_size,
                                                           self.num_epochs if epoch is None else epoch, self.lr)
        return (model_dir + '/' + self.model_name + '_G' + tail, model_dir + '/' + self.model_name + '_D' + tail,
                model_dir + '/' + self.model_name + '_G_param_pretrain.pkl')

    def save_model(self, epoch=None, is_pretrain=False):   # srgan.py:483-506
        g_name, d_name, pre = self._names(epoch)
        cpu = lambda m: {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
        if is_pretrain:
            torch.save(cpu(self.G), pre)
            print('Pre-trained generator model is saved.')
        else:
            torch.save(cpu(self.G), g_name)
            torch.save(cpu(self.D), d_name)
            print('Trained models are saved.')

    def load_model(self, is_pretrain=False):   # srgan.py:508-526
        g_name, _, pre = self._names(None)
        name = pre if is_pretrain else g_name
        if os.path.exists(name):
            self.G.load_state_dict(torch.load(name))
            print('Trained generator model is loaded.')
            return True
        return False


TRAINERS = {"SRCNN": SRCNN, "VDSR": VDSR, "ESPCN": ESPCN, "FSRCNN": FSRCNN, "SRGAN": SRGAN, "LapSRN": LapSRN, "EDSR": EDSR}
