"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol declared in
include/srk.h, the nn.Module surface matches the reference's (class names, ctor signatures,
state_dict keys, init distributions), the product path refuses CPU tensors (no fallback), and the
data-parallel plumbing works with world_size 2 over gloo."""
import inspect
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import fill, ref_modules as R

import __graft_entry__


@pytest.fixture(scope="module")
def pkg():
    __graft_entry__.build()
    import pytorch_super_resolution_model_collection_amd as p
    return p


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg._lib.load()
    syms = pkg._lib.header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), s
    assert set(syms) == set(pkg._lib._PROTOTYPES), "ctypes prototypes out of sync with include/srk.h"
    assert lib.srk_version() == 600
    assert lib.srk_status_string(-2) == b"unsupported configuration"
    # geometry helper is pure host code
    assert lib.srk_conv_out_dim(256, 5, 1, 0, 0, 0) == 252
    assert lib.srk_conv_out_dim(28, 9, 4, 3, 1, 1) == 112      # FSRCNN deconv (fsrcnn.py:33)
    assert lib.srk_conv_out_dim(32, 4, 2, 1, 1, 0) == 64       # LapSRN / Upsample2x deconv
    assert lib.srk_conv_out_dim(128, 3, 2, 1, 0, 0) == 64      # SRGAN-D stride 2
    assert lib.srk_conv_out_dim(0, 3, 1, 1, 0, 0) == -1


def test_argument_validation_without_gpu(pkg):
    """Bad arguments are rejected before anything is launched (works without a device)."""
    lib = pkg._lib.load()
    assert lib.srk_pixel_shuffle_forward(None, None, 1, 1, 1, 1, 2, None) == -1
    d = pkg._lib.ConvDesc(1, 8, 8, 3, 8, 8, 4, 3, 3, 1, 1, 0, 0, 0)
    assert lib.srk_conv2d_forward(d, None, None, None, None, None) == -1
    bad = pkg._lib.ConvDesc(1, 8, 8, 3, 7, 7, 4, 3, 3, 1, 1, 0, 0, 0)  # wrong OH/OW
    assert lib.srk_conv2d_backward_weight_workspace_bytes(bad) >= 0
    import ctypes
    buf = ctypes.c_void_p(16)
    assert lib.srk_conv2d_forward(bad, buf, buf, buf, None, None) == -1
    assert b"OH/OW" in lib.srk_last_error_string()


PAIRS = [("SRCNNNet", R.SRCNN, (3, 64)), ("ESPCNNet", R.ESPCN, (3, 64, 4)), ("FSRCNNNet", R.FSRCNN, (3, 4, 56, 12, 4)),
         ("VDSRNet", R.VDSR, (3, 64, 18)), ("EDSRNet", R.EDSR, (3, 64, 16)), ("LapSRNNet", R.LapSRN, (3, 64, 10)),
         ("SRGANGenerator", R.Generator, (3, 64, 16)), ("SRGANDiscriminator", R.Discriminator, (3, 64, 32))]


@pytest.mark.parametrize("name,ora,args", PAIRS, ids=[p[0] for p in PAIRS])
def test_state_dict_layout_matches_reference(pkg, name, ora, args):
    a, b = getattr(pkg, name)(*args), ora(*args)
    sa, sb = a.state_dict(), b.state_dict()
    assert list(sa.keys()) == list(sb.keys())
    for k in sa:
        assert tuple(sa[k].shape) == tuple(sb[k].shape), k
    b.load_state_dict(sa)  # loads without key/shape errors both ways
    a.load_state_dict(sb)


def test_block_constructor_signatures_match_reference(pkg):
    for cls in ("DenseBlock", "ConvBlock", "DeconvBlock", "ResnetBlock", "PSBlock", "Upsample2xBlock"):
        pa = inspect.signature(getattr(pkg.base_networks, cls).__init__).parameters
        pb = inspect.signature(getattr(R, cls).__init__).parameters
        assert list(pa) == list(pb), cls
        for k in pa:
            assert pa[k].default == pb[k].default, (cls, k)


def test_block_variants_construct_like_the_reference(pkg):
    """Every (upsample, norm) variant the reference's blocks can be built with constructs here too, with the same
    state_dict keys and shapes: 'rnc' upsampler, instance norm, DenseBlock's BatchNorm1d."""
    B = pkg.base_networks
    cases = [(B.Upsample2xBlock, R.Upsample2xBlock, (8, 12), dict(upsample='rnc', activation='relu', norm=None)),
             (B.Upsample2xBlock, R.Upsample2xBlock, (8, 12), dict(upsample='rnc', activation='prelu', norm='batch')),
             (B.ConvBlock, R.ConvBlock, (8, 16, 3, 1, 1), dict(norm='instance')),
             (B.DeconvBlock, R.DeconvBlock, (8, 16), dict(norm='instance')),
             (B.ResnetBlock, R.ResnetBlock, (16,), dict(norm='instance')),
             (B.PSBlock, R.PSBlock, (8, 8, 2), dict(norm='instance')),
             (B.DenseBlock, R.DenseBlock, (24, 10), dict()),                       # default norm='batch' -> BatchNorm1d
             (B.DenseBlock, R.DenseBlock, (24, 10), dict(activation='prelu', norm='batch')),
             (B.DenseBlock, R.DenseBlock, (24, 10), dict(norm='instance'))]     # InstanceNorm1d on the [B, F] activation
    for ours, theirs, args, kw in cases:
        a, b = ours(*args, **kw), theirs(*args, **kw)
        sa, sb = a.state_dict(), b.state_dict()
        assert list(sa.keys()) == list(sb.keys()), (ours.__name__, kw)
        for k in sa:
            assert tuple(sa[k].shape) == tuple(sb[k].shape), k
        b.load_state_dict(sa)
        a.load_state_dict(sb)
        for m in a.modules():          # the reference's initialiser walks every module by class name
            pkg.utils.weights_init_normal(m)
    with pytest.raises(ValueError):
        B.Upsample2xBlock(8, 8, upsample='bogus')


def test_vgg_feature_extractor_layout(pkg):
    """srgan.py:84-90: vgg19.features[:9] — same Sequential indices, so a torchvision vgg19 state_dict loads directly."""
    fe = pkg.FeatureExtractor()
    assert list(fe.state_dict().keys()) == ["features.%d.%s" % (i, k) for i in (0, 2, 5, 7) for k in ("weight", "bias")]
    assert [tuple(fe.state_dict()["features.%d.weight" % i].shape) for i in (0, 2, 5, 7)] == \
        [(64, 3, 3, 3), (64, 64, 3, 3), (128, 64, 3, 3), (128, 128, 3, 3)]
    assert len(fe.features) == 9 and isinstance(fe.features[4], torch.nn.MaxPool2d)
    assert not any(p.requires_grad for p in fe.parameters())
    full = {k: torch.zeros_like(v) for k, v in fe.state_dict().items()}
    full["features.10.weight"] = torch.zeros(256, 128, 3, 3)     # deeper vgg19 entries are ignored
    full["classifier.0.weight"] = torch.zeros(8, 8)
    fe.load_vgg19(full)
    assert float(fe.features[0].weight.abs().max()) == 0.0
    with pytest.raises(KeyError):
        fe.load_vgg19({"features.0.weight": torch.zeros(64, 3, 3, 3)})
    assert len(pkg.FeatureExtractor(feature_layer=36).features) == 37   # the whole vgg19.features stack


def test_lr_decay_rules_match_reference(pkg):
    """vdsr.py:127-129 (/10 every 20), edsr.py:131-133 (/2 every 40), lapsrn.py:173-175 (/10 every 100),
    srgan.py:239-244 (/10 every 20, G and D); srcnn / espcn / fsrcnn never decay."""
    from pytorch_super_resolution_model_collection_amd import sr_trainers as T

    class Opt(object):
        def __init__(self, lr):
            self.param_groups = [{"lr": lr}, {"lr": lr * 2}]

    def run(kind, epochs, n_opts=1):
        opts = [Opt(1.0) for _ in range(n_opts)]
        for e in range(epochs):
            T.apply_lr_decay(kind, e, *opts)
        return [o.param_groups[0]["lr"] for o in opts], opts[0].param_groups[1]["lr"]

    def ref(every, factor, epochs):
        lr = 1.0
        for e in range(epochs):
            if (e + 1) % every == 0:
                lr /= factor
        return lr

    assert run("vdsr", 45)[0] == [ref(20, 10.0, 45)]
    assert run("edsr", 100)[0] == [ref(40, 2.0, 100)]
    assert run("lapsrn", 250)[0] == [ref(100, 10.0, 250)]
    assert run("lapsrn", 99)[0] == [1.0]
    assert run("srgan", 60, 2)[0] == [ref(20, 10.0, 60)] * 2
    assert run("srgan", 60, 2)[1] == 2 * ref(20, 10.0, 60)       # every param group
    for kind in ("srcnn", "espcn", "fsrcnn"):
        assert run(kind, 300)[0] == [1.0]
    assert T.LR_DECAY == {"vdsr": (20, 10.0), "edsr": (40, 2.0), "lapsrn": (100, 10.0), "srgan": (20, 10.0)}


def test_integration_md_stubs_match_header(pkg):
    """INTEGRATION.md's ctypes stub is what a maintainer pastes: its Structure `_fields_` must match include/srk.h
    (names, order, count) and the package's own binding."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    header = open(os.path.join(root, "include", "srk.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)

    def header_fields(struct):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), header, flags=re.S).group(1)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            decl = re.sub(r"^(const\s+)?(float|int32_t|int64_t|double|int)\s*\*?", "", decl).strip()
            names += [n.strip().lstrip("*") for n in decl.split(",")]
        return names

    def md_fields(cls):
        m = re.search(r"class %s\(ctypes\.Structure\):.*?_fields_\s*=\s*\[(.*?)\]\n" % cls, text, flags=re.S)
        assert m, "INTEGRATION.md has no ctypes stub for %s" % cls
        return re.findall(r'\(\s*"(\w+)"', m.group(1))

    for struct, cls, ours in (("srk_conv_desc", "ConvDesc", pkg._lib.ConvDesc), ("srk_epilogue", "Epilogue", pkg._lib.Epilogue),
                              ("srk_bwd_mask", "BwdMask", pkg._lib.BwdMask)):
        want = header_fields(struct)
        assert md_fields(cls) == want, (cls, md_fields(cls), want)
        assert [f[0] for f in ours._fields_] == want, cls
    n_units = len([f for f in os.listdir(os.path.join(root, "pytorch_super_resolution_model_collection_amd", "csrc"))
                   if f.endswith(".hip")])
    m = re.search(r"(\d+) translation units", text)
    assert m and int(m.group(1)) == n_units == len(pkg._build.SOURCES), (m and m.group(1), n_units)


def test_weight_init_distributions(pkg):
    torch.manual_seed(0)
    net = pkg.EDSRNet(3, 64, 16)
    net.weight_init()
    w = net.residual_layers[3].conv1.weight
    assert abs(float(w.std()) - 0.02) < 2e-3 and abs(float(w.mean())) < 1e-3
    assert float(net.residual_layers[3].conv1.bias.abs().max()) == 0.0
    s = pkg.SRCNNNet(3, 64)
    s.weight_init()
    assert abs(float(s.layers[1].conv.weight.std()) - 0.001) < 1e-4
    v = pkg.VDSRNet(3, 64, 18)
    v.weight_init()
    assert abs(float(v.residual_layers[0].conv.weight.std()) - (2.0 / (64 * 9)) ** 0.5) < 5e-3
    lap = pkg.LapSRNNet(3, 64, 10)
    lap.weight_init()
    assert torch.equal(lap.convt_I1.deconv.weight[0, 0], pkg.models.get_upsample_filter(4))
    f = pkg.FSRCNNNet(3, 4, 56, 12, 4)
    f.weight_init()
    assert abs(float(f.last_part.weight.std()) - 1e-4) < 2e-5


def test_no_cpu_fallback(pkg):
    net = pkg.ESPCNNet(3, 64, 4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.zeros(1, 3, 16, 16))
    with pytest.raises(RuntimeError):
        pkg.ops.mse_loss(torch.zeros(1, 3, 4, 4), torch.zeros(1, 3, 4, 4))
    with pytest.raises(RuntimeError):
        pkg.optim.FlatParams(net)


def test_product_package_never_imports_the_oracle():
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                        "pytorch_super_resolution_model_collection_amd")
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f


def test_shard_range_covers_batch_exactly(pkg):
    for n in (1, 7, 16, 128, 129):
        for world in (1, 2, 3, 8):
            spans = [pkg.dp.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
                assert a1 == b0 and a1 >= a0
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_dp_bucket_plan(pkg):
    """dp.DataParallel.plan: which ranges of the flat gradient buffer leave after which weight-gradient group —
    every element exactly once, never before the group that finalises it, small runs held back for a neighbour."""
    class FakeFlat(object):
        pass
    sizes = [12, 400, 400, 400, 400, 4, 200, 8]      # input conv, 4 trunk layers, a PReLU slope, upsampler, output conv
    flat = FakeFlat()
    flat.offsets, total = [], 0
    for n in sizes:
        flat.offsets.append(total)
        total += n
    flat.grad = torch.zeros(total)
    flat.data = torch.zeros(total)
    views = [flat.grad[o:o + n] for o, n in zip(flat.offsets, sizes)]
    dp = pkg.dp.DataParallel(flat, min_bucket_bytes=4 * 100)

    def rec(i):   # (key, desc, x, dy, mask, slope, wacc, bacc)
        return ("k%d" % i, None, None, None, None, 0.0, views[i], None)
    # backward order: output conv, upsampler, trunk in two chunks (layers 4,3 then 2,1), input conv; index 5 never deferred
    groups = [[rec(7)], [rec(6)], [rec(4), rec(3)], [rec(2), rec(1)], [rec(0)]]
    sends = dp.plan(groups)
    assert len(sends) == len(groups)
    assert sends[0] == []                                   # 8 floats: held back
    assert sends[1] == [(flat.offsets[5], total)]           # slope (final all along) + upsampler + output conv
    assert sends[2] == [(flat.offsets[3], flat.offsets[5])]
    assert sends[3] == [(flat.offsets[1], flat.offsets[3])]
    assert sends[4] == [(0, flat.offsets[1])]               # last group: whatever is left, however small
    covered = sorted(r for s_ in sends for r in s_)
    assert covered[0][0] == 0 and covered[-1][1] == total and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
    assert dp.plan([]) == [[(0, total)]]


class _Flat(object):
    def __init__(self, n, seed):
        self.data = fill.randn((n,), seed)
        self.grad = fill.randn((n,), seed + 100)


def _dp_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import pytorch_super_resolution_model_collection_amd as pkg
    r, w, _ = pkg.dp.init_from_env("gloo")
    assert (r, w) == (rank, world)
    flat = _Flat(1000, 10 + rank)           # different replicas before the broadcast
    dp = pkg.dp.DataParallel(flat, bucket_bytes=1024)   # 256-float buckets -> 4 buckets
    assert len(dp.buckets()) == 4 and abs(float(dp.loss_seed) - 1.0 / world) < 1e-7
    dp.broadcast_params()
    local_grad = flat.grad.clone()
    dp.allreduce_grads()
    # shard a batch, compute a "mean-loss gradient" per shard seeded with 1/world, check it equals the full-batch one
    batch = fill.randn((8, 5), 3)
    sh = pkg.dp.shard(batch, rank, world)
    g = (sh.mean(0) * float(dp.loss_seed)).clone()
    dist.all_reduce(g)
    torch.save({"data": flat.data, "sum": flat.grad, "local": local_grad, "g": g, "full": batch.mean(0),
                "loss": dp.allreduce_scalar(torch.tensor(float(rank)))}, out % rank)
    dist.destroy_process_group()


def test_data_parallel_gloo_world2(tmp_path):
    world, port = 2, 29600 + os.getpid() % 200
    out = str(tmp_path / "r%d.pt")
    mp.spawn(_dp_worker, args=(world, port, out), nprocs=world, join=True)
    r0, r1 = torch.load(out % 0), torch.load(out % 1)
    assert torch.equal(r0["data"], r1["data"]) and torch.equal(r0["data"], fill.randn((1000,), 10))
    assert torch.allclose(r0["sum"], r0["local"] + r1["local"]) and torch.equal(r0["sum"], r1["sum"])
    assert torch.allclose(r0["g"], r0["full"], atol=1e-6)
    assert float(r0["loss"]) == 0.5


def test_bench_prints_its_line_when_a_side_metric_hangs():
    """bench.py at N > 1: a side metric that never returns (one rank dropped out of a collective) must not cost the
    headline line -- the watchdog prints it with what was collected and every rank exits 0."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "res = {'metric': 'm', 'value': 1.0}; extra = {'c4_edsr_ms_per_step': 6.6}\n"
            "bench.extras_watchdog(res, extra, 0, 1)\n"
            "time.sleep(30)\n"
            "print('not reached')\n" % root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-500:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and "not reached" not in r.stdout
    rec = json.loads(lines[0])
    assert rec["value"] == 1.0 and rec["extra"]["c4_edsr_ms_per_step"] == 6.6 and "extras_error" in rec["extra"]


def test_patch_loader_shards_one_permutation_over_the_ranks(pkg):
    """Data-parallel training from image folders (sr_trainers.load_dataset): every rank draws the same per-epoch
    permutation and takes a disjoint 1/world of it -- together the ranks cover the folder once per epoch (round-2
    advisor finding: every rank used to load the same images)."""
    class FakeFolder(object):
        def __init__(self, n):
            self.image_filenames = ["%d.png" % i for i in range(n)]

        def __len__(self):
            return len(self.image_filenames)

    for n, world in ((800, 8), (10, 4), (7, 2)):
        loaders = [pkg.data.PatchLoader(FakeFolder(n), 2, shuffle=True, num_threads=1, seed=1234, rank=r, world=world,
                                        background=False) for r in range(world)]
        assert len({len(ld) for ld in loaders}) == 1                      # the same number of batches on every rank
        for epoch in range(3):
            orders = [ld._order() for ld in loaders]
            assert len({len(o) for o in orders}) == 1
            flat = [i for o in orders for i in o]
            assert set(flat) == set(range(n))                             # the whole folder, once per epoch ...
            assert len(flat) - n < world                                  # ... plus at most world-1 wrapped indices
            if n % world == 0:
                assert len(set(flat)) == len(flat)                        # disjoint shards
        assert loaders[0]._order() != loaders[0]._order()                 # a new permutation every epoch
    # one rank: the old behaviour, seedable
    a = pkg.data.PatchLoader(FakeFolder(9), 2, seed=5, background=False)._order()
    b = pkg.data.PatchLoader(FakeFolder(9), 2, seed=5, background=False)._order()
    assert a == b and sorted(a) == list(range(9))


def test_cli_dataset_names_and_synthetic_flag(tmp_path):
    """`--train_dataset DIV2K` is one name (the reference's `type=list` splits it into characters, main.py:18);
    `--synthetic` is the only way to random patches."""
    import main as cli
    a = cli.parse_args(["--train_dataset", "DIV2K", "--test_dataset", "Set5,Set14", "--save_dir", str(tmp_path)])
    assert a.train_dataset == ["DIV2K"] and a.test_dataset == ["Set5", "Set14"] and a.synthetic is False
    a = cli.parse_args(["--save_dir", str(tmp_path), "--synthetic"])
    assert a.train_dataset == ["DIV2K"] and a.test_dataset == ["Set5", "Set14", "Urban100"] and a.synthetic is True


# The reference's `random` consumption per training item, from its source (torchvision is not installed here, so its
# dataset classes cannot run: SURVEY.md 8c / VERDICT r2 weak #3).  (dataset.py line, call, condition); the two torchvision
# transforms are written out as torchvision 0.2 implements them: RandomCrop.get_params draws i = randint(0, h - th) then
# j = randint(0, w - tw) unless the image already has the crop size, RandomHorizontalFlip tests random() < 0.5.
REFERENCE_DRAWS = [
    (54, "random.randint(5, 10)", "random_scale"),
    (66, "RandomCrop(self.crop_size)", "always"),
    (71, "random.randint(1, 3)", "rotate"),
    (76, "RandomHorizontalFlip()", "fliplr"),
    (81, "random.random() < 0.5", "fliptb"),
]


def test_training_item_draws_follow_the_reference_source_order(pkg, tmp_path, monkeypatch):
    """data.TrainDatasetFromFolder.draw consumes Python's `random` in the order of dataset.py:51-83 -- the table above,
    checked against the reference's source lines when /root/reference is present (this container) and replayed against
    draw() for every flag combination: same calls, same arguments, same order."""
    ref = "/root/reference/dataset.py"
    if os.path.exists(ref):
        lines = open(ref).read().splitlines()
        for lineno, text, _ in REFERENCE_DRAWS:
            assert text in lines[lineno - 1], (lineno, text, lines[lineno - 1])
        order = [lineno for lineno, _, _ in REFERENCE_DRAWS]
        assert order == sorted(order)
    calls = []

    class Recorder(object):
        @staticmethod
        def randint(a, b):
            calls.append(("randint", a, b))
            return a          # smallest value: the random-scale ratio 0.5 < 1 takes the reference's "force >= crop" branch
        @staticmethod
        def random():
            calls.append(("random",))
            return 0.25

    monkeypatch.setattr(pkg.data, "random", Recorder)
    img_dir = tmp_path / "imgs"
    img_dir.mkdir()
    import itertools
    for random_scale, rotate, fliplr, fliptb in itertools.product([True, False], repeat=4):
        ds = pkg.data.TrainDatasetFromFolder([str(img_dir)], random_scale=random_scale, crop_size=32, rotate=rotate,
                                             fliplr=fliplr, fliptb=fliptb, scale_factor=4, device="cpu")
        for (w, h) in ((48, 40), (32, 32)):
            del calls[:]
            scale, (x0, y0), rot, fl, ft = ds.draw(w, h)
            want = []
            cw, ch = w, h
            if random_scale:
                want.append(("randint", 5, 10))
                # dataset.py:54-60: ratio forced to crop / crop + eps -> the whole image becomes crop x crop (App. B-8)
                cw = ch = int(32 * (32 / 32 + 1e-3))
                assert scale == (cw, ch)
            else:
                assert scale is None
            if (cw, ch) != (32, 32):       # torchvision 0.2 RandomCrop.get_params
                want += [("randint", 0, ch - 32), ("randint", 0, cw - 32)]
            if rotate:
                want.append(("randint", 1, 3))
            if fliplr:
                want.append(("random",))
            if fliptb:
                want.append(("random",))
            assert calls == want, (random_scale, rotate, fliplr, fliptb, w, h, calls, want)
            assert rot == (1 if rotate else 0) and fl == fliplr and ft == fliptb
