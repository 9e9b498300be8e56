"""Op-level parity on the MI355X, through the C ABI: every kernel family against the committed
golden vectors (generated from the torch.nn classes the reference instantiates).  Tolerance: the
contract is 1e-3 relative (BASELINE.json north_star); the exact-fp32 kernels are held to 2e-5.

Environment: the suite runs the library with SRK_ENV_LIVE=1 (tests/conftest.py) -- the kernel-variant tests flip SRK_* switches
between calls of one process, which the library otherwise reads once per process.  Only the reading of the switches differs
from the product configuration (`bench.py` runs without it); every kernel and dispatch decision is the same."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import TOL_TIGHT, assert_close_elementwise, rel_err
from oracle import fill
from oracle.kat_table import CONV_KATS, conv_case_inputs

pytestmark = pytest.mark.gpu

ALGOS = {"auto": 0, "generic": 1, "mfma_fp32": 2, "bf16x3": 4, "bf16x6": 5}
# AUTO runs the bf16x3 split-MFMA kernels where they apply (~1e-5 rel); generic / fp32 MFMA are exact-order fp32
# and bf16x6 (exact 3-way operand split, 6 bf16 MFMAs) is held to the same tolerance as them.
# variant -> (algo, env): the env switches pick the kernel family / block configuration that a problem of
# benchmark size would get (the KAT shapes are all "small problems" for the automatic choice)
VARIANTS = {"auto": ("auto", {}), "generic": ("generic", {}), "mfma_fp32": ("mfma_fp32", {}),
            "auto_lds_weights": ("auto", {"SRK_BFD_SMALL": "0"}),
            "auto_global_weights_big": ("auto", {"SRK_BF3_DIRECT": "1", "SRK_BFD_SMALL": "0"}),
            "auto_wave_specialized": ("auto", {"SRK_BFW": "1"}),
            "bf16x6": ("bf16x6", {}), "bf16x6_big": ("bf16x6", {"SRK_BFD_SMALL": "0"}),
            # the fp32-faithful class as f16x3 (two fp16 planes of the power-of-two scaled operands, three MFMAs) wherever
            # the fp16 kernels cover the layer; every input takes the srk_absmax pass (ops.F16X3_ALWAYS)
            "f16x3": ("bf16x6", {}), "f16x3_big": ("bf16x6", {"SRK_BFD_SMALL": "0"}),
            "f16x3_big_4x64": ("bf16x6", {"SRK_BFD_SMALL": "0", "SRK_BFD_F16_CFG": "0"})}
TOL_ALGO = {"auto": 1e-4, "generic": TOL_TIGHT, "mfma_fp32": TOL_TIGHT, "bf16x6": TOL_TIGHT}
ACTS = {None: 0, "relu": 1, "lrelu": 3}


def _pkg():
    import pytorch_super_resolution_model_collection_amd as pkg
    return pkg


@pytest.mark.parametrize("variant", list(VARIANTS))
@pytest.mark.parametrize("idx", range(len(CONV_KATS)), ids=[c[0] for c in CONV_KATS])
def test_conv_forward_backward(gpu, ops_kat, idx, variant, monkeypatch):
    algo, env = VARIANTS[variant]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    pkg = _pkg()
    ops = pkg.ops
    monkeypatch.setattr(ops, "F16X3_ALWAYS", variant.startswith("f16x3"))
    if variant.startswith("bf16x6"):
        monkeypatch.setattr(ops, "F16X3", False)
    tag, cin, cout, k, s, p, tr, op, H, W, N, act = CONV_KATS[idx]
    x, w, b, g = conv_case_inputs(idx)
    xg = x.to(gpu).requires_grad_(True)
    wg = w.to(gpu).requires_grad_(True)
    bg = b.to(gpu).requires_grad_(True)
    cfg = ops.ConvCfg(s, p, bool(tr), op, ACTS[act], 0.2 if act == "lrelu" else 0.0, 0, ALGOS[algo])
    y = ops.conv2d(xg, wg, bg, None, cfg)
    assert tuple(y.shape) == tuple(g.shape)
    if variant.startswith("f16x3") and min(cin, cout) >= 8:
        kern = pkg._lib.load().srk_last_kernel_name().decode()
        assert kern.startswith("k_conv_bfd") and ",f16" in kern, kern       # the fp16 kernel did run ...
        tagged = getattr(y, "_srk_amax", None)                              # ... and left the output's maximum behind
        assert tagged is not None and tagged[0] is not None
        assert float(tagged[0].max()) == float(y.detach().abs().max())
    tol = TOL_ALGO[algo]
    assert rel_err(y, ops_kat["conv.%s.y" % tag]) < tol
    y.backward(g.to(gpu))
    assert rel_err(xg.grad, ops_kat["conv.%s.dx" % tag]) < tol
    assert rel_err(wg.grad, ops_kat["conv.%s.dw" % tag]) < tol
    assert rel_err(bg.grad, ops_kat["conv.%s.db" % tag]) < tol


@pytest.mark.parametrize("idx", [0, 3, 9, 10], ids=lambda i: CONV_KATS[i][0])
def test_conv_infer_fused_epilogue(gpu, idx):
    """Fully fused no-grad path: bias + PReLU + residual in one launch vs torch CPU."""
    pkg = _pkg()
    ops = pkg.ops
    tag, cin, cout, k, s, p, tr, op, H, W, N, act = CONV_KATS[idx]
    x, w, b, g = conv_case_inputs(idx)
    slope = torch.tensor([0.3])
    res = fill.randn(tuple(g.shape), 555)
    ref = torch.nn.functional.prelu(torch.nn.functional.conv2d(x, w, b, s, p), slope) + res
    cfg = ops.ConvCfg(s, p, False, 0, pkg._lib.ACT_PRELU)
    with torch.no_grad():
        y = ops.conv2d_infer(x.to(gpu), w.to(gpu), b.to(gpu), res.to(gpu), cfg, slope.to(gpu))
    assert rel_err(y, ref) < 1e-4


@pytest.mark.parametrize("cin,cout,k,p,H,W,N", [
    (64, 3, 3, 1, 37, 29, 2),    # several ragged 16x16 tiles per image (EDSR / VDSR reconstruction conv)
    (64, 3, 3, 0, 20, 45, 1),    # no padding, wide
    (32, 2, 3, 1, 17, 17, 3),    # one 32-channel step, two output channels
    (64, 1, 3, 1, 16, 16, 1),    # single output channel: one 16-column fragment
    (64, 3, 1, 0, 19, 23, 2),    # 1x1
    (32, 3, 2, 1, 9, 12, 1),     # even kernel
    (64, 3, 9, 4, 30, 26, 2),    # SRGAN-G output conv: 9 tap groups of 27 columns over the same activation registers
    (32, 3, 5, 0, 20, 24, 2),    # SRCNN output conv: 3 tap groups (10 + 10 + 5 taps)
    (64, 2, 7, 3, 15, 17, 1),    # 16 taps per group, 49 taps
])
@pytest.mark.parametrize("algo", ["auto", "bf16x6"])
def test_conv_few_output_channels(gpu, monkeypatch, cin, cout, k, p, H, W, N, algo):
    """The taps-as-N kernel (KH*KW*Cout <= 32 columns, conv_tapn.hip) against torch fp64 on CPU, held to the
    exact-fp32 tolerance in every precision class (it always runs the exact 3-way split); bias + LeakyReLU +
    residual go through its scalar epilogue.  Kernels with more taps than one 32-column group (9x9, 5x5, 7x7) run
    k_conv_rown (conv_rown.hip: kernel rows on N, one GEMM per input row) -- bf16x6 here in both classes too ("auto" in the
    fp32-faithful mode of the suite resolves to the exact split)."""
    pkg = _pkg()
    ops = pkg.ops
    monkeypatch.setenv("SRK_ROWN", "2")     # (k_conv_rown on problems of any size: by default small ones stay on k_conv_tapn)
    x = fill.randn((N, cin, H, W), 71)
    w = fill.randn((cout, cin, k, k), 72, (2.0 / (cin * k * k)) ** 0.5)
    b = fill.randn((cout,), 73, 0.1)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), 1, p)
    res = fill.randn(tuple(ref.shape), 74)
    ref = torch.nn.functional.leaky_relu(ref, 0.2) + res.double()
    cfg = ops.ConvCfg(1, p, False, 0, ACTS["lrelu"], 0.2, 0, ALGOS[algo])
    with torch.no_grad():
        y = ops.conv2d_infer(x.to(gpu), w.to(gpu), b.to(gpu), res.to(gpu), cfg)
    name = pkg._lib.load().srk_last_kernel_name().decode()
    assert name.startswith("k_conv_rown<" if k * k * cout > 32 else "k_conv_tapn<"), name
    assert rel_err(y, ref.float()) < TOL_TIGHT


@pytest.mark.parametrize("cin,cout,k,p,H,W,N,th", [
    (64, 3, 9, 4, 70, 150, 2, 0),    # SRGAN-G output conv: three 64-column tiles (the last ragged), several row tiles
    (64, 3, 9, 4, 70, 150, 2, 8),    # ... 8-row tiles: 16 input rows per tile, the ring wraps inside a tile
    (64, 3, 9, 4, 33, 64, 1, 32),    # ... 32-row tiles, the second one a single row; exactly one column tile
    (64, 3, 9, 0, 40, 90, 2, 16),    # no padding (output 32 x 82)
    (32, 3, 5, 0, 52, 52, 3, 0),     # SRCNN output conv (c1 size): one 32-channel step, one 16-column N tile
    (32, 2, 5, 2, 20, 100, 2, 16),   # two output channels, one N tile
    (64, 2, 7, 3, 31, 65, 2, 0),     # 7x7, two output channels; one column past the first tile
    (64, 3, 9, 4, 128, 128, 2, 0),   # c5 size per image
])
@pytest.mark.parametrize("mode", ["bf16x3", "bf16x6"])
def test_conv_rows_on_n_kernel(gpu, monkeypatch, cin, cout, k, p, H, W, N, th, mode):
    """k_conv_rown (conv_rown.hip): few output channels, more taps than one 32-column group.  Both arithmetics (bf16x3: 1e-4,
    the exact split: fp32 tolerance), every tile height the host may pick (SRK_ROWN_TH), ragged row / column tiles, rows of
    the halo outside the image (skipped matrix phases), bias + LeakyReLU + residual through the epilogue; vs torch fp64."""
    pkg = _pkg()
    ops = pkg.ops
    monkeypatch.setenv("SRK_ROWN", "2")
    if th:
        monkeypatch.setenv("SRK_ROWN_TH", str(th))
    x = fill.randn((N, cin, H, W), 171)
    w = fill.randn((cout, cin, k, k), 172, (2.0 / (cin * k * k)) ** 0.5)
    b = fill.randn((cout,), 173, 0.1)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), 1, p)
    res = fill.randn(tuple(ref.shape), 174)
    ref = torch.nn.functional.leaky_relu(ref, 0.2) + res.double()
    cfg = ops.ConvCfg(1, p, False, 0, ACTS["lrelu"], 0.2, 0, ALGOS[mode])
    with torch.no_grad():
        y = ops.conv2d_infer(x.to(gpu), w.to(gpu), b.to(gpu), res.to(gpu), cfg)
        y2 = ops.conv2d_infer(x.to(gpu), w.to(gpu), b.to(gpu), res.to(gpu), cfg)
    name = pkg._lib.load().srk_last_kernel_name().decode()
    assert name.startswith("k_conv_rown<%d,%d,%d,%d," % (cin // 32, cout, k, 3 if mode == "bf16x6" else 2)), name
    assert torch.equal(y, y2)                                   # fixed summation order: run-to-run identical
    assert rel_err(y, ref.float()) < (TOL_TIGHT if mode == "bf16x6" else 1e-4)
    assert_close_elementwise(y, ref.float(), 1e-4 if mode == "bf16x6" else 1e-3, what="k_conv_rown")


@pytest.mark.parametrize("cin,cout,k,p,H,W,N", [
    (64, 3, 3, 1, 41, 41, 3),    # VDSR patch: the second 32-pixel K step of every row is ragged
    (64, 3, 3, 0, 20, 70, 2),    # no padding: dy smaller than x
    (32, 2, 3, 1, 17, 33, 2),    # half of the channel lanes idle
    (48, 1, 3, 1, 16, 16, 1),    # one 16-column fragment, Cin not a multiple of 32 (forward runs k_conv_direct)
    (64, 3, 1, 0, 19, 23, 2),    # 1x1
])
def test_conv_few_output_channels_backward(gpu, cin, cout, k, p, H, W, N):
    """Training step pieces of the reconstruction convs: forward (taps-as-N kernel, bf16x6), data gradient, and the
    transposition-free weight-gradient kernel with its fused bias-gradient partials (conv_tapn.hip), vs torch fp64."""
    pkg = _pkg()
    ops = pkg.ops
    x = fill.randn((N, cin, H, W), 81)
    w = fill.randn((cout, cin, k, k), 82, (2.0 / (cin * k * k)) ** 0.5)
    b = fill.randn((cout,), 83, 0.1)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    ref = torch.nn.functional.conv2d(xr, wr, br, 1, p)
    g = fill.randn(tuple(ref.shape), 84)
    ref.backward(g.double())
    xg, wg, bg = (t.to(gpu).requires_grad_(True) for t in (x, w, b))
    y = ops.conv2d(xg, wg, bg, None, ops.ConvCfg(1, p, False, 0, 0, 0.0, 0, ALGOS["auto"]))
    y.backward(g.to(gpu))
    assert rel_err(y, ref.detach().float()) < TOL_TIGHT
    assert rel_err(xg.grad, xr.grad.float()) < 1e-4
    assert rel_err(wg.grad, wr.grad.float()) < 1e-4
    assert rel_err(bg.grad, br.grad.float()) < 1e-4


@pytest.mark.parametrize("cin,cout,k,p,H,W,N,act", [
    (64, 64, 3, 1, 32, 32, 4, None),       # SRGAN-D conv2 geometry at a small size: four phases of 16 x 16, taps 2x2 / 2x1 / 1x2 / 1x1
    (512, 512, 3, 1, 16, 16, 16, None),    # the discriminator's last strided layer at c5 size: 16 chunks, K split over two wave groups
    (128, 128, 3, 1, 15, 21, 3, "lrelu"),  # odd sizes: the phases differ in height and width; LeakyReLU mask on dy
    (64, 128, 4, 1, 18, 18, 2, None),      # even kernel: every phase has 2x2 taps
    (256, 256, 3, 0, 17, 17, 2, "lrelu"),  # no padding
])
def test_strided_data_gradient_phases_in_one_launch(gpu, monkeypatch, cin, cout, k, p, H, W, N, act):
    """Round 6: the s x s phases of a stride-2 data gradient on the small-problem block run as ONE launch (k_conv_bfd_mp: a
    block finds its phase's parameters from its index) instead of four.  Same tiles, same arithmetic, same order per output
    element: bit-equal to the phase-by-phase launches (SRK_BFD_MP=0), and both against torch fp64."""
    pkg = _pkg()
    ops, L = pkg.ops, pkg._lib
    lib = L.load()
    x = fill.randn((N, cin, H, W), 381)
    w = fill.randn((cout, cin, k, k), 382, (2.0 / (cin * k * k)) ** 0.5)
    b = fill.randn((cout,), 383, 0.1)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = torch.nn.functional.conv2d(xr, wr, b.double(), 2, p)
    if act:
        yr = torch.nn.functional.leaky_relu(yr, 0.2)
    g = fill.randn(tuple(yr.shape), 384)
    yr.backward(g.double())
    cfg = ops.ConvCfg(2, p, False, 0, 0, 0.0, 0, ALGOS["auto"])
    d = ops._make_desc(x.shape, w, cfg, "bwd")
    dy = g.to(gpu).contiguous(memory_format=torch.channels_last)
    yg = yr.detach().float().to(gpu).contiguous(memory_format=torch.channels_last)
    wpb = ops.pack_weight_bwd(w.to(gpu), False, 0)
    mask = L.BwdMask(L.ptr(yg), 0.2) if act else None
    dx = {}
    for mp in ("1", "0"):
        monkeypatch.setenv("SRK_BFD_MP", mp)
        out = torch.full((N, cin, H, W), float("nan"), device=gpu).contiguous(memory_format=torch.channels_last)
        assert lib.srk_conv2d_backward_data(ctypes.byref(d), L.ptr(dy), L.ptr(wpb), L.ptr(out),
                                            ctypes.byref(mask) if mask is not None else None, None, L.stream_ptr()) == 0
        name = lib.srk_last_kernel_name().decode()
        if mp == "1":
            assert name.startswith("k_conv_bfd_mp<1,1,4,2,2,") and name.endswith("x4"), name
        else:
            assert name.startswith("k_conv_bfd<1,1,4,2,2,"), name
        dx[mp] = (out, name.split("x4")[0].rstrip(">").split(",")[-1])    # (output, K-split factor of the launch)
        assert rel_err(out, xr.grad.float()) < 1e-4
    # the K split over two wave groups is decided on the blocks of a LAUNCH (one phase: few blocks -> split; all phases
    # together may fill the CUs without it): same split -> the same sums in the same order, bit for bit
    if dx["1"][1] == dx["0"][1]:
        assert torch.equal(dx["1"][0], dx["0"][0])
    else:
        assert rel_err(dx["1"][0], dx["0"][0]) < 2e-5
    # ... and the layer through autograd (forward, dx, dw) on the merged path
    monkeypatch.setenv("SRK_BFD_MP", "1")
    xg, wg, bg = (t.to(gpu).requires_grad_(True) for t in (x, w, b))
    cfg2 = ops.ConvCfg(2, p, False, 0, ACTS[act], 0.2 if act else 0.0, 0, ALGOS["auto"])
    y = ops.conv2d(xg, wg, bg, None, cfg2)
    y.backward(g.to(gpu))
    assert rel_err(xg.grad, xr.grad.float()) < 1e-4
    assert rel_err(wg.grad, wr.grad.float()) < 1e-4


@pytest.mark.parametrize("cin,cout,k,p,H,W,N,slope", [
    (3, 64, 3, 1, 37, 29, 2, 0.2),     # SRGAN-D first layer (srgan.py:51): dx[3] from dy[64] under the LeakyReLU gradient
    (3, 64, 3, 1, 16, 50, 1, 0.0),     # ReLU mask
    (2, 32, 3, 0, 20, 21, 2, 0.2),     # one 32-channel step, two image channels, no padding
    (1, 64, 5, 2, 18, 18, 1, 0.2),     # 25 taps x 1 channel
])
def test_conv_image_gradient_with_mask(gpu, cin, cout, k, p, H, W, N, slope):
    """The data gradient of a FIRST layer (few image channels in, 32 / 64 out, activation behind it): a TRANS gather with 32 /
    64 input and <= 3 output channels and the activation-gradient mask on its input -- k_conv_tapn's single pass with the mask
    prologue (k_conv_direct until round 6).  Through the C ABI, exact three-way split: fp32 tolerance vs torch fp64."""
    pkg = _pkg()
    ops, L = pkg.ops, pkg._lib
    lib = L.load()
    x = fill.randn((N, cin, H, W), 191)
    w = fill.randn((cout, cin, k, k), 192, (2.0 / (cin * k * k)) ** 0.5)
    xr = x.double().requires_grad_(True)
    y = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(xr, w.double(), None, 1, p), slope)
    g = fill.randn(tuple(y.shape), 193)
    y.backward(g.double())
    cfg = ops.ConvCfg(1, p, False, 0, 0, 0.0, 0, ALGOS["auto"])
    d = ops._make_desc(x.shape, w, cfg, "bwd")
    dy = g.to(gpu).contiguous(memory_format=torch.channels_last)
    yg = y.detach().float().to(gpu).contiguous(memory_format=torch.channels_last)
    wpb = ops.pack_weight_bwd(w.to(gpu), False, 0)
    dx = torch.empty((N, cin, H, W), device=gpu).contiguous(memory_format=torch.channels_last)
    mask = L.BwdMask(L.ptr(yg), slope)
    assert lib.srk_conv2d_backward_data(ctypes.byref(d), L.ptr(dy), L.ptr(wpb), L.ptr(dx), ctypes.byref(mask), None,
                                        L.stream_ptr()) == 0
    assert lib.srk_last_kernel_name().decode() == "k_conv_tapn<%d,%d,mask>" % (cout // 32, cin)
    assert rel_err(dx, xr.grad.float()) < TOL_TIGHT


@pytest.mark.parametrize("cin,cout,k,p,H,W,N,resid", [
    (64, 3, 9, 4, 30, 40, 2, False),    # SRGAN-G output conv: dx[64] from dy[3], 9 kernel rows of 27 K slots, 4 channel tiles
    (64, 3, 9, 4, 17, 70, 1, True),     # ragged 16-pixel groups, an odd number of them; fused "+ residual" (gradient fan-in)
    (32, 3, 5, 2, 20, 24, 2, False),    # SRCNN output conv: two channel tiles, 15 K slots per row
    (64, 2, 7, 3, 15, 33, 1, False),    # two gradient channels: 14 slots per row
    (48, 3, 9, 0, 24, 24, 1, False),    # no padding (dy smaller than dx), three channel tiles
    (64, 3, 9, 4, 128, 128, 1, False),  # c5 image size
])
def test_conv_many_tap_data_gradient(gpu, monkeypatch, cin, cout, k, p, H, W, N, resid):
    """k_conv_tapkm (conv_tapn.hip): the data gradient of a few-output-channel conv with more than 32 / Cout taps -- a TRANS
    gather with <= 4 input channels, one kernel row per K step, filter fragments in LDS.  Through the C ABI
    (srk_conv2d_backward_data: the dispatch must pick it) and through autograd (whole layer: forward, dx, dw, db); bf16x3
    arithmetic: 1e-4 of the tensor's maximum, 1e-3 element-wise; vs torch fp64."""
    pkg = _pkg()
    ops, L = pkg.ops, pkg._lib
    lib = L.load()
    monkeypatch.setenv("SRK_WGRAD_TAPNW", "2")   # the column-split weight gradient (k_wgrad_tapnw) on these small problems too
    monkeypatch.setenv("SRK_ROWN", "2")
    x = fill.randn((N, cin, H, W), 181)
    w = fill.randn((cout, cin, k, k), 182, (2.0 / (cin * k * k)) ** 0.5)
    b = fill.randn((cout,), 183, 0.1)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    ref = torch.nn.functional.conv2d(xr, wr, br, 1, p)
    g = fill.randn(tuple(ref.shape), 184)
    ref.backward(g.double())
    cfg = ops.ConvCfg(1, p, False, 0, 0, 0.0, 0, ALGOS["auto"])
    # the C ABI call: dx = conv^T(dy) [+ add_to]
    d = ops._make_desc(x.shape, w, cfg, "bwd")
    dy = g.to(gpu).contiguous(memory_format=torch.channels_last)
    wpb = ops.pack_weight_bwd(w.to(gpu), False, 0)
    dx = torch.empty((N, cin, H, W), device=gpu).contiguous(memory_format=torch.channels_last)
    add = fill.randn((N, cin, H, W), 185).to(gpu).contiguous(memory_format=torch.channels_last) if resid else None
    assert lib.srk_conv2d_backward_data(ctypes.byref(d), L.ptr(dy), L.ptr(wpb), L.ptr(dx), None, L.ptr(add),
                                        L.stream_ptr()) == 0
    assert lib.srk_last_kernel_name().decode() == "k_conv_tapkm<%d>" % (cin // 16)
    want = xr.grad.float() + (add.cpu() if resid else 0.0)
    assert rel_err(dx, want) < 1e-4
    assert_close_elementwise(dx, want, 1e-3, what="k_conv_tapkm")
    # the layer through autograd
    xg, wg, bg = (t.to(gpu).requires_grad_(True) for t in (x, w, b))
    y = ops.conv2d(xg, wg, bg, None, cfg)
    y.backward(g.to(gpu))
    assert rel_err(y, ref.detach().float()) < TOL_TIGHT
    assert rel_err(xg.grad, xr.grad.float()) < 1e-4
    assert rel_err(wg.grad, wr.grad.float()) < 1e-4
    assert rel_err(bg.grad, br.grad.float()) < 1e-4
    assert_close_elementwise(wg.grad, wr.grad.float(), 1e-3, what="k_wgrad_tapnw dw")
    # ... and the same weight gradient from the kernel this one replaces (role-swapped exact fp32): both are the layer's dw
    monkeypatch.setenv("SRK_WGRAD_TAPNW", "0")
    xg2, wg2, bg2 = (t.to(gpu).requires_grad_(True) for t in (x, w, b))
    ops.conv2d(xg2, wg2, bg2, None, cfg).backward(g.to(gpu))
    assert rel_err(wg.grad, wg2.grad) < 1e-4 and rel_err(bg.grad, bg2.grad) < 1e-4


@pytest.mark.parametrize("cin,cout,k,p,H,W,N,act,ps", [
    (64, 32, 3, 0, 30, 41, 2, "relu", 0),     # c2 layer 2 shape class: two chunks, unrolled taps + deferred stores
    (32, 48, 3, 0, 21, 37, 2, None, 4),       # c2 layer 3: one chunk, fused pixel-shuffle store
    (96, 16, 3, 1, 19, 19, 1, "lrelu", 0),    # three chunks, one 16-channel tile, padding
    (40, 64, 3, 1, 17, 23, 2, "prelu", 0),    # Cin not a multiple of 32 (last chunk partly empty), 64 channels
    (16, 32, 5, 2, 20, 20, 1, "relu", 0),     # 5x5: dynamic tap loop, immediate epilogue
    (32, 48, 1, 0, 33, 9, 3, None, 0),        # 1x1
    (64, 48, 3, 1, 40, 40, 1, "relu", 2),     # pixel shuffle r = 2 with 12-channel runs
])
def test_conv_wave_specialized(gpu, monkeypatch, cin, cout, k, p, H, W, N, act, ps):
    """k_conv_bfw (producer / consumer waves, filter resident in LDS) forced onto small problems: every template
    variant (1-4 channel tiles, unrolled / dynamic taps, 32- / 64-pixel consumer waves), ragged tiles on both image
    edges, partial channel chunks, all activations it accepts, the fused pixel-shuffle store; vs torch fp64."""
    monkeypatch.setenv("SRK_BFW", "1")
    pkg = _pkg()
    ops = pkg.ops
    x = fill.randn((N, cin, H, W), 91)
    w = fill.randn((cout, cin, k, k), 92, (2.0 / (cin * k * k)) ** 0.5)
    b = fill.randn((cout,), 93, 0.1)
    slope = torch.tensor([0.3])
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), 1, p)
    if act == "relu":
        ref = torch.relu(ref)
    elif act == "lrelu":
        ref = torch.nn.functional.leaky_relu(ref, 0.2)
    elif act == "prelu":
        ref = torch.nn.functional.prelu(ref, slope.double())
    if ps:
        ref = torch.nn.functional.pixel_shuffle(ref, ps)
    code = {None: 0, "relu": 1, "lrelu": 3, "prelu": pkg._lib.ACT_PRELU}[act]
    cfg = ops.ConvCfg(1, p, False, 0, code, 0.2 if act == "lrelu" else 0.0, ps, ALGOS["auto"])
    with torch.no_grad():
        y = ops.conv2d_infer(x.to(gpu), w.to(gpu), b.to(gpu), None, cfg, slope.to(gpu) if act == "prelu" else None)
    assert tuple(y.shape) == tuple(ref.shape)
    assert rel_err(y, ref.float()) < 1e-4


@pytest.mark.parametrize("cin,cout,k,p,H,W,N,act,mode", [
    (3, 64, 5, 0, 40, 44, 2, "relu", "mixed"),      # ESPCN first layer (c2), f16x3, NCHW input read in place
    (3, 64, 5, 0, 40, 44, 2, "relu", "bf16x3"),
    (3, 64, 3, 1, 41, 41, 3, "relu", "mixed"),      # VDSR / EDSR first layer: 3 K steps (filter in registers), padding
    (1, 64, 5, 2, 19, 70, 1, "lrelu", "bf16x3"),    # one channel, ragged tiles
    (4, 128, 3, 1, 17, 23, 2, None, "bf16x3"),      # four channels, two 64-channel slices
    (3, 64, 5, 2, 8, 300, 1, None, "mixed"),        # fewer tiles than XCDs
    (2, 64, 3, 0, 33, 18, 5, "relu", "mixed"),      # two channels, no padding
    (3, 64, 5, 2, 24, 16, 200, "relu", "mixed"),    # 600 one-tile bands on 512 blocks: one or two bands per block
    (3, 64, 3, 1, 50, 100, 30, None, "bf16x3"),     # 210 bands of seven tiles, ragged last column chunk
    (3, 64, 5, 0, 70, 64, 9, "relu", "mixed"),      # no padding: the input is as wide as the tiles' columns
])
def test_conv_first_layer_wave_specialized(gpu, monkeypatch, cin, cout, k, p, H, W, N, act, mode):
    """k_conv_rowsw (persistent, producer / consumer waves, deferred stores) forced onto small problems, against the
    per-tile row-packed kernel it replaces at benchmark size (same arithmetic and summation order: equal outputs) and
    torch fp64."""
    pkg = _pkg()
    ops = pkg.ops
    x = fill.randn((N, cin, H, W), 291)
    w = fill.randn((cout, cin, k, k), 292, (2.0 / (cin * k * k)) ** 0.5)
    b = fill.randn((cout,), 293, 0.1)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), 1, p)
    if act == "relu":
        ref = torch.relu(ref)
    elif act == "lrelu":
        ref = torch.nn.functional.leaky_relu(ref, 0.2)
    code = {None: 0, "relu": 1, "lrelu": 3}[act]
    cfg = ops.ConvCfg(1, p, False, 0, code, 0.2 if act == "lrelu" else 0.0, 0, ALGOS["auto"])
    outs = {}
    monkeypatch.setattr(ops, "F16X3_ALWAYS", True)   # (small problems: the size policy would keep the fp32 MFMA kernel)
    ops.set_precision(mode)
    try:
        # "0": per-tile kernel; "1": 8-wave persistent kernel; "r": its row-reuse form (4-wave blocks, filter in registers);
        # "b": that kernel walking whole bands of tiles with one aligned 16-column chunk staged per tile (round 6) -- where
        # the input is no wider than the tiles' columns, else the library stays with "r"
        PW = W + 2 * p - k + 1
        band_ok = W <= 16 * ((PW + 15) // 16)
        for sw in ("0", "1", "r", "b"):
            monkeypatch.setenv("SRK_ROWSW", "0" if sw == "0" else "1")
            monkeypatch.setenv("SRK_ROWSR", "1" if sw in "rb" else "0")
            monkeypatch.setenv("SRK_ROWSB", "2" if sw == "b" else "0")
            with torch.no_grad():
                outs[sw] = ops.conv2d_infer(x.to(gpu), w.to(gpu), b.to(gpu), None, cfg)
            name = pkg._lib.load().srk_last_kernel_name().decode()
            assert name.startswith({"0": "k_conv_bf3_rows<", "1": "k_conv_rowsw<", "r": "k_conv_rowsr<", "b": "k_conv_rowsr<"}[sw]), name
            assert ("f16" in name) == (mode == "mixed"), name
            assert ("band" in name) == (sw == "b" and band_ok), name
    finally:
        ops.set_precision("mixed")
    assert torch.equal(outs["0"], outs["1"])
    assert torch.equal(outs["0"], outs["r"])
    assert torch.equal(outs["0"], outs["b"])
    assert rel_err(outs["1"], ref.float()) < (2e-6 if mode == "mixed" else 1e-4)


@pytest.mark.parametrize("cin,cout,H,W,N,act,ps", [
    (64, 64, 41, 41, 3, "relu", 0),     # VDSR body layer: two 32-channel slices (the whole filter is 147 KB)
    (64, 256, 16, 24, 2, None, 2),      # EDSR up-sampler: eight slices over four 64-channel filter blocks, PS store
    (32, 96, 19, 19, 1, "lrelu", 0),    # three slices, one chunk
    (64, 128, 9, 50, 1, None, 0),       # fewer tiles than XCDs: slice groups without work
])
def test_conv_wave_specialized_channel_slices(gpu, monkeypatch, cin, cout, H, W, N, act, ps):
    """k_conv_bfw on layers wider than its LDS-resident filter allows: 32-channel slices on neighbouring blocks
    (BfwParams.nsl).  Same arithmetic and accumulation order as the per-tile bf16x3 kernel -> equal outputs; and both
    vs torch fp64."""
    pkg = _pkg()
    ops = pkg.ops
    x = fill.randn((N, cin, H, W), 191)
    w = fill.randn((cout, cin, 3, 3), 192, (2.0 / (cin * 9)) ** 0.5)
    b = fill.randn((cout,), 193, 0.1)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), 1, 1)
    if act == "relu":
        ref = torch.relu(ref)
    elif act == "lrelu":
        ref = torch.nn.functional.leaky_relu(ref, 0.2)
    if ps:
        ref = torch.nn.functional.pixel_shuffle(ref, ps)
    code = {None: 0, "relu": 1, "lrelu": 3}[act]
    cfg = ops.ConvCfg(1, 1, False, 0, code, 0.2 if act == "lrelu" else 0.0, ps, ALGOS["auto"])
    monkeypatch.setenv("SRK_BF3_DIRECT", "0")   # reference kernel: the per-tile k_conv_bf3 (same summation order)
    outs = {}
    ops.set_precision("bf16x3")
    try:
        for mode in ("0", "1"):
            monkeypatch.setenv("SRK_BFW", mode)
            with torch.no_grad():
                outs[mode] = ops.conv2d_infer(x.to(gpu), w.to(gpu), b.to(gpu), None, cfg)
            name = pkg._lib.load().srk_last_kernel_name().decode()
            assert name.startswith(("k_conv_bfw<2,9,2", "k_conv_bfr<2")) == (mode == "1"), name
    finally:
        ops.set_precision("mixed")
    assert torch.equal(outs["0"], outs["1"])
    assert rel_err(outs["1"], ref.float()) < 1e-4


@pytest.mark.parametrize("cin,cout,H,W,N,act", [
    (64, 64, 41, 41, 3, "relu"),      # VDSR body layer
    (64, 64, 20, 33, 2, "lrelu"),     # slope on the masked side
    (32, 64, 24, 24, 2, "relu"),      # data gradient 64 -> 32: one slice, masked producers
    (64, 64, 17, 17, 2, None),        # no activation: plain producers, flipped taps only
])
def test_conv_wave_specialized_data_gradient(gpu, monkeypatch, cin, cout, H, W, N, act):
    """Data gradient of a stride-1 conv behind a fused activation on k_conv_bfw: TRANS gather (flipped taps), the
    activation-gradient mask applied by the producer waves, output-channel slices; vs torch fp64 and equal to the
    per-tile kernel."""
    pkg = _pkg()
    ops = pkg.ops
    ops.set_precision("mixed")
    x = fill.randn((N, cin, H, W), 201)
    w = fill.randn((cout, cin, 3, 3), 202, (2.0 / (cin * 9)) ** 0.5)
    b = fill.randn((cout,), 203, 0.1)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    ref = torch.nn.functional.conv2d(xr, wr, br, 1, 1)
    if act == "relu":
        ref = torch.relu(ref)
    elif act == "lrelu":
        ref = torch.nn.functional.leaky_relu(ref, 0.2)
    g = fill.randn(tuple(ref.shape), 204)
    ref.backward(g.double())
    code = {None: 0, "relu": 1, "lrelu": 3}[act]
    monkeypatch.setenv("SRK_BF3_DIRECT", "0")   # reference kernel: the per-tile k_conv_bf3 (same summation order)
    grads = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("SRK_BFW", mode)
        xg, wg, bg = (t.to(gpu).requires_grad_(True) for t in (x, w, b))
        y = ops.conv2d(xg, wg, bg, None, ops.ConvCfg(1, 1, False, 0, code, 0.2 if act == "lrelu" else 0.0, 0))
        y.backward(g.to(gpu))
        ops.flush_wgrads()
        grads[mode] = xg.grad.clone()
        assert rel_err(wg.grad, wr.grad.float()) < 1e-4
    assert torch.equal(grads["0"], grads["1"])
    assert rel_err(grads["1"], xr.grad.float()) < 1e-4


@pytest.mark.parametrize("cin,cout,H,W,N,act,ps,mode", [
    (64, 32, 44, 40, 3, "relu", 0, "mixed"),      # c2 second layer: two chunks, f16x3
    (32, 48, 37, 50, 2, None, 4, "mixed"),        # c2 third layer: one chunk, fused pixel-shuffle store, f16x3
    (64, 32, 9, 300, 1, "lrelu", 0, "bf16x3"),    # fewer tiles than XCDs, ragged rows and columns
    (32, 32, 19, 19, 1, "lrelu", 0, "bf16x3"),    # one chunk, odd tile counts per block
    (40, 32, 23, 17, 5, "prelu", 0, "mixed"),     # partial channel chunk (40 = 32 + 8)
    (64, 32, 130, 200, 4, "relu", 0, "mixed"),    # several pairs of tiles per block
    (32, 48, 90, 150, 6, "relu", 0, "bf16x3"),    # several pairs of tiles per block, 48 channels
])
def test_conv_wave_specialized_ring(gpu, monkeypatch, cin, cout, H, W, N, act, ps, mode):
    """k_conv_bfr (conv_bfr.hip): the wave-specialised kernel with row-reused fragments (4 x 16 pixels per consumer wave)
    and a ring of halo buffers with full / free counters in LDS instead of the per-stage workgroup barrier.  Same products
    as k_conv_bfw, summed in a different order (kernel column before kernel row): equal to it within fp32 summation
    noise, for every ring depth the LDS allows; no poll may run into its iteration cap; and vs torch fp64."""
    monkeypatch.setenv("SRK_BFW", "1")
    monkeypatch.setenv("SRK_BF3_DIRECT", "0")   # (bf16x3: small problems would take the per-tile k_conv_bfd)
    pkg = _pkg()
    ops = pkg.ops
    lib = pkg._lib.load()
    x = fill.randn((N, cin, H, W), 391)
    w = fill.randn((cout, cin, 3, 3), 392, (2.0 / (cin * 9)) ** 0.5)
    b = fill.randn((cout,), 393, 0.1)
    slope = torch.tensor([0.3])
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), 1, 0)
    if act == "relu":
        ref = torch.relu(ref)
    elif act == "lrelu":
        ref = torch.nn.functional.leaky_relu(ref, 0.2)
    elif act == "prelu":
        ref = torch.nn.functional.prelu(ref, slope.double())
    if ps:
        ref = torch.nn.functional.pixel_shuffle(ref, ps)
    code = {None: 0, "relu": 1, "lrelu": 3, "prelu": pkg._lib.ACT_PRELU}[act]
    cfg = ops.ConvCfg(1, 0, False, 0, code, 0.2 if act == "lrelu" else 0.0, ps, ALGOS["auto"])
    lib.srk_ring_timeouts(1)
    outs = {}
    monkeypatch.setattr(ops, "F16X3_ALWAYS", mode == "mixed")
    ops.set_precision(mode)
    ring_ok = not (cout == 48 and cin > 32)
    try:
        # (the ring kernel first: a kernel that stored nothing must not find the barrier kernel's result in recycled memory)
        for ring in ("1", "3", "4", "0"):
            monkeypatch.setenv("SRK_BFR", "0" if ring == "0" else "1")
            monkeypatch.setenv("SRK_BFR_NBUF", "0" if ring in "01" else ring)
            with torch.no_grad():
                outs[ring] = ops.conv2d_infer(x.to(gpu), w.to(gpu), b.to(gpu), None, cfg, slope.to(gpu) if act == "prelu" else None)
            name = lib.srk_last_kernel_name().decode()
            assert name.startswith("k_conv_bfr<" if ring != "0" and ring_ok else "k_conv_bfw<"), name
            assert ("f16" in name) == (mode == "mixed"), name
    finally:
        ops.set_precision("mixed")
    assert lib.srk_ring_timeouts(1) == 0
    for ring in ("1", "3", "4"):
        assert rel_err(outs[ring], outs["0"]) < 2e-6, ring
    assert torch.equal(outs["1"], outs["3"]) and torch.equal(outs["1"], outs["4"])
    assert rel_err(outs["1"], ref.float()) < (2e-6 if mode == "mixed" else 1e-4)


@pytest.mark.parametrize("cout,H,W,N,mode", [
    (64, 41, 41, 7, "mixed"),      # VDSR body layer: two 32-channel slices, 7 patches on a canvas, f16x3
    (64, 41, 41, 19, "bf16x3"),    # ... more patches than one canvas row holds
    (32, 20, 33, 5, "mixed"),      # one slice
    (64, 9, 17, 40, "bf16x3"),     # the smallest patches the canvas takes
    (64, 48, 50, 3, "mixed"),      # patches larger than a tile row of the canvas is tall
    (64, 130, 200, 1, "mixed"),    # one large image (VDSR's test() forward): a canvas of one cell, for the slices
    (64, 32, 32, 6, "mixed"),      # patches of whole tiles (EDSR's residual blocks): stacked without separators ("canvas0")
    (64, 16, 48, 9, "bf16x3"),     # ... the smallest such patches
])
def test_conv_ring_on_a_canvas(gpu, monkeypatch, cout, H, W, N, mode):
    """k_conv_bfr<2,2,..,canvas> (round 6): the ring kernel's fixed 8 x 16 tiles laid over the batch as a grid of
    (H + 1) x (W + 1) cells -- separator rows / columns read as the convolution's zero padding and are never stored -- with
    32-channel output slices.  Forward (f16x3 / bf16x3) against k_conv_bfw (same products, another summation order) and
    torch fp64; then the data gradient of a two-layer ReLU chain under ops.premasked_gradients() -- plain for the first
    layer, multiplied by the ReLU gradient of the layer below (ep.out_relu) for the second -- against k_conv_bfw's."""
    monkeypatch.setenv("SRK_BFW", "1")
    monkeypatch.setenv("SRK_BF3_DIRECT", "0")
    pkg = _pkg()
    ops = pkg.ops
    lib = pkg._lib.load()
    x = fill.randn((N, 64, H, W), 491)
    w = fill.randn((cout, 64, 3, 3), 492, (2.0 / (64 * 9)) ** 0.5)
    b = fill.randn((cout,), 493, 0.1)
    ref = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), b.double(), 1, 1))
    cfg = ops.ConvCfg(1, 1, False, 0, 1, 0.0, 0, ALGOS["auto"])
    lib.srk_ring_timeouts(1)
    outs = {}
    monkeypatch.setattr(ops, "F16X3_ALWAYS", mode == "mixed")
    ops.set_precision(mode)
    try:
        for cv in ("2", "0"):
            monkeypatch.setenv("SRK_BFR_CV", cv)
            monkeypatch.setenv("SRK_BFR", "1" if cv == "2" else "0")
            with torch.no_grad():
                outs[cv] = ops.conv2d_infer(x.to(gpu), w.to(gpu), b.to(gpu), None, cfg)
            name = lib.srk_last_kernel_name().decode()
            assert name.startswith("k_conv_bfr<2,2" if cv == "2" else "k_conv_bfw<"), name
            assert ("canvas" in name) == (cv == "2") and ("f16" in name) == (mode == "mixed"), name
            assert ("canvas0" in name) == (cv == "2" and H % 8 == 0 and W % 16 == 0), name
    finally:
        ops.set_precision("mixed")
    assert lib.srk_ring_timeouts(1) == 0
    assert rel_err(outs["2"], outs["0"]) < 2e-6
    assert rel_err(outs["2"], ref.float()) < (2e-6 if mode == "mixed" else 1e-4)
    if H % 8 == 0 and W % 16 == 0:   # the same layer on the canvas WITH separators: same products, same order -> equal outputs
        monkeypatch.setenv("SRK_BFR_CV", "2")
        monkeypatch.setenv("SRK_BFR", "1")
        monkeypatch.setenv("SRK_BFR_CV_EXACT", "0")
        ops.set_precision(mode)
        try:
            with torch.no_grad():
                sep = ops.conv2d_infer(x.to(gpu), w.to(gpu), b.to(gpu), None, cfg)
            name = lib.srk_last_kernel_name().decode()
            assert "canvas" in name and "canvas0" not in name, name
        finally:
            ops.set_precision("mixed")
            monkeypatch.setenv("SRK_BFR_CV_EXACT", "1")
        assert torch.equal(sep, outs["2"])
    # conv + residual (the second conv of a residual block, no activation): the canvas variant adds it when a tile is parked;
    # without the canvas the layer leaves the wave-specialised family (k_conv_bfw has no residual)
    res = fill.randn((N, cout, H, W), 497)
    ref_r = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), 1, 1) + res.double()
    cfg_r = ops.ConvCfg(1, 1, False, 0, 0, 0.0, 0, ALGOS["auto"])
    outs_r = {}
    monkeypatch.setenv("SRK_C64", "0")   # (small activation-free 64 -> 64 problems would take the per-tile k_c64)
    ops.set_precision(mode)
    try:
        for cv in ("2", "0"):
            monkeypatch.setenv("SRK_BFR_CV", cv)
            monkeypatch.setenv("SRK_BFR", "1" if cv == "2" else "0")
            with torch.no_grad():
                outs_r[cv] = ops.conv2d_infer(x.to(gpu), w.to(gpu), b.to(gpu), res.to(gpu), cfg_r)
            name = lib.srk_last_kernel_name().decode()
            tag = "canvas0" if (H % 8 == 0 and W % 16 == 0) else "canvas"
            assert (name == "k_conv_bfr<2,2%s,%s,res>" % (",f16" if mode == "mixed" else "", tag)) == (cv == "2"), name
            assert not name.startswith("k_conv_bfw<"), name
    finally:
        ops.set_precision("mixed")
    assert lib.srk_ring_timeouts(1) == 0
    assert rel_err(outs_r["2"], outs_r["0"]) < (2e-6 if mode == "mixed" else 2e-5)
    assert rel_err(outs_r["2"], ref_r.float()) < (2e-6 if mode == "mixed" else 1e-4)
    if cout != 64:
        return
    # data gradients: x -> three conv + ReLU layers, 64 -> 64.  The top layer masks its own dy (k_conv_bfw: the canvas takes
    # no input mask), the middle one gets it pre-masked and masks its dx (canvas, relu), the first one is plain
    w2 = fill.randn((64, 64, 3, 3), 494, (2.0 / (64 * 9)) ** 0.5)
    w3 = fill.randn((64, 64, 3, 3), 496, (2.0 / (64 * 9)) ** 0.5)
    g = fill.randn((N, 64, H, W), 495)
    xr = x.double().requires_grad_(True)
    h = torch.relu(torch.nn.functional.conv2d(xr, w.double(), None, 1, 1))
    h = torch.relu(torch.nn.functional.conv2d(h, w2.double(), None, 1, 1))
    torch.relu(torch.nn.functional.conv2d(h, w3.double(), None, 1, 1)).backward(g.double())
    grads, names = {}, {}
    real = lib.srk_conv2d_backward_data_relu
    cfg2 = ops.ConvCfg(1, 1, False, 0, 1, 0.0, 0)
    try:
        for cv in ("2", "0"):
            monkeypatch.setenv("SRK_BFR_CV", cv)
            monkeypatch.setenv("SRK_BFR", "1" if cv == "2" else "0")
            seen = []
            lib.srk_conv2d_backward_data_relu = lambda *a: (real(*a), seen.append(lib.srk_last_kernel_name().decode()))[0]
            xg = x.to(gpu).requires_grad_(True)
            h1 = ops.conv2d(xg, w.to(gpu), None, None, cfg2)
            h2 = ops.conv2d(h1, w2.to(gpu), None, None, cfg2)
            out = ops.conv2d(h2, w3.to(gpu), None, None, cfg2)
            with ops.premasked_gradients():
                out.backward(g.to(gpu))
            names[cv] = (seen, lib.srk_last_kernel_name().decode())
            grads[cv] = xg.grad.clone()
    finally:
        lib.srk_conv2d_backward_data_relu = real
    assert lib.srk_ring_timeouts(1) == 0
    assert names["2"][0] == ["k_conv_bfw<2,9,2,mask,relu>", "k_conv_bfr<2,2,%s,relu>" % tag], names
    assert all(n.startswith("k_conv_bfw<") for n in names["0"][0]), names   # (the first layer's plain gradient: whatever the
                                                                           #  dispatch takes in this precision mode)
    # (a ReLU whose pre-activation is within rounding of zero may decide differently under another summation order, and flips
    #  the gradient of the 7 x 7 pixels below it: compare pixel by pixel and allow a handful of such neighbourhoods)
    def bad_pixels(a, r, tol):
        a, r = a.detach().cpu().double(), r.detach().cpu().double()
        return float(((a - r).abs().amax(dim=1) > tol * float(r.abs().max())).double().mean())
    assert bad_pixels(grads["2"], grads["0"], 2e-5) < 5e-3
    assert bad_pixels(grads["2"], xr.grad, 1e-4) < 5e-3


@pytest.mark.parametrize("fan_out", [False, True])
def test_premasked_gradients(gpu, monkeypatch, fan_out):
    """Chain of conv + ReLU layers whose data gradients run on k_conv_bfw: each dx leaves multiplied by the ReLU gradient
    of the layer below (srk_conv2d_backward_data_relu) and that layer skips its own masks.  Same numbers as with the
    masks applied by every layer itself (multiplications by 0 / 1 commute with nothing in between) and vs torch fp64.
    fan_out: the middle activation also feeds a skip connection -- its gradient is a sum, the mark must not survive."""
    pkg = _pkg()
    ops = pkg.ops
    monkeypatch.setenv("SRK_BFW", "1")
    monkeypatch.setenv("SRK_BF3_DIRECT", "0")   # (small problems: keep the K-split blocks of k_conv_bfd out of the comparison)
    ops.set_precision("mixed")
    N, C, H, W = 2, 64, 20, 33
    x = fill.randn((N, C, H, W), 301)
    ws = [fill.randn((C, C, 3, 3), 302 + i, (2.0 / (C * 9)) ** 0.5) for i in range(4)]
    g = fill.randn((N, C, H, W), 310)

    def chain(xx, ww, conv):
        h1 = conv(xx, ww[0])
        h2 = conv(h1, ww[1])
        h3 = conv(h2, ww[2])
        out = conv(h3, ww[3])
        return out + h2 if fan_out else out

    xr = x.double().requires_grad_(True)
    wr = [w.double().requires_grad_(True) for w in ws]
    chain(xr, wr, lambda a, w: torch.relu(torch.nn.functional.conv2d(a, w, None, 1, 1))).backward(g.double())
    res = {}
    for on in (False, True):
        monkeypatch.setattr(ops, "PREMASK", on)
        ops.PREMASK_STATS.update(masked_dx=0, masks_skipped=0)
        xg = x.to(gpu).requires_grad_(True)
        wg = [w.to(gpu).requires_grad_(True) for w in ws]
        cfg = ops.ConvCfg(1, 1, False, 0, 1, 0.0, 0)
        out = chain(xg, wg, lambda a, w: ops.conv2d(a, w, None, None, cfg))
        with ops.premasked_gradients():     # (what trainers._backward does; a plain backward() never pre-masks: below)
            out.backward(g.to(gpu))
        ops.flush_wgrads()
        res[on] = [xg.grad.clone()] + [w.grad.clone() for w in wg]
        if on:   # three data gradients write into a ReLU output; without the skip all three marks reach their layer
            assert ops.PREMASK_STATS["masked_dx"] == 3
            assert ops.PREMASK_STATS["masks_skipped"] == (2 if fan_out else 3)
        else:
            assert ops.PREMASK_STATS == {"masked_dx": 0, "masks_skipped": 0}
    for a, b in zip(res[False], res[True]):
        assert torch.equal(a, b)
    # outside the context the gradient of an intermediate activation is the standard one (nothing pre-masked)
    monkeypatch.setattr(ops, "PREMASK", True)
    ops.PREMASK_STATS.update(masked_dx=0, masks_skipped=0)
    xg = x.to(gpu).requires_grad_(True)
    wg = [w.to(gpu) for w in ws]
    h1 = ops.conv2d(xg, wg[0], None, None, ops.ConvCfg(1, 1, False, 0, 1, 0.0, 0))
    h1.retain_grad()
    ops.conv2d(h1, wg[1], None, None, ops.ConvCfg(1, 1, False, 0, 1, 0.0, 0)).backward(g.to(gpu))
    h1r = torch.relu(torch.nn.functional.conv2d(x.double(), ws[0].double(), None, 1, 1)).requires_grad_(True)
    torch.relu(torch.nn.functional.conv2d(h1r, ws[1].double(), None, 1, 1)).backward(g.double())
    assert ops.PREMASK_STATS == {"masked_dx": 0, "masks_skipped": 0}
    assert rel_err(h1.grad, h1r.grad.float()) < 1e-4
    assert rel_err(res[True][0], xr.grad.float()) < 1e-4
    for a, w in zip(res[True][1:], wr):
        assert rel_err(a, w.grad.float()) < 1e-4


@pytest.mark.parametrize("cin,cout,k,s,p,op,H,W", [
    (64, 8, 9, 4, 3, 1, 8, 8),     # data gradient = 9x9 stride-4 gather over 64 channels: halo chunk > half the LDS
    (56, 32, 8, 4, 2, 0, 6, 9),
    (64, 64, 5, 2, 2, 1, 12, 7),
])
def test_deconv_large_halo_small_problem(gpu, cin, cout, k, s, p, op, H, W):
    """Small-problem blocks whose halo chunk is large (big strided kernels): the K-split must not ask for two chunk
    buffers that exceed the LDS (found by tools/fuzz_deconv.py); forward and all gradients vs torch fp64."""
    pkg = _pkg()
    ops = pkg.ops
    x = fill.randn((2, cin, H, W), 31)
    w = fill.randn((cin, cout, k, k), 32, (2.0 / (cin * k * k)) ** 0.5)
    b = fill.randn((cout,), 33, 0.1)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    ref = torch.nn.functional.conv_transpose2d(xr, wr, br, s, p, op)
    g = fill.randn(tuple(ref.shape), 34)
    ref.backward(g.double())
    xg, wg, bg = (t.to(gpu).requires_grad_(True) for t in (x, w, b))
    y = ops.conv2d(xg, wg, bg, None, ops.ConvCfg(s, p, True, op, 0, 0.0, 0, ALGOS["auto"]))
    y.backward(g.to(gpu))
    assert rel_err(y, ref.detach().float()) < 1e-4
    assert rel_err(xg.grad, xr.grad.float()) < 1e-4
    assert rel_err(wg.grad, wr.grad.float()) < 1e-4
    assert rel_err(bg.grad, br.grad.float()) < 1e-4


@pytest.mark.parametrize("r,C", [(2, 64), (4, 3), (3, 2)])
def test_conv_fused_pixel_shuffle(gpu, r, C):
    """conv + PixelShuffle store (PSBlock, base_networks.py:179-181), forward and backward."""
    pkg = _pkg()
    ops = pkg.ops
    cin = 16
    x, w = fill.randn((2, cin, 7, 9), 1), fill.randn((C * r * r, cin, 3, 3), 2, 0.1)
    b = fill.randn((C * r * r,), 3, 0.1)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = torch.nn.functional.pixel_shuffle(torch.nn.functional.conv2d(xr, wr, br, 1, 1), r)
    g = fill.randn(tuple(ref.shape), 4)
    ref.backward(g)
    xg, wg, bg = (t.to(gpu).requires_grad_(True) for t in (x, w, b))
    y = ops.conv2d(xg, wg, bg, None, ops.ConvCfg(1, 1, False, 0, 0, 0.0, r))
    assert rel_err(y, ref) < 1e-4
    y.backward(g.to(gpu))
    assert rel_err(xg.grad, xr.grad) < 1e-4
    assert rel_err(wg.grad, wr.grad) < 1e-4
    assert rel_err(bg.grad, br.grad) < 1e-4


@pytest.mark.parametrize("r", [2, 4, 3])
def test_pixel_shuffle(gpu, ops_kat, r):
    pkg = _pkg()
    x = torch.from_numpy(ops_kat["ps.r%d.x" % r]).to(gpu).requires_grad_(True)
    y = pkg.ops.pixel_shuffle(x, r)
    assert rel_err(y, ops_kat["ps.r%d.y" % r]) == 0.0  # pure permutation: bit-exact
    y.backward(torch.from_numpy(ops_kat["ps.r%d.g" % r]).to(gpu))
    assert rel_err(x.grad, ops_kat["ps.r%d.dx" % r]) == 0.0


@pytest.mark.parametrize("r,C,N,H,W", [(4, 3, 2, 5, 70), (4, 3, 1, 3, 248), (2, 64, 2, 6, 13), (3, 3, 2, 4, 115), (2, 3, 1, 7, 257),
                                       (2, 1, 3, 5, 40), (4, 1, 1, 2, 1000), (8, 3, 1, 2, 19), (2, 5, 1, 3, 9), (3, 64, 1, 2, 7)])
@pytest.mark.parametrize("tile", ["1", "0"])
def test_pixel_shuffle_tiled_kernel(gpu, monkeypatch, r, C, N, H, W, tile):
    """The tiled pixel-shuffle kernel (16-byte runs on both sides, permutation in LDS; base_networks.py:157,181) on the (r, C)
    pairs it is instantiated for, with ragged last tiles, and the gather kernel it falls back to ((2, 5): no instance;
    SRK_PS_TILE=0): both directions bit-equal to torch's pixel_shuffle / pixel_unshuffle."""
    pkg = _pkg()
    monkeypatch.setenv("SRK_PS_TILE", tile)
    x = fill.randn((N, C * r * r, H, W), 17 + r).to(gpu).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = pkg.ops.pixel_shuffle(x, r)
    assert torch.equal(y, torch.nn.functional.pixel_shuffle(x.detach(), r))
    g = fill.randn(tuple(y.shape), 23).to(gpu)
    y.backward(g)
    assert torch.equal(x.grad, torch.nn.functional.pixel_unshuffle(g, r))


@pytest.mark.parametrize("name", ["relu", "prelu", "prelu_c", "lrelu", "tanh", "sigmoid"])
def test_activation(gpu, ops_kat, name):
    pkg = _pkg()
    kinds = {"relu": 1, "prelu": 2, "prelu_c": 2, "lrelu": 3, "tanh": 4, "sigmoid": 5}
    x = torch.from_numpy(ops_kat["act.%s.x" % name]).to(gpu).requires_grad_(True)
    pw = None
    if name.startswith("prelu"):
        pw = torch.from_numpy(ops_kat["act.%s.w" % name]).to(gpu).requires_grad_(True)
    y = pkg.ops.activation(x, kinds[name], 0.2, pw)
    assert rel_err(y, ops_kat["act.%s.y" % name]) < 1e-6
    y.backward(torch.from_numpy(ops_kat["act.%s.g" % name]).to(gpu))
    assert rel_err(x.grad, ops_kat["act.%s.dx" % name]) < 1e-6
    if pw is not None:
        assert rel_err(pw.grad, ops_kat["act.%s.dw" % name]) < 1e-5


@pytest.mark.parametrize("name", ["mse", "l1", "charbonnier", "bce"])
@pytest.mark.parametrize("target_layout", ["nchw", "nhwc"])
def test_losses(gpu, ops_kat, name, target_layout):
    pkg = _pkg()
    fn = {"mse": pkg.ops.mse_loss, "l1": pkg.ops.l1_loss, "charbonnier": pkg.ops.charbonnier_loss,
          "bce": pkg.ops.bce_loss}[name]
    p = torch.from_numpy(ops_kat["loss.pred"]).to(gpu).requires_grad_(True)
    t = torch.from_numpy(ops_kat["loss.target"]).to(gpu)
    if target_layout == "nhwc":
        t = pkg.ops.to_nhwc(t)
    l = fn(p, t)
    assert rel_err(l, ops_kat["loss.%s.value" % name]) < 1e-6
    l.backward()
    assert rel_err(p.grad, ops_kat["loss.%s.dpred" % name]) < 1e-6


def test_add_and_fork(gpu):
    pkg = _pkg()
    a, b = fill.randn((2, 5, 6, 7), 1), fill.randn((2, 5, 6, 7), 2)
    ag, bg = a.to(gpu).requires_grad_(True), b.to(gpu).requires_grad_(True)
    y = pkg.ops.add(ag, bg)
    assert rel_err(y, a + b) == 0.0
    u, v = pkg.ops.fork(ag)
    z = pkg.ops.add(pkg.ops.activation(u, 1), pkg.ops.activation(v, 4))
    z.backward(torch.ones_like(z))
    ar = a.clone().requires_grad_(True)
    (torch.relu(ar) + torch.tanh(ar)).sum().backward()
    assert rel_err(ag.grad, ar.grad) < 1e-6


def test_batchnorm_shared_double_call(gpu, ops_kat):
    """One BN applied twice in a forward (ResnetBlock's shared bn, base_networks.py:117,137,145):
    outputs, gradients and the twice-updated running statistics."""
    pkg = _pkg()
    bn = pkg.layers.BatchNorm2d(16)
    bn.weight.data.copy_(torch.from_numpy(ops_kat["bn.gamma"]))
    bn.bias.data.copy_(torch.from_numpy(ops_kat["bn.beta"]))
    bn.to(gpu).train()
    x = torch.from_numpy(ops_kat["bn.x"]).to(gpu).requires_grad_(True)
    y1 = bn(x)
    y2 = bn(_affine(pkg, y1))
    assert rel_err(y1, ops_kat["bn.y1"]) < TOL_TIGHT
    assert rel_err(y2, ops_kat["bn.y2"]) < TOL_TIGHT
    y2.backward(torch.from_numpy(ops_kat["bn.g"]).to(gpu))
    assert rel_err(x.grad, ops_kat["bn.dx"]) < 1e-4
    assert rel_err(bn.weight.grad, ops_kat["bn.dgamma"]) < 1e-4
    assert rel_err(bn.bias.grad, ops_kat["bn.dbeta"]) < 1e-4
    assert rel_err(bn.running_mean, ops_kat["bn.running_mean"]) < 1e-5
    assert rel_err(bn.running_var, ops_kat["bn.running_var"]) < 1e-5
    assert int(bn.num_batches_tracked) == 2
    bn.eval()
    with torch.no_grad():
        assert rel_err(bn(x), ops_kat["bn.eval_y"]) < TOL_TIGHT


@pytest.mark.parametrize("n,c,h,w", [(16, 64, 32, 32), (3, 72, 40, 37), (8, 512, 8, 8)])
def test_batchnorm_split_reduction_blocks(gpu, monkeypatch, n, c, h, w):
    """k_bn_reduce16 (16 channels per block, SRK_BN_RED16 = 1, default) against the 64-channel blocks it replaces and
    torch's float64 BatchNorm: outputs, input / parameter gradients, running statistics -- more than 64 row splits, channel
    counts that are not a multiple of 16 included."""
    pkg = _pkg()
    x0 = fill.randn((n, c, h, w), 61 + c) * 1.7 + 0.3
    g0 = fill.randn((n, c, h, w), 62 + c)
    ref = torch.nn.BatchNorm2d(c).double()
    ref.weight.data.copy_(fill.randn((c,), 63).double() * 0.2 + 1.0)
    ref.bias.data.copy_(fill.randn((c,), 64).double() * 0.1)
    xr = x0.double().requires_grad_(True)
    yr = ref(xr)
    yr.backward(g0.double())
    res = {}
    for tag in ("1", "0"):
        monkeypatch.setenv("SRK_BN_RED16", tag)
        bn = pkg.layers.BatchNorm2d(c)
        bn.weight.data.copy_(ref.weight.data.float())
        bn.bias.data.copy_(ref.bias.data.float())
        bn.to(gpu).train()
        x = x0.to(gpu).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        y = bn(x)
        y.backward(g0.to(gpu).contiguous(memory_format=torch.channels_last))
        res[tag] = (y.detach(), x.grad, bn.weight.grad, bn.bias.grad, bn.running_mean.clone(), bn.running_var.clone())
        assert rel_err(y, yr.detach().float()) < 2e-5
        assert rel_err(x.grad, xr.grad.float()) < 1e-4
        assert rel_err(bn.weight.grad, ref.weight.grad.float()) < 1e-4
        assert rel_err(bn.bias.grad, ref.bias.grad.float()) < 1e-4
        assert rel_err(bn.running_mean, ref.running_mean.float()) < 1e-5
        assert rel_err(bn.running_var, ref.running_var.float()) < 1e-5
    for a, b in zip(res["1"], res["0"]):    # double sums in another order: equal to fp32 rounding of the results
        assert rel_err(a, b) < 1e-6


class _Affine(torch.autograd.Function):
    """y = 0.5*x + 0.1 built from srk_axpby (test helper; keeps the graph on our kernels)."""

    @staticmethod
    def forward(ctx, x, pkg):
        ctx.pkg = pkg
        return pkg.ops._axpby(x, torch.full_like(x, 0.1), 0.5, 1.0)

    @staticmethod
    def backward(ctx, g):
        return ctx.pkg.ops._axpby(g, g, 0.5, 0.0), None


def _affine(pkg, x):
    return _Affine.apply(x, pkg)


def test_linear(gpu, ops_kat):
    pkg = _pkg()
    x = torch.from_numpy(ops_kat["fc.x"]).to(gpu).requires_grad_(True)
    w = torch.from_numpy(ops_kat["fc.w"]).to(gpu).requires_grad_(True)
    b = torch.from_numpy(ops_kat["fc.b"]).to(gpu).requires_grad_(True)
    y = pkg.ops.linear(x, w, b, 3, 0.2)
    assert rel_err(y, ops_kat["fc.y"]) < TOL_TIGHT
    y.backward(torch.from_numpy(ops_kat["fc.g"]).to(gpu))
    assert rel_err(x.grad, ops_kat["fc.dx"]) < TOL_TIGHT
    assert rel_err(w.grad, ops_kat["fc.dw"]) < TOL_TIGHT
    assert rel_err(b.grad, ops_kat["fc.db"]) < TOL_TIGHT


@pytest.mark.parametrize("shape", [(16, 18432, 1024), (5, 2048, 70), (20, 4096, 65), (1, 2052, 64)])
def test_linear_wide_layers(gpu, shape):
    """The weight-streaming kernels of wide dense layers (SRGAN discriminator 512*6*6 -> 1024, srgan.py:66-70; taken
    for In >= 2048, Out >= 64) against float64 matmuls: forward with fused LeakyReLU, dx, dw (beta = 0 and accumulate),
    db; batch above / below the 16-row pass, Out not a multiple of the 4- / 16-row blocks, In not a multiple of 64."""
    pkg = _pkg()
    B, In, Out = shape
    x = fill.randn((B, In), 71).to(gpu).requires_grad_(True)
    w = (fill.randn((Out, In), 72) * 0.02).to(gpu).requires_grad_(True)
    b = fill.randn((Out,), 73).to(gpu).requires_grad_(True)
    g = fill.randn((B, Out), 74).to(gpu)
    y = pkg.ops.linear(x, w, b, 3, 0.2)   # 3 = LeakyReLU
    z = x.detach().double().cpu() @ w.detach().double().cpu().t() + b.detach().double().cpu()
    yr = torch.where(z > 0, z, 0.2 * z)
    assert_close_elementwise(y, yr, 2e-6, what="wide linear forward")
    y.backward(g)
    gz = g.double().cpu() * torch.where(z > 0, 1.0, 0.2)
    assert_close_elementwise(x.grad, gz @ w.detach().double().cpu(), 2e-6, what="wide linear dx")
    assert_close_elementwise(w.grad, gz.t() @ x.detach().double().cpu(), 2e-6, what="wide linear dw")
    assert_close_elementwise(b.grad, gz.sum(0), 2e-6, what="wide linear db")
    first = w.grad.clone()
    y2 = pkg.ops.linear(x, w, b, 3, 0.2)
    y2.backward(g)                           # autograd accumulates: 2x
    assert_close_elementwise(w.grad, 2 * first, 1e-6, what="wide linear dw accumulated")


@pytest.mark.parametrize("act", ["relu", "lrelu", "prelu1", "preluC", "none+res", "prelu1+res"])
def test_batchnorm_with_folded_activation_and_residual(gpu, act):
    """act(bn(x)) [+ residual] in the BatchNorm's own launches (srk_bn_apply_act / _backward_*_act: the backward
    recomputes z from x) against torch's float64 BatchNorm -> activation -> add on the CPU: output, dx, dgamma, dbeta,
    dprelu, d(residual); training statistics, running statistics untouched by the fusion."""
    import torch.nn.functional as F
    pkg = _pkg()
    ops = pkg.ops
    n, c, h, w = 3, 8, 7, 9
    x = fill.randn((n, c, h, w), 81) * 1.7 + 0.3
    res = fill.randn((n, c, h, w), 82) if act.endswith("+res") else None
    dy = fill.randn((n, c, h, w), 83)
    gamma, beta = fill.rand((c,), 84, 0.5, 1.5), fill.randn((c,), 85) * 0.3
    kind = act.split("+")[0]
    pw = None
    if kind.startswith("prelu"):
        pw = fill.rand((1 if kind == "prelu1" else c,), 86, 0.1, 0.4)
    code = {"relu": pkg._lib.ACT_RELU, "lrelu": pkg._lib.ACT_LRELU, "prelu1": pkg._lib.ACT_PRELU,
            "preluC": pkg._lib.ACT_PRELU, "none": pkg._lib.ACT_NONE}[kind]
    # float64 reference
    xr = x.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    pr = pw.double().requires_grad_(True) if pw is not None else None
    rr = res.double().requires_grad_(True) if res is not None else None
    z = F.batch_norm(xr, None, None, gr, br, True, 0.1, 1e-5)
    if kind == "relu":
        yr = F.relu(z)
    elif kind == "lrelu":
        yr = F.leaky_relu(z, 0.2)
    elif pw is not None:
        yr = F.prelu(z, pr)
    else:
        yr = z
    if rr is not None:
        yr = yr + rr
    yr.backward(dy.double())
    # fused path
    xg = x.to(gpu).requires_grad_(True)
    gg, bg = gamma.to(gpu).requires_grad_(True), beta.to(gpu).requires_grad_(True)
    pg = pw.to(gpu).requires_grad_(True) if pw is not None else None
    rg = res.to(gpu).requires_grad_(True) if res is not None else None
    rm, rv = torch.zeros(c, device=gpu), torch.ones(c, device=gpu)
    assert ops.bn_fusable(xg, code, pg)
    y = ops.batch_norm(xg, gg, bg, rm, rv, True, 0.1, 1e-5, None, None, code, 0.2, pg, rg)
    y.backward(dy.to(gpu))
    assert_close_elementwise(y, yr.detach(), 2e-6, what="fused bn+act forward")
    assert_close_elementwise(xg.grad, xr.grad, 2e-5, what="dx")
    assert_close_elementwise(gg.grad, gr.grad, 2e-6, what="dgamma")
    assert_close_elementwise(bg.grad, br.grad, 2e-6, what="dbeta")
    if pw is not None:
        assert_close_elementwise(pg.grad, pr.grad, 2e-6, what="dprelu")
    if res is not None:
        assert torch.equal(rg.grad.cpu(), dy)
    # running statistics as nn.BatchNorm2d updates them (momentum 0.1, unbiased variance)
    m = x.double().mean((0, 2, 3))
    v = x.double().var((0, 2, 3), unbiased=True)
    assert_close_elementwise(rm, 0.1 * m, 2e-6, what="running_mean")
    assert_close_elementwise(rv, 0.9 + 0.1 * v, 2e-6, what="running_var")
    # error behaviour: unsupported activations are refused, not mis-computed
    lib = pkg._lib.load()
    P = pkg._lib.ptr
    xs = xg.detach().contiguous(memory_format=torch.channels_last)
    rc = lib.srk_bn_apply_act(P(xs), P(torch.empty_like(xs)), P(rm), P(rv), None, None, n * h * w, c, pkg._lib.ACT_BY_NAME["tanh"]
                              if hasattr(pkg._lib, "ACT_BY_NAME") else 4, 0.0, None, 0, None, None, pkg._lib.stream_ptr())
    assert rc != 0


@pytest.mark.parametrize("shape", [(16, 64, 32, 32), (5, 16, 9, 11), (8, 512, 8, 8), (3, 128, 37, 20), (16, 64, 64, 64)])
@pytest.mark.parametrize("act", ["none", "lrelu", "prelu1", "preluC", "none+res", "prelu1+res"])
def test_batchnorm_finalize_in_apply(gpu, monkeypatch, shape, act):
    """Round 6: the apply kernels finish the split reduction themselves (srk_bn_finalize_apply_act /
    srk_bn_backward_finalize_apply_act: one launch less per BatchNorm and direction).  Against torch's float64 BatchNorm ->
    activation -> add, and against the separate-launch path (SRK_BN_FIN_APPLY=0): the forward statistics and outputs of the two
    paths must be bit-equal (same summation order, same z), the backward equal to fp32 rounding; one and several row ranges per
    slab, 1 .. 32 slabs, ragged last range, 64 .. 512 row splits."""
    import torch.nn.functional as F
    pkg = _pkg()
    ops = pkg.ops
    n, c, h, w = shape
    x = fill.randn((n, c, h, w), 281) * 1.7 + 0.3
    res = fill.randn((n, c, h, w), 282) if act.endswith("+res") else None
    dy = fill.randn((n, c, h, w), 283)
    gamma, beta = fill.rand((c,), 284, 0.5, 1.5), fill.randn((c,), 285) * 0.3
    kind = act.split("+")[0]
    pw = fill.rand((1 if kind == "prelu1" else c,), 286, 0.1, 0.4) if kind.startswith("prelu") else None
    code = {"lrelu": pkg._lib.ACT_LRELU, "prelu1": pkg._lib.ACT_PRELU, "preluC": pkg._lib.ACT_PRELU, "none": pkg._lib.ACT_NONE}[kind]
    xr = x.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    pr = pw.double().requires_grad_(True) if pw is not None else None
    z = F.batch_norm(xr, None, None, gr, br, True, 0.1, 1e-5)
    yr = F.leaky_relu(z, 0.2) if kind == "lrelu" else (F.prelu(z, pr) if pw is not None else z)
    if res is not None:
        yr = yr + res.double()
    yr.backward(dy.double())
    assert pkg._lib.load().srk_bn_fused_supported(c) == 1
    out = {}
    for fin in (True, False):
        monkeypatch.setattr(ops, "BN_FIN_APPLY", fin)
        xg = x.to(gpu).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        gg, bg = gamma.to(gpu).requires_grad_(True), beta.to(gpu).requires_grad_(True)
        pg = pw.to(gpu).requires_grad_(True) if pw is not None else None
        rg = res.to(gpu).contiguous(memory_format=torch.channels_last) if res is not None else None
        rm, rv = torch.zeros(c, device=gpu), torch.ones(c, device=gpu)
        nbt = torch.zeros((), dtype=torch.int64, device=gpu)
        y = ops.batch_norm(xg, gg, bg, rm, rv, True, 0.1, 1e-5, None, nbt, code, 0.2, pg, rg)
        y.backward(dy.to(gpu).contiguous(memory_format=torch.channels_last))
        out[fin] = (y.detach(), rm, rv, xg.grad, gg.grad, bg.grad, None if pg is None else pg.grad)
        assert int(nbt) == 1
        assert_close_elementwise(y, yr.detach(), 4e-6, what="forward")
        assert_close_elementwise(xg.grad, xr.grad, 5e-5, what="dx")
        assert_close_elementwise(gg.grad, gr.grad, 1e-5, what="dgamma")
        assert_close_elementwise(bg.grad, br.grad, 1e-5, what="dbeta")
        if pw is not None:
            assert_close_elementwise(pg.grad, pr.grad, 1e-5, what="dprelu")
        m = x.double().mean((0, 2, 3))
        v = x.double().var((0, 2, 3), unbiased=True)
        assert_close_elementwise(rm, 0.1 * m, 2e-6, what="running_mean")
        assert_close_elementwise(rv, 0.9 + 0.1 * v, 2e-6, what="running_var")
    a, b = out[True], out[False]
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])          # statistics: the reduce kernel's summation order
    if act != "none":                                                    # (the plain apply kernel writes its affine differently)
        assert torch.equal(a[0], b[0])
    for u, v_ in zip(a[3:], b[3:]):
        if u is not None:
            assert rel_err(u, v_) < 2e-6


def test_declared_input_bound_replaces_the_absmax_pass(gpu):
    """ops.declare_absmax(x, bound): a caller-declared upper bound of |x| stands in for the srk_absmax pass in front of an
    fp32-faithful (f16x3) first layer -- the ABI only asks for slots >= max|x|.  ESPCN forward on a [0, 1) batch: same accuracy
    against fp64 with the measured maximum, the tight bound 1.0 and a loose bound 8.0 (three bits of 22 given away)."""
    from oracle import ref_modules as R
    pkg = _pkg()
    ops = pkg.ops
    ops.set_precision("mixed")
    ora = fill.fill_module(R.ESPCN(3, 64, 4))
    net = pkg.ESPCNNet(3, 64, 4)
    net.load_state_dict(ora.state_dict())
    net.to(gpu).eval()
    x = fill.rand((2, 3, 300, 260), 401)          # (large enough for the f16x3 class to be taken)
    ref = ora.double()(x.double())
    xg = x.to(gpu)
    with torch.no_grad():
        y0 = net(xg.view(xg.shape))
        assert getattr(xg, "_srk_amax", None) is None
        y1 = net(ops.declare_absmax(xg.view(xg.shape), 1.0))
        y8 = net(ops.declare_absmax(xg.view(xg.shape), 8.0))
    assert float(ops.declare_absmax(xg.view(xg.shape), 1.0)._srk_amax[0].max()) == 1.0
    for y, bar in ((y0, 2e-6), (y1, 2e-6), (y8, 4e-6)):
        assert rel_err(y, ref.float()) < bar


def test_layout_roundtrip_and_ragged(gpu):
    """NCHW<->NHWC copies at ragged sizes (non multiples of the 32x32 transpose tile)."""
    pkg = _pkg()
    for shape in ((1, 1, 1, 1), (2, 3, 5, 7), (3, 33, 9, 31), (1, 64, 41, 41)):
        x = fill.randn(shape, 11).to(gpu)
        y = pkg.ops.to_nhwc(x)
        assert torch.equal(y.cpu(), x.cpu())
        assert y.stride() == x.contiguous(memory_format=torch.channels_last).stride() or shape[1] == 1 or shape[2] * shape[3] == 1
        z = pkg.ops.to_nchw(y)
        assert z.is_contiguous() and torch.equal(z.cpu(), x.cpu())


def test_errors_are_loud(gpu):
    pkg = _pkg()
    with pytest.raises(RuntimeError):
        pkg.ops.conv2d(torch.zeros(1, 3, 8, 8), torch.zeros(4, 3, 3, 3))  # CPU tensors: no fallback
    with pytest.raises(RuntimeError):
        pkg.ops.conv2d(torch.zeros(1, 3, 2, 2, device=gpu), torch.zeros(4, 3, 5, 5, device=gpu))  # empty output
    with pytest.raises(RuntimeError):
        pkg.ops.pixel_shuffle(torch.zeros(1, 6, 2, 2, device=gpu), 2)
    lib = pkg._lib.load()
    assert lib.srk_axpby(None, None, None, 0, 1.0, 1.0, None) == -1
    assert b"null" in lib.srk_last_error_string()


class _OneParam(torch.nn.Module):
    def __init__(self, p0):
        super(_OneParam, self).__init__()
        self.p = torch.nn.Parameter(p0.clone())


@pytest.mark.parametrize("kind", ["srcnn", "fsrcnn", "vdsr", "edsr", "srgan_d"])
def test_optimizers_kat(gpu, ops_kat, kind):
    """srk_sgd_step / srk_adam_step against the reference's optimizer choices (srcnn.py:79 plain SGD; fsrcnn.py:105
    momentum; vdsr.py:86-90 momentum + weight decay; edsr.py:93 Adam with the device step counter / bias correction;
    srgan.py:149 Nesterov at lr/100): 3 steps on a flat 1000-vector with the committed gradients."""
    pkg = _pkg()
    mod = _OneParam(torch.from_numpy(ops_kat["opt.p0"])).to(gpu)
    flat = pkg.optim.FlatParams(mod)
    opt = pkg.optim.make_optimizer(kind, flat, 1e-2)
    for g in ops_kat["opt.grads"]:
        opt.zero_grad()
        flat.grad[:1000].copy_(torch.from_numpy(g).to(gpu))
        opt.step()
    want = ops_kat["opt.%s.final" % kind]
    got = mod.p.detach().cpu().numpy()
    assert float(np.abs(got - ops_kat["opt.p0"]).max()) > 1e-4          # the parameters did move
    assert float(np.abs(got - want).max()) <= 1e-6 * max(1.0, float(np.abs(want).max())), kind
    if kind == "edsr":
        assert opt.step_dev.tolist() == [3, 0]      # {step count, the kernel's arrival ticket back at 0}


def test_grad_norm_clip_kat(gpu, ops_kat):
    """srk_grad_norm_clip = torch.nn.utils.clip_grad_norm (vdsr.py:149): the norm, and the scale the next optimizer
    step applies (the flat gradient itself stays unscaled)."""
    pkg = _pkg()
    mod = _OneParam(torch.from_numpy(ops_kat["opt.p0"])).to(gpu)
    flat = pkg.optim.FlatParams(mod)
    opt = pkg.optim.SGD(flat, 1.0)
    g = torch.from_numpy(ops_kat["opt.grads"][0]).to(gpu) * 3
    opt.zero_grad()
    flat.grad[:1000].copy_(g)
    norm = opt.clip_grad_norm(0.4)
    assert abs(float(norm) - float(ops_kat["opt.clip.norm"])) <= 1e-6 * float(ops_kat["opt.clip.norm"])
    clipped = (flat.grad[:1000] * opt.scale_dev).cpu().numpy()
    assert float(np.abs(clipped - ops_kat["opt.clip.grad"]).max()) <= 1e-6 * float(np.abs(ops_kat["opt.clip.grad"]).max())
    # the SGD step (lr 1) applies exactly that scaled gradient
    p_before = mod.p.detach().clone()
    opt.step()
    step = (p_before - mod.p.detach()).cpu().numpy()
    assert float(np.abs(step - ops_kat["opt.clip.grad"]).max()) <= 2e-6 * float(np.abs(ops_kat["opt.clip.grad"]).max()) + 1e-7
    # a gradient already inside the ball is left alone (scale 1)
    opt.zero_grad()
    flat.grad[:1000].copy_(g * 1e-3)
    opt.clip_grad_norm(0.4)
    assert float(opt.scale_dev) == 1.0


@pytest.mark.parametrize("n,N,cin,cout,k,p,H,W,act,bias", [
    (5, 2, 64, 64, 3, 1, 12, 12, "relu", True),     # EDSR / SRResNet body class: two input-channel chunks per layer
    (7, 1, 64, 64, 3, 1, 33, 21, None, False),      # VDSR class (bias-free), ragged tiles
    (3, 2, 32, 48, 3, 0, 17, 19, "lrelu", True),    # one chunk, 48 output channels, no padding
    (4, 1, 24, 16, 1, 0, 9, 40, None, True),        # 1x1, 16 output channels (one-tile waves)
    (2, 1, 16, 32, 5, 2, 14, 14, "relu", True),     # 5x5: no grouped kernel -> the per-layer path behind the same call
    (45, 1, 64, 64, 3, 1, 8, 8, "relu", True),      # more layers than one launch holds (chunks of 40)
])
def test_wgrad_grouped(gpu, n, N, cin, cout, k, p, H, W, act, bias):
    """srk_conv2d_backward_weight_grouped: the weight (+ bias) gradients of n same-geometry convs in one call against
    n per-layer srk_conv2d_backward_weight calls (same kernels, other split-K counts) and torch fp64; beta = 1
    accumulates; layers with and without an activation mask mix freely."""
    pkg = _pkg()
    lib, L = pkg._lib.load(), pkg._lib
    OH = lib.srk_conv_out_dim(H, k, 1, p, 0, 0)
    OW = lib.srk_conv_out_dim(W, k, 1, p, 0, 0)
    d = L.ConvDesc(N, H, W, cin, OH, OW, cout, k, k, 1, p, 0, 0, 0, 0, 0)
    st = L.stream_ptr()
    slope = 0.2 if act == "lrelu" else 0.0
    xs, dys, ys, dws, dbs, refs = [], [], [], [], [], []
    for l in range(n):
        x = fill.randn((N, cin, H, W), 900 + l)
        dy = fill.randn((N, cout, OH, OW), 950 + l)
        y = fill.randn((N, cout, OH, OW), 990 + l) if (act and l % 3 != 2) else None   # every third layer unmasked
        dym = dy.double()
        if y is not None:
            dym = torch.where(y > 0, dym, dym * slope)
        xr = x.double().requires_grad_(True)
        wr = torch.zeros(cout, cin, k, k, dtype=torch.float64, requires_grad=True)
        br = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
        torch.nn.functional.conv2d(xr, wr, br, 1, p).backward(dym)
        refs.append((wr.grad, br.grad))
        cl = lambda t: t.to(gpu).contiguous(memory_format=torch.channels_last)
        xs.append(cl(x)); dys.append(cl(dy)); ys.append(None if y is None else cl(y))
        dws.append(torch.full((cout, cin, k, k), 0.5, device=gpu)); dbs.append(torch.full((cout,), -0.25, device=gpu) if bias else None)
    vp = ctypes.c_void_p
    arr = lambda ts: (vp * n)(*[None if t is None else t.data_ptr() for t in ts])
    masks = (L.BwdMask * n)(*[L.BwdMask(None if y is None else y.data_ptr(), slope) for y in ys])
    nbytes = int(lib.srk_conv2d_backward_weight_grouped_workspace_bytes(ctypes.byref(d), n))
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=gpu)
    L.check(lib.srk_conv2d_backward_weight_grouped(ctypes.byref(d), n, arr(xs), arr(dys), masks, arr(dws),
                                                   arr(dbs) if bias else None, 1.0, L.ptr(ws), ws.numel(), st), "grouped")
    # per-layer reference through the single-layer entry point
    nb1 = int(lib.srk_conv2d_backward_weight_workspace_bytes(ctypes.byref(d)))
    ws1 = torch.empty(max(nb1, 16), dtype=torch.uint8, device=gpu)
    for l in range(n):
        dw1 = torch.full((cout, cin, k, k), 0.5, device=gpu)
        db1 = torch.full((cout,), -0.25, device=gpu) if bias else None
        m = L.BwdMask(None if ys[l] is None else ys[l].data_ptr(), slope)
        L.check(lib.srk_conv2d_backward_weight(ctypes.byref(d), L.ptr(xs[l]), L.ptr(dys[l]),
                                               ctypes.byref(m) if ys[l] is not None else None, L.ptr(dw1), L.ptr(db1), 1.0,
                                               L.ptr(ws1), ws1.numel(), st), "single")
        assert rel_err(dws[l], dw1) < 2e-5, l
        assert rel_err(dws[l] - 0.5, refs[l][0].float()) < 1e-4, l          # beta = 1: accumulated onto the 0.5 fill
        if bias:
            assert rel_err(dbs[l], db1) < 2e-5, l
            assert rel_err(dbs[l] + 0.25, refs[l][1].float()) < 1e-4, l
    # shared weights in one call are refused (LapSRN's aliased branches go into separate calls)
    if n >= 2:
        dup = (vp * 2)(dws[0].data_ptr(), dws[0].data_ptr())
        rc = lib.srk_conv2d_backward_weight_grouped(ctypes.byref(d), 2, arr(xs[:2] + [None] * (n - 2)), arr(dys[:2] + [None] * (n - 2)),
                                                    masks, dup, None, 1.0, L.ptr(ws), ws.numel(), st)
        assert rc == -1 and b"same dw" in lib.srk_last_error_string()


@pytest.mark.parametrize("N,cin,cout,H,W,p,act,bias,ps_r,n", [
    (32, 64, 64, 32, 32, 1, "relu", True, 0, 1),     # EDSR body layer: 4 x 32 tiles, ring of halo rows, ReLU mask
    (16, 64, 64, 41, 41, 1, None, False, 0, 1),      # VDSR body layer: 2 x 48 tiles, ragged right / bottom edges, K steps across rows
    (8, 64, 256, 40, 36, 1, None, True, 2, 1),       # EDSR up-sampler: dY handed over pixel-shuffled, four 64-channel blocks
    (64, 32, 128, 23, 29, 0, "lrelu", True, 0, 1),   # one input-channel chunk, no padding, LeakyReLU mask, ragged everything
    (4, 128, 64, 64, 64, 1, None, True, 0, 1),       # four input-channel chunks
    (16, 64, 64, 32, 32, 1, "relu", True, 0, 3),     # grouped launch (three layers, the middle one unmasked)
])
def test_wgrad_transpose_read_kernel(gpu, monkeypatch, N, cin, cout, H, W, p, act, bias, ps_r, n):
    """k_wgrad_tr (pixel-major LDS image, ds_read_b64_tr_b16 fragments) against k_wgrad_bf (SRK_WGRAD_TR=0) and float64:
    the weight gradients of the two kernels are BIT-EQUAL (same products, same summation order per accumulator and slab),
    the bias gradients agree to fp32 summation noise; beta = 1 accumulates onto what the tensors held."""
    pkg = _pkg()
    lib, L = pkg._lib.load(), pkg._lib
    k = 3
    OH = lib.srk_conv_out_dim(H, k, 1, p, 0, 0)
    OW = lib.srk_conv_out_dim(W, k, 1, p, 0, 0)
    d = L.ConvDesc(N, H, W, cin, OH, OW, cout, k, k, 1, p, 0, 0, 0, 0, ps_r)
    st = L.stream_ptr()
    slope = 0.2 if act == "lrelu" else 0.0
    cl = lambda t: t.to(gpu).contiguous(memory_format=torch.channels_last)
    xs, dys, ys, refs = [], [], [], []
    for l in range(n):
        x = fill.randn((N, cin, H, W), 1300 + l)
        dy = fill.randn((N, cout, OH, OW), 1350 + l)
        y = fill.randn((N, cout, OH, OW), 1390 + l) if (act and l != 1) else None
        dym = dy.double()
        if y is not None:
            dym = torch.where(y > 0, dym, dym * slope)
        wr = torch.zeros(cout, cin, k, k, dtype=torch.float64, requires_grad=True)
        br = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
        torch.nn.functional.conv2d(x.double(), wr, br, 1, p).backward(dym)
        refs.append((wr.grad, br.grad))
        if ps_r > 1:   # the gradient as the pixel-shuffle's input gradient would arrive: [N, C / r^2, OH r, OW r]
            dy = torch.nn.functional.pixel_shuffle(dy, ps_r)
        xs.append(cl(x)); dys.append(cl(dy)); ys.append(None if y is None else cl(y))
    vp = ctypes.c_void_p
    arr = lambda ts: (vp * n)(*[None if t is None else t.data_ptr() for t in ts])
    out = {}
    for tag in ("1", "0"):
        monkeypatch.setenv("SRK_WGRAD_TR", tag)
        dws = [torch.full((cout, cin, k, k), 0.5, device=gpu) for _ in range(n)]
        dbs = [torch.full((cout,), -0.25, device=gpu) if bias else None for _ in range(n)]
        if n == 1:
            ws = torch.empty(int(lib.srk_conv2d_backward_weight_workspace_bytes(ctypes.byref(d))) + 16, dtype=torch.uint8, device=gpu)
            m = L.BwdMask(None if ys[0] is None else ys[0].data_ptr(), slope)
            L.check(lib.srk_conv2d_backward_weight(ctypes.byref(d), L.ptr(xs[0]), L.ptr(dys[0]),
                                                   ctypes.byref(m) if ys[0] is not None else None, L.ptr(dws[0]), L.ptr(dbs[0]), 1.0,
                                                   L.ptr(ws), ws.numel(), st), "wgrad")
        else:
            masks = (L.BwdMask * n)(*[L.BwdMask(None if y is None else y.data_ptr(), slope) for y in ys])
            ws = torch.empty(int(lib.srk_conv2d_backward_weight_grouped_workspace_bytes(ctypes.byref(d), n)) + 16, dtype=torch.uint8,
                             device=gpu)
            L.check(lib.srk_conv2d_backward_weight_grouped(ctypes.byref(d), n, arr(xs), arr(dys), masks, arr(dws),
                                                           arr(dbs) if bias else None, 1.0, L.ptr(ws), ws.numel(), st), "grouped")
        name = lib.srk_last_kernel_name().decode()
        assert name.startswith("k_wgrad_tr<" if tag == "1" else "k_wgrad_bf<2,2,2,spec"), name
        torch.cuda.synchronize()
        out[tag] = (dws, dbs)
    for l in range(n):
        assert torch.equal(out["1"][0][l], out["0"][0][l]), l
        assert rel_err(out["1"][0][l] - 0.5, refs[l][0].float()) < 1e-4, l
        if bias:
            assert rel_err(out["1"][1][l], out["0"][1][l]) < 2e-6, l
            assert rel_err(out["1"][1][l] + 0.25, refs[l][1].float()) < 2e-5, l


@pytest.mark.parametrize("scale_x,scale_w", [(1.0, 0.05), (1e-5, 40.0), (3e4, 1e-6), (1e-30, 1e20)])
def test_f16x3_is_fp32_faithful_at_any_operand_scale(gpu, scale_x, scale_w):
    """SRK_ALGO_MFMA_F16X3 (two fp16 planes of the operands scaled by exact powers of two from their running maxima):
    error against float64 at the level of the exact-fp32 MFMA kernel -- far from fp16's own range in either direction,
    where unscaled fp16 planes would flush to zero or overflow -- and ReLU decisions that agree wherever float64 can
    decide them.  The output's running maximum (what the next layer scales by) is exact."""
    pkg = _pkg()
    ops, lib = pkg.ops, pkg._lib.load()
    x = (fill.randn((4, 64, 24, 20), 71).clamp(min=0) * scale_x)
    w = fill.randn((64, 64, 3, 3), 72) * scale_w
    b = fill.randn((64,), 73) * (2.4 * scale_x * scale_w)
    ref = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1))
    out = {}
    for name, algo in (("f16x3", pkg._lib.ALGO_MFMA_F16X3), ("fp32", pkg._lib.ALGO_MFMA), ("bf16x3", pkg._lib.ALGO_MFMA_BF16X3)):
        cfg = ops.ConvCfg(1, 1, False, 0, 1, 0.0, 0, algo)
        with torch.no_grad():
            out[name] = ops.conv2d_infer(x.to(gpu), w.to(gpu), b.to(gpu), None, cfg)
        if name == "f16x3":
            assert ",f16" in lib.srk_last_kernel_name().decode()
            slots = out[name]._srk_amax[0]
            assert float(slots.max()) == float(out[name].abs().max())
    err = {k: float((v.cpu().double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()) for k, v in out.items()}
    assert err["f16x3"] < 6e-7 and err["f16x3"] < 2.0 * err["fp32"] + 1e-8, err     # fp32-class ...
    assert err["f16x3"] < 0.25 * err["bf16x3"], err                                 # ... an order below the 3-term bf16 split
    decidable = ref.abs() > 1e-5 * float(ref.pow(2).mean().sqrt())
    assert bool((((out["f16x3"].cpu() > 0) == (ref > 0)) | ~decidable).all())


def test_forward_under_inference_mode(gpu):
    """Tensors created under torch.inference_mode() track no version counter (reading `_version` raises): the running-
    maximum / ReLU-output tags must not touch it.  The default precision runs every inference forward through the tagged
    f16x3 / bf16x6 path, so this is simply `with torch.inference_mode(): net(x)` -- same values as under no_grad."""
    pkg = _pkg()
    torch.manual_seed(5)
    net = pkg.ESPCNNet(3, 64, 4)
    net.weight_init()
    net.to(gpu).eval()
    x = torch.rand(2, 3, 40, 36, device=gpu)
    with torch.no_grad():
        ref = net(x)
    with torch.inference_mode():
        xi = x.clone()                       # an inference tensor as the network input, too
        assert xi.is_inference()
        y = net(xi)
        assert torch.equal(y, ref)
        # ops that tag / test tags directly: fork, to_nhwc, a block with a fused skip
        a, b = pkg.ops.fork(y)
        assert a.data_ptr() == y.data_ptr() and b.data_ptr() == y.data_ptr()
        edsr = pkg.EDSRNet(3, 64, 2)
        edsr.weight_init()
        edsr.to(gpu).eval()
        assert torch.isfinite(edsr(xi)).all()


def test_pack_batched_amax_scans_an_unaligned_layer_completely(gpu):
    """srk_pack_weights_batched with a weight offset that is not 16-byte aligned takes the scalar max|w| scan (256 floats
    per block and stride): its blocks must cover the whole layer.  The largest weight sits at the END of a 2016-element
    filter; the fp16 planes (scaled by 2^kw from that maximum) must equal those of the same filter packed from an aligned
    offset -- an under-estimated maximum would overflow them to inf."""
    pkg = _pkg()
    lib = pkg._lib.load()
    cout, cin, k = 8, 28, 3
    elems = cout * cin * k * k
    w = fill.randn((elems,), 91) * 0.05
    w[-1] = 900.0
    params = torch.zeros(2 * elems + 64, dtype=torch.float32, device=gpu)
    params[4:4 + elems] = w.to(gpu)                   # aligned copy (offset 4 floats = 16 bytes)
    params[elems + 33:elems + 33 + elems] = w.to(gpu)  # unaligned copy (offset % 4 == 1)
    nf = int(lib.srk_packed_weight_bytes(cout, cin, k, k, 0))
    nb = int(lib.srk_packed_weight_bytes(cout, cin, k, k, 1))
    pad = lambda n: (n + 255) // 256 * 256
    offs, rows, off = [], [], 0
    for w_off in (4, elems + 33):
        fo, bo = off, off + pad(nf)
        off = bo + pad(nb)
        offs.append((fo, bo))
        rows.append([w_off, fo, bo, cout, cin, k, k, 0, 0, -1, -1, -1, -1, 0])
    scratch = off
    for i, r in enumerate(rows):
        r[11] = scratch + 64 * i
    buf = torch.zeros(scratch + 256, dtype=torch.uint8, device=gpu)
    table = torch.tensor(rows, dtype=torch.int64, device=gpu)
    rc = lib.srk_pack_weights_batched(pkg._lib.ptr(params), pkg._lib.ptr(buf), pkg._lib.ptr(table), 2, -8, None, 0,
                                      pkg._lib.stream_ptr())
    assert rc == 0, lib.srk_last_error_string()
    torch.cuda.synchronize()
    a = buf[offs[0][0]:offs[0][0] + nf].cpu()
    b = buf[offs[1][0]:offs[1][0] + nf].cpu()
    assert torch.equal(a, b)
    assert torch.equal(buf[offs[0][1]:offs[0][1] + nb].cpu(), buf[offs[1][1]:offs[1][1] + nb].cpu())


def test_conv_says_whether_it_wrote_the_running_maximum(gpu, monkeypatch):
    """srk_last_conv_wrote_amax(): 1 after a forward whose kernel keeps the running maximum of its output in y_amax, 0 after
    one that does not (exact-fp32 kernels) -- ops.conv_forward_raw tags the output with the slots only in the first
    case, instead of inferring it from the kernel's name."""
    pkg = _pkg()
    ops, lib = pkg.ops, pkg._lib.load()
    x = fill.randn((2, 64, 20, 20), 3).to(gpu)
    w = (fill.randn((64, 64, 3, 3), 4) * 0.05).to(gpu)
    monkeypatch.setattr(ops, "F16X3_ALWAYS", True)   # (a problem this small would not ask for the maximum otherwise)
    with torch.no_grad():
        y = ops.conv2d_infer(x, w, None, None, ops.ConvCfg(1, 1, False, 0, 1, 0.0, 0, pkg._lib.ALGO_MFMA_BF16X6))
        assert lib.srk_last_conv_wrote_amax() == 1
        assert y._srk_amax[0] is not None and float(y._srk_amax[0].max()) == float(y.abs().max())
        # the exact-fp32 kernels keep the maximum on their 16-byte store path since round 4 (a first layer on them used to
        # cost the trunk behind it an srk_absmax pass) ...
        ya = torch.zeros(pkg._lib.AMAX_FLOATS, device=gpu)
        d = ops._make_desc(x.shape, w, ops.ConvCfg(1, 1, False, 0, 1, 0.0, 0, pkg._lib.ALGO_MFMA))
        xs = x.contiguous(memory_format=torch.channels_last)
        yy = torch.empty_like(xs)
        ep = pkg._lib.Epilogue(None, None, None, 0.0, 1, 0, 0, None, pkg._lib.ptr(ya))
        wp = ops.pack_weight_fwd(w, False, 0)
        rc = lib.srk_conv2d_forward(ctypes.byref(d), pkg._lib.ptr(xs), pkg._lib.ptr(wp), pkg._lib.ptr(yy), ctypes.byref(ep),
                                    pkg._lib.stream_ptr())
        assert rc == 0 and lib.srk_last_conv_wrote_amax() == 1
        assert float(ya.max()) == float(yy.abs().max())
        # the same answers in the caller's srk_conv_result (srk_conv2d_forward_ex; the srk_last_conv_* queries above are
        # deprecated aliases of its fields)
        res = pkg._lib.ConvResult()
        res.wrote_amax, res.bn_partial_rows = 7, 7
        assert lib.srk_conv2d_forward_ex(ctypes.byref(d), pkg._lib.ptr(xs), pkg._lib.ptr(wp), pkg._lib.ptr(yy), ctypes.byref(ep),
                                         ctypes.byref(res), pkg._lib.stream_ptr()) == 0
        assert (res.wrote_amax, res.bn_partial_rows) == (1, 0)
        # a caller compiled against a SHORTER struct (here: one that ends behind wrote_amax) gets only the fields it knows
        res.struct_size, res.wrote_amax, res.bn_partial_rows = 8, 7, 7
        assert lib.srk_conv2d_forward_ex(ctypes.byref(d), pkg._lib.ptr(xs), pkg._lib.ptr(wp), pkg._lib.ptr(yy), ctypes.byref(ep),
                                         ctypes.byref(res), pkg._lib.stream_ptr()) == 0
        assert (res.wrote_amax, res.bn_partial_rows) == (1, 7)
        assert lib.srk_conv2d_forward_ex(ctypes.byref(d), pkg._lib.ptr(xs), pkg._lib.ptr(wp), pkg._lib.ptr(yy), ctypes.byref(ep),
                                         None, pkg._lib.stream_ptr()) == 0
        # ... a call without y_amax says 0, and so does the shape-agnostic kernel
        ep = pkg._lib.Epilogue(None, None, None, 0.0, 1, 0, 0, None, None)
        assert lib.srk_conv2d_forward(ctypes.byref(d), pkg._lib.ptr(xs), pkg._lib.ptr(wp), pkg._lib.ptr(yy), ctypes.byref(ep),
                                      pkg._lib.stream_ptr()) == 0 and lib.srk_last_conv_wrote_amax() == 0
        d.algo = 1   # SRK_ALGO_GENERIC
        ep = pkg._lib.Epilogue(None, None, None, 0.0, 1, 0, 0, None, pkg._lib.ptr(ya))
        assert lib.srk_conv2d_forward(ctypes.byref(d), pkg._lib.ptr(xs), pkg._lib.ptr(wp), pkg._lib.ptr(yy), ctypes.byref(ep),
                                      pkg._lib.stream_ptr()) == 0 and lib.srk_last_conv_wrote_amax() == 0


def test_batchnorm_on_an_unaligned_view_takes_the_scalar_path(gpu):
    """A BatchNorm whose input is not 16-byte aligned cannot use the float4 kernels (which keep the running maximum of
    their output): the forward must fall back to the scalar apply kernel instead of raising."""
    pkg = _pkg()
    ops = pkg.ops
    base = torch.randn(1 + 2 * 8 * 8 * 8, device=gpu)
    x = base[1:].view(2, 8, 8, 8).permute(0, 3, 1, 2)        # [2, 8, 8, 8] NCHW view of NHWC storage at a 4-byte offset
    assert x.data_ptr() % 16 != 0 and ops._is_nhwc_dense(x)
    gamma, beta = torch.rand(8, device=gpu) + 0.5, torch.rand(8, device=gpu)
    rm, rv = torch.zeros(8, device=gpu), torch.ones(8, device=gpu)
    prev = ops.F16X3_ALWAYS
    ops.F16X3_ALWAYS = True       # (the small problem would not ask for the maximum otherwise)
    try:
        y = ops.batch_norm(x, gamma, beta, rm, rv, True)
    finally:
        ops.F16X3_ALWAYS = prev
    ref = torch.nn.functional.batch_norm(x.contiguous(), None, None, gamma, beta, True)
    assert rel_err(y, ref) < 1e-5


C64_SHAPES = [(16, 32, 32), (3, 13, 21), (1, 8, 8), (2, 5, 40), (1, 1, 1), (2, 17, 9)]


@pytest.mark.parametrize("shape", C64_SHAPES)
@pytest.mark.parametrize("bias,res", [(True, False), (False, True), (True, True)])
def test_conv_c64_small_problem_kernel(gpu, shape, bias, res):
    """k_c64 (conv_c64.hip): the 3x3 pad-1 64 -> 64 convolution of a small problem, one 8x8 tile per block -- what SRGAN's
    BatchNorm-separated generator convs (srgan.py:14-46 through base_networks.py:128-150) and EDSR's body-end conv run at
    the reference's batch of 16.  Forward in the three arithmetics and the data gradient (flipped taps, gradient fan-in
    add) against float64, through the C ABI; the dispatcher must really have picked the kernel."""
    import torch.nn.functional as F
    pkg = _pkg()
    ops, L, lib = pkg.ops, pkg._lib, pkg._lib.load()
    n, h, w = shape
    x = fill.randn((n, 64, h, w), 31 + h)
    wt = fill.randn((64, 64, 3, 3), 32 + w) * (2.0 / 576) ** 0.5
    b = fill.randn((64,), 33) * 0.1 if bias else None
    r = fill.randn((n, 64, h, w), 34) if res else None
    ref = F.conv2d(x.double(), wt.double(), None if b is None else b.double(), padding=1)
    if r is not None:
        ref = ref + r.double()
    xg, wg = x.to(gpu), wt.to(gpu)
    bg = None if b is None else b.to(gpu)
    rg = None if r is None else r.to(gpu)
    for algo, tol, tag in ((L.ALGO_MFMA_BF16X6, TOL_TIGHT, "k_c64<3,0>"), (L.ALGO_MFMA_F16X3, TOL_TIGHT, "k_c64<2,0,f16>"),
                           (L.ALGO_MFMA_BF16X3, 1e-4, "k_c64<2,0>")):
        with torch.no_grad():
            y = ops.conv2d_infer(xg, wg, bg, rg, ops.ConvCfg(1, 1, False, 0, 0, 0.0, 0, algo))
        assert lib.srk_last_kernel_name().decode() == tag
        assert rel_err(y, ref) < tol, (tag, rel_err(y, ref))
        if algo == L.ALGO_MFMA_F16X3:     # asked for by name: the output carries the running maximum of what was stored
            assert float(y._srk_amax[0].max()) == float(y.abs().max())
    # data gradient: dx = conv^T(dy) [+ add_to]
    dy = fill.randn((n, 64, h, w), 35)
    dref = F.conv_transpose2d(dy.double(), wt.double(), padding=1)
    if r is not None:
        dref = dref + r.double()
    CL = torch.channels_last
    dyg = dy.to(gpu).contiguous(memory_format=CL)
    addg = None if r is None else rg.contiguous(memory_format=CL)
    dx = torch.empty_like(dyg)
    d = ops._make_desc(xg.shape, wg, ops.ConvCfg(1, 1, False, 0, 0, 0.0, 0), "bwd")
    wpb = ops.pack_weight_bwd(wg, False, 0)
    rc = lib.srk_conv2d_backward_data(ctypes.byref(d), L.ptr(dyg), L.ptr(wpb), L.ptr(dx), None, L.ptr(addg), L.stream_ptr())
    assert rc == 0, lib.srk_last_error_string()
    assert lib.srk_last_kernel_name().decode() == "k_c64<2,1>"
    assert rel_err(dx, dref) < 1e-4
    # with an activation the layer stays on the general kernels
    with torch.no_grad():
        ops.conv2d_infer(xg, wg, bg, None, ops.ConvCfg(1, 1, False, 0, 1, 0.0, 0))
    assert not lib.srk_last_kernel_name().decode().startswith("k_c64")


@pytest.mark.parametrize("shape", [(4, 10), (16, 1024), (3, 257), (1, 5)])
def test_dense_block_instance_norm(gpu, shape):
    """DenseBlock(norm='instance') (base_networks.py:12-13): nn.InstanceNorm1d on the [B, F] output of the Linear, which
    torch reads as one unbatched sample of B channels x F positions -- a per-row normalisation with biased statistics
    (srk_rownorm_*).  Block output, input gradient and parameter gradients against the oracle's block (stock torch.nn)."""
    import warnings
    from oracle import ref_modules as R
    pkg = _pkg()
    b, f = shape
    ref = R.DenseBlock(24, f, activation='lrelu', norm='instance')
    fill.fill_module(ref, 77)
    blk = pkg.base_networks.DenseBlock(24, f, activation='lrelu', norm='instance').to(gpu)
    blk.load_state_dict(ref.state_dict())
    x = fill.randn((b, 24), 78)
    g = fill.randn((b, f), 79)
    xr = x.clone().requires_grad_(True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")      # ("input's size at dim=0 does not match num_features")
        yr = ref(xr)
        yr.backward(g)
    xg = x.to(gpu).requires_grad_(True)
    yg = blk(xg)
    yg.backward(g.to(gpu))
    assert rel_err(yg, yr) < 2e-5
    if f > 1:
        assert rel_err(xg.grad, xr.grad) < 1e-4
        assert rel_err(blk.fc.weight.grad, ref.fc.weight.grad) < 1e-4


@pytest.mark.parametrize("cin,cout,kh,kw,s,p,tr", [(8, 16, 3, 5, 1, 1, 0), (64, 64, 1, 3, 1, 1, 0), (3, 64, 5, 3, 1, 2, 0),
                                                   (64, 3, 3, 1, 1, 1, 0), (16, 24, 2, 3, 2, 1, 0), (12, 8, 4, 2, 2, 1, 1)])
@pytest.mark.parametrize("mode", ["mixed", "bf16x3", "fp32"])
def test_conv_non_square_kernels(gpu, cin, cout, kh, kw, s, p, tr, mode):
    """kernel_size = (kh, kw) with kh != kw (base_networks.py:42,77 hand kernel_size to torch.nn.Conv2d /
    ConvTranspose2d, which take pairs): forward, input gradient and parameter gradients of the blocks against float64, in the
    default, the fast and the exact arithmetic (every kernel family behind them must index taps by (kh, kw), not k^2)."""
    import torch.nn.functional as F
    pkg = _pkg()
    prev = pkg.ops.get_precision()
    pkg.ops.set_precision(mode)
    try:
        B = pkg.base_networks
        blk = (B.DeconvBlock if tr else B.ConvBlock)(cin, cout, (kh, kw), s, p, activation='lrelu', norm=None).to(gpu)
        w = fill.randn(tuple(blk.state_dict()[("deconv" if tr else "conv") + ".weight"].shape), 41) * (2.0 / (cin * kh * kw)) ** 0.5
        b = fill.randn((cout,), 42) * 0.1
        blk.load_state_dict({("deconv" if tr else "conv") + ".weight": w, ("deconv" if tr else "conv") + ".bias": b})
        x = fill.randn((2, cin, 11, 9), 43)
        xr = x.double().requires_grad_(True)
        wr, br = w.double().requires_grad_(True), b.double().requires_grad_(True)
        yr = F.leaky_relu((F.conv_transpose2d if tr else F.conv2d)(xr, wr, br, stride=s, padding=p), 0.2)
        g = fill.randn(tuple(yr.shape), 44)
        yr.backward(g.double())
        xg = x.to(gpu).requires_grad_(True)
        yg = blk(xg)
        assert tuple(yg.shape) == tuple(yr.shape)
        yg.backward(g.to(gpu))
        tol = 2e-5 if mode == "fp32" else 2e-4
        conv = blk.deconv if tr else blk.conv
        assert rel_err(yg, yr) < tol
        assert rel_err(xg.grad, xr.grad) < tol
        assert rel_err(conv.weight.grad, wr.grad) < tol
        assert rel_err(conv.bias.grad, br.grad) < tol
    finally:
        pkg.ops.set_precision(prev)


@pytest.mark.parametrize("cin,cout,H,W,N", [(64, 64, 32, 32, 2), (128, 128, 16, 12, 1), (64, 128, 9, 9, 1), (32, 64, 7, 5, 3),
                                            (64, 64, 128, 128, 1), (256, 256, 6, 6, 2)])
@pytest.mark.parametrize("bias", [True, False])
def test_wgrad_stride2_bf16x3(gpu, monkeypatch, cin, cout, H, W, N, bias):
    """k_wgrad_s2 (conv_wgrad_bf16.hip): the weight / bias gradient of a 3x3 stride-2 pad-1 convolution (SRGAN's
    discriminator, srgan.py:57-63) on the bf16 matrix cores with the halo columns de-interleaved into even / odd halves.
    Against float64, at the tolerance of the stride-1 bf16x3 weight gradient, with odd and ragged sizes; accumulation
    (beta = 1) on top of existing gradients; and equal to the exact-fp32 kernel it replaces within that tolerance."""
    import torch.nn.functional as F
    pkg = _pkg()
    ops, L, lib = pkg.ops, pkg._lib, pkg._lib.load()
    x = fill.randn((N, cin, H, W), 51)
    oh, ow = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    dy = fill.randn((N, cout, oh, ow), 52)
    wt = torch.zeros(cout, cin, 3, 3)
    ref_w = torch.nn.grad.conv2d_weight(x.double(), wt.shape, dy.double(), stride=2, padding=1)
    ref_b = dy.double().sum((0, 2, 3))
    CL = torch.channels_last
    xg, dyg = x.to(gpu).contiguous(memory_format=CL), dy.to(gpu).contiguous(memory_format=CL)
    d = ops._make_desc(xg.shape, wt, ops.ConvCfg(2, 1, False, 0, 0, 0.0, 0), "bwd")
    out = {}
    for tag, env in (("s2", "1"), ("fp32", "0")):
        monkeypatch.setenv("SRK_WGRAD_S2", env)
        ws = torch.empty(int(lib.srk_conv2d_backward_weight_workspace_bytes(ctypes.byref(d))) + 16, dtype=torch.uint8, device=gpu)
        dw = torch.full((cout, cin, 3, 3), 0.5, device=gpu)
        db = torch.full((cout,), -2.0, device=gpu) if bias else None
        rc = lib.srk_conv2d_backward_weight(ctypes.byref(d), L.ptr(xg), L.ptr(dyg), None, L.ptr(dw), L.ptr(db), 1.0, L.ptr(ws),
                                            ws.numel(), L.stream_ptr())
        assert rc == 0, lib.srk_last_error_string()
        torch.cuda.synchronize()
        out[tag] = (dw - 0.5, None if db is None else db + 2.0)
    assert rel_err(out["fp32"][0], ref_w) < 2e-5
    assert rel_err(out["s2"][0], ref_w) < 1e-4, rel_err(out["s2"][0], ref_w)
    assert not torch.equal(out["s2"][0], out["fp32"][0])      # (the bf16x3 kernel did run)
    if bias:
        assert rel_err(out["s2"][1], ref_b) < 2e-5


@pytest.mark.parametrize("shape", [(16, 32, 32), (3, 13, 21), (2, 17, 9)])
def test_conv_c64_leaves_batchnorm_column_sums(gpu, shape):
    """srk_epilogue.bn_partial (base_networks.py:46,117: conv -> bn): k_c64's forward leaves {sum y, sum y^2} per tile and
    channel of what it stored, srk_bn_finalize_partials turns them into the statistics srk_bn_stats_finalize computes
    from the activation itself -- through the C ABI against float64, then through ResnetBlock(norm='batch') with the
    protocol on and off (ragged tiles included)."""
    pkg = _pkg()
    ops, L, lib = pkg.ops, pkg._lib, pkg._lib.load()
    n, h, w = shape
    x = fill.randn((n, 64, h, w), 41 + h).to(gpu).contiguous(memory_format=torch.channels_last)
    wt = (fill.randn((64, 64, 3, 3), 42 + w) * (2.0 / 576) ** 0.5).to(gpu)
    b = (fill.randn((64,), 43) * 0.1).to(gpu)
    cfg = ops.ConvCfg(1, 1, False, 0, 0, 0.0, 0, L.ALGO_MFMA_BF16X6)
    d = ops._make_desc(x.shape, wt, cfg, "train_fwd")
    wp, bp = ops.pack_weight_fwd(wt, False, 0), ops.pack_bias_ps(b, 0)
    y = torch.empty_like(x)
    tiles = n * ((h + 7) // 8) * ((w + 7) // 8)
    part = torch.full((tiles, 128), float("nan"), dtype=torch.float64, device=gpu)
    ep = L.Epilogue(L.ptr(bp), None, None, 0.0, 0, 0, 0, None, None, L.ptr(part))
    assert lib.srk_conv2d_forward(ctypes.byref(d), L.ptr(x), L.ptr(wp), L.ptr(y), ctypes.byref(ep), L.stream_ptr()) == 0
    assert lib.srk_last_kernel_name().decode().startswith("k_c64<")
    assert lib.srk_last_conv_bn_partial_rows() == tiles
    res = L.ConvResult()                # ... and in the caller's srk_conv_result
    assert lib.srk_conv2d_forward_ex(ctypes.byref(d), L.ptr(x), L.ptr(wp), L.ptr(y), ctypes.byref(ep), ctypes.byref(res),
                                     L.stream_ptr()) == 0
    assert res.bn_partial_rows == tiles
    yd = y.double()
    s_ref = torch.cat([yd.sum((0, 2, 3)), (yd * yd).sum((0, 2, 3))])
    assert rel_err(part.sum(0), s_ref) < 1e-12
    rows = n * h * w
    stats = torch.empty(128, dtype=torch.float64, device=gpu)
    mean, rstd = torch.empty(64, device=gpu), torch.empty(64, device=gpu)
    rm, rv = torch.zeros(64, device=gpu), torch.ones(64, device=gpu)
    assert lib.srk_bn_finalize_partials(L.ptr(part), tiles, L.ptr(stats), rows, 64, L.ptr(mean), L.ptr(rstd), L.ptr(rm),
                                        L.ptr(rv), 0.1, 1e-5, None, L.stream_ptr()) == 0
    m_ref = yd.mean((0, 2, 3))
    v_ref = yd.var((0, 2, 3), unbiased=False)
    assert rel_err(mean, m_ref.float()) < 1e-6 and rel_err(rstd, (v_ref + 1e-5).rsqrt().float()) < 1e-6
    assert rel_err(rm, (0.1 * m_ref).float()) < 1e-6
    # a kernel that does not keep them says so
    ep2 = L.Epilogue(L.ptr(bp), None, None, 0.0, 1, 0, 0, None, None, L.ptr(part))   # with ReLU: not k_c64
    assert lib.srk_conv2d_forward(ctypes.byref(d), L.ptr(x), L.ptr(wp), L.ptr(y), ctypes.byref(ep2), L.stream_ptr()) == 0
    assert lib.srk_last_conv_bn_partial_rows() == 0
    # the block: same outputs, gradients and running statistics with the protocol on and off
    outs = {}
    for on in (True, False):
        torch.manual_seed(5)
        blk = pkg.base_networks.ResnetBlock(64, norm="batch").to(gpu).train()
        xin = x.clone().requires_grad_(True)
        old = ops.BN_PARTIAL
        ops.BN_PARTIAL = on
        # (the fused path must actually RUN in the `on` arm -- if the conv output lost its tag on the way to the BatchNorm,
        #  both arms would take the classic path and agree trivially: count the finalize calls)
        #  round 6: the consumer of the sums is srk_bn_finalize_apply_act, and WITHOUT them that path first launches
        #  srk_bn_stats_partials -- count those)
        real, calls = lib.srk_bn_stats_partials, []
        lib.srk_bn_stats_partials = lambda *a: (calls.append(1), real(*a))[1]
        try:
            out = blk(xin)
            out.square().mean().backward()
        finally:
            ops.BN_PARTIAL = old
            lib.srk_bn_stats_partials = real
        assert len(calls) == (0 if on else 2), (on, len(calls))    # conv1 -> bn and conv2 -> bn (one shared BatchNorm)
        outs[on] = (out.detach(), xin.grad, blk.bn.running_mean.clone(), blk.bn.running_var.clone(), blk.conv1.weight.grad)
    for a, bb in zip(outs[True], outs[False]):
        assert rel_err(a, bb) < 1e-5
