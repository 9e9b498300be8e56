"""ORACLE helper — TEST INFRASTRUCTURE ONLY.  The op-level known-answer cases shared by the
fixture generator (tests/golden/make_golden.py) and the tests.  Inputs are regenerated from seeds
with oracle.fill (platform-independent numpy MT19937), so the fixture file only stores expected
outputs / gradients."""
from . import fill

CONV_KATS = [
    # (tag, Cin, Cout, k, stride, pad, transposed, out_pad, H, W, N, act)
    ("c3_64_64", 64, 64, 3, 1, 1, 0, 0, 9, 11, 2, "relu"),          # VDSR / EDSR / SRGAN body
    ("c5_3_64", 3, 64, 5, 1, 0, 0, 0, 16, 14, 2, "relu"),           # ESPCN layer 1
    ("c3_64_32", 64, 32, 3, 1, 0, 0, 0, 12, 12, 2, "relu"),         # ESPCN layer 2
    ("c3_32_48", 32, 48, 3, 1, 0, 0, 0, 12, 10, 2, None),           # ESPCN layer 3 (pre-shuffle)
    ("c9_3_64", 3, 64, 9, 1, 0, 0, 0, 16, 16, 1, "relu"),           # SRCNN layer 1
    ("c5_32_3", 32, 3, 5, 1, 0, 0, 0, 12, 12, 2, None),             # SRCNN layer 3
    ("c1_56_12", 56, 12, 1, 1, 0, 0, 0, 9, 9, 2, "lrelu"),          # FSRCNN shrink
    ("c3_12_12", 12, 12, 3, 1, 1, 0, 0, 9, 9, 2, None),             # FSRCNN map
    ("c1_12_56", 12, 56, 1, 1, 0, 0, 0, 9, 9, 2, "lrelu"),          # FSRCNN expand
    ("c3_64_3", 64, 3, 3, 1, 1, 0, 0, 9, 9, 2, None),               # VDSR / EDSR tail
    ("c3_3_64", 3, 64, 3, 1, 1, 0, 0, 9, 9, 2, "lrelu"),            # VDSR / EDSR head
    ("c9_64_3", 64, 3, 9, 1, 4, 0, 0, 12, 12, 1, None),             # SRGAN-G tail
    ("c9_3_64p", 3, 64, 9, 1, 4, 0, 0, 12, 12, 1, None),            # SRGAN-G head
    ("c3s2_64_64", 64, 64, 3, 2, 1, 0, 0, 12, 12, 2, "lrelu"),      # SRGAN-D stride 2
    ("c3s2_64_128", 64, 128, 3, 2, 1, 0, 0, 9, 9, 1, None),         # SRGAN-D stride 2, Cout > 64
    ("c3_80_96", 80, 96, 3, 1, 1, 0, 0, 6, 7, 1, None),             # Cin, Cout beyond one 64-chunk
    ("d9s4_56_3", 56, 3, 9, 4, 3, 1, 1, 7, 7, 2, None),             # FSRCNN deconv (fsrcnn.py:33)
    ("d4s2_64_64", 64, 64, 4, 2, 1, 1, 0, 6, 6, 2, "lrelu"),        # LapSRN feature deconv
    ("d4s2_3_3", 3, 3, 4, 2, 1, 1, 0, 7, 7, 2, None),               # LapSRN image deconv
    ("c4s2_default", 8, 16, 4, 2, 1, 0, 0, 10, 10, 2, "relu"),      # ConvBlock ctor defaults (base_networks.py:40)
]


def conv_case_inputs(i):
    """(x NCHW, weight torch-layout, bias, upstream gradient g) of CONV_KATS[i]."""
    tag, cin, cout, k, s, p, tr, op, H, W, N, act = CONV_KATS[i]
    wshape = (cin, cout, k, k) if tr else (cout, cin, k, k)
    w = fill.randn(wshape, 1000 + i, (2.0 / (cin * k * k)) ** 0.5)
    b = fill.randn((cout,), 2000 + i, 0.1)
    x = fill.randn((N, cin, H, W), 3000 + i)
    if tr:
        oh, ow = (H - 1) * s - 2 * p + k + op, (W - 1) * s - 2 * p + k + op
    else:
        oh, ow = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    g = fill.randn((N, cout, oh, ow), 4000 + i)
    return x, w, b, g
