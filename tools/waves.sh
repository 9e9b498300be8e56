#!/bin/bash
cd $(dirname $0)/..
for shape in vdsr espcn2 espcn3 edsrup; do
for nw in 4 2 1; do
  SRK_BF3_WAVES=$nw SRK_DBG=32 SHAPE=$shape python - <<'PY' 2>&1 | grep -E "waves=|\[srk\]" | sort -u | head -3
import os, sys, torch
sys.path.insert(0, ".")
import pytorch_super_resolution_model_collection_amd as pkg
ops = pkg.ops
SH = {"vdsr": (256, 64, 41, 41, 64, 3, 1, 1, 0), "espcn2": (64, 64, 252, 252, 32, 3, 0, 1, 0), "espcn3": (64, 32, 250, 250, 48, 3, 0, 0, 4), "edsrup": (128, 64, 64, 64, 256, 3, 1, 0, 2)}
shape = os.environ["SHAPE"]
N, cin, H, W, cout, k, pad, act, ps = SH[shape]
dev = torch.device("cuda:0")
x = torch.randn(N, cin, H, W, device=dev).contiguous(memory_format=torch.channels_last)
w = torch.randn(cout, cin, k, k, device=dev) * 0.05
b = torch.randn(cout, device=dev)
cfg = ops.ConvCfg(1, pad, False, 0, act, 0.0, ps, 0)
wp, bp = ops.pack_weight_fwd(w, False, ps), ops.pack_bias_ps(b, ps)
ref = None
with torch.no_grad():
    for _ in range(3): y = ops.conv2d_infer(x, w, b, None, cfg, None, (wp, bp))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.conv2d_infer(x, w, b, None, cfg, None, (wp, bp))
    e1.record(); torch.cuda.synchronize()
print("%s waves=%s  %.3f ms  checksum %.6e" % (shape, os.environ["SRK_BF3_WAVES"], e0.elapsed_time(e1) / 10, float(y.double().abs().sum())))
PY
done; done
