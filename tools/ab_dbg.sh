#!/bin/bash
# time of the c2 layers with parts of k_conv_bfw switched off (SRK_DBG bits: 1 no halo loads, 16 no split + LDS write,
# 4 no MFMA loop, 2 no epilogue) -- results are wrong by construction, only the layer times mean anything
cd $(dirname $0)/..
for d in "$@"; do
  echo "SRK_DBG=$d $(SRK_DBG=$d python bench.py --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        r = json.loads(ln); print(r['value'], r['roofline']['layer_ms'])
")"
done
