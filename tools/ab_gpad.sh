#!/bin/bash
cd $(dirname $0)/..
for r in 1 2; do for p in 0,0 0,1 1,0 1,1; do
  echo "PERM,W16=$p $(SRK_BFW_PERM=${p%%,*} SRK_BFW_W16=${p##*,} python bench.py --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        r = json.loads(ln); print(r['value'], r['roofline']['layer_ms'])
")"
done; done
