#!/bin/bash
# A/B an environment switch on the c2 headline:  tools/ab_env_c2.sh VAR valueA valueB ...
cd $(dirname $0)/..
VAR=$1; shift
for round in 1 2 3; do for v in "$@"; do
  echo "$VAR=$v $(env $VAR=$v python bench.py --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        r = json.loads(ln); print(r['value'], r['roofline']['layer_ms'])
")"
done; done
