#!/bin/bash
# Board power and clocks while a command loops (run on the GPU box):  tools/power_probe.sh <seconds> <command ...>
# Samples rocm-smi every ~0.2 s while the command runs; prints min / median / max of power and sclk.
SECS=$1; shift
"$@" > /tmp/pp_cmd.log 2>&1 &
PID=$!
sleep 2.5
: > /tmp/pp.log
END=$(( $(date +%s) + SECS ))
while [ $(date +%s) -lt $END ] && kill -0 $PID 2>/dev/null; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Average Graphics Package Power|Current Socket Graphics Package Power|sclk|mclk|fclk|Temperature \(Sensor (junction|memory)" >> /tmp/pp.log
  echo "--" >> /tmp/pp.log
done
wait $PID
python3 - <<'PY'
import re, statistics
txt = open("/tmp/pp.log").read()
def vals(pat):
    return [float(x) for x in re.findall(pat, txt)]
for name, pat in (("power W", r"Power \(W\): ([0-9.]+)"), ("sclk MHz", r"sclk clock level: \d+: \((\d+)Mhz\)"), ("mclk MHz", r"mclk clock level: \d+: \((\d+)Mhz\)"),
                  ("fclk MHz", r"fclk clock level: \d+: \((\d+)Mhz\)"), ("junction C", r"junction\) \(C\): ([0-9.]+)"), ("memory C", r"memory\) \(C\): ([0-9.]+)")):
    v = vals(pat)
    if v: print("  %-10s n %3d  min %7.1f  median %7.1f  max %7.1f" % (name, len(v), min(v), statistics.median(v), max(v)))
    else: print("  %-10s (no samples)" % name)
PY
tail -2 /tmp/pp_cmd.log
