set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/profile_round.sh r03 > gpurun_out/profile_r03.log 2>&1
tail -2 gpurun_out/profile_r03.log
mkdir -p profiles_tmp && cp gpurun_out/prof_r03/reduced/* profiles/ 2>/dev/null
python bench.py > gpurun_out/bench_r03.json 2> gpurun_out/bench_r03.err; tail -2 gpurun_out/bench_r03.err
python -c "
import json; d=json.load(open('gpurun_out/bench_r03.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic']); print(json.dumps(d['extra'], indent=0)[:3500]); print(d['cpu_baseline'])"
