set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_train_gpu.py tests/test_dp_gpu.py tests/test_ops_gpu.py tests/test_interp.py -m gpu -q 2>&1 | tail -3
python tools/shard_step.py 16 50 2>&1 | tail -1; python tools/shard_step.py 128 20 2>&1 | tail -1; python tools/vdsr_graph_step.py 2>&1 | tail -1; python tools/srgan_graph_step.py 2>&1 | tail -1
