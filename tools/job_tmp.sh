cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
DINER_TRAIN_BATCH=0 timeout 1200 python -m pytest tests/test_train_gpu.py -x -q 2>&1 | tail -15
