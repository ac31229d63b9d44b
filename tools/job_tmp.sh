cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_train_gpu.py tests/test_generic_gpu.py -x -q 2>&1 | tail -4
timeout 600 python tools/time_train.py --objects 4 --rays 4096 2>&1 | grep "rays x" | cut -c1-130
