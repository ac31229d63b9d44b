cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_train_gpu.py -x -q 2>&1 | tail -4
for rep in 1 2; do for v in 1 0; do
  echo "== DINER_TRAIN_VIEW_SHARED=$v"
  DINER_TRAIN_VIEW_SHARED=$v timeout 600 python tools/time_train.py --objects 4 --rays 4096 2>&1 | grep "rays x" | cut -c1-130
done; done
