set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 > gpurun_out/c35_tests.txt
cat gpurun_out/c35_tests.txt
python __graft_entry__.py smoke 2>&1 | tail -3
