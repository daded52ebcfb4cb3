set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r5_gputests_tail.txt
cat gpurun_out/r5_gputests_tail.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
