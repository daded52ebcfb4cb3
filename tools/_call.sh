set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_golden_gpu.py -x -q -m gpu -k "attention or attn" 2>&1 | tail -3
echo "# A = round-4 tree, B = this tree" > gpurun_out/c28_ab.txt
bash tools/ab_rounds.sh r4 2 >> gpurun_out/c28_ab.txt 2>&1
cat gpurun_out/c28_ab.txt
