set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "layernorm_qkv or k0_v0t" 2>&1 | tail -5
timeout -k 10 900 python -m pytest tests/test_full_width_gpu.py tests/test_golden_gpu.py -x -q -m gpu -k "transformer or t2d or block or unet" 2>&1 | tail -3
echo "# A = round-4 tree, B = this tree" > gpurun_out/c33_ab.txt
bash tools/ab_rounds.sh r4 2 >> gpurun_out/c33_ab.txt 2>&1
cat gpurun_out/c33_ab.txt
