set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "fused or motion or feed_forward or layernorm_qkv" 2>&1 | tail -2
echo "# A = round-4 tree, B = this tree" > gpurun_out/c22_ab.txt
bash tools/ab_rounds.sh r4 2 >> gpurun_out/c22_ab.txt 2>&1
echo "# A = I2V_FF_TAIL=0, B = default" >> gpurun_out/c22_ab.txt
bash tools/ab_env.sh I2V_FF_TAIL=0 2 >> gpurun_out/c22_ab.txt 2>&1
cat gpurun_out/c22_ab.txt
