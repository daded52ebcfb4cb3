set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_full_width_gpu.py tests/test_golden_gpu.py tests/test_modules_gpu.py -x -q -m gpu 2>&1 | tail -4
echo "# A = I2V_QKV_FUSED=0, B = default" > gpurun_out/c21_ab.txt
bash tools/ab_env.sh I2V_QKV_FUSED=0 2 >> gpurun_out/c21_ab.txt 2>&1
cat gpurun_out/c21_ab.txt
