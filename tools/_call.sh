set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "layernorm_qkv" 2>&1 | tail -2
I2V_LNQKV_WG2=0 timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "layernorm_qkv" 2>&1 | tail -2
echo "--- two workgroups per CU (default)"
timeout -k 10 300 python tools/ln_qkv_probe.py 2>&1 | grep -v amdgpu.ids | tail -2
echo "--- one workgroup per CU"
I2V_LNQKV_WG2=0 timeout -k 10 300 python tools/ln_qkv_probe.py 2>&1 | grep -v amdgpu.ids | tail -2
