set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/ln_qkv_probe.py 2>&1 | grep -v amdgpu.ids | tail -1
echo "--- all tiles store to the first tile's rows"
I2V_LIB_PATH=$PWD/.ab_libs/lq_smallout.so timeout -k 10 300 python tools/ln_qkv_probe.py 2>&1 | grep -v amdgpu.ids | tail -1
