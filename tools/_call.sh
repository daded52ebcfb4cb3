set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "motion_attention_sub_block or feed_forward or geglu or gelu" 2>&1 | tail -3
timeout -k 10 900 python -m pytest tests/test_full_width_gpu.py -x -q -m gpu -k "motion_module" 2>&1 | tail -3
