set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_full_width_gpu.py -x -q -m gpu -k "motion or cross_attention_sub or transformer_2d" 2>&1 | tail -8
