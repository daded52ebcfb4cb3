set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "model_handle" 2>&1 | tail -15
