set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_kernels_gpu.py tests/test_training_gpu.py tests/test_modules_gpu.py -x -q -m gpu -k "norm or training or pipeline or train or adapter or block" 2>&1 | tail -3
python bench.py --no-cpu-baseline --shapes gpurun_out/c27_step_shapes.txt 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],3), round(d['ms_per_step'],3), {k: v['ms'] for k, v in d['kernel_classes'].items()})"
head -60 gpurun_out/c27_step_shapes.txt
