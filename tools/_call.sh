set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/c25_configs.jsonl
timeout -k 10 300 python bench.py --no-cpu-baseline --frames 8 --size 256 2>/dev/null >> gpurun_out/c25_configs.jsonl
timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null >> gpurun_out/c25_configs.jsonl
timeout -k 10 300 python bench.py --no-cpu-baseline --ip 2>/dev/null >> gpurun_out/c25_configs.jsonl
timeout -k 10 600 python bench.py --no-cpu-baseline --frames 32 --size 768 --steps 10 --windows 3 2>/dev/null >> gpurun_out/c25_configs.jsonl
I2V_MOTION_FUSED=0 timeout -k 10 600 python bench.py --no-cpu-baseline --frames 32 --size 768 --steps 10 --windows 3 2>/dev/null >> gpurun_out/c25_configs.jsonl
python -c "
import json
for l in open('gpurun_out/c25_configs.jsonl'):
    d=json.loads(l); print(d['config']['workload'][:70], round(d['value'],3), round(d['ms_per_step'],2))
"
timeout -k 10 1000 python bench.py --parity-only --frames 32 --size 768 2> gpurun_out/c25_p5.err | tee gpurun_out/c25_parity_config5.json | cut -c1-400
