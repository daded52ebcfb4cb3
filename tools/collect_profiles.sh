#!/bin/bash
# One GPU-box call that produces every measurement artefact of the round (copy gpurun_out/<tag>_* to profiles/):
#   <tag>_bench_default.json      python bench.py (configs[1], with the CPU-oracle baseline + parity)
#   <tag>_bench_configs.jsonl     configs[0] (8 f x 256^2), configs[2] (--ip), configs[4] (32 f x 768^2), configs[3] on one GPU
#                                 (--pairs 8, batch 1/2/4)
#   <tag>_step_shapes.txt         where one step's time sits per (kernel, problem shape) (bench.py --shapes)
#   <tag>_kernel_stats.txt        rocprofv3 --kernel-trace --stats of `bench.py --steps 10 --warmup 2`
#   <tag>_pmc_summary.txt         per-kernel PMC table (tools/pmc_step.sh)
#   <tag>_traffic.json            bytes below L2 per launch by kernel class (tools/pmc_traffic.sh)
#   <tag>_ceilings.json           measured ceilings of THIS chip: stream copy / read, MFMA-saturating loops (tools/ceilings.hip,
#                                 built beforehand in the container: hipcc --offload-arch=gfx950 -O3 tools/ceilings.hip -o
#                                 tools/build/ceilings); bench.py reads profiles/<tag>_ceilings.json for frac_of_measured
set -e
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out
tag=${1:-r2}
export TMPDIR=/tmp
if [ -x tools/build/ceilings ]; then tools/build/ceilings > gpurun_out/${tag}_ceilings.json 2> gpurun_out/${tag}_ceilings.err; fi
: > gpurun_out/${tag}_bench_configs.jsonl
python bench.py --frames 8 --size 256 --no-cpu-baseline 2>/dev/null >> gpurun_out/${tag}_bench_configs.jsonl
python bench.py --ip --no-cpu-baseline 2>/dev/null >> gpurun_out/${tag}_bench_configs.jsonl
python bench.py --frames 32 --size 768 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null >> gpurun_out/${tag}_bench_configs.jsonl
for b in 1 2 4; do
  python bench.py --pairs 8 --batch $b --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null >> gpurun_out/${tag}_bench_configs.jsonl
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_prof.log 2>&1
python tools/summarize_prof.py gpurun_out/${tag}_prof gpurun_out/${tag}_kernel_stats.txt "python bench.py --steps 10 --warmup 2 --no-cpu-baseline on one MI355X" > /dev/null
rm -rf gpurun_out/${tag}_prof
bash tools/pmc_step.sh ${tag}_pmc > /dev/null 2>&1
bash tools/pmc_traffic.sh > gpurun_out/${tag}_traffic.log 2>&1
cp gpurun_out/traffic.json gpurun_out/${tag}_traffic.json
# the headline line LAST, with this box's traffic and ceilings where bench.py looks for them (the copies under profiles/ on
# the box are scratch; the ones to commit are the gpurun_out/ files)
cp gpurun_out/${tag}_traffic.json profiles/${tag}_traffic.json
cp gpurun_out/${tag}_kernel_stats.txt profiles/${tag}_kernel_stats.txt      # bench.py's roofline.dominant_kernel reads it (same source stamp)
if [ -s gpurun_out/${tag}_ceilings.json ]; then cp gpurun_out/${tag}_ceilings.json profiles/${tag}_ceilings.json; fi
python bench.py --shapes gpurun_out/${tag}_step_shapes.txt > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err
python -c "
import json
d = json.load(open('gpurun_out/${tag}_bench_default.json'))
print('default', round(d['value'], 3), 'steps/s', round(d['ms_per_step'], 2), 'ms', d['roofline']['frac'], d['cpu_baseline']['value'], d['parity'])
for l in open('gpurun_out/${tag}_bench_configs.jsonl'):
    d = json.loads(l); print(round(d['value'], 3), round(d['ms_per_step'], 2), d['config']['workload'][-90:])
"
