#!/bin/bash
# BatchNorm-apply grid size (A3D_BN_GRID) under the prefetched-backbone step and in the sequential step
#   usage: bash profiles/r06_bn_grid_lease.sh "2048 1" "1024 1" "2048 0"      (grid, prefetch on / off)
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r06/p; mkdir -p $O
for cfg in "$@"; do set -- $cfg
  A3D_BN_GRID=$1 A3D_PREFETCH_BACKBONE=$2 timeout 300 python bench.py --skip-secondary --skip-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('A3D_BN_GRID=$1 prefetch=$2', round(d['value'],1), round(d['ms_per_step'],3))" | tee -a $O/bn_grid.txt
done
