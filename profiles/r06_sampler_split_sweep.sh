#!/bin/bash
# persistent sampler: key splits per queue item (A3D_DN_PERSIST_SPLIT) at cfg-3 and at the script horizon
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r06/f; mkdir -p $O
for sp in 8 4 6 12 16; do
  for shape in "64 16" "24 50"; do set -- $shape
    A3D_DN_PERSIST_SPLIT=$sp timeout 300 python bench_denoise.py --mode sample --batch $1 --horizon $2 2>/dev/null | tail -1 > $O/s${sp}_B$1_L$2.json
    python -c "
import json; d=json.load(open('$O/s${sp}_B$1_L$2.json')); print('split $sp B $1 L $2', round(d['value'],1), d['unit'], round(d['ms_per_denoise_step'],4), 'ms/step', d['config'].get('graph_vs_eager_max_abs_diff'))"
  done
done
