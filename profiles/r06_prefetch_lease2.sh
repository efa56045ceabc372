#!/bin/bash
# prefetched backbone: 2-rank DP variant of the test; capture-stream priority A/B
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r06/p; mkdir -p $O
timeout 900 python -m pytest tests/test_engine_gpu.py -q -x -s -k "two_ranks" 2>&1 | grep -E "parity\] DP|passed|failed|Error|error|assert" | head -30 | tee $O/tests2.txt
for prio in 1 0; do
  A3D_PREFETCH_BACKBONE=1 A3D_PREFETCH_HIPRIO=$prio timeout 600 python bench.py --skip-secondary --skip-cpu-baseline --steps 20 --warmup 5 > $O/bench_prefetch1_prio$prio.json 2> $O/bench_prefetch1_prio$prio.err
  python -c "
import json; d=json.load(open('$O/bench_prefetch1_prio$prio.json')); print('A3D_PREFETCH_HIPRIO=$prio', round(d['value'],1), round(d['ms_per_step'],3), d['config'].get('backbone_prefetch'), d['config'].get('graph_capture_error'))" || tail -5 $O/bench_prefetch1_prio$prio.err
done
