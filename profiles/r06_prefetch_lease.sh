#!/bin/bash
# backbone of the next batch prefetched on a side stream inside the step's graph (engine.GraphedStep(prefetch=...)): test + whole-step A/B
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r06/p; mkdir -p $O
timeout 600 python -m pytest tests/test_engine_gpu.py -q -x -s -k "prefetched or graphed" 2>&1 | grep -E "parity\] prefetched|passed|failed|Error|error|assert" | head -30 | tee $O/tests.txt
for mode in 1 0; do
  A3D_PREFETCH_BACKBONE=$mode timeout 600 python bench.py --skip-secondary --skip-cpu-baseline --steps 20 --warmup 5 > $O/bench_prefetch$mode.json 2> $O/bench_prefetch$mode.err
  python -c "
import json; d=json.load(open('$O/bench_prefetch$mode.json')); print('A3D_PREFETCH_BACKBONE=$mode', round(d['value'],1), round(d['ms_per_step'],3), d['config'].get('backbone_prefetch'), d['config'].get('graph_capture_error'))" || tail -5 $O/bench_prefetch$mode.err
done
