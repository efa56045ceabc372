#!/bin/bash
# prefetched backbone in the trajectory-diffusion training step: graph-replay test (both modes), training-step A/B at the script shape
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r06/p; mkdir -p $O
timeout 900 python -m pytest tests/test_joint_gpu.py -q -x -s -k "graphed_trajectory" 2>&1 | grep -E "passed|failed|Error|error|assert" | head -10 | tee $O/tests4.txt
for mode in 1 0; do
  A3D_PREFETCH_BACKBONE=$mode timeout 600 python bench_denoise.py --mode train --batch 22 --horizon 50 2>/dev/null | tail -1 > $O/train_prefetch$mode.json
  python -c "
import json; d=json.load(open('$O/train_prefetch$mode.json')); print('A3D_PREFETCH_BACKBONE=$mode', round(d['value'],1), d['unit'], round(d['ms_per_step'],3), 'ms', d['config'].get('backbone_prefetch'), d['config'].get('graph_capture_error'))"
done
