#!/bin/bash
# FPN forward on own kernels (lateral 1x1 + bias + top-down add in one launch; 3x3 output convolutions on the stream kernel): kernel /
# FPN / model tests, whole-step A/B
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r06/h; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_act3d_gpu.py -q -x -s -k "fpn or lateral or golden or bf16 or graphed" 2>&1 | grep -E "parity\] (fpn lateral|fused)|passed|failed|Error|error" | head -40 | tee $O/tests.txt
for mode in "1 1" "0 1" "1 0" "0 0"; do set -- $mode
  A3D_FPN_LATERAL=$1 A3D_FPN_OUT3X3=$2 timeout 600 python bench.py --skip-secondary --skip-cpu-baseline --steps 20 --warmup 5 > $O/bench_lat$1_out$2.json 2> $O/bench_lat$1_out$2.err
  python -c "
import json; d=json.load(open('$O/bench_lat$1_out$2.json')); print('A3D_FPN_LATERAL=$1 A3D_FPN_OUT3X3=$2', round(d['value'],1), round(d['ms_per_step'],3))" || tail -5 $O/bench_lat$1_out$2.err
done
