#!/bin/bash
# deep-layer 1x1 GEMM: kernel + backbone tests, whole-step A/B (default per-shape set vs A3D_CONV1X1_DEEP=0 vs =2)
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r06/d3; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "conv1x1 or backbone" 2>&1 | tail -4 | tee $O/tests.txt
for mode in 1 0 2; do
  A3D_CONV1X1_DEEP=$mode timeout 600 python bench.py --skip-secondary --skip-cpu-baseline --steps 20 --warmup 5 > $O/bench_mode$mode.json 2> $O/bench_mode$mode.err
  python -c "
import json; d=json.load(open('$O/bench_mode$mode.json')); print('A3D_CONV1X1_DEEP=$mode', round(d['value'],1), round(d['ms_per_step'],3))"
done
