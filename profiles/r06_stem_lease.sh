#!/bin/bash
# stem convolution (csrc/stem.hip): kernel + backbone tests, whole-step A/B
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r06/e; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "stem or backbone" 2>&1 | tail -4 | tee $O/tests.txt
for mode in 1 0; do
  A3D_FUSED_STEM=$mode timeout 600 python bench.py --skip-secondary --skip-cpu-baseline --steps 20 --warmup 5 > $O/bench_stem$mode.json 2> $O/bench_stem$mode.err
  python -c "
import json; d=json.load(open('$O/bench_stem$mode.json')); print('A3D_FUSED_STEM=$mode', round(d['value'],1), round(d['ms_per_step'],3), d['roofline'].get('kernel'), d['roofline'].get('frac'))"
done
