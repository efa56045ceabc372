#!/bin/bash
# diffusion head: bf16 FPN (round 6) vs the fp32 FPN -- consistency test, training-step A/B; persistent sampler key-split sweep
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r06/g; mkdir -p $O
timeout 600 python -m pytest tests/test_diffusion_gpu.py -q -x -s -k "bf16_fpn" 2>&1 | grep -E "parity|passed|failed|Error" | head -20 | tee $O/tests.txt
for fp32 in 0 1; do
  A3D_DIFFUSION_FPN_FP32=$fp32 timeout 600 python bench_denoise.py --mode train --batch 22 --horizon 50 2>/dev/null | tail -1 > $O/train_fp32fpn$fp32.json
  python -c "
import json; d=json.load(open('$O/train_fp32fpn$fp32.json')); print('A3D_DIFFUSION_FPN_FP32=$fp32', round(d['value'],1), d['unit'], round(d['ms_per_step'],3), 'ms')"
done
for sp in 8 4 6; do
  A3D_DN_PERSIST_SPLIT=$sp timeout 300 python bench_denoise.py --mode sample --batch 64 --horizon 16 2>/dev/null | tail -1 > $O/s${sp}_cfg3.json
  python -c "
import json; d=json.load(open('$O/s${sp}_cfg3.json')); print('split $sp cfg3', round(d['value'],1), round(d['ms_per_denoise_step'],4), d['config'].get('graph_vs_eager_max_abs_diff'))"
done
A3D_DN_PERSIST_SPLIT=4 timeout 300 python bench_denoise.py --mode sample --batch 24 --horizon 50 2>/dev/null | tail -1 > $O/s4_L50.json
python -c "
import json; d=json.load(open('$O/s4_L50.json')); print('split 4 L50', round(d['value'],1), round(d['ms_per_denoise_step'],4))"
