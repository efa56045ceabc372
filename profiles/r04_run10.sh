#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04i; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -s -k "conv3x3 or backbone" > $O/k.log 2>&1; grep -E "passed|failed" $O/k.log | tail -3; grep -E "^FAILED|Error|max err" $O/k.log | head -20
grep "parity. conv3x3\|parity. backbone" $O/k.log | head -30
timeout 600 python profiles/conv3x3_probe.py > $O/conv3x3_probe.json 2> $O/probe.err; tail -3 $O/probe.err; cat $O/conv3x3_probe.json
timeout 600 python bench.py --skip-secondary --skip-cpu-baseline > $O/bench_kp.json 2> $O/bench_kp.err
A3D_FUSED_CONV3X3=0 timeout 600 python bench.py --skip-secondary --skip-cpu-baseline > $O/bench_kp_no3.json 2> /dev/null
for f in bench_kp bench_kp_no3; do python - <<P
import json
try:
    d=json.load(open("$O/$f.json")); print("$f", round(d["value"],1), round(d["ms_per_step"],3), d.get("hot_path_only"))
except Exception as e: print("$f", "failed", e)
P
done
