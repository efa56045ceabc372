#!/bin/bash
# Second GPU call of round 5: the whole parity suite with its [parity] lines (-s) after the gate-node gradient sink, the
# cfg-3 graph test three more times (a capture-vs-replay difference was seen once in the first call), the bench line.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05b; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -s > $O/pytest_all.log 2>&1; echo "all rc=$? $(grep -E 'passed|failed' $O/pytest_all.log | tail -1)"; grep -E "^FAILED|^ERROR" $O/pytest_all.log | head -20
grep "\[parity\]" $O/pytest_all.log > $O/parity_report.txt; wc -l $O/parity_report.txt
for i in 1 2 3; do timeout 300 python -m pytest tests/test_diffusion_gpu.py -q -k "cfg3_full_shape" 2>&1 | tail -3 | grep -E "passed|failed|max abs diff"; done
timeout 600 python bench.py --skip-secondary --skip-cpu-baseline > $O/bench_kp.json 2> $O/bench_kp.err
python - <<P
import json
try:
    d=json.load(open("$O/bench_kp.json")); print("bench_kp", round(d["value"],1), round(d["ms_per_step"],3), d.get("hot_path_only"))
except Exception as e: print("bench_kp failed", e)
P
