#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04d; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -q -s -k "query_stream" > $O/qs.log 2>&1; grep -E "passed|failed|Error|error" $O/qs.log | tail -5; grep -E "parity.*fused vs|Assertion" $O/qs.log | tail -12
python -m pytest tests/test_act3d_gpu.py tests/test_engine_gpu.py tests/test_joint_gpu.py -q -s > $O/model.log 2>&1; grep -E "passed|failed" $O/model.log | tail -3; grep -E "^FAILED" $O/model.log | head
python bench.py --skip-secondary --skip-cpu-baseline > $O/bench_kp.json 2> $O/bench_kp.err; echo "bench rc=$?"
A3D_OVERLAP_STREAMS=1 python bench.py --skip-secondary --skip-cpu-baseline > $O/bench_kp_overlap.json 2> $O/bench_kp_overlap.err
A3D_QS_FUSED=0 python bench.py --skip-secondary --skip-cpu-baseline > $O/bench_kp_noqs.json 2> /dev/null
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/$O/trace" -o kp -- python "$GRAFT_REPO_ROOT/bench.py" --skip-secondary --skip-cpu-baseline --no-graph --steps 10 --warmup 4 > "$GRAFT_REPO_ROOT/$O/trace.log" 2>&1 )
DB=$(find $O/trace -name '*.db' | head -1); python profiles/summarize.py "$DB" > $O/kernel_trace_B64.txt 2>&1; rm -rf $O/trace
for f in bench_kp bench_kp_overlap bench_kp_noqs; do python - <<P
import json
try:
    d=json.load(open("$O/$f.json")); print("$f", round(d["value"],1), round(d["ms_per_step"],3), d.get("hot_path_only"))
except Exception as e: print("$f", "failed", e)
P
done
head -30 $O/kernel_trace_B64.txt
