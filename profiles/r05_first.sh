#!/bin/bash
# First GPU call of round 5 on branch r5-prep: every prepared change is unverified; run the parity tests grouped by the change they
# cover so that a failure is attributable in ONE call, then the bench line and the eager kernel trace (compare kernel by kernel with
# profiles/r04_kernel_trace_B64.txt of main), then the sq_bwd phase probe.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05a; mkdir -p $O
run() { name=$1; shift; timeout 600 python -m pytest "$@" -q -x > $O/t_$name.log 2>&1; echo "$name rc=$? $(grep -E 'passed|failed|error' $O/t_$name.log | tail -1)"; grep -E "^FAILED|^ERROR|Error" $O/t_$name.log | head -5; }
run bn       tests/test_kernels_gpu.py -k "batchnorm or backbone or fpn"
run convs    tests/test_kernels_gpu.py -k "conv1x1 or conv3x3"
run sq       tests/test_kernels_gpu.py -k "single_query or query_stream or sq_"
run rope     tests/test_kernels_gpu.py -k "rope or projection or attn_block or operand"
run ln       tests/test_kernels_gpu.py -k "layernorm or add_ln or mlp or linear"
run sink     tests/test_act3d_gpu.py -k "sink or golden"
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; echo "all rc=$? $(grep -E 'passed|failed' $O/pytest_all.log | tail -1)"; grep -E "^FAILED|^ERROR" $O/pytest_all.log | head
timeout 600 python bench.py --skip-secondary --skip-cpu-baseline > $O/bench_kp.json 2> $O/bench_kp.err
A3D_CTX_SINK=0 A3D_FOLD_DS_BN=0 timeout 600 python bench.py --skip-secondary --skip-cpu-baseline > $O/bench_kp_nosink_nofold.json 2> /dev/null
for f in bench_kp bench_kp_nosink_nofold; do python - <<P
import json
try:
    d=json.load(open("$O/$f.json")); print("$f", round(d["value"],1), round(d["ms_per_step"],3), d.get("hot_path_only"))
except Exception as e: print("$f", "failed", e)
P
done
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/$O/trace" -o kp -- python "$GRAFT_REPO_ROOT/bench.py" --skip-secondary --skip-cpu-baseline --no-graph --steps 10 --warmup 4 > "$GRAFT_REPO_ROOT/$O/trace.log" 2>&1 )
DB=$(find $O/trace -name '*.db' | head -1); python profiles/summarize.py "$DB" > $O/kernel_trace_B64.txt 2>&1; rm -rf $O/trace
head -40 $O/kernel_trace_B64.txt | cut -c1-140
timeout 120 python profiles/sq_bwd_phases.py > $O/sq_bwd_phases.json 2> $O/sq_phases.err; cat $O/sq_bwd_phases.json
