#!/bin/bash
# First GPU call of round 5 (merged preparation branch): the whole parity suite once (no -x: every failure is attributable in one
# call), the keypose bench line with its A/B switches, the eager kernel traces of the keypose and the diffusion training step (compare
# kernel by kernel with profiles/r04_kernel_trace_*.txt), then the sq_bwd phase probe.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; echo "all rc=$? $(grep -E 'passed|failed' $O/pytest_all.log | tail -1)"; grep -E "^FAILED|^ERROR" $O/pytest_all.log | head -20
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
head -45 $O/kernel_trace_B64.txt | cut -c1-140
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/$O/trace2" -o dt -- python "$GRAFT_REPO_ROOT/bench_denoise.py" --mode train --no-graph --reps 5 > "$GRAFT_REPO_ROOT/$O/trace_dt.log" 2>&1 )
DB=$(find $O/trace2 -name '*.db' | head -1); python profiles/summarize.py "$DB" > $O/kernel_trace_diffusion_train.txt 2>&1; rm -rf $O/trace2
head -30 $O/kernel_trace_diffusion_train.txt | cut -c1-140
timeout 200 python bench_denoise.py --mode train > $O/denoise_train.json 2> $O/denoise_train.err; cat $O/denoise_train.json | cut -c1-400
timeout 200 python bench_denoise.py --mode sample > $O/denoise_sample.json 2> $O/denoise_sample.err; cat $O/denoise_sample.json | cut -c1-400
timeout 120 python profiles/sq_bwd_phases.py > $O/sq_bwd_phases.json 2> $O/sq_phases.err; cat $O/sq_bwd_phases.json
