#!/bin/bash
# Round-5 evidence run: the whole parity suite with its [parity] lines, the default bench line (secondary entries + CPU baseline),
# eager kernel traces of the keypose step, the diffusion training step and the samplers.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05z; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -s > $O/pytest_all.log 2>&1; echo "all rc=$? $(grep -E 'passed|failed' $O/pytest_all.log | tail -1)"; grep -E "^FAILED|^ERROR|fault" $O/pytest_all.log | head -20
grep "\[parity\]" $O/pytest_all.log > $O/parity_report.txt; wc -l $O/parity_report.txt
timeout 900 python bench.py > $O/bench_B64.json 2> $O/bench_B64.err; echo "bench rc=$?"
python - <<P
import json
try:
    d=json.load(open("$O/bench_B64.json")); print("bench", round(d["value"],1), round(d["ms_per_step"],3), d.get("hot_path_only"), "roofline", d["roofline"].get("kernel"), round(d["roofline"].get("frac",0),4))
    for s in d.get("secondary", []): print("  ", s.get("name"), s.get("value"), s.get("unit"), s.get("ms_per_step") or s.get("ms_per_denoise_step"), s.get("error"), (s.get("config") or {}).get("sampler"))
    print("  cpu", d.get("cpu_baseline"))
    print("  kernels", {k: round(v["ms"]*1e3,1) for k,v in d.get("kernels",{}).items()})
except Exception as e: print("bench parse failed", e, open("$O/bench_B64.err").read()[-500:])
P
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/$O/trace" -o kp -- python "$GRAFT_REPO_ROOT/bench.py" --skip-secondary --skip-cpu-baseline --no-graph --steps 10 --warmup 4 > "$GRAFT_REPO_ROOT/$O/trace.log" 2>&1 )
DB=$(find $O/trace -name '*.db' | head -1); python profiles/summarize.py "$DB" > $O/kernel_trace_B64.txt 2>&1; rm -rf $O/trace
head -30 $O/kernel_trace_B64.txt | cut -c1-130
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/$O/trace2" -o dt -- python "$GRAFT_REPO_ROOT/bench_denoise.py" --mode train --no-graph --reps 5 > "$GRAFT_REPO_ROOT/$O/trace_dt.log" 2>&1 )
DB=$(find $O/trace2 -name '*.db' | head -1); python profiles/summarize.py "$DB" > $O/kernel_trace_diffusion_train.txt 2>&1; rm -rf $O/trace2
head -12 $O/kernel_trace_diffusion_train.txt | cut -c1-130
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/trace3" -o ds -- python "$GRAFT_REPO_ROOT/bench_denoise.py" --mode sample --reps 3 > "$GRAFT_REPO_ROOT/$O/trace_ds.log" 2>&1 )
find $O/trace3 -name '*kernel_stats.csv' | head -1 | xargs -r head -12 > $O/kernel_stats_sampling.txt; rm -rf $O/trace3; cat $O/kernel_stats_sampling.txt | cut -c1-160
