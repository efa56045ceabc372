#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04e; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -q -s -k "knn or query_stream" > $O/k.log 2>&1; grep -E "passed|failed" $O/k.log | tail -3; grep -E "^FAILED|Error" $O/k.log | head
python -m pytest tests/test_diffusion_gpu.py tests/test_act3d_gpu.py -q -s > $O/model.log 2>&1; grep -E "passed|failed" $O/model.log | tail -3; grep -E "^FAILED" $O/model.log | head
python bench.py --skip-secondary --skip-cpu-baseline > $O/bench_kp.json 2> $O/bench_kp.err; echo "bench rc=$?"
A3D_QS_FUSED=0 python bench.py --skip-secondary --skip-cpu-baseline > $O/bench_kp_noqs.json 2> /dev/null
python bench_denoise.py --mode sample > $O/denoise.json 2> $O/denoise.err; tail -c 600 $O/denoise.json
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/$O/trace" -o kp -- python "$GRAFT_REPO_ROOT/bench.py" --skip-secondary --skip-cpu-baseline --no-graph --steps 10 --warmup 4 > "$GRAFT_REPO_ROOT/$O/trace.log" 2>&1 )
DB=$(find $O/trace -name '*.db' | head -1); python profiles/summarize.py "$DB" > $O/kernel_trace_B64.txt 2>&1; rm -rf $O/trace
for f in bench_kp bench_kp_noqs; do python - <<P
import json
try:
    d=json.load(open("$O/$f.json")); print("$f", round(d["value"],1), round(d["ms_per_step"],3), d.get("hot_path_only")); print({k:(round(v["ms"],4), round(v["frac"],4)) for k,v in d.get("kernels",{}).items()}, d.get("kernels_error"))
except Exception as e: print("$f", "failed", e)
P
done
grep -E "qs_|knn|sq_|dispatches" $O/kernel_trace_B64.txt | head -20
