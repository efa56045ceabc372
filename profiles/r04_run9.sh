#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04h; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py tests/test_act3d_gpu.py -q -s -k "projection or attn_block or act3d or query_stream" > $O/k.log 2>&1; grep -E "passed|failed" $O/k.log | tail -3; grep -E "^FAILED|Error" $O/k.log | head
python bench.py --kernels-only > $O/kern_res.json 2>/dev/null
A3D_PROJ_RES=0 python bench.py --kernels-only > $O/kern_nores.json 2>/dev/null
python bench.py --skip-secondary --skip-cpu-baseline > $O/bench_kp.json 2> $O/bench_kp.err
A3D_PROJ_RES=0 python bench.py --skip-secondary --skip-cpu-baseline > $O/bench_kp_nores.json 2> /dev/null
for f in bench_kp bench_kp_nores; do python - <<P
import json
try:
    d=json.load(open("$O/$f.json")); print("$f", round(d["value"],1), round(d["ms_per_step"],3), d.get("hot_path_only"))
except Exception as e: print("$f", "failed", e)
P
done
for f in res nores; do python - <<P
import json
d=json.load(open("$O/kern_$f.json"))["kernels"]; print("$f", {k:round(v["ms"],4) for k,v in d.items() if isinstance(v,dict)})
P
done
