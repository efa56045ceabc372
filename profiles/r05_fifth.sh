#!/bin/bash
# Fifth GPU call of round 5: whole parity suite after the call-free sincos (determinism of the fused denoise step, joint test),
# forward-kernel occupancy A/B (2 vs 3 workgroups per CU), bench line.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05e; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -s > $O/pytest_all.log 2>&1; echo "all rc=$? $(grep -E 'passed|failed' $O/pytest_all.log | tail -1)"; grep -E "^FAILED|^ERROR" $O/pytest_all.log | head -20
grep "\[parity\]" $O/pytest_all.log > $O/parity_report.txt; wc -l $O/parity_report.txt
timeout 200 python profiles/cfg3_graph_probe.py 2>&1 | grep -E "n_steps" | head -4
for cfg in "libact3d_hip.so 512" "libact3d_hip_occ3.so 512" "libact3d_hip_occ3.so 768"; do set -- $cfg; A3D_LIB=$1 A3D_SQ_WGS=$2 timeout 300 python bench.py --kernels-only > $O/k.json 2> /dev/null; python - <<P
import json
try:
    k=json.load(open("$O/k.json"))["kernels"]; print("$1 WGS=$2", {n: round(k[n]["ms"]*1e3,1) for n in ("sq_fwd","sq_bwd") if n in k})
except Exception as e: print("failed", e)
P
done
timeout 600 python bench.py --skip-secondary --skip-cpu-baseline > $O/bench_kp.json 2> $O/bench_kp.err
python - <<P
import json
try:
    d=json.load(open("$O/bench_kp.json")); print("bench_kp", round(d["value"],1), round(d["ms_per_step"],3), d.get("hot_path_only"))
except Exception as e: print("bench_kp failed", e)
P
