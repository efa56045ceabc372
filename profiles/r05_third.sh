#!/bin/bash
# Third GPU call of round 5: the capture-vs-replay probe of the cfg-3 sampler, the wave-local single-query kernels (parity tests,
# micro-benchmark A/B against the round-4 kernels), the keypose tests that use them, the joint-iteration test.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05c; mkdir -p $O
timeout 300 python profiles/cfg3_graph_probe.py > $O/cfg3_probe.txt 2>&1; cat $O/cfg3_probe.txt | tail -12
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "single_query or query_stream or sq_" > $O/t_sq.log 2>&1; echo "sq rc=$? $(grep -E 'passed|failed' $O/t_sq.log | tail -1)"; grep -E "^FAILED|^ERROR|^E  " $O/t_sq.log | head -10
timeout 600 python -m pytest tests/test_act3d_gpu.py tests/test_joint_gpu.py tests/test_engine_gpu.py -q > $O/t_act3d.log 2>&1; echo "act3d rc=$? $(grep -E 'passed|failed' $O/t_act3d.log | tail -1)"; grep -E "^FAILED|^ERROR" $O/t_act3d.log | head -10
for w in 1 0; do A3D_SQ_WAVE=$w timeout 300 python bench.py --kernels-only > $O/kernels_wave$w.json 2> $O/kernels_wave$w.err; python - <<P
import json
try:
    k=json.load(open("$O/kernels_wave$w.json"))["kernels"]; print("A3D_SQ_WAVE=$w", {n: round(k[n]["ms"]*1e3,1) for n in ("sq_fwd","sq_bwd","attn_fwd","attn_bwd","kv_proj_rope") if n in k})
except Exception as e: print("kernels $w failed", e)
P
done
timeout 600 python bench.py --skip-secondary --skip-cpu-baseline > $O/bench_kp.json 2> $O/bench_kp.err
python - <<P
import json
try:
    d=json.load(open("$O/bench_kp.json")); print("bench_kp", round(d["value"],1), round(d["ms_per_step"],3), d.get("hot_path_only"))
except Exception as e: print("bench_kp failed", e)
P
