#!/bin/bash
# Fourth GPU call of round 5: (1) which tree makes ONE fused denoise step run-to-run non-deterministic: the round-4 tree (.ab/r04),
# the current tree with round 4's sincos helper (.ab/v1), the current tree; (2) wave-local single-query kernels after the LDS /
# split-count fix: micro-benchmark A/B.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05d; mkdir -p $O
for d in .ab/r04 .ab/v1 .; do echo "== probe in $d"; (cd $d && timeout 200 python profiles/cfg3_graph_probe.py 2>&1 | grep -E "context run|n_steps|Error|error" | head -8); done > $O/cfg3_probe_ab.txt 2>&1; cat $O/cfg3_probe_ab.txt
for w in 1 0; do A3D_SQ_WAVE=$w timeout 300 python bench.py --kernels-only > $O/kernels_wave$w.json 2> $O/kernels_wave$w.err; python - <<P
import json
try:
    k=json.load(open("$O/kernels_wave$w.json"))["kernels"]; print("A3D_SQ_WAVE=$w", {n: round(k[n]["ms"]*1e3,1) for n in ("sq_fwd","sq_bwd") if n in k})
except Exception as e: print("kernels $w failed", e)
P
done
A3D_SQ_WGS=1024 timeout 300 python bench.py --kernels-only > $O/kernels_wave1_1024.json 2> /dev/null; python - <<P
import json
try:
    k=json.load(open("$O/kernels_wave1_1024.json"))["kernels"]; print("A3D_SQ_WAVE=1 WGS=1024", {n: round(k[n]["ms"]*1e3,1) for n in ("sq_fwd","sq_bwd") if n in k})
except Exception as e: print("failed", e)
P
A3D_SQ_WGS=256 timeout 300 python bench.py --kernels-only > $O/kernels_wave1_256.json 2> /dev/null; python - <<P
import json
try:
    k=json.load(open("$O/kernels_wave1_256.json"))["kernels"]; print("A3D_SQ_WAVE=1 WGS=256", {n: round(k[n]["ms"]*1e3,1) for n in ("sq_fwd","sq_bwd") if n in k})
except Exception as e: print("failed", e)
P
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "single_query or query_stream or sq_" 2>&1 | tail -2
