#!/bin/bash
# Round-5 counter passes (FETCH_SIZE | WRITE_SIZE | MfmaUtil VALUBusy, one --pmc set per pass, --kernel-trace only): the dominant
# keypose kernels (bench.py --kernels-only) and the samplers (persistent and per-phase).
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05p; mkdir -p $O
timeout 600 bash profiles/pmc_json_cmd.sh $O/pmc_B64.json 64 python "$GRAFT_REPO_ROOT/bench.py" --kernels-only > $O/pmc_B64.log 2>&1; tail -3 $O/pmc_B64.log
python - <<P
import json
try:
    d=json.load(open("$O/pmc_B64.json"))["kernels"]
    for k,v in d.items(): print(k, round(v["hbm_bytes"]/1e6,1), "MB", v.get("pmc"))
except Exception as e: print("pmc B64 failed", e)
P
A3D_DN_PERSIST=1 timeout 400 bash profiles/pmc_json_cmd.sh $O/pmc_denoise_persist.json 64 python "$GRAFT_REPO_ROOT/bench_denoise.py" --mode sample --reps 1 > $O/pmc_dn1.log 2>&1; tail -3 $O/pmc_dn1.log
A3D_DN_PERSIST=0 timeout 400 bash profiles/pmc_json_cmd.sh $O/pmc_denoise_perphase.json 64 python "$GRAFT_REPO_ROOT/bench_denoise.py" --mode sample --reps 1 > $O/pmc_dn0.log 2>&1; tail -3 $O/pmc_dn0.log
python - <<P
import json
for f in ("pmc_denoise_persist", "pmc_denoise_perphase"):
    try:
        d=json.load(open("$O/%s.json" % f))["kernels"]
        for k,v in d.items(): print(f, k, round(v["hbm_bytes"]/1e6,1), "MB", v.get("pmc"))
    except Exception as e: print(f, "failed", e)
P
