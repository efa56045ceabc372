#!/bin/bash
# second attention lease of round 6: bf16 split variants of the dK / dV kernel (errors + timings), m0 handling, key splits, QT = 1
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r06/b; mkdir -p $O; export PYTHONUNBUFFERED=1
run() { tag=$1; shift; env "$@" timeout 300 python profiles/attn_ab.py $EXTRA 2>$O/$tag.err | sed "s/^{/{\"tag\": \"$tag\", /" | tee -a $O/attn_ab.jsonl; }
EXTRA="--check" run base A3D_LIB=libact3d_hip_base.so
EXTRA="--check" run new A3D_LIB=libact3d_hip.so
EXTRA="--check" run rne A3D_LIB=libact3d_hip_rne.so
EXTRA="--check" run trunc2 A3D_LIB=libact3d_hip_trunc2.so
EXTRA="" run m0 A3D_LIB=libact3d_hip_m0.so
EXTRA="--nsplit 2" run ns2 A3D_LIB=libact3d_hip.so
EXTRA="--nsplit 4" run ns4 A3D_LIB=libact3d_hip.so
EXTRA="" run qt1 A3D_LIB=libact3d_hip.so A3D_ATTN_QT=1
python - <<P
import json
for l in open("$O/attn_ab.jsonl"):
    if not l.startswith("{"): continue
    r=json.loads(l)
    print("%-8s fwd_train %.4f fwd_nograd %.4f bwd %.4f ns %d sha %s" % (r["tag"], r["fwd_train_ms"], r["fwd_nograd_ms"], r["bwd_ms"], r["nsplit"], " ".join(r["sha"].values())))
    for k in ("err_mild","err_sharp"):
        if k in r: print("         ", k, {a: "%.2e" % b for a,b in r[k].items()})
P
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/trace" -o ab -- python "$GRAFT_REPO_ROOT/profiles/attn_ab.py" > "$GRAFT_REPO_ROOT/$O/trace.log" 2>&1 )
find $O/trace -type f | head; f=$(find $O/trace -name '*kernel_stats*' | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-200 > $O/kernel_stats.txt; rm -rf $O/trace; cat $O/kernel_stats.txt
