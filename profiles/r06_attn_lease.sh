#!/bin/bash
# Round-6 attention lease: co-issue microbenchmark, A/B timings + digests of the library builds named on the command line, kernel
# trace of the default build, the attention / golden / sampler tests.   usage: bash profiles/r06_attn_lease.sh <tag> lib1 lib2 ...
cd "${GRAFT_REPO_ROOT:-.}"; TAG=$1; shift; O=gpurun_out/r06/$TAG; mkdir -p $O
export PYTHONUNBUFFERED=1
if [ -x profiles/ubench/coissue ] && [ ! -f gpurun_out/r06/coissue.txt ]; then timeout 300 profiles/ubench/coissue > gpurun_out/r06/coissue.txt 2>&1; fi
for lib in "$@"; do
  A3D_LIB=$lib timeout 300 python profiles/attn_ab.py 2>$O/ab_$lib.err | tee -a $O/attn_ab.jsonl
done
python - <<P
import json
rows=[json.loads(l) for l in open("$O/attn_ab.jsonl") if l.startswith("{")]
for r in rows: print("%-28s fwd_train %.4f fwd_nograd %.4f bwd %.4f  sha %s" % (r["lib"], r["fwd_train_ms"], r["fwd_nograd_ms"], r["bwd_ms"], " ".join(r["sha"].values())))
P
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/trace" -o ab -- python "$GRAFT_REPO_ROOT/profiles/attn_ab.py" > "$GRAFT_REPO_ROOT/$O/trace.log" 2>&1 )
f=$(find $O/trace -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cut -d, -f1-8 "$f" | head -12 > $O/kernel_stats.txt; rm -rf $O/trace; cat $O/kernel_stats.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "attn or rows_only" 2>&1 | tail -4 | tee $O/tests_attn.txt
timeout 900 python -m pytest tests/test_act3d_gpu.py -q -x -k "golden or cfg4" 2>&1 | tail -4 | tee $O/tests_act3d.txt
