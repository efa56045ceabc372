#!/bin/bash
# A/B of the constant-divisor rope_merge_bwd kernel inside the eager keypose step (kernel trace, per-kernel averages).
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05r; mkdir -p $O
for v in 0 1; do
( cd /tmp && export TMPDIR=/tmp && A3D_ROPE_MERGE_PAIRS=$v timeout 200 rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/$O/trace$v" -o kp -- python "$GRAFT_REPO_ROOT/bench.py" --skip-secondary --skip-cpu-baseline --no-graph --steps 10 --warmup 4 > "$GRAFT_REPO_ROOT/$O/trace$v.log" 2>&1 )
DB=$(find $O/trace$v -name '*.db' | head -1); python profiles/summarize.py "$DB" > $O/kernel_trace_pairs$v.txt 2>&1; rm -rf $O/trace$v
echo "pairs=$v"; head -1 $O/kernel_trace_pairs$v.txt; grep -E "rope_merge|attn16_fwd|sqw_bwd" $O/kernel_trace_pairs$v.txt | cut -c1-110
done
