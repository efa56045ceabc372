#!/bin/bash
# round-4 evidence: parity report of the full GPU suite, the bench line, eager kernel traces (keypose, diffusion training),
# counter passes (keypose kernels at B = 64; the diffusion training attention micro-benchmark; the diffusion training step)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04z; mkdir -p $O
python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED" $O/pytest.log | head
grep -E "^\.*\[parity\]|^\[parity\]" $O/pytest.log | sed 's/^\.*//' > $O/parity_report.txt
profiles/pmc_json_cmd.sh $O/pmc_B64.json 64 python "$GRAFT_REPO_ROOT/bench.py" --kernels-only --batch 64
profiles/pmc_json_cmd.sh $O/pmc_diffusion_attn_B22_L50.json 22 python "$GRAFT_REPO_ROOT/bench_denoise.py" --mode attn --batch 22 --horizon 50
profiles/pmc_json_cmd.sh $O/pmc_diffusion_attn_B64_L16.json 64 python "$GRAFT_REPO_ROOT/bench_denoise.py" --mode attn --batch 64 --horizon 16
cp $O/pmc_B64.json profiles/r04_pmc_B64.json; cp $O/pmc_diffusion_attn_B22_L50.json profiles/r04_pmc_diffusion_attn_B22_L50.json; cp $O/pmc_diffusion_attn_B64_L16.json profiles/r04_pmc_diffusion_attn_B64_L16.json
python bench.py > $O/bench_B64.json 2> $O/bench_B64.err; echo "bench rc=$?" >> $O/rc.txt
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/$O/trace" -o kp -- python "$GRAFT_REPO_ROOT/bench.py" --skip-secondary --skip-cpu-baseline --no-graph --steps 10 --warmup 4 > "$GRAFT_REPO_ROOT/$O/trace.log" 2>&1 )
DB=$(find $O/trace -name '*.db' | head -1); python profiles/summarize.py "$DB" > $O/kernel_trace_B64.txt 2>&1; rm -rf $O/trace
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/$O/trace2" -o dt -- python "$GRAFT_REPO_ROOT/bench_denoise.py" --mode train --no-graph > "$GRAFT_REPO_ROOT/$O/trace2.log" 2>&1 )
DB=$(find $O/trace2 -name '*.db' | head -1); python profiles/summarize.py "$DB" 4 12 > $O/kernel_trace_diffusion_train.txt 2>&1; python profiles/trace_summary.py "$DB" > $O/kernel_totals_diffusion_train.txt 2>&1; rm -rf $O/trace2
cat $O/rc.txt; head -c 500 $O/bench_B64.json; echo; head -12 $O/kernel_trace_B64.txt; head -8 $O/kernel_trace_diffusion_train.txt
