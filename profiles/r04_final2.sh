#!/bin/bash
# round-4 evidence refresh after conv3x3.hip / the hardware bf16 rounding in conv1x1.hip: parity report of the full GPU suite, the
# bench line, the eager kernel trace of the keypose step, the convolution probes (the counter passes of r04_final.sh cover the
# attention / single-query / k-NN / projection kernels, which did not change)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04y; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest.log | head
grep -E "^\.*\[parity\]|^\[parity\]" $O/pytest.log | sed 's/^\.*//' > $O/parity_report.txt
timeout 300 python profiles/conv3x3_probe.py > $O/conv3x3_probe.json 2> $O/probe3.err; cat $O/conv3x3_probe.json
timeout 300 python profiles/conv1x1_probe.py > $O/conv1x1_probe.json 2> $O/probe1.err; cat $O/conv1x1_probe.json
timeout 900 python bench.py > $O/bench_B64.json 2> $O/bench_B64.err; echo "bench rc=$?" >> $O/rc.txt
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/$O/trace" -o kp -- python "$GRAFT_REPO_ROOT/bench.py" --skip-secondary --skip-cpu-baseline --no-graph --steps 10 --warmup 4 > "$GRAFT_REPO_ROOT/$O/trace.log" 2>&1 )
DB=$(find $O/trace -name '*.db' | head -1); python profiles/summarize.py "$DB" > $O/kernel_trace_B64.txt 2>&1; rm -rf $O/trace
cat $O/rc.txt; head -c 400 $O/bench_B64.json; echo; head -30 $O/kernel_trace_B64.txt | cut -c1-150
