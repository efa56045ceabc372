#!/bin/bash
# after the 64 -> 64 output-channel split and the unrolled 32 -> 64 instance of conv3x3.hip: convolution tests, the probe with and
# without the split, the bench line, the keypose kernel trace
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04x; mkdir -p $O
timeout 200 python -m pytest tests/test_kernels_gpu.py -q -s -k "conv3x3 or backbone or conv1x1" > $O/k.log 2>&1; echo "pytest rc=$?" > $O/rc.txt; grep -E "passed|failed" $O/k.log | tail -2; grep -E "^FAILED|^ERROR|max err" $O/k.log | head
timeout 120 python profiles/conv3x3_probe.py > $O/conv3x3_probe.json 2> $O/probe3.err; cat $O/conv3x3_probe.json
A3D_C3_SPLIT=0 timeout 120 python profiles/conv3x3_probe.py > $O/conv3x3_probe_nosplit.json 2> /dev/null; cat $O/conv3x3_probe_nosplit.json
timeout 400 python bench.py > $O/bench_B64.json 2> $O/bench_B64.err; echo "bench rc=$?" >> $O/rc.txt
cat $O/rc.txt; head -c 300 $O/bench_B64.json; echo
( cd /tmp && export TMPDIR=/tmp && timeout 150 rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/$O/trace" -o kp -- python "$GRAFT_REPO_ROOT/bench.py" --skip-secondary --skip-cpu-baseline --no-graph --steps 10 --warmup 4 > "$GRAFT_REPO_ROOT/$O/trace.log" 2>&1 )
DB=$(find $O/trace -name '*.db' | head -1); python profiles/summarize.py "$DB" > $O/kernel_trace_B64.txt 2>&1; rm -rf $O/trace
head -12 $O/kernel_trace_B64.txt | cut -c1-150; grep conv3x3 $O/kernel_trace_B64.txt | cut -c1-150
