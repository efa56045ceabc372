#!/bin/bash
# round-4 GPU call 1: parity suite, keypose bench line, eager kernel trace of the keypose step
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04a
python -m pytest tests -m gpu -q -s > gpurun_out/r04a/pytest.log 2>&1; echo "pytest rc=$?" > gpurun_out/r04a/rc.txt
grep -E "passed|failed|error" gpurun_out/r04a/pytest.log | tail -5
python bench.py --skip-secondary --skip-cpu-baseline > gpurun_out/r04a/bench_kp.json 2> gpurun_out/r04a/bench_kp.err; echo "bench rc=$?" >> gpurun_out/r04a/rc.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/gpurun_out/r04a/trace" -o kp -- python "$GRAFT_REPO_ROOT/bench.py" --skip-secondary --skip-cpu-baseline --no-graph --steps 10 --warmup 4 > "$GRAFT_REPO_ROOT/gpurun_out/r04a/trace.log" 2>&1
cd "$GRAFT_REPO_ROOT"
DB=$(find gpurun_out/r04a/trace -name '*.db' | head -1)
python profiles/summarize.py "$DB" > gpurun_out/r04a/kernel_trace_B64.txt 2>&1
rm -rf gpurun_out/r04a/trace
cat gpurun_out/r04a/rc.txt; head -c 600 gpurun_out/r04a/bench_kp.json; head -45 gpurun_out/r04a/kernel_trace_B64.txt
