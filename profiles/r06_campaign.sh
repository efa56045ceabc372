#!/bin/bash
# Round-6 measurement campaign (one gpurun call per section; sections selected by name): counter passes, kernel traces, parity report.
#   usage: bash profiles/r06_campaign.sh pmc|trace|parity|bench
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r06; mkdir -p $O
case "$1" in
pmc)
  # FETCH_SIZE | WRITE_SIZE | MfmaUtil VALUBusy, one --pmc set per pass, --kernel-trace only (profiles/pmc_json_cmd.sh)
  bash profiles/pmc_json_cmd.sh $O/r06_pmc_B64.json 64 python "$PWD/bench.py" --kernels-only
  bash profiles/pmc_json_cmd.sh $O/r06_pmc_denoise_persist.json 64 python "$PWD/bench_denoise.py" --mode sample --batch 64 --horizon 16 --reps 1
  bash profiles/pmc_json_cmd.sh $O/r06_pmc_denoise_persist_L50.json 24 python "$PWD/bench_denoise.py" --mode sample --batch 24 --horizon 50 --reps 1
  python - <<P
import json
for f in ("r06_pmc_B64", "r06_pmc_denoise_persist", "r06_pmc_denoise_persist_L50"):
    try:
        d = json.load(open("$O/%s.json" % f))["kernels"]
        print(f, {k: (round(v["hbm_bytes"] / 1e6, 1), v.get("pmc")) for k, v in d.items()})
    except Exception as e:
        print(f, "ERR", e)
P
  ;;
pmc5)
  bash profiles/pmc_json_cmd.sh $O/r06_pmc_cfg5.json 16 python "$PWD/bench.py" --only-cfg5
  python -c "
import json; d=json.load(open('$O/r06_pmc_cfg5.json'))['kernels']; print({k: (round(v['hbm_bytes']/1e6,1), v.get('pmc')) for k,v in d.items()})"
  ;;
pmcdt)
  # the trajectory -> context cross-attention of the diffusion training step (bench_denoise.py's roofline entries)
  for shape in "22 50" "64 16"; do set -- $shape
    bash profiles/pmc_json_cmd.sh $O/r06_pmc_diffusion_attn_B$1_L$2.json $1 python "$PWD/bench_denoise.py" --mode attn --batch $1 --horizon $2
  done
  ;;
trace)
  ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/$O/trace" -o kp -- python "$GRAFT_REPO_ROOT/bench.py" --skip-secondary --skip-cpu-baseline --no-graph --steps 10 --warmup 4 > "$GRAFT_REPO_ROOT/$O/trace.log" 2>&1 )
  DB=$(find $O/trace -name '*.db' | head -1); python profiles/summarize.py "$DB" > $O/r06_kernel_trace_B64.txt 2>&1; rm -rf $O/trace
  head -42 $O/r06_kernel_trace_B64.txt | cut -c1-150
  ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/$O/trace2" -o dt -- python "$GRAFT_REPO_ROOT/bench_denoise.py" --mode train --no-graph --reps 5 > "$GRAFT_REPO_ROOT/$O/trace_dt.log" 2>&1 )
  DB=$(find $O/trace2 -name '*.db' | head -1); python profiles/summarize.py "$DB" > $O/r06_kernel_trace_diffusion_train.txt 2>&1; rm -rf $O/trace2
  head -32 $O/r06_kernel_trace_diffusion_train.txt | cut -c1-150
  ;;
parity)
  timeout 2400 python -m pytest tests -m gpu -q -s > $O/pytest_all.log 2>&1; echo "all rc=$? $(grep -E 'passed|failed' $O/pytest_all.log | tail -1)"; grep -E "^FAILED|^ERROR" $O/pytest_all.log | head -20
  grep "\[parity\]" $O/pytest_all.log > $O/r06_parity_report.txt; wc -l $O/r06_parity_report.txt
  ;;
bench)
  timeout 900 python bench.py > $O/r06_bench_B64.json 2> $O/r06_bench_B64.err; tail -c 600 $O/r06_bench_B64.json
  ;;
esac
