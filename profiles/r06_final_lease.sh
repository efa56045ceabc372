#!/bin/bash
# Round-6 evidence on the final tree: (1) full GPU test suite + parity report, default bench line, kernel traces, counter passes of the
# keypose kernels, attention A/B record;  (2) the slower counter passes (fp8 attention, diffusion attention);  (3) refresh after the
# last changes: full suite, default bench line, traces, sampler / diffusion-training entries.   usage: ... 1|2|3
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r06; mkdir -p $O
case "${1:-1}" in
1)
  bash profiles/r06_campaign.sh parity 2>&1 | tail -4
  bash profiles/r06_campaign.sh bench 2>&1 | tail -c 300
  bash profiles/r06_campaign.sh trace > $O/call_trace.log 2>&1; tail -2 $O/call_trace.log | cut -c1-200
  A3D_LIB=libact3d_hip.so timeout 300 python profiles/attn_ab.py --check > $O/attn_ab_final.json 2>/dev/null; cut -c1-300 $O/attn_ab_final.json
  bash profiles/pmc_json_cmd.sh $O/r06_pmc_B64.json 64 python "$PWD/bench.py" --kernels-only > $O/call_pmc.log 2>&1
  python -c "
import json; d=json.load(open('$O/r06_pmc_B64.json'))['kernels']; print({k: (round(v['hbm_bytes']/1e6,1), v.get('pmc')) for k,v in d.items()})" | cut -c1-1500
  ;;
2)
  # counters of the attention kernels AS THE TRAINING STEP LAUNCHES THEM (eager step: the captured graph hides the kernel names)
  bash profiles/pmc_json_cmd.sh $O/r06_pmc_step_B64.json 64 python "$PWD/bench.py" --skip-secondary --skip-cpu-baseline --no-graph --steps 3 --warmup 2 > $O/call_pmc_step.log 2>&1
  python -c "
import json; d=json.load(open('$O/r06_pmc_step_B64.json'))['kernels']; print({k: (round(v['hbm_bytes']/1e6,1), v.get('pmc')) for k,v in d.items() if k.startswith('attn') or k.startswith('kv')})" | cut -c1-900
  bash profiles/r06_campaign.sh pmcdt > $O/call_pmcdt.log 2>&1; tail -2 $O/call_pmcdt.log | cut -c1-300
  bash profiles/r06_campaign.sh pmc5 > $O/call_pmc5.log 2>&1; tail -1 $O/call_pmc5.log | cut -c1-400
  ;;
3)
  bash profiles/r06_campaign.sh parity 2>&1 | tail -4
  bash profiles/r06_campaign.sh bench 2>&1 | tail -c 300
  bash profiles/r06_campaign.sh trace > $O/call_trace.log 2>&1; tail -2 $O/call_trace.log | cut -c1-200
  for a in "train 22 50" "sample 64 16" "sample 24 50"; do set -- $a
    timeout 600 python bench_denoise.py --mode $1 --batch $2 --horizon $3 2>/dev/null | tail -1 > $O/r06_denoise_$1_B$2_L$3.json
    python -c "
import json; d=json.load(open('$O/r06_denoise_$1_B$2_L$3.json')); print('$1 B=$2 L=$3', round(d['value'],1), d['unit'], d.get('ms_per_step'), d.get('ms_per_denoise_step'))"
  done
  ;;
esac
