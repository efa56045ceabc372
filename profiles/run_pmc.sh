#!/bin/bash
# rocprofv3 counter passes over the dominant-kernel micro-benchmarks (bench.py --kernels-only), one --pmc set per pass
# as the MI355X guide prescribes (TCC has 4 slots: FETCH_SIZE and WRITE_SIZE cannot share a pass).  No sys/hip/hsa
# trace domains are combined with --pmc.   usage: [PASSES="fetch write derived"] profiles/run_pmc.sh <out-dir> [batch]
set -u
OUT=${1:-gpurun_out/pmc}
B=${2:-16}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
case "$OUT" in /*) ;; *) OUT="$R/$OUT";; esac
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$OUT/counters_available.txt" 2>&1
pass() {
  name=$1; shift
  case " ${PASSES:-fetch write sq stall insts derived} " in *" $name "*) ;; *) return;; esac
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o "$name" -- \
      python "$R/bench.py" --kernels-only --batch "$B" > "$OUT/$name.log" 2>&1
  echo "pass $name rc=$?" >> "$OUT/passes.txt"
  python "$R/profiles/pmc_summary.py" "$OUT/$name" attn_ linear_ bn_ proj_rope > "$OUT/$name.summary.txt" 2>&1
  find "$OUT/$name" -name '*.csv' -size +8M -delete
}
: > "$OUT/passes.txt"
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
pass stall SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
pass insts SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_IDX_ACTIVE SQ_WAVES
pass derived MfmaUtil VALUBusy
cat "$OUT/passes.txt"
