#!/bin/bash
# rocprofv3 counter passes over the attention-core micro-benchmark (profiles/attn16_check.py --time-only), one --pmc set per
# pass, no trace domains mixed in.   usage: profiles/run_pmc_attn16.sh <out-dir> [family]
set -u
OUT=${1:-gpurun_out/r03/pmc_attn16}
FAM=${2:-f16}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
case "$OUT" in /*) ;; *) OUT="$R/$OUT";; esac
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
pass() {
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o "$name" -- python "$R/profiles/attn16_check.py" --time-only --family "$FAM" > "$OUT/$name.log" 2>&1
  echo "pass $name rc=$?" >> "$OUT/passes.txt"
  python "$R/profiles/pmc_summary.py" "$OUT/$name" attn > "$OUT/$name.summary.txt" 2>&1
  rm -rf "$OUT/$name"
}
: > "$OUT/passes.txt"
pass sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVES
pass stall SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
pass insts SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM
pass derived MfmaUtil VALUBusy
pass fetch FETCH_SIZE
pass write WRITE_SIZE
cat "$OUT/passes.txt"
cat "$OUT"/*.summary.txt
