#!/bin/bash
# rocprofv3 counter passes (one --pmc set per pass, no trace domains mixed in) over an arbitrary command.
# usage: profiles/run_pmc_cmd.sh <out-dir> "<kernel filters for the summary>" <command...>
set -u
OUT=$1; FILT=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
case "$OUT" in /*) ;; *) OUT="$R/$OUT";; esac
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
pass() {
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o "$name" -- "${CMD[@]}" > "$OUT/$name.log" 2>&1
  echo "pass $name rc=$?" >> "$OUT/passes.txt"
  python "$R/profiles/pmc_summary.py" "$OUT/$name" $FILT > "$OUT/$name.summary.txt" 2>&1
  rm -rf "$OUT/$name"
}
CMD=("$@")
: > "$OUT/passes.txt"
pass derived MfmaUtil VALUBusy
pass fetch FETCH_SIZE
pass write WRITE_SIZE
cat "$OUT/passes.txt"
