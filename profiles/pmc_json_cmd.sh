#!/bin/bash
# rocprofv3 counter passes (FETCH_SIZE | WRITE_SIZE | MfmaUtil VALUBusy: one --pmc set per pass, --kernel-trace only, no
# sys / hip / hsa trace domains) over an arbitrary command, folded by profiles/pmc_to_json.py into one JSON keyed by bench.py's
# kernel names.   usage: profiles/pmc_json_cmd.sh <out.json> <label> <command...>
set -u
OUTJ=$1; LABEL=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
case "$OUTJ" in /*) ;; *) OUTJ="$R/$OUTJ";; esac
TMP=$(mktemp -d /tmp/pmcXXXX)
CMD=("$@")
cd /tmp && export TMPDIR=/tmp
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "derived MfmaUtil VALUBusy"; do
  set -- $pass; name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$TMP/$name" -o "$name" -- "${CMD[@]}" > "$TMP/$name.log" 2>&1
  echo "pass $name rc=$?"
done
python "$R/profiles/pmc_to_json.py" "$TMP" "$LABEL" > "$OUTJ"
rm -rf "$TMP"
