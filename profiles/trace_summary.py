#!/usr/bin/env python3
"""Per-kernel totals of a rocprofv3 --kernel-trace database (rocpd sqlite): python profiles/trace_summary.py <results.db> [substr]
Prints calls, total ms, average us per kernel name (optionally only names containing `substr`), largest first."""
import collections
import sqlite3
import sys

db, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
rows = sqlite3.connect(db).execute("select name, start, end from kernels").fetchall()
agg = collections.defaultdict(lambda: [0, 0.0])
for name, s, e in rows:
    k = name.replace("void ", "").split("(")[0][:100]
    if sub in k:
        agg[k][0] += 1
        agg[k][1] += e - s
tot = sum(v[1] for v in agg.values())
print(f"# {len(rows)} dispatches; {tot / 1e6:.3f} ms in the selected kernels")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{v[1] / 1e6:10.3f} ms {v[0]:8d} calls {v[1] / v[0] / 1e3:9.2f} us avg  {k}")
