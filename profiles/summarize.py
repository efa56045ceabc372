#!/usr/bin/env python3
"""Summarises a rocprofv3 --kernel-trace database (rocpd sqlite, ROCm 7.2) into a per-training-step kernel table.

usage: python profiles/summarize.py <results.db> [first_step last_step]
Steps are delimited by the fused AdamW kernel (one launch per training step); the default window skips the first
steps (MIOpen solver search, allocator warm-up)."""
import collections
import sqlite3
import sys


def main():
    db = sys.argv[1]
    rows = sqlite3.connect(db).execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if "adamw_kernel" in r[0]]
    rows3 = rows
    a = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    b = int(sys.argv[3]) if len(sys.argv) > 3 else len(marks) - 1
    b = min(b, len(marks) - 1)
    a = min(a, b - 1)
    n = b - a
    win = rows[marks[a] + 1:marks[b] + 1]
    agg = collections.defaultdict(lambda: [0, 0.0])
    shapes = collections.defaultdict(lambda: [0, 0.0])
    for name, s, e, gx, gy, gz, wx in win:
        k = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        k = k.split("<")[0] if k.startswith("at::native::") and "elementwise" not in k else k[:110]
        agg[k][0] += 1
        agg[k][1] += e - s
        if "a3d::" in k:
            kk = (k, gx // max(wx, 1), gy, gz)
            shapes[kk][0] += 1
            shapes[kk][1] += e - s
    tot = sum(v[1] for v in agg.values())
    mine = sum(v[1] for k, v in agg.items() if "a3d::" in k)
    print(f"# steps {a}..{b} ({n} steps); wall/step {(win[-1][2] - win[0][1]) / n / 1e6:.3f} ms; kernel time/step {tot / n / 1e6:.3f} ms; "
          f"dispatches/step {len(win) / n:.0f}; libact3d_hip kernels/step {mine / n / 1e6:.3f} ms")
    print(f"{'us/step':>10} {'calls/step':>10} {'avg us':>9} {'% of step':>9}  kernel")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{v[1] / n / 1e3:10.1f} {v[0] / n:10.1f} {v[1] / v[0] / 1e3:9.1f} {100 * v[1] / tot:9.2f}  {k}")
    print("\n# libact3d_hip kernels by launch grid (workgroups x, y, z)")
    for kk, v in sorted(shapes.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"{v[1] / n / 1e3:10.1f} {v[0] / n:10.1f} {v[1] / v[0] / 1e3:9.1f}  {kk[0]} grid=({kk[1]},{kk[2]},{kk[3]})")


if __name__ == "__main__":
    main()
