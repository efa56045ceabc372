#!/usr/bin/env python3
"""Instruction histogram per basic block of one kernel of a csrc/*.hip file (what the compiler actually emitted for a
main loop).   usage: python profiles/isa_blocks.py attention16.hip attn16_fwd_kernelILb0ELi2 [min_block_size]"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "act3d-chained-diffuser_amd"))
import build as B  # noqa: E402

src, pat = sys.argv[1], sys.argv[2]
minsz = int(sys.argv[3]) if len(sys.argv) > 3 else 30
out = "/tmp/isa_%s.s" % src.replace(".hip", "")
subprocess.run(["hipcc"] + B.FLAGS + B.EXTRA_FLAGS.get(src, []) + ["-S", "--cuda-device-only", os.path.join(B.CSRC, src), "-o", out],
               check=True, capture_output=True)
lines = open(out).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*%s\S*:" % re.escape(pat), l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
print(lines[start].split(":")[0])
blocks = collections.OrderedDict()
cur = "entry"
blocks[cur] = []
for l in lines[start + 1:end]:
    l = l.strip()
    m = re.match(r"(\.LBB\d+_\d+):", l)
    if m:
        cur = m.group(1)
        blocks[cur] = []
        continue
    if not l or l.startswith(";") or l.startswith("."):
        continue
    blocks[cur].append(l)


def klass(i):
    op = i.split()[0]
    for pre, k in (("v_mfma", "mfma"), ("v_exp", "exp"), ("v_log", "log"), ("v_rcp", "rcp"), ("v_cvt_pk", "cvt_pk"), ("v_max3", "max3"), ("v_max", "max"),
                   ("v_pk_", "pk"), ("v_permlane", "permlane"), ("v_accvgpr", "accvgpr"), ("v_cndmask", "cndmask"), ("v_cmp", "cmp"), ("v_mov", "mov"), ("v_", "valu_other"),
                   ("ds_read", "ds_read"), ("ds_write", "ds_write"), ("ds_", "ds_other"), ("global_load", "gload"), ("buffer_load", "gload"),
                   ("global_store", "gstore"), ("s_waitcnt", "waitcnt"), ("s_nop", "nop"), ("s_barrier", "barrier"), ("s_cbranch", "branch"), ("s_", "salu")):
        if op.startswith(pre):
            return k
    return op


for b, ins in blocks.items():
    if len(ins) < minsz:
        continue
    c = collections.Counter(klass(i) for i in ins)
    valu = sum(v for k, v in c.items() if k in ("exp", "log", "rcp", "cvt_pk", "max3", "max", "pk", "permlane", "accvgpr", "cndmask", "cmp", "mov", "valu_other"))
    print("%-10s n=%4d  VALU(non-mfma)=%4d  %s" % (b, len(ins), valu, dict(sorted(c.items(), key=lambda kv: -kv[1]))))
