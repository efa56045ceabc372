"""Phase breakdown of the persistent sampler (a3d_dn_persist) at the cfg-3 shape (B = 64, L = 16, 3 cameras): timestamps of sample
workgroup 0 per layer of the SECOND denoise step (head | per layer: q projection, publish, staging, wait for the streamers, combine,
layer remainder | tail) and of streamer workgroup 0's first 32 queue items (ticket -> entry ready -> item done).
usage (GPU box): A3D_DN_PROF=1 python profiles/dn_persist_phases.py [n_steps]"""
import ctypes
import importlib
import json
import os
import sys

import torch

os.environ.setdefault("A3D_DN_PROF", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_denoise as BD  # noqa: E402

a3d = importlib.import_module("act3d-chained-diffuser_amd")
dev = torch.device("cuda:0")
n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
B, Ln, C = 64, 16, 3
m = BD.build_planner(a3d, dev, train=False)
s = BD.synthetic_inputs(B, Ln, C, dev)
with torch.no_grad():
    tokens = m.prediction_head.encode_images(s["rgbs"], None).contiguous()
for _ in range(2):
    out = m.compute_trajectory(s["trajectory_mask"], None, s["pcds"], s["instr"], s["curr_gripper"], s["action"],
                               init_noise=torch.randn(B, Ln, 9, device=dev), step_noise=torch.randn(100, B, Ln, 9, device=dev),
                               visual_tokens=tokens, n_steps=n_steps)
torch.cuda.synchronize()
ps = m.prediction_head._last_persist
buf = (ctypes.c_longlong * 256)()
nl = sum(ps["stacks"])
a3d.lib.call("a3d_dn_persist_prof", ps["sync"].data_ptr(), B, Ln, nl, n_steps, ctypes.cast(buf, ctypes.c_void_p).value)
t = [buf[i] for i in range(256)]
us = lambda a, b: round((b - a) * 0.01, 2)
layers = []
for l in range(nl):
    k = t[96 + 7 * l: 96 + 7 * l + 7]
    layers.append({"q projection + rope": us(k[0], k[1]), "publish (write q, release, enqueue)": us(k[1], k[2]), "stage vectors / warm L2 / table": us(k[2], k[3]),
                   "wait for the streamers": us(k[3], k[4]), "combine": us(k[4], k[5]), "layer remainder (13 ops)": us(k[5], k[6]), "layer total": us(k[0], k[6])})
items = [{"wait for ticket + entry": us(t[3 * i], t[3 * i + 1]), "stream + publish": us(t[3 * i + 1], t[3 * i + 2])} for i in range(32) if t[3 * i + 2]]
# wave 0 of streamer 0, first 14 items (round 6): entry ready -> queries loaded -> fragment loads issued -> key range consumed -> item done
for i in range(min(14, len(items))):
    k = t[208 + 3 * i: 208 + 3 * i + 3]
    if k[2]:
        items[i].update({"q loads": us(t[3 * i + 1], k[0]), "issue first fragments": us(k[0], k[1]), "key loop": us(k[1], k[2]),
                         "partials + completion": us(k[2], t[3 * i + 2])})
print(json.dumps({"shape": {"B": B, "L": Ln, "S": C * 1024 + 2, "nsplit": ps["nsplit"], "n_steps": n_steps},
                  "sample_0_step_1": {"head_us": us(t[250], t[96]), "layers": layers, "tail_us": us(t[251], t[252]), "step_us": us(t[250], t[252])},
                  "streamer_0_items": items, "abort_word": int(ps["sync"][2].item())}, indent=1))
