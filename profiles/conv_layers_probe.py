"""Per-layer timing of the frozen backbone's convolutions on MIOpen / CK (bf16 NHWC, find mode) at the bench shape
(256 images 256 x 256), with each layer's algorithmic FLOPs and bytes: the numbers behind the decision whether a hand-written
1x1-convolution GEMM with fused BatchNorm can pay (frozen backbone).   usage (GPU box): python profiles/conv_layers_probe.py [images]"""
import importlib
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
a3d = importlib.import_module("act3d-chained-diffuser_amd")
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
bb = a3d.nn.SyntheticCLIPResNet50().to(dev).train()
shapes = {}
hooks = []


def hook(name):
    def fn(m, inp, out):
        x = inp[0]
        key = (m.in_channels, m.out_channels, m.kernel_size[0], m.stride[0], x.shape[-1])
        shapes.setdefault(key, []).append(name)
    return fn


for name, m in bb.named_modules():
    if isinstance(m, torch.nn.Conv2d):
        hooks.append(m.register_forward_hook(hook(name)))
with torch.no_grad():
    bb(torch.rand(2, 3, 256, 256, device=dev))
for h in hooks:
    h.remove()
rows = []
tot = {"1x1": 0.0, "3x3": 0.0}
for (cin, cout, k, stride, hw), names in sorted(shapes.items(), key=lambda kv: (-kv[0][4], kv[0][2], kv[0][0])):
    x = torch.randn(N, cin, hw, hw, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, k, k, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    for _ in range(3):
        y = F.conv2d(x, w, None, stride, k // 2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        y = F.conv2d(x, w, None, stride, k // 2)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100.0
    ho = y.shape[-1]
    gflop = 2.0 * N * ho * ho * cout * cin * k * k / 1e9
    mb = 2.0 * (x.numel() + y.numel() + w.numel()) / 1e6
    rows.append({"cin": cin, "cout": cout, "k": k, "stride": stride, "hw": hw, "layers": len(names), "us": round(us, 1), "gflop": round(gflop, 1),
                 "MB": round(mb, 1), "TFLOPs": round(gflop / us * 1e3 / 1e3, 1), "TBps": round(mb / us, 2),
                 "roofline_us": round(max(gflop / 2.5e3 * 1e3 / 1e3, mb / 8.0), 1)})
    tot["1x1" if k == 1 else "3x3"] += us * len(names)
    print(rows[-1], flush=True)
print(json.dumps({"images": N, "total_ms_1x1": round(tot["1x1"] / 1e3, 3), "total_ms_3x3": round(tot["3x3"] / 1e3, 3), "layers": rows}))
