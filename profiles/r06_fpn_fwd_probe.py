"""Forward convolutions of the FPN at the bench shape (256 images 256 x 256, 64 padded channels, bf16 NHWC): the library's kernels
against the backbone's own streaming kernels (a3d_conv1x1_bn_fwd / a3d_conv3x3_bn_fwd without their BatchNorm folds), and the
top-down pass that a fused lateral epilogue would absorb.   usage (GPU box): python profiles/r06_fpn_fwd_probe.py"""
import importlib
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
a3d = importlib.import_module("act3d-chained-diffuser_amd")
nnm = importlib.import_module("act3d-chained-diffuser_amd.nn")
dev = torch.device("cuda:0")
torch.manual_seed(0)
cl = torch.channels_last


def timed(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3          # us


N = 256
out = {}
with torch.no_grad():
    for name, K, hw in (("res1", 64, 128), ("res2", 256, 64)):
        x = torch.randn(N, K, hw, hw, device=dev).to(torch.bfloat16).contiguous(memory_format=cl)
        conv = torch.nn.Conv2d(K, 64, 1, bias=False).to(dev).to(torch.bfloat16)
        top = torch.randn(N, 64, hw // 2, hw // 2, device=dev).to(torch.bfloat16).contiguous(memory_format=cl)
        bias = torch.randn(60, device=dev)
        lib = lambda: F.conv2d(x, conv.weight)
        own = lambda: nnm.conv1x1_bn(x, conv, want_stats=False)[0]
        lat = lib()
        td = lambda: nnm._Upsample2AddFn.apply(lat, top, bias)
        d = (own().float() - lat.float()).abs().max().item()
        out["lateral " + name] = {"library_us": timed(lib), "own_conv1x1_us": timed(own), "top_down_us": timed(td), "max_abs_diff": d,
                                  "bytes_fused_MB": (x.numel() + top.numel() + lat.numel()) * 2 / 1e6}
        print("lateral", name, out["lateral " + name], flush=True)
        del x, top, lat
    for name, hw in (("res1", 128), ("res3", 32)):
        x = torch.randn(N, 64, hw, hw, device=dev).to(torch.bfloat16).contiguous(memory_format=cl)
        conv = torch.nn.Conv2d(64, 64, 3, padding=1, bias=False).to(dev).to(torch.bfloat16)
        conv.weight.data = conv.weight.data.contiguous(memory_format=cl)
        lib = lambda: F.conv2d(x, conv.weight, None, 1, 1)
        own = lambda: nnm.conv3x3_bn(x, conv, want_stats=False)[0]
        d = (own().float() - lib().float()).abs().max().item()
        out["output 3x3 " + name] = {"library_us": timed(lib), "own_conv3x3_us": timed(own), "max_abs_diff": d}
        print("output 3x3", name, out["output 3x3 " + name], flush=True)
        del x
os.makedirs("gpurun_out/r06", exist_ok=True)
json.dump(out, open("gpurun_out/r06/r06_fpn_fwd_probe.json", "w"), indent=1)
