"""Times a3d_stem_conv_bn_fwd at the bench shape (256 images of 256 x 256) against the path it replaces (normalisation pass + MIOpen
convolution + a3d_bn_stats).   usage (GPU box): python profiles/stem_probe.py"""
import importlib
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
a3d = importlib.import_module("act3d-chained-diffuser_amd")
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
x = torch.rand(N, 3, 256, 256, device=dev)
norm = a3d.nn.ClipNormalize().to(dev)
conv = torch.nn.Conv2d(3, 32, 3, stride=2, padding=1, bias=False).to(dev).to(torch.bfloat16)
conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


L = a3d.lib
M = N * 128 * 128
part = torch.empty(L.load().a3d_bn_nslab(M, 32), 2, 32, device=dev)


def unfused():
    xb = a3d.nn.normalize_to_nhwc_bf16(x, norm)
    y = F.conv2d(xb, conv.weight, None, 2, 1)
    L.call("a3d_bn_stats", y.data_ptr(), part.data_ptr(), M, 32, part.shape[0], L.stream())
    return y


rec = {"images": N, "fused_us": timeit(lambda: a3d.nn.stem_conv_bn(x, conv, norm)), "normalize_conv_stats_us": timeit(unfused),
       "normalize_us": timeit(lambda: a3d.nn.normalize_to_nhwc_bf16(x, norm))}
print(json.dumps(rec))
