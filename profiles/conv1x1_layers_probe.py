"""Per-layer A/B of the frozen backbone's 1x1 convolutions at the bench shape (256 images): MIOpen's convolution alone and
followed by the BatchNorm statistics pass it needs (a3d_bn_stats), against the fused GEMM of csrc/conv1x1.hip with the
statistics in its epilogue (and, for the bottleneck's conv3 position, the producer's BatchNorm-apply + ReLU on its operand
load instead of a separate a3d_bn_apply pass).  Decides per shape which path the backbone runner takes (nn.py).
usage (GPU box): python profiles/conv1x1_layers_probe.py [images]"""
import importlib
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
a3d = importlib.import_module("act3d-chained-diffuser_amd")
L = a3d.lib
lib = L.load()
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
# (cin, cout, hw, count, position): position "conv1" (input already normalised), "conv3" (input = raw 3x3 output + bn2 apply), "ds"
SHAPES = [(64, 64, 64, 1, "conv1"), (64, 256, 64, 3, "conv3"), (64, 256, 64, 1, "ds"), (256, 64, 64, 2, "conv1"), (256, 128, 64, 1, "conv1"),
          (128, 512, 32, 4, "conv3"), (256, 512, 32, 1, "ds"), (512, 128, 32, 3, "conv1"), (512, 256, 32, 1, "conv1"),
          (256, 1024, 16, 6, "conv3"), (512, 1024, 16, 1, "ds"), (1024, 256, 16, 5, "conv1"), (1024, 512, 16, 1, "conv1"),
          (512, 2048, 8, 3, "conv3"), (1024, 2048, 8, 1, "ds"), (2048, 512, 8, 2, "conv1")]


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


tot = {"miopen": 0.0, "miopen+bn": 0.0, "fused": 0.0}
rows = []
for cin, cout, hw, count, pos in SHAPES:
    M = N * hw * hw
    x = torch.randn(N, cin, hw, hw, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, 1, 1, device=dev) / cin ** 0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w2 = w.reshape(cout, cin).contiguous()
    y = torch.empty(N, cout, hw, hw, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    xn = torch.empty_like(x)
    sc = torch.rand(2, cin, device=dev) + 0.5
    nslab = lib.a3d_bn_nslab(M, cout)
    part = torch.empty(nslab, 2, cout, device=dev)
    nslab_c = max(1, lib.a3d_conv1x1_nslab(M, cin, cout))
    part_c = torch.empty(nslab_c, 2, cout, device=dev)
    st = L.stream()
    t_conv = timeit(lambda: F.conv2d(x, w))
    yy = F.conv2d(x, w)

    def unfused():
        if pos == "conv3":        # bn2 apply + relu of the 3x3 output, then the convolution, then the statistics of its output
            L.call("a3d_bn_apply", x.data_ptr(), None, None, None, sc[0].data_ptr(), sc[1].data_ptr(), xn.data_ptr(), M, cin, 1, st)
            o = F.conv2d(xn, w)
        else:
            o = F.conv2d(x, w)
        L.call("a3d_bn_stats", o.data_ptr(), part.data_ptr(), M, cout, nslab, st)

    def fused():
        L.call("a3d_conv1x1_bn_fwd", x.data_ptr(), w2.data_ptr(), sc[0].data_ptr() if pos == "conv3" else None,
               sc[1].data_ptr() if pos == "conv3" else None, 1 if pos == "conv3" else 0, y.data_ptr(), part_c.data_ptr(), M, cin, cout, st)

    ok = bool(lib.a3d_conv1x1_streams(cin, cout))
    t_unf = timeit(unfused)
    t_f = timeit(fused) if ok else float("nan")
    mb = 2.0 * (x.numel() + y.numel()) / 1e6
    rows.append({"cin": cin, "cout": cout, "hw": hw, "n": count, "pos": pos, "miopen_us": round(t_conv, 1), "miopen_bn_us": round(t_unf, 1),
                 "fused_us": round(t_f, 1), "MB": round(mb, 1), "fused_TBps": round(mb / t_f, 2) if ok else None})
    print(rows[-1], flush=True)
    tot["miopen"] += t_conv * count
    tot["miopen+bn"] += t_unf * count
    tot["fused"] += (t_f if ok else t_unf) * count
print(json.dumps({"images": N, "total_ms": {k: round(v / 1e3, 3) for k, v in tot.items()}, "layers": rows}))
