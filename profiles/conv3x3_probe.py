"""A/B of the 3x3 implicit-GEMM path of the frozen backbone (A3D_FUSED_CONV3X3): per-layer time of a3d_conv3x3_bn_fwd (BatchNorm-apply
of the producer + statistics of the consumer folded in) against what it replaces (a3d_bn_apply + MIOpen convolution + a3d_bn_stats)
at the bench shapes (256 images 256x256), and the whole backbone forward with the path on / off.
usage (GPU box): python profiles/conv3x3_probe.py"""
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
torch.manual_seed(0)


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


layers = []
for (cin, cout, hw) in [(32, 32, 128), (32, 64, 128), (64, 64, 64)]:
    N = 256
    x = torch.randn(N, cin, hw, hw, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1, bias=False).to(dev).to(torch.bfloat16)
    conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
    bn_in = torch.nn.BatchNorm2d(cin).to(dev).train()
    bn_out = torch.nn.BatchNorm2d(cout).to(dev).train()
    with torch.no_grad():
        scale = a3d.nn.bn_scale_shift(x, bn_in)
        fused = lambda: a3d.nn.conv3x3_bn(x, conv, in_scale=scale, in_relu=True, want_stats=True)

        def unfused():
            y = torch.empty_like(x)
            a3d.lib.call("a3d_bn_apply", x.data_ptr(), None, None, None, scale[0].data_ptr(), scale[1].data_ptr(), y.data_ptr(), N * hw * hw, cin, 1,
                         a3d.lib.stream())
            c = F.conv2d(y, conv.weight, None, 1, 1)
            nslab = a3d.lib.load().a3d_bn_nslab(N * hw * hw, cout)
            part = torch.empty((nslab, 2, cout), device=dev, dtype=torch.float32)
            a3d.lib.call("a3d_bn_stats", c.data_ptr(), part.data_ptr(), N * hw * hw, cout, nslab, a3d.lib.stream())
            return c

        conv_only = lambda: F.conv2d(x, conv.weight, None, 1, 1)
        t_f, t_u, t_c = timed(fused), timed(unfused), timed(conv_only)
        ya, _ = fused()
        yb = unfused()
        diff = (ya.float() - yb.float()).abs().max().item()
    px = N * hw * hw
    layers.append({"cin": cin, "cout": cout, "hw": hw, "fused_us": round(t_f, 1), "bn_apply+miopen+bn_stats_us": round(t_u, 1),
                   "miopen_conv_only_us": round(t_c, 1), "MB_in_out": round(px * (cin + cout) * 2 / 1e6, 1),
                   "fused_TBps": round(px * (cin + cout) * 2 / t_f / 1e6, 2), "fused_PFLOPs": round(2 * px * 9 * cin * cout / t_f / 1e9, 3),
                   "max_abs_diff_vs_unfused": diff})
    del x

bb = a3d.nn.SyntheticCLIPResNet50().to(dev).train()
x = torch.rand(256, 3, 256, 256, device=dev)
norm = a3d.nn.ClipNormalize().to(dev)
res = {}
for label, flag in [("miopen_3x3", False), ("fused_3x3", True)]:
    a3d.nn.FUSED_CONV3X3 = flag
    with torch.no_grad():
        res[label] = timed(lambda: a3d.nn.run_frozen_backbone(bb, x, torch.bfloat16, keep_dtype=True, normalize=norm), reps=5) / 1e3
a3d.nn.FUSED_CONV3X3 = True
print(json.dumps({"layers": layers, "backbone_forward_ms": res, "images": 256}))
