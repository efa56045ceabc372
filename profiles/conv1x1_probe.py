"""A/B of the opt-in fused 1x1-convolution path of the frozen backbone (A3D_FUSED_CONV1X1): backbone forward time at the
bench shape (256 images 256x256, bf16) with MIOpen convolutions + separate BatchNorm kernels vs the fused GEMM.
usage (GPU box): python profiles/conv1x1_probe.py"""
import importlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
a3d = importlib.import_module("act3d-chained-diffuser_amd")
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
torch.manual_seed(0)
bb = a3d.nn.SyntheticCLIPResNet50().to(dev).train()
x = torch.rand(256, 3, 256, 256, device=dev)
norm = a3d.nn.ClipNormalize().to(dev)
res = {}
for flag in (False, True, False, True):
    a3d.nn.FUSED_CONV1X1 = flag
    with torch.no_grad():
        for _ in range(3):
            a3d.nn.run_frozen_backbone(bb, x, torch.bfloat16, keep_dtype=True, normalize=norm)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            a3d.nn.run_frozen_backbone(bb, x, torch.bfloat16, keep_dtype=True, normalize=norm)
        e1.record()
        torch.cuda.synchronize()
    res.setdefault("fused_1x1" if flag else "miopen", []).append(e0.elapsed_time(e1) / 5)
a3d.nn.FUSED_CONV1X1 = False
print(json.dumps({"backbone_forward_ms": res, "images": 256}))
