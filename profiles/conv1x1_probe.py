"""A/B of the opt-in fused 1x1-convolution path of the frozen backbone (A3D_FUSED_CONV1X1): backbone forward time at the
bench shape (256 images 256x256, bf16) with MIOpen convolutions + separate BatchNorm kernels vs the fused GEMM.
usage (GPU box): python profiles/conv1x1_probe.py
Round 3 ran this with ten wider-tile candidates as well (profiles/r03_conv1x1_probe.json: best 12.6 ms against MIOpen's 11.9 ms,
and none of the ten reproduced the reference output); they were deleted, the verified 4-wave kernel stays opt-in."""
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
for label, flag in [("miopen", False), ("fused_stream", True)]:
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
    res[label] = e0.elapsed_time(e1) / 5
a3d.nn.FUSED_CONV1X1 = True
print(json.dumps({"backbone_forward_ms": res, "images": 256}))
