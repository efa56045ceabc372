"""A/B of the opt-in fused 1x1-convolution path of the frozen backbone (A3D_FUSED_CONV1X1): backbone forward time at the
bench shape (256 images 256x256, bf16) with MIOpen convolutions + separate BatchNorm kernels vs the fused GEMM.
usage (GPU box): python profiles/conv1x1_probe.py [tile ...]     (tile = "wm,wn,ksub", see csrc/conv1x1.hip)"""
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
# (label, fused flag, A3D_C1_TILE): the MIOpen path, the verified default tile, then the tuning candidates
runs = [("miopen", False, None), ("fused_default", True, None)] + \
       [(f"fused_tile_{t}", True, t) for t in (sys.argv[1:] or ["1,4,2", "2,4,1", "2,4,2"])]
for label, flag, tile in runs:
    a3d.nn.FUSED_CONV1X1 = flag
    if tile is None:
        os.environ.pop("A3D_C1_TILE", None)
    else:
        os.environ["A3D_C1_TILE"] = tile
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
a3d.nn.FUSED_CONV1X1 = False
os.environ.pop("A3D_C1_TILE", None)
print(json.dumps({"backbone_forward_ms": res, "images": 256}))
