"""Times a3d_resize_crop (the GPU `Resize` augmentation, csrc/data.hip) on one cfg-2 batch: 64 keyframes x 4 cameras, RGB and XYZ.
usage (GPU box): python profiles/resize_probe.py"""
import importlib
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
a3d = importlib.import_module("act3d-chained-diffuser_amd")
dev = torch.device("cuda:0")
B, N, H = 64, 4, 256
x = torch.rand(B, N, 3, H, H, device=dev)
rs = np.random.RandomState(0)
params = []
for f in range(B):
    sc = rs.uniform(0.75, 1.25)
    rh = int(H * sc)
    i, j = (rs.randint(0, rh - H + 1), rs.randint(0, rh - H + 1)) if rh > H else (0, 0)
    params.append((rh, rh, i, j))
p = torch.tensor(params, dtype=torch.int32, device=dev)
for _ in range(3):
    y = a3d.data.resize_crop(x, p)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    y = a3d.data.resize_crop(x, p)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
byt = 2 * x.numel() * 4
print(json.dumps({"kernel": "a3d_resize_crop", "shape": [B, N, 3, H, H], "ms": ms, "algorithmic_bytes": byt,
                  "GBps": byt / ms / 1e6, "frac_of_8TBps": byt / ms / 1e6 / 8000.0}))
