#!/usr/bin/env python3
"""Probe: does MIOpen run the FPN's 3x3 E->E convolution (N=256 images, 128x128, bf16 NHWC, fwd + dgrad + wgrad) faster
with the channel count padded from 60 to 64?  Prints ms per direction for both widths (find mode on)."""
import sys
import time

import torch
import torch.nn.functional as F

torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for C in (60, 64):
    x = torch.randn(N, C, 128, 128, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_()
    w = torch.randn(C, C, 3, 3, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_()
    b = torch.zeros(C, device=dev, dtype=torch.bfloat16, requires_grad=True)
    dy = torch.randn(N, C, 128, 128, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)

    def fwd():
        return F.conv2d(x, w, b, padding=1)

    def both():
        y = fwd()
        y.backward(dy)
        x.grad = w.grad = b.grad = None

    for fn, name in ((fwd, "fwd"), (both, "fwd+bwd")):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        print(f"C={C} {name}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms")
