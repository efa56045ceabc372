#!/usr/bin/env python3
"""Secondary benchmark (BASELINE.json configs[2]): ChainedDiffuser DDPM sampling, 100 denoise steps, horizon 16,
batch 64, hipGraph-captured, on ONE MI355X.  Not the driver's bench.py contract; prints one JSON line with
trajectories/s and the K/V-cache streaming roofline of the cross-attention layers (HBM-bound).

  python bench_denoise.py [--batch 64] [--horizon 16] [--cams 3] [--reps 5] [--no-graph]
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
DIFFUSION_BOUNDS = np.array([[-0.7342, -0.7915, 0.7098], [0.6944, 0.8437, 1.8645]])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--horizon", type=int, default=16)
    ap.add_argument("--cams", type=int, default=3)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--no-graph", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    a3d = importlib.import_module("act3d-chained-diffuser_amd")
    a3d.lib.load()
    torch.manual_seed(0)
    B, Ln, C, E, H = args.batch, args.horizon, args.cams, 120, 8
    m = a3d.DiffusionPlanner(embedding_dim=E, output_dim=7, num_vis_ins_attn_layers=2, num_query_cross_attn_layers=6,
                             use_instruction=True, use_goal=True, use_goal_at_test=True, weight_tying=True,
                             gripper_loc_bounds=DIFFUSION_BOUNDS, rotation_parametrization="6D", diffusion_timesteps=100).to(dev)
    for mod in m.modules():                      # AdaLN is zero-initialised in the reference; give it non-trivial weights
        if isinstance(mod, a3d.nn.AdaLN):
            torch.nn.init.normal_(mod.modulation[1].weight, std=0.02)
    m.eval()
    m.prediction_head.backbone_dtype = torch.bfloat16
    g = torch.Generator().manual_seed(1)
    lo, hi = torch.tensor(DIFFUSION_BOUNDS[0], dtype=torch.float32), torch.tensor(DIFFUSION_BOUNDS[1], dtype=torch.float32)
    rgb = torch.rand(B, C, 3, 256, 256, generator=g).to(dev)
    pcd = (lo.view(1, 1, 3, 1, 1) + torch.rand(B, C, 3, 256, 256, generator=g) * (hi - lo).view(1, 1, 3, 1, 1)).to(dev)

    def pose():
        q = torch.randn(B, 4, generator=g)
        return torch.cat([lo + 0.15 * (hi - lo) + torch.rand(B, 3, generator=g) * 0.7 * (hi - lo), q / q.norm(dim=-1, keepdim=True)], -1).to(dev)

    cg, gg = pose(), pose()
    instr = torch.randn(B, 53, 512, generator=g).to(dev)
    mask = torch.zeros(B, Ln, dtype=torch.bool, device=dev)
    with torch.no_grad():
        tokens = m.prediction_head.encode_images(rgb, None).contiguous()      # one-off per trajectory batch (adjacent)
    kw = dict(visual_tokens=tokens, use_graph=not args.no_graph)

    def run():
        return m.compute_trajectory(mask, None, pcd, instr, cg, gg, init_noise=torch.randn(B, Ln, 9, device=dev),
                                    step_noise=torch.randn(100, B, Ln, 9, device=dev), **kw)

    out = run()
    out = run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        out = run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.reps
    assert torch.isfinite(out).all()
    S = C * 1024 + 2
    Sp = (S + 63) // 64 * 64
    # K/V cache actually streamed per denoise step: 8 cross-attention layers x (K: 96 B (hi|lo|lo2) + V: 64 B per key and head)
    KV_BYTES = 2 * a3d.ops.QKW + 64
    bytes_step = 8 * B * H * Sp * KV_BYTES
    flops_step = 8 * 4.0 * Ln * S * E * B                  # QK^T + PV of the 8 cross-attention layers
    # live timing of one cross-attention core launch at these shapes
    O = a3d.ops
    Lqp = 64
    Qs = torch.randn(B, H, Lqp, O.QKW, device=dev).to(torch.bfloat16)
    Ks = torch.randn(B, H, Sp, O.QKW, device=dev).to(torch.bfloat16)
    Vt = torch.randn(B, H, 2, 16, Sp, device=dev).to(torch.bfloat16)
    ns = O.pick_nsplit(B, H, Lqp, Sp)
    for _ in range(3):
        O.attn_core_fwd(Qs, Ks, Vt, None, B, H, Ln, Lqp, S, Sp, ns)
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    st.record()
    for _ in range(20):
        O.attn_core_fwd(Qs, Ks, Vt, None, B, H, Ln, Lqp, S, Sp, ns)
    en.record()
    torch.cuda.synchronize()
    t_attn = st.elapsed_time(en) / 20 * 1e-3
    kv_bytes_launch = B * H * Sp * KV_BYTES
    res = {
        "metric": "DDPM trajectory sampling, 100 denoise steps (trajectories/s)", "value": B / dt, "unit": "trajectories/s",
        "n_gpus": 1, "ms_per_100_step_batch": dt * 1e3, "ms_per_denoise_step": dt * 10, "higher_is_better": True,
        "dtype": "bf16 MFMA on split operands (q,k hi+lo+lo2; p,v hi+lo) attention, fp32 elsewhere", "data": "synthetic",
        "config": {"workload": f"ChainedDiffuser compute_trajectory: B={B}, horizon={Ln}, {C} cameras (S={S} context tokens), "
                               "E=120, H=8, 100 steps, context + K/V cache built once, loop hipGraph-captured"
                               if not args.no_graph else "eager loop", "hipgraph": not args.no_graph},
        "roofline": {"bound": "hbm", "kernel": "attn_fwd (cross-attention against the K/V cache)",
                     "achieved": kv_bytes_launch / t_attn / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": kv_bytes_launch / t_attn / 1e9 / 8000.0, "ms": t_attn * 1e3, "traffic": None,
                     "kv_cache_bytes_per_step": bytes_step, "attn_flops_per_step": flops_step},
    }
    print(json.dumps(res))


if __name__ == "__main__":
    main()
