"""Debug probe for tests/test_diffusion_gpu.py::test_cfg3_full_shape_graph_vs_oracle: which of {eager, eager again, graph capture
call, graph replay call} differ, and whether the step-invariant context (encode_context / build_fused tensors) is run-to-run
deterministic.  usage (GPU box): python profiles/cfg3_graph_probe.py"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import common as C  # noqa: E402
import test_diffusion_gpu as T  # noqa: E402

a3d = importlib.import_module("act3d-chained-diffuser_amd")
dev = torch.device("cuda:0")
r = T.load("diffusion.pt")
E, B, Ln, ncam, H = 120, 64, 16, 3, 8
m = a3d.DiffusionPlanner(embedding_dim=E, output_dim=7, num_vis_ins_attn_layers=2, num_query_cross_attn_layers=6,
                         use_instruction=True, use_goal=True, use_goal_at_test=True, weight_tying=True,
                         gripper_loc_bounds=C.DIFFUSION_BOUNDS, rotation_parametrization="6D", diffusion_timesteps=100)
m.load_state_dict(T._diffusion_params(r), strict=False)
m.to(dev).eval()
inp = C.trajectory_inputs(91, B, Ln, ncam, E, pad_last=3)
tokens = C.tokens_from_maps(inp["fmap"]).to(dev)
d = {k: v.to(dev) for k, v in inp.items()}


def run(**kw):
    return m.compute_trajectory(d["mask"], None, d["pcd"], d["instr"], d["curr_gripper"], d["goal_gripper"], init_noise=d["init_noise"],
                                step_noise=d["step_noise"], visual_tokens=tokens, **kw).cpu()


def diff(a, b):
    return (a - b).abs().max().item()


# context determinism
head = m.prediction_head
with torch.no_grad():
    ctxs = []
    for _ in range(3):
        tk, cx, cg, gg = m._prepare(None, d["pcd"], d["curr_gripper"], d["goal_gripper"], tokens)
        ctx, cxyz, instr = head.encode_context(tk, cx, d["instr"], cg, gg)
        st = head.build_fused(ctx, cxyz, instr, d["mask"].to(torch.uint8).contiguous(), m.tables(dev) and m._time_tables["sin"], Ln)
        torch.cuda.synchronize()
        ctxs.append((ctx.clone(), instr.clone(), [t_.clone() for t_ in st["tensors"]]))
    for i in (1, 2):
        print("context run", i, "vs 0: ctx", diff(ctxs[i][0], ctxs[0][0]), "instr", diff(ctxs[i][1], ctxs[0][1]),
              "fused tensors", max(diff(a.float(), b.float()) for a, b in zip(ctxs[i][2], ctxs[0][2])))
def abort_word():
    ps = getattr(m.prediction_head, "_last_persist", None)
    return None if ps is None else int(ps["sync"][2].item())


for n in (1, 5, 30, 100):
    print(f"-- n_steps={n}: eager", flush=True)
    e1 = run(n_steps=n)
    print("   abort word", abort_word(), flush=True)
    e2 = run(n_steps=n)
    m._graph = None
    print(f"-- n_steps={n}: graph", flush=True)
    g1 = run(n_steps=n, use_graph=True)
    print("   captured + replayed once; abort word", abort_word(), flush=True)
    g2, g3 = run(n_steps=n, use_graph=True), run(n_steps=n, use_graph=True)
    print(f"n_steps={n}: eager-eager {diff(e1, e2):.3e}  graph1-eager {diff(g1, e1):.3e}  graph2-eager {diff(g2, e1):.3e}  graph3-eager {diff(g3, e1):.3e}  "
          f"graph1-graph2 {diff(g1, g2):.3e}")
