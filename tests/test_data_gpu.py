"""GPU tests of the data plane and the train / evaluate drivers:
  * a3d_resize_crop (the `Resize` augmentation of datasets/utils.py:40-100 as one gather kernel) against the oracle's index
    map -- bit-exact: the operation only moves values;
  * DeviceLoader over the product dataset == the batch the REFERENCE's dataset + collate produced (tests/golden/dataset.pt);
  * evaluate_nsteps of both drivers == the mean over batches of the criterion's metrics on free-running forwards;
  * KeyposeTrainTester.main end to end on synthetic episode files: loaders -> training steps -> evaluation -> checkpoints.
"""
import os
import pickle
import random
import sys
import types

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import common as C  # noqa: E402
from oracle import data as OD  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("H,scale,shift", [(256, 1.0, 0.0), (128, 0.5, 0.5), (20, 1.0, 0.0)])
def test_resize_crop_kernel_bit_exact(a3d, dev, H, scale, shift):
    rs = np.random.RandomState(H)
    F_, N = 5, 2
    x = torch.from_numpy(rs.uniform(-1, 1, size=(F_, N, 3, H, H)).astype(np.float32))
    params = []
    for f in range(F_):
        np.random.seed(100 + f)
        torch.manual_seed(100 + f)
        params.append(OD.resize_params((0.75, 1.25), H, H))
    params[0] = (H, H, 0, 0)                                        # identity frame
    p = torch.tensor(params, dtype=torch.int32)
    got = a3d.data.resize_crop(x.to(dev), p.to(dev), scale, shift).cpu().numpy()
    for f in range(F_):
        ref = OD.resize_crop(x[f].numpy(), *params[f])
        ref = (ref * np.float32(scale) + np.float32(shift)).astype(np.float32)
        assert np.array_equal(got[f], ref), (f, params[f])
    assert {pp[0] < H for pp in params[1:]} == {True, False}, "both the pad and the crop branch must be covered"


@pytest.mark.parametrize("tag", ["train_traj", "eval_traj", "train_keypose"])
def test_device_loader_equals_reference_batches(a3d, dev, tag, tmp_path):
    r = torch.load(os.path.join(HERE, "golden", "dataset.pt"), weights_only=False)[tag]
    instr = C.write_synthetic_dataset(str(tmp_path))
    training, traj = tag.startswith("train"), tag.endswith("traj")
    random.seed(5)
    np.random.seed(5)
    torch.manual_seed(5)
    ds = a3d.data.RLBenchDataset(root=str(tmp_path), instructions=instr, taskvar=C.DATASET_TASKVAR, max_episode_length=5,
                                 cache_size=0, max_episodes_per_task=100, cameras=C.DATASET_CAMERAS, training=training,
                                 gripper_loc_bounds=C.PERACT_BOUNDS, image_rescale=(0.75, 1.25),
                                 point_cloud_rotate_yaw_range=0.0, return_low_lvl_trajectory=traj, dense_interpolation=traj,
                                 interpolation_length=12, action_dim=8, predict_short=False)
    collate = a3d.data.traj_collate_fn if traj else a3d.data.keypose_collate_fn

    class Five(torch.utils.data.Sampler):                  # items 0..4 in order, as the golden generator drew them
        def __iter__(self):
            return iter(range(5))

        def __len__(self):
            return 5
    # own generator: the DataLoader iterator draws its base seed from it instead of from the global torch RNG, whose stream
    # the dataset's crop offsets must see exactly as the reference's direct ds[i] calls did
    loader = torch.utils.data.DataLoader(ds, batch_size=5, sampler=Five(), num_workers=0, collate_fn=collate, pin_memory=True,
                                         generator=torch.Generator().manual_seed(0))
    batches = list(a3d.data.DeviceLoader(loader, dev))
    assert len(batches) == 1
    b = batches[0]
    torch.cuda.synchronize()
    assert "resize_params" not in b and b["task"] == r["task"]
    for k in ["rgbs", "pcds", "curr_gripper", "action"] + (["trajectory", "trajectory_mask"] if traj else []):
        assert b[k].is_cuda and b[k].dtype == r[k].dtype, k
        assert torch.equal(b[k].cpu(), r[k]), k
    assert torch.equal(b["instr"][:, ::13, ::64].cpu(), r["instr_sample"])


def _keypose_batches(dev, n, B=3):
    out = []
    for i in range(n):
        inp = C.keypose_inputs(40 + i, B, 1, 60, 2, image=128)
        rs = np.random.RandomState(90 + i)
        s = {"rgbs": torch.from_numpy(rs.uniform(0, 1, size=(B, 1, 3, 128, 128)).astype(np.float32)), "pcds": inp["pcd"],
             "instr": inp["instr"], "curr_gripper": inp["curr_gripper"], "action": inp["action"]}
        s = {k: v.to(dev) for k, v in s.items()}
        s["task"] = ["task_a", "task_b", "task_a"][:B]
        out.append(s)
    return out


def test_keypose_evaluate_nsteps(a3d, dev):
    """main_keypose.py:236-281: mean over the first val_iters batches of compute_metrics on gt-free forwards.  The forwards
    themselves are recorded with a hook (a second free-running pass of an untrained model is not comparable: its near-tied
    mask logits turn 1e-6 differences between MIOpen's convolution algorithms into different argmax points)."""
    torch.manual_seed(0)
    m = a3d.Act3D(image_size=(128, 128), gripper_loc_bounds=C.PERACT_BOUNDS, num_ghost_points=200, num_ghost_points_val=400,
                  num_sampling_level=2, sampler_seed=3).to(dev)
    crit = a3d.LossAndMetrics(position_loss="ce", rotation_parametrization="quat_from_query", ground_truth_gaussian_spread=0.01)
    tt = a3d.KeyposeTrainTester(types.SimpleNamespace(log_dir=None))
    batches = _keypose_batches(dev, 3)
    seen = []
    m.register_forward_hook(lambda mod, args, kwargs, out: seen.append((kwargs, out)), with_kwargs=True)
    m.train()
    ret = tt.evaluate_nsteps(m, crit, batches, step_id=7, val_iters=2, split="val")
    assert ret is None                       # the reference looks up 'val-losses/action_mse', which no metric is called
    assert not m.training and len(seen) == 2
    acc = {}
    for (kwargs, out), s in zip(seen, batches):
        assert kwargs.get("gt_action", "missing") is None           # no ground-truth anchor at validation time
        assert out["ghost_pcd_pyramid"][0].shape[-1] == 200         # num_ghost_points_val // levels
        assert not out["position"].requires_grad
        for k, v in crit.compute_metrics(out, s).items():
            acc.setdefault(f"val-losses/{k}", []).append(float(v))
    got = {k: v[0] for k, v in tt.scalars.items()}
    assert set(got) == set(acc) and "val-losses/task_a/pos_l2_final" in got and "val-losses/mean/rot_l1" in got
    for k, vs in acc.items():
        assert abs(got[k] - float(np.mean(np.float32(vs)))) <= 1e-6 + 1e-6 * abs(got[k]), k
    assert all(step == 7 for _, step in tt.scalars.values())


def test_trajectory_evaluate_nsteps(a3d, dev):
    """main_trajectory.py:206-274: sampling (run_inference=True) per batch, summary + per-task metrics."""
    import bench_denoise as BD
    torch.manual_seed(0)
    m = BD.build_planner(a3d, dev, train=True)
    crit = a3d.TrajectoryCriterion()
    batches = []
    for i in range(2):
        s = BD.synthetic_inputs(2, 8, 1, dev, seed=5 + i)
        s["task"] = ["task_a", "task_b"]
        batches.append(s)
    tt = a3d.TrajectoryTrainTester(types.SimpleNamespace(log_dir=None))
    seen = []
    m.register_forward_hook(lambda mod, args, kwargs, out: seen.append((kwargs, out)), with_kwargs=True)
    ret = tt.evaluate_nsteps(m, crit, batches, step_id=3, val_iters=5, split="val")
    assert not m.training and len(seen) == 2
    acc = {}
    for (kwargs, traj), s in zip(seen, batches):
        assert kwargs.get("run_inference") is True and traj.shape == s["trajectory"].shape
        summ, per = crit.compute_metrics(traj, s["trajectory"], s["trajectory_mask"])
        for k, v in summ.items():
            acc.setdefault(f"val-losses/{k}", []).append(float(v))
        for k, v in per.items():
            for j, t in enumerate(s["task"]):
                acc.setdefault(f"val-loss/{t}/{k}", []).append(float(v[j]))
    got = {k: v[0] for k, v in tt.scalars.items()}
    assert set(got) == set(acc)
    for k, vs in acc.items():
        assert abs(got[k] - float(np.mean(np.float32(vs)))) <= 1e-5 + 1e-5 * abs(got[k]), k
    assert abs(ret - got["val-losses/traj_action_mse"]) < 1e-12


def test_keypose_train_tester_main_end_to_end(a3d, dev, tmp_path):
    """engine.py:104-181 on synthetic episode files: DistributedSampler loaders -> DeviceLoader (GPU augmentation) ->
    train_one_step x 2 -> evaluate_nsteps (train + val split) -> best.pth / last.pth in the reference's layout -> resume."""
    root = tmp_path / "data"
    for task, seed, T in (("task_a", 31, 4), ("task_b", 32, 3)):
        d = root / f"{task}+0"
        d.mkdir(parents=True)
        for e in range(2):
            ep = C.synthetic_episode(seed + 10 * e, T, ncam=2, H=128)
            # world-frame clouds / poses inside the workspace (the synthetic generator draws N(0, 1))
            lo, hi = torch.tensor(C.PERACT_BOUNDS[0]).float(), torch.tensor(C.PERACT_BOUNDS[1]).float()
            for t in range(T):
                ep[1][t][:, 1] = lo.view(1, 3, 1, 1) + (ep[1][t][:, 1] * 0.5 + 0.5) * (hi - lo).view(1, 3, 1, 1)
                for lst in (ep[2], ep[4]):
                    lst[t][:, :3] = lo + torch.sigmoid(lst[t][:, :3]) * (hi - lo)
            with open(d / f"ep{e}.pkl", "wb") as f:
                pickle.dump(ep, f)
    rs = np.random.RandomState(3)
    with open(tmp_path / "instructions.pkl", "wb") as f:
        pickle.dump({"task_a": {0: C.rs_tensor(rs, (2, 53, 512))}, "task_b": {0: C.rs_tensor(rs, (2, 53, 512))}}, f)
    log_dir = tmp_path / "logs"
    log_dir.mkdir()
    args = types.SimpleNamespace(
        local_rank=0, cameras=C.DATASET_CAMERAS, image_size="128,128", max_episodes_per_task=100,
        instructions=str(tmp_path / "instructions.pkl"), seed=0, tasks=("task_a", "task_b"), variations=(0,), checkpoint=None,
        accumulate_grad_batches=1, val_freq=2, gripper_loc_bounds=C.PERACT_BOUNDS, eval_only=0, dataset=str(root),
        valset=str(root), log_dir=log_dir, num_workers=0, batch_size=2, batch_size_val=2, cache_size=0, cache_size_val=0,
        lr=1e-4, train_iters=2, max_episode_length=5, image_rescale="0.75,1.25", point_cloud_rotate_yaw_range=0.0,
        position_prediction_only=0, position_loss="ce", ground_truth_gaussian_spread=0.01, compute_loss_at_all_layers=0,
        position_loss_coeff=1.0, position_offset_loss_coeff=10000.0, rotation_loss_coeff=10.0, symmetric_rotation_loss=0,
        gripper_loss_coeff=1.0, label_smoothing=0.0, regress_position_offset=0, num_sampling_level=2,
        fine_sampling_ball_diameter=0.16, weight_tying=1, gp_emb_tying=1, num_ghost_points=200, num_ghost_points_val=400,
        use_ground_truth_position_for_sampling_train=1, backbone="clip", embedding_dim=60,
        num_ghost_point_cross_attn_layers=2, num_query_cross_attn_layers=2, num_vis_ins_attn_layers=2,
        rotation_parametrization="quat_from_query", use_instruction=1)
    random.seed(0)
    np.random.seed(0)
    torch.manual_seed(0)
    tt = a3d.KeyposeTrainTester(args)
    model = tt.main(collate_fn=a3d.data.keypose_collate_fn)
    torch.cuda.synchronize()
    sc = tt.scalars
    assert "train-loss/noise_mse" in sc and np.isfinite(sc["train-loss/noise_mse"][0])
    assert "train-losses/mean/pos_l2_final" in sc and "val-losses/mean/pos_l2_final" in sc
    assert all(np.isfinite(v) for v, _ in sc.values())
    for name in ("best.pth", "last.pth"):
        ck = torch.load(log_dir / name, map_location="cpu", weights_only=False)
        assert ck["iter"] == 2 and set(ck) == {"weight", "optimizer", "iter", "best_loss"}
        assert all(k.startswith("module.") for k in ck["weight"])
        assert set(ck["optimizer"]) == {"state", "param_groups"}
    # resume from last.pth (engine.py:195-212) and evaluate only
    args.checkpoint, args.eval_only = str(log_dir / "last.pth"), 1
    tt2 = a3d.KeyposeTrainTester(args)
    m2 = tt2.main(collate_fn=a3d.data.keypose_collate_fn)
    for (n1, p1), (n2, p2) in zip(model.named_parameters(), m2.named_parameters()):
        if not n1.startswith("backbone"):
            assert n1 == n2 and torch.equal(p1, p2), n1
    assert "val-losses/mean/pos_l2_final" in tt2.scalars


# ---------------------------------------------------------------------------------------------- reference-produced harness records
def _harness():
    return torch.load(os.path.join(HERE, "golden", "harness.pt"), weights_only=False)


def _ref_seeded_act3d(a3d, dev, hk, train=False, **kw):
    from test_act3d_gpu import build_model
    from test_oracle_golden import _act3d_case, act3d_params
    names = _act3d_case("train_L3_C1_N64")[2]
    cfg = dict(hk["cfg"], use_instruction=False, image=256)
    P = act3d_params(cfg, hk["seed"], hk["gain"], names)
    return build_model(a3d, dev, cfg, P, cfg["Ng"], train, **kw), cfg


def test_keypose_evaluate_nsteps_reproduces_the_reference_harness_record(a3d, dev):
    """tests/golden/harness.pt["keypose"] is what the REFERENCE's main_keypose.TrainTester.evaluate_nsteps (main_keypose.py:236-281)
    logged and returned for two seeded batches through the reference Act3D (numpy ghost sampler, injected FPN outputs).  The
    product driver on the same batches, the same parameters and the same numpy seed must log the same keys with the same
    values at the same step, and return the same thing (None: the reference looks up a key no keypose metric has)."""
    hk = _harness()["keypose"]
    m, cfg = _ref_seeded_act3d(a3d, dev, hk, ghost_sampler="numpy")
    B, ncam, E, levels = cfg["B"], cfg["ncam"], cfg["E"], cfg["levels"]
    batches, feats = [], []
    for s_ in hk["batch_seeds"]:
        inp = C.keypose_inputs(s_, B, ncam, E, levels)
        maps = [inp["feats"][0]] + [inp["feats"][1]] * (levels - 1)
        feats.append([C.tokens_from_maps(f.to(dev)) for f in maps])
        batches.append({"rgbs": torch.zeros(B, ncam, 3, 8, 8, device=dev), "pcds": inp["pcd"].to(dev), "instr": inp["instr"].to(dev),
                        "curr_gripper": inp["curr_gripper"].to(dev), "action": inp["action"].to(dev), "task": ["task_a", "task_b"]})
    state = {"j": 0}
    m.compute_visual_tokens = lambda rgb: feats[state["j"]]
    m.register_forward_hook(lambda mod, args, out: state.__setitem__("j", state["j"] + 1))
    crit = a3d.LossAndMetrics(position_loss="ce", rotation_parametrization="quat_from_query", ground_truth_gaussian_spread=0.01)
    tt = a3d.KeyposeTrainTester(types.SimpleNamespace(log_dir=None))
    np.random.seed(hk["np_seed"])
    ret = tt.evaluate_nsteps(m, crit, batches, step_id=hk["step_id"], val_iters=5, split="val")
    assert ret is hk["returned"] is None and state["j"] == 2
    got = tt.scalars
    assert set(got) == set(hk["scalars"]), sorted(set(got) ^ set(hk["scalars"]))
    worst = 0.0
    for k, (val, step) in hk["scalars"].items():
        assert got[k][1] == step == hk["step_id"], k
        err = abs(got[k][0] - val)
        worst = max(worst, err / max(1.0, abs(val)))
        assert err <= 1e-3 * max(1.0, abs(val)), (k, got[k][0], val)
    print(f"[parity] keypose evaluate_nsteps vs the reference harness: {len(got)} scalars, worst error {worst:.2e} of scale")


def test_trajectory_evaluate_nsteps_reproduces_the_reference_harness_record(a3d, dev):
    """harness.pt["trajectory"]: the reference's main_trajectory.TrainTester.evaluate_nsteps (main_trajectory.py:206-274) over two
    batches with two task names -- 100-step sampling with injected noise, summary + per-task keys, the returned
    'val-losses/traj_action_mse'.  (Its tensorboard trajectory plot is out of scope.)"""
    from test_oracle_golden import _diffusion_params, load
    ht = _harness()["trajectory"]
    cfg = ht["cfg"]
    r = load("diffusion.pt")
    assert (ht["seed"], ht["gain"]) == (r["seed"], r["gain"])          # the same seeded parameters as the diffusion goldens
    m = a3d.DiffusionPlanner(embedding_dim=cfg["E"], output_dim=7, num_vis_ins_attn_layers=2, num_query_cross_attn_layers=6,
                             use_instruction=True, use_goal=True, use_goal_at_test=True, weight_tying=True,
                             gripper_loc_bounds=C.DIFFUSION_BOUNDS, rotation_parametrization="6D", diffusion_timesteps=100, dropout=0.0)
    assert not m.load_state_dict(_diffusion_params(r), strict=False).unexpected_keys
    m.to(dev)
    batches, inject = [], []
    for s_ in ht["batch_seeds"]:
        inp = {k: v.to(dev) for k, v in C.trajectory_inputs(s_, cfg["B"], cfg["L"], cfg["ncam"], cfg["E"], pad_last=cfg["pad_last"]).items()}
        batches.append({"trajectory": inp["trajectory"], "trajectory_mask": inp["mask"], "rgbs": torch.zeros(cfg["B"], cfg["ncam"], 3, 8, 8, device=dev),
                        "pcds": inp["pcd"], "instr": inp["instr"], "curr_gripper": inp["curr_gripper"], "action": inp["goal_gripper"],
                        "task": ["task_a", "task_b"]})
        inject.append(dict(visual_tokens=C.tokens_from_maps(inp["fmap"]), init_noise=inp["init_noise"], step_noise=inp["step_noise"]))
    state = {"j": 0}
    sample_loop = m.compute_trajectory

    def injected(*a, **k):
        k.update(inject[state["j"]])
        state["j"] += 1
        return sample_loop(*a, **k)
    m.compute_trajectory = injected
    tt = a3d.TrajectoryTrainTester(types.SimpleNamespace(log_dir=None))
    ret = tt.evaluate_nsteps(m, a3d.TrajectoryCriterion(), batches, step_id=ht["step_id"], val_iters=5, split="val")
    got = tt.scalars
    assert state["j"] == 2 and set(got) == set(ht["scalars"]), sorted(set(got) ^ set(ht["scalars"]))
    worst = 0.0
    for k, (val, step) in ht["scalars"].items():
        assert got[k][1] == step, k
        tol = 0.13 if "acc" in k else 2e-3 * max(1.0, abs(val))      # accuracies: one of 8-16 binary outcomes may sit on the threshold
        err = abs(got[k][0] - val)
        worst = max(worst, 0.0 if "acc" in k else err / max(1.0, abs(val)))
        assert err <= tol, (k, got[k][0], val)
    assert abs(ret - ht["returned"]) <= 2e-3 * max(1.0, abs(ht["returned"]))
    print(f"[parity] trajectory evaluate_nsteps vs the reference harness: {len(got)} scalars, worst error {worst:.2e} of scale")


def test_checkpoint_written_here_matches_the_one_the_reference_engine_wrote(a3d, dev, tmp_path):
    """harness.pt["checkpoint"] describes last.pth as the REFERENCE's engine wrote it (BaseTrainTester.get_optimizer + one
    torch.optim.AdamW step + save_checkpoint, engine.py:89-102,214-230) for a seeded Act3D and seeded gradients: file names, keys,
    'iter' / 'best_loss', the two parameter groups (options and membership), which parameters carry optimizer state, every
    hot-path tensor's (sum, abs-sum) of weight / exp_avg / exp_avg_sq and two whole tensors.  The product, driven the same
    way (same parameters, same gradients, its own fused AdamW step, KeyposeTrainTester.save_checkpoint), must write the
    same thing; FPN / backbone VALUES are not compared (the reference model draws them from the unseeded torch RNG)."""
    hc = _harness()["checkpoint"]
    m, _ = _ref_seeded_act3d(a3d, dev, dict(cfg=dict(hc["model"], B=2), seed=hc["model"]["seed"], gain=hc["model"]["gain"]), train=True)
    named = [(n, p) for n, p in m.named_parameters() if not n.startswith("backbone")]
    assert [n for n, _ in named] == hc["named_parameters"], "parameter names / order differ from the reference model's"
    flat, opt = a3d.engine.get_optimizer(m, lr=1e-4)
    g = torch.Generator().manual_seed(hc["grad_seed"])
    flat.zero_grad()
    for n, p in named:
        if p.requires_grad and n not in hc["unused"]:
            p.grad.copy_((torch.randn(p.shape, generator=g) * hc["grad_scale"]).to(dev))
    opt.step()
    tt = a3d.KeyposeTrainTester(types.SimpleNamespace(log_dir=str(tmp_path)))
    best = tt.save_checkpoint(m, opt, 4, None, None)
    assert best is hc["returned_best"] is None and sorted(os.listdir(tmp_path)) == hc["files"]
    ck = torch.load(os.path.join(tmp_path, "last.pth"), map_location="cpu", weights_only=False)
    assert sorted(ck.keys()) == hc["keys"] and ck["iter"] == hc["iter"] and ck["best_loss"] == hc["best_loss"]
    weight = {(k[7:] if k.startswith("module.") else k): v for k, v in ck["weight"].items()}       # DDP's prefix (engine.py:121-124)
    assert [k for k in weight if not k.startswith("backbone")] == hc["weight_keys"]
    chk = lambda t: (float(t.double().sum()), float(t.double().abs().sum()))
    close = lambda a, b: abs(a[0] - b[0]) <= 2e-5 * max(1e-6, b[1]) and abs(a[1] - b[1]) <= 2e-5 * max(1e-6, b[1])
    for k, ref in hc["weight_checksums"].items():
        assert close(chk(weight[k]), ref), ("weight", k, chk(weight[k]), ref)
    for k, t in hc["samples"].items():
        assert torch.allclose(weight[k], t, rtol=0, atol=2e-6), k
    groups = ck["optimizer"]["param_groups"]
    for gp, ref in zip(groups, hc["param_group_options"]):
        for key in ("weight_decay", "lr", "betas", "eps", "amsgrad"):
            assert gp[key] == ref[key] or tuple(gp[key]) == tuple(ref[key]), (key, gp[key], ref[key])
    index_name = [n for grp in flat.torch_groups for n in grp]
    mine = [[index_name[i] for i in gp["params"] if not index_name[i].startswith("backbone")] for gp in groups]
    assert mine == hc["group_names"], "AdamW group membership / order differs from the reference optimizer's"
    state = {index_name[int(i)]: st for i, st in ck["optimizer"]["state"].items()}
    assert set(state) == set(hc["state_by_name"]), sorted(set(state) ^ set(hc["state_by_name"]))[:6]
    for n, ref in hc["state_by_name"].items():
        assert float(state[n]["step"]) == ref["step"] == 1.0, n
        if "feature_pyramid" in n:
            continue
        assert close(chk(state[n]["exp_avg"]), ref["exp_avg"]) and close(chk(state[n]["exp_avg_sq"]), ref["exp_avg_sq"]), n
    # and the file loads back into a fresh product model + optimizer through the resume path
    m2, _ = _ref_seeded_act3d(a3d, dev, dict(cfg=dict(hc["model"], B=2), seed=7, gain=1.0), train=True)
    flat2, opt2 = a3d.engine.get_optimizer(m2, lr=1e-4)
    it, bl = a3d.engine.load_checkpoint(os.path.join(tmp_path, "last.pth"), m2, opt2)
    assert it == hc["iter"] and bl is None and torch.equal(flat2.flat, flat.flat) and torch.equal(opt2.exp_avg, opt.exp_avg)
    assert torch.equal(opt2.param_steps, opt.param_steps)
    print(f"[parity] checkpoint vs the reference engine's: {len(hc['weight_checksums'])} weight and {len(state)} optimizer-state entries agree")
