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
