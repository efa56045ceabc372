"""Deterministic, platform-independent construction of the golden fixtures' inputs.

Fixtures store only seeds + the reference's OUTPUTS; parameters and inputs are regenerated from numpy's MT19937
(`np.random.RandomState`), which is bit-stable across machines, so the same function feeds the reference (when the
goldens are generated), the CPU oracle and the HIP path.
"""
import numpy as np
import torch

PERACT_BOUNDS = np.array([[-0.1101, -0.5558, 0.7129], [0.6481, 0.5184, 1.5116]])
HIVEFORMER_BOUNDS = np.array([[-0.9439, -0.5644, 0.7106], [0.7039, 0.5821, 1.5122]])       # 74 tasks, SURVEY §8d
DIFFUSION_BOUNDS = np.array([[-0.7342, -0.7915, 0.7098], [0.6944, 0.8437, 1.8645]])


def rs_tensor(rs, shape, scale=1.0, kind="normal"):
    if kind == "normal":
        a = rs.standard_normal(size=shape)
    else:
        a = rs.uniform(0.0, 1.0, size=shape)
    return torch.from_numpy((a * scale).astype(np.float32))


def seeded_state_dict(named_shapes, seed, gain=1.0):
    """Fills every parameter (sorted by name) from RandomState(seed).  Tied duplicates must be resolved by the caller
    (pass each distinct tensor once).  Matrices ~ N(0, gain^2/fan_in); LayerNorm weights ~ 1 + 0.1 N; biases and
    embeddings ~ 0.1-0.5 N -- non-zero everywhere so that no code path is hidden by zero-initialised parameters
    (e.g. AdaLN's zero-init modulation, layers.py:279-280)."""
    rs = np.random.RandomState(seed)
    sd = {}
    for name in sorted(named_shapes):
        shape = tuple(named_shapes[name])
        if name.endswith("norm.weight") or ".norm" in name and name.endswith(".weight"):
            t = 1.0 + 0.1 * rs.standard_normal(size=shape)
        elif len(shape) >= 2 and "embed" not in name:
            fan_in = int(np.prod(shape[1:]))
            t = rs.standard_normal(size=shape) * (gain / np.sqrt(fan_in))
        elif "embed" in name:
            t = rs.standard_normal(size=shape) * 0.5
        else:
            t = rs.standard_normal(size=shape) * 0.1
        sd[name] = torch.from_numpy(t.astype(np.float32))
    return sd


def unique_param_shapes(module):
    """name -> shape for trainable, non-backbone, non-FPN parameters, tied parameters listed under every alias but
    mapped to one canonical name (the first alias in sorted order)."""
    by_id, alias = {}, {}
    for name, p in sorted(module.state_dict(keep_vars=True).items()):
        if name.startswith("backbone.") or "feature_pyramid" in name or ".backbone." in name:
            continue
        key = id(p)
        if key not in by_id:
            by_id[key] = name
        alias[name] = by_id[key]
    shapes = {canon: tuple(module.state_dict()[canon].shape) for canon in set(alias.values())}
    return shapes, alias


def expand_aliases(sd_canon, alias):
    return {name: sd_canon[canon] for name, canon in alias.items()}


def keypose_inputs(seed, B, ncam, E, levels, image=256, bounds=PERACT_BOUNDS):
    """Synthetic post-FPN feature maps + point clouds + poses, numpy-seeded.

    Returns feature maps in the reference's layout (B, ncam, E, h, w) and the point cloud (B, ncam, 3, H, W)."""
    rs = np.random.RandomState(seed)
    lo, hi = bounds[0], bounds[1]
    feats = []
    for i in range(levels):
        f = (8 if image == 256 else 4) if i == 0 else 2       # act3d.py:78-87: 128x128 images use res2 @ 1/4
        feats.append(rs_tensor(rs, (B, ncam, E, image // f, image // f)))
        if i >= 1:
            feats[-1] = feats[1]            # levels >= 1 share the res1 map (act3d.py:86)
    # smooth-ish cloud: low-res noise up-sampled + small jitter, inside the workspace
    base = rs.uniform(0.0, 1.0, size=(B, ncam, 3, image // 16, image // 16))
    base = np.kron(base, np.ones((1, 1, 1, 16, 16))) + 0.05 * rs.standard_normal(size=(B, ncam, 3, image, image))
    base = np.clip(base, 0.0, 1.0)
    pcd = lo[None, None, :, None, None] + base * (hi - lo)[None, None, :, None, None]
    pcd = torch.from_numpy(pcd.astype(np.float32))
    shrink = 0.1 * (hi - lo)

    def pose():
        xyz = rs.uniform(lo + shrink, hi - shrink, size=(B, 3))
        q = rs.standard_normal(size=(B, 4))
        q /= np.linalg.norm(q, axis=-1, keepdims=True)
        op = rs.randint(0, 2, size=(B, 1)).astype(np.float64)
        return torch.from_numpy(np.concatenate([xyz, q, op], axis=-1).astype(np.float32))

    curr_gripper, action = pose(), pose()
    instr = rs_tensor(rs, (B, 53, 512))
    return dict(feats=feats, pcd=pcd, curr_gripper=curr_gripper, action=action, instr=instr)


def tokens_from_maps(fmap):
    """(B, ncam, E, h, w) -> (B, ncam*h*w, E): "b ncam c h w -> b (ncam h w) c" (act3d.py:240-248)."""
    B, C, E, h, w = fmap.shape
    return fmap.permute(0, 1, 3, 4, 2).reshape(B, C * h * w, E).contiguous()


def trajectory_inputs(seed, B, L, ncam, E, image=256, bounds=DIFFUSION_BOUNDS, pad_last=0):
    rs = np.random.RandomState(seed)
    lo, hi = bounds[0], bounds[1]
    fmap = rs_tensor(rs, (B, ncam, E, image // 8, image // 8))
    base = rs.uniform(0.0, 1.0, size=(B, ncam, 3, image, image))
    pcd = torch.from_numpy((lo[None, None, :, None, None] + base * (hi - lo)[None, None, :, None, None]).astype(np.float32))
    shrink = 0.15 * (hi - lo)

    def pose7():
        xyz = rs.uniform(lo + shrink, hi - shrink, size=(B, 3))
        q = rs.standard_normal(size=(B, 4))
        q /= np.linalg.norm(q, axis=-1, keepdims=True)
        return np.concatenate([xyz, q], axis=-1)

    cg, gg = pose7(), pose7()
    w = np.linspace(0.0, 1.0, L)[None, :, None]
    traj = cg[:, None] * (1 - w) + gg[:, None] * w + 0.01 * rs.standard_normal(size=(B, L, 7))
    traj[..., 3:] /= np.linalg.norm(traj[..., 3:], axis=-1, keepdims=True)
    mask = np.zeros((B, L), dtype=bool)
    if pad_last:
        mask[B // 2:, L - pad_last:] = True
    instr = rs_tensor(rs, (B, 53, 512))
    f32 = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))
    return dict(fmap=fmap, pcd=pcd, curr_gripper=f32(cg), goal_gripper=f32(gg), trajectory=f32(traj),
                mask=torch.from_numpy(mask), instr=instr,
                noise=rs_tensor(rs, (B, L, 9)), timesteps=torch.from_numpy(rs.randint(0, 100, size=(B,))).long(),
                init_noise=rs_tensor(rs, (B, L, 9)), step_noise=rs_tensor(rs, (100, B, L, 9)))


def fine_feature_map(seed, B, ncam, E, image=256):
    """Seeded stand-in for the FPN's res1 map (stride 2) used by the multi-scale diffusion fixtures."""
    rs = np.random.RandomState(seed + 1000)
    return rs_tensor(rs, (B, ncam, E, image // 2, image // 2))


def metrics_inputs():
    """Seeded predictions / targets for the metric and optional-loss fixtures (shared with the tests)."""
    rs = np.random.RandomState(17)
    B, Ln = 6, 7
    gt = rs_tensor(rs, (B, Ln, 7))
    gt[..., 3:] = gt[..., 3:] / gt[..., 3:].norm(dim=-1, keepdim=True)
    pred = gt + 0.02 * rs_tensor(rs, (B, Ln, 7))
    pred[0] = gt[0] + 0.002 * rs_tensor(rs, (Ln, 7))            # inside the 0.01 / 0.025 thresholds
    pred[1, :, 3:] = -gt[1, :, 3:] + 0.003 * rs_tensor(rs, (Ln, 4))   # the antipodal quaternion: symmetric branch
    action = torch.cat([rs_tensor(rs, (B, 3)), torch.nn.functional.normalize(rs_tensor(rs, (B, 4)), dim=-1),
                        torch.tensor([[1.0], [0.0], [1.0], [1.0], [0.0], [0.0]])], dim=-1)
    kp = {"position_pyramid": [action[:, None, :3] + s * rs_tensor(rs, (B, 1, 3)) for s in (0.05, 0.01, 0.004)],
          "rotation": torch.nn.functional.normalize(action[:, 3:7] + 0.01 * rs_tensor(rs, (B, 4)), dim=-1),
          "gripper": torch.tensor([[0.9], [0.2], [0.4], [0.7], [0.6], [0.1]])}
    kp["rotation"][2] = -kp["rotation"][2]
    kp["position"] = kp["position_pyramid"][-1][:, 0].clone()
    tasks = ["close_jar", "open_drawer", "close_jar", "stack_cups", "open_drawer", "close_jar"]
    return pred, gt, action, kp, tasks


# ---------------------------------------------------------------------------------------------------- data plane fixtures
DATASET_CAMERAS = ("wrist", "left_shoulder")
DATASET_TASKVAR = [("task_a", 0), ("task_b", 0)]
DATASET_IMAGE = 20


def synthetic_episode(seed, T, ncam=2, H=DATASET_IMAGE):
    """One episode in the on-disk layout of data_preprocessing/data_gen.py:122-132:
    [frame_ids, obs (n_cam, 2, 3, H, W) per frame, actions (1, 8), camera dicts, grippers (1, 8), trajectories (N_i, 8)].
    The camera dicts list the cameras in REVERSED order so that the dataset's camera re-mapping is exercised."""
    rs = np.random.RandomState(seed)
    frame_ids = list(range(T))
    obs = [rs_tensor(rs, (ncam, 2, 3, H, H), kind="uniform") * 2 - 1 for _ in range(T)]

    def pose():
        p = rs_tensor(rs, (1, 8))
        p[:, 3:7] = p[:, 3:7] / p[:, 3:7].norm(dim=-1, keepdim=True)
        p[:, 7] = (p[:, 7] > 0).float()
        return p
    actions = [pose() for _ in range(T)]
    cams = [{c: None for c in reversed(DATASET_CAMERAS)} for _ in range(T)]
    grippers = [pose() for _ in range(T)]
    trajs = []
    for _ in range(T):
        n = int(rs.randint(4, 9))
        t = rs_tensor(rs, (n, 8))
        t[:, 3:7] = t[:, 3:7] / t[:, 3:7].norm(dim=-1, keepdim=True)
        t[:, 7] = (t[:, 7] > 0).float()
        trajs.append(t)
    return [frame_ids, obs, actions, cams, grippers, trajs]


def write_synthetic_dataset(root):
    """task_a+0/ep0.npy (7 frames: two chunks of <= 5) and task_b+0/ep0.pkl (3 frames); returns the instruction dict."""
    import os
    import pickle
    os.makedirs(os.path.join(root, "task_a+0"), exist_ok=True)
    os.makedirs(os.path.join(root, "task_b+0"), exist_ok=True)
    ep = synthetic_episode(11, 7)
    arr = np.empty(len(ep), dtype=object)
    for i, e in enumerate(ep):
        arr[i] = e
    np.save(os.path.join(root, "task_a+0", "ep0.npy"), arr, allow_pickle=True)
    with open(os.path.join(root, "task_b+0", "ep0.pkl"), "wb") as f:
        pickle.dump(synthetic_episode(12, 3), f)
    rs = np.random.RandomState(13)
    return {"task_a": {0: rs_tensor(rs, (3, 53, 512))}, "task_b": {0: rs_tensor(rs, (2, 53, 512))}}


def assert_topk_equal_up_to_exact_ties(got, ref, pos, xyz, what=""):
    """k-NN indices `got` vs the reference's `ref` (B, k): bit-exact, except that torch.topk leaves the order AMONG EXACTLY TIED
    distances unspecified (act3d.py:244-245) while the oracle / the HIP kernel order ties by index.  Every differing position must
    therefore hold two points whose fp32 distances sqrt((dx^2 + dy^2) + dz^2) to the centre are the SAME BITS (0 ulp apart), and the
    two index SETS must be equal.  Returns the number of tied positions that differ."""
    got, ref = np.asarray(got), np.asarray(ref)
    pos, xyz = np.asarray(pos, dtype=np.float32).reshape(-1, 3), np.asarray(xyz, dtype=np.float32)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert np.array_equal(np.sort(got, -1), np.sort(ref, -1)), f"{what}: k-NN index sets differ"
    mis = np.argwhere(got != ref)
    for b, j in mis:
        def dist_bits(ix):
            d = xyz[b, ix] - pos[b]
            return np.sqrt(np.float32(np.float32(d[0] * d[0] + d[1] * d[1]) + d[2] * d[2])).astype(np.float32).view(np.int32)
        a, c = dist_bits(got[b, j]), dist_bits(ref[b, j])
        assert a == c, f"{what}: position ({b}, {j}) differs and the two distances are {abs(int(a) - int(c))} ulp apart (not a tie)"
    return len(mis)
