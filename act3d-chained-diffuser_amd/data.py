"""Data plane: episode files -> collated device batches.

Mirrors `datasets/dataset_engine.py:14-258` (RLBenchDataset), `datasets/utils.py` (loader, Resize, Rotate,
TrajectoryInterpolator) and the collate functions of `main_keypose.py:284-292` / `main_trajectory.py:277-292`, re-cut for
the MI355X: DataLoader workers only decode (pickle / blosc) and slice the episode; the per-pixel work -- the `Resize`
augmentation (nearest resize by a random scale, reflect pad, random crop, shared between RGB and XYZ) -- is ONE gather
kernel over the whole collated batch after the pinned host->device copy (`a3d_resize_crop`), issued on a copy stream
one batch ahead of the training step (`DeviceLoader`).  The random draws are made in the worker with exactly the
reference's RNG consumption (`sample_resize_params`) and travel with the batch as four integers per frame.
"""
import itertools
import math
import pickle
import random
from collections import Counter, defaultdict
from pathlib import Path
from pickle import UnpicklingError
from time import time

import numpy as np
import torch
from torch.utils.data import Dataset

from . import ops as O


def loader(file):
    """datasets/utils.py:16-37: .npy (pickled object array), .dat (blosc-compressed pickle), .pkl."""
    name = str(file)
    try:
        if name.endswith(".npy"):
            return np.load(file, allow_pickle=True)
        if name.endswith(".dat"):
            try:
                import blosc
            except ImportError as e:
                raise RuntimeError(f"{file}: .dat episodes are blosc-compressed and the `blosc` module is not installed") from e
            with open(file, "rb") as f:
                return pickle.loads(blosc.decompress(f.read()))
        if name.endswith(".pkl"):
            with open(file, "rb") as f:
                return pickle.load(f)
    except UnpicklingError as e:
        print(f"Can't load {file}: {e}")
    return None


def sample_resize_params(scales, raw_h, raw_w):
    """The draws of Resize.__call__ (datasets/utils.py:60-62, 89-92): the scale from numpy's global RNG, then the crop offsets
    i, j from torch's (none when the padded image already has the output size).  Returns (rh, rw, i, j) for a3d_resize_crop."""
    sc = np.random.uniform(*scales)
    rh, rw = int(raw_h * sc), int(raw_w * sc)
    if rh < 1 or rw < 1 or raw_h - rh >= rh or raw_w - rw >= rw:
        raise ValueError(f"image_rescale {scales}: reflect padding needs the resized image to be larger than half the original")
    ph, pw = max(rh, raw_h), max(rw, raw_w)
    if ph == raw_h and pw == raw_w:
        return rh, rw, 0, 0
    i = int(torch.randint(0, ph - raw_h + 1, size=(1,)).item())
    j = int(torch.randint(0, pw - raw_w + 1, size=(1,)).item())
    return rh, rw, i, j


def resize_crop(x, params, scale=1.0, shift=0.0):
    """x (F, N, C, H, W) device fp32, params (F, 4) int32 device -> augmented copy (a3d_resize_crop)."""
    O.L.require_gpu(x, params)
    x = O._c(x.float())
    F_, N, C, H, W = x.shape
    params = O._c(params.to(torch.int32))
    assert params.shape == (F_, 4)
    out = torch.empty_like(x)
    O.L.call("a3d_resize_crop", x.data_ptr(), out.data_ptr(), params.data_ptr(), F_, N * C, H, W, float(scale), float(shift),
             O.L.stream())
    return out


class Rotate:
    """datasets/utils.py:103-183.  The dataset asserts point_cloud_rotate_yaw_range == 0 (dataset_engine.py:82), for which
    the augmentation is the identity; other ranges are unreachable through RLBenchDataset."""

    def __init__(self, gripper_loc_bounds, yaw_range, num_tries=10):
        if float(yaw_range) != 0.0:
            raise NotImplementedError("point_cloud_rotate_yaw_range != 0 is rejected by the reference's dataset as well")

    def __call__(self, pcds, gripper, action, mask, trajectory=None):
        return pcds, gripper, action, trajectory


class TrajectoryInterpolator:
    """datasets/utils.py:186-214: resample a (n, 8) trajectory to a fixed length (cubic spline; the gripper-open channel
    linearly), quaternion re-normalised.  Host-side: it runs once per keyframe inside the DataLoader worker."""

    def __init__(self, use=False, interpolation_length=50):
        self._use, self._interpolation_length = use, interpolation_length

    def __call__(self, trajectory):
        if not self._use:
            return trajectory
        from scipy.interpolate import CubicSpline, interp1d
        t = trajectory.numpy()
        old, new = np.linspace(0, 1, len(t)), np.linspace(0, 1, self._interpolation_length)
        cols = [(interp1d(old, t[:, c]) if c == 7 else CubicSpline(old, t[:, c]))(new) for c in range(t.shape[1])]
        out = torch.tensor(np.stack(cols, axis=1))
        q = out[:, 3:7]
        out[:, 3:7] = q / torch.clamp(q.square().sum(dim=-1).sqrt().unsqueeze(-1), min=1e-10)
        return out


class RLBenchDataset(Dataset):
    """Episode files -> per-chunk training items; constructor and item dictionary of datasets/dataset_engine.py:14-258.

    On-disk episode (data_preprocessing/data_gen.py:122-132), indexed by frame id:
      [0] frame ids  [1] observations (n_cam, 2 = rgb | xyz, 3, H, W)  [2] actions (1, 8)  [3] camera dicts
      [4] gripper poses (1, 8)  [5] low-level trajectories (N_i, 8)
    One difference from the reference, by design: in training mode the item carries the UN-augmented `rgbs` / `pcds` plus
    `resize_params` (n_frames, 4) int32 -- the draws of the `Resize` augmentation -- and `DeviceLoader` applies them on the
    GPU.  The host-side RNG streams (`random` for the chunk and the instruction, numpy / torch for the augmentation) are
    consumed in the reference's order, so seeded runs pick the same chunks, instructions and crops."""

    _EPISODE_PATTERNS = ("*.npy", "*.dat", "*.pkl")        # the reference's listing order (dataset_engine.py:94-97)

    def __init__(self, root, instructions=None, taskvar=[('close_door', 0)], max_episode_length=5, cache_size=0,
                 max_episodes_per_task=100, num_iters=None, cameras=("wrist", "left_shoulder", "right_shoulder"),
                 training=True, gripper_loc_bounds=None, image_rescale=(1.0, 1.0), point_cloud_rotate_yaw_range=0.0,
                 return_low_lvl_trajectory=False, dense_interpolation=False, interpolation_length=100, action_dim=8,
                 predict_short=None):
        self._cache, self._cache_size = {}, cache_size
        self._cameras = cameras
        self._max_episode_length = max_episode_length
        self._num_iters = num_iters
        self._training = training
        self._taskvar = taskvar
        self._action_dim = action_dim
        self._predict_short = predict_short
        self._root = [Path(r).expanduser() for r in ([root] if isinstance(root, (Path, str)) else root)]
        self._return_low_lvl_trajectory = return_low_lvl_trajectory
        if return_low_lvl_trajectory:
            assert dense_interpolation or predict_short
            self._interpolate_traj = TrajectoryInterpolator(use=dense_interpolation, interpolation_length=interpolation_length)
        if training:
            self._image_rescale = tuple(image_rescale)
            self._rotate = Rotate(gripper_loc_bounds=gripper_loc_bounds, yaw_range=point_cloud_rotate_yaw_range)
        self._instructions, self._num_vars = self._collect_instructions(instructions)
        self._data_dirs, self._episodes = self._index_episodes(max_episodes_per_task)
        self._num_episodes = len(self._episodes)
        print(f"Created dataset from {self._root} with {self._num_episodes}")

    # ---- construction
    def _task_dirs(self):
        for r, (task, var) in itertools.product(self._root, self._taskvar):
            yield task, var, r / f"{task}+{var}"

    def _collect_instructions(self, instructions):
        """Only the instructions of (task, variation) folders that exist; variations counted per task (:62-70)."""
        kept, num_vars = defaultdict(dict), Counter()
        for task, var, d in self._task_dirs():
            if d.is_dir():
                if instructions is not None:
                    kept[task][var] = instructions[task][var]
                num_vars[task] += 1
        return kept, num_vars

    def _index_episodes(self, max_per_task):
        """(task, variation, file) per episode: an equal share per variation, then at most max_per_task per task (:84-117)."""
        dirs, by_task = [], defaultdict(list)
        for task, var, d in self._task_dirs():
            if not d.is_dir():
                print(f"Can't find dataset folder {d}")
                continue
            files = [(task, var, ep) for pat in self._EPISODE_PATTERNS for ep in d.glob(pat)]
            if max_per_task > -1:
                files = files[:max_per_task // self._num_vars[task] + 1]
            if not files:
                print(f"Can't find episodes at folder {d}")
                continue
            dirs.append(d)
            by_task[task] += files
        episodes = []
        for task, eps in by_task.items():
            if -1 < max_per_task < len(eps):
                eps = random.sample(eps, max_per_task)
            episodes += eps
        return dirs, episodes

    def read_from_cache(self, args):
        """Bounded episode cache with the reference's time-based eviction (:119-137)."""
        if self._cache_size == 0:
            return loader(args)
        if args not in self._cache:
            value = loader(args)
            if len(self._cache) == self._cache_size:
                del self._cache[list(self._cache.keys())[int(time()) % self._cache_size]]
            if len(self._cache) >= self._cache_size:
                return value
            self._cache[args] = value
        return self._cache[args]

    @staticmethod
    def _unnormalize_rgb(rgb):
        return rgb / 2 + 0.5

    # ---- one item
    def _observations(self, episode, frame_ids):
        """(rgb in [0, 1], xyz), each (n_frames, n_cam, 3, H, W), cameras in the order the dataset was asked for."""
        frames = [episode[1][i] for i in frame_ids]
        states = torch.stack([f if isinstance(f, torch.Tensor) else torch.from_numpy(f) for f in frames])
        if episode[3]:
            stored = list(episode[3][0].keys())
            assert all(c in stored for c in self._cameras)
            states = states[:, torch.tensor([stored.index(c) for c in self._cameras])]
        return self._unnormalize_rgb(states[:, :, 0]), states[:, :, 1]

    def _trajectories(self, episode, frame_ids):
        """Zero-padded (n_frames, T_max, 8) low-level trajectories, their lengths and the padding mask (1 = padded)."""
        items = [self._interpolate_traj(episode[5][i]) for i in frame_ids]
        lens = torch.as_tensor([len(it) for it in items])
        traj = torch.zeros(len(items), int(lens.max()), 8)
        mask = torch.zeros(traj.shape[:-1])
        for k, it in enumerate(items):
            traj[k, :len(it)] = it
            mask[k, len(it):] = 1
        return traj, lens, mask

    def __getitem__(self, episode_id):
        task, variation, file = self._episodes[episode_id % self._num_episodes]
        episode = self.read_from_cache(file)
        if episode is None:
            return None
        # one random chunk of at most max_episode_length keyframes (dynamic chunking, :153-162)
        n_chunks = math.ceil(len(episode[0]) / self._max_episode_length)
        chunk = random.randint(0, n_chunks - 1)
        frame_ids = episode[0][chunk * self._max_episode_length:(chunk + 1) * self._max_episode_length]
        rgbs, pcds = self._observations(episode, frame_ids)
        n = len(rgbs)
        action = torch.cat([episode[2][i] for i in frame_ids])
        if self._instructions:
            instr = random.choice(self._instructions[task][variation])[None].repeat(n, 1, 1)
        else:
            instr = torch.zeros((n, 53, 512))
        poses = episode[4]
        gripper = torch.cat([poses[i] for i in frame_ids])
        history = torch.stack([torch.cat([poses[max(0, i - back)] for i in frame_ids]) for back in (2, 1)] + [gripper], dim=1)
        traj = traj_mask = None
        if self._return_low_lvl_trajectory:
            traj, traj_lens, traj_mask = self._trajectories(episode, frame_ids)

        H, W = rgbs.shape[-2:]
        draws = (H, W, 0, 0)                                   # identity: evaluation items are not augmented
        if self._training:
            pcds, gripper, action, traj = self._rotate(pcds, gripper, action, None, traj)
            if traj is not None:
                for k, tlen in enumerate(traj_lens):
                    traj[k, tlen:] = 0
            # the Resize draws (one set per item, shared by its frames and by RGB / XYZ); applied by DeviceLoader
            draws = sample_resize_params(self._image_rescale, H, W)
        d = self._action_dim
        item = {"task": [task] * n, "rgbs": rgbs, "pcds": pcds, "action": action[..., :d], "instr": instr,
                "curr_gripper": gripper[..., :d], "curr_gripper_history": history[..., :d],
                "resize_params": torch.tensor([draws], dtype=torch.int32).repeat(n, 1)}
        if traj is not None:
            item["trajectory"], item["trajectory_mask"] = traj[..., :d], traj_mask.bool()
        return item

    def __len__(self):
        return self._num_iters if self._num_iters is not None else self._num_episodes


def _collate(batch, keys):
    batch = [item for item in batch if item is not None]
    ret = {key: torch.cat([item[key].float() if key not in ("trajectory_mask", "resize_params") else item[key]
                           for item in batch]) for key in keys if key in batch[0]}
    ret["task"] = [t for item in batch for t in item["task"]]
    return ret


def keypose_collate_fn(batch):
    """main_keypose.py:284-292: unfold multi-step demos into one longer batch (+ the deferred augmentation draws)."""
    return _collate(batch, ["rgbs", "pcds", "curr_gripper", "action", "instr", "resize_params"])


def traj_collate_fn(batch):
    """main_trajectory.py:277-292"""
    return _collate(batch, ["trajectory", "trajectory_mask", "rgbs", "pcds", "curr_gripper", "action", "instr", "resize_params"])


class DeviceLoader:
    """Wraps a DataLoader of collated host batches: pinned host->device copies and the deferred `Resize` augmentation run on
    a copy stream one batch AHEAD of the consumer, so that neither appears in the training step's critical path (a step's
    input is 64 x 4 x 2 x 3 x 256^2 x 4 B = 403 MB, ~6.5 ms of PCIe Gen5 x16 -- DESIGN.md §7)."""

    def __init__(self, loader, device, augment=True):
        self.loader, self.device, self.augment = loader, torch.device(device), augment
        self._stream = None

    def __len__(self):
        return len(self.loader)

    def _upload(self, batch):
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=self.device)
        out = {}
        with torch.cuda.stream(self._stream):
            for k, v in batch.items():
                if torch.is_tensor(v):
                    v = v if v.is_pinned() else v.pin_memory()
                    out[k] = v.to(self.device, non_blocking=True)
                else:
                    out[k] = v
            params = out.pop("resize_params", None)
            if self.augment and params is not None:
                host = batch["resize_params"]
                H, W = batch["rgbs"].shape[-2:]
                identity = bool(((host[:, 0] == H) & (host[:, 1] == W) & (host[:, 2] == 0) & (host[:, 3] == 0)).all())
                if not identity:
                    out["rgbs"] = resize_crop(out["rgbs"], params)
                    out["pcds"] = resize_crop(out["pcds"], params)
        return out

    def __iter__(self):
        it = iter(self.loader)
        nxt = None
        for batch in it:
            cur, nxt = nxt, self._upload(batch)
            if cur is not None:
                yield self._ready(cur)
        if nxt is not None:
            yield self._ready(nxt)

    def _ready(self, batch):
        # NOTE: the NEXT batch's upload was enqueued before this one is handed out: the consumer only waits for work that is
        # already complete or in flight on the copy stream
        cur = torch.cuda.current_stream(self.device)
        cur.wait_stream(self._stream)
        for v in batch.values():
            if torch.is_tensor(v):
                v.record_stream(cur)
        return batch
