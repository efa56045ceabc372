"""Data plane: episode files -> collated device batches.

Provides what `datasets/dataset_engine.py:14-258` (RLBenchDataset), `datasets/utils.py` (loader, Resize, Rotate,
TrajectoryInterpolator) and the collate functions of `main_keypose.py:284-292` / `main_trajectory.py:277-292` provide --
same constructor keywords, item keys and generator consumption (pinned by tests/golden/dataset.pt) -- organised for
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
from collections import Counter, OrderedDict
from pathlib import Path
from pickle import UnpicklingError
from typing import NamedTuple, Optional

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


# ------------------------------------------------------------------------------------------------ the dataset
# Design: three small pieces instead of one class that does everything in __getitem__ --
#   EpisodeIndex  which (task, variation, file) triples exist, built once (the only place that touches the directory tree);
#   EpisodeStore  file -> decoded episode, with a bounded least-recently-used cache;
#   draw_plan     ALL random decisions of one item (which chunk of keyframes, which instruction, the Resize draws) in one
#                 function, consuming python's / numpy's / torch's generators exactly as the reference's __getitem__ does
#                 (datasets/dataset_engine.py:153-216), so seeded runs pick the same chunks, instructions and crops;
#   EpisodeView   tensors of a set of keyframes, sliced out of the decoded episode by the plan.
# On-disk episode (data_preprocessing/data_gen.py:122-132), indexed by frame id:
#   [0] frame ids  [1] observations (n_cam, 2 = rgb | xyz, 3, H, W)  [2] actions (1, 8)  [3] camera dicts
#   [4] gripper poses (1, 8)  [5] low-level trajectories (N_i, 8)
class EpisodeRef(NamedTuple):
    task: str
    variation: int
    path: Path


class EpisodeIndex:
    """The episodes a dataset draws from.  Selection rule of the reference (dataset_engine.py:84-117): per (task, variation)
    folder the files in glob order (*.npy, then *.dat, then *.pkl), truncated to an equal share of `max_per_task` per
    variation (+ 1); then, per task, a `random.sample` down to `max_per_task` if still above it."""

    PATTERNS = ("*.npy", "*.dat", "*.pkl")

    def __init__(self, roots, taskvar, max_per_task):
        folders = [(task, var, Path(root).expanduser() / f"{task}+{var}") for root, (task, var) in itertools.product(roots, taskvar)]
        self.present = [(task, var, d) for task, var, d in folders if d.is_dir()]
        for task, var, d in folders:
            if not d.is_dir():
                print(f"[dataset] no folder {d}")
        self.variations_per_task = Counter(task for task, _, _ in self.present)
        per_task = {}
        self.folders = []
        for task, var, d in self.present:
            refs = [EpisodeRef(task, var, f) for pattern in self.PATTERNS for f in d.glob(pattern)]
            if max_per_task > -1:
                refs = refs[:max_per_task // self.variations_per_task[task] + 1]
            if refs:
                self.folders.append(d)
                per_task.setdefault(task, []).extend(refs)
            else:
                print(f"[dataset] no episodes in {d}")
        self.refs = []
        for task, refs in per_task.items():
            self.refs += random.sample(refs, max_per_task) if -1 < max_per_task < len(refs) else refs

    def __len__(self):
        return len(self.refs)

    def __getitem__(self, i):
        return self.refs[i % len(self.refs)]


class EpisodeStore:
    """Decoded episodes by path; at most `capacity` kept, least recently used evicted (capacity 0: no cache)."""

    def __init__(self, capacity):
        self.capacity = capacity
        self._kept = OrderedDict()

    def __call__(self, path):
        if self.capacity == 0:
            return loader(path)
        if path in self._kept:
            self._kept.move_to_end(path)
            return self._kept[path]
        episode = loader(path)
        self._kept[path] = episode
        while len(self._kept) > self.capacity:
            self._kept.popitem(last=False)
        return episode


class ItemPlan(NamedTuple):
    chunk: int                    # which block of max_episode_length keyframes
    instruction: Optional[int]    # index into the (task, variation)'s instruction embeddings, None without instructions
    resize: tuple                 # (rh, rw, i, j) of a3d_resize_crop; identity for evaluation items


def draw_plan(n_keyframes, chunk_len, n_instructions, image_hw, rescale):
    """Every random decision of one item.  Generator consumption == the reference's: python `random` for the chunk
    (randint, :160) and then the instruction (choice, :180); numpy then torch for the Resize draws (datasets/utils.py:60-92),
    only for training items (rescale not None)."""
    chunk = random.randrange(math.ceil(n_keyframes / chunk_len))
    instruction = random.randrange(n_instructions) if n_instructions else None
    H, W = image_hw
    resize = sample_resize_params(rescale, H, W) if rescale is not None else (H, W, 0, 0)
    return ItemPlan(chunk, instruction, resize)


class EpisodeView:
    """Tensor access to the keyframes `ids` of one decoded episode."""

    def __init__(self, episode, ids, cameras):
        self.episode, self.ids, self.cameras = episode, list(ids), cameras

    def __len__(self):
        return len(self.ids)

    def _per_frame(self, field):
        return torch.cat([self.episode[field][i] for i in self.ids])

    def observations(self):
        """(rgb in [0, 1], xyz), each (n_frames, n_cam, 3, H, W), cameras in the order the dataset was asked for"""
        obs = torch.stack([torch.as_tensor(self.episode[1][i]) for i in self.ids])
        if self.episode[3]:
            stored = list(self.episode[3][0])
            missing = [c for c in self.cameras if c not in stored]
            assert not missing, f"cameras {missing} are not in the episode ({stored})"
            obs = obs[:, torch.tensor([stored.index(c) for c in self.cameras])]
        return obs[:, :, 0] / 2 + 0.5, obs[:, :, 1]

    def actions(self):
        return self._per_frame(2)

    def grippers(self):
        return self._per_frame(4)

    def gripper_history(self):
        """(n_frames, 3, 8): the pose two keyframes back, one back, and now (clamped at the episode start)"""
        poses = self.episode[4]
        return torch.stack([torch.cat([poses[max(0, i - back)] for i in self.ids]) for back in (2, 1, 0)], dim=1)

    def trajectories(self, resample):
        """zero-padded (n_frames, T_max, 8) low-level trajectories and the padding mask (True = padded)"""
        pieces = [resample(self.episode[5][i]) for i in self.ids]
        longest = max(len(p) for p in pieces)
        traj = torch.zeros(len(pieces), longest, 8)
        padded = torch.ones(len(pieces), longest, dtype=torch.bool)
        for k, p in enumerate(pieces):
            traj[k, :len(p)] = p
            padded[k, :len(p)] = False
        return traj, padded


class RLBenchDataset(Dataset):
    """Constructor keywords and item dictionary of datasets/dataset_engine.py:14-258; one item = one random block of at most
    `max_episode_length` keyframes of one episode.

    One difference from the reference, by design: a training item carries the UN-augmented `rgbs` / `pcds` plus
    `resize_params` (n_frames, 4) int32 -- the draws of the `Resize` augmentation -- and `DeviceLoader` applies them on the
    GPU."""

    def __init__(self, root, instructions=None, taskvar=[('close_door', 0)], max_episode_length=5, cache_size=0,
                 max_episodes_per_task=100, num_iters=None, cameras=("wrist", "left_shoulder", "right_shoulder"),
                 training=True, gripper_loc_bounds=None, image_rescale=(1.0, 1.0), point_cloud_rotate_yaw_range=0.0,
                 return_low_lvl_trajectory=False, dense_interpolation=False, interpolation_length=100, action_dim=8,
                 predict_short=None):
        roots = [root] if isinstance(root, (Path, str)) else list(root)
        self._cameras, self._chunk_len, self._num_iters = cameras, max_episode_length, num_iters
        self._training, self._action_dim = training, action_dim
        self._resample = None
        if return_low_lvl_trajectory:
            assert dense_interpolation or predict_short
            self._resample = TrajectoryInterpolator(use=dense_interpolation, interpolation_length=interpolation_length)
        # training-only augmentations: the Resize draws (applied on the device) and the yaw rotation (identity: the
        # reference asserts a zero range, dataset_engine.py:82)
        self._rescale = tuple(image_rescale) if training else None
        if training:
            Rotate(gripper_loc_bounds=gripper_loc_bounds, yaw_range=point_cloud_rotate_yaw_range)
        self._index = EpisodeIndex(roots, taskvar, max_episodes_per_task)
        self._store = EpisodeStore(cache_size)
        # instruction embeddings of the (task, variation) folders that exist (dataset_engine.py:62-70)
        self._instructions = {}
        if instructions is not None:
            for task, var, _ in self._index.present:
                self._instructions.setdefault(task, {})[var] = instructions[task][var]
        print(f"[dataset] {len(self._index)} episodes under {[str(r) for r in roots]}")

    def __len__(self):
        return self._num_iters if self._num_iters is not None else len(self._index)

    def read_from_cache(self, path):
        return self._store(path)

    def __getitem__(self, episode_id):
        ref = self._index[episode_id]
        episode = self._store(ref.path)
        if episode is None:
            return None
        options = self._instructions[ref.task][ref.variation] if self._instructions else None
        first = episode[1][episode[0][0]]
        plan = draw_plan(len(episode[0]), self._chunk_len, 0 if options is None else len(options), tuple(first.shape[-2:]),
                         self._rescale)
        ids = episode[0][plan.chunk * self._chunk_len:(plan.chunk + 1) * self._chunk_len]
        view = EpisodeView(episode, ids, self._cameras)
        n, d = len(view), self._action_dim
        rgbs, pcds = view.observations()
        instr = torch.zeros((n, 53, 512)) if options is None else options[plan.instruction][None].repeat(n, 1, 1)
        item = {"task": [ref.task] * n, "rgbs": rgbs, "pcds": pcds, "action": view.actions()[..., :d], "instr": instr,
                "curr_gripper": view.grippers()[..., :d], "curr_gripper_history": view.gripper_history()[..., :d],
                "resize_params": torch.tensor([plan.resize], dtype=torch.int32).repeat(n, 1)}
        if self._resample is not None:
            traj, padded = view.trajectories(self._resample)
            item["trajectory"], item["trajectory_mask"] = traj[..., :d], padded
        return item


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
