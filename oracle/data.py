"""ORACLE (test infrastructure only) -- CPU restatement of the data plane's arithmetic.

`Resize` (datasets/utils.py:40-100) is written by the reference in terms of torchvision, which is absent from this image and
from /root/reference: transforms_f.resize(NEAREST) on a float tensor is torch.nn.functional.interpolate(mode="nearest"),
transforms_f.pad(reflect) is F.pad(mode="reflect"), RandomCrop.get_params draws i then j with torch.randint (none when the
sizes already match), transforms_f.crop is a slice (torchvision 0.14 transforms/functional_tensor.py).  This module
restates the composition as an explicit index map in numpy; tests/test_oracle_golden.py pins it against the
F.interpolate / F.pad / slice composition of the installed torch (bit-exact: the operation only moves values).
Parity status: pinned to torch's operators, NOT to torchvision itself (third-party, not installed).
"""
import numpy as np
import torch


def resize_params(scales, raw_h, raw_w):
    """Draws of Resize.__call__ (datasets/utils.py:60-62,90-92): np.random.uniform for the scale, then torch.randint for the
    crop offsets -- consumed exactly as the reference does.  Returns (rh, rw, i, j)."""
    sc = np.random.uniform(*scales)
    rh, rw = int(raw_h * sc), int(raw_w * sc)
    ph, pw = max(rh, raw_h), max(rw, raw_w)                # extent after the reflect padding (utils.py:75-87)
    if ph == raw_h and pw == raw_w:                        # RandomCrop.get_params returns (0, 0) without drawing
        return rh, rw, 0, 0
    i = int(torch.randint(0, ph - raw_h + 1, size=(1,)).item())
    j = int(torch.randint(0, pw - raw_w + 1, size=(1,)).item())
    return rh, rw, i, j


def nearest_index(out_size, in_size):
    """ATen `nearest`: src = min(floorf(dst * (float)in / out), in - 1)"""
    scale = np.float32(in_size) / np.float32(out_size)
    d = np.arange(out_size, dtype=np.float32)
    return np.minimum(np.floor(d * scale).astype(np.int64), in_size - 1)


def resize_crop(x, rh, rw, i, j):
    """x (..., H, W) numpy -> nearest resize to (rh, rw), reflect-pad bottom/right to >= (H, W), crop (H, W) at (i, j)."""
    H, W = x.shape[-2:]
    yy = np.arange(H) + i
    xx = np.arange(W) + j
    yy = np.where(yy >= rh, 2 * (rh - 1) - yy, yy)
    xx = np.where(xx >= rw, 2 * (rw - 1) - xx, xx)
    iy = nearest_index(rh, H)[yy]
    ix = nearest_index(rw, W)[xx]
    return x[..., iy[:, None], ix[None, :]]


def unnormalize_rgb(rgb):
    """dataset_engine.py:134-137"""
    return rgb / 2 + 0.5


def interpolate_trajectory(traj, length):
    """TrajectoryInterpolator.__call__ (datasets/utils.py:186-214): cubic spline per channel, linear for the gripper-open
    channel (index 7), quaternion re-normalised.  traj (n, 8) float tensor -> (length, 8) float64 tensor."""
    from scipy.interpolate import CubicSpline, interp1d
    t = traj.numpy()
    old, new = np.linspace(0, 1, len(t)), np.linspace(0, 1, length)
    out = np.empty((length, t.shape[1]))
    for c in range(t.shape[1]):
        f = interp1d(old, t[:, c]) if c == 7 else CubicSpline(old, t[:, c])
        out[:, c] = f(new)
    out = torch.tensor(out)
    q = out[:, 3:7]
    out[:, 3:7] = q / torch.clamp(q.square().sum(-1).sqrt().unsqueeze(-1), min=1e-10)
    return out
