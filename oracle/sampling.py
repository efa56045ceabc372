"""ORACLE (test infrastructure only) -- ghost-point samplers, k-NN context selection, point-cloud down-sampling.

numpy restatements; citations are into /root/reference.
"""
import numpy as np


# ----------------------------------------------------------------------------- the reference's host sampler
def ref_sample_cube(bounds, num_points):
    """model/utils/utils.py:68-73 -- three global-RNG float64 np.random.uniform calls (x, then y, then z)."""
    x = np.random.uniform(bounds[0][0], bounds[1][0], num_points)
    y = np.random.uniform(bounds[0][1], bounds[1][1], num_points)
    z = np.random.uniform(bounds[0][2], bounds[1][2], num_points)
    return np.stack([x, y, z], axis=1)


def ref_sample_sphere(center, radius, bounds, num_points, max_rounds=10000):
    """model/utils/utils.py:76-84 -- batch rejection: draw num_points cube points, keep |p - c| < r, repeat, truncate.
    The reference has no round limit and never returns when the clipped box misses the ball; the
    oracle raises instead."""
    pts = np.empty((0, 3))
    rounds = 0
    while pts.shape[0] < num_points:
        cand = ref_sample_cube(bounds, num_points)
        l2 = np.linalg.norm(cand - center, axis=1)
        pts = np.concatenate([pts, cand[l2 < radius]])
        rounds += 1
        if rounds > max_rounds:
            raise RuntimeError("rejection sampler does not terminate: anchor outside the workspace")
    return pts[:num_points]


def ref_sample_ghost_points(gripper_loc_bounds, B, Ng, level, anchor=None, diameter=None):
    """act3d.py:394-440 (_sample_ghost_points): level 0 = workspace box, level >= 1 = ball around `anchor` (B,3)
    inside the box clipped to the workspace.  Consumes the global numpy RNG exactly like the reference and returns
    float32 (B, Ng, 3)."""
    glb = np.asarray(gripper_loc_bounds)
    if level == 0:
        pts = np.stack([ref_sample_cube(glb, Ng) for _ in range(B)])
    else:
        anchor = np.asarray(anchor)
        lo = np.clip(anchor - diameter / 2, a_min=glb[0], a_max=glb[1])
        hi = np.clip(anchor + diameter / 2, a_min=glb[0], a_max=glb[1])
        bounds = np.stack([lo, hi], axis=1)
        pts = np.stack([ref_sample_sphere(anchor[i], diameter / 2, bounds[i], Ng) for i in range(B)])
    return pts.astype(np.float32)


# ----------------------------------------------------------------------------- CPU twin of the device Philox sampler
_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = np.uint32(0x9E3779B9)
_W1 = np.uint32(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 (Salmon et al., SC'11), vectorised over uint32 arrays.  Twin of philox4x32_10 in heads.hip."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32) for c in (c0, c1, c2, c3))
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = _M0 * c0.astype(np.uint64)
            p1 = _M1 * c2.astype(np.uint64)
            n0 = ((p1 >> np.uint64(32)) & _MASK).astype(np.uint32) ^ c1 ^ k0
            n1 = (p1 & _MASK).astype(np.uint32)
            n2 = ((p0 >> np.uint64(32)) & _MASK).astype(np.uint32) ^ c3 ^ k1
            n3 = (p0 & _MASK).astype(np.uint32)
            c0, c1, c2, c3 = n0, n1, n2, n3
            k0 = np.uint32((int(k0) + int(_W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(_W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def _u01(x):
    return (x >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)


def philox_ghost_points(seed, offset, bounds, anchor, radius, B, Ng, level, max_attempts=64):
    """Bit-exact CPU twin of a3d_sample_ghost_points (heads.hip): counter = (i, b, level | attempt << 8, offset_lo),
    key = (seed_lo, seed_hi ^ offset_hi); u = (r >> 8) * 2^-24; p = lo + u * (hi - lo) in fp32 (no fma)."""
    bounds = np.asarray(bounds, dtype=np.float32)
    radius = np.float32(radius)
    k0 = seed & 0xFFFFFFFF
    k1 = ((seed >> 32) ^ (offset >> 32)) & 0xFFFFFFFF
    out = np.zeros((B, Ng, 3), dtype=np.float32)
    ii = np.arange(Ng, dtype=np.uint32)
    for b in range(B):
        if anchor is None:
            lo, hi = bounds[0], bounds[1]
        else:
            ctr = np.asarray(anchor[b], dtype=np.float32)
            lo = np.minimum(np.maximum(ctr - radius, bounds[0]), bounds[1])
            hi = np.minimum(np.maximum(ctr + radius, bounds[0]), bounds[1])
        done = np.zeros(Ng, dtype=bool)
        tries = 1 if anchor is None else max_attempts
        for a in range(tries):
            todo = ~done
            if not todo.any():
                break
            r = philox4x32_10(ii, np.full(Ng, b, np.uint32), np.full(Ng, level | (a << 8), np.uint32),
                              np.full(Ng, offset & 0xFFFFFFFF, np.uint32), k0, k1)
            p = np.stack([lo[c] + _u01(r[c]) * (hi[c] - lo[c]) for c in range(3)], axis=1).astype(np.float32)
            if anchor is None:
                ok = np.ones(Ng, dtype=bool)
            else:
                d = p - ctr
                ok = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]) < radius
            take = todo & ok
            out[b][take] = p[take]
            done |= take
        if anchor is not None and not done.all():
            out[b][~done] = np.minimum(np.maximum(ctr, bounds[0]), bounds[1])
    return out


class DropoutTwin:
    """CPU twin of the device dropout masks (csrc/a3d_common.h drop_keep8, csrc/dropout.hip, attention kernels):
    one Philox4x32-10 call per block of 8 elements, counter = (c0, c1, c2, site), key = the two 32-bit halves of
    seed + offset * 0x9E3779B97F4A7C15 (mod 2^64); element j of the block is kept iff the j-th 16-bit field of the 128 output bits is
    >= round(p * 65536); kept elements are scaled by 1 / (1 - p) (fp32).  Restates nn.Dropout / F.dropout in training
    mode (layers.py:34,58,82-84; multihead_custom_attention.py:413; diffusion_head.py:46,183,193) up to the random
    stream, which is the device's own."""

    def __init__(self, seed, offset, p):
        key = (int(seed) + int(offset) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        self.k0 = key & 0xFFFFFFFF
        self.k1 = key >> 32
        self.p = float(p)
        self.thr = int(np.rint(np.float32(p) * np.float32(65536.0)))
        self.scale = np.float32(1.0) / (np.float32(1.0) - np.float32(p))

    @staticmethod
    def site_id(name, sub=0):
        import zlib
        return ((zlib.crc32(name.encode()) & 0xFFFFFFF8) | sub) & 0xFFFFFFFF

    def _keep(self, c0, c1, c2, site):
        n = c0.shape[0]
        r = philox4x32_10(c0.astype(np.uint32), c1.astype(np.uint32), c2.astype(np.uint32), np.full(n, site, np.uint32),
                          self.k0, self.k1)
        f = np.stack([r[0] & np.uint32(0xFFFF), r[0] >> np.uint32(16), r[1] & np.uint32(0xFFFF), r[1] >> np.uint32(16),
                      r[2] & np.uint32(0xFFFF), r[2] >> np.uint32(16), r[3] & np.uint32(0xFFFF), r[3] >> np.uint32(16)], axis=1)
        return f >= self.thr                                           # (n, 8) bool

    def flat(self, site, shape):
        """scaled keep mask (float32) of a contiguous tensor of `shape`, flat element indexing"""
        n = int(np.prod(shape))
        nblk = (n + 7) // 8
        blk = np.arange(nblk, dtype=np.uint64)
        keep = self._keep(blk & np.uint64(0xFFFFFFFF), blk >> np.uint64(32), np.full(nblk, 0xFFFFFFFF, np.uint64), site)
        return (keep.reshape(-1)[:n].reshape(shape).astype(np.float32) * self.scale)

    def attn(self, site, B, H, Lq, S):
        """scaled keep mask (B, H, Lq, S) of the attention weights: block = 8 consecutive keys of one (b, h, query)"""
        nblk = (S + 7) // 8
        bh, q, kb = np.meshgrid(np.arange(B * H), np.arange(Lq), np.arange(nblk), indexing="ij")
        keep = self._keep(kb.reshape(-1), q.reshape(-1), bh.reshape(-1), site)
        keep = keep.reshape(B, H, Lq, nblk * 8)[..., :S]
        return keep.astype(np.float32) * self.scale


# ----------------------------------------------------------------------------- scene selection
def pcd_downsample(pcd, factor):
    """act3d.py:379-383 / encoder.py:147-158: F.interpolate(pcd, scale_factor=1/f, mode='bilinear') followed by
    "(bt ncam) c h w -> bt (ncam h w) c".  For even f, source index (dst + 0.5) * f - 0.5 = f*dst + f/2 - 1 + 0.5:
    the sample is the mean of the 2x2 block at offset f/2 - 1.  The order of the three additions follows the ATen
    CPU kernel that the reference run takes for the given output size (pinned by tests/test_oracle_cpu.py against
    F.interpolate itself).  pcd: (B, C, 3, H, W) float32 -> (B, C*h*w, 3)."""
    pcd = np.asarray(pcd, dtype=np.float32)
    B, C, _, H, W = pcd.shape
    o = factor // 2 - 1
    p00 = pcd[..., o::factor, o::factor]
    p01 = pcd[..., o::factor, o + 1::factor]
    p10 = pcd[..., o + 1::factor, o::factor]
    p11 = pcd[..., o + 1::factor, o + 1::factor]
    half, quarter = np.float32(0.5), np.float32(0.25)
    if (H // factor) + (W // factor) <= 128:
        # ATen's "vectorized" CPU kernel (UpSampleKernel.cpp, _use_vectorized_kernel_cond_2d): left-to-right sum
        out = ((quarter * p00 + quarter * p01) + quarter * p10) + quarter * p11
    else:
        # ATen's generic separable kernel: rows first, then columns
        t0 = half * p00 + half * p01
        t1 = half * p10 + half * p11
        out = half * t0 + half * t1                               # (B, C, 3, h, w)
    return np.ascontiguousarray(out.transpose(0, 1, 3, 4, 2)).reshape(B, -1, 3)


def knn_topk(pos, xyz, k):
    """act3d.py:244-245: l2 = ((pos - pcd)**2).sum(-1).sqrt(); topk(k, largest=False).indices.
    fp32, sum order (dx^2 + dy^2) + dz^2; result ordered by ascending (distance, index) -- torch leaves the order of
    exactly tied distances unspecified, the oracle (and the HIP kernel) define it by index."""
    pos = np.asarray(pos, dtype=np.float32).reshape(-1, 1, 3)
    xyz = np.asarray(xyz, dtype=np.float32)
    d = pos - xyz
    d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
    dist = np.sqrt(d2).astype(np.float32)
    B, N = dist.shape
    order = np.lexsort((np.broadcast_to(np.arange(N), (B, N)), dist), axis=-1)[:, :k]
    return order.astype(np.int64), np.take_along_axis(dist, order, axis=1)


def traj_nn_topk(traj_xyz, xyz, k):
    """find_traj_nn (model/utils/utils.py:39-48): squared distance of every scene point to its nearest trajectory point,
    then the k smallest; fp32, sum order (dx^2 + dy^2) + dz^2, ascending (distance, index) like knn_topk above."""
    t = np.asarray(traj_xyz, dtype=np.float32)[:, :, None, :]          # (B, L, 1, 3)
    x = np.asarray(xyz, dtype=np.float32)[:, None, :, :]               # (B, 1, N, 3)
    d = t - x
    d2 = ((d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]).astype(np.float32)
    dist = d2.min(axis=1)
    B, N = dist.shape
    order = np.lexsort((np.broadcast_to(np.arange(N), (B, N)), dist), axis=-1)[:, :k]
    return order.astype(np.int64), np.take_along_axis(dist, order, axis=1)
