"""CPU checks of the index logic of csrc/conv3x3.hip (no GPU, no HIP code is executed): a lane-level numpy model of the kernel --
weight / halo staging with the LDS swizzles, the carry-propagating tile cursor, the XCD tile remap, the fragment gathers in the
16x16x32 MFMA lane layout (A operand: lane (li, g) holds row li, k = 8 g .. 8 g + 7; B operand: column li, same k; result: lane
(li, g) holds rows 4 g .. 4 g + 3 of column li -- the layout conv1x1.hip is validated with on the GPU), the permuted weight rows and
the stores -- against F.conv2d (model/utils/clip.py:22-43: Conv2d(.., 3, padding=1, bias=False)), and the enumeration over
ds_read_b128's lane groups behind the header's "conflict-free from any base row" claim.  The GPU parity test of the kernel itself
is tests/test_kernels_gpu.py::test_conv3x3_gemm_with_folded_batchnorm."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

HW_, HP = 34, 340                      # halo tile: 10 rows of 34 pixels


def woff(row, seg):                   # c3_woff (conv1x1.hip's c1_woff is the same function)
    return row * 32 + ((seg ^ (((row >> 1) & 1) | (((row >> 4) & 1) << 1))) << 3)


def xoff(hp, hx, seg):                # c3_xoff
    return hp * 32 + ((seg ^ ((hx >> 1) & 3)) << 3)


class Cursor:                          # C3Cursor
    def __init__(self, img, ty, tx):
        self.img, self.ty, self.tx = img, ty, tx

    def advance(self, dimg, dty, dtx, tiles_y, tiles_x):
        self.tx += dtx
        if self.tx >= tiles_x:
            self.tx -= tiles_x
            self.ty += 1
        self.ty += dty
        if self.ty >= tiles_y:
            self.ty -= tiles_y
            self.img += 1
        self.img += dimg


def model_kernel(x, w, HALVES, NT, nwg, D, yblocks=1):
    """x [n][H][W][Cin], w [Ctot][3][3][Cin] fp32; returns y [n][H][W][Ctot] computed the way the kernel's lanes compute it."""
    nimg, H, W, CIN = x.shape
    COUT = 16 * NT
    ctot = COUT * yblocks
    assert CIN == 32 * HALVES and w.shape[0] == ctot and D % HALVES == 0
    y = np.full((nimg, H, W, ctot), np.nan, np.float32)
    xf, wf, yf = x.reshape(-1), w.reshape(-1), y.reshape(-1)
    tiles_x, tiles_y = W // 32, H // 8
    tpi = tiles_x * tiles_y
    ntiles = nimg * tpi
    for by in range(yblocks):
        co0 = by * COUT
        for bid in range(nwg):
            lb = (bid & 7) * (nwg >> 3) + (bid >> 3) if nwg % 8 == 0 else bid
            Ws = np.zeros(HALVES * 9 * COUT * 32, np.float32)
            for i in range(HALVES * 9 * COUT * 4):
                seg, r = i & 3, i >> 2
                co, tap, h = r % COUT, (r // COUT) % 9, r // (COUT * 9)
                src = ((co0 + co) * 9 + tap) * CIN + h * 32 + seg * 8
                Ws[woff((h * 9 + tap) * COUT + co, seg):][:8] = wf[src:src + 8]
            my_tiles = (ntiles - lb + nwg - 1) // nwg if lb < ntiles else 0
            total = my_tiles * HALVES
            if total == 0:
                continue
            dimg = nwg // tpi
            dty = (nwg - dimg * tpi) // tiles_x
            dtx = nwg - dimg * tpi - dty * tiles_x
            img0 = lb // tpi
            ty0 = (lb - img0 * tpi) // tiles_x
            tx0 = lb - img0 * tpi - ty0 * tiles_x
            cl, cc = Cursor(img0, ty0, tx0), Cursor(img0, ty0, tx0)
            st = {"lk": 0, "lhalf": 0}

            def load_next():
                y0, x0, lh = cl.ty * 8, cl.tx * 32, st["lhalf"]
                regs = {}
                for t in range(256):
                    for i in range(6):
                        hp = (t + i * 256) >> 2
                        hy = hp // HW_
                        hx = hp - hy * HW_
                        iy, ix = min(max(y0 - 1 + hy, 0), H - 1), min(max(x0 - 1 + hx, 0), W - 1)
                        off = cl.img * H * W * CIN + lh * 32 + (iy * W + ix) * CIN + (t & 3) * 8
                        regs[(t, i)] = xf[off:off + 8].copy()
                st["lhalf"] += 1
                if st["lhalf"] == HALVES:
                    st["lhalf"] = 0
                    if st["lk"] + 1 < my_tiles:
                        st["lk"] += 1
                        cl.advance(dimg, dty, dtx, tiles_y, tiles_x)
                return regs

            xr = [load_next() for _ in range(D)]
            Xs = np.zeros((2, HP * 32), np.float32)
            acc = np.zeros((4, NT, 4, 64, 4), np.float32)          # wave, tn, tm, lane, r
            for s0 in range(0, total, D):
                for j in range(D):
                    s = s0 + j
                    if s >= total:
                        break
                    half, buf = j % HALVES, s & 1
                    y0, x0 = cc.ty * 8, cc.tx * 32
                    for t in range(256):                             # stage
                        for i in range(6):
                            hp = (t + i * 256) >> 2
                            hy = hp // HW_
                            hx = hp - hy * HW_
                            iy, ix = y0 - 1 + hy, x0 - 1 + hx
                            v = xr[j][(t, i)].copy()
                            if not (0 <= iy < H and 0 <= ix < W):
                                v[:] = 0
                            if hp < HP:
                                Xs[buf][xoff(hp, hx, t & 3):][:8] = v
                    xr[j] = load_next()
                    for wave in range(4):
                        for tap in range(9):
                            kh, kw = tap // 3, tap % 3
                            for tn in range(NT):
                                A = np.zeros((16, 32), np.float32)
                                for li in range(16):
                                    for g in range(4):
                                        o = woff((li >> 2) * (4 * NT) + tn * 4 + (li & 3), g)
                                        A[li, g * 8:g * 8 + 8] = Ws[half * 9 * COUT * 32 + (kh * 3 + kw) * COUT * 32 + o:][:8]
                                for tm in range(4):
                                    B = np.zeros((32, 16), np.float32)
                                    for li in range(16):
                                        col = (tm & 1) * 16 + li + kw
                                        for g in range(4):
                                            o = xoff((2 * wave + (tm >> 1)) * HW_ + col, col, g)
                                            B[g * 8:g * 8 + 8, li] = Xs[buf][o + kh * (HW_ * 32):][:8]
                                    Dm = A @ B
                                    for lane in range(64):
                                        li, g = lane & 15, lane >> 4
                                        acc[wave, tn, tm, lane, :] += Dm[g * 4:g * 4 + 4, li]
                    if half == HALVES - 1:
                        for wave in range(4):
                            for tm in range(4):
                                for lane in range(64):
                                    li, g = lane & 15, lane >> 4
                                    oy, ox = y0 + 2 * wave + (tm >> 1), x0 + (tm & 1) * 16 + li
                                    dst = cc.img * H * W * ctot + co0 + (oy * W + ox) * ctot + g * 4 * NT
                                    for tn in range(NT):
                                        yf[dst + tn * 4:dst + tn * 4 + 4] = acc[wave, tn, tm, lane, :]
                        acc[:] = 0
                        cc.advance(dimg, dty, dtx, tiles_y, tiles_x)
    return y


@pytest.mark.parametrize("nimg,H,W,HALVES,NT,nwg,D,yblocks", [
    (2, 16, 32, 2, 4, 3, 2, 1),       # 64 -> 64 as one block (A3D_C3_SPLIT=0), grid not a multiple of 8
    (2, 8, 64, 2, 2, 2, 2, 2),        # 64 -> 64 as two 32-channel blocks (default)
    (3, 8, 96, 1, 2, 4, 3, 1),        # 32 -> 32, three tile columns, three steps in flight
    (1, 16, 64, 1, 4, 8, 2, 1),       # 32 -> 64, the XCD remap (grid a multiple of 8) with fewer tiles than workgroups
    (5, 8, 32, 1, 2, 2, 3, 1),        # one tile per image: the grid advance carries into the image digit
])
def test_lane_level_model_of_the_kernel_equals_conv2d(nimg, H, W, HALVES, NT, nwg, D, yblocks):
    rng = np.random.default_rng(nimg * 100 + W + NT)
    cin, ctot = 32 * HALVES, 16 * NT * yblocks
    x = rng.standard_normal((nimg, H, W, cin)).astype(np.float32)
    w = rng.standard_normal((ctot, 3, 3, cin)).astype(np.float32)
    y = model_kernel(x, w, HALVES, NT, nwg, D, yblocks)
    ref = F.conv2d(torch.from_numpy(x).permute(0, 3, 1, 2).double(), torch.from_numpy(w).permute(0, 3, 1, 2).double(),
                   padding=1).permute(0, 2, 3, 1).numpy()
    assert not np.isnan(y).any()                                    # every output element written exactly by the tiling
    assert np.abs(y - ref).max() < 1e-3 * max(1.0, np.abs(ref).max())


# ds_read_b128 is serviced in four groups of 16 lanes, one LDS cycle per group when the 16-byte slots (of 16 per 256-byte row of
# banks) are distinct (MI355X micro-architecture guide, LDS section)
B128_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
               list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
               [32, 33, 34, 35, 44, 45, 46, 47] + list(range(52, 60)),
               list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]


def b128_cycles(byte_addr_of_lane):
    total = 0
    for grp in B128_GROUPS:
        slots = {}
        for lane in grp:
            a = byte_addr_of_lane(lane)
            slots.setdefault((a // 16) % 16, set()).add(a)
        total += max(len(v) for v in slots.values())
    return total


def test_pixel_fragment_reads_are_bank_conflict_free_from_any_base():
    """Pixel fragments: lane (li, g) reads segment g of halo pixel base + li (16 consecutive pixels of one halo row, any tap shift):
    4 cycles = one per lane group = conflict-free with the column-keyed swizzle."""
    for hy in range(10):
        for hx0 in range(0, 34 - 15):
            cyc = b128_cycles(lambda l: 2 * xoff(hy * HW_ + hx0 + (l & 15), hx0 + (l & 15), l >> 4))
            assert cyc == 4, (hy, hx0, cyc)
    # the layouts that were rejected: plain 64-byte rows conflict 2-way from every base
    assert b128_cycles(lambda l: 64 * (5 + (l & 15)) + 16 * (l >> 4)) == 8


def test_weight_fragment_reads_are_bank_conflict_free():
    """Weight fragments read the PERMUTED rows (i >> 2) 4 NT + tn 4 + (i & 3) of a tile (conv3x3.hip, NT = 2 / 4) or
    (i >> 2) 16 + tn 4 + (i & 3) of a 64-row wave block (conv1x1.hip): conflict-free with the swizzle keyed on row bits 1 and 4;
    the round-4 keyings ((row >> 1) & 3 / plane_off) were 2-way conflicted for this row order."""
    def cycles(swz, rows_of_lane):
        def addr(l):
            row = rows_of_lane(l)
            return 2 * (row * 32 + (((l >> 4) ^ swz(row)) << 3))
        return b128_cycles(addr)
    new = lambda r: ((r >> 1) & 1) | (((r >> 4) & 1) << 1)
    for NT in (2, 4):
        for tn in range(NT):
            rows = lambda l, NT=NT, tn=tn: ((l & 15) >> 2) * (4 * NT) + tn * 4 + (l & 3)
            assert b128_cycles(lambda l: 2 * woff(rows(l), l >> 4)) == 4
            assert cycles(new, rows) == 4 and cycles(lambda r: (r >> 1) & 3, rows) == 8
    for wn in range(4):
        for tn in range(4):
            rows = lambda l, wn=wn, tn=tn: wn * 64 + ((l & 15) >> 2) * 16 + tn * 4 + (l & 3)
            assert cycles(new, rows) == 4 and cycles(lambda r: (0 - (r >> 2)) & 3, rows) == 8
