"""Index model of the dropout epilogues (csrc/linear.hip: linear_fwd_kernel<*, true>, add_ln_bwd_rows_kernel<32, true>), on the CPU.

The separate launch (csrc/dropout.hip) draws ONE Philox block per eight consecutive elements of the flat output: block (m * N + n) >> 3,
bit (m * N + n) & 7.  The folded form must read exactly those blocks from inside the producing kernel's own thread layout:
  * the MFMA epilogue owns, per 16 x 16 tile, column n = lane & 15 and rows (lane >> 4) * 4 + r; lane (g, li) generates the block of row
    g*4 + (li & 3), column half (li >> 2) & 1, and the four rows of a lane's column fetch theirs with a wave shuffle from lane
    g*16 + r + 4 * (li >> 3);
  * the LayerNorm backward owns four consecutive channels per lane = one half of a block.
Pure index arithmetic (the GPU tests then check the bits: tests/test_dropout_gpu.py)."""
import itertools

import pytest


@pytest.mark.parametrize("N", [8, 120, 480, 128])
def test_linear_epilogue_fetches_the_block_of_every_element(N):
    for m0, n0 in itertools.product((0, 64, 1088), (0, 64)):
        if n0 >= N and N > 64:
            continue
        for wave, nt in itertools.product(range(4), range(4)):
            generated = {}
            for lane in range(64):
                li, g = lane & 15, lane >> 4
                generated[lane] = ((m0 + wave * 16 + g * 4 + (li & 3)) * N + (n0 + nt * 16 + ((li >> 2) & 1) * 8)) >> 3
            for lane in range(64):
                li, g = lane & 15, lane >> 4
                n = n0 + nt * 16 + li
                if n >= N:
                    continue                                        # the kernel skips the store, not the shuffle
                for r in range(4):
                    m = m0 + wave * 16 + g * 4 + r
                    src = (lane & 48) + r + 4 * (li >> 3)
                    flat = m * N + n
                    assert generated[src] == flat >> 3 and (li & 7) == flat & 7, (N, m, n, lane, r)


@pytest.mark.parametrize("E", [72, 120, 128])
def test_layernorm_backward_lane_owns_half_a_block(E):
    for m in (0, 1, 7, 1099, 67583):
        for lane in range(32):
            e0 = 4 * lane
            if e0 >= E:
                continue
            idx = m * E + e0
            for j in range(4):
                flat = m * E + e0 + j
                assert flat >> 3 == idx >> 3 and flat & 7 == (idx & 4) + j
