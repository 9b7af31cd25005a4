"""CPU: the paired bf16 row stores of field_fwd16_kernel<2> and of the bf16-delta dgrad (csrc/field_fwd_bf16.hip
store_pair, csrc/field_device_bf16.h store_tile3h_pair), emulated lane by lane in numpy: every lane packs its two values,
swaps the word with its neighbour (DPP quad_perm [1,0,3,2]), selects with v_perm_b32, and stores ONE dword.  Checked:
each (row, point) of the tile is written exactly once with the right value, one store instruction covers two full
128-byte lines, and hip_backend.saved_rows() inverts the layout."""
import numpy as np
import torch

import nerf_pytorch_amd as npa


def v_perm_b32(s0, s1, sel):
    """D.byte[i] = {s0, s1}.byte[sel.byte[i]] with bytes 0-3 = s1, 4-7 = s0 (the selector values used here are 0..7)."""
    pool = np.concatenate([s1.view(np.uint8).reshape(-1, 4), s0.view(np.uint8).reshape(-1, 4)], 1)     # [lanes, 8]
    idx = sel.view(np.uint8).reshape(-1, 4)
    return np.take_along_axis(pool, idx, 1).copy().view(np.uint32).reshape(-1)


def pack_bf16x2(lo, hi):
    b = lambda x: (torch.tensor(x, dtype=torch.float32).bfloat16().view(torch.int16).numpy().astype(np.uint32) & 0xffff)
    return b(lo) | (b(hi) << 16)


def pair_words(v0, v1, lane):
    own = pack_bf16x2(v0, v1)
    nbr = own[lane ^ 1]                                             # v_mov_b32_dpp quad_perm:[1,0,3,2]
    sel = np.where(lane & 1, 0x03020706, 0x05040100).astype(np.uint32)
    return v_perm_b32(nbr, own, sel)


def test_fwd16_paired_rows_fill_a_16_point_tile_in_row16h_order():
    F = 256
    lane = np.arange(64)
    pt, q = lane & 15, lane >> 4
    rng = np.random.RandomState(0)
    val = rng.randn(16, F).astype(np.float32)                        # [point][feature] of one wave's tile
    tile = np.full(F * 8, 0xdeadbeef, dtype=np.uint32)               # [F rows][8 dwords] = 16 points x 2 B per row
    writes = np.zeros(F * 8, dtype=int)
    lane_pair_off = (2 * q + (lane & 1)) * 8 + (pt >> 1)
    for nb in range(16):
        for r0 in (0, 2):                                            # one store instruction
            f0 = 16 * nb + 4 * q + r0                                # the lane's features r0, r0 + 1 of block nb
            word = pair_words(val[pt, f0], val[pt, f0 + 1], lane)
            addr = (16 * nb + 4 * r0) * 8 + lane_pair_off
            tile[addr] = word
            writes[addr] += 1
            # one instruction = 8 consecutive rows of 32 B = 256 contiguous bytes = two full 128-byte lines
            assert sorted(addr) == list(range(addr.min(), addr.min() + 64)) and (addr.min() * 4) % 128 == 0
    assert (writes == 1).all()
    got = torch.tensor(tile.view(np.int16)).view(torch.bfloat16).float().reshape(F, 16)      # [row][point]
    row16h = npa.hip_backend._row16h(torch.arange(F))
    want = torch.tensor(val).bfloat16().float()                                             # [point][feature]
    assert torch.equal(got[row16h].T, want)
    # and the host-side view of a whole region made of such tiles
    P = 48
    reg = npa.hip_backend.buffer_regions(P, 1, True)
    full = torch.zeros(reg["total"])                                                        # floats holding 2-byte elements
    vals = torch.randn(P, F).bfloat16()
    p, f = torch.meshgrid(torch.arange(P), torch.arange(F), indexing="ij")
    full.view(torch.bfloat16)[2 * reg["h3"] + (p // 16) * F * 16 + npa.hip_backend._row16h(f) * 16 + p % 16] = vals
    assert torch.equal(npa.hip_backend.saved_rows(full, P, 1, "h3", "bf16x3"), vals.float())


def test_dgrad_paired_deltas_fill_a_32_point_tile():
    F = 256
    lane = np.arange(64)
    pt, half = lane & 31, lane >> 5
    rng = np.random.RandomState(1)
    val = rng.randn(32, F).astype(np.float32)                        # [point][feature]
    tile = np.zeros(F * 16, dtype=np.uint32)                         # [F rows][16 dwords] = 32 points x 2 B per row
    writes = np.zeros(F * 16, dtype=int)
    base = (half * 4 + (lane & 1)) * 16 + (pt >> 1)
    d32row = lambda r, h: (r & 3) + 8 * (r >> 2) + 4 * h
    for ob in range(8):
        for r in range(0, 16, 2):                                    # lane value 16*ob + r is feature 32*ob + d32row(r, half)
            f0 = 32 * ob + d32row(r, half)
            assert (32 * ob + d32row(r + 1, half) == f0 + 1).all()   # (r, r+1) are adjacent rows of the tile
            word = pair_words(val[pt, f0], val[pt, f0 + 1], lane)
            addr = (32 * ob + (r & 3) + 8 * (r >> 2)) * 16 + base
            tile[addr] = word
            writes[addr] += 1
            for h in (0, 1):                                         # each half-wave writes rows R, R+1 = one full 128-byte line
                a = np.sort(addr[half == h])
                assert list(a) == list(range(a[0], a[0] + 32)) and (a[0] * 4) % 128 == 0
    assert (writes == 1).all()
    got = torch.tensor(tile.view(np.int16)).view(torch.bfloat16).float().reshape(F, 32)      # [feature][point]
    assert torch.equal(got.T, torch.tensor(val).bfloat16().float())
