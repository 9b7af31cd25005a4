"""CPU check of the TRANSPOSED fragment streams of the split datapaths' delta chain (csrc/field_bwd_ring.hip): numpy emulation of one
wavefront (32 points, v_mfma_f32_32x32x16_{bf16,f16} lane maps: A lane (row = l&31, half = l>>5) pairs element j with B lane
(col = l&31, same half) element j; D lane (col = l&31, half) register r = row d32row(r, half)) with the library's own gather table,
against W^T delta computed directly.  hi parts are taken as the full fp64 weight and lo parts as zero, so this checks the layout / slot
algebra exactly; the hi/lo arithmetic itself is checked on the GPU (fp64 autograd comparisons)."""
import numpy as np
import torch

import nerf_oracle as orc
import nerf_pytorch_amd as npa

LANE = np.arange(64)
PT, HALF = LANE & 31, LANE >> 5
K8 = 4096                    # words per k-step (8 output blocks x (hi, lo) x 64 lanes x 16 B)
P3B_VIEWS = 0
P3B_FEAT = P3B_VIEWS + 8 * K8
P3B_L7 = P3B_FEAT + 16 * K8
P3B_END = P3B_L7 + 7 * 16 * K8


def d32row(r, half):
    return (r & 3) + 8 * (r >> 2) + 4 * half


def feature_of(i, half):
    """feature held in lane value i (0..127) of lane half `half`"""
    return 32 * (i >> 4) + d32row(i & 15, half)


def layer(w16, base_word, bvals):
    """acc[nb][r][lane] of one contraction: bvals = list over k-steps of [8][64] per-lane B elements; w16 = weight value per 16-bit
    element of the stream"""
    acc = np.zeros((8, 16, 64))
    per = 8 * 2 * 64 * 8
    for s, b in enumerate(bvals):
        for nb in range(8):
            off = 2 * base_word + s * per + (nb * 2) * 512          # hi fragment of block nb
            A = w16[off:off + 512].reshape(64, 8)                   # [lane][j]
            Dm = np.zeros((32, 32))                                 # D[row][col] = sum_{half, j} A[(row, half)][j] * B[(col, half)][j]
            for hf in range(2):
                Dm += A[hf * 32:(hf + 1) * 32] @ b[:, hf * 32:(hf + 1) * 32]
            for r in range(16):
                acc[nb, r] += Dm[d32row(r, HALF), PT]
    return acc


def lane_values(mat):
    """[feature][point] -> the lane values i = 0 .. F/2-1 of the delta chain: value i of lane (pt, half) = mat[feature_of(i, half)][pt]"""
    n = mat.shape[0] // 2
    return [mat[feature_of(i, HALF), PT] for i in range(n)]


def ksteps(vals, n):
    return [np.stack(vals[8 * s: 8 * s + 8]) for s in range(n)]


def test_transposed_table_covers_every_weight_once_per_part():
    tab = npa.hip_backend.pack_table3()
    assert tab.size == 2 * P3B_END
    hi = tab[(tab >= 0) & (tab % 2 == 0)] // 2
    lo = tab[(tab >= 0) & (tab % 2 == 1)] // 2
    expect = []
    for nm, off, shp in npa.hip_backend.param_table():
        if nm.startswith("pts_linears") and nm.endswith("weight") and not nm.startswith("pts_linears.0."):
            idx = np.arange(off, off + shp[0] * shp[1]).reshape(shp)
            if nm == "pts_linears.5.weight":        # the skip connection's encoding columns receive no delta
                idx = idx[:, 63:]
            expect.append(idx.reshape(-1))
        if nm == "feature_linear.weight":           # packed for the stream's fixed shape, skipped by the kernel (folded layer)
            expect.append(np.arange(off, off + shp[0] * shp[1]))
    expect.append(np.arange(595844, 595844 + 128 * 256))       # W' = Wv[:, :256] Wf (csrc/nerf_common.h, DERIVED_WVF)
    expect = np.sort(np.concatenate(expect))
    assert np.array_equal(np.sort(hi), expect) and np.array_equal(np.sort(lo), expect)


def test_delta_chain_wave_emulation_matches_the_transposed_products():
    torch.manual_seed(0)
    Pc, _ = orc.scene_params()
    flat = np.concatenate([Pc[nm].double().numpy().reshape(-1) for nm, _ in orc.param_shapes()])
    Wv, Wf = Pc["views_linears.0.weight"].double().numpy(), Pc["feature_linear.weight"].double().numpy()
    Wfold = Wv[:, :256] @ Wf                                                    # [128][256]
    flat = np.concatenate([flat, Wfold.reshape(-1), np.zeros(128)])            # derived W', b'
    tab = npa.hip_backend.pack_table3()
    w16 = np.where((tab >= 0) & (tab % 2 == 0), flat[np.maximum(tab, 0) // 2], 0.0)
    rng = np.random.RandomState(3)
    # view branch: delta of the trunk output (before its ReLU mask) = W'^T delta_hv, delta_hv [128][32 points]
    d_hv = rng.randn(128, 32)
    acc = layer(w16, P3B_VIEWS, ksteps(lane_values(d_hv), 8))
    want = Wfold.T @ d_hv                                                       # [256][32]
    for i in range(128):
        np.testing.assert_allclose(acc[i >> 4, i & 15], want[feature_of(i, HALF), PT], rtol=1e-10, atol=1e-10)
    # trunk: delta_{l-1} (before the mask) = W_l^T delta_l for l = 7 .. 1; layer 5's weight has 63 leading encoding columns
    for t, l in enumerate(range(7, 0, -1)):
        Wl = Pc[f"pts_linears.{l}.weight"].double().numpy()
        if l == 5:
            Wl = Wl[:, 63:]
        d_l = rng.randn(256, 32)
        acc = layer(w16, P3B_L7 + t * 16 * K8, ksteps(lane_values(d_l), 16))
        want = Wl.T @ d_l
        for i in range(128):
            np.testing.assert_allclose(acc[i >> 4, i & 15], want[feature_of(i, HALF), PT], rtol=1e-10, atol=1e-10, err_msg=f"layer {l}")
