"""CPU check of the split-bf16 repack: numpy emulation of one wavefront of field_fwd3_kernel (32 points,
v_mfma_f32_32x32x16_bf16 lane maps: A lane (row = l&31, half = l>>5) pairs element j with B lane
(col = l&31, same half) element j; D lane (col = l&31, half) register r = row d32row(r, half)) with the
library's own gather table, against the oracle MLP.  hi parts are taken as the full fp64 weight and lo parts
as zero, so this checks the layout / slot algebra exactly; the hi/lo arithmetic itself is checked on the GPU."""
import numpy as np
import torch

import nerf_oracle as orc
import nerf_pytorch_amd as npa

LANE = np.arange(64)
PT, HALF = LANE & 31, LANE >> 5
K8, K4 = 4096, 2048          # words per k-step (8 / 4 output blocks)
P3F_L0 = 0
P3F_L1 = 4 * K8
P3F_L5 = P3F_L1 + 64 * K8
P3F_L6 = P3F_L5 + 20 * K8
P3F_FEAT = P3F_L6 + 32 * K8
P3F_VIEWS = P3F_FEAT + 16 * K8
P3F_END = P3F_VIEWS + 18 * K4


def d32row(r, half):
    return (r & 3) + 8 * (r >> 2) + 4 * half


def enc3slot(v, half):
    if half == 0:
        i, fn = v >> 1, v & 1
        return 3 + (i // 3) * 6 + fn * 3 + (i % 3)
    if v < 28:
        i, fn = 16 + (v >> 1), v & 1
        return 3 + (i // 3) * 6 + fn * 3 + (i % 3)
    return v - 28 if v < 31 else -1


def dir3slot(v, half):
    if half == 0:
        i, fn = v >> 1, v & 1
        return 3 + (i // 3) * 6 + fn * 3 + (i % 3)
    if v < 8:
        i, fn = 8 + (v >> 1), v & 1
        return 3 + (i // 3) * 6 + fn * 3 + (i % 3)
    return v - 8 if v < 11 else -1


def layer(w16, base_word, nblk, bvals, acc):
    """acc[nb][r][lane]; bvals: list over k-steps of [8][64] per-lane B elements; w16: weight value per 16-bit element"""
    per = nblk * 2 * 64 * 8
    for s, b in enumerate(bvals):
        for nb in range(nblk):
            off = 2 * base_word + s * per + (nb * 2) * 512          # hi fragment of block nb
            A = w16[off:off + 512].reshape(64, 8)                   # [lane][j]
            # D[row][col] = sum_{half, j} A[(row, half)][j] * B[(col, half)][j]
            Dm = np.zeros((32, 32))
            for hf in range(2):
                Dm += A[hf * 32:(hf + 1) * 32] @ b[:, hf * 32:(hf + 1) * 32]
            for r in range(16):
                acc[nb, r] += Dm[d32row(r, HALF), PT]
    return acc


def lane_vals_from_acc(acc, relu):
    out = []
    for nb in range(acc.shape[0]):
        for r in range(16):
            out.append(np.maximum(acc[nb, r], 0.0) if relu else acc[nb, r].copy())
    return out          # index 16*nb + r


def ksteps(vals, n):
    return [np.stack(vals[8 * s: 8 * s + 8]) for s in range(n)]


def lane_bias(bias, nblk):
    acc = np.zeros((nblk, 16, 64))
    for nb in range(nblk):
        for r in range(16):
            acc[nb, r] = bias[32 * nb + d32row(r, HALF)]
    return acc


def test_bf16x3_table_covers_every_weight_once_per_part():
    tab = npa.hip_backend.pack_table3()
    fwd = tab[:2 * P3F_END]
    hi = fwd[(fwd >= 0) & (fwd % 2 == 0)] // 2
    lo = fwd[(fwd >= 0) & (fwd % 2 == 1)] // 2
    expect = []
    for nm, off, shp in npa.hip_backend.param_table():
        if nm.endswith("weight") and not nm.startswith(("alpha", "rgb")):
            idx = np.arange(off, off + shp[0] * shp[1])
            if nm == "views_linears.0.weight":      # its feature columns are folded with feature_linear into the derived W'
                idx = idx.reshape(shp)[:, 256:].reshape(-1)
            expect.append(idx)
    expect.append(np.arange(595844, 595844 + 128 * 256))       # W' = Wv[:, :256] Wf (csrc/nerf_common.h, DERIVED_WVF)
    expect = np.sort(np.concatenate(expect))
    assert np.array_equal(np.sort(hi), expect) and np.array_equal(np.sort(lo), expect)


def test_bf16x3_forward_wave_emulation_matches_oracle():
    torch.manual_seed(0)
    Pc, _ = orc.scene_params()
    flat = np.concatenate([Pc[nm].double().numpy().reshape(-1) for nm, _ in orc.param_shapes()])
    Wv, Wf = Pc["views_linears.0.weight"].double().numpy(), Pc["feature_linear.weight"].double().numpy()
    b_fold = Wv[:, :256] @ Pc["feature_linear.bias"].double().numpy() + Pc["views_linears.0.bias"].double().numpy()
    flat = np.concatenate([flat, (Wv[:, :256] @ Wf).reshape(-1), b_fold])       # derived W', b' (folded feature layer)
    tab = npa.hip_backend.pack_table3()
    w16 = np.where((tab >= 0) & (tab % 2 == 0), flat[np.maximum(tab, 0) // 2], 0.0)
    P64 = {k: v.double() for k, v in Pc.items()}
    g = lambda nm: P64[nm].numpy()
    pts = torch.randn(32, 3, dtype=torch.float64) * 2.0
    dirs = torch.nn.functional.normalize(torch.randn(32, 3, dtype=torch.float64), dim=-1)
    enc, encd = orc.posenc(pts, 10).numpy(), orc.posenc(dirs, 4).numpy()
    want, hidden, feat, hv = orc.field_mlp(P64, torch.cat([torch.tensor(enc), torch.tensor(encd)], -1), return_hidden=True)
    e = [np.array([enc[PT[l], enc3slot(v, HALF[l])] if enc3slot(v, HALF[l]) >= 0 else 0.0 for l in LANE]) for v in range(32)]
    dv = [np.array([encd[PT[l], dir3slot(v, HALF[l])] if dir3slot(v, HALF[l]) >= 0 else 0.0 for l in LANE]) for v in range(16)]

    acc = layer(w16, P3F_L0, 8, ksteps(e, 4), lane_bias(g("pts_linears.0.bias"), 8))
    h = lane_vals_from_acc(acc, True)
    base = P3F_L1
    for l in range(1, 8):
        acc = lane_bias(g(f"pts_linears.{l}.bias"), 8)
        if l == 5:
            base = P3F_L5
            acc = layer(w16, base, 8, ksteps(e, 4), acc)
            base += 4 * K8
        if l == 6:
            base = P3F_L6
        acc = layer(w16, base, 8, ksteps(h, 16), acc)
        base += 16 * K8
        h = lane_vals_from_acc(acc, True)
        ref = hidden[l].numpy()
        for i in range(128):
            np.testing.assert_allclose(h[i], ref[PT, 32 * (i >> 4) + d32row(i & 15, HALF)], rtol=1e-9, atol=1e-9)
    wa = g("alpha_linear.weight")[0]
    sigma = sum(h[i] * wa[32 * (i >> 4) + d32row(i & 15, HALF)] for i in range(128))
    sigma = sigma.reshape(2, 32).sum(0) + g("alpha_linear.bias")[0]
    np.testing.assert_allclose(sigma, want[:, 3].numpy(), rtol=1e-9, atol=1e-9)
    # the view branch runs on the trunk output with the folded W' / b' (the feature_linear region of the stream is skipped)
    acc = layer(w16, P3F_VIEWS, 4, ksteps(h, 16) + ksteps(dv, 2), lane_bias(b_fold, 4))
    hvr = lane_vals_from_acc(acc, True)
    wr = g("rgb_linear.weight")
    for c in range(3):
        tot = sum(hvr[i] * wr[c, 32 * (i >> 4) + d32row(i & 15, HALF)] for i in range(64))
        tot = tot.reshape(2, 32).sum(0) + g("rgb_linear.bias")[c]
        np.testing.assert_allclose(tot, want[:, c].numpy(), rtol=1e-9, atol=1e-9)
