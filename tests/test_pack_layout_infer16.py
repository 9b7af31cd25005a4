"""CPU check of the 16-point inference stream (nerf_common.h P16F): numpy emulation of one wavefront of
field_fwd16_kernel (16 points, v_mfma_f32_16x16x32_bf16 lane maps: A lane (row = l&15, k-group = l>>4) pairs element
j with B lane (col = l&15, same k-group) element j; D lane (col = l&15, q = l>>4) register r = row 4q + r) with the
library's own gather table, against the oracle MLP.  hi parts are taken as the full fp64 weight and lo parts as zero,
so this checks the layout / slot algebra exactly; the hi/lo arithmetic itself is checked on the GPU."""
import numpy as np
import torch

import nerf_oracle as orc
import nerf_pytorch_amd as npa

LANE = np.arange(64)
PT, Q = LANE & 15, LANE >> 4
K16, K8 = 8192, 4096          # words per k-step (16 / 8 output blocks)
L1 = 2 * K16
L5 = L1 + 4 * 8 * K16
L6 = L5 + 10 * K16
FEAT = L6 + 2 * 8 * K16
VIEWS = FEAT + 8 * K16
END = VIEWS + 9 * K8


def hcol(s, q):
    return 16 * (s >> 2) + 4 * q + (s & 3)


def encslot(s, q):
    m, fn = s >> 1, s & 1
    i = q + 4 * m
    if i < 30:
        return 3 + (i // 3) * 6 + fn * 3 + (i % 3)
    if q == 2:
        return fn
    return 2 if fn == 0 else -1


def dirslot(s, q):
    if s < 6:
        m, fn = s >> 1, s & 1
        i = q + 4 * m
        return 3 + (i // 3) * 6 + fn * 3 + (i % 3)
    return q if q < 3 else -1


def layer(w16, base_word, nblk, bvals, acc):
    """acc[nb][r][lane]; bvals: list over k-steps of [8][64] per-lane B elements; w16: weight value per 16-bit element"""
    per = nblk * 2 * 64 * 8
    for s, b in enumerate(bvals):
        for nb in range(nblk):
            off = 2 * base_word + s * per + (nb * 2) * 512          # hi fragment of block nb
            A = w16[off:off + 512].reshape(64, 8)                   # [lane][j]
            Dm = np.zeros((16, 16))                                 # D[row][col] = sum_{kq, j} A[(row, kq)][j] * B[(col, kq)][j]
            for kq in range(4):
                Dm += A[kq * 16:(kq + 1) * 16] @ b[:, kq * 16:(kq + 1) * 16]
            for r in range(4):
                acc[nb, r] += Dm[4 * Q + r, PT]
    return acc


def lane_vals(acc, relu):
    return [np.maximum(acc[nb, r], 0.0) if relu else acc[nb, r].copy() for nb in range(acc.shape[0]) for r in range(4)]   # index 4*nb + r


def ksteps(vals, n):
    return [np.stack(vals[8 * s: 8 * s + 8]) for s in range(n)]


def lane_bias(bias, nblk):
    acc = np.zeros((nblk, 4, 64))
    for nb in range(nblk):
        for r in range(4):
            acc[nb, r] = bias[16 * nb + 4 * Q + r]
    return acc


def test_infer16_table_covers_every_weight_once_per_part():
    tab = npa.hip_backend.pack_table16()
    assert tab.shape == (2 * END,)
    hi = tab[(tab >= 0) & (tab % 2 == 0)] // 2
    lo = tab[(tab >= 0) & (tab % 2 == 1)] // 2
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


def test_infer16_forward_wave_emulation_matches_oracle():
    torch.manual_seed(0)
    Pc, _ = orc.scene_params()
    flat = np.concatenate([Pc[nm].double().numpy().reshape(-1) for nm, _ in orc.param_shapes()])
    # derived parameters of the folded feature layer, appended behind the canonical vector like in the pack tables
    Wv, Wf = Pc["views_linears.0.weight"].double().numpy(), Pc["feature_linear.weight"].double().numpy()
    b_fold = Wv[:, :256] @ Pc["feature_linear.bias"].double().numpy() + Pc["views_linears.0.bias"].double().numpy()
    flat = np.concatenate([flat, (Wv[:, :256] @ Wf).reshape(-1), b_fold])
    tab = npa.hip_backend.pack_table16()
    w16 = np.where((tab >= 0) & (tab % 2 == 0), flat[np.maximum(tab, 0) // 2], 0.0)
    P64 = {k: v.double() for k, v in Pc.items()}
    g = lambda nm: P64[nm].numpy()
    pts = torch.randn(16, 3, dtype=torch.float64) * 2.0
    dirs = torch.nn.functional.normalize(torch.randn(16, 3, dtype=torch.float64), dim=-1)
    enc, encd = orc.posenc(pts, 10).numpy(), orc.posenc(dirs, 4).numpy()
    want, hidden, feat, hv = orc.field_mlp(P64, torch.cat([torch.tensor(enc), torch.tensor(encd)], -1), return_hidden=True)
    pick = lambda table, slot, v: np.array([table[PT[l], slot(v, Q[l])] if slot(v, Q[l]) >= 0 else 0.0 for l in LANE])
    e = [pick(enc, encslot, v) for v in range(16)]
    dv = [pick(encd, dirslot, v) if v < 7 else np.zeros(64) for v in range(8)]

    acc = layer(w16, 0, 16, ksteps(e, 2), lane_bias(g("pts_linears.0.bias"), 16))
    h = lane_vals(acc, True)
    base = L1
    for l in range(1, 8):
        acc = lane_bias(g(f"pts_linears.{l}.bias"), 16)
        if l == 5:
            base = L5
            acc = layer(w16, base, 16, ksteps(e, 2), acc)
            base += 2 * K16
        if l == 6:
            base = L6
        acc = layer(w16, base, 16, ksteps(h, 8), acc)
        base += 8 * K16
        h = lane_vals(acc, True)
        ref = hidden[l].numpy()
        for i in range(64):
            np.testing.assert_allclose(h[i], ref[PT, hcol(i, Q)], rtol=1e-9, atol=1e-9)
    wa = g("alpha_linear.weight")[0]
    sigma = sum(h[i] * wa[hcol(i, Q)] for i in range(64)).reshape(4, 16).sum(0) + g("alpha_linear.bias")[0]
    np.testing.assert_allclose(sigma, want[:, 3].numpy(), rtol=1e-9, atol=1e-9)
    # the view branch runs on the trunk output with the folded W' / b' (the feature_linear region of the stream is skipped)
    acc = layer(w16, VIEWS, 8, ksteps(h, 8) + ksteps(dv, 1), lane_bias(b_fold, 8))
    hvr = lane_vals(acc, True)
    wr = g("rgb_linear.weight")
    for c in range(3):
        tot = sum(hvr[i] * wr[c, hcol(i, Q)] for i in range(32)).reshape(4, 16).sum(0) + g("rgb_linear.bias")[c]
        np.testing.assert_allclose(tot, want[:, c].numpy(), rtol=1e-9, atol=1e-9)
