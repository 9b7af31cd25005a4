"""CPU check of the fragment repack + slot algebra (no GPU needed).

Emulates, in numpy, exactly the data flow one wavefront of field_fwd_kernel /
field_dgrad_kernel executes -- packed A-fragments from the library's own gather
table, the v_mfma_f32_16x16x4_f32 lane maps (A[i=l&15][k=l>>4], B[k=l>>4][j=l&15],
D[i=4*(l>>4)+r][j=l&15]) and the k-slot permutations -- and compares the result
with the oracle MLP (forward) and with autograd (backward deltas).  A wrong
column permutation, layer order or chunk offset in pack.hip / nerf_common.h
fails here before any GPU time is spent.
"""
import numpy as np
import torch

import nerf_oracle as orc
import nerf_pytorch_amd as npa

LANE = np.arange(64)
PT = LANE & 15          # point of the lane
Q = LANE >> 4           # lane quarter

# constants mirrored from csrc/nerf_common.h
KSTEP16, KSTEP8 = 1024, 512
FWD_L0 = 0
FWD_L1 = FWD_L0 + 16 * KSTEP16
FWD_L5 = FWD_L1 + 4 * 64 * KSTEP16
FWD_L6 = FWD_L5 + 80 * KSTEP16
FWD_FEAT = FWD_L6 + 2 * 64 * KSTEP16
FWD_VIEWS = FWD_FEAT + 64 * KSTEP16
FWD_END = FWD_VIEWS + 71 * KSTEP8
BWD_VIEWS = FWD_END
BWD_FEAT = BWD_VIEWS + 32 * KSTEP16
BWD_L7 = BWD_FEAT + 64 * KSTEP16
BWD_END = BWD_L7 + 7 * 64 * KSTEP16


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


def mfma(a, b, acc):
    """acc[r][lane] += sum_k A[i=4*(lane>>4)+r][k] * B[k][j=lane&15], A/B given per lane."""
    A = a.reshape(4, 16)        # [k][i]
    B = b.reshape(4, 16)        # [k][j]
    Dm = A.T @ B                # [i][j]
    for r in range(4):
        acc[r] += Dm[4 * Q + r, PT]
    return acc


def layer(packed, base, nb_blocks, bregs, acc):
    """acc[nb][r][lane]; bregs: list over k-steps of per-lane B registers."""
    G = nb_blocks // 4
    per = nb_blocks * 64
    for s, b in enumerate(bregs):
        for g in range(G):
            frag = packed[base + s * per + g * 256: base + s * per + (g + 1) * 256].reshape(64, 4)
            for j in range(4):
                mfma(frag[:, j], b, acc[4 * g + j])
    return acc


def lane_bias(bias, nb_blocks):
    acc = np.zeros((nb_blocks, 4, 64), dtype=np.float64)
    for nb in range(nb_blocks):
        for r in range(4):
            acc[nb, r] = bias[16 * nb + 4 * Q + r]
    return acc


def regs_from_acc(acc, relu):
    out = []
    for nb in range(acc.shape[0]):
        for r in range(4):
            v = acc[nb, r]
            out.append(np.maximum(v, 0.0) if relu else v.copy())
    return out


def gather_packed(P_canon_flat):
    tab = npa.hip_backend.pack_table()
    packed = np.where(tab >= 0, P_canon_flat[np.maximum(tab, 0)], 0.0)
    return packed, tab


def flat_from_params(P):
    return np.concatenate([P[nm].double().numpy().reshape(-1) for nm, _ in orc.param_shapes()])


def test_pack_table_is_a_permutation_of_the_weights():
    tab = npa.hip_backend.pack_table()
    assert tab.shape[0] == npa.hip_backend.lib().nerf_packed_floats()
    fwd = tab[:FWD_END]
    used = fwd[fwd >= 0]
    # forward stream holds every weight matrix entry except the two VALU heads exactly once
    names = dict((nm, (off, shp)) for nm, off, shp in npa.hip_backend.param_table())
    expect = []
    for nm, (off, shp) in names.items():
        if nm.endswith("weight") and not nm.startswith(("alpha", "rgb")):
            expect.append(np.arange(off, off + shp[0] * shp[1]))
    expect = np.sort(np.concatenate(expect))
    assert np.array_equal(np.sort(used), expect)
    assert (fwd < 0).sum() == FWD_END - expect.size


def test_forward_wave_emulation_matches_oracle():
    torch.manual_seed(0)
    Pc, _ = orc.scene_params()
    packed, _ = gather_packed(flat_from_params(Pc))
    pts = torch.randn(16, 3, dtype=torch.float64) * 2.0
    dirs = torch.nn.functional.normalize(torch.randn(16, 3, dtype=torch.float64), dim=-1)
    P64 = {k: v.double() for k, v in Pc.items()}
    enc = orc.posenc(pts, 10).numpy()
    encd = orc.posenc(dirs, 4).numpy()
    want, hidden, feat, hv = orc.field_mlp(P64, torch.cat([torch.tensor(enc), torch.tensor(encd)], -1), return_hidden=True)

    e = [np.array([enc[PT[l], encslot(s, Q[l])] if encslot(s, Q[l]) >= 0 else 0.0 for l in LANE]) for s in range(16)]
    v = [np.array([encd[PT[l], dirslot(s, Q[l])] if dirslot(s, Q[l]) >= 0 else 0.0 for l in LANE]) for s in range(7)]
    g = lambda nm: P64[nm].numpy()

    acc = layer(packed, FWD_L0, 16, e, lane_bias(g("pts_linears.0.bias"), 16))
    h = regs_from_acc(acc, True)
    base = FWD_L1
    for l in range(1, 8):
        acc = lane_bias(g(f"pts_linears.{l}.bias"), 16)
        if l == 5:
            base = FWD_L5
            acc = layer(packed, base, 16, e, acc)
            base += 16 * KSTEP16
        if l == 6:
            base = FWD_L6
        acc = layer(packed, base, 16, h, acc)
        base += 64 * KSTEP16
        h = regs_from_acc(acc, True)
        # lane layout claim: register 4*nb+r of lane (p,q) is feature 16*nb+4*q+r of point p
        got = np.stack([h[4 * nb + r] for nb in range(16) for r in range(4)])      # [64 regs][64 lanes]
        ref = hidden[l].numpy()
        for reg in range(64):
            nb, r = reg // 4, reg % 4
            np.testing.assert_allclose(got[reg], ref[PT, 16 * nb + 4 * Q + r], rtol=1e-9, atol=1e-9)
    # density head
    wa = g("alpha_linear.weight")[0]
    sigma = np.zeros(64)
    for reg in range(64):
        nb, r = reg // 4, reg % 4
        sigma += h[reg] * wa[16 * nb + 4 * Q + r]
    sigma = sigma.reshape(4, 16).sum(0) + g("alpha_linear.bias")[0]
    np.testing.assert_allclose(sigma, want[:, 3].numpy(), rtol=1e-9, atol=1e-9)
    # feature + view branch + rgb
    acc = layer(packed, FWD_FEAT, 16, h, lane_bias(g("feature_linear.bias"), 16))
    f = regs_from_acc(acc, False)
    acc = layer(packed, FWD_VIEWS, 8, f + v, lane_bias(g("views_linears.0.bias"), 8))
    hvr = regs_from_acc(acc, True)
    wr = g("rgb_linear.weight")
    for c in range(3):
        tot = np.zeros(64)
        for reg in range(32):
            nb, r = reg // 4, reg % 4
            tot += hvr[reg] * wr[c, 16 * nb + 4 * Q + r]
        tot = tot.reshape(4, 16).sum(0) + g("rgb_linear.bias")[c]
        np.testing.assert_allclose(tot, want[:, c].numpy(), rtol=1e-9, atol=1e-9)


def test_backward_wave_emulation_matches_autograd():
    torch.manual_seed(1)
    Pc, _ = orc.scene_params()
    packed, _ = gather_packed(flat_from_params(Pc))
    P64 = {k: v.double() for k, v in Pc.items()}
    pts = torch.randn(16, 3, dtype=torch.float64) * 2.0
    dirs = torch.nn.functional.normalize(torch.randn(16, 3, dtype=torch.float64), dim=-1)
    feats = torch.cat([orc.posenc(pts, 10), orc.posenc(dirs, 4)], -1)
    # autograd reference deltas = gradient w.r.t. biases per point (bias enters pre-activation additively)
    lin = torch.nn.functional.linear
    xyz, dd = feats[:, :63], feats[:, 63:]
    pre = []
    h = xyz
    for i in range(8):
        z = lin(h, P64[f"pts_linears.{i}.weight"], P64[f"pts_linears.{i}.bias"]).requires_grad_(True) if False else None
        a = lin(h, P64[f"pts_linears.{i}.weight"], P64[f"pts_linears.{i}.bias"])
        a = a.detach().requires_grad_(True) if False else a
        pre.append(a)
        h = torch.relu(a)
        if i == 4:
            h = torch.cat([xyz, h], -1)
    # simpler: recompute with leaf pre-activations via hooks
    pres = {}

    def run():
        hh = xyz
        outs = {}
        for i in range(8):
            a = lin(hh, P64[f"pts_linears.{i}.weight"], P64[f"pts_linears.{i}.bias"])
            a.retain_grad()
            outs[f"h{i}"] = a
            hh = torch.relu(a)
            if i == 4:
                hh = torch.cat([xyz, hh], -1)
        sigma = lin(hh, P64["alpha_linear.weight"], P64["alpha_linear.bias"])
        ft = lin(hh, P64["feature_linear.weight"], P64["feature_linear.bias"])
        ft.retain_grad()
        outs["feat"] = ft
        av = lin(torch.cat([ft, dd], -1), P64["views_linears.0.weight"], P64["views_linears.0.bias"])
        av.retain_grad()
        outs["hv"] = av
        rgb = lin(torch.relu(av), P64["rgb_linear.weight"], P64["rgb_linear.bias"])
        return torch.cat([rgb, sigma], -1), outs

    for k in P64:
        P64[k].requires_grad_(True)
    out, outs = run()
    d_raw = torch.randn(16, 4, dtype=torch.float64)
    (out * d_raw).sum().backward()

    g = lambda nm: P64[nm].detach().numpy()
    dr = d_raw.numpy()
    mask = lambda t: (t.detach().numpy() > 0)
    # rgb^T + relu mask
    wr = g("rgb_linear.weight")
    dhv = []
    for reg in range(32):
        nb, r = reg // 4, reg % 4
        col = 16 * nb + 4 * Q + r
        val = sum(dr[PT, c] * wr[c, col] for c in range(3))
        dhv.append(np.where(mask(outs["hv"])[PT, col], val, 0.0))
    for reg in range(32):
        nb, r = reg // 4, reg % 4
        np.testing.assert_allclose(dhv[reg], outs["hv"].grad.numpy()[PT, 16 * nb + 4 * Q + r], rtol=1e-9, atol=1e-12)
    acc = layer(packed, BWD_VIEWS, 16, dhv, np.zeros((16, 4, 64)))
    d = regs_from_acc(acc, False)
    for reg in range(64):
        nb, r = reg // 4, reg % 4
        np.testing.assert_allclose(d[reg], outs["feat"].grad.numpy()[PT, 16 * nb + 4 * Q + r], rtol=1e-9, atol=1e-12)
    wa = g("alpha_linear.weight")[0]
    acc = np.zeros((16, 4, 64))
    for nb in range(16):
        for r in range(4):
            acc[nb, r] = dr[PT, 3] * wa[16 * nb + 4 * Q + r]
    acc = layer(packed, BWD_FEAT, 16, d, acc)

    def masked(acc, key):
        m = mask(outs[key])
        return [np.where(m[PT, 16 * (reg // 4) + 4 * Q + (reg % 4)], acc[reg // 4, reg % 4], 0.0) for reg in range(64)]
    d = masked(acc, "h7")
    base = BWD_L7
    for l in range(7, 0, -1):
        ref = outs[f"h{l}"].grad.numpy()
        for reg in range(64):
            nb, r = reg // 4, reg % 4
            np.testing.assert_allclose(d[reg], ref[PT, 16 * nb + 4 * Q + r], rtol=1e-8, atol=1e-11)
        acc = layer(packed, base, 16, d, np.zeros((16, 4, 64)))
        base += 64 * KSTEP16
        d = masked(acc, f"h{l - 1}")
    ref = outs["h0"].grad.numpy()
    for reg in range(64):
        nb, r = reg // 4, reg % 4
        np.testing.assert_allclose(d[reg], ref[PT, 16 * nb + 4 * Q + r], rtol=1e-8, atol=1e-11)
    assert base == BWD_END
