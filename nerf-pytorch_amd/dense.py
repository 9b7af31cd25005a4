"""The reference's NeRF for ANY constructor arguments (run_nerf_helpers.py:67-119) and its render_rays (run_nerf.py:308-418).

The fused kernels cover the architecture every BASELINE config uses (D=8, W=256, 10 / 4 frequencies, view directions).
Everything else the reference's command line can build -- --netdepth / --netwidth (run_nerf.py:435-442), --multires /
--multires_views, --i_embed -1 (helpers:48-50), use_viewdirs=False with its output_linear head (helpers:93-94, :117) --
runs here, layer by layer: plain library SGEMMs (rocBLAS, exact fp32) behind the C ABI (csrc/dense.hip: nerf_dense_fwd /
_dgrad / _wgrad) with HIP epilogues, the network input built by one kernel (nerf_build_inputs), sampling and compositing by the
same per-ray kernels as the fused path.  Same state_dict keys and registration order as the reference, so checkpoints and
torch.optim.Adam carry over.  This is the reference's arithmetic at library-GEMM speed, not the MI355X fast path.
"""
import torch
import torch.nn as nn

from . import hip_backend as hb


def _cols(t, lo, width):
    """(data pointer of column `lo`, leading dimension) of a row-major 2-D tensor"""
    assert t.dim() == 2 and t.is_contiguous() and t.dtype == torch.float32 and 0 <= lo and lo + width <= t.shape[1]
    return t.data_ptr() + 4 * lo, t.shape[1]


def _fwd(x, x_lo, K, w, w_lo, bias, y, y_lo, N, accumulate=False, relu=False):
    xp, ldx = _cols(x, x_lo, K)
    wp, ldw = _cols(w, w_lo, K)
    yp, ldy = _cols(y, y_lo, N)
    hb._check(hb.lib().nerf_dense_fwd(xp, ldx, K, wp, ldw, None if bias is None else bias.data_ptr(), yp, ldy, N, x.shape[0],
                                      int(accumulate), int(relu), hb._stream()), "nerf_dense_fwd")


def _dgrad(dy, dy_lo, N, w, w_lo, dx, K, accumulate=False, act=None):
    dyp, lddy = _cols(dy, dy_lo, N)
    wp, ldw = _cols(w, w_lo, K)
    dxp, lddx = _cols(dx, 0, K)
    hb._check(hb.lib().nerf_dense_dgrad(dyp, lddy, N, wp, ldw, dxp, lddx, K, dy.shape[0], int(accumulate),
                                        None if act is None else act.data_ptr(), 0 if act is None else act.shape[1], hb._stream()),
              "nerf_dense_dgrad")


def _wgrad(dy, dy_lo, N, x, x_lo, K, dw, dw_lo, dbias=None):
    P = dy.shape[0]
    dyp, lddy = _cols(dy, dy_lo, N)
    xp, ldx = _cols(x, x_lo, K)
    dwp, lddw = _cols(dw, dw_lo, K)
    scratch = None
    if dbias is not None:
        scratch = torch.empty(max(1, hb.lib().nerf_dense_wgrad_scratch_floats(P, N)), dtype=torch.float32, device=dy.device)
    hb._check(hb.lib().nerf_dense_wgrad(dyp, lddy, N, xp, ldx, K, P, dwp, lddw, None if dbias is None else dbias.data_ptr(),
                                        None if scratch is None else scratch.data_ptr(), 0, hb._stream()), "nerf_dense_wgrad")


class _DenseMLP(torch.autograd.Function):
    """helpers:96-119 on x [P, input_ch + input_ch_views] -> [P, 4] (view directions: rgb | alpha) or [P, output_ch]."""

    @staticmethod
    def forward(ctx, model, x, *params):
        m = model
        P = x.shape[0]
        dev = x.device
        cx, W = m.input_ch, m.W
        new = lambda n: torch.empty((P, n), dtype=torch.float32, device=dev)
        hs = []
        for i, lin in enumerate(m.pts_linears):
            y = new(W)
            if i == 0:
                _fwd(x, 0, cx, lin.weight, 0, lin.bias, y, 0, W, relu=True)
            elif (i - 1) in m.skips:            # input = cat([input_pts, h]) (helpers:102-103): two GEMMs, no concatenation
                _fwd(x, 0, cx, lin.weight, 0, None, y, 0, W)
                _fwd(hs[-1], 0, W, lin.weight, cx, lin.bias, y, 0, W, accumulate=True, relu=True)
            else:
                _fwd(hs[-1], 0, W, lin.weight, 0, lin.bias, y, 0, W, relu=True)
            hs.append(y)
        h = hs[-1]
        feat = hv = None
        if m.use_viewdirs:
            cd, Wh = m.input_ch_views, W // 2
            out = new(4)
            _fwd(h, 0, W, m.alpha_linear.weight, 0, m.alpha_linear.bias, out, 3, 1)
            feat, hv = new(W), new(Wh)
            _fwd(h, 0, W, m.feature_linear.weight, 0, m.feature_linear.bias, feat, 0, W)
            vl = m.views_linears[0]
            _fwd(feat, 0, W, vl.weight, 0, None, hv, 0, Wh)                                  # cat([feature, input_views]) (:111)
            _fwd(x, cx, cd, vl.weight, W, vl.bias, hv, 0, Wh, accumulate=True, relu=True)
            _fwd(hv, 0, Wh, m.rgb_linear.weight, 0, m.rgb_linear.bias, out, 0, 3)
        else:
            out = new(m.output_ch)
            _fwd(h, 0, W, m.output_linear.weight, 0, m.output_linear.bias, out, 0, m.output_ch)
        ctx.model = m
        ctx.names = [n for n, _ in m.named_parameters()]
        ctx.save_for_backward(x, *hs, *([feat, hv] if m.use_viewdirs else []))
        ctx.needs = any(p.requires_grad for p in params)
        return out

    @staticmethod
    def backward(ctx, d_out):
        m = ctx.model
        if not ctx.needs:
            return (None, None) + (None,) * len(ctx.names)
        saved = ctx.saved_tensors
        x = saved[0]
        D, W, cx = m.D, m.W, m.input_ch
        hs = list(saved[1:1 + D])
        d_out = d_out.to(torch.float32).contiguous()
        P, dev = x.shape[0], x.device
        new = lambda n: torch.empty((P, n), dtype=torch.float32, device=dev)
        g = {n: torch.empty_like(p) for n, p in m.named_parameters()}
        h = hs[-1]
        d_h = new(W)
        if m.use_viewdirs:
            feat, hv = saved[1 + D], saved[2 + D]
            cd, Wh = m.input_ch_views, W // 2
            vl = m.views_linears[0]
            _wgrad(d_out, 0, 3, hv, 0, Wh, g["rgb_linear.weight"], 0, g["rgb_linear.bias"])
            d_hv = new(Wh)
            _dgrad(d_out, 0, 3, m.rgb_linear.weight, 0, d_hv, Wh, act=hv)
            _wgrad(d_hv, 0, Wh, feat, 0, W, g["views_linears.0.weight"], 0, g["views_linears.0.bias"])
            _wgrad(d_hv, 0, Wh, x, cx, cd, g["views_linears.0.weight"], W)
            d_feat = new(W)
            _dgrad(d_hv, 0, Wh, vl.weight, 0, d_feat, W)
            _wgrad(d_feat, 0, W, h, 0, W, g["feature_linear.weight"], 0, g["feature_linear.bias"])
            _wgrad(d_out, 3, 1, h, 0, W, g["alpha_linear.weight"], 0, g["alpha_linear.bias"])
            _dgrad(d_feat, 0, W, m.feature_linear.weight, 0, d_h, W)
            _dgrad(d_out, 3, 1, m.alpha_linear.weight, 0, d_h, W, accumulate=True, act=h)
        else:
            oc = m.output_ch
            _wgrad(d_out, 0, oc, h, 0, W, g["output_linear.weight"], 0, g["output_linear.bias"])
            _dgrad(d_out, 0, oc, m.output_linear.weight, 0, d_h, W, act=h)
            g["views_linears.0.weight"] = g["views_linears.0.bias"] = None           # unused without view directions (helpers:82)
        for i in range(D - 1, -1, -1):          # d_h = dL / d(pre-activation of layer i)
            lin = m.pts_linears[i]
            wn, bn = f"pts_linears.{i}.weight", f"pts_linears.{i}.bias"
            if i == 0:
                _wgrad(d_h, 0, W, x, 0, cx, g[wn], 0, g[bn])
                break
            d_prev = new(W)
            if (i - 1) in m.skips:
                _wgrad(d_h, 0, W, x, 0, cx, g[wn], 0, g[bn])
                _wgrad(d_h, 0, W, hs[i - 1], 0, W, g[wn], cx)
                _dgrad(d_h, 0, W, lin.weight, cx, d_prev, W, act=hs[i - 1])
            else:
                _wgrad(d_h, 0, W, hs[i - 1], 0, W, g[wn], 0, g[bn])
                _dgrad(d_h, 0, W, lin.weight, 0, d_prev, W, act=hs[i - 1])
            d_h = d_prev
        return (None, None) + tuple(g[n] for n in ctx.names)


class DenseNeRF(nn.Module):
    """run_nerf_helpers.py:67-119 with the reference's constructor, attributes, registration order and state_dict."""

    def __init__(self, D=8, W=256, input_ch=3, input_ch_views=3, output_ch=4, skips=[4], use_viewdirs=False):
        super().__init__()
        self.D, self.W = D, W
        self.input_ch, self.input_ch_views = input_ch, input_ch_views
        self.output_ch = output_ch
        self.skips, self.use_viewdirs = list(skips), use_viewdirs
        if (D - 1) in self.skips:
            raise ValueError(f"NeRF(D={D}, skips={skips}): a skip connection behind the last layer feeds {W + input_ch} features into a "
                             f"{W}-wide head (the reference fails at its first forward for the same reason)")
        self.pts_linears = nn.ModuleList([nn.Linear(input_ch, W)] + [nn.Linear(W, W) if i not in self.skips else nn.Linear(W + input_ch, W)
                                                                      for i in range(D - 1)])
        self.views_linears = nn.ModuleList([nn.Linear(input_ch_views + W, W // 2)])
        if use_viewdirs:
            self.feature_linear = nn.Linear(W, W)
            self.alpha_linear = nn.Linear(W, 1)
            self.rgb_linear = nn.Linear(W // 2, 3)
        else:
            self.output_linear = nn.Linear(W, output_ch)

    @property
    def multires(self):
        """frequencies of the positional encoding get_embedder() built for input_ch (-1: the identity of i_embed = -1)"""
        return _freqs_of(self.input_ch)

    @property
    def multires_views(self):
        return _freqs_of(self.input_ch_views) if self.use_viewdirs else -1

    def forward(self, x):
        """x [..., input_ch + input_ch_views] as run_network builds it (run_nerf.py:41-47)"""
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1]).to(torch.float32).contiguous()
        if x2.shape[1] != self.input_ch + self.input_ch_views:
            raise ValueError(f"NeRF.forward: {x2.shape[1]} input features for input_ch={self.input_ch} + input_ch_views={self.input_ch_views}")
        out = _DenseMLP.apply(self, x2, *[p for _, p in self.named_parameters()])
        return out.reshape(*lead, out.shape[-1])

    def query(self, rays, z_vals):
        """raw [N, S, 4 | output_ch] of the sample points o + d z of `rays` [N, 8 | 11]: nerf_build_inputs + the layer stack"""
        n, S = z_vals.shape
        # without gradients nothing has to outlive a layer: bound the transient activations (D + 2 arrays of P x W floats) by
        # evaluating ~2^20 points at a time -- what the reference's netchunk does (run_nerf.py:27-34; results do not depend on it)
        rays_per_slice = max(1, (1 << 20) // max(S, 1))
        if n > rays_per_slice and not (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())):
            return torch.cat([self.query(rays[i:i + rays_per_slice].contiguous(), z_vals[i:i + rays_per_slice].contiguous())
                              for i in range(0, n, rays_per_slice)], 0)
        C = self.input_ch + self.input_ch_views
        x = torch.zeros((n * S, C), dtype=torch.float32, device=rays.device)      # (columns the kernel does not write stay defined)
        hb._check(hb.lib().nerf_build_inputs(hb._ptr(rays, "rays"), rays.shape[1], hb._ptr(z_vals, "z_vals"), n, S, self.multires,
                                             self.multires_views, int(self.use_viewdirs), x.data_ptr(), C, hb._stream()), "nerf_build_inputs")
        out = _DenseMLP.apply(self, x, *[p for _, p in self.named_parameters()])
        return out.reshape(n, S, out.shape[-1])


def _freqs_of(ch):
    if ch == 3:
        return -1
    if ch < 3 or (ch - 3) % 6:
        raise NotImplementedError(f"input width {ch} is not 3 (identity) or 3 + 6 L (get_embedder, helpers:48-63)")
    return (ch - 3) // 6


def render_rays_dense(cfg, rays, rnd, model_c, model_f):
    """run_nerf.py:351-412 for DenseNeRF networks: the same per-ray kernels as the fused path around DenseNeRF.query; autograd
    through the compositing (render._Composite) and the layer stack (_DenseMLP).  Returns the reference's dict."""
    from .render import _Composite, _linspace01
    n_c, n_f = cfg["N_samples"], cfg["N_importance"]
    std, wb, dev = cfg["raw_noise_std"], cfg["white_bkgd"], rays.device
    if model_c.use_viewdirs and rays.shape[1] < 11:
        raise ValueError("render_rays: a network with view directions needs ray records with 11 columns")
    rays_d = rays[:, 3:6].contiguous()
    z = hb.sample_coarse(rays, _linspace01(n_c, dev), cfg["lindisp"], rnd.get("t_rand"))
    raw = model_c.query(rays, z)
    rgb, disp, acc, weights, _ = _Composite.apply(raw[..., :4].contiguous(), z, rays_d, rnd.get("noise_c"), std, wb)
    ret = {}
    if n_f > 0:
        ret.update(rgb0=rgb, disp0=disp, acc0=acc)
        u = rnd.get("u")
        z, z_std, _ = hb.sample_fine(z, weights.detach().contiguous(), n_f, u, None if u is not None else _linspace01(n_f, dev))
        raw = (model_c if model_f is None else model_f).query(rays, z)
        rgb, disp, acc, _, _ = _Composite.apply(raw[..., :4].contiguous(), z, rays_d, rnd.get("noise_f"), std, wb)
        ret["z_std"] = z_std
    ret.update(rgb_map=rgb, disp_map=disp, acc_map=acc, raw=raw)
    return ret
