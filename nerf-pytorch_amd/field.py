"""Field model host objects: ``Embedder`` / ``get_embedder`` / ``NeRF``.

Same names, constructor arguments and ``state_dict`` layout as the reference
(run_nerf_helpers.py:15-63, :67-119) so ``create_nerf`` / checkpoints /
``torch.optim.Adam`` keep working unchanged -- but the arithmetic runs in the
HIP library: ``NeRF`` keeps its 24 parameters as views into ONE flat fp32 vector
(state_dict order) that the kernels read through a fragment repack, and
``.grad`` of all of them are views into one flat gradient vector that a single
RCCL all-reduce covers.
"""
import torch
import torch.nn as nn

from . import hip_backend as hb


# --------------------------------------------------------------------------- positional encoding
class Embedder:
    """run_nerf_helpers.py:15-45.  Only the configuration get_embedder() builds is
    supported (3 inputs, include_input, log-sampled power-of-two bands, [sin, cos])."""

    def __init__(self, **kwargs):
        self.kwargs = kwargs
        d = kwargs["input_dims"]
        n = kwargs["num_freqs"]
        fns = list(kwargs.get("periodic_fns", []))
        ok = (d == 3 and kwargs.get("include_input", True) and kwargs.get("log_sampling", True)
              and kwargs["max_freq_log2"] == n - 1 and fns == [torch.sin, torch.cos])
        if not ok:
            raise NotImplementedError("Embedder: only get_embedder()'s configuration is implemented on gfx950")
        self.num_freqs = n
        self.out_dim = d + 2 * n * d

    def embed(self, inputs):
        return hb.embed(inputs.float(), self.num_freqs)


def get_embedder(multires, i=0):
    """run_nerf_helpers.py:48-63."""
    if i == -1:
        return nn.Identity(), 3
    eo = Embedder(include_input=True, input_dims=3, max_freq_log2=multires - 1, num_freqs=multires,
                  log_sampling=True, periodic_fns=[torch.sin, torch.cos])
    embed = lambda x, eo=eo: eo.embed(x)
    embed.num_freqs = multires
    return embed, eo.out_dim


# --------------------------------------------------------------------------- the MLP
class NeRF(nn.Module):
    """Reference NeRF MLP (run_nerf_helpers.py:67-119), evaluated by fused HIP kernels.

    This class is the architecture of every BASELINE config (D=8, W=256, input_ch=63, input_ch_views=27, skips=[4],
    use_viewdirs=True).  ``NeRF(...)`` with any other arguments the reference accepts returns a ``dense.DenseNeRF``: same
    constructor, attributes and state_dict, evaluated layer by layer (library GEMMs behind the C ABI) instead of by the fused
    kernels.  There is no eager / CPU fallback on either path."""

    @staticmethod
    def fused(D=8, W=256, input_ch=3, input_ch_views=3, output_ch=4, skips=[4], use_viewdirs=False):
        """whether these constructor arguments are the architecture of the fused kernels"""
        return bool(D == 8 and W == 256 and input_ch == 63 and input_ch_views == 27 and list(skips) == [4] and use_viewdirs)

    def __new__(cls, *args, force_dense=False, **kwargs):
        # NeRF(...) with any other arguments the reference accepts builds the layer-by-layer module (dense.py): same constructor,
        # attributes and state_dict; the reference's arithmetic on library GEMMs instead of the fused kernels.  force_dense
        # (keyword-only, not in the reference): the layer-by-layer module also for the fused architecture -- create_nerf uses it
        # when the OTHER network of the pair is outside the fused architecture (render_rays runs a pair on one path).
        if cls is NeRF and (force_dense or not NeRF.fused(*args, **kwargs)):
            from .dense import DenseNeRF
            return DenseNeRF(*args, **kwargs)
        return super().__new__(cls)

    def __getnewargs_ex__(self):        # copy.deepcopy / pickle re-create the object through __new__: keep it on this class
        return (), dict(D=8, W=256, input_ch=63, input_ch_views=27, skips=[4], use_viewdirs=True)

    def __init__(self, D=8, W=256, input_ch=3, input_ch_views=3, output_ch=4, skips=[4], use_viewdirs=False, force_dense=False):
        super().__init__()
        if not (D == 8 and W == 256 and input_ch == 63 and input_ch_views == 27 and list(skips) == [4]
                and use_viewdirs):
            raise NotImplementedError(
                "nerf-pytorch_amd implements NeRF(D=8, W=256, input_ch=63, input_ch_views=27, skips=[4], "
                f"use_viewdirs=True); got D={D} W={W} input_ch={input_ch} input_ch_views={input_ch_views} "
                f"skips={skips} use_viewdirs={use_viewdirs}")
        self.D, self.W = D, W
        self.input_ch, self.input_ch_views = input_ch, input_ch_views
        self.skips, self.use_viewdirs = skips, use_viewdirs
        # same construction order as the reference => same default init from the same RNG state
        self.pts_linears = nn.ModuleList(
            [nn.Linear(input_ch, W)] + [nn.Linear(W, W) if i not in self.skips else nn.Linear(W + input_ch, W)
                                        for i in range(D - 1)])
        self.views_linears = nn.ModuleList([nn.Linear(input_ch_views + W, W // 2)])
        self.feature_linear = nn.Linear(W, W)
        self.alpha_linear = nn.Linear(W, 1)
        self.rgb_linear = nn.Linear(W // 2, 3)
        self._flat = None
        self._packed = None
        self._packed_key = None
        self.last_flat_grad = None      # set by the backward: flat gradient the .grad views live in
        self._bind_flat()

    # ---- flat parameter vector -------------------------------------------------
    def _ordered_params(self):
        sd = dict(self.named_parameters())
        return [(nm, off, shape, sd[nm]) for nm, off, shape in _param_table()]

    def _bind_flat(self):
        """(Re)create the flat vector from the current parameter values and make every
        parameter a view into it."""
        params = self._ordered_params()
        dev = params[0][3].device
        flat = torch.empty(hb.N_PARAMS, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for nm, off, shape, p in params:
                n = p.numel()
                flat[off:off + n].copy_(p.detach().reshape(-1))
                p.data = flat[off:off + n].view(shape)
        self._flat = flat
        self._packed = None

    def _is_bound(self):
        base = self._flat.data_ptr()
        for nm, off, shape, p in self._ordered_params():
            if p.data_ptr() != base + 4 * off or p.dtype != torch.float32:
                return False
        return True

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        self._bind_flat()
        return out

    def flat_params(self):
        if self._flat is None or not self._is_bound():
            self._bind_flat()
        return self._flat

    def _refresh_packed_cache(self):
        """drop the cached repacks if the parameters changed since they were made; returns the flat parameter vector"""
        flat = self.flat_params()
        # torch-side in-place updates advance the version counters of the parameters / of the flat vector; the fused
        # Adam kernel advances hb.param_epoch(flat).  Writers that do neither (c10d collectives such as the DP parameter
        # broadcast, `.data` writes, raw pointers) must call invalidate_packed().
        key = tuple(p._version for p in self.parameters()) + (flat.data_ptr(), flat._version, hb.param_epoch(flat))
        if self._packed is None or key != self._packed_key:
            self._packed = {}
            self._packed_key = key
        return flat

    def packed_params(self, precision="fp32"):
        """Fragment repack of the current parameters for the given datapath (cached on the parameters' versions)."""
        flat = self._refresh_packed_cache()
        precision = hb.PACK_OF.get(precision, precision)        # (e.g. "fp16x3w" reads fp16x3's fragments)
        if precision not in self._packed:
            # fresh tensor each time: a pending backward keeps a reference to the old one
            self._packed[precision] = hb.pack_params(flat, precision=precision)
        return self._packed[precision]

    def invalidate_packed(self):
        """Drop the cached fragment repacks: call after writing the parameters through a path that does not advance
        tensor version counters (dist.broadcast / all_reduce on flat_params(), `.data` writes, raw device pointers)."""
        self._packed = None
        self._packed_key = None

    def param_list(self):
        return [p for _, _, _, p in self._ordered_params()]

    # ---- reference call surface --------------------------------------------------
    def forward(self, x):
        """x [..., 90] = cat(embed(pts), embed(viewdirs)) as run_network builds it
        (run_nerf.py:41-47).  The encoding is recomputed in-kernel from its identity
        columns (x[..., 0:3] and x[..., 63:66]), so x must be a genuine embedding."""
        from .render import query_points
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1])
        out = query_points(self, x2[:, 0:3], x2[:, self.input_ch:self.input_ch + 3])
        return out.reshape(*lead, 4)

    def load_weights_from_keras(self, weights):
        """run_nerf_helpers.py:121-148 (interop utility; copies into the flat vector)."""
        import numpy as np
        with torch.no_grad():
            def put(lin, w, b):
                lin.weight.copy_(torch.from_numpy(np.transpose(w)))
                lin.bias.copy_(torch.from_numpy(np.transpose(b)))
            for i in range(self.D):
                put(self.pts_linears[i], weights[2 * i], weights[2 * i + 1])
            put(self.feature_linear, weights[2 * self.D], weights[2 * self.D + 1])
            put(self.views_linears[0], weights[2 * self.D + 2], weights[2 * self.D + 3])
            put(self.rgb_linear, weights[2 * self.D + 4], weights[2 * self.D + 5])
            put(self.alpha_linear, weights[2 * self.D + 6], weights[2 * self.D + 7])


def packed_params_pair(model_a, model_b, precision):
    """(model_a.packed_params(precision), model_b.packed_params(precision)); when BOTH repacks are stale -- every training step, after
    the optimizer step -- the two networks are repacked in the two launches one takes (hb.pack_params_pair) instead of four."""
    precision = hb.PACK_OF.get(precision, precision)
    if model_a is model_b or precision not in ("fp16x3", "bf16x3") or not (isinstance(model_a, NeRF) and isinstance(model_b, NeRF)):
        return model_a.packed_params(precision), model_b.packed_params(precision)
    fa, fb = model_a._refresh_packed_cache(), model_b._refresh_packed_cache()
    if precision not in model_a._packed and precision not in model_b._packed and fa.device == fb.device:
        model_a._packed[precision], model_b._packed[precision] = hb.pack_params_pair(fa, fb, precision)
    return model_a.packed_params(precision), model_b.packed_params(precision)


_TABLE = None


def _param_table():
    global _TABLE
    if _TABLE is None:
        _TABLE = hb.param_table()
    return _TABLE
