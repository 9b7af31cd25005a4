"""Volumetric renderer host code: the reference's hot-path call surface
(run_nerf.py:27-134, :262-418; run_nerf_helpers.py:153-239) on top of the HIP
library.  Function names, argument order, defaults and returned structures are
the reference's; the bodies enqueue fused kernels instead of ATen op chains.

Random draws (stratified jitter, density noise, CDF samples) are made here with
torch, in the reference's order and shapes, and handed to the
kernels, so a seeded run consumes the generator exactly like the reference.
"""
import numpy as np
import torch

from . import hip_backend as hb
from .field import NeRF, packed_params_pair

_LINSPACE_CACHE = {}
# The datapath a user gets without asking (round 5): the fp16 three-term split -- fp32-class products (~2^-22), the configuration
# bench.py's headline measures and the north-star gate admits at 4e-6 dB.  NERF_PRECISION=fp32 (or set_precision("fp32")) selects the
# exact-fp32 anchor; the test suite runs under it unless a test names a datapath (tests/conftest.py).
DEFAULT_PRECISION = "fp16x3"
_PRECISION = __import__("os").environ.get("NERF_PRECISION", DEFAULT_PRECISION)
if _PRECISION not in hb.PRECISIONS:
    raise ValueError(f"NERF_PRECISION={_PRECISION!r}: must be one of {hb.PRECISIONS}")


def set_precision(mode):
    """Select the field datapath (hip_backend.PRECISIONS):
      "fp16x3"    (default) every product as three fp16 MFMAs, W_hi x_hi + W_hi x_lo + W_lo x_hi with hi = fp16(v), lo = fp16(v - hi):
                  ~2^-22 per product (fp32-class), fp32 accumulation / activations / gradients; the bench headline;
      "fp32"      exact fp32 MFMA (v_mfma_f32_16x16x4_f32, bitwise an fmaf chain): the parity anchor;
      "bf16x3"    the same with bf16 parts (~2^-17 per product, 8-bit operands for the weight-gradient GEMM; fp32's exponent range:
                  the datapath for activations beyond fp16's 65504);
      "fp16_fp8c" fp16x3 for everything that needs gradients; no_grad rendering on fp16 main term + fp8 correction terms (~2^-15);
      "fp16x3w"   fp16x3 whose weight-gradient GEMM contracts TWO-WORD operands (the forward and the delta chain also save the lo
                  words): gradients of the forward's product class instead of 11-bit operands, ~1.4x the step time; forward values
                  bit-identical to fp16x3's.  The instrument that prices the default's operand storage, not a default."""
    global _PRECISION
    if mode not in hb.PRECISIONS:
        raise ValueError(f"precision must be one of {hb.PRECISIONS}")
    _PRECISION = mode


def get_precision():
    return _PRECISION


def check_range():
    """The fp16 split's range guard rail on demand (hb.RangeMonitor.report): waits for the scans in flight and returns
    {"max_activation": largest post-ReLU activation a scanned training forward saved, "warn_at": 32768, "limit": 65504, ...}.
    hip_backend.RANGE_MONITOR.every (NERF_RANGE_CHECK_EVERY, default 64) = every how many training renders a scan runs; 0 = never."""
    return hb.RANGE_MONITOR.report()


def _linspace01(n, device):
    key = (n, str(device))
    t = _LINSPACE_CACHE.get(key)
    if t is None:
        t = torch.linspace(0.0, 1.0, steps=n, dtype=torch.float32, device=device)
        _LINSPACE_CACHE[key] = t
    return t


def _f32c(t):
    return t.detach().to(torch.float32).contiguous()


# --------------------------------------------------------------------------- autograd glue
_FREED_MSG = ("nerf-pytorch_amd: the saved activations of this render were already consumed by a backward pass and "
              "freed; backward through the same graph a second time is not supported (sum the losses and call "
              "backward once)")


_STALE_MSG = ("nerf-pytorch_amd: the network parameters changed between this render's forward and its backward "
              "(optimizer.step / load_state_dict / broadcast in between); the split-bf16 backward combines activations saved "
              "by the forward with the current feature_linear / views_linears weights, so the gradient would mix two "
              "parameter states.  Call backward before updating the parameters.")


# Called as hook(model, flat_grad) from inside the backward pass the moment a network's flat gradient vector is final
# (all of its kernels are enqueued on the current stream).  parallel.GradientSync uses it to start that network's
# all-reduce while the other network's backward still runs (the coarse and fine backward are independent: the
# reference detaches z_samples, run_nerf.py:394).
GRAD_READY_HOOKS = []


def _grad_ready(model, flat_grad):
    model.last_flat_grad = flat_grad
    for hook in GRAD_READY_HOOKS:
        hook(model, flat_grad)


class _FieldQuery(torch.autograd.Function):
    """raw = field(rays, z) for explicit rays/depths; gradients w.r.t. the parameters only."""

    @staticmethod
    def forward(ctx, model, rays, z_vals, need, *params):
        prec = _PRECISION
        if prec == "fp16_fp8c" and need:        # the reduced class is an inference form; gradients: the fp16x3 datapath
            prec = "fp16x3"
        packed = model.packed_params(prec)
        raw, act = hb.field_fwd(packed, rays, z_vals, save_act=need, precision=prec,
                                guard_packed=model.packed_params("fp16x3") if prec == "fp16_fp8c" else None)
        ctx.model, ctx.packed, ctx.act, ctx.prec, ctx.saved_any = model, packed, act, prec, bool(need)
        ctx.param_state = _param_state(model)
        ctx.set_materialize_grads(False)
        return raw

    @staticmethod
    def backward(ctx, d_raw):
        model = ctx.model
        if d_raw is None or not ctx.saved_any:      # nothing requires grad: forward saved nothing on purpose
            return (None, None, None, None) + (None,) * len(_param_slices(model))
        if ctx.act is None:
            raise RuntimeError(_FREED_MSG)
        if ctx.prec != "fp32" and _param_state(model) != ctx.param_state:
            raise RuntimeError(_STALE_MSG)
        grad = torch.empty(hb.N_PARAMS, dtype=torch.float32, device=d_raw.device)
        hb.field_bwd(ctx.packed, ctx.act, d_raw.contiguous(), grad, accumulate=False, precision=ctx.prec,
                     params=model.flat_params())
        hb.WORKSPACE.give(ctx.act)
        ctx.act = None      # ~10 KB per point: back to the workspace pool as soon as the gradient exists
        _grad_ready(model, grad)
        return (None, None, None, None) + _grad_views(model, grad)


def _param_slices(model):
    from .field import _param_table
    return _param_table()


def _param_state(model):
    """Identity of the parameter values a forward saw: storage, in-place version, fused-Adam epoch."""
    flat = model.flat_params()
    return (flat.data_ptr(), flat._version, tuple(p._version for p in model.parameters()), hb.param_epoch(flat))


def _grad_views(model, flat_grad):
    return tuple(flat_grad[off:off + int(np.prod(shape))].view(shape) for _, off, shape in _param_slices(model))


def _field_pass(cfg, rays, rnd, model_c, model_f, save):
    """One evaluation of render_rays' pipeline (run_nerf.py:351-412) on `rays`: coarse depths -> field -> composite
    [-> hierarchical depths -> field -> composite].  save=True leases workspace buffers for the saved activations
    (hb.Workspace) and returns them in the dict; everything else is small ([N,S]-sized)."""
    n_c, n_f = cfg["N_samples"], cfg["N_importance"]
    dev = rays.device
    std, wb, prec = cfg["raw_noise_std"], cfg["white_bkgd"], cfg.get("precision", "fp32")
    r = {}
    guard = (lambda m: m.packed_params("fp16x3")) if prec == "fp16_fp8c" else (lambda m: None)
    r["z_c"] = hb.sample_coarse(rays, _linspace01(n_c, dev), cfg["lindisp"], rnd.get("t_rand"))
    mf = model_c if (model_f is None or model_f is model_c) else model_f
    nxt = raw_f = None
    # The reduced inference class with a refining pass: hierarchical sampling divides by ~1e-5 in bins the coarse pass found empty
    # (run_nerf_helpers.py:234-236), so a 2^-15 perturbation of the coarse weights moves single fine samples by whole bins -- measured
    # on the reference's fixtures at BASELINE's batch sizes: no flipped ray, but images at 62-79 dB of the reference's instead of
    # fp16x3's 89-120 dB (tools/EXPERIMENTS.md, round 5).  The coarse pass (a quarter of the points) therefore runs on the three-term
    # products and only the refining pass (whose errors reach the image unamplified) on the reduced ones.
    prec_c = "fp16x3" if (prec == "fp16_fp8c" and n_f > 0 and hb.REDUCED_COARSE_THREE_TERM) else prec
    if n_f > 0 and mf is not model_c and prec_c == prec:
        r["packed_c"], packed_f = packed_params_pair(model_c, mf, prec)       # (both stale after an optimizer step: one pair of launches)
    else:
        r["packed_c"], packed_f = model_c.packed_params(prec_c), None
    if prec_c == "fp16_fp8c" and n_f > 0:     # the guard launch of the coarse pass also evaluates the fine pass's last sample (hb.field_fwd)
        raw_f = torch.empty((rays.shape[0], n_c + n_f, 4), dtype=torch.float32, device=dev)
        nxt = (guard(mf), raw_f)
    r["raw_c"], r["act_c"] = hb.field_fwd(r["packed_c"], rays, r["z_c"], save_act=save, precision=prec_c,
                                          guard_packed=guard(model_c) if prec_c == "fp16_fp8c" else None, next_guard=nxt)
    r["rgb_c"], r["disp_c"], r["acc_c"], w_c, _ = hb.raw2outputs(r["raw_c"], r["z_c"], rays, rays.shape[1], rnd.get("noise_c"), std, wb,
                                                              want_weights=n_f > 0, want_depth=False, rays_d_offset=3)
    if n_f <= 0:
        return r
    u = rnd.get("u")
    r["z_f"], r["z_std"], _ = hb.sample_fine(r["z_c"], w_c, n_f, u, None if u is not None else _linspace01(n_f, dev))
    r["packed_f"] = packed_f if packed_f is not None else mf.packed_params(prec)
    r["raw_f"], r["act_f"] = hb.field_fwd(r["packed_f"], rays, r["z_f"], save_act=save, precision=prec,
                                          guard_packed="done" if raw_f is not None else guard(mf), raw=raw_f)
    r["rgb_f"], r["disp_f"], r["acc_f"], _, _ = hb.raw2outputs(r["raw_f"], r["z_f"], rays, rays.shape[1], rnd.get("noise_f"), std, wb,
                                                             want_weights=False, want_depth=False, rays_d_offset=3)
    return r


def _release(r):
    for k in ("act_c", "act_f"):
        hb.WORKSPACE.give(r.get(k))
        r[k] = None


# how the last training render_rays call kept its backward state: ("one launch" | "resident sub-chunks" | "recompute", rays, rays per
# sub-chunk) -- benchmarks read it to report (and refuse to mislabel) the recompute fallback
LAST_BACKWARD_PLAN = None


class _RenderRays(torch.autograd.Function):
    """The whole of render_rays (run_nerf.py:308-418) as one autograd node.

    Memory: the backward needs 4.8 KB of saved activations per sample point on the split datapaths (16-bit tiles; 10.7 KB of fp32
    rows on the fp32 datapath) plus as much for the deltas.  Up to hb.max_saved_rays(...) rays per call (default budget 48 GiB:
    ~22k rays at 64+128 samples on the split datapaths, ~10k on fp32, i.e. every N_rand of the BASELINE configs) they are saved by
    the forward into buffers leased from hb.WORKSPACE (persistent across steps, no per-step allocation).  Larger ray chunks (the
    reference's default chunk is 32768 rays) are rendered in equal sub-chunks.  If the saved activations of ALL sub-chunks fit
    hb.SAVE_TOTAL_BYTES (default 160 GiB of the 288 GB: the 32768-ray batch of configs[3] needs ~40 GB on the split datapaths,
    ~90 GB on fp32) every sub-chunk keeps its own lease and the backward walks them: no
    recomputation, deltas and partial sums re-use one sub-chunk-sized scratch.  Beyond that the forward runs WITHOUT
    saving and the backward re-runs it, with saving, one sub-chunk at a time (the kernels are deterministic, so the
    recomputed pass is bit-identical to the first): bounded memory for +1 inference-speed forward."""

    @staticmethod
    def forward(ctx, cfg, rays, rnd, model_c, model_f, *params):
        n_f = cfg["N_importance"]
        need = cfg["need_grad"]
        n = rays.shape[0]
        sub = hb.max_saved_rays(cfg["N_samples"], n_f, cfg.get("precision", "fp32")) if need else n
        ctx.checkpoint = bool(need and n > sub)
        ctx.tiles = None
        if ctx.checkpoint:
            ceil_div = lambda a, b: -(-a // b)
            sub = min(sub, 64 * ceil_div(ceil_div(n, ceil_div(n, sub)), 64))        # equal sub-chunks, multiples of 64 rays
            # resident sub-chunks only if they fit the budget AND what the device actually has free right now: what the driver reports
            # + the pool's idle leases LARGE ENOUGH to hold a sub-chunk's saved activations (smaller ones serve nothing) + half of
            # what torch's caching allocator holds without using (cached blocks are fragmented: only part of them can back a
            # multi-GB request).  The estimate can still be wrong: the forward below falls back to recomputation on out-of-memory.
            prec_ = cfg.get("precision", "fp32")
            lease = min(hb.act_floats(sub, cfg["N_samples"], prec_), hb.act_floats(sub, cfg["N_samples"] + n_f, prec_) if n_f > 0 else 1 << 62)
            cached = torch.cuda.memory_reserved(rays.device) - torch.cuda.memory_allocated(rays.device)
            free_now = torch.cuda.mem_get_info(rays.device)[0] + cached // 2 + hb.WORKSPACE.idle_bytes(rays.device, lease)
            if hb.saved_bytes(sub, cfg["N_samples"], n_f, prec_) * ceil_div(n, sub) <= min(hb.SAVE_TOTAL_BYTES, int(0.9 * free_now)):
                ctx.checkpoint = False
                ctx.tiles = [(lo, min(lo + sub, n)) for lo in range(0, n, sub)]
        global LAST_BACKWARD_PLAN
        LAST_BACKWARD_PLAN = ("recompute" if ctx.checkpoint else "resident sub-chunks" if ctx.tiles is not None else "one launch", n, sub)
        ctx.sub_rays = sub
        prec = cfg.get("precision", "fp32")
        if not need and hb.INFER_ONE_LAUNCH and hb.render_infer_supported(cfg["N_samples"], n_f, prec):
            # no gradients: coarse depths -> network -> compositing -> hierarchical depths -> network -> compositing in ONE launch
            mf = None if (model_f is None or model_f is model_c or n_f <= 0) else model_f
            r = hb.render_rays_infer(model_c.packed_params(prec), None if mf is None else mf.packed_params(prec), rays, cfg["N_samples"],
                                     n_f, cfg["lindisp"], cfg["white_bkgd"], cfg["raw_noise_std"], prec, rnd)
        elif ctx.tiles is None:
            r = _field_pass(cfg, rays, rnd, model_c, model_f, save=need and not ctx.checkpoint)
        else:
            # every sub-chunk keeps its saved activations; the node's outputs are the concatenations (new tensors)
            parts = []
            try:
                for lo, hi in ctx.tiles:
                    parts.append(_field_pass(cfg, rays[lo:hi], {k_: v[lo:hi] for k_, v in rnd.items()}, model_c, model_f, save=True))
            except torch.cuda.OutOfMemoryError:
                # the free-memory estimate was wrong (fragmented cache, another tenant of the device): hand everything back and take
                # the recompute plan -- forward without saving, the backward re-runs it per sub-chunk (bit-identical)
                for p_ in parts:
                    _release(p_)
                parts = None
                hb.WORKSPACE.clear()
                torch.cuda.empty_cache()
                ctx.tiles, ctx.checkpoint = None, True
                LAST_BACKWARD_PLAN = ("recompute", n, sub)
            if parts is None:
                r = _field_pass(cfg, rays, rnd, model_c, model_f, save=False)
            else:
                out_keys = ("rgb_c", "disp_c", "acc_c", "raw_c") if n_f <= 0 else ("rgb_f", "disp_f", "acc_f", "raw_f", "rgb_c", "disp_c", "acc_c", "z_std")
                r = {k_: torch.cat([p_[k_] for p_ in parts], 0) for k_ in out_keys}
        if need and not ctx.checkpoint and prec in ("fp16x3", "fp16x3w"):
            # the fp16 split's range guard rail: every RANGE_MONITOR.every-th training render scans what the forward saved
            # (hb.RangeMonitor; replaces run_nerf.py:414-416's DEBUG-gated NaN / Inf check)
            for r_, n_ in ([(r, n)] if ctx.tiles is None else [(p_, hi - lo) for p_, (lo, hi) in zip(parts, ctx.tiles)]):
                hb.RANGE_MONITOR.after_forward([(r_[k_], s_) for k_, s_ in (("act_c", cfg["N_samples"]), ("act_f", cfg["N_samples"] + n_f))
                                                if r_.get(k_) is not None], n_)
        ctx.cfg, ctx.model_c, ctx.model_f = cfg, model_c, model_f
        ctx.same_net = model_f is None or model_f is model_c
        ctx.n_params_c = len(_param_slices(model_c))
        ctx.need = need
        ctx.set_materialize_grads(False)
        ctx.rays, ctx.rnd = rays, rnd
        # What the backward needs, WITHOUT the node's own outputs: an output carries grad_fn = this node, so keeping it
        # in ctx.__dict__ is a node -> ctx -> output -> node cycle the garbage collector cannot break (a graph dropped
        # without backward would pin its ~11 GB of saved activations for good).  The one output the backward reads,
        # `raw` of the last pass, goes through save_for_backward, which autograd knows how to hold without a cycle.
        ctx.saved = None
        if ctx.tiles is not None:
            keep = ("packed_c", "z_c", "act_c", "raw_c", "packed_f", "z_f", "act_f", "raw_f")
            ctx.saved = [{k_: p_[k_] for k_ in keep if k_ in p_} for p_ in parts]     # per-sub-chunk tensors, none is an output
        elif need and not ctx.checkpoint:
            keep = ("packed_c", "z_c", "act_c", "packed_f", "z_f", "act_f") + (("raw_c",) if n_f > 0 else ())
            ctx.saved = {k: r[k] for k in keep if k in r}
            ctx.save_for_backward(r["raw_f"] if n_f > 0 else r["raw_c"])
        elif not need:
            _release(r)
        # the folded feature layer of the split datapaths' backward reads the LIVE parameters (Wf, bf, Wv) next to
        # fragments packed at forward time: remember which parameter state this forward saw
        ctx.param_state = tuple(_param_state(m) for m in (model_c, model_f) if m is not None)
        ctx.consumed = False
        if n_f <= 0:
            return r["rgb_c"], r["disp_c"], r["acc_c"], r["raw_c"]
        ctx.mark_non_differentiable(r["z_std"])     # the reference detaches z_samples (run_nerf.py:394)
        return r["rgb_f"], r["disp_f"], r["acc_f"], r["raw_f"], r["rgb_c"], r["disp_c"], r["acc_c"], r["z_std"]

    @staticmethod
    def backward(ctx, *gouts):
        if ctx.consumed:
            raise RuntimeError(_FREED_MSG)
        cfg = ctx.cfg
        n_lead = 5
        none_c = (None,) * ctx.n_params_c
        none_all = (None,) * n_lead + none_c + (() if ctx.same_net else none_c)
        if not ctx.need:
            return none_all
        now = tuple(_param_state(m) for m in (ctx.model_c, ctx.model_f) if m is not None)
        if cfg.get("precision", "fp32") != "fp32" and now != ctx.param_state:
            raise RuntimeError(_STALE_MSG)
        rays_all, rnd_all = ctx.rays, ctx.rnd
        std, wb, prec = cfg["raw_noise_std"], cfg["white_bkgd"], cfg.get("precision", "fp32")
        dev = rays_all.device
        n_all = rays_all.shape[0]
        n_f = cfg["N_importance"]
        fine = n_f > 0
        # upstream gradients of (rgb, disp, acc, raw) of the fine (or only) pass and of the coarse pass
        up_f = (gouts[0], gouts[1], gouts[2], gouts[3])
        up_c = (gouts[4], gouts[5], gouts[6], None) if fine else None
        if not fine:
            up_c, up_f = up_f, None
        has = lambda up: up is not None and any(g is not None for g in up)
        if not has(up_c) and not has(up_f):
            ctx.consumed = True
            if ctx.saved is not None:
                for r_ in (ctx.saved if isinstance(ctx.saved, list) else [ctx.saved]):
                    _release(r_)
                ctx.saved = None
            return none_all
        grad_c = torch.empty(hb.N_PARAMS, dtype=torch.float32, device=dev)
        grad_f = None if (ctx.same_net or not fine) else torch.empty(hb.N_PARAMS, dtype=torch.float32, device=dev)
        wrote = {"c": False, "f": False}

        def field_grad(rays, model, packed, act, raw, z, noise, up, lo, hi, grad, key):
            d_rgb, d_disp, d_acc, d_raw_up = (None if g is None else g[lo:hi] for g in up)
            m = hi - lo
            if d_rgb is None and (d_disp is not None or d_acc is not None):
                d_rgb = torch.zeros((m, 3), dtype=torch.float32, device=dev)
            c = lambda t: t.to(torch.float32).contiguous() if t is not None else None
            if d_rgb is None:       # only `raw` itself (extras['raw'], e.g. a sigma regulariser) carries a gradient
                d_raw = c(d_raw_up)
            else:
                d_raw = hb.raw2outputs_bwd(raw, z, rays, rays.shape[1], noise, std, wb, c(d_rgb), c(d_acc), c(d_disp),
                                           rays_d_offset=3)
                if d_raw_up is not None:
                    d_raw += d_raw_up
            hb.field_bwd(packed, act, d_raw, grad, wrote[key], precision=prec, params=model.flat_params())
            wrote[key] = True

        shared = ctx.same_net and fine and has(up_f)      # the fine pass adds into the coarse network's gradient

        def backprop(r, rays, rnd, lo, hi, last=True):
            if has(up_c):
                field_grad(rays, ctx.model_c, r["packed_c"], r["act_c"], r["raw_c"], r["z_c"], rnd.get("noise_c"), up_c, lo, hi, grad_c, "c")
                if last and not shared:     # final: its all-reduce may start under the fine network's backward
                    _grad_ready(ctx.model_c, grad_c)
            hb.WORKSPACE.give(r["act_c"])
            r["act_c"] = None
            if fine and has(up_f):
                if ctx.same_net:
                    field_grad(rays, ctx.model_c, r["packed_f"], r["act_f"], r["raw_f"], r["z_f"], rnd.get("noise_f"), up_f, lo, hi, grad_c, "c")
                    if last:
                        _grad_ready(ctx.model_c, grad_c)
                else:
                    field_grad(rays, ctx.model_f, r["packed_f"], r["act_f"], r["raw_f"], r["z_f"], rnd.get("noise_f"), up_f, lo, hi, grad_f, "f")
                    if last:
                        _grad_ready(ctx.model_f, grad_f)
            _release(r)

        if ctx.tiles is not None:
            for i, (lo, hi) in enumerate(ctx.tiles):
                backprop(ctx.saved[i], rays_all[lo:hi], {k: v[lo:hi] for k, v in rnd_all.items()}, lo, hi, last=i == len(ctx.tiles) - 1)
        elif not ctx.checkpoint:
            r = dict(ctx.saved)
            r["raw_f" if fine else "raw_c"] = ctx.saved_tensors[0]
            backprop(r, rays_all, rnd_all, 0, n_all)
            ctx.saved["act_c"] = ctx.saved["act_f"] = None
        else:
            step = ctx.sub_rays
            for lo in range(0, n_all, step):
                hi = min(lo + step, n_all)
                rays = rays_all[lo:hi]
                rnd = {k: v[lo:hi] for k, v in rnd_all.items()}
                backprop(_field_pass(cfg, rays, rnd, ctx.model_c, ctx.model_f, save=True), rays, rnd, lo, hi, last=hi == n_all)
        ctx.saved = None
        ctx.consumed = True
        out_c = none_c
        if wrote["c"]:
            out_c = _grad_views(ctx.model_c, grad_c)
        if ctx.same_net:
            return (None,) * n_lead + out_c
        out_f = none_c
        if wrote["f"]:
            out_f = _grad_views(ctx.model_f, grad_f)
        return (None,) * n_lead + out_c + out_f


class _Composite(torch.autograd.Function):
    """raw2outputs with a gradient w.r.t. raw (for callers that use it standalone)."""

    @staticmethod
    def forward(ctx, raw, z_vals, rays_d, noise, raw_noise_std, white_bkgd):
        rgb, disp, acc, w, depth = hb.raw2outputs(raw, z_vals, rays_d, 3, noise, raw_noise_std, white_bkgd)
        ctx.args = (raw, z_vals, rays_d, noise, raw_noise_std, white_bkgd)
        ctx.set_materialize_grads(False)
        return rgb, disp, acc, w, depth

    @staticmethod
    def backward(ctx, d_rgb, d_disp, d_acc, d_w, d_depth):
        # all five outputs carry gradients to raw, as in the reference (a depth / weight / sparsity loss term works)
        raw, z_vals, rays_d, noise, std, wb = ctx.args
        if all(g is None for g in (d_rgb, d_disp, d_acc, d_w, d_depth)):
            return (None,) * 6
        if d_rgb is None:
            d_rgb = torch.zeros((raw.shape[0], 3), dtype=torch.float32, device=raw.device)
        c = lambda t: t.to(torch.float32).contiguous() if t is not None else None
        d_raw = hb.raw2outputs_bwd(raw, z_vals, rays_d, 3, noise, std, wb, c(d_rgb), c(d_acc), c(d_disp),
                                   d_weights=c(d_w), d_depth=c(d_depth))
        return d_raw, None, None, None, None, None


# --------------------------------------------------------------------------- reference call surface
def query_points(model, pts, viewdirs_per_point):
    """Evaluate the field at explicit points: every point is its own ray record
    (o = pt, d = 0, z = 0  =>  o + d*z == pt exactly)."""
    pts = _f32c(pts)
    vd = _f32c(viewdirs_per_point)
    n = pts.shape[0]
    rays = torch.zeros((n, 11), dtype=torch.float32, device=pts.device)
    rays[:, 0:3] = pts
    rays[:, 8:11] = vd
    z = torch.zeros((n, 1), dtype=torch.float32, device=pts.device)
    plist = model.param_list()
    need = torch.is_grad_enabled() and any(p.requires_grad for p in plist)
    raw = _FieldQuery.apply(model, rays, z, need, *plist)
    return raw.reshape(n, 4)


def batchify(fn, chunk):
    """run_nerf.py:27-34."""
    if chunk is None:
        return fn

    def ret(inputs):
        return torch.cat([fn(inputs[i:i + chunk]) for i in range(0, inputs.shape[0], chunk)], 0)
    return ret


def run_network(inputs, viewdirs, fn, embed_fn, embeddirs_fn, netchunk=1024 * 64):
    """run_nerf.py:37-51.  inputs [N,S,3], viewdirs [N,3], fn a NeRF module.  The embedders are
    accepted for signature parity; encoding happens inside the fused kernel (and the
    netchunk loop disappears: the kernel tiles the points itself)."""
    from .dense import DenseNeRF
    if isinstance(fn, DenseNeRF):       # any other architecture: the reference's own composition (embed, broadcast, network)
        flat = torch.reshape(inputs, [-1, inputs.shape[-1]])
        embedded = embed_fn(flat)
        if viewdirs is not None:
            dirs = torch.reshape(viewdirs[:, None].expand(inputs.shape), [-1, inputs.shape[-1]])
            embedded = torch.cat([embedded, embeddirs_fn(dirs)], -1)
        out = batchify(fn, netchunk)(embedded)
        return torch.reshape(out, list(inputs.shape[:-1]) + [out.shape[-1]])
    if not isinstance(fn, NeRF):
        raise NotImplementedError("run_network: fn must be a nerf-pytorch_amd NeRF module")
    if viewdirs is None:
        raise ValueError("run_network: this network was built with use_viewdirs=True")
    flat = torch.reshape(inputs, [-1, inputs.shape[-1]])
    dirs = viewdirs[:, None].expand(inputs.shape)
    dirs_flat = torch.reshape(dirs, [-1, dirs.shape[-1]])
    out = query_points(fn, flat, dirs_flat)
    return torch.reshape(out, list(inputs.shape[:-1]) + [out.shape[-1]])


def raw2outputs(raw, z_vals, rays_d, raw_noise_std=0, white_bkgd=False, pytest=False):
    """run_nerf.py:262-305: (rgb_map, disp_map, acc_map, weights, depth_map)."""
    noise = None
    std = float(raw_noise_std)
    if raw_noise_std > 0.0:
        noise = torch.randn(raw[..., 3].shape, device=raw.device)
        if pytest:
            np.random.seed(0)
            noise = torch.Tensor(np.random.rand(*list(raw[..., 3].shape)) * raw_noise_std).to(raw.device)
            std = 1.0
        noise = noise.contiguous()
    raw_c = raw.to(torch.float32).contiguous()
    return _Composite.apply(raw_c, _f32c(z_vals), _f32c(rays_d), noise, std, bool(white_bkgd))


def sample_pdf(bins, weights, N_samples, det=False, pytest=False):
    """run_nerf_helpers.py:196-239 (output is a constant: the reference detaches it, run_nerf.py:394)."""
    bins, weights = _f32c(bins), _f32c(weights)
    dev = bins.device
    lead = list(bins.shape[:-1])
    u = None
    if not det:
        u = torch.rand(lead + [N_samples], device=dev)
    if pytest:
        np.random.seed(0)
        if det:
            u = torch.Tensor(np.broadcast_to(np.linspace(0.0, 1.0, N_samples), lead + [N_samples]).copy()).to(dev)
        else:
            u = torch.Tensor(np.random.rand(*(lead + [N_samples]))).to(dev)
    b2 = bins.reshape(-1, bins.shape[-1])
    w2 = weights.reshape(-1, weights.shape[-1])
    u2 = u.reshape(-1, N_samples).contiguous() if u is not None else None
    out = hb.sample_pdf(b2, w2, N_samples, u2, None if u2 is not None else _linspace01(N_samples, dev))
    return out.reshape(lead + [N_samples])


def builtin_query_fn(fn):
    """Mark `fn` as the stock network_query_fn (run_network with the stock embedders, run_nerf.py:201-204): render_rays then takes the
    fused path, in which encoding, network and netchunk tiling happen inside one kernel.  create_nerf marks the lambda it builds."""
    fn._nerf_amd_builtin = True
    return fn


def _is_builtin_query(network_query_fn):
    return network_query_fn is None or getattr(network_query_fn, "_nerf_amd_builtin", False)


def _render_rays_hooked(rays, rnd, network_fn, network_query_fn, N_samples, n_f, network_fine, lindisp, white_bkgd, std, retraw):
    """render_rays with a USER-SUPPLIED network_query_fn (run_nerf.py:385, :401 call it for every pass): the hook sees the reference's
    arguments -- pts [N, S, 3], viewdirs [N, 3], the network module -- and whatever it returns is composited, exactly as in the
    reference (a density regulariser, a clamp, a different network call all work).  Stage by stage through the C ABI: coarse depths
    (nerf_sample_coarse), the hook (stock run_network -> nerf_field_fwd / dgrad / wgrad per point), raw2outputs with its adjoint
    (nerf_raw2outputs[_bwd]), hierarchical depths (nerf_sample_fine: sample_pdf + sort, a constant as in run_nerf.py:394)."""
    dev = rays.device
    rays_o, rays_d, viewdirs = rays[:, 0:3], rays[:, 3:6].contiguous(), rays[:, 8:11]

    def one_pass(z_vals, net, noise):
        pts = rays_o[:, None, :] + rays_d[:, None, :] * z_vals[:, :, None]       # run_nerf.py:381 / :397
        raw = network_query_fn(pts, viewdirs, net)
        return raw, _Composite.apply(raw.to(torch.float32).contiguous(), z_vals, rays_d, noise, std, bool(white_bkgd))

    z_c = hb.sample_coarse(rays, _linspace01(N_samples, dev), lindisp, rnd.get("t_rand"))
    raw, (rgb, disp, acc, weights, _) = one_pass(z_c, network_fn, rnd.get("noise_c"))
    ret = {}
    if n_f > 0:
        ret.update(rgb0=rgb, disp0=disp, acc0=acc)
        u = rnd.get("u")
        z_f, z_std, _ = hb.sample_fine(z_c, weights.detach().contiguous(), n_f, u, None if u is not None else _linspace01(n_f, dev))
        raw, (rgb, disp, acc, _, _) = one_pass(z_f, network_fn if network_fine is None else network_fine, rnd.get("noise_f"))
        ret["z_std"] = z_std
    ret.update(rgb_map=rgb, disp_map=disp, acc_map=acc)
    if retraw:
        ret["raw"] = raw
    return ret


def render_rays(ray_batch, network_fn, network_query_fn, N_samples, retraw=False, lindisp=False, perturb=0.,
                N_importance=0, network_fine=None, white_bkgd=False, raw_noise_std=0., verbose=False, pytest=False,
                *, randoms=None):
    """run_nerf.py:308-418.  Same arguments, same returned dict.

    ``network_query_fn``: None or the function create_nerf built (builtin_query_fn) -> the fused path; ANY other callable is called
    for every pass with (pts, viewdirs, network) like the reference does (_render_rays_hooked).  ``verbose`` is accepted and prints
    nothing (run_nerf.py:414-416's DEBUG-gated NaN check: see render.check_range()); the reference's ``netchunk`` has no counterpart
    (the kernels tile the points themselves).

    ``randoms`` (keyword-only, not in the reference) injects the random tensors
    {t_rand [N,N_samples], noise_c [N,N_samples], u [N,N_importance], noise_f [N,N_samples+N_importance]}
    instead of drawing them: the explicit form of the reference's ``pytest=`` hook."""
    from .dense import DenseNeRF
    nets = [network_fn] + ([network_fine] if network_fine is not None else [])
    dense = all(isinstance(m, DenseNeRF) for m in nets)         # architectures outside the fused kernels: layer by layer (dense.py)
    if not dense and not all(isinstance(m, NeRF) for m in nets):
        raise NotImplementedError("render_rays: network_fn / network_fine must both be fused-kernel NeRF modules (D=8, W=256, 10 / 4 "
                                  "frequencies, view directions) or both general ones (nerf_pytorch_amd.NeRF builds either)")
    if not dense and ray_batch.shape[-1] <= 8:
        raise ValueError("render_rays: these networks use view directions; the ray records need 11 columns (render(use_viewdirs=True))")
    rays = ray_batch.to(torch.float32).contiguous().detach()
    n = rays.shape[0]
    dev = rays.device
    n_f = int(N_importance)
    if n == 0:      # empty batch: the reference returns empty tensors of the right trailing shapes
        e = lambda *tail: torch.zeros((0,) + tail, dtype=torch.float32, device=dev)
        ret = {'rgb_map': e(3), 'disp_map': e(), 'acc_map': e()}
        if retraw:
            ret['raw'] = e(N_samples + n_f, 4)
        if n_f > 0:
            ret.update(rgb0=e(3), disp0=e(), acc0=e(), z_std=e())
        return ret
    rnd = {}
    if randoms is not None:
        keys = (["t_rand"] if perturb > 0. else []) + (["noise_c"] if raw_noise_std > 0. else [])
        if n_f > 0:
            keys += (["u"] if perturb > 0. else []) + (["noise_f"] if raw_noise_std > 0. else [])
        rnd = {k: randoms[k].to(device=dev, dtype=torch.float32).contiguous() for k in keys}
        for k, v in rnd.items():
            if v.shape[0] != n:
                raise ValueError(f"render_rays: randoms[{k!r}] has {v.shape[0]} rows for {n} rays")
        perturb_draw = 0.
    else:
        perturb_draw = perturb
    # draw order of the reference: t_rand (:371) -> noise coarse (:285) -> u (helpers:208) -> noise fine (:285)
    if perturb_draw > 0.:
        rnd["t_rand"] = torch.rand((n, N_samples), device=dev)
        if pytest:
            np.random.seed(0)
            rnd["t_rand"] = torch.Tensor(np.random.rand(n, N_samples)).to(dev)
    std = float(raw_noise_std)

    def draw_noise(S):
        nz = torch.randn((n, S), device=dev)
        if pytest:
            np.random.seed(0)
            nz = torch.Tensor(np.random.rand(n, S) * raw_noise_std).to(dev)
        return nz.contiguous()
    if raw_noise_std > 0. and randoms is None:
        rnd["noise_c"] = draw_noise(N_samples)
    if n_f > 0:
        if perturb_draw > 0.:
            rnd["u"] = torch.rand((n, n_f), device=dev)
            if pytest:
                np.random.seed(0)
                rnd["u"] = torch.Tensor(np.random.rand(n, n_f)).to(dev)
        elif pytest and randoms is None:
            # det + pytest: the reference builds u with np.linspace in float64 and casts it (helpers:213-215), which is NOT
            # torch.linspace's fp32 sequence (one ulp apart in 30 of 64 / 8 of 128 entries: tests/test_host_cpu.py) -- hand the
            # kernel the reference's numbers as explicit draws
            rnd["u"] = torch.Tensor(np.broadcast_to(np.linspace(0., 1., n_f), (n, n_f)).copy()).to(dev)
        if raw_noise_std > 0. and randoms is None:
            rnd["noise_f"] = draw_noise(N_samples + n_f)
    if pytest and raw_noise_std > 0. and randoms is None:
        std = 1.0       # pytest noise is pre-scaled in float64 like the reference (run_nerf.py:290)
    cfg = dict(N_samples=int(N_samples), N_importance=n_f, lindisp=bool(lindisp), white_bkgd=bool(white_bkgd),
               raw_noise_std=std, precision=_PRECISION)
    if not _is_builtin_query(network_query_fn):
        return _render_rays_hooked(rays, rnd, network_fn, network_query_fn, int(N_samples), n_f, network_fine if n_f > 0 else None,
                                   bool(lindisp), white_bkgd, std, retraw)
    if dense:
        from .dense import render_rays_dense
        ret = render_rays_dense(cfg, rays, rnd, network_fn, network_fine if n_f > 0 else None)
        if not retraw:
            ret.pop("raw")
        return ret
    params = network_fn.param_list()
    same = network_fine is None or network_fine is network_fn
    if n_f > 0 and not same:
        params = params + network_fine.param_list()
    # activations are saved only when a backward can follow (Function.forward itself always runs in no-grad mode)
    cfg["need_grad"] = torch.is_grad_enabled() and any(p.requires_grad for p in params)
    if cfg["precision"] == "fp16_fp8c" and cfg["need_grad"]:
        cfg["precision"] = "fp16x3"         # the reduced class is an inference form; gradients: the fp16x3 datapath, unchanged
    outs = _RenderRays.apply(cfg, rays, rnd, network_fn, None if (same or n_f <= 0) else network_fine, *params)
    if n_f <= 0:
        rgb_map, disp_map, acc_map, raw = outs
        ret = {'rgb_map': rgb_map, 'disp_map': disp_map, 'acc_map': acc_map}
        if retraw:
            ret['raw'] = raw
        return ret
    rgb_map, disp_map, acc_map, raw, rgb0, disp0, acc0, z_std = outs
    ret = {'rgb_map': rgb_map, 'disp_map': disp_map, 'acc_map': acc_map}
    if retraw:
        ret['raw'] = raw
    ret['rgb0'] = rgb0
    ret['disp0'] = disp0
    ret['acc0'] = acc0
    ret['z_std'] = z_std
    return ret


def batchify_rays(rays_flat, chunk=1024 * 32, **kwargs):
    """run_nerf.py:54-66.  Injected ``randoms`` (one row per ray) are sliced with the rays, so a chunked call consumes
    the same draws as an unchunked one."""
    all_ret = {}
    randoms = kwargs.pop("randoms", None)
    if randoms is not None:
        for k, v in randoms.items():
            if v.shape[0] != rays_flat.shape[0]:
                raise ValueError(f"randoms[{k!r}] has {v.shape[0]} rows for {rays_flat.shape[0]} rays")
    for i in range(0, rays_flat.shape[0], chunk):
        if randoms is not None:
            kwargs["randoms"] = {k: v[i:i + chunk] for k, v in randoms.items()}
        ret = render_rays(rays_flat[i:i + chunk], **kwargs)
        for k in ret:
            all_ret.setdefault(k, []).append(ret[k])
    return {k: (v[0] if len(v) == 1 else torch.cat(v, 0)) for k, v in all_ret.items()}


# ---- ray geometry: per-ray (not per-sample) work, stays PyTorch behind render()
def get_rays(H, W, K, c2w):
    """run_nerf_helpers.py:153-162."""
    dev = c2w.device if isinstance(c2w, torch.Tensor) else None
    i, j = torch.meshgrid(torch.linspace(0, W - 1, W, device=dev), torch.linspace(0, H - 1, H, device=dev), indexing='ij')
    i = i.t()
    j = j.t()
    dirs = torch.stack([(i - K[0][2]) / K[0][0], -(j - K[1][2]) / K[1][1], -torch.ones_like(i)], -1)
    rays_d = torch.sum(dirs[..., None, :] * c2w[:3, :3], -1)
    rays_o = c2w[:3, -1].expand(rays_d.shape)
    return rays_o, rays_d


def get_rays_np(H, W, K, c2w):
    """run_nerf_helpers.py:165-172."""
    i, j = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing='xy')
    dirs = np.stack([(i - K[0][2]) / K[0][0], -(j - K[1][2]) / K[1][1], -np.ones_like(i)], -1)
    rays_d = np.sum(dirs[..., np.newaxis, :] * c2w[:3, :3], -1)
    rays_o = np.broadcast_to(c2w[:3, -1], np.shape(rays_d))
    return rays_o, rays_d


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """run_nerf_helpers.py:175-192."""
    t = -(near + rays_o[..., 2]) / rays_d[..., 2]
    rays_o = rays_o + t[..., None] * rays_d
    o0 = -1. / (W / (2. * focal)) * rays_o[..., 0] / rays_o[..., 2]
    o1 = -1. / (H / (2. * focal)) * rays_o[..., 1] / rays_o[..., 2]
    o2 = 1. + 2. * near / rays_o[..., 2]
    d0 = -1. / (W / (2. * focal)) * (rays_d[..., 0] / rays_d[..., 2] - rays_o[..., 0] / rays_o[..., 2])
    d1 = -1. / (H / (2. * focal)) * (rays_d[..., 1] / rays_d[..., 2] - rays_o[..., 1] / rays_o[..., 2])
    d2 = -2. * near / rays_o[..., 2]
    return torch.stack([o0, o1, o2], -1), torch.stack([d0, d1, d2], -1)


def render(H, W, K, chunk=1024 * 32, rays=None, c2w=None, ndc=True, near=0., far=1., use_viewdirs=False,
           c2w_staticcam=None, **kwargs):
    """run_nerf.py:69-134: [rgb_map, disp_map, acc_map, extras].  With c2w the ray records are built by one HIP
    launch (get_rays + view directions + ndc_rays + near / far: SURVEY 8 f-2), no [H,W,3] intermediates."""
    # (use_viewdirs=False: the 11-column ray records are built all the same; networks without view directions ignore columns 8-10)
    if c2w is not None:
        net = kwargs.get("network_fn")
        dev = next(net.parameters()).device if net is not None else (c2w.device if isinstance(c2w, torch.Tensor) else None)
        rays = hb.make_rays(H, W, K, c2w, c2w_staticcam, ndc, near, far, dev)
        sh = (H, W, 3)
    elif (c2w_staticcam is None and isinstance(rays[0], torch.Tensor) and rays[0].is_cuda
          and isinstance(near, (int, float)) and isinstance(far, (int, float))):
        # one launch: view directions + NDC warp + near / far columns -> [N, 11] records (run_nerf.py:100-123)
        rays_o, rays_d = rays
        sh = rays_d.shape
        rays = hb.assemble_rays(_f32c(rays_o).reshape(-1, 3), _f32c(rays_d).reshape(-1, 3), ndc, H, W, K[0][0], near, far)
    else:
        rays_o, rays_d = rays
        viewdirs = rays_d
        if c2w_staticcam is not None:       # run_nerf.py:103-105
            rays_o, rays_d = get_rays(H, W, K, c2w_staticcam)
        viewdirs = viewdirs / torch.norm(viewdirs, dim=-1, keepdim=True)
        viewdirs = torch.reshape(viewdirs, [-1, 3]).float()
        sh = rays_d.shape
        if ndc:
            rays_o, rays_d = ndc_rays(H, W, K[0][0], 1., rays_o, rays_d)
        rays_o = torch.reshape(rays_o, [-1, 3]).float()
        rays_d = torch.reshape(rays_d, [-1, 3]).float()
        near, far = near * torch.ones_like(rays_d[..., :1]), far * torch.ones_like(rays_d[..., :1])
        rays = torch.cat([rays_o, rays_d, near, far, viewdirs], -1)
    all_ret = batchify_rays(rays, chunk, **kwargs)
    for k in all_ret:
        k_sh = list(sh[:-1]) + list(all_ret[k].shape[1:])
        all_ret[k] = torch.reshape(all_ret[k], k_sh)
    k_extract = ['rgb_map', 'disp_map', 'acc_map']
    ret_list = [all_ret[k] for k in k_extract]
    ret_dict = {k: all_ret[k] for k in all_ret if k not in k_extract}
    return ret_list + [ret_dict]


to8b = lambda x: (255 * np.clip(x, 0, 1)).astype(np.uint8)
_MSE_SCRATCH = {}


class _Img2Mse(torch.autograd.Function):
    """run_nerf_helpers.py:11 on the device in one launch (+ one for the gradient) instead of sub / pow / mean and their three
    backward kernels: the loss of run_nerf.py:765-772 is evaluated twice per step."""

    @staticmethod
    def forward(ctx, x, y):
        xc, yc = x.contiguous(), y.contiguous()
        key = (str(x.device), hb._stream())         # (per stream: the kernel's ticket word must not be shared by concurrent launches)
        if key not in _MSE_SCRATCH:
            _MSE_SCRATCH[key] = torch.zeros(hb.lib().nerf_mse_scratch_floats(), dtype=torch.float32, device=x.device)
        out = torch.empty((), dtype=torch.float32, device=x.device)
        hb._check(hb.lib().nerf_mse_fwd(xc.data_ptr(), yc.data_ptr(), xc.numel(), _MSE_SCRATCH[key].data_ptr(), out.data_ptr(), hb._stream()),
                  "nerf_mse_fwd")
        ctx.save_for_backward(xc, yc)
        return out

    @staticmethod
    def backward(ctx, g):
        xc, yc = ctx.saved_tensors
        dx = torch.empty_like(xc)
        hb._check(hb.lib().nerf_mse_bwd(xc.data_ptr(), yc.data_ptr(), xc.numel(), g.to(torch.float32).contiguous().data_ptr(), dx.data_ptr(),
                                        hb._stream()), "nerf_mse_bwd")
        return dx, (-dx if ctx.needs_input_grad[1] else None)


def img2mse(x, y):
    """run_nerf_helpers.py:11.  fp32 tensors of one shape on the GPU: one HIP launch; anything else (CPU tensors, broadcasting,
    other dtypes): the reference's expression."""
    if (isinstance(x, torch.Tensor) and isinstance(y, torch.Tensor) and x.is_cuda and y.is_cuda and x.dtype == torch.float32
            and y.dtype == torch.float32 and x.shape == y.shape and x.numel() > 0):
        return _Img2Mse.apply(x, y)
    return torch.mean((x - y) ** 2)


_LOG10 = {}


def mse2psnr(x):
    """run_nerf_helpers.py:12, same arithmetic and result shape ([1]); log(10) lives on x's device once instead of being uploaded at
    every call (a pageable host-to-device copy per step would make the host wait for the stream in the train() loop)"""
    t = _LOG10.get(x.device)
    if t is None:
        t = _LOG10[x.device] = torch.log(torch.tensor([10.])).to(x.device)
    return -10. * torch.log(x) / t


def _write_png(path, rgb8):
    """Minimal PNG writer (imageio is not a dependency of the hot path)."""
    import struct
    import zlib
    h, w, c = rgb8.shape
    rows = b"".join(b"\x00" + rgb8[y].tobytes() for y in range(h))

    def chunk(tag, data):
        body = tag + data
        return struct.pack(">I", len(data)) + body + struct.pack(">I", zlib.crc32(body) & 0xffffffff)
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2 if c == 3 else 6, 0, 0, 0))
    png += chunk(b"IDAT", zlib.compress(rows, 6)) + chunk(b"IEND", b"")
    with open(path, "wb") as f:
        f.write(png)


class _FrameSink:
    """Output side of render_path.  The reference blocks on `.cpu().numpy()` and encodes every frame on
    the rendering thread (run_nerf.py:155-169).  Here frame i's device->host copies go to pinned memory asynchronously,
    `to8b` runs on the device, and the PNG of frame i-1 is encoded by a worker thread while frame i+1 renders; the
    returned arrays and files are the same."""

    def __init__(self, savedir=None, workers=2):
        self.savedir = savedir
        self.rgbs, self.disps, self.jobs = [], [], []
        self.pending = None
        self.pool = None
        if savedir is not None:
            from concurrent.futures import ThreadPoolExecutor
            self.pool = ThreadPoolExecutor(max_workers=workers)

    @staticmethod
    def _to_host(t):
        if not t.is_cuda:
            return t.detach().clone()
        h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        h.copy_(t.detach(), non_blocking=True)
        return h

    def push(self, index, rgb, disp):
        rgb8 = None
        if self.savedir is not None:
            rgb8 = self._to_host((255 * rgb.detach().clamp(0, 1)).to(torch.uint8))      # to8b (helpers:11) on the device
        entry = (index, self._to_host(rgb), self._to_host(disp), rgb8)
        event = None
        if rgb.is_cuda:
            event = torch.cuda.Event()
            event.record()
        self._finish()                      # the previous frame: its copies had a whole frame time to land
        self.pending = (entry, event)

    def _finish(self):
        if self.pending is None:
            return
        (index, rgb, disp, rgb8), event = self.pending
        self.pending = None
        if event is not None:
            event.synchronize()
        self.rgbs.append(rgb.numpy())
        self.disps.append(disp.numpy())
        if rgb8 is not None:
            import os
            path = os.path.join(self.savedir, '{:03d}.png'.format(index))
            self.jobs.append(self.pool.submit(_write_png, path, rgb8.numpy()))

    def close(self):
        self._finish()
        for j in self.jobs:
            j.result()                      # re-raises an encoder / IO error
        if self.pool is not None:
            self.pool.shutdown()
        return np.stack(self.rgbs, 0), np.stack(self.disps, 0)


def render_path(render_poses, hwf, K, chunk, render_kwargs, gt_imgs=None, savedir=None, render_factor=0):
    """run_nerf.py:137-175: (rgbs[F,H,W,3], disps[F,H,W]) as numpy."""
    import os
    H, W, focal = hwf
    if render_factor != 0:
        H = H // render_factor
        W = W // render_factor
        focal = focal / render_factor
    sink = _FrameSink(savedir)
    for i, c2w in enumerate(render_poses):
        rgb, disp, acc, _ = render(H, W, K, chunk=chunk, c2w=c2w[:3, :4], **render_kwargs)
        sink.push(i, rgb, disp)
    return sink.close()

