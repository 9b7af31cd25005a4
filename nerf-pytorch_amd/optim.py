"""Fused Adam over the flat parameter vectors.

``FlatAdam`` is a ``torch.optim.Optimizer`` whose ``state_dict()`` has exactly the layout of
``torch.optim.Adam`` (per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq``; same param_group keys), so the
reference's checkpoints (run_nerf.py:792-800) load into it and vice versa -- but the moments of all parameters
that live in one flat vector (a ``NeRF`` module) are views into one flat buffer each, and ``step()`` is ONE HIP
launch per flat vector instead of six foreach kernels over 48 tensors (run_nerf.py:776).
"""
import torch

from . import hip_backend as hb


def _segments(params):
    """Group parameters into maximal runs that are contiguous in memory (same storage, back to back)."""
    ps = sorted([p for p in params], key=lambda p: p.data_ptr())
    segs, cur = [], []
    for p in ps:
        if cur and p.data_ptr() == cur[-1].data_ptr() + 4 * cur[-1].numel() and p.dtype == torch.float32 \
                and p.untyped_storage().data_ptr() == cur[-1].untyped_storage().data_ptr():
            cur.append(p)
        else:
            if cur:
                segs.append(cur)
            cur = [p]
    if cur:
        segs.append(cur)
    return segs


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, maximize=False, foreach=None,
                        capturable=False, differentiable=False, fused=None, decoupled_weight_decay=False)
        super().__init__(params, defaults)
        self._flat = {}     # id(first param of segment) -> (segment params, flat exp_avg, flat exp_avg_sq)

    def _segment_state(self, seg):
        key = (seg[0].data_ptr(), len(seg))
        if key not in self._flat:
            n = sum(p.numel() for p in seg)
            dev = seg[0].device
            m = torch.zeros(n, dtype=torch.float32, device=dev)
            v = torch.zeros(n, dtype=torch.float32, device=dev)
            off = 0
            for p in seg:
                st = self.state[p]
                k = p.numel()
                mv, vv = m[off:off + k].view_as(p), v[off:off + k].view_as(p)
                if "exp_avg" in st:                       # state loaded from a checkpoint: adopt its values
                    mv.copy_(st["exp_avg"])
                    vv.copy_(st["exp_avg_sq"])
                st["exp_avg"], st["exp_avg_sq"] = mv, vv
                st.setdefault("step", torch.tensor(0.0))
                off += k
            self._flat[key] = (m, v)
        return self._flat[key]

    _UNSUPPORTED = (("weight_decay", 0), ("amsgrad", False), ("maximize", False), ("decoupled_weight_decay", False))

    def _check_groups(self):
        """The fused kernel is plain Adam (run_nerf.py:207).  The other torch.optim.Adam keys exist in the groups only
        so that state_dicts interchange; a group that sets one of them must not silently train as plain Adam."""
        for group in self.param_groups:
            for key, off in self._UNSUPPORTED:
                if group.get(key, off) not in (off, None):
                    raise NotImplementedError(f"FlatAdam: {key}={group[key]!r} is not implemented by the fused kernel "
                                              "(use torch.optim.Adam for this configuration)")

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._check_groups()
        self._flat = {}        # moments are re-flattened (values adopted) at the next step

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._check_groups()
        for group in self.param_groups:
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            b1, b2 = group["betas"]
            for seg in _segments(params):
                m, v = self._segment_state(seg)
                n = m.numel()
                first = seg[0]
                pflat = torch.as_strided(first.data, (n,), (1,))        # the run is contiguous in memory
                g0 = first.grad
                contiguous = all(q.grad.data_ptr() == g0.data_ptr() + 4 * sum(r.numel() for r in seg[:i])
                                 for i, q in enumerate(seg))
                gflat = torch.as_strided(g0, (n,), (1,)) if contiguous and g0.is_contiguous() else \
                    torch.cat([q.grad.reshape(-1) for q in seg])
                step_t = self.state[first]["step"]
                step = int(step_t.item()) + 1
                if first.is_cuda:
                    hb.adam_step(pflat, gflat.contiguous(), m, v, group["lr"], b1, b2, group["eps"], step)
                else:       # CPU tensors (tests of the bookkeeping): same arithmetic with torch ops
                    m.lerp_(gflat, 1 - b1)
                    v.mul_(b2).addcmul_(gflat, gflat, value=1 - b2)
                    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
                    pflat.addcdiv_(m, (v.sqrt() / (bc2 ** 0.5)).add_(group["eps"]), value=-group["lr"] / bc1)
                for q in seg:
                    self.state[q]["step"] = torch.tensor(float(step))
        return loss
