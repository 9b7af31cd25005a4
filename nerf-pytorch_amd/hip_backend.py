"""ctypes binding of libnerf_hip.so (C ABI: include/nerf_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; every function
below hands raw device pointers of fp32 CUDA(ROCm) tensors to the library and
enqueues HIP kernels on ``torch.cuda.current_stream()``.

There is no fallback: if the library is missing or a call fails, this raises.
"""
import ctypes
import os

import torch  # import torch BEFORE loading the library: one HIP runtime per process (torch's bundled one)

from . import build as _build

_c_float_p = ctypes.c_void_p
_LIB = None
ABI_VERSION = 10       # NERF_ABI_VERSION of include/nerf_hip.h this binding was written against


class NerfHipError(RuntimeError):
    pass


def _declare(lib):
    i, l, f, p, sz = ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t
    sig = {
        "nerf_abi_version": (i, []),
        "nerf_last_error": (ctypes.c_char_p, []),
        "nerf_param_count": (i, []),
        "nerf_param_offset": (i, [i, ctypes.POINTER(i), ctypes.POINTER(i)]),
        "nerf_packed_floats": (i, []),
        "nerf_pack_params": (i, [p, p, p]),
        "nerf_debug_pack_table": (i, [p]),
        "nerf_embed": (i, [p, l, i, p, p]),
        "nerf_sample_coarse": (i, [p, i, i, p, i, i, p, p, p]),
        "nerf_sample_ray_batch": (i, [i, i, p, p, i, p, i, i, i, i, i, ctypes.c_uint, ctypes.c_uint, p, p, p, p]),
        "nerf_make_rays": (i, [i, i, p, p, p, i, f, f, p, i, p]),
        "nerf_assemble_rays": (i, [p, p, l, i, i, i, f, f, f, p, i, p]),
        "nerf_buffer_layout": (i, [p, ctypes.POINTER(i), ctypes.POINTER(i), ctypes.POINTER(i)]),
        "nerf_debug_layout": (i, [i, i, i, i, ctypes.POINTER(ctypes.c_longlong)]),
        "nerf_act_floats": (sz, [i, i]),
        "nerf_workspace_floats": (sz, [i, i, i, i]),
        "nerf_pack_params_split_pair": (i, [p, p, p, p, i, i, p]),
        "nerf_act_floats_dp": (sz, [i, i, i]),
        "nerf_delta_floats_dp": (sz, [i, i, i]),
        "nerf_workspace_floats_dp": (sz, [i, i, i, i, i]),
        "nerf_field_fwd": (i, [p, p, i, p, i, i, p, p, p]),
        "nerf_raw2outputs": (i, [p, p, p, i, i, i, p, f, i, p, p, p, p, p, p]),
        "nerf_raw2outputs_bwd": (i, [p, p, p, i, i, i, p, f, i, p, p, p, p, p, p, p]),
        "nerf_sample_fine": (i, [p, p, i, i, i, p, p, p, p, p, p]),
        "nerf_sample_pdf": (i, [p, p, i, i, i, p, p, p, p]),
        "nerf_delta_floats": (sz, [i, i]),
        "nerf_wgrad_partial_floats": (sz, [i, i]),
        "nerf_field_bwd": (i, [p, p, p, i, i, p, p, p, i, p]),
        "nerf_field_dgrad": (i, [p, p, p, i, i, p, p]),
        "nerf_field_wgrad": (i, [p, p, p, i, i, p, p, i, p]),
        "nerf_packed3_floats": (i, []),
        "nerf_debug_pack3_table": (i, [p]),
        "nerf_field_wgrad_phase": (i, [p, p, p, i, i, p, p, i, i, i, p, p]),
        "nerf_pack_params_split": (i, [p, p, i, i, p]),
        "nerf_field_fwd_split": (i, [p, p, i, p, i, i, p, p, i, p]),
        "nerf_field_dgrad_split": (i, [p, p, p, i, i, p, i, p]),
        "nerf_field_fwd_last_sample": (i, [p, p, i, p, i, i, p, p, p, i, p]),
        "nerf_debug_pack16_table": (i, [p]),
        "nerf_adam_step": (i, [p, p, p, p, i, f, f, f, f, i, p]),
        "nerf_render_workspace_floats": (sz, [p, i, i]),
        "nerf_render_rays_fwd": (i, [p, p, p, p, i, i, p, p, p, p, p, p, p, p, p, p, p, p, p, i, p]),
        "nerf_render_rays_bwd": (i, [p, p, p, p, p, p, i, i, p, p, p, p, p, p, p, p, p, p, p, p, p, i, p]),
        "nerf_render_infer_supported": (i, [p]),
        "nerf_mse_scratch_floats": (i, []),
        "nerf_mse_fwd": (i, [p, p, l, p, p, p]),
        "nerf_mse_bwd": (i, [p, p, l, p, p, p]),
        "nerf_build_inputs": (i, [p, i, p, i, i, i, i, i, p, i, p]),
        "nerf_dense_fwd": (i, [p, i, i, p, i, p, p, i, i, l, i, i, p]),
        "nerf_dense_dgrad": (i, [p, i, i, p, i, p, i, i, l, i, p, i, p]),
        "nerf_dense_wgrad_scratch_floats": (sz, [l, i]),
        "nerf_dense_wgrad": (i, [p, i, i, p, i, i, l, p, i, p, p, i, p]),
        "nerf_render_rays_infer": (i, [p, p, p, p, i, i, p, p, p, p, p, p, p, p, p, p, p, p, p, p]),
        "nerf_range_scan": (i, [p, i, i, p, p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)      # AttributeError here = header / library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    return sig


EXPORTS = ["nerf_abi_version", "nerf_last_error", "nerf_param_count", "nerf_param_offset", "nerf_packed_floats",
           "nerf_pack_params", "nerf_debug_pack_table", "nerf_embed", "nerf_make_rays", "nerf_assemble_rays", "nerf_sample_coarse", "nerf_sample_ray_batch", "nerf_buffer_layout", "nerf_debug_layout", "nerf_act_floats", "nerf_workspace_floats", "nerf_field_fwd",
           "nerf_raw2outputs", "nerf_raw2outputs_bwd", "nerf_sample_fine", "nerf_sample_pdf", "nerf_delta_floats", "nerf_pack_params_split_pair", "nerf_act_floats_dp", "nerf_delta_floats_dp", "nerf_workspace_floats_dp",
           "nerf_wgrad_partial_floats", "nerf_field_bwd", "nerf_field_dgrad", "nerf_field_wgrad",
           "nerf_packed3_floats", "nerf_debug_pack3_table",
           "nerf_field_wgrad_phase", "nerf_debug_pack16_table",
           "nerf_pack_params_split", "nerf_field_fwd_split", "nerf_field_dgrad_split", "nerf_field_fwd_last_sample",
           "nerf_adam_step",
           "nerf_render_workspace_floats", "nerf_render_rays_fwd", "nerf_render_rays_bwd", "nerf_render_infer_supported",
           "nerf_render_rays_infer", "nerf_mse_scratch_floats", "nerf_mse_fwd", "nerf_mse_bwd", "nerf_build_inputs", "nerf_dense_fwd", "nerf_dense_dgrad", "nerf_dense_wgrad_scratch_floats",
           "nerf_dense_wgrad", "nerf_range_scan"]


def lib():
    """Load (once) and return the shared library.  Raises if it has not been built."""
    global _LIB
    if _LIB is None:
        # (NERF_HIP_LIB: another build of the same ABI -- kernel A/B timing on one box, tools/time_kernels.py)
        path = os.environ.get("NERF_HIP_LIB") or _build.LIB_PATH
        if not os.path.exists(path):
            raise NerfHipError(
                f"{path} not found: build it with `python __graft_entry__.py build` (hipcc --offload-arch=gfx950). "
                "There is no PyTorch fallback for the render hot path.")
        _LIB = ctypes.CDLL(path)
        _declare(_LIB)
        if _LIB.nerf_abi_version() != ABI_VERSION:
            raise NerfHipError("libnerf_hip.so ABI version mismatch")
    return _LIB


def _check(rc, name):
    if rc != 0:
        raise NerfHipError(f"{name} failed (code {rc}): {lib().nerf_last_error().decode()}")


def _ptr(t, name="tensor", optional=False):
    if t is None:
        if optional:
            return None
        raise NerfHipError(f"{name} is required")
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise NerfHipError(f"{name} must be a contiguous fp32 tensor on the GPU "
                           f"(got {type(t).__name__} {getattr(t, 'dtype', None)} {getattr(t, 'device', None)})")
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


class KernelTimer:
    """Optional HIP-event bracketing of the heavy launches (bench.py).  Events are recorded on torch's
    current stream, which is the stream the kernels are enqueued on.  `work` = algorithmic FLOPs."""

    def __init__(self):
        self.records = []       # (name, start_event, end_event, algorithmic flops, algorithmic HBM bytes)

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, e0, e1, fl, by in self.records:
            d = out.setdefault(name, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += fl
            d["bytes"] += by
        return out


TIMER = None        # set to a KernelTimer() to enable


class _timed:
    def __init__(self, name, flops, nbytes=0.0):
        self.name, self.flops, self.nbytes = name, flops, nbytes

    def __enter__(self):
        if TIMER is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *exc):
        if TIMER is not None:
            self.e1.record()
            TIMER.records.append((self.name, self.e0, self.e1, self.flops, self.nbytes))


FLOP_FWD_PER_POINT = 2 * 593408
FLOP_DGRAD_PER_POINT = 2 * 557696
FLOP_WGRAD_PER_POINT = 2 * 593408
FLOP_WGRAD_BIG_PER_POINT = 2 * 8 * 256 * 256            # the eight full-width jobs
# algorithmic HBM bytes per point (SURVEY §8d / DESIGN.md §2): what a launch must move once
BYTES_ACT_PER_POINT = 4 * (9 * 256 + 128 + 64 + 32) + 32 * 9 + 16  # saved activations + encodings + ReLU bitmasks (256 bits x 9 layers) + raw
BYTES_DELTA_PER_POINT = 4 * (9 * 256 + 128)             # deltas written by dgrad
BYTES_WGRAD_BIG_PER_POINT = 4 * 8 * (256 + 256)         # each full-width job reads its delta and its input once
BYTES_WGRAD_SMALL_PER_POINT = 4 * (2 * (256 + 64) + (4 + 256) + (128 + 256) + (128 + 32) + (4 + 128))
# split datapaths: feature_linear is folded into the view branch (csrc/nerf_common.h): one 256x256 layer
# less is EXECUTED in each of the three kernels, `feature` and its delta are neither written nor re-read
FOLD_MAC = 256 * 256
FLOP_FWD3_PER_POINT = 2 * (593408 - FOLD_MAC)
FLOP_DGRAD3_PER_POINT = 2 * (557696 - FOLD_MAC)
FLOP_WGRAD3_PER_POINT = 2 * (593408 - FOLD_MAC)
BYTES_ACT3_PER_POINT = BYTES_ACT_PER_POINT - 4 * 256
BYTES_DELTA3_PER_POINT = BYTES_DELTA_PER_POINT - 4 * 256
BYTES_ACT3_BF16_PER_POINT = 2 * (8 * 256 + 128 + 64) + 32 * 9 + 16  # bf16 rows + encodings, ReLU bitmasks, raw
BYTES_DELTA3_BF16_PER_POINT = 2 * (8 * 256 + 128 + 4)
BYTES_ACT3_LO_PER_POINT = 2 * (8 * 256 + 128 + 64)        # "fp16x3w": the lo words of the saved rows and of the xyz encoding
# 12 jobs on 16-bit operands (round 5: the alpha head's row rides on the (delta_hv, h7) job -- h7 is not re-read for it): 9,808 B per
# point (PMC: 9.83 KB)
BYTES_WGRAD_MIXED_PER_POINT = 0.5 * (BYTES_WGRAD_BIG_PER_POINT + BYTES_WGRAD_SMALL_PER_POINT - 4 * (256 + 256)) - 2 * 256


N_PARAMS = 595844


def param_table():
    """[(name, offset, shape)] of the flat parameter vector, from the library itself."""
    L = lib()
    names = []
    for i in range(8):
        names += [f"pts_linears.{i}.weight", f"pts_linears.{i}.bias"]
    names += ["views_linears.0.weight", "views_linears.0.bias", "feature_linear.weight", "feature_linear.bias",
              "alpha_linear.weight", "alpha_linear.bias", "rgb_linear.weight", "rgb_linear.bias"]
    out = []
    for idx, nm in enumerate(names):
        r, c = ctypes.c_int(), ctypes.c_int()
        off = L.nerf_param_offset(idx, ctypes.byref(r), ctypes.byref(c))
        shape = (r.value, c.value) if nm.endswith("weight") else (r.value,)
        out.append((nm, off, shape))
    return out


def pack_table():
    """Host-side gather table of the fragment repack (numpy int32, -1 = zero padding)."""
    import numpy as np
    L = lib()
    tab = np.empty(L.nerf_packed_floats(), dtype=np.int32)
    _check(L.nerf_debug_pack_table(tab.ctypes.data_as(ctypes.c_void_p)), "nerf_debug_pack_table")
    return tab


# "fp16x3" / "bf16x3": the three-term split W x = W_hi x_hi + W_hi x_lo + W_lo x_hi with fp16 / bf16 parts (csrc/split_types.h) on the
# weight-ring kernels; the operands of the weight-gradient GEMM are the stored hi words (11 / 8 significant bits).
# "fp16x3" (round 4) = the same three-term split with IEEE-half parts (csrc/split_types.h): ~2^-22 per product instead of 2^-17
# and 11-bit instead of 8-bit operands for the weight-gradient GEMM, at the bf16 MFMA count; needs |activations| < 65520.
# "fp16_fp8c" (round 4, INFERENCE class): no_grad rendering with every product of the 256-wide layers as fp16 main term + two fp8
# correction terms (csrc/field_ring8.h: ~2^-15 per product, 2 instead of 3 MFMA-equivalents; every ray's last sample re-evaluated
# with the three-term fp16 products); anything that needs gradients runs the fp16x3 datapath unchanged.  Never the bench headline.
# "fp16x3w" (round 6): fp16x3 with TWO-WORD operands in the weight-gradient GEMM -- the forward and the delta chain save the lo words
# next to the hi words and the GEMM contracts d_hi X_hi + d_hi X_lo + d_lo X_hi (the forward's product class, ~2^-22, instead of 11-bit
# operands) at twice the saved bytes and three times the GEMM's MFMAs.  Forward values are bit-identical to fp16x3's.  Not the default:
# it is the instrument that prices the one-word operand storage (profiles/r06_*).
PRECISIONS = ("fp32", "fp16x3", "bf16x3", "fp16_fp8c", "fp16x3w")
SPLIT = ("fp16x3", "bf16x3", "fp16_fp8c", "fp16x3w")       # datapaths on the three-term-split kernels (folded feature layer, tiled saves)
PACK_OF = {"fp16x3w": "fp16x3"}                 # datapaths that read another datapath's fragment repack


def pack_table3():
    """Host-side gather table of the TRANSPOSED (hi, lo) fragment streams of the delta chain (the first region of the packed3 buffer:
    W'^T | feature_linear^T | L7^T .. L1^T): per 16-bit element 2*canonical_index + is_lo, -1 = padding."""
    import numpy as np
    L = lib()
    # packed3 = transposed (hi, lo) streams | fp32 small parameters | 16-point forward stream (P16F) | derived W', b'
    n16 = 2 * (L.nerf_packed3_floats() - (L.nerf_packed_floats() - _small_offset()) - P16F_WORDS - N_DERIVED)
    tab = np.empty(n16, dtype=np.int32)
    _check(L.nerf_debug_pack3_table(tab.ctypes.data_as(ctypes.c_void_p)), "nerf_debug_pack3_table")
    return tab


P16F_WORDS = 593920        # csrc/nerf_common.h: words of the 16-point forward stream
N_DERIVED = 128 * 256 + 128  # csrc/nerf_common.h: W' = Wv[:, :256] Wf and b' (folded feature layer), appended to packed3;
#                              in the pack tables they are "canonical" indices N_PARAMS + k*256 + j, N_PARAMS + 32768 + k


def pack_table16():
    """Host-side gather table of the 16-point inference stream: per 16-bit element 2*canonical_index + is_lo, -1 = padding."""
    import numpy as np
    tab = np.empty(2 * P16F_WORDS, dtype=np.int32)
    _check(lib().nerf_debug_pack16_table(tab.ctypes.data_as(ctypes.c_void_p)), "nerf_debug_pack16_table")
    return tab


def _small_offset():
    # SM_BIAS of csrc/nerf_common.h = first word after the fp32 forward + backward weight streams
    return 593408 + 557056


def pack_params(flat, out=None, precision="fp32"):
    L = lib()
    precision = PACK_OF.get(precision, precision)
    if precision == "fp16_fp8c":    # the reduced inference stream (fp16 main + fp8 correction fragments) in the 16-point forward stream's slot
        if out is None:
            out = torch.empty(L.nerf_packed3_floats(), dtype=torch.float32, device=flat.device)
        _check(L.nerf_pack_params_split(_ptr(flat, "params"), _ptr(out, "packed"), 1, 2, _stream()), "nerf_pack_params_split")
        return out
    if precision == "fp16x3":       # fp16 (hi, lo) fragments: the 16-point forward stream and the transposed streams of the delta chain
        if out is None:
            out = torch.empty(L.nerf_packed3_floats(), dtype=torch.float32, device=flat.device)
        _check(L.nerf_pack_params_split(_ptr(flat, "params"), _ptr(out, "packed"), 1 | 4, 1, _stream()), "nerf_pack_params_split")
        return out
    if precision == "bf16x3":       # bf16 (hi, lo) fragments of the same two streams
        if out is None:
            out = torch.empty(L.nerf_packed3_floats(), dtype=torch.float32, device=flat.device)
        _check(L.nerf_pack_params_split(_ptr(flat, "params"), _ptr(out, "packed"), 1 | 4, 0, _stream()), "nerf_pack_params_split")
        return out
    if out is None:
        out = torch.empty(L.nerf_packed_floats(), dtype=torch.float32, device=flat.device)
    _check(L.nerf_pack_params(_ptr(flat, "params"), _ptr(out, "packed"), _stream()), "nerf_pack_params")
    return out


def pack_params_pair(flat_a, flat_b, precision):
    """the (hi, lo) fragment repack of two networks in the two launches one takes (nerf_pack_params_split_pair; fp16x3 / bf16x3)"""
    L = lib()
    split = {"bf16x3": 0, "fp16x3": 1}[PACK_OF.get(precision, precision)]
    out_a = torch.empty(L.nerf_packed3_floats(), dtype=torch.float32, device=flat_a.device)
    out_b = torch.empty(L.nerf_packed3_floats(), dtype=torch.float32, device=flat_a.device)
    _check(L.nerf_pack_params_split_pair(_ptr(flat_a, "params"), _ptr(out_a, "packed"), _ptr(flat_b, "params"), _ptr(out_b, "packed"), 1 | 4, split,
                                         _stream()), "nerf_pack_params_split_pair")
    return out_a, out_b


def embed(x, n_freqs):
    x = x.contiguous()
    out = torch.empty(x.shape[:-1] + (3 + 6 * n_freqs,), dtype=torch.float32, device=x.device)
    n = x.numel() // 3
    _check(lib().nerf_embed(_ptr(x, "x"), n, n_freqs, _ptr(out), _stream()), "nerf_embed")
    return out


def sample_coarse(rays, t_vals, lindisp, t_rand):
    n, stride = rays.shape
    S = t_vals.numel()
    z = torch.empty((n, S), dtype=torch.float32, device=rays.device)
    _check(lib().nerf_sample_coarse(_ptr(rays, "rays"), stride, n, _ptr(t_vals, "t_vals"), S, int(bool(lindisp)),
                                    _ptr(t_rand, "t_rand", True), _ptr(z), _stream()), "nerf_sample_coarse")
    return z


def make_rays(H, W, K, c2w, c2w_staticcam, ndc, near, far, device):
    """[H*W, 11] ray records of render(c2w=...) in one launch (get_rays + viewdirs + ndc_rays + near/far)."""
    import numpy as np

    def host(m, shape):
        if m is None:
            return None
        if isinstance(m, torch.Tensor):
            m = m.detach().cpu().numpy()
        return np.ascontiguousarray(np.asarray(m, dtype=np.float32)[:shape[0], :shape[1]])
    Kh, ph, sh = host(K, (3, 3)), host(c2w, (3, 4)), host(c2w_staticcam, (3, 4))
    rays = torch.empty((H * W, 11), dtype=torch.float32, device=device)
    as_p = lambda arr: None if arr is None else arr.ctypes.data_as(ctypes.c_void_p)
    _check(lib().nerf_make_rays(int(H), int(W), as_p(Kh), as_p(ph), as_p(sh), int(bool(ndc)), float(near), float(far),
                                _ptr(rays), 11, _stream()), "nerf_make_rays")
    return rays


def assemble_rays(rays_o, rays_d, ndc, H, W, focal, near, far):
    """[N, 11] ray records of render(rays=(rays_o, rays_d), ...) in one launch (view directions + ndc_rays + near / far)."""
    n = rays_o.numel() // 3
    rays = torch.empty((n, 11), dtype=torch.float32, device=rays_o.device)
    _check(lib().nerf_assemble_rays(_ptr(rays_o, "rays_o"), _ptr(rays_d, "rays_d"), n, int(bool(ndc)), int(H), int(W),
                                    float(focal), float(near), float(far), _ptr(rays), 11, _stream()), "nerf_assemble_rays")
    return rays


def sample_ray_batch(H, W, K, pose, image, n_rand, window, key, want_pixels=False):
    """nerf_sample_ray_batch: n_rand distinct pixels of `image` inside window = (h0, w0, nh, nw), their rays and colours in one launch.
    pose: device tensor holding c2w[:3,:4] (a view into a [N,4,4] pose table is fine); key: two 32-bit words."""
    import numpy as np
    Kh = np.ascontiguousarray(np.asarray(K.detach().cpu().numpy() if isinstance(K, torch.Tensor) else K, dtype=np.float32)[:3, :3])
    if not (isinstance(pose, torch.Tensor) and pose.is_cuda and pose.dtype == torch.float32 and pose.dim() == 2 and pose.shape[0] >= 3
            and pose.shape[1] >= 4 and pose.stride(1) == 1):
        raise NerfHipError("sample_ray_batch: pose must be a float32 [3+, 4] device tensor with contiguous rows")
    dev = image.device
    rays = torch.empty((2, n_rand, 3), dtype=torch.float32, device=dev)
    target = torch.empty((n_rand, 3), dtype=torch.float32, device=dev)
    pix = torch.empty((n_rand,), dtype=torch.int32, device=dev) if want_pixels else None
    h0, w0, nh, nw = (int(v) for v in window)
    _check(lib().nerf_sample_ray_batch(int(H), int(W), Kh.ctypes.data_as(ctypes.c_void_p), pose.data_ptr(), int(pose.stride(0)),
                                       _ptr(image, "image"), h0, w0, nh, nw, int(n_rand), int(key[0]) & 0xffffffff, int(key[1]) & 0xffffffff,
                                       _ptr(rays), _ptr(target), pix.data_ptr() if pix is not None else None, _stream()),
           "nerf_sample_ray_batch")
    return (rays, target, pix) if want_pixels else (rays, target)


def _dp(precision):
    """datapath index of the *_dp size entry points: 0 = fp32 rows, 1 = 16-bit tiles of the split datapaths, 2 = hi + lo tiles"""
    if precision not in PRECISIONS:
        raise ValueError(f"precision must be one of {PRECISIONS}")
    return 0 if precision == "fp32" else (2 if precision == "fp16x3w" else 1)


def act_floats(n_rays, n_samples, precision=None):
    """floats of a save buffer: for `precision`'s layout (fp32 rows: 10.6 KB / point; split datapaths' 16-bit tiles: 4.8 KB / point), or,
    without a precision, the larger of the two (a buffer any datapath may write)"""
    if precision is None:
        return lib().nerf_act_floats(int(n_rays), int(n_samples))
    return lib().nerf_act_floats_dp(int(n_rays), int(n_samples), _dp(precision))


def delta_floats(n_rays, n_samples, precision=None):
    """floats of the delta scratch of one backward pass (see act_floats)"""
    if precision is None:
        return lib().nerf_delta_floats(int(n_rays), int(n_samples))
    return lib().nerf_delta_floats_dp(int(n_rays), int(n_samples), _dp(precision))


class Workspace:
    """Caller-owned, persistent scratch of the backward path (the library allocates nothing and keeps no
    pointer).  The saved activations of a forward live from the forward to its backward; the deltas and the per-chunk
    partial gradients live inside one field_bwd call.  Buffers are LEASED from this pool (take) and handed back (give):
    a training loop of fixed shape re-uses the same device buffers every step -- same pointers, no allocation -- and a
    forward whose backward is still pending simply keeps its lease (a second forward leases another buffer).  A lease
    that is never returned (graph dropped without backward) is an ordinary tensor and is freed with its owner."""
    MAX_FREE = 16

    def __init__(self):
        self._free = {}

    def take(self, n_floats, device):
        free = self._free.setdefault(str(device), [])
        best = None
        for i, t in enumerate(free):
            # best fit; among equally sized buffers the lowest address, so that the choice does not depend on the order
            # in which earlier leases came back
            if t.numel() >= n_floats and (best is None or (t.numel(), t.data_ptr()) < (free[best].numel(), free[best].data_ptr())):
                best = i
        if best is not None and free[best].numel() <= 2 * n_floats + (1 << 20):
            return free.pop(best)
        return torch.empty(max(int(n_floats), 1), dtype=torch.float32, device=device)

    def give(self, t):
        if t is None:
            return
        free = self._free.setdefault(str(t.device), [])
        if any(f.data_ptr() == t.data_ptr() for f in free):
            return
        free.append(t)
        if len(free) > self.MAX_FREE:       # (list.remove would compare tensors elementwise)
            free.pop(min(range(len(free)), key=lambda i: free[i].numel()))

    def idle_bytes(self, device, at_least_floats=0):
        """bytes of the idle leases on `device` that a take() of at_least_floats can re-use instead of allocating (a lease smaller than
        the request serves nothing; take() also passes over leases more than twice the request)"""
        return 4 * sum(t.numel() for t in self._free.get(str(device), []) if t.numel() >= at_least_floats)

    def clear(self):
        self._free.clear()


WORKSPACE = Workspace()
SAVE_BUDGET_BYTES = int(float(os.environ.get("NERF_SAVE_BUDGET_GB", "48")) * (1 << 30))
# what ONE render_rays call may keep alive for its backward in total (a ray chunk above SAVE_BUDGET_BYTES is rendered in
# sub-chunks that each keep their own saved activations, as long as all of them fit here: 160 of the 288 GB of an MI355X)
SAVE_TOTAL_BYTES = int(float(os.environ.get("NERF_SAVE_TOTAL_GB", "160")) * (1 << 30))


def workspace_floats(n_rays, n_coarse, n_fine, training=True, precision=None):
    """nerf_workspace_floats[_dp](): floats of scratch one training render_rays call needs (saved activations of both
    passes + deltas + partial gradients of the larger pass) on `precision`'s layouts (None: the larger of the two); 0 for inference."""
    if precision is None:
        return lib().nerf_workspace_floats(int(n_rays), int(n_coarse), int(n_fine), int(bool(training)))
    return lib().nerf_workspace_floats_dp(int(n_rays), int(n_coarse), int(n_fine), int(bool(training)), _dp(precision))


def max_saved_rays(n_coarse, n_fine, precision=None):
    """Largest ray count whose backward scratch fits SAVE_BUDGET_BYTES (multiple of 1024, at least 1024): larger ray
    chunks are back-propagated in sub-chunks of this size (render._RenderRays).  64 + 128 samples under the default 48 GiB:
    10,240 rays on fp32 rows, 21,504 on the split datapaths' 16-bit tiles."""
    per_1024 = 4 * workspace_floats(1024, n_coarse, n_fine, True, precision)
    return max(1, SAVE_BUDGET_BYTES // max(per_1024, 1)) * 1024


def saved_bytes(n_rays, n_coarse, n_fine, precision=None):
    """bytes of saved activations (both passes) a training render_rays call over n_rays keeps until its backward"""
    return 4 * (act_floats(n_rays, n_coarse, precision) + (act_floats(n_rays, n_coarse + n_fine, precision) if n_fine > 0 else 0))


class NerfRenderCfg(ctypes.Structure):
    """include/nerf_hip.h NerfRenderCfg (render_rays in one call)"""
    _fields_ = [("n_coarse", ctypes.c_int), ("n_fine", ctypes.c_int), ("lindisp", ctypes.c_int), ("white_bkgd", ctypes.c_int),
                ("raw_noise_std", ctypes.c_float), ("precision", ctypes.c_int), ("reserved", ctypes.c_int)]


ACT_LAYOUTS = {0: "fp32 rows", 1: "tile32 fp32", 2: "tile32 bf16", 3: "tile16 fp32", 4: "tile16 bf16", 5: "tile16 fp16", 6: "tile16 fp16 hi + lo"}


def buffer_layout(buf):
    """What the library recorded for a scratch buffer it wrote (nerf_buffer_layout): (kind, is_delta, n_rays, n_samples),
    kind -1 = unknown.  act kinds: ACT_LAYOUTS; delta kinds: 0 fp32 rows, 1 / 2 tiles fp32 / bf16."""
    d, n, s_ = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    kind = lib().nerf_buffer_layout(buf.data_ptr(), ctypes.byref(d), ctypes.byref(n), ctypes.byref(s_))
    return kind, bool(d.value), n.value, s_.value


def _row16(f):
    """csrc/nerf_common.h row16(): row of feature f inside a 16-point tile (numpy / torch integer arrays or ints)."""
    return (f & ~15) + 8 * ((f >> 3) & 1) + 2 * (f & 3) + ((f >> 2) & 1)


def _row16h(f):
    """csrc/nerf_common.h row16h(): row of feature f inside a 16-point tile of bf16 rows."""
    return (f & ~15) + 8 * ((f >> 1) & 1) + 2 * ((f >> 2) & 3) + (f & 1)


_REGIONS = {**{"h%d" % i: (i, 256) for i in range(8)}, "feat": (8, 256), "hv": (9, 128)}


def buffer_regions(n_rays, n_samples, split, is_delta=False):
    """nerf_debug_layout: region offsets (floats) of a save / delta buffer for n_rays x n_samples points; split = the split
    datapaths' tiles of 16-bit elements (False: the fp32 datapath's point-major rows; 2: the two-word layout, whose "lo" entry is the
    offset of the mirror that holds the lo words)."""
    out = (ctypes.c_longlong * 16)()
    _check(lib().nerf_debug_layout(int(n_rays), int(n_samples), int(split), int(bool(is_delta)), out), "nerf_debug_layout")
    keys = [f"h{i}" for i in range(8)] + ["feat", "hv"] + (["graw", "scale"] if is_delta else ["enc", "dir", "dir_pt", "mask"])
    reg = {k: int(out[i]) for i, k in enumerate(keys)}
    reg["total"] = int(out[14])
    reg["lo"] = int(out[15]) if int(split) == 2 else 0
    return reg


def _family(precision):
    """nerf_debug_layout family of a datapath"""
    return 0 if precision not in SPLIT else (2 if precision == "fp16x3w" else 1)


def _elem16(precision):
    return torch.bfloat16 if precision == "bf16x3" else torch.float16


def _tile32(flat16, Pa, F, P):
    """32-point feature-major tiles of 16-bit elements -> point-major [P, F] fp32"""
    return flat16[:Pa * F].float().view(Pa // 32, F, 32).permute(0, 2, 1).reshape(Pa, F)[:P]


def saved_rows(buf, n_rays, n_samples, region, precision="fp32", part="hi"):
    """Debug / test view of one region of a save buffer as a point-major [P, F] fp32 tensor.  region: "h0".."h7", "hv", "enc", and
    on the fp32 datapath "feat" (the split datapaths fold feature_linear into the view branch and never write it).  fp32 datapath:
    point-major fp32 rows; split datapaths: 16-bit elements (fp16 / bf16 by `precision`), the 256- / 128-wide rows in 16-point tiles
    with the row16h row order, the encoding in 32-point feature-major tiles (csrc/nerf_common.h)."""
    split = precision in SPLIT
    P = n_rays * n_samples
    reg = buffer_regions(n_rays, n_samples, _family(precision))
    F = 64 if region == "enc" else _REGIONS[region][1]
    off = reg[region] + (reg["lo"] if part == "lo" else 0)      # part="lo" ("fp16x3w" only): the remainders T(v - hi)
    if part == "lo" and not reg["lo"]:
        raise NerfHipError("saved_rows(part='lo'): only the two-word layout (\"fp16x3w\") holds lo words")
    if not split:
        return buf[off:off + P * F].view(P, F)
    Pa = (P + 31) // 32 * 32
    flat = buf[off:off + (Pa * F + 1) // 2].view(_elem16(precision))
    if region == "enc":
        return _tile32(flat, Pa, F, P)
    rows = flat[:Pa * F].float().view(Pa // 16, F, 16).permute(0, 2, 1).reshape(Pa, F)[:P]      # [P, row]
    return rows[:, _row16h(torch.arange(F, device=rows.device))]                              # feature f sits at row16h(f)


def saved_dir(buf, n_rays, n_samples, precision="fp32"):
    """the per-ray direction encoding a saving forward wrote: [n_rays, 32] fp32 (27 used)"""
    off = buffer_regions(n_rays, n_samples, _family(precision))["dir"]
    return buf[off:off + n_rays * 32].view(n_rays, 32)


def saved_masks(buf, n_rays, n_samples, precision="fp32"):
    """the ReLU bitmask words of a save buffer as int32 [9, P, 8] (layers 0..7 + view branch; 256 bits per point and layer, in the
    lane order of the datapath's kernels: csrc/nerf_common.h)"""
    P = n_rays * n_samples
    off = buffer_regions(n_rays, n_samples, _family(precision))["mask"]
    return buf[off:off + 9 * P * 8].view(torch.int32).view(9, P, 8)


def relu_patterns(buf, n_rays, n_samples, precision="fp16x3"):
    """Debug / test view: the ReLU patterns a split datapath's forward saved (and its delta chain applies), as 9 boolean tensors
    [P, width] on the buffer's device -- trunk layers 0..7 (256 wide) and the view branch (128).  Word (layer, p, half), bit i <->
    feature 32*(i>>4) + d32row(i&15, half) (csrc/nerf_common.h).  A checker that forces THIS pattern onto an fp64 autograd of the
    reference network measures the backward's arithmetic alone (ReLU units within rounding of zero legitimately take either side)."""
    if precision not in SPLIT:
        raise NerfHipError("relu_patterns: the split datapaths' bitmask order")
    P = n_rays * n_samples
    words = saved_masks(buf, n_rays, n_samples, precision).view(9, P, 2, 4)
    dev = buf.device
    i = torch.arange(128, device=dev)
    shifts = torch.arange(32, device=dev)
    out = []
    for layer in range(9):
        width = 256 if layer < 8 else 128
        m = torch.zeros(P, width, dtype=torch.bool, device=dev)
        for half in range(2):
            feat = 32 * (i >> 4) + ((i & 15) & 3) + 8 * ((i & 15) >> 2) + 4 * half
            bits = ((words[layer, :, half, :, None] >> shifts) & 1).reshape(P, 128).bool()
            n = width // 2
            m[:, feat[:n]] = bits[:, :n]
        out.append(m)
    return out


def delta_rows(buf, n_rays, n_samples, region, precision="fp32", part="hi"):
    """Debug / test view of one region of a delta buffer as point-major [P, F] fp32: "h0".."h7", "hv", "feat" (fp32 datapath only),
    "graw" (split datapaths: the tiled 4-wide copy of the scaled d_raw)."""
    split = precision in SPLIT
    P = n_rays * n_samples
    reg = buffer_regions(n_rays, n_samples, _family(precision), is_delta=True)
    F = 4 if region == "graw" else _REGIONS[region][1]
    off = reg[region] + (reg["lo"] if part == "lo" else 0)
    if part == "lo" and not reg["lo"]:
        raise NerfHipError("delta_rows(part='lo'): only the two-word layout (\"fp16x3w\") holds lo words")
    if not split:
        return buf[off:off + P * F].view(P, F)
    Pa = (P + 31) // 32 * 32
    flat = buf[off:off + (Pa * F + 1) // 2].view(_elem16(precision))
    return _tile32(flat, Pa, F, P)


def delta_scale_word(buf, n_rays, n_samples):
    """fp16 split: the bit pattern of the launch's max|d_raw| the dgrad left in its delta buffer (int32 scalar tensor)"""
    off = buffer_regions(n_rays, n_samples, 1, is_delta=True)["scale"]       # (the same word in the one- and the two-word layout)
    return buf[off:off + 1].view(torch.int32)


# render_rays without gradients as ONE launch on the split datapaths (csrc/render_fused.hip; bit-identical to the
# chain of launches and as fast or faster for every ray count measured).  NERF_INFER_ONE_LAUNCH=0 keeps the chain:
# sample_coarse -> field forward -> composite -> sample_fine -> field forward -> composite
INFER_ONE_LAUNCH = os.environ.get("NERF_INFER_ONE_LAUNCH", "1") != "0"
# the reduced inference class ("fp16_fp8c") evaluates the COARSE pass of a coarse + fine rendering on the three-term fp16 products
# (render._field_pass); NERF_REDUCED_COARSE=reduced keeps round 4's all-reduced chain (with its two-pass last-sample guard)
REDUCED_COARSE_THREE_TERM = os.environ.get("NERF_REDUCED_COARSE", "fp16x3") != "reduced"


def render_cfg(n_coarse, n_fine, lindisp, white_bkgd, raw_noise_std, precision):
    if precision == "fp16_fp8c":
        raise NerfHipError("render_cfg: the one-call entry points run the fp32 / bf16x3 / fp16x3 datapaths; the reduced inference class "
                           "\"fp16_fp8c\" is a chain of launches with its last-sample guard in between (render._field_pass)")
    return NerfRenderCfg(int(n_coarse), int(n_fine), int(bool(lindisp)), int(bool(white_bkgd)), float(raw_noise_std),
                         {"fp32": 0, "bf16x3": 1, "fp16x3": 3, "fp16x3w": 5}[precision], 1)


def render_infer_supported(n_coarse, n_fine, precision):
    """whether nerf_render_rays_infer takes these sample counts on this datapath"""
    if precision in ("fp32", "fp16_fp8c"):      # (the reduced class runs the chain of launches: its last-sample guard sits between them)
        return False
    cfg = render_cfg(n_coarse, n_fine, 0, 0, 0.0, precision)
    return bool(lib().nerf_render_infer_supported(ctypes.byref(cfg)))


def render_rays_infer(packed_c, packed_f, rays, n_coarse, n_fine, lindisp, white_bkgd, raw_noise_std, precision, rnd):
    """nerf_render_rays_infer: the whole no-grad render_rays of `rays` in one launch.  rnd: the optional draws {t_rand, noise_c,
    u, noise_f}.  Returns the dict of render._field_pass's outputs (rgb_c, disp_c, acc_c, raw_c[, rgb_f, disp_f, acc_f, raw_f,
    z_std])."""
    L = lib()
    n, dev = rays.shape[0], rays.device
    fine = n_fine > 0
    S2 = n_coarse + n_fine
    cfg = render_cfg(n_coarse, n_fine, lindisp, white_bkgd, raw_noise_std, precision)
    e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    rgb, disp, acc, raw = e(n, 3), e(n), e(n), e(n, S2, 4)
    rgb0, disp0, acc0, z_std = (e(n, 3), e(n), e(n), e(n)) if fine else (None, None, None, None)
    ws = WORKSPACE.take(L.nerf_render_workspace_floats(ctypes.byref(cfg), n, 0), dev)
    opt = lambda k: _ptr(rnd[k], k) if rnd.get(k) is not None else None
    try:
        with _timed("render_infer_kernel", FLOP_FWD3_PER_POINT * n * (n_coarse + (S2 if fine else 0)), 0.0):
            _check(L.nerf_render_rays_infer(ctypes.byref(cfg), _ptr(packed_c, "packed3"), _ptr(packed_f, "packed3") if packed_f is not None else None,
                                            _ptr(rays, "rays"), rays.shape[1], n, opt("t_rand"), opt("noise_c"), opt("u"), opt("noise_f"),
                                            _ptr(rgb), _ptr(disp), _ptr(acc), _ptr(raw), _ptr(rgb0) if fine else None,
                                            _ptr(disp0) if fine else None, _ptr(acc0) if fine else None, _ptr(z_std) if fine else None,
                                            _ptr(ws), _stream()), "nerf_render_rays_infer")
    finally:        # stream-ordered: the next lease is written by kernels enqueued after this one
        WORKSPACE.give(ws)
    if not fine:
        return {"rgb_c": rgb, "disp_c": disp, "acc_c": acc, "raw_c": raw}
    return {"rgb_f": rgb, "disp_f": disp, "acc_f": acc, "raw_f": raw, "rgb_c": rgb0, "disp_c": disp0, "acc_c": acc0, "z_std": z_std}


class RangeMonitor:
    """The guard rail of the fp16 split's range (|activation| and |scaled delta| < 65520; beyond it `raw` / the gradient turn NaN): every
    `every`-th saving forward on fp16x3 / fp16x3w is followed by nerf_range_scan over what it saved, the delta chains of the same step
    by a scan of their deltas, the result words travel to pinned host memory without a synchronisation, and the NEXT calls poll the
    copies' events -- when a value has reached 32768 (half the range) a RuntimeWarning names the way out BEFORE the NaN:
    set_precision("bf16x3") (fp32's exponent range).  Replaces the reference's DEBUG-gated NaN / Inf check (run_nerf.py:414-416).
    every = 0 switches it off; report() is the on-demand form."""

    def __init__(self):
        self.every = int(os.environ.get("NERF_RANGE_CHECK_EVERY", "64"))
        self.calls = 0
        self.words = {}             # device -> int32[4]: (flag, max pattern) of the rows, (flag, max pattern) of the deltas
        self.inflight = []          # (pinned int32[4], event)
        self.max_seen = 0.0         # largest activation any finished scan has seen
        self.max_delta = 0.0        # largest |scaled delta| (the chain runs on s * d_raw with max|s d_raw| in [16, 32))
        self.warned_at = {"activation": 0.0, "scaled delta": 0.0}
        self.warnings = 0
        self.delta_scans_due = 0    # delta chains of the current step still to be scanned

    @staticmethod
    def _f16(bits):
        import numpy as np
        v = float(np.array([bits & 0xffff], dtype=np.uint16).view(np.float16)[0])
        return float("inf") if v != v else v       # a NaN pattern in the buffer: the overflow has already happened

    def _scan(self, bufs, n_rays):
        dev = bufs[0][0].device
        w = self.words.get(dev)
        if w is None:
            w = self.words[dev] = torch.zeros(4, dtype=torch.int32, device=dev)
        for buf, S in bufs:
            _check(lib().nerf_range_scan(buf.data_ptr(), int(n_rays), int(S), w.data_ptr(), _stream()), "nerf_range_scan")
        host = torch.empty(4, dtype=torch.int32, pin_memory=True)
        host.copy_(w, non_blocking=True)
        w.zero_()                   # (stream-ordered behind the copy: the next scan starts from zero)
        ev = torch.cuda.Event()
        ev.record()
        self.inflight.append((host, ev))

    def after_forward(self, acts, n_rays):
        """acts: [(act buffer, n_samples)] of one saving fp16 forward over n_rays rays"""
        self.poll()
        if self.every <= 0:
            return
        self.calls += 1
        if (self.calls - 1) % self.every:
            return
        self._scan(acts, n_rays)
        self.delta_scans_due = len(acts)        # ... and the delta chains of this step's backward

    def after_dgrad(self, delta, n_rays, n_samples):
        """called by field_bwd between the delta chain and the weight-gradient GEMM"""
        if self.delta_scans_due > 0:
            self.delta_scans_due -= 1
            self._scan([(delta, n_samples)], n_rays)

    def poll(self, wait=False):
        import warnings
        while self.inflight and (wait or self.inflight[0][1].query()):
            host, ev = self.inflight.pop(0)
            ev.synchronize()
            for what, flag, top in (("activation", int(host[0]), int(host[1])), ("scaled delta", int(host[2]), int(host[3]))):
                val = self._f16(top)
                if what == "activation":
                    self.max_seen = max(self.max_seen, val)
                else:
                    self.max_delta = max(self.max_delta, val)
                if flag and val > self.warned_at[what]:       # again only when it got worse
                    self.warned_at[what] = val
                    self.warnings += 1
                    where = ("`raw` (and the loss) turn" if what == "activation" else "the parameter gradients turn")
                    warnings.warn(f"nerf-pytorch_amd: {'an' if what == 'activation' else 'a'} {what} of the fp16x3 datapath reached {val:.4g}; the fp16 split's "
                                  f"operands end at 65504 -- beyond that {where} NaN.  Switch to nerf_pytorch_amd.set_precision(\"bf16x3\") (the same "
                                  "kernels with bf16 parts: fp32's exponent range) or to \"fp32\" before it does; weights and optimizer state carry over "
                                  "unchanged.", RuntimeWarning, stacklevel=3)

    def report(self):
        """wait for the scans in flight; {"max_activation", "max_scaled_delta", "limit", "warnings"}"""
        self.poll(wait=True)
        return {"max_activation": self.max_seen, "max_scaled_delta": self.max_delta, "warn_at": 32768.0, "limit": 65504.0,
                "warnings": self.warnings, "every": self.every}


RANGE_MONITOR = RangeMonitor()


def field_fwd(packed, rays, z_vals, save_act=False, precision="fp32", guard_packed=None, raw=None, next_guard=None):
    """guard_packed (precision "fp16_fp8c"): the fp16x3 repack of the same parameters for the last-sample guard, or the string
    "done" when an earlier call's next_guard has already written this pass's last samples into `raw`.
    next_guard = (fp16x3 repack of the refining pass's network, that pass's preallocated raw [n, S_next, 4]): the guard launch also
    evaluates the refining pass's last sample (its depth is this pass's last depth)."""
    n, stride = rays.shape
    S = z_vals.shape[1]
    if raw is None:
        raw = torch.empty((n, S, 4), dtype=torch.float32, device=rays.device)
    elif tuple(raw.shape) != (n, S, 4) or raw.dtype != torch.float32 or not raw.is_contiguous():
        raise NerfHipError(f"field_fwd: raw= must be a contiguous float32 [{n}, {S}, 4] tensor")
    act = WORKSPACE.take(act_floats(n, S, precision), rays.device) if save_act else None
    nbytes = BYTES_ACT_PER_POINT * n * S if save_act else 16.0 * n * S
    if precision in SPLIT:
        nbytes = BYTES_ACT3_PER_POINT * n * S if save_act else 16.0 * n * S
    if precision == "fp16_fp8c":
        if save_act:
            raise NerfHipError("field_fwd: \"fp16_fp8c\" is an inference class (no saved activations); train on \"fp16x3\"")
        if guard_packed is None:
            raise NerfHipError("field_fwd: \"fp16_fp8c\" needs guard_packed= (the fp16x3 repack) for the last-sample guard")
        with _timed("field_fwd16r_kernel<fp16 + fp8c>", FLOP_FWD3_PER_POINT * n * S, nbytes):
            _check(lib().nerf_field_fwd_split(_ptr(packed, "packed3"), _ptr(rays, "rays"), stride, _ptr(z_vals, "z_vals"),
                                              n, S, _ptr(raw), None, 3, _stream()), "nerf_field_fwd_split")
        if not (isinstance(guard_packed, str) and guard_packed == "done"):
            nxt, raw_n, S_n = (None, None, 0) if next_guard is None else (next_guard[0], next_guard[1], next_guard[1].shape[1])
            with _timed("field_fwd16r_kernel<fp16> (last samples)", FLOP_FWD3_PER_POINT * n * (1 if nxt is None else 2), 16.0 * n):
                _check(lib().nerf_field_fwd_last_sample(_ptr(guard_packed, "packed3"), _ptr(rays, "rays"), stride, _ptr(z_vals, "z_vals"),
                                                        n, S, _ptr(raw), _ptr(nxt, "packed3", True), _ptr(raw_n, "raw", True), S_n,
                                                        _stream()), "nerf_field_fwd_last_sample")
        return raw, act
    if precision in ("fp16x3", "fp16x3w"):
        two = precision == "fp16x3w" and save_act
        with _timed("field_fwd16r_kernel<fp16" + (", save hi+lo>" if two else ", save>" if save_act else ">"), FLOP_FWD3_PER_POINT * n * S,
                    (BYTES_ACT3_BF16_PER_POINT + (BYTES_ACT3_LO_PER_POINT if two else 0)) * n * S if save_act else nbytes):
            _check(lib().nerf_field_fwd_split(_ptr(packed, "packed3"), _ptr(rays, "rays"), stride, _ptr(z_vals, "z_vals"),
                                              n, S, _ptr(raw), _ptr(act, "act", True), 5 if two else 1, _stream()), "nerf_field_fwd_split")
        return raw, act
    if precision == "bf16x3":
        with _timed("field_fwd16r_kernel" + ("<save bf16>" if save_act else ""), FLOP_FWD3_PER_POINT * n * S,
                    BYTES_ACT3_BF16_PER_POINT * n * S if save_act else nbytes):
            _check(lib().nerf_field_fwd_split(_ptr(packed, "packed3"), _ptr(rays, "rays"), stride, _ptr(z_vals, "z_vals"),
                                              n, S, _ptr(raw), _ptr(act, "act", True), 0, _stream()), "nerf_field_fwd_split")
        return raw, act
    with _timed("field_fwd_kernel<save>" if save_act else "field_fwd_kernel", FLOP_FWD_PER_POINT * n * S, nbytes):
        _check(lib().nerf_field_fwd(_ptr(packed, "packed"), _ptr(rays, "rays"), stride, _ptr(z_vals, "z_vals"), n, S,
                                    _ptr(raw), _ptr(act, "act", True), _stream()), "nerf_field_fwd")
    return raw, act


def raw2outputs(raw, z_vals, rays_d, dir_stride, noise, raw_noise_std, white_bkgd, want_weights=True, want_depth=True,
                rays_d_offset=0):
    """rays_d: tensor whose element [rays_d_offset] is ray 0's first direction component."""
    n, S = z_vals.shape
    dev = raw.device
    rgb = torch.empty((n, 3), dtype=torch.float32, device=dev)
    disp = torch.empty((n,), dtype=torch.float32, device=dev)
    acc = torch.empty((n,), dtype=torch.float32, device=dev)
    weights = torch.empty((n, S), dtype=torch.float32, device=dev) if want_weights else None
    depth = torch.empty((n,), dtype=torch.float32, device=dev) if want_depth else None
    dptr = _ptr(rays_d, "rays_d") + 4 * rays_d_offset
    _check(lib().nerf_raw2outputs(_ptr(raw, "raw"), _ptr(z_vals, "z_vals"), dptr, dir_stride, n, S,
                                  _ptr(noise, "noise", True), float(raw_noise_std), int(bool(white_bkgd)),
                                  _ptr(rgb), _ptr(disp), _ptr(acc), _ptr(weights, "w", True), _ptr(depth, "d", True),
                                  _stream()), "nerf_raw2outputs")
    return rgb, disp, acc, weights, depth


def raw2outputs_bwd(raw, z_vals, rays_d, dir_stride, noise, raw_noise_std, white_bkgd, d_rgb, d_acc, d_disp,
                    rays_d_offset=0, d_weights=None, d_depth=None):
    n, S = z_vals.shape
    d_raw = torch.empty((n, S, 4), dtype=torch.float32, device=raw.device)
    dptr = _ptr(rays_d, "rays_d") + 4 * rays_d_offset
    _check(lib().nerf_raw2outputs_bwd(_ptr(raw, "raw"), _ptr(z_vals, "z_vals"), dptr, dir_stride, n, S,
                                      _ptr(noise, "noise", True), float(raw_noise_std), int(bool(white_bkgd)),
                                      _ptr(d_rgb, "d_rgb"), _ptr(d_acc, "d_acc", True), _ptr(d_disp, "d_disp", True),
                                      _ptr(d_weights, "d_weights", True), _ptr(d_depth, "d_depth", True),
                                      _ptr(d_raw), _stream()), "nerf_raw2outputs_bwd")
    return d_raw


def sample_fine(z_vals, weights, n_fine, u, u_lin, want_samples=False):
    n, Sc = z_vals.shape
    dev = z_vals.device
    z_all = torch.empty((n, Sc + n_fine), dtype=torch.float32, device=dev)
    z_std = torch.empty((n,), dtype=torch.float32, device=dev)
    z_samples = torch.empty((n, n_fine), dtype=torch.float32, device=dev) if want_samples else None
    _check(lib().nerf_sample_fine(_ptr(z_vals, "z_vals"), _ptr(weights, "weights"), n, Sc, n_fine,
                                  _ptr(u, "u", True), _ptr(u_lin, "u_lin", True), _ptr(z_all),
                                  _ptr(z_samples, "z_samples", True), _ptr(z_std), _stream()), "nerf_sample_fine")
    return z_all, z_std, z_samples


def sample_pdf(bins, weights, n_samples, u, u_lin):
    n, nb = bins.shape
    out = torch.empty((n, n_samples), dtype=torch.float32, device=bins.device)
    _check(lib().nerf_sample_pdf(_ptr(bins, "bins"), _ptr(weights, "weights"), n, nb, n_samples,
                                 _ptr(u, "u", True), _ptr(u_lin, "u_lin", True), _ptr(out), _stream()),
           "nerf_sample_pdf")
    return out


def field_bwd(packed, act, d_raw, grad, accumulate, precision="fp32", params=None):
    """Parameter gradients of one field evaluation into the flat vector `grad`.  `params`: the canonical (flat) parameter
    vector `packed` was made from -- required by the split datapaths, whose folded feature layer needs Wf, bf
    and Wv[:, :256] to turn G = delta_hv^T h7 into their gradients (csrc/nerf_common.h)."""
    if precision != "fp32" and params is None:
        raise NerfHipError("field_bwd: the split datapaths need params= (the flat parameter vector)")
    n, S, _ = d_raw.shape
    L = lib()
    dev = d_raw.device
    delta = WORKSPACE.take(delta_floats(n, S, precision), dev)
    partial = WORKSPACE.take(L.nerf_wgrad_partial_floats(n, S), dev)
    try:
        return _field_bwd(L, packed, act, d_raw, grad, accumulate, precision, delta, partial, n, S, params)
    finally:        # stream-ordered: the next lease is written by kernels enqueued after these
        WORKSPACE.give(delta)
        WORKSPACE.give(partial)


def _field_bwd(L, packed, act, d_raw, grad, accumulate, precision, delta, partial, n, S, params):
    split = {"bf16x3": 0, "fp16x3": 1, "fp16x3w": 5}.get(precision)
    two = split == 5
    # what the forward wrote into `act` (the library's own record, nerf_buffer_layout); the weight-gradient call below passes
    # datapath = -1 ("as recorded"), and a mismatched pairing is refused by the library (NERF_E_BADARG)
    kind = buffer_layout(act)[0]
    if split is not None and kind == -1:
        raise NerfHipError("field_bwd: `act` is not a save buffer this library's forward wrote (no layout record): the split "
                           "datapaths cannot guess its tiling and element type")
    P = n * S
    if split is not None:
        with _timed("field_dgrad3r_kernel<fp16, hi+lo>" if two else "field_dgrad3r_kernel<fp16>" if split else "field_dgrad3r_kernel<bf16 out>",
                    FLOP_DGRAD3_PER_POINT * P, BYTES_DELTA3_BF16_PER_POINT * P * (2 if two else 1)):
            _check(L.nerf_field_dgrad_split(_ptr(packed, "packed3"), _ptr(act, "act"), _ptr(d_raw, "d_raw"), n, S, _ptr(delta), split, _stream()),
                   "nerf_field_dgrad_split")
    else:
        with _timed("field_dgrad_kernel", FLOP_DGRAD_PER_POINT * P, BYTES_DELTA_PER_POINT * P):
            _check(L.nerf_field_dgrad(_ptr(packed, "packed"), _ptr(act, "act"), _ptr(d_raw, "d_raw"), n, S, _ptr(delta),
                                      _stream()), "nerf_field_dgrad")
    if split in (1, 5):
        RANGE_MONITOR.after_dgrad(delta, n, S)      # (scans the deltas on the steps whose forward was scanned; nothing otherwise)
    gemm16 = split is not None          # 16-bit operands streamed straight into the MFMA (wgrad1_kernel)
    datapath = -1
    args = (_ptr(act, "act"), _ptr(delta), _ptr(d_raw, "d_raw"), n, S, _ptr(partial), _ptr(grad, "grad"),
            int(bool(accumulate)), datapath)
    tail = (_ptr(params, "params", True), _stream())
    if TIMER is None:
        _check(L.nerf_field_wgrad_phase(*args, 7, *tail), "nerf_field_wgrad_phase")
        return grad
    if gemm16:      # all 12 jobs stream 16-bit operands straight into the MFMA
        with _timed("wgrad1_kernel<fp16, 3 terms>" if two else "wgrad1_kernel<fp16>" if split else "wgrad1_kernel", FLOP_WGRAD3_PER_POINT * P,
                    BYTES_WGRAD_MIXED_PER_POINT * P * (2 if two else 1)):
            _check(L.nerf_field_wgrad_phase(*args, 3, *tail), "nerf_field_wgrad_phase")
    else:
        with _timed("wgrad256_kernel", FLOP_WGRAD_BIG_PER_POINT * P, BYTES_WGRAD_BIG_PER_POINT * P):
            _check(L.nerf_field_wgrad_phase(*args, 1, *tail), "nerf_field_wgrad_phase")
        with _timed("wgrad_kernel(narrow jobs)", (FLOP_WGRAD_PER_POINT - FLOP_WGRAD_BIG_PER_POINT) * P, BYTES_WGRAD_SMALL_PER_POINT * P):
            _check(L.nerf_field_wgrad_phase(*args, 2, *tail), "nerf_field_wgrad_phase")
    # chunks of partial sums the reduction reads (csrc/field_bwd.hip, wgrad_chunks)
    n_chunks = min((19 if P < 400000 else 39) if gemm16 else 128, max(1, (P + 255) // 256))
    with _timed("wgrad_reduce_kernel", 0.0, 4.0 * N_PARAMS * (n_chunks + 1)):
        _check(L.nerf_field_wgrad_phase(*args, 4, *tail), "nerf_field_wgrad_phase")
    return grad


# Bumped by every raw-pointer update of parameters (the fused Adam kernel writes through data_ptr(), which does not
# advance the tensors' autograd version counters): NeRF.packed_params() keys its fragment-repack cache on it.
PARAM_EPOCH = 0             # total number of raw-pointer updates (any vector)
_PARAM_EPOCHS = {}          # ... per updated vector (keyed by the device address of its storage)


def _epoch_key(t):
    # the STORAGE a vector lives in, not the address of its first element: FlatAdam hands the kernel the run of parameters that
    # have gradients, which starts anywhere inside a network's flat vector (frozen first layer, heads-only fine-tuning)
    return t.untyped_storage().data_ptr()


def param_epoch(flat):
    """number of raw-pointer (fused Adam) updates of the storage THIS flat parameter vector lives in: an optimizer step on an
    unrelated network neither invalidates another network's fragment repack nor trips the stale-parameter guard of its pending
    backward; a step on any part of this network's vector does both"""
    return _PARAM_EPOCHS.get(_epoch_key(flat), 0)


def adam_step(params, grads, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step):
    """In-place fused Adam over flat fp32 vectors (one launch)."""
    global PARAM_EPOCH
    PARAM_EPOCH += 1
    _PARAM_EPOCHS[_epoch_key(params)] = _PARAM_EPOCHS.get(_epoch_key(params), 0) + 1
    n = params.numel()
    _check(lib().nerf_adam_step(_ptr(params, "params"), _ptr(grads, "grads"), _ptr(exp_avg, "exp_avg"),
                                _ptr(exp_avg_sq, "exp_avg_sq"), n, float(lr), float(beta1), float(beta2), float(eps),
                                int(step), _stream()), "nerf_adam_step")
