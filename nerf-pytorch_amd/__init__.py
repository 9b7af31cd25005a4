"""nerf-pytorch_amd: MI355X (gfx950) implementation of the nerf-pytorch render hot path.

Drop-in surface (same names / signatures as the reference's run_nerf.py and
run_nerf_helpers.py): ``render``, ``render_path``, ``render_rays``, ``batchify_rays``,
``raw2outputs``, ``run_network``, ``sample_pdf``, ``Embedder``, ``get_embedder``, ``NeRF``,
``create_nerf``, ``config_parser``.  The directory name carries a hyphen (contract of
this build); import it as ``nerf_pytorch_amd`` (root-level shim module).
"""
from .field import Embedder, NeRF, get_embedder  # noqa: F401
from .render import (batchify, batchify_rays, get_rays, get_rays_np, img2mse, mse2psnr, ndc_rays,  # noqa: F401
                     query_points, raw2outputs, render, render_path, render_rays, run_network, sample_pdf, to8b,
                     set_precision, get_precision, check_range, DEFAULT_PRECISION)
from .nerf_setup import config_parser, create_nerf  # noqa: F401
from . import hip_backend, parallel  # noqa: F401
from .optim import FlatAdam  # noqa: F401
from .sampling import sample_ray_batch  # noqa: F401

__all__ = ["Embedder", "NeRF", "get_embedder", "batchify", "batchify_rays", "get_rays", "get_rays_np", "img2mse",
           "mse2psnr", "ndc_rays", "query_points", "raw2outputs", "render", "render_path", "render_rays",
           "run_network", "sample_pdf", "to8b", "config_parser", "create_nerf", "hip_backend", "parallel", "set_precision", "get_precision", "FlatAdam", "sample_ray_batch", "DEFAULT_PRECISION", "check_range"]
