"""Device-side ray-batch sampling for train().

The reference builds a full [H, W, 3] ray grid per step, a meshgrid of pixel coordinates and draws
``np.random.choice(H*W, N_rand, replace=False)`` on the host, then copies one image host->device
(run_nerf.py:730-757).  At fused-kernel speed that host work (an O(H*W) permutation, ~10 ms at 800x800) would cap
rays/s, so here everything stays on the device and rays are generated ONLY for the selected pixels:

    batch_rays, target_s = sample_ray_batch(H, W, K, pose, image, N_rand, precrop_frac=...)

returns exactly what train() feeds to render(): ``batch_rays [2, N_rand, 3]`` (rays_o, rays_d as get_rays would
give them for those pixels, run_nerf_helpers.py:153-162) and ``target_s [N_rand, 3]``.  Selection is uniform without
replacement like the reference's.  On the GPU it is ONE launch of N_rand threads (nerf_sample_ray_batch: pixel k = a keyed
bijection of the window's pixel range applied to k -- distinct by construction, no O(H*W) permutation, no device sort), keyed by
two words drawn per step from the HOST generator (torch's default CPU generator, i.e. torch.manual_seed, or a CPU
``generator=``): no device synchronisation.  CPU tensors take the torch formulation (tests of the bookkeeping).
"""
import torch

from . import hip_backend as hb


def _window(H, W, precrop_frac):
    if precrop_frac is not None:
        dH, dW = int(H // 2 * precrop_frac), int(W // 2 * precrop_frac)
        return H // 2 - dH, W // 2 - dW, 2 * dH, 2 * dW
    return 0, 0, H, W


def sample_ray_batch(H, W, K, pose, image, N_rand, precrop_frac=None, generator=None, return_pixels=False):
    """pose [3,4] or [4,4] camera-to-world, image [H,W,3] (device tensors).  precrop_frac: central crop used during
    the first precrop_iters steps (run_nerf.py:738-747).  generator: a CPU torch.Generator (default: torch's global CPU generator)
    for the sync-free kernel path; a device generator works too (its draw is read back: one synchronisation)."""
    dev = image.device
    h0, w0, nh, nw = _window(H, W, precrop_frac)
    if N_rand > nh * nw:        # np.random.choice(..., replace=False) raises here too (run_nerf.py:752)
        raise ValueError(f"cannot take N_rand={N_rand} rays without replacement from {nh}x{nw} = {nh * nw} pixels")
    if image.is_cuda:
        gdev = generator.device if generator is not None else torch.device("cpu")
        key = torch.randint(0, 2 ** 31 - 1, (2,), generator=generator, device=gdev).tolist()
        c2w = pose if (isinstance(pose, torch.Tensor) and pose.is_cuda and pose.dtype == torch.float32 and pose.stride(-1) == 1) \
            else torch.as_tensor(pose, dtype=torch.float32).to(dev).contiguous()
        out = hb.sample_ray_batch(H, W, K, c2w, image if image.dtype == torch.float32 and image.is_contiguous() else image.float().contiguous(),
                                  N_rand, (h0, w0, nh, nw), key, want_pixels=return_pixels)
        return out
    sel = torch.randperm(nh * nw, device=dev, generator=generator)[:N_rand]
    jj = h0 + torch.div(sel, nw, rounding_mode="floor")      # row (y)
    ii = w0 + sel - torch.div(sel, nw, rounding_mode="floor") * nw      # column (x)
    i = ii.to(torch.float32)
    j = jj.to(torch.float32)
    dirs = torch.stack([(i - K[0][2]) / K[0][0], -(j - K[1][2]) / K[1][1], -torch.ones_like(i)], -1)
    c2w = pose[:3, :4].to(dev)
    rays_d = torch.sum(dirs[..., None, :] * c2w[:3, :3], -1)
    rays_o = c2w[:3, -1].expand(rays_d.shape)
    target_s = image[jj, ii]
    out = (torch.stack([rays_o, rays_d], 0), target_s)
    return out + ((jj * W + ii).to(torch.int32),) if return_pixels else out
