"""Device-side ray-batch sampling for train().

The reference builds a full [H, W, 3] ray grid per step, a meshgrid of pixel coordinates and draws
``np.random.choice(H*W, N_rand, replace=False)`` on the host, then copies one image host->device
(run_nerf.py:730-757).  At fused-kernel speed that host work (an O(H*W) permutation, ~10 ms at 800x800) would cap
rays/s, so here everything stays on the device and rays are generated ONLY for the selected pixels:

    batch_rays, target_s = sample_ray_batch(H, W, K, pose, image, N_rand, precrop_frac=...)

returns exactly what train() feeds to render(): ``batch_rays [2, N_rand, 3]`` (rays_o, rays_d as get_rays would
give them for those pixels, run_nerf_helpers.py:153-162) and ``target_s [N_rand, 3]``.  Selection is uniform without
replacement like the reference's, but drawn from the torch device generator (not numpy's stream).
"""
import torch


def sample_ray_batch(H, W, K, pose, image, N_rand, precrop_frac=None, generator=None):
    """pose [3,4] or [4,4] camera-to-world, image [H,W,3] (device tensors).  precrop_frac: central crop used during
    the first precrop_iters steps (run_nerf.py:738-747)."""
    dev = image.device
    if precrop_frac is not None:
        dH, dW = int(H // 2 * precrop_frac), int(W // 2 * precrop_frac)
        h0, w0, nh, nw = H // 2 - dH, W // 2 - dW, 2 * dH, 2 * dW
    else:
        h0, w0, nh, nw = 0, 0, H, W
    if N_rand > nh * nw:        # np.random.choice(..., replace=False) raises here too (run_nerf.py:752)
        raise ValueError(f"cannot take N_rand={N_rand} rays without replacement from {nh}x{nw} = {nh * nw} pixels")
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
    return torch.stack([rays_o, rays_d], 0), target_s
