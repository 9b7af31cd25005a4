"""Data-parallel ray sharding (one process per GPU, RCCL over xGMI).

The reference is single-process (run_nerf.py:22); this is the only new control
flow around its loop.  Rays are independent through forward and backward, only
the parameter gradient couples them:

  * training: rank r renders rays [r*N/G, (r+1)*N/G) of each batch; after
    ``loss.backward()`` (run_nerf.py:775) ONE all-reduce per network sums the flat
    fp32 gradient vector (595,844 floats = 2.38 MB) that every ``.grad`` is a view
    of, scaled by 1/G; every rank then takes the identical Adam step.
  * ``render_only``: frames are dealt round-robin (frame i -> rank i mod G), no
    collective on the data path; rank 0 gathers the finished frames.

``torch.distributed`` backend "nccl" IS RCCL on ROCm; "gloo" is used by the CPU
tests of this logic.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Initialise from torchrun's environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).
    Returns (rank, world_size, device).  A single process (no env) is world_size 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = torch.cuda.is_available()
    if use_cuda and os.environ.get("NERF_ALLOW_SHARED_GPU") == "1":
        # test rigs with fewer GPUs than ranks (RCCL refuses two ranks on one device: combine with backend="gloo")
        local = local % torch.cuda.device_count()
    device = torch.device("cuda", local) if use_cuda else torch.device("cpu")
    if use_cuda:
        torch.cuda.set_device(device)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC only on this host driver
        kw = {}
        if backend is None:
            backend = "nccl" if use_cuda else "gloo"
        if backend == "nccl":
            kw["device_id"] = device
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, device


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def shard_slice(n_items, rank_, world_):
    """Contiguous, equal shard [lo, hi) of n_items (n_items must divide evenly: the loss is a mean)."""
    if n_items % world_ != 0:
        raise ValueError(f"batch of {n_items} rays does not divide over {world_} ranks")
    per = n_items // world_
    return rank_ * per, (rank_ + 1) * per


def shard_rays(batch_rays, target, rank_=None, world_=None):
    """batch_rays [2, N, 3] (rays_o, rays_d as train() builds them, run_nerf.py:756), target [N, 3]."""
    rank_ = rank() if rank_ is None else rank_
    world_ = world_size() if world_ is None else world_
    lo, hi = shard_slice(batch_rays.shape[1], rank_, world_)
    return batch_rays[:, lo:hi], target[lo:hi]


def frames_of_rank(n_frames, rank_=None, world_=None):
    rank_ = rank() if rank_ is None else rank_
    world_ = world_size() if world_ is None else world_
    return list(range(rank_, n_frames, world_))


def _flat_grad_of(model):
    """The flat gradient vector if every .grad of `model` is a view into model.last_flat_grad."""
    flat = getattr(model, "last_flat_grad", None)
    if flat is None:
        return None
    base, nbytes = flat.data_ptr(), flat.numel() * flat.element_size()
    for p in model.parameters():
        g = p.grad
        if g is None or not (base <= g.data_ptr() < base + nbytes):
            return None
    return flat


def allreduce_gradients(models, group=None):
    """Average gradients over ranks: one all-reduce per network on its flat gradient bucket
    (falls back to a packed copy when the .grad tensors are not views of one bucket)."""
    world_ = dist.get_world_size(group) if dist.is_initialized() else 1
    if world_ == 1:
        return
    for m in models:
        if m is None:
            continue
        flat = _flat_grad_of(m)
        if flat is not None:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
            flat.mul_(1.0 / world_)
            continue
        grads = [p.grad for p in m.parameters() if p.grad is not None]
        if not grads:
            continue
        bucket = torch.cat([g.reshape(-1) for g in grads])
        dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=group)
        bucket.mul_(1.0 / world_)
        off = 0
        for g in grads:
            g.copy_(bucket[off:off + g.numel()].view_as(g))
            off += g.numel()


class GradientSync:
    """Gradient averaging overlapped with the backward pass.

    The backward of render_rays produces the coarse network's flat gradient first and then runs the (3x larger) fine
    network's backward; the two are independent.  While a GradientSync is installed, the all-reduce of a
    network's bucket is started (async, on the communicator's own stream — RCCL orders it after the kernels already
    enqueued) the moment render.py reports the bucket final, so the coarse bucket travels under the fine backward
    and only the last bucket's exchange is exposed.  finish() waits for the started exchanges, reduces whatever was
    not started (networks whose .grad already existed — accumulation reads the bucket — or that did not come through
    the hook) the plain way, and applies the 1/G.

        sync = GradientSync([model, model_fine])      # once
        ...
        loss.backward(); sync.finish(); optimizer.step()
    """

    def __init__(self, models, group=None):
        from .render import GRAD_READY_HOOKS
        self.models = [m for m in models if m is not None]
        self.group = group
        self.pending = {}           # id(model) -> (flat, work)
        self._hooks = GRAD_READY_HOOKS
        self._hooks.append(self._on_ready)
        self.started = 0            # exchanges started under the backward (for tests / reporting)

    def close(self):
        if self._on_ready in self._hooks:
            self._hooks.remove(self._on_ready)

    def _on_ready(self, model, flat):
        if not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return
        if not any(model is m for m in self.models) or id(model) in self.pending:
            return
        if any(p.grad is not None for p in model.parameters()):
            return      # autograd will ADD the bucket into the existing .grad: it must not change under that read
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.pending[id(model)] = (flat, work)
        self.started += 1

    def finish(self):
        world_ = dist.get_world_size(self.group) if dist.is_initialized() else 1
        if world_ == 1:
            return
        rest = []
        for m in self.models:
            started = self.pending.pop(id(m), None)
            if started is None:
                rest.append(m)
                continue
            flat, work = started
            work.wait()
            flat.mul_(1.0 / world_)
            if _flat_grad_of(m) is not flat:    # autograd copied instead of adopting the views: hand the averages over
                from .render import _grad_views
                for p, g in zip(m.param_list(), _grad_views(m, flat)):
                    p.grad.copy_(g)
        allreduce_gradients(rest, group=self.group)


def broadcast_parameters(models, src=0, group=None):
    """Make every rank start from rank `src`'s parameters (one broadcast per network)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for m in models:
        if m is None:
            continue
        flat = m.flat_params() if hasattr(m, "flat_params") else None
        if flat is not None:
            dist.broadcast(flat, src=src, group=group)
            # c10d collectives write through the storage WITHOUT advancing tensor version counters, so the
            # fragment-repack cache (NeRF.packed_params) cannot see this update: drop it explicitly
            m.invalidate_packed()
        else:
            for p in m.parameters():
                dist.broadcast(p.data, src=src, group=group)


def gather_frames(local_frames, frame_ids, n_frames, group=None):
    """Collect per-rank rendered frames (numpy arrays) on rank 0 in frame order."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local_frames
    gathered = [None] * dist.get_world_size(group)
    dist.all_gather_object(gathered, (frame_ids, local_frames), group=group)
    out = [None] * n_frames
    for ids, frames in gathered:
        for i, f in zip(ids, frames):
            out[i] = f
    return out
