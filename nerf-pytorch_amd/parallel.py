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

# The host driver of the MI355X boxes supports dmabuf IPC only: without this, RCCL (and CUDA-tensor sharing across processes) fails with
# `hipIpcGetMemHandle: invalid argument`.  The HSA runtime reads it when the FIRST HIP call of the process initialises it, so it is
# set at import time -- init_distributed() would be too late (torch.cuda.is_available() has run by then); a value already in the
# environment wins.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# NERF_FORCE_PROCESS_GROUP=1 (or init_distributed(force_group=True)): build the process group and run EVERY collective of this module
# even when the world is one rank.  A 1-GPU box then executes the RCCL branch as written -- ProcessGroupNCCL, the communicator's
# stream, the async work objects GradientSync waits on, broadcast, all-gather -- instead of the world-size-1 short cuts.
FORCE_GROUP = os.environ.get("NERF_FORCE_PROCESS_GROUP") == "1"
_FORCE_GROUP_ENV = FORCE_GROUP      # what the environment asked for: shutdown() goes back to it


def _single(group=None):
    """True when the collectives of this module have nothing to do: no process group, or a world of one that was not forced"""
    if not (dist.is_available() and dist.is_initialized()):
        return True
    return dist.get_world_size(group) == 1 and not FORCE_GROUP


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def init_distributed(backend=None, force_group=None):
    """Initialise from torchrun's environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).
    Returns (rank, world_size, device).  A single process (no env) is world_size 1 and builds no process group unless
    force_group / NERF_FORCE_PROCESS_GROUP=1 asks for a one-rank group (FORCE_GROUP above; shutdown() undoes it).

    Knobs for a node this code has never run on (no multi-GPU node was available to any round; every one of them is an environment
    variable so that the driver's `bench.py --gpus 8` can be re-run with a different setting without a code change):
      HSA_ENABLE_IPC_MODE_LEGACY   defaults to 0 (dmabuf IPC: the only mode the build host's driver supports), set when this module
                                   is imported -- before the process's first HIP call; a value already in the environment wins
      NERF_DIST_NO_DEVICE_ID=1     do not bind the process group to the device at init (eager communicator creation): the communicator
                                   is then created lazily by the first collective
      NERF_DIST_TIMEOUT_S          rendezvous / collective timeout (default 180 s)
      NERF_ALLOW_SHARED_GPU=1      test rigs with fewer GPUs than ranks (with backend="gloo")"""
    global FORCE_GROUP
    if force_group is not None:
        FORCE_GROUP = bool(force_group)
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
    if (world > 1 or FORCE_GROUP) and not dist.is_initialized():
        import datetime
        kw = {}
        if backend is None:
            backend = "nccl" if use_cuda else "gloo"
        if backend == "nccl" and os.environ.get("NERF_DIST_NO_DEVICE_ID") != "1":
            kw["device_id"] = device
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
        else:
            # the forced one-rank group rendezvous on a private TCP store: nothing is written into os.environ (a later subprocess or
            # spawned worker must not inherit a stale MASTER_PORT)
            kw["init_method"] = f"tcp://127.0.0.1:{_free_port()}"
        timeout = datetime.timedelta(seconds=float(os.environ.get("NERF_DIST_TIMEOUT_S", "180")))
        try:
            dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=timeout, **kw)
            # first collective right here, so that a broken fabric / IPC setup fails at a known place with the environment
            # in the message instead of hanging inside the first training step
            probe = torch.ones(1, device=device if backend == "nccl" else "cpu")
            dist.all_reduce(probe)
            if int(probe.item()) != world:
                raise RuntimeError(f"probe all-reduce returned {probe.item()} on a world of {world}")
        except Exception as e:
            env = {k: v for k, v in os.environ.items()
                   if k.startswith(("NCCL_", "RCCL_", "HSA_", "HIP_", "ROCR_", "MASTER_", "GLOO_", "NERF_DIST_")) or k in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
            raise RuntimeError(
                f"nerf-pytorch_amd: process group init failed on rank {rank}/{world} (backend {backend}, device {device}, "
                f"{torch.cuda.device_count() if use_cuda else 0} GPU(s) visible, timeout {timeout.total_seconds():.0f} s): "
                f"{type(e).__name__}: {e}\nenvironment: {env}") from e
    return rank, world, device


def shutdown():
    """Destroy the process group (if any) and forget a force_group= request: the next init_distributed starts from the environment."""
    global FORCE_GROUP
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
    FORCE_GROUP = _FORCE_GROUP_ENV


def fabric_topology(max_chars=4000):
    """`rocm-smi --showtopo` (link type / hops / weight between the GPUs of this node: xGMI vs PCIe) as text, or None where the tool is
    missing -- bench.py puts it into the N > 1 line so that a scaling number can be read against the fabric it was measured on."""
    import shutil
    import subprocess
    if not shutil.which("rocm-smi"):
        return None
    try:
        out = subprocess.run(["rocm-smi", "--showtopo"], capture_output=True, text=True, timeout=30).stdout
    except Exception:       # noqa: BLE001 (diagnostics only)
        return None
    lines = [l.rstrip() for l in out.splitlines() if l.strip() and not set(l.strip()) <= set("=")]
    return "\n".join(lines)[:max_chars] or None


def ranks_seen(group=None):
    """Sorted list of the ranks that answer a collective (all-gather of every rank's id): world_size entries 0..G-1 when
    the communicator really spans every process."""
    if _single(group):
        return [0]
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    mine = torch.tensor([dist.get_rank(group)], dtype=torch.int64, device=dev)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size(group))]
    dist.all_gather(out, mine, group=group)
    return sorted(int(t.item()) for t in out)


def ranks_identical(tensors, group=None):
    """True when every rank holds bit-identical copies of `tensors` (all-gather of a 64-bit checksum of their bytes): the
    data-parallel invariant after broadcast + identical Adam steps on averaged gradients."""
    if _single(group):
        return True
    sums = []
    for t in tensors:
        b = t.detach().contiguous().view(torch.uint8).to(torch.int64)
        w = torch.arange(1, b.numel() + 1, dtype=torch.int64, device=b.device) % 65521
        sums.append((b * w).sum().reshape(1))
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    mine = torch.cat(sums).to(dev)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size(group))]
    dist.all_gather(out, mine, group=group)
    return all(torch.equal(o, out[0]) for o in out)


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


def global_randoms(n_global, n_coarse, n_fine, raw_noise_std, generator, device=None):
    """The random draws of ONE GLOBAL batch of n_global rays, in the reference's order and shapes (t_rand run_nerf.py:371, coarse noise
    :285, u helpers:208, fine noise :285), from a generator every rank seeds identically.  Each rank keeps its rows
    (shard_randoms) and hands them to render(..., randoms=...): the G-rank step then computes exactly the one-process step over the
    same n_global rays (BASELINE configs[3] = 8 shards of 4096 of ONE 32,768-ray batch)."""
    kw = dict(device=device, generator=generator)
    r = {"t_rand": torch.rand((n_global, n_coarse), **kw)}
    if raw_noise_std > 0:
        r["noise_c"] = torch.randn((n_global, n_coarse), **kw)
    if n_fine > 0:
        r["u"] = torch.rand((n_global, n_fine), **kw)
        if raw_noise_std > 0:
            r["noise_f"] = torch.randn((n_global, n_coarse + n_fine), **kw)
    return r


def shard_randoms(randoms, rank_=None, world_=None):
    """this rank's rows of global_randoms() (the same contiguous shard as shard_rays)"""
    rank_ = rank() if rank_ is None else rank_
    world_ = world_size() if world_ is None else world_
    out = {}
    for k, v in randoms.items():
        lo, hi = shard_slice(v.shape[0], rank_, world_)
        out[k] = v[lo:hi].contiguous()
    return out


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
    if _single(group):
        return
    world_ = dist.get_world_size(group)
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


_LIVE_SYNCS = []        # weak references to the GradientSync objects whose hook is installed


class GradientSync:
    """Gradient averaging overlapped with the backward pass.

    The backward of render_rays produces the coarse network's flat gradient first and then runs the (3x larger) fine
    network's backward; the two are independent.  While a GradientSync is installed, the all-reduce of a
    network's bucket is started (async, on the communicator's own stream — RCCL orders it after the kernels already
    enqueued) the moment render.py reports the bucket final, so the coarse bucket travels under the fine backward
    and only the last bucket's exchange is exposed.  finish() waits for the started exchanges, reduces whatever was
    not started (networks whose .grad already existed — accumulation reads the bucket — or that did not come through
    the hook) the plain way, and applies the 1/G.

    A network may report MORE than one bucket inside one loss.backward() (render(chunk < N_rand): one _RenderRays node
    per chunk; two render() calls before one backward; repeated query_points).  Autograd then SUMS the buckets out of
    place, i.e. it reads bucket #1 — whose in-place exchange was started early.  The second report therefore waits for
    that exchange, turns the bucket back into a per-rank share (x 1/G: every rank now holds sum/G, and the shares still
    add up to the sum), starts nothing more for this network, and finish() exchanges its final .grad the plain way.

        sync = GradientSync([model, model_fine])      # once (a context manager; a second one for the same networks
        ...                                           #       retires the first)
        loss.backward(); sync.finish(); optimizer.step()
    """

    def __init__(self, models, group=None):
        import weakref
        from .render import GRAD_READY_HOOKS
        self.models = [m for m in models if m is not None]
        self.group = group
        self.pending = {}           # id(model) -> (flat, work) of the exchange started under the backward
        self.multi = set()          # id(model) of networks that reported a second bucket in this backward
        self._hooks = GRAD_READY_HOOKS
        # one hook per network: a forgotten GradientSync (created per step / per run without close()) must not leave a
        # second all-reduce of the same bucket behind
        for ref in list(_LIVE_SYNCS):
            other = ref()
            if other is None or any(a is b for a in other.models for b in self.models):
                if other is not None:
                    other.close()
                if ref in _LIVE_SYNCS:
                    _LIVE_SYNCS.remove(ref)
        self._hooks.append(self._on_ready)
        self._ref = weakref.ref(self)
        _LIVE_SYNCS.append(self._ref)
        self.started = 0            # exchanges started under the backward (for tests / reporting)

    def close(self):
        if self._on_ready in self._hooks:
            self._hooks.remove(self._on_ready)
        if getattr(self, "_ref", None) in _LIVE_SYNCS:
            _LIVE_SYNCS.remove(self._ref)
        self.pending.clear()
        self.multi.clear()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def _on_ready(self, model, flat):
        if _single(self.group):
            return
        if not any(model is m for m in self.models) or id(model) in self.multi:
            return
        earlier = self.pending.pop(id(model), None)
        if earlier is not None:
            # a second bucket of this network (or a bucket left over from a backward whose finish() never ran): autograd is
            # about to read the first one.  Finish its exchange and make it a share again (class docstring).
            first, work = earlier
            work.wait()
            first.mul_(1.0 / dist.get_world_size(self.group))
            self.multi.add(id(model))
            return
        if any(p.grad is not None for p in model.parameters()):
            return      # autograd will ADD the bucket into the existing .grad: it must not change under that read
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.pending[id(model)] = (flat, work)
        self.started += 1

    def finish(self):
        if _single(self.group):
            self.pending.clear()
            self.multi.clear()
            return
        world_ = dist.get_world_size(self.group)
        try:
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
        finally:
            self.pending.clear()
            self.multi.clear()


def broadcast_parameters(models, src=0, group=None):
    """Make every rank start from rank `src`'s parameters (one broadcast per network)."""
    if _single(group):
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


def gather_frames(local_frames, frame_ids, n_frames, group=None, dst=None):
    """Collect per-rank rendered frames (numpy arrays, or tuples of them) in frame order: on every rank (dst=None, all_gather_object),
    or on rank `dst` only (gather_object: the other ranks send, receive nothing and get None)."""
    if _single(group):
        return local_frames
    world_ = dist.get_world_size(group)
    if dst is None:
        gathered = [None] * world_
        dist.all_gather_object(gathered, (frame_ids, local_frames), group=group)
    else:
        gathered = [None] * world_ if dist.get_rank(group) == dst else None
        dist.gather_object((frame_ids, local_frames), gathered, dst=dst, group=group)
        if gathered is None:
            return None
    out = [None] * n_frames
    for ids, frames in gathered:
        for i, f in zip(ids, frames):
            out[i] = f
    return out


def render_path(render_poses, hwf, K, chunk, render_kwargs, gt_imgs=None, savedir=None, render_factor=0, group=None):
    """Frame-parallel render_path (BASELINE configs[4]; the single-process function is run_nerf.py:137-175 = render.render_path):
    pose i is rendered by rank i mod G (frames_of_rank), no collective on the data path; every rank writes the PNGs of ITS frames into
    `savedir` under their global frame numbers (the reference's '{:03d}.png'), and rank 0 returns exactly what render_path returns --
    (rgbs [F, H, W, 3], disps [F, H, W]) float32 numpy in pose order -- after ONE gather of the finished frames (host arrays: off the
    rendering path).  The other ranks return (None, None).  One rank (or no process group): plain render_path."""
    import numpy as np
    from .render import render_path as render_path_single, _FrameSink, render
    if _single(group):
        return render_path_single(render_poses, hwf, K, chunk, render_kwargs, gt_imgs=gt_imgs, savedir=savedir, render_factor=render_factor)
    H, W, focal = hwf
    if render_factor != 0:                          # run_nerf.py:141-145
        H, W, focal = H // render_factor, W // render_factor, focal / render_factor
    n_frames = len(render_poses)
    mine = frames_of_rank(n_frames, dist.get_rank(group), dist.get_world_size(group))
    sink = _FrameSink(savedir)
    for i in mine:
        rgb, disp, _acc, _extras = render(H, W, K, chunk=chunk, c2w=render_poses[i][:3, :4], **render_kwargs)
        sink.push(i, rgb, disp)                     # PNG name = the GLOBAL frame number
    rgbs, disps = sink.close() if mine else (np.zeros((0, H, W, 3), np.float32), np.zeros((0, H, W), np.float32))
    frames = gather_frames([(rgbs[j], disps[j]) for j in range(len(mine))], mine, n_frames, group=group, dst=0)
    if frames is None:
        return None, None
    return np.stack([f[0] for f in frames], 0), np.stack([f[1] for f in frames], 0)
