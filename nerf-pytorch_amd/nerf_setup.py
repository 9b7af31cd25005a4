"""``config_parser`` and ``create_nerf`` with the reference's names and behaviour
(run_nerf.py:421-531, :178-259).

``configargparse`` is not installed in this environment, so the parser is a small
argparse subclass that reads the same ``key = value`` config files
(configs/*.txt of the reference) and the same 44 flags with the same defaults.
``create_nerf`` is the injection point of the build: it puts HIP-backed ``NeRF``
modules behind the very same ``render_kwargs`` dictionaries.
"""
import argparse
import os

import torch

from .field import NeRF, get_embedder
from .render import builtin_query_fn, run_network

# (flag, kwargs) in the reference's order; store_true flags take `key = True` in config files
_FLAGS = [
    ("expname", dict(type=str, help="experiment name")),
    ("basedir", dict(type=str, default="./logs/", help="where to store ckpts and logs")),
    ("datadir", dict(type=str, default="./data/llff/fern", help="input data directory")),
    ("netdepth", dict(type=int, default=8, help="layers in network")),
    ("netwidth", dict(type=int, default=256, help="channels per layer")),
    ("netdepth_fine", dict(type=int, default=8, help="layers in fine network")),
    ("netwidth_fine", dict(type=int, default=256, help="channels per layer in fine network")),
    ("N_rand", dict(type=int, default=32 * 32 * 4, help="batch size (number of random rays per gradient step)")),
    ("lrate", dict(type=float, default=5e-4, help="learning rate")),
    ("lrate_decay", dict(type=int, default=250, help="exponential learning rate decay (in 1000 steps)")),
    ("chunk", dict(type=int, default=1024 * 32, help="number of rays processed in parallel")),
    ("netchunk", dict(type=int, default=1024 * 64, help="number of pts sent through network in parallel")),
    ("no_batching", dict(action="store_true", help="only take random rays from 1 image at a time")),
    ("no_reload", dict(action="store_true", help="do not reload weights from saved ckpt")),
    ("ft_path", dict(type=str, default=None, help="specific weights npy file to reload for coarse network")),
    ("N_samples", dict(type=int, default=64, help="number of coarse samples per ray")),
    ("N_importance", dict(type=int, default=0, help="number of additional fine samples per ray")),
    ("perturb", dict(type=float, default=1., help="set to 0. for no jitter, 1. for jitter")),
    ("use_viewdirs", dict(action="store_true", help="use full 5D input instead of 3D")),
    ("i_embed", dict(type=int, default=0, help="set 0 for default positional encoding, -1 for none")),
    ("multires", dict(type=int, default=10, help="log2 of max freq for positional encoding (3D location)")),
    ("multires_views", dict(type=int, default=4, help="log2 of max freq for positional encoding (2D direction)")),
    ("raw_noise_std", dict(type=float, default=0., help="std dev of noise added to regularize sigma_a output")),
    ("render_only", dict(action="store_true", help="do not optimize, reload weights and render out render_poses path")),
    ("render_test", dict(action="store_true", help="render the test set instead of render_poses path")),
    ("render_factor", dict(type=int, default=0, help="downsampling factor to speed up rendering")),
    ("precrop_iters", dict(type=int, default=0, help="number of steps to train on central crops")),
    ("precrop_frac", dict(type=float, default=.5, help="fraction of img taken for central crops")),
    ("dataset_type", dict(type=str, default="llff", help="options: llff / blender / deepvoxels")),
    ("testskip", dict(type=int, default=8, help="will load 1/N images from test/val sets")),
    ("shape", dict(type=str, default="greek", help="options : armchair / cube / greek / vase")),
    ("white_bkgd", dict(action="store_true", help="render synthetic data on a white bkgd")),
    ("half_res", dict(action="store_true", help="load blender synthetic data at 400x400 instead of 800x800")),
    ("factor", dict(type=int, default=8, help="downsample factor for LLFF images")),
    ("no_ndc", dict(action="store_true", help="do not use normalized device coordinates")),
    ("lindisp", dict(action="store_true", help="sampling linearly in disparity rather than depth")),
    ("spherify", dict(action="store_true", help="set for spherical 360 scenes")),
    ("llffhold", dict(type=int, default=8, help="will take every 1/N images as LLFF test set")),
    ("i_print", dict(type=int, default=100, help="frequency of console printout and metric logging")),
    ("i_img", dict(type=int, default=500, help="frequency of tensorboard image logging")),
    ("i_weights", dict(type=int, default=10000, help="frequency of weight ckpt saving")),
    ("i_testset", dict(type=int, default=50000, help="frequency of testset saving")),
    ("i_video", dict(type=int, default=50000, help="frequency of render_poses video saving")),
]


class _ConfigFileParser(argparse.ArgumentParser):
    """argparse + `--config file` holding `key = value` lines (configargparse's default syntax).
    Command-line values override the file, as in configargparse."""

    def parse_known_args(self, args=None, namespace=None):
        import sys
        args = list(sys.argv[1:] if args is None else args)
        file_args = []
        for i, a in enumerate(args):
            path = None
            if a == "--config" and i + 1 < len(args):
                path = args[i + 1]
            elif a.startswith("--config="):
                path = a.split("=", 1)[1]
            if path:
                file_args += self._read(path)
        return super().parse_known_args(file_args + args, namespace)

    def _read(self, path):
        out = []
        store_true = {a.dest for a in self._actions if isinstance(a, argparse._StoreTrueAction)}
        with open(path) as f:
            for line in f:
                line = line.split("#", 1)[0].strip()
                if not line or line.startswith(";"):
                    continue
                key, _, val = line.partition("=")
                key, val = key.strip(), val.strip()
                if key in store_true:
                    if val.lower() in ("true", "1", "yes", ""):
                        out.append("--" + key)
                else:
                    out += ["--" + key, val]
        return out


def config_parser():
    """run_nerf.py:421-531."""
    parser = _ConfigFileParser()
    parser.add_argument("--config", type=str, default=None, help="config file path")
    for name, kw in _FLAGS:
        parser.add_argument("--" + name, **kw)
    return parser


def create_nerf(args, device=None, fused_adam=None):
    """run_nerf.py:178-259: (render_kwargs_train, render_kwargs_test, start, grad_vars, optimizer).
    The optimizer is nerf_pytorch_amd.FlatAdam -- torch.optim.Adam's arithmetic, param_groups and state_dict (the reference's
    checkpoints load into it and its checkpoints load into torch.optim.Adam), ONE HIP launch per network instead of foreach kernels
    over 48 tensors -- whenever the networks are the fused architecture on a GPU (fused_adam=None, the default); fused_adam=False
    returns torch.optim.Adam itself, fused_adam=True forces FlatAdam."""
    if device is None:
        device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    embed_fn, input_ch = get_embedder(args.multires, args.i_embed)
    input_ch_views = 0
    embeddirs_fn = None
    if args.use_viewdirs:
        embeddirs_fn, input_ch_views = get_embedder(args.multires_views, args.i_embed)
    output_ch = 5 if args.N_importance > 0 else 4
    skips = [4]
    arch = lambda D, W: dict(D=D, W=W, input_ch=input_ch, output_ch=output_ch, skips=skips, input_ch_views=input_ch_views,
                             use_viewdirs=args.use_viewdirs)
    # render_rays evaluates a (coarse, fine) pair on ONE path: if only one of the two is the fused kernels' architecture
    # (e.g. --netwidth_fine 128 next to the default coarse network), both are built layer by layer
    mixed_pair = args.N_importance > 0 and (NeRF.fused(**arch(args.netdepth, args.netwidth)) !=
                                            NeRF.fused(**arch(args.netdepth_fine, args.netwidth_fine)))
    model = NeRF(D=args.netdepth, W=args.netwidth, input_ch=input_ch, output_ch=output_ch, skips=skips,
                 input_ch_views=input_ch_views, use_viewdirs=args.use_viewdirs, force_dense=mixed_pair).to(device)
    grad_vars = list(model.parameters())
    model_fine = None
    if args.N_importance > 0:
        model_fine = NeRF(D=args.netdepth_fine, W=args.netwidth_fine, input_ch=input_ch, output_ch=output_ch,
                          skips=skips, input_ch_views=input_ch_views, use_viewdirs=args.use_viewdirs, force_dense=mixed_pair).to(device)
        grad_vars += list(model_fine.parameters())

    netchunk = getattr(args, "netchunk", 1024 * 64)
    # (marked as the stock query function: render_rays evaluates it inside the fused kernels; a user's own callable in this slot of
    # render_kwargs is called per pass like the reference does, run_nerf.py:385 / :401)
    network_query_fn = builtin_query_fn(lambda inputs, viewdirs, network_fn: run_network(
        inputs, viewdirs, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn, netchunk=netchunk))

    if fused_adam is None:      # the configuration bench.py measures: fused kernels + fused optimizer
        fused_adam = isinstance(model, NeRF) and (model_fine is None or isinstance(model_fine, NeRF)) and torch.device(device).type == "cuda"
    if fused_adam:
        from .optim import FlatAdam
        optimizer = FlatAdam(params=grad_vars, lr=args.lrate, betas=(0.9, 0.999))
    else:
        optimizer = torch.optim.Adam(params=grad_vars, lr=args.lrate, betas=(0.9, 0.999))

    start = 0
    basedir, expname = args.basedir, args.expname
    ft_path = getattr(args, "ft_path", None)
    if ft_path is not None and ft_path != 'None':
        ckpts = [ft_path]
    else:
        d = os.path.join(basedir, expname) if (basedir and expname) else None
        ckpts = [os.path.join(d, f) for f in sorted(os.listdir(d)) if 'tar' in f] if d and os.path.isdir(d) else []
    print('Found ckpts', ckpts)
    if len(ckpts) > 0 and not getattr(args, "no_reload", False):
        ckpt_path = ckpts[-1]
        print('Reloading from', ckpt_path)
        ckpt = torch.load(ckpt_path, map_location=device)
        start = ckpt['global_step']
        optimizer.load_state_dict(ckpt['optimizer_state_dict'])
        model.load_state_dict(ckpt['network_fn_state_dict'])
        if model_fine is not None:
            model_fine.load_state_dict(ckpt['network_fine_state_dict'])

    render_kwargs_train = {
        'network_query_fn': network_query_fn,
        'perturb': args.perturb,
        'N_importance': args.N_importance,
        'network_fine': model_fine,
        'N_samples': args.N_samples,
        'network_fn': model,
        'use_viewdirs': args.use_viewdirs,
        'white_bkgd': args.white_bkgd,
        'raw_noise_std': args.raw_noise_std,
    }
    if args.dataset_type != 'llff' or args.no_ndc:
        print('Not ndc!')
        render_kwargs_train['ndc'] = False
        render_kwargs_train['lindisp'] = args.lindisp
    render_kwargs_test = {k: render_kwargs_train[k] for k in render_kwargs_train}
    render_kwargs_test['perturb'] = False
    render_kwargs_test['raw_noise_std'] = 0.
    return render_kwargs_train, render_kwargs_test, start, grad_vars, optimizer
