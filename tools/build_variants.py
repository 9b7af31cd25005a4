"""Build experimental variants of libnerf_hip.so (timing probes, not product code): patched copies of csrc/ compiled to
nerf-pytorch_amd/libexp_<name>.so (git-ignored, shipped by gpurun); tools/exp_fwd3.py <libname> times them."""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nerf-pytorch_amd", "csrc")
SOURCES = ["api.hip", "pack.hip", "ray_ops.hip", "field_fwd.hip", "field_bwd.hip", "field_fwd_bf16.hip", "field_bwd_bf16.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]

def patch(path, old, new, count=1):
    s = open(path).read()
    assert s.count(old) >= 1, (path, old[:60])
    open(path, "w").write(s.replace(old, new) if count == 0 else s.replace(old, new, count))

VARIANTS = {
    "base": lambda d: None,
    # weight-gradient staging with the round-1 lane map (2-way ds_write_b64 bank conflicts)
    "oldwg": lambda d: open(os.path.join(d, "field_bwd.hip"), "w").write(
        subprocess.run(["git", "-C", ROOT, "show", "18bf52c:nerf-pytorch_amd/csrc/field_bwd.hip"], capture_output=True, text=True, check=True).stdout),
    # forward: no ReLU bitmasks (upper bound of what cheaper mask construction can buy)
    "nomask": lambda d: patch(os.path.join(d, "field_fwd_bf16.hip"), "        if (with_mask) {\n            unsigned w[4]", "        if (false) {\n            unsigned w[4]"),
    # forward: no row stores (masks kept)
    "norows": lambda d: patch(os.path.join(d, "field_fwd_bf16.hip"), "        if (valid) {\n#pragma unroll\n            for (int nb = 0; nb < 16; ++nb)\n#pragma unroll\n                for (int r = 0; r < 4; ++r) store_val(region, W,", "        if (false) {\n#pragma unroll\n            for (int nb = 0; nb < 16; ++nb)\n#pragma unroll\n                for (int r = 0; r < 4; ++r) store_val(region, W,"),
    # dgrad: no delta stores
    "dgrad_nostore": lambda d: patch(os.path.join(d, "field_bwd_bf16.hip"), "    auto store_q = [&](auto part, size_t off) {\n        if (!valid) return;", "    auto store_q = [&](auto part, size_t off) {\n        return;"),
}

def main(names):
    inc = os.path.join(ROOT, "include")
    for name in names:
        d = os.path.join("/tmp/exp_src", name)
        shutil.rmtree(d, ignore_errors=True)
        shutil.copytree(CSRC, d)
        VARIANTS[name](d)
        # the sources include "../../include/nerf_hip.h": give the copy the same relative layout
        os.makedirs(os.path.join("/tmp/exp_src", "..", "include"), exist_ok=True)
        out = os.path.join(ROOT, "nerf-pytorch_amd", f"libexp_{name}.so")
        src = [os.path.join(d, s) for s in SOURCES]
        for s in src:
            patch_inc = open(s).read().replace('"../../include/nerf_hip.h"', f'"{inc}/nerf_hip.h"')
            open(s, "w").write(patch_inc)
        res = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + src + ["-o", out], capture_output=True, text=True)
        print(name, "->", out if res.returncode == 0 else res.stderr[-2000:])

if __name__ == "__main__":
    main(sys.argv[1:] or list(VARIANTS))
