"""Build libnerf_hip.so (gfx950) in-tree with hipcc.

No JIT cache, no torch.utils.cpp_extension: the library is a plain C-ABI shared
object (include/nerf_hip.h) that ctypes loads, so it travels with the source
tree to the GPU box.  ``python -m nerf_pytorch_amd.build`` or
``__graft_entry__.build()`` runs this.
"""
import hashlib
import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libnerf_hip.so")
STAMP_PATH = os.path.join(PKG_DIR, "libnerf_hip.stamp")
SOURCES = ["api.hip", "render_abi.hip", "pack.hip", "ray_ops.hip", "field_fwd.hip", "field_bwd.hip", "field_fwd_ring.hip", "field_bwd_ring.hip", "render_fused.hip", "dense.hip"]
# test-only library of the superseded split-bf16 kernels (bit-identity references of the ring kernels): build_ref()
REF_LIB_PATH = os.path.join(PKG_DIR, "libnerf_hip_ref.so")
REF_SOURCES = [os.path.join("ref", "ref_api.hip"), os.path.join("ref", "field_fwd_bf16.hip"), os.path.join("ref", "field_bwd_bf16.hip")]
HEADERS = ["nerf_common.h", "field_device.h", "field_device_bf16.h", "split_types.h", "field_ring.h", "field_ring8.h", "field_fwd_ring_body.h", "ray_device.h", "api_util.h", "launchers.h", os.path.join("..", "..", "include", "nerf_hip.h")]
# -ffp-contract=off: the per-ray arithmetic is written in the reference's operation
# order (separate multiply / add) so z_vals, dists and sample points round identically.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC"]
LIBS = ["-ldl"]        # dense.hip resolves rocBLAS with dlopen on first use


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm with gfx950 support)")


def source_digest():
    h = hashlib.sha256()
    for name in SOURCES + HEADERS:
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode())
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def is_current():
    if not (os.path.exists(LIB_PATH) and os.path.exists(STAMP_PATH)):
        return False
    with open(STAMP_PATH) as f:
        return f.read().strip() == source_digest()


def build(force=False, verbose=False):
    """Compile every HIP translation unit into nerf-pytorch_amd/libnerf_hip.so."""
    if not force and is_current():
        return LIB_PATH
    # one hipcc process per translation unit, in parallel (the ring kernels dominate: ~35 s each), then one link
    from concurrent.futures import ThreadPoolExecutor
    objdir = os.path.join(PKG_DIR, "build")
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()

    def compile_one(src):
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n" + res.stdout + res.stderr)
        return obj
    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared"] + objs + ["-o", LIB_PATH] + LIBS
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + res.stdout + res.stderr)
    with open(STAMP_PATH, "w") as f:
        f.write(source_digest())
    return LIB_PATH


def build_ref(force=False):
    """Compile nerf-pytorch_amd/libnerf_hip_ref.so (tests only): the superseded kernels under csrc/ref/."""
    from concurrent.futures import ThreadPoolExecutor
    srcs = [os.path.join(CSRC, s_) for s_ in REF_SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(CSRC, "ref", "ref_launchers.h")]
    if not force and os.path.exists(REF_LIB_PATH) and os.path.getmtime(REF_LIB_PATH) >= max(os.path.getmtime(d) for d in deps):
        return REF_LIB_PATH
    objdir = os.path.join(PKG_DIR, "build")
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()

    def compile_one(src):
        obj = os.path.join(objdir, "ref_" + os.path.splitext(os.path.basename(src))[0] + ".o")
        res = subprocess.run([hipcc] + FLAGS + ["-I", CSRC, "-c", src, "-o", obj], capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n" + res.stdout + res.stderr)
        return obj
    with ThreadPoolExecutor(max_workers=len(srcs)) as pool:
        objs = list(pool.map(compile_one, srcs))
    res = subprocess.run([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared"] + objs + ["-o", REF_LIB_PATH], capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + res.stdout + res.stderr)
    return REF_LIB_PATH


if __name__ == "__main__":
    if "--ref" in sys.argv:
        print(build_ref(force="--force" in sys.argv))
        sys.exit(0)
    print(build(force="--force" in sys.argv, verbose=True))
