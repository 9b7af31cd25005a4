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
RESOURCE_PATH = os.path.join(PKG_DIR, "libnerf_hip.resources.json")
SOURCES = ["api.hip", "render_abi.hip", "pack.hip", "ray_ops.hip", "field_fwd.hip", "field_bwd.hip", "field_fwd_ring.hip", "field_bwd_ring.hip", "render_fused.hip", "dense.hip"]
HEADERS = ["nerf_common.h", "field_device.h", "split_types.h", "field_ring.h", "field_ring8.h", "field_fwd_ring_body.h", "ray_device.h", "api_util.h", "launchers.h", os.path.join("..", "..", "include", "nerf_hip.h")]
# -ffp-contract=off: the per-ray arithmetic is written in the reference's operation
# order (separate multiply / add) so z_vals, dists and sample points round identically.
# -Rpass-analysis=kernel-resource-usage: the register / scratch / LDS table of every kernel is recorded at every build
# (RESOURCE_PATH; tests/test_host_cpu.py fails the build if a heavy kernel spills).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Rpass-analysis=kernel-resource-usage"]
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


def _parse_resource_remarks(text):
    """{kernel (demangled where c++filt is available): {"sgprs", "vgprs", "agprs", "scratch_bytes_per_lane", "occupancy_waves_per_simd",
    "lds_bytes"}} from hipcc's -Rpass-analysis=kernel-resource-usage remarks"""
    import re
    out, cur = {}, None
    keys = {"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch_bytes_per_lane",
            "Occupancy [waves/SIMD]": "occupancy_waves_per_simd", "LDS Size [bytes/block]": "lds_bytes", "VGPRs Spill": "vgpr_spills",
            "SGPRs Spill": "sgpr_spills"}
    for line in text.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", line)
        if m and cur is not None and m.group(1).strip() in keys:
            cur[keys[m.group(1).strip()]] = int(m.group(2))
    filt = shutil.which("c++filt") or "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
    if os.path.exists(filt) and out:
        names = list(out)
        res = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True)
        if res.returncode == 0 and len(res.stdout.splitlines()) == len(names):
            out = {d.replace("void ", ""): out[n] for n, d in zip(names, res.stdout.splitlines())}
    return out


def resource_usage():
    """the table recorded by the last build (None if the library was built without it)"""
    import json
    if not os.path.exists(RESOURCE_PATH):
        return None
    with open(RESOURCE_PATH) as f:
        return json.load(f)


def is_current():
    if not (os.path.exists(LIB_PATH) and os.path.exists(STAMP_PATH) and os.path.exists(RESOURCE_PATH)):
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
        warnings = [l for l in res.stderr.splitlines() if "warning:" in l]
        if warnings:
            print(f"{src}: {len(warnings)} warning(s), first: {warnings[0]}", file=sys.stderr)
        return obj, _parse_resource_remarks(res.stderr)
    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as pool:
        compiled = list(pool.map(compile_one, SOURCES))
    objs = [c[0] for c in compiled]
    resources = {}
    for src, (_o, table) in zip(SOURCES, compiled):
        for k, v in table.items():
            resources[k] = dict(v, source=src)
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared"] + objs + ["-o", LIB_PATH] + LIBS
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + res.stdout + res.stderr)
    import json
    with open(RESOURCE_PATH, "w") as f:
        json.dump(resources, f, indent=1, sort_keys=True)
    with open(STAMP_PATH, "w") as f:
        f.write(source_digest())
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
