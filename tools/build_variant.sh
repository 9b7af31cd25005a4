#!/bin/bash
# build a VARIANT of the library for A/B timing on one box (tools/ab.sh): tools/build_variant.sh <name> [extra hipcc flags, e.g. -DNERF_WG_CHUNKS_FINE=19]
# -> nerf-pytorch_amd/build/variants/libnerf_hip_<name>.so (same ABI; select with NERF_HIP_LIB=...)
set -e
NAME=$1; shift
PKG=$(cd "$(dirname "$0")/../nerf-pytorch_amd" && pwd)
OBJ=$PKG/build/variant_$NAME; mkdir -p $OBJ $PKG/build/variants
HIPCC=${HIPCC:-$(command -v hipcc || echo /opt/rocm/bin/hipcc)}
pids=()
for src in api render_abi pack ray_ops field_fwd field_bwd field_fwd_ring field_bwd_ring render_fused dense; do
  $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC "$@" -I $PKG/csrc -I $PKG/../include -c $PKG/csrc/$src.hip -o $OBJ/$src.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -fPIC -shared $OBJ/*.o -o $PKG/build/variants/libnerf_hip_$NAME.so -ldl
echo $PKG/build/variants/libnerf_hip_$NAME.so
