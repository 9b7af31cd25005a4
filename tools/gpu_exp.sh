#!/bin/bash
# scratch driver of the experiment of the moment (gpurun): tools/gpu_exp.sh <script> [args] ; more scripts separated by --
mkdir -p gpurun_out
args=()
for a in "$@"; do
  if [ "$a" == "--" ]; then timeout 300 python "${args[@]}" 2>&1 | tail -60; args=(); else args+=("$a"); fi
done
[ ${#args[@]} -gt 0 ] && timeout 300 python "${args[@]}" 2>&1 | tail -60
