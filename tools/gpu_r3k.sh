#!/bin/bash
for v in - libexp_dma_m0.so -; do python tools/exp_fwd3.py $v --bwd 2>&1 | grep -v amdgpu | grep "fwd16\|dgrad3\|\["; done
