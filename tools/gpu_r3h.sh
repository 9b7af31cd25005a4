#!/bin/bash
for v in - libexp_nodma.so; do python tools/exp_fwd3.py $v --bwd 2>&1 | grep -v amdgpu | grep "fwd16\|dgrad3\|fwd3\|\["; done
