#!/bin/bash
mkdir -p gpurun_out
L=gpurun_out/r2u.log; : > $L
for v in - libexp_f16_norows.so libexp_f16_nosave.so libexp_f16_nomask.so libexp_dgrad_nostore.so -; do python tools/exp_fwd3.py $v --bwd 2>&1 | grep -v amdgpu.ids >> $L; done
cat $L
