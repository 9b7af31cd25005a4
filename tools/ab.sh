#!/bin/bash
# A/B kernel timing on one box: tools/ab.sh <libA.so> <libB.so> [time_kernels.py flags]; alternates the two libraries three times
A=$1; B=$2; shift 2
for i in 1 2 3; do
  NERF_HIP_LIB=$A python tools/time_kernels.py "$@" | tail -1
  NERF_HIP_LIB=$B python tools/time_kernels.py "$@" | tail -1
done
