#!/bin/bash
# A/B builds of libslam_hip.so on the same box: tools/ab_*.so (built with SLAM_EXTRA_FLAGS=..., copied there) vs each other.
# usage: tools/gpu_ab.sh f64|f32 variant lib1.so lib2.so ...
dt=$1; var=$2; shift 2
for rep in 1 2; do
  for lib in "$@"; do
    echo -n "$(basename $lib): "; SLAM_HIP_LIB=$PWD/$lib python tools/gpu_variants.py $dt $var 2>&1 | tail -1
  done
done
