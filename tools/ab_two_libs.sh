#!/bin/bash
# Interleaved A/B of two builds of libglimship.so on one GPU box: _ab_prev/libglimship.so against _ab_prev/libglimship_new.so
# (both built here, git-ignored, travel with the snapshot).  Swaps the library file between runs of tools/ab_rank_sized.py.
#   gpurun -- 'bash tools/ab_two_libs.sh "c4:107 c4o:107" "cheb cached f64" > gpurun_out/r05/ab_libs.txt'
specs=${1:-"c4:107 c4o:107"}
only=${2:-"cheb cached f64"}
export OPENBLAS_NUM_THREADS=1 TIME_KERNELS=0 ROUNDS=1 WARMUP=${WARMUP:-10} STEPS=${STEPS:-40}
for rnd in 1 2 3; do
  for lib in libglimship.so libglimship_new.so; do
    cp _ab_prev/$lib glimslib_amd/libglimship.so
    echo "## round $rnd $lib"
    ONLY="$only" python3 tools/ab_rank_sized.py $specs 2>&1 | grep -E "^==|round 0"
  done
done
cp _ab_prev/libglimship_new.so glimslib_amd/libglimship.so
