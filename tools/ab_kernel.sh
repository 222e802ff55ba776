#!/bin/bash
# usage: tools/ab_kernel.sh "<bench args>" <kernel-name-prefix> "VAR=a" "VAR=b" ...  -- profiled median of one kernel per setting
export TMPDIR=/tmp
args="$1"; kern="$2"; shift; shift
for cfg in "$@"; do
  rm -rf gpurun_out/abk
  env $cfg rocprofv3 --kernel-trace --output-format csv -d $PWD/gpurun_out/abk -- python3 bench.py $args --no-cpu-baseline > gpurun_out/abk.log 2>&1
  echo "$cfg: $(python tools/kernel_medians.py gpurun_out/abk 2>/dev/null | grep "^$kern" | head -1)"
done
rm -rf gpurun_out/abk
