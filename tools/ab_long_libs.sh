#!/bin/bash
# Long runs of several BUILDS of the library on one box: for every library file given (under _ab_prev/) and every workload, the
# summary line of tools/run_long.py (mean ms per step, Newton iterations and Krylov passes per step, take-backs, completed steps).
#   gpurun -- 'bash tools/ab_long_libs.sh "libglimship.so libglimship_o1.so" "c4:500 c4o:107:300" > gpurun_out/r05/ab_long.txt'
libs=${1:-"libglimship.so libglimship_new.so"}
runs=${2:-"c4:215:500 c4:107:300 c4o:107:300"}
export OPENBLAS_NUM_THREADS=1
mkdir -p gpurun_out/ab_long
for run in $runs; do
  wl=${run%:*}; steps=${run##*:}
  for lib in $libs; do
    cp _ab_prev/$lib glimslib_amd/libglimship.so
    python3 tools/run_long.py $wl $steps 20 gpurun_out/ab_long/s.json > gpurun_out/ab_long/log.txt 2>&1
    python3 - "$lib" "$wl" <<'P'
import json, sys
d = json.load(open('gpurun_out/ab_long/s.json'))
print("%-22s %-10s steps %3d/%3d status %d  mean %.3f ms/step (windows %.3f .. %.3f)  Newton %.2f  Krylov %.2f per step  take-backs %d" %
      (sys.argv[1], sys.argv[2], d['steps_completed'], d['steps_requested'], d['final_status'], d['ms_per_step_mean'],
       d['ms_per_step_min_window'] or 0, d['ms_per_step_max_window'] or 0, d['newton_its_per_step'], d['krylov_passes_per_step'],
       d['chebyshev_fallbacks']), flush=True)
P
  done
done
