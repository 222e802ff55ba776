#!/bin/bash
# Linear tolerance of a step's first solve (glims_options.cg_rtol) against iterations and ms per step, early and late in a run.
#   gpurun -- 'mkdir -p gpurun_out/r04 && bash tools/ab_cg_rtol.sh > gpurun_out/r04/ab_cg_rtol.txt 2>&1'
export GLIMS_MESH_CACHE=/tmp/glims_mesh_cache
python3 -c "import sys; sys.path.insert(0, '.'); from glimslib_amd import workloads; workloads.config_brain_like(1000000)"
for w in ${1:-c4 c3 bl}; do
  for win in "5 20" "120 40"; do
    set -- $win
    for rt in 1e-3 3e-4 1e-4 3e-5; do
      python3 bench.py --workload $w --warmup $1 --steps $2 --cg-rtol $rt --no-cpu-baseline --no-alt 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.readline())
c = d['config']
print('$w warmup $1 steps $2 cg_rtol $rt: %.3f ms/step, Newton %.2f, PCG %.2f, sweeps %.2f, cheap %.2f, status %d' % (d['ms_per_step'], c['newton_its_per_step'], c['cg_its_per_step'], c['assemblies_per_step'], c['quadratic_residual_updates_per_step'], d['solver_status']), flush=True)"
    done
  done
done
