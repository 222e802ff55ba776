#!/bin/bash
# XCD chunk length (GLIMS_XCD_CHUNK) against in-step kernel times, one bench.py process per value, same box.
#   gpurun -- 'bash tools/ab_xcd_chunk.sh "bl c3 c4" "64 128 256 512"'
out=gpurun_out/r04
mkdir -p $out
export GLIMS_MESH_CACHE=/tmp/glims_mesh_cache
python3 -c "import sys; sys.path.insert(0, '.'); from glimslib_amd import workloads; workloads.config_brain_like(1000000)"
for w in ${1:-bl c3 c4}; do
  for g in ${2:-64 128 256 512} 64; do
    GLIMS_XCD_CHUNK=$g python3 bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-alt 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.readline())
k = {q['name'].split('<')[0]: q['median_us'] for q in d['roofline']['kernels']}
print('$w G=$g: %.3f ms/step (Newton %.2f, PCG %.2f)' % (d['ms_per_step'], d['config']['newton_its_per_step'], d['config']['cg_its_per_step']), {a: round(b, 1) for a, b in k.items()}, 'isolated spmv %.1f' % d['roofline']['isolated_launch_us'], flush=True)"
  done
done
