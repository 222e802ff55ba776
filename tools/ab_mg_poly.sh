#!/bin/bash
# Polynomial acceleration of the second grid (mg_poly_setup, opt-in) off / degree 4, on sliver and quality meshes.
#   gpurun -- 'mkdir -p gpurun_out/r04 && bash tools/ab_mg_poly.sh > gpurun_out/r04/ab_mg_poly.txt 2>&1'
export GLIMS_VERBOSE=1
for poly in ${POLYS:-0 4}; do
  export GLIMS_MG_POLY=$poly
  echo "=== GLIMS_MG_POLY=$poly : RD stiff steps, 200 k random-point Delaunay mesh"
  MESH=delaunay DSCALE=100 ONLY=multigrid STEPS=6 timeout -k 10 300 python3 tools/run_rd_precond.py 200000 2>&1 | grep -i "spectrum\|pcg_per" | cut -c1-420
  echo "=== GLIMS_MG_POLY=$poly : elasticity, ${NEL:-300000} random points"
  timeout -k 10 600 python3 tools/run_c5.py ${NEL:-300000} 4 2>&1 | grep -i "spectrum\|step\|levels"
done
unset GLIMS_MG_POLY
echo "=== default : elasticity, brain-like 300 k"
MESH=bl timeout -k 10 400 python3 tools/run_c5.py 300000 4 2>&1 | grep -i "spectrum\|step"
echo "=== default: C5 lattice"
timeout -k 10 300 python3 tools/run_c5.py 99 4 2>&1 | grep -i "spectrum\|step"
