#!/bin/bash
# Collects the round's rocprofv3 evidence on a GPU box into gpurun_out/r05/ (copied into profiles/ afterwards).
#   tools/collect_profiles.sh [tag] [part]   e.g.  gpurun -- 'bash tools/collect_profiles.sh r05_a bench'
#   part: bench | trace | pmc | all (default); the parts fit one gpurun call each
# Kernel traces and PMC passes are SEPARATE runs (MI355X_MICROARCH.md, HBM / rocprofv3 section); the program follows `--`
# directly (no env / bash -c hop under the profiler).
set -o pipefail
tag=${1:-r05_a}
out=gpurun_out/r05
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export OPENBLAS_NUM_THREADS=1   # (profiles/r05_blas_threads_throttle.txt)

trace() {   # name, program args ...
  local name=$1; shift
  rm -rf $out/prof_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$name -- "$@" > $out/${tag}_${name}_profiled.log 2>&1
  python3 tools/kernel_medians.py $out/prof_$name > $out/${tag}_${name}_real_launch_medians.txt
  cp $(ls $out/prof_$name/*/*kernel_stats.csv | head -1) $out/${tag}_${name}_kernel_stats.csv
  rm -rf $out/prof_$name
  echo "[collect] trace $name done"
}
pmc() {     # name, n_rows, nnz, program args ...
  local name=$1 rows=$2 nnz=$3; shift 3
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $out/pmc_${name}_$c
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_${name}_$c -- "$@" > $out/pmc_${name}_$c.log 2>&1
  done
  python3 tools/pmc_summary.py $out/pmc_${name}_FETCH_SIZE $out/pmc_${name}_WRITE_SIZE $rows $nnz $out/${tag}_pmc_$name.json > $out/${tag}_pmc_${name}_summary.txt
  rm -rf $out/pmc_${name}_FETCH_SIZE $out/pmc_${name}_WRITE_SIZE $out/pmc_${name}_*.log
  echo "[collect] pmc $name done"
}

part=${2:-all}
# the brain-like mesh is built by worker processes: once, un-profiled, into a cache the profiled runs read
export GLIMS_MESH_CACHE=/tmp/glims_mesh_cache
python3 -c "import sys; sys.path.insert(0, '.'); from glimslib_amd import workloads; workloads.config_brain_like(1000000)"
if [ $part = bench ] || [ $part = all ]; then
# bench lines (un-profiled)
python3 bench.py --steps 20 --warmup 5 > $out/${tag}_c4_bench.json 2> $out/${tag}_c4_bench.log; echo "[collect] bench c4 done (the driver's flags: --steps 20 --warmup 5)"
python3 bench.py --workload c3 --steps 40 --warmup 5 --no-cpu-baseline --no-alt > $out/${tag}_c3_bench.json 2>/dev/null
python3 bench.py --workload c4 --size 107 --steps 40 --warmup 5 --no-cpu-baseline --no-alt > $out/${tag}_c4_107_bench.json 2>/dev/null
python3 bench.py --workload c5 --steps 20 --warmup 10 --no-cpu-baseline --no-alt > $out/${tag}_c5_bench.json 2>/dev/null
python3 bench.py --workload c5 --size 215 --steps 10 --warmup 10 --no-cpu-baseline --no-alt > $out/${tag}_c5_10m_bench.json 2>/dev/null
python3 bench.py --workload bl --steps 20 --warmup 2 --no-cpu-baseline --no-alt > $out/${tag}_bl_bench.json 2>/dev/null
python3 bench.py --workload u --size 1000000 --steps 10 --warmup 2 --no-cpu-baseline --no-alt > $out/${tag}_u1m_bench.json 2>/dev/null
python3 tools/run_rd_precond.py 46 99 215 > $out/${tag}_rd_precond.json 2> $out/${tag}_rd_precond.txt
DIM=2 python3 tools/run_rd_precond.py 1000 > $out/${tag}_rd_precond_2d.json 2> $out/${tag}_rd_precond_2d.txt
echo "[collect] bench lines done"
fi
if [ $part = trace ] || [ $part = all ]; then

trace c4 python3 bench.py --workload c4 --steps 20 --warmup 5 --no-cpu-baseline --no-alt
trace c4_107 python3 bench.py --workload c4 --size 107 --steps 20 --warmup 5 --no-cpu-baseline --no-alt
trace bl python3 bench.py --workload bl --steps 10 --warmup 2 --no-cpu-baseline --no-alt
trace c5 python3 bench.py --workload c5 --steps 20 --warmup 10 --no-cpu-baseline --no-alt
ONLY=multigrid trace rdmg_c2 python3 tools/run_rd_precond.py 46
ONLY=multigrid trace rdmg_10m python3 tools/run_rd_precond.py 215
fi
if [ $part = pmc ] || [ $part = all ]; then

pmc c4 10077696 150048286 python3 bench.py --workload c4 --steps 3 --warmup 1 --spmv-reps 10 --no-cpu-baseline --no-alt
pmc c4_107 1259712 18663974 python3 bench.py --workload c4 --size 107 --steps 3 --warmup 1 --spmv-reps 10 --no-cpu-baseline --no-alt
pmc bl 1040364 16800788 python3 bench.py --workload bl --steps 3 --warmup 1 --spmv-reps 10 --no-cpu-baseline --no-alt
pmc c5 1000000 14761198 python3 bench.py --workload c5 --steps 3 --warmup 1 --spmv-reps 10 --no-cpu-baseline --no-alt
ONLY=multigrid STEPS=3 pmc rdmg_10m 10077696 150048286 python3 tools/run_rd_precond.py 215
fi
ls -la $out
