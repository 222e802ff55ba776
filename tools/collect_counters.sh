#!/bin/bash
# SQ / TCP / TCC counters of the hot kernels on the brain-like mesh (round-4 review, item 3): one rocprofv3 --pmc pass per
# counter group (the guide's HBM / rocprofv3 section: counters in their own runs, program directly after `--`).
#   gpurun -- 'bash tools/collect_counters.sh r05_a bl'
set -o pipefail
tag=${1:-r05_a}
spec=${2:-bl}
out=gpurun_out/r05
mkdir -p $out
export TMPDIR=/tmp OPENBLAS_NUM_THREADS=1
export GLIMS_MESH_CACHE=/tmp/glims_mesh_cache
python3 -c "import sys; sys.path.insert(0, '.'); from glimslib_amd import workloads; workloads.config_brain_like(1000000)"
rocprofv3 -L > $out/${tag}_counters_available.txt 2>&1 || true
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TA_BUSY_avr" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf $out/ctr_$i
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/ctr_$i -- python3 tools/pmc_bl_kernels.py $spec > $out/ctr_$i.log 2>&1 \
    || echo "[counters] group $i ($grp) failed: $(tail -2 $out/ctr_$i.log | head -1)"
  python3 tools/pmc_counters.py $out/ctr_$i k_spmv k_rd_assemble k_rd_quad k_cheb >> $out/${tag}_counters_${spec//:/_}.txt 2>/dev/null
  rm -rf $out/ctr_$i
  echo "[counters] group $i done"
done
rm -f $out/ctr_*.log
