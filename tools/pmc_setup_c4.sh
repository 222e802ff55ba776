set -o pipefail
out=gpurun_out/r05; tag=r05_b; mkdir -p $out
export TMPDIR=/tmp OPENBLAS_NUM_THREADS=1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $out/pmc_c4_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_c4_$c -- python3 bench.py --workload c4 --steps 3 --warmup 1 --spmv-reps 10 --no-cpu-baseline --no-alt > $out/pmc_c4_$c.log 2>&1
done
python3 tools/pmc_summary.py $out/pmc_c4_FETCH_SIZE $out/pmc_c4_WRITE_SIZE 10077696 150048286 $out/${tag}_pmc_c4.json > $out/${tag}_pmc_c4_summary.txt
rm -rf $out/pmc_c4_FETCH_SIZE $out/pmc_c4_WRITE_SIZE $out/pmc_c4_*.log
rm -rf $out/prof_c4
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_c4 -- python3 bench.py --workload c4 --steps 3 --warmup 1 --spmv-reps 10 --no-cpu-baseline --no-alt > /dev/null 2>&1
python3 tools/kernel_medians.py $out/prof_c4 > $out/${tag}_c4_setup_medians.txt
rm -rf $out/prof_c4
grep "assemble_static\|fill_pattern\|corner_weights\|row_lengths" $out/${tag}_c4_setup_medians.txt
python3 -c "
import json
q=json.load(open('$out/${tag}_pmc_c4.json'))
for k,v in q['kernels'].items():
    if any(t in k for t in ('fill_pattern','corner_weights','assemble_static','row_lengths','egeo','translate')): print(k, round(v['hbm_bytes_per_launch']/1e9,2),'GB')
"
timeout -k 10 300 python -m pytest tests/test_gpu_symbolic.py tests/test_gpu_parity.py -q -m gpu -k "symbolic or operators or golden or unstructured_numbering" 2>&1 | tail -3
