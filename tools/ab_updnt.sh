for v in 0 1 0 1; do GLIMS_UPD_NT=$v python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('upd_nt=$v', round(d['ms_per_step'],3), 'in-step spmv', round(r['avg_launch_us'],1), 'isolated', round(r['isolated_launch_us'],1))"; done
