for r in 1e-3 3e-3 3e-4 1e-4 1e-3 1e-2; do timeout -k 10 200 python bench.py --cg-rtol $r --steps 40 --warmup 3 --no-cpu-baseline 2>/dev/null > gpurun_out/m.json && python -c "
import json; d=json.load(open('gpurun_out/m.json')); print('cg_rtol=$r', round(d['ms_per_step'],3), d['config']['newton_its_per_step'], d['config']['cg_its_per_step'])"; done
