for wl in c4 c3; do for m in 0.1 0.3 0.5 0.1 0.5; do GLIMS_LIN_MARGIN=$m timeout -k 10 200 python bench.py --workload $wl --steps 40 --warmup 3 --no-cpu-baseline 2>/dev/null > gpurun_out/m.json && python -c "
import json; d=json.load(open('gpurun_out/m.json')); print('$wl margin=$m', round(d['ms_per_step'],3), d['config']['newton_its_per_step'], d['config']['cg_its_per_step'])"; done; done
