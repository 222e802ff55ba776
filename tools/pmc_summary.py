#!/usr/bin/env python3
"""
Turns two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs as MI355X_MICROARCH.md section HBM prescribes)
into profiles/<name>.json: per kernel median counter values of the real launches, the gfx950 correction
(FETCH_SIZE counts 128-B requests at 64 B -> x2, calibrated below on k_cg_update / k_perm_in whose byte counts are
known exactly) and HBM-side bytes per launch = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024.

usage: tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write N_ROWS NNZ out.json
"""
import collections, csv, glob, json, re, sys
import numpy as np


def load(d, counter):
    f = glob.glob(d + '/*/*_counter_collection.csv')[0]
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != counter:
            continue
        m = re.search(r'(k_[a-z_0-9]+(<[^>]*>)?)', r['Kernel_Name'])
        out[m.group(1) if m else r['Kernel_Name'][:30]].append(float(r['Counter_Value']))
    res = {}
    for k, v in out.items():
        v = np.array(v)
        real = v[v > 0.2 * v.max()] if v.max() > 0 else v     # drop launches skipped by the device-side done flag
        res[k] = float(np.median(real))
    return res


def main():
    dfetch, dwrite, n_rows, nnz, outp = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    fe, wr = load(dfetch, 'FETCH_SIZE'), load(dwrite, 'WRITE_SIZE')
    vec_kib = n_rows * 8 / 1024.0
    calib = {
        "k_cg_update<1> reads 7 vectors": {"true_KiB": 7 * vec_kib, "FETCH_SIZE": fe.get('k_cg_update<1>'),
                                             "ratio": fe.get('k_cg_update<1>', 0) / (7 * vec_kib)},
        "k_cg_update<1> writes 5 vectors": {"true_KiB": 5 * vec_kib, "WRITE_SIZE": wr.get('k_cg_update<1>'),
                                              "ratio": wr.get('k_cg_update<1>', 0) / (5 * vec_kib)},
        "k_perm_in reads 8-B vector + 4-B index": {"true_KiB": 1.5 * vec_kib, "FETCH_SIZE": fe.get('k_perm_in'),
                                                     "ratio": fe.get('k_perm_in', 0) / (1.5 * vec_kib)},
    }
    kernels = {}
    for k in sorted(set(fe) | set(wr)):
        if not k.startswith('k_'):
            continue
        kernels[k] = {"FETCH_SIZE_KiB": fe.get(k), "WRITE_SIZE_KiB": wr.get(k),
                      "hbm_bytes_per_launch": 1024.0 * (2.0 * fe.get(k, 0.0) + wr.get(k, 0.0))}
    out = {"n_rows": n_rows, "nnz": nnz, "fetch_correction": 2.0, "unit": "KiB (counter) / bytes (derived)",
           "calibration": calib, "kernels": kernels,
           "algorithmic_spmv_bytes": 12 * nnz + 20 * n_rows}
    json.dump(out, open(outp, 'w'), indent=1)
    print(json.dumps({k: kernels[k] for k in kernels if 'spmv' in k or 'cheb' in k or 'rd_assemble' in k or 'cg_update' in k or 'fill_pattern' in k or 'corner_weights' in k or 'assemble_static' in k or 'row_lengths' in k}, indent=1))
    print(json.dumps(calib, indent=1))


if __name__ == "__main__":
    main()
