#!/usr/bin/env python3
"""
Generates the committed fixtures under tests/golden/.  Run in the build container only:

  (1) logistic_growth.json -- outputs of the reference's own ``compute_growth_logistic``
      (/root/reference/glimslib/simulation_helpers/math_reaction_diffusion.py:2-3), the only arithmetic of the
      reference's hot path that imports without FEniCS.  Inputs + expected outputs only.
  (2) oracle_c1.npz / oracle_box3d.npz -- final fields of the CPU oracle on BASELINE config C1 and on a small
      3-D two-tissue box (regression anchors for both the oracle and the HIP path; NOT FEniCS output).
"""
import importlib.util
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def make_logistic():
    path = "/root/reference/glimslib/simulation_helpers/math_reaction_diffusion.py"
    spec = importlib.util.spec_from_file_location("ref_mrd", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rows = []
    for rho in (0.0, 0.05, 0.1, 1.3):
        for cmax in (1.0, 2.5):
            for c in (0.0, 0.1, 0.25, 0.5, 0.9, 1.0, 1.7, -0.2):
                rows.append({"conc": c, "prolif_rate": rho, "conc_max": cmax,
                             "expected": float(mod.compute_growth_logistic(c, rho, cmax))})
    arr = np.array([0.0, 0.25, 0.5, 1.0])
    vec = {"conc": arr.tolist(), "prolif_rate": 0.1, "conc_max": 1.0,
           "expected": mod.compute_growth_logistic(arr, 0.1, 1.0).tolist()}
    with open(os.path.join(HERE, "logistic_growth.json"), "w") as f:
        json.dump({"source": "glimslib/simulation_helpers/math_reaction_diffusion.py:2-3 (imported, not copied)",
                   "scalar_cases": rows, "vector_case": vec}, f, indent=1)


def make_oracle_runs():
    from oracle.glims_oracle import OracleTumorGrowth
    from glimslib_amd import workloads
    from glimslib_amd.mesh import BoxMesh
    w = workloads.config_c1()
    dofs = (w.dirichlet_nodes[:, None] * 2 + np.arange(2)).ravel()
    o = OracleTumorGrowth(w.mesh.points, w.mesh.cells, w.per_cell('D'), w.per_cell('rho'), w.per_cell('gamma'),
                          w.per_cell('E'), w.per_cell('nu'), w.dt, dirichlet_u=(dofs, np.zeros(len(dofs))))
    u, c = o.run(w.c0, w.n_steps * w.dt)
    np.savez_compressed(os.path.join(HERE, "oracle_c1.npz"), c=c, u=u, n_steps=w.n_steps)

    mesh = BoxMesh((0.0, 0.0, 0.0), (8.0, 9.0, 7.0), 9, 8, 7)
    label = np.where(mesh.cell_midpoints()[:, 0] > 4.0, 2, 1).astype(np.int32)
    tabs = dict(D=[0.0, 0.1, 0.02], rho=[0.0, 0.1, 0.05], gamma=[0.0, 0.2, 0.1], E=[1.0, 1e-3, 3e-3],
                nu=[0.3, 0.40, 0.45])
    per = {k: np.asarray(v)[label] for k, v in tabs.items()}
    f = mesh.facets()
    bn = np.unique(f['vertices'][f['exterior']])
    dofs = (bn[:, None] * 3 + np.arange(3)).ravel()
    c0 = np.exp(-0.5 * ((mesh.points - np.array([4.0, 4.5, 3.5])) ** 2).sum(axis=1))
    o = OracleTumorGrowth(mesh.points, mesh.cells, per['D'], per['rho'], per['gamma'], per['E'], per['nu'], 1.0,
                          dirichlet_u=(dofs, np.zeros(len(dofs))))
    u, c = o.run(c0, 5.0)
    np.savez_compressed(os.path.join(HERE, "oracle_box3d.npz"), c=c, u=u, c0=c0, label=label, n_steps=5)


if __name__ == "__main__":
    make_logistic()
    make_oracle_runs()
    print("fixtures written to", HERE)
