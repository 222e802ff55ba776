#!/usr/bin/env python3
"""Wall time of glims_create / glims_setup for a workload:  GLIMS_VERBOSE=1 python tools/setup_time.py c4 [mechanics]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glimslib_amd import workloads
from glimslib_amd._backend import Handle
w = workloads.by_name(sys.argv[1])
mech = len(sys.argv) > 2
for rep in range(2):
    t = time.perf_counter()
    h = Handle(w.mesh.points, w.mesh.cells, w.cell_label)
    t1 = time.perf_counter()
    tb = w.tables
    h.set_materials(tb['D'], tb['rho'], tb['gamma'], tb['E'], tb['nu'])
    h.set_options(dt=w.dt)
    h.setup(mech)
    t2 = time.perf_counter()
    print("%s: glims_create %.3f s, glims_setup(mechanics=%d) %.3f s" % (w.name, t1 - t, mech, t2 - t1), flush=True)
    h.close()
