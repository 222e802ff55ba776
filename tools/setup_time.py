import sys, time, os
sys.path.insert(0, '/root/repo')
from glimslib_amd import workloads
from glimslib_amd._backend import Handle
w = workloads.by_name(sys.argv[1])
t=time.perf_counter()
h = Handle(w.mesh.points, w.mesh.cells, w.cell_label)
print("Handle() %.2f s" % (time.perf_counter()-t))
