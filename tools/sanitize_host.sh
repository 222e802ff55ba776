#!/bin/bash
# AddressSanitizer + UBSan run of the HOST side of libglimship (symbolic phase: adjacency, neighbour lists, Morton /
# sigma sort, SELL-64 + corner packing, 16-bit column codes) on three small meshes.  Device code is not instrumented
# (GPU ASan is unavailable on this pool): -fno-gpu-sanitize.  Usage: tools/sanitize_host.sh   (from the repo root)
set -e
T=${TMPDIR:-/tmp}/glims_asan; mkdir -p $T
F="-O1 -g -std=c++17 -fopenmp --offload-arch=gfx950 -fsanitize=address,undefined -fno-gpu-sanitize -I include -I glimslib_amd/csrc"
hipcc $F -x hip -c glimslib_amd/csrc/setup_host.cpp -o $T/setup_host.o
hipcc $F -x hip -c tools/pattern_stats.cpp -o $T/pattern_stats.o
hipcc --offload-arch=gfx950 -fopenmp -fsanitize=address,undefined -fno-gpu-sanitize $T/pattern_stats.o $T/setup_host.o -L/opt/rocm/lib -lrccl -o $T/pattern_stats_asan
python - "$T" <<'PY'
import subprocess, sys, os
sys.path.insert(0, os.getcwd())
from glimslib_amd import workloads
from glimslib_amd.mesh import BoxMesh, RectangleMesh
T = sys.argv[1]
bad = 0
for name, mesh in (("box 20x18x16", BoxMesh((0, 0, 0), (1, 1, 1), 20, 18, 16)),
                   ("delaunay 20k", workloads.config_unstructured(20000).mesh),
                   ("rectangle 70x50", RectangleMesh((0, 0), (1, 1), 70, 50))):
    mesh.points.astype('<f8').tofile(T + '/p.bin'); mesh.cells.astype('<i4').tofile(T + '/c.bin')
    r = subprocess.run([T + '/pattern_stats_asan', str(mesh.points.shape[1]), str(mesh.num_vertices()),
                        str(mesh.num_cells()), T + '/p.bin', T + '/c.bin'], capture_output=True, text=True,
                       env={"OMP_NUM_THREADS": "4", "ASAN_OPTIONS": "detect_leaks=0", "PATH": "/usr/bin:/bin"})
    print("%-18s rc %d %s" % (name, r.returncode, "clean" if r.returncode == 0 and not r.stderr.strip() else r.stderr[-600:]))
    bad += r.returncode != 0 or bool(r.stderr.strip())
sys.exit(bad)
PY
