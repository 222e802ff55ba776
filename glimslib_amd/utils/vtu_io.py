"""
Minimal ASCII VTK-XML (.vtu) writer for P1 simplicial meshes -- stands in for ``fenics.File('x.pvd') << f``
(glimslib/simulation_helpers/helper_classes.py:1376-1380) without the vtk / DOLFIN dependency.
"""
import numpy as np


def _data_array(name, arr, ncomp=1, typ="Float64"):
    arr = np.asarray(arr)
    fmt = "%d" if typ.startswith("Int") or typ.startswith("UInt") else "%.17g"
    body = " ".join(fmt % v for v in arr.reshape(-1))
    return '<DataArray type="%s" Name="%s" NumberOfComponents="%d" format="ascii">%s</DataArray>\n' % (
        typ, name, ncomp, body)


def write_vtu(path, points, cells, point_fields=None, cell_fields=None):
    points = np.asarray(points, dtype=np.float64)
    cells = np.asarray(cells, dtype=np.int64)
    n, d = points.shape
    m, nv = cells.shape
    p3 = np.zeros((n, 3))
    p3[:, :d] = points
    vtk_type = 5 if nv == 3 else 10   # VTK_TRIANGLE / VTK_TETRA
    with open(path, "w") as f:
        f.write('<?xml version="1.0"?>\n<VTKFile type="UnstructuredGrid" version="0.1" byte_order="LittleEndian">\n')
        f.write('<UnstructuredGrid>\n<Piece NumberOfPoints="%d" NumberOfCells="%d">\n' % (n, m))
        f.write('<Points>\n' + _data_array("Points", p3, 3) + '</Points>\n')
        f.write('<Cells>\n')
        f.write(_data_array("connectivity", cells, 1, "Int64"))
        f.write(_data_array("offsets", (np.arange(m) + 1) * nv, 1, "Int64"))
        f.write(_data_array("types", np.full(m, vtk_type), 1, "UInt8"))
        f.write('</Cells>\n<PointData>\n')
        for name, v in (point_fields or {}).items():
            v = np.asarray(v, dtype=np.float64)
            if v.ndim == 2:
                v3 = np.zeros((n, 3))
                v3[:, :v.shape[1]] = v
                f.write(_data_array(str(name), v3, 3))
            else:
                f.write(_data_array(str(name), v, 1))
        f.write('</PointData>\n<CellData>\n')
        for name, v in (cell_fields or {}).items():
            f.write(_data_array(str(name), np.asarray(v), 1, "Int64" if np.issubdtype(np.asarray(v).dtype, np.integer) else "Float64"))
        f.write('</CellData>\n</Piece>\n</UnstructuredGrid>\n</VTKFile>\n')
    return path
