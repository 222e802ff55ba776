"""
XDMF time-series writer / reader for P1 fields: ``<name>.xdmf`` (XML, temporal grid collection) + ``<name>.bin``
(raw little-endian heavy data referenced through ``Format="Binary" Seek=...`` items, which ParaView's Xdmf3 reader
opens directly).

Stands in for ``fenics.XDMFFile(...).write(mesh)`` / ``write_checkpoint(function, name, time)`` as used by
``Results.save_solution_start / save_function`` (glimslib/simulation_helpers/helper_classes.py:1350-1375,
1411-1440).  The reference's heavy data goes to ``solution.h5``; there is no HDF5 library in this environment
(no h5py, no DOLFIN), hence the raw binary side file -- same light-data structure, same field names.
"""
import os
import xml.etree.ElementTree as ET

import numpy as np

_TOPOLOGY = {3: "Triangle", 4: "Tetrahedron"}


class XDMFFile:
    def __init__(self, path):
        self.path = path if path.endswith(".xdmf") else path + ".xdmf"
        self.bin_path = os.path.splitext(self.path)[0] + ".bin"
        self._bin = None
        self._mesh_items = None
        self._steps = []          # [(time, [(name, ncomp, offset, n)])]
        self._n_points = self._n_cells = 0

    # -- writing ---------------------------------------------------------------------------------------------
    def _append(self, arr):
        if self._bin is None:
            os.makedirs(os.path.dirname(os.path.abspath(self.bin_path)), exist_ok=True)
            self._bin = open(self.bin_path, "wb")
        off = self._bin.tell()
        self._bin.write(np.ascontiguousarray(arr).tobytes())
        return off

    def write(self, mesh):
        pts = np.asarray(mesh.points, dtype="<f8")
        cells = np.asarray(mesh.cells, dtype="<i8")
        self._n_points, self._dim = pts.shape
        self._n_cells, self._nv = cells.shape
        self._mesh_items = (self._append(cells), self._append(pts))
        self._flush_xml()

    def write_checkpoint(self, function, function_name, time):
        """Appends the nodal values of a single-space Function under `function_name` at `time`; consecutive calls
        with the same time land in the same time step (one Attribute per subspace, as the reference writes them)."""
        if self._mesh_items is None:
            self.write(function.mesh)
        v = np.asarray(function.values(), dtype="<f8")
        ncomp = 1 if v.ndim == 1 else v.shape[1]
        if ncomp == 2:                                   # XDMF vectors have three components
            v = np.concatenate([v, np.zeros((v.shape[0], 1))], axis=1)
            ncomp = 3
        off = self._append(v)
        if not self._steps or self._steps[-1][0] != float(time) or any(a[0] == function_name for a in self._steps[-1][1]):
            self._steps.append((float(time), []))
        self._steps[-1][1].append((function_name, ncomp, off, v.shape[0]))
        self._flush_xml()

    def _flush_xml(self):
        if self._bin is not None:
            self._bin.flush()
        binname = os.path.basename(self.bin_path)
        root = ET.Element("Xdmf", Version="3.0")
        dom = ET.SubElement(root, "Domain")
        coll = ET.SubElement(dom, "Grid", Name="TimeSeries", GridType="Collection", CollectionType="Temporal")

        def item(parent, dims, number_type, precision, seek):
            d = ET.SubElement(parent, "DataItem", Format="Binary", NumberType=number_type, Precision=str(precision),
                              Endian="Little", Dimensions=dims, Seek=str(seek))
            d.text = binname

        steps = self._steps or [(0.0, [])]
        for k, (t, attrs) in enumerate(steps):
            g = ET.SubElement(coll, "Grid", Name="step_%05d" % k, GridType="Uniform")
            ET.SubElement(g, "Time", Value=repr(t))
            topo = ET.SubElement(g, "Topology", TopologyType=_TOPOLOGY[self._nv], NumberOfElements=str(self._n_cells))
            item(topo, "%d %d" % (self._n_cells, self._nv), "Int", 8, self._mesh_items[0])
            geo = ET.SubElement(g, "Geometry", GeometryType="XY" if self._dim == 2 else "XYZ")
            item(geo, "%d %d" % (self._n_points, self._dim), "Float", 8, self._mesh_items[1])
            for name, ncomp, off, n in attrs:
                a = ET.SubElement(g, "Attribute", Name=name, AttributeType="Scalar" if ncomp == 1 else "Vector",
                                  Center="Node")
                item(a, "%d" % n if ncomp == 1 else "%d %d" % (n, ncomp), "Float", 8, off)
        ET.ElementTree(root).write(self.path, xml_declaration=True, encoding="utf-8")

    def close(self):
        if self._bin is not None:
            self._bin.close()
            self._bin = None


def read_xdmf(path):
    """Reads a file written by XDMFFile: returns (points, cells, [(time, {name: array})])."""
    root = ET.parse(path).getroot()
    base = os.path.dirname(os.path.abspath(path))

    def load(d):
        dims = [int(x) for x in d.get("Dimensions").split()]
        dt = np.dtype("<i8" if d.get("NumberType") == "Int" else "<f8")
        with open(os.path.join(base, d.text.strip()), "rb") as f:
            f.seek(int(d.get("Seek")))
            return np.fromfile(f, dtype=dt, count=int(np.prod(dims))).reshape(dims)

    points = cells = None
    series = []
    for g in root.iter("Grid"):
        if g.get("GridType") != "Uniform":
            continue
        if cells is None:
            cells = load(g.find("Topology/DataItem"))
            points = load(g.find("Geometry/DataItem"))
        t = float(g.find("Time").get("Value"))
        fields = {}
        for a in g.findall("Attribute"):
            v = load(a.find("DataItem"))
            if v.ndim == 2 and points.shape[1] == 2:
                v = v[:, :2]
            fields[a.get("Name")] = v
        series.append((t, fields))
    return points, cells, series
