"""
Mesh / label ingestion without vtk, meshio, SimpleITK or h5py -- the step before the forward path
(SURVEY.md section 8f row 3).  Counterparts in the reference's glimslib/utils/data_io.py:

  read_vtu + convert_vtu_to_mesh      <->  meshio.read + convert_meshio_to_fenics_mesh      (:469-524)
  identify/remove_orphaned_vertices   <->  identify_orphaned_vertices / remove_orphaned_vertices (:405-467)
  read_mha + image2fct2D + get_labelfunction_from_image  <->  sitk.ReadImage + (:31-63, :256-269)
  save_mesh_hdf5 / read_mesh_hdf5     <->  (:663-713)   (.npz container instead of DOLFIN HDF5)
  merge_VTUs                          <->  (:649-661)   (writes a .pvd collection of the per-step .vtu files)

The bundled atlas files of the reference are git-LFS pointer stubs, so these readers are tested on files written by
this package and on hand-built binary / compressed variants (tests/test_data_io.py).
"""
from __future__ import annotations

import base64
import os
import re
import xml.etree.ElementTree as ET
import zlib

import numpy as np

from ..mesh import Mesh, RectangleMesh
from .vtu_io import write_vtu  # noqa: F401  (re-exported)

_VTK_DTYPES = {'Int8': np.int8, 'UInt8': np.uint8, 'Int16': np.int16, 'UInt16': np.uint16, 'Int32': np.int32,
               'UInt32': np.uint32, 'Int64': np.int64, 'UInt64': np.uint64, 'Float32': np.float32,
               'Float64': np.float64}
VTK_TRIANGLE, VTK_TETRA = 5, 10


def _decode_binary(text, dtype, header_dtype, compressed):
    raw = text.strip().encode()
    hsize = np.dtype(header_dtype).itemsize
    if not compressed:
        # [nbytes header][data], both base64 -- either encoded together or header separately
        head_len = ((hsize + 2) // 3) * 4
        nbytes = int(np.frombuffer(base64.b64decode(raw[:head_len])[:hsize], dtype=header_dtype)[0])
        data = base64.b64decode(raw)
        if len(data) >= hsize + nbytes:
            payload = data[hsize:hsize + nbytes]
        else:
            payload = base64.b64decode(raw[head_len:])[:nbytes]
        return np.frombuffer(payload, dtype=dtype)
    # compressed: header = [nblocks, blocksize, lastblocksize, csize_0 ... csize_n-1] (base64), then the blocks (base64)
    first = base64.b64decode(raw[:((3 * hsize + 2) // 3) * 4])
    nblocks = int(np.frombuffer(first[:hsize], dtype=header_dtype)[0])
    head_bytes = (3 + nblocks) * hsize
    head_b64 = ((head_bytes + 2) // 3) * 4
    head = np.frombuffer(base64.b64decode(raw[:head_b64])[:head_bytes], dtype=header_dtype)
    sizes = head[3:3 + nblocks].astype(np.int64)
    blob = base64.b64decode(raw[head_b64:])
    out, off = [], 0
    for s in sizes:
        out.append(zlib.decompress(blob[off:off + int(s)]))
        off += int(s)
    return np.frombuffer(b"".join(out), dtype=dtype)


def _read_data_array(el, header_dtype, compressed, appended):
    dtype = _VTK_DTYPES[el.get('type')]
    fmt = el.get('format', 'ascii')
    ncomp = int(el.get('NumberOfComponents', '1'))
    if fmt == 'ascii':
        arr = np.array(el.text.split(), dtype=np.float64).astype(dtype) if el.text and el.text.strip() else np.zeros(0, dtype)
    elif fmt == 'binary':
        arr = _decode_binary(el.text, dtype, header_dtype, compressed)
    elif fmt == 'appended':
        if appended is None:
            raise ValueError("appended DataArray but no <AppendedData> section")
        enc, blob = appended
        off = int(el.get('offset'))
        if enc == 'base64':
            arr = _decode_binary(blob[off:].decode(), dtype, header_dtype, compressed)
        else:
            hsize = np.dtype(header_dtype).itemsize
            if compressed:
                nblocks = int(np.frombuffer(blob[off:off + hsize], dtype=header_dtype)[0])
                head = np.frombuffer(blob[off:off + (3 + nblocks) * hsize], dtype=header_dtype)
                p = off + (3 + nblocks) * hsize
                out = []
                for s in head[3:3 + nblocks]:
                    out.append(zlib.decompress(blob[p:p + int(s)]))
                    p += int(s)
                arr = np.frombuffer(b"".join(out), dtype=dtype)
            else:
                nbytes = int(np.frombuffer(blob[off:off + hsize], dtype=header_dtype)[0])
                arr = np.frombuffer(blob[off + hsize:off + hsize + nbytes], dtype=dtype)
    else:
        raise ValueError("unknown DataArray format %r" % fmt)
    return arr.reshape(-1, ncomp) if ncomp > 1 else arr


def read_vtu(path):
    """
    Reads a VTK XML UnstructuredGrid (.vtu): ascii, inline base64 (optionally zlib-compressed) and appended
    (raw or base64) DataArrays.  Returns dict(points [N,3], cells {'triangle': [M,3], 'tetrahedron': [M,4]},
    cell_index {type: indices into the file's cell order}, point_data, cell_data).
    """
    with open(path, 'rb') as f:
        content = f.read()
    appended = None
    m = re.search(rb'<AppendedData\s+encoding="(\w+)"\s*>\s*_', content)
    if m:
        end = content.rfind(b'</AppendedData>')
        appended = (m.group(1).decode(), content[m.end():end])
        content = content[:m.start()] + b'</VTKFile>'
    root = ET.fromstring(content)
    header_dtype = _VTK_DTYPES[root.get('header_type', 'UInt32')]
    compressed = root.get('compressor') is not None
    piece = root.find('./UnstructuredGrid/Piece')
    rd = lambda el: _read_data_array(el, header_dtype, compressed, appended)
    points = rd(piece.find('./Points/DataArray')).reshape(-1, 3).astype(np.float64)
    carr = {el.get('Name'): rd(el) for el in piece.find('./Cells')}
    conn, offs, types = carr['connectivity'].astype(np.int64), carr['offsets'].astype(np.int64), carr['types']
    starts = np.concatenate([[0], offs[:-1]])
    cells, cell_index = {}, {}
    for name, vt, nv in (('triangle', VTK_TRIANGLE, 3), ('tetrahedron', VTK_TETRA, 4)):
        idx = np.flatnonzero(types == vt)
        if len(idx):
            cells[name] = conn[starts[idx][:, None] + np.arange(nv)[None, :]]
            cell_index[name] = idx
    pdata = {el.get('Name'): rd(el) for el in (piece.find('./PointData') if piece.find('./PointData') is not None else [])}
    cdata = {el.get('Name'): rd(el) for el in (piece.find('./CellData') if piece.find('./CellData') is not None else [])}
    return dict(points=points, cells=cells, cell_index=cell_index, point_data=pdata, cell_data=cdata)


def identify_orphaned_vertices(points, cells):
    """Vertices not referenced by any cell (data_io.py:405-427); these make the operator singular (PETSc error 76
    in the reference, GLIMS_E_USAGE 'orphaned vertex' here)."""
    used = np.zeros(len(points), dtype=bool)
    used[np.asarray(cells).ravel()] = True
    return np.flatnonzero(~used)


def remove_orphaned_vertices(points, cells, vertex_ids=None):
    """data_io.py:429-467 -- drops the given (default: all) orphaned vertices and renumbers the connectivity."""
    points, cells = np.asarray(points), np.asarray(cells)
    if vertex_ids is None:
        vertex_ids = identify_orphaned_vertices(points, cells)
    keep = np.ones(len(points), dtype=bool)
    keep[np.asarray(vertex_ids, dtype=np.int64)] = False
    if not keep[cells.ravel()].all():
        raise ValueError("a vertex to be removed is referenced by a cell")
    new_id = np.cumsum(keep) - 1
    return points[keep], new_id[cells].astype(np.int32)


def convert_vtu_to_mesh(path_or_dict, domain_array_name='ElementBlockIds'):
    """
    data_io.py:469-524: first cell type of the file, 2-D meshes lose an all-zero third coordinate, orphaned vertices
    are removed, the named cell array becomes the subdomain labels.  Returns (Mesh, subdomains int array or None).
    """
    d = read_vtu(path_or_dict) if isinstance(path_or_dict, str) else path_or_dict
    if 'tetrahedron' in d['cells']:
        ctype, dim = 'tetrahedron', 3
    elif 'triangle' in d['cells']:
        ctype, dim = 'triangle', 2
    else:
        raise ValueError("no triangle / tetrahedron cells in the file")
    cells, points = d['cells'][ctype], d['points']
    if dim == 2:
        if not np.all(points[:, 2] == 0):
            raise ValueError("2-D mesh expected: third coordinate of all points must be 0")
        points = points[:, :2]
    points, cells = remove_orphaned_vertices(points, cells)
    sub = None
    if domain_array_name in d['cell_data']:
        sub = np.asarray(d['cell_data'][domain_array_name])[d['cell_index'][ctype]].astype(np.int64)
    return Mesh(points, cells), sub


# ---- images ------------------------------------------------------------------------------------------------------------
_MET = {'MET_UCHAR': np.uint8, 'MET_CHAR': np.int8, 'MET_USHORT': np.uint16, 'MET_SHORT': np.int16,
        'MET_UINT': np.uint32, 'MET_INT': np.int32, 'MET_FLOAT': np.float32, 'MET_DOUBLE': np.float64}


def read_mha(path):
    """MetaImage (.mha, ElementDataFile = LOCAL; raw or zlib).  Returns dict(array [z, y, x] (or [y, x]), origin,
    spacing) -- array axis order as sitk.GetArrayFromImage."""
    with open(path, 'rb') as f:
        blob = f.read()
    hdr, pos = {}, 0
    while True:
        end = blob.index(b'\n', pos)
        line = blob[pos:end].decode('ascii', 'replace').strip()
        pos = end + 1
        if '=' in line:
            k, v = [t.strip() for t in line.split('=', 1)]
            hdr[k] = v
            if k == 'ElementDataFile':
                break
    if hdr['ElementDataFile'] != 'LOCAL':
        raise NotImplementedError("only ElementDataFile = LOCAL is supported")
    dims = [int(t) for t in hdr['DimSize'].split()]
    dtype = np.dtype(_MET[hdr['ElementType']])
    if hdr.get('BinaryDataByteOrderMSB', hdr.get('ElementByteOrderMSB', 'False')).lower() == 'true':
        dtype = dtype.newbyteorder('>')
    data = blob[pos:]
    if hdr.get('CompressedData', 'False').lower() == 'true':
        data = zlib.decompress(data)
    nch = int(hdr.get('ElementNumberOfChannels', '1'))
    arr = np.frombuffer(data, dtype=dtype, count=int(np.prod(dims)) * nch)
    shape = tuple(reversed(dims)) + ((nch,) if nch > 1 else ())
    origin = [float(t) for t in hdr.get('Offset', hdr.get('Origin', ' '.join(['0'] * len(dims)))).split()]
    spacing = [float(t) for t in hdr.get('ElementSpacing', ' '.join(['1'] * len(dims))).split()]
    return dict(array=arr.reshape(shape).astype(dtype.newbyteorder('=')), origin=origin, spacing=spacing)


def write_mha(path, array, origin=None, spacing=None, compressed=False, channels=False):
    array = np.ascontiguousarray(array)
    met = {v: k for k, v in _MET.items()}[array.dtype.type]
    dims = list(reversed(array.shape[:-1] if channels else array.shape))
    origin = origin or [0.0] * len(dims)
    spacing = spacing or [1.0] * len(dims)
    data = array.tobytes()
    hdr = ["ObjectType = Image", "NDims = %d" % len(dims), "BinaryData = True", "BinaryDataByteOrderMSB = False",
           "CompressedData = %s" % ("True" if compressed else "False"),
           "Offset = %s" % " ".join(repr(float(v)) for v in origin),
           "ElementSpacing = %s" % " ".join(repr(float(v)) for v in spacing),
           "DimSize = %s" % " ".join(str(v) for v in dims)] + \
          (["ElementNumberOfChannels = %d" % array.shape[-1]] if channels else []) + \
          ["ElementType = %s" % met, "ElementDataFile = LOCAL"]
    with open(path, 'wb') as f:
        f.write(("\n".join(hdr) + "\n").encode())
        f.write(zlib.compress(data) if compressed else data)


def image2fct2D(image2d, origin=(0.0, 0.0), spacing=(1.0, 1.0)):
    """data_io.py:31-63: rectangle mesh with one vertex per pixel ((width-1) x (height-1) cells) and the pixel values
    as nodal values.  Returns (Mesh, nodal values)."""
    image2d = np.asarray(image2d)
    height, width = image2d.shape[:2]
    mesh = RectangleMesh((origin[0], origin[1]),
                         (origin[0] + spacing[0] * width, origin[1] + spacing[1] * height), width - 1, height - 1)
    return mesh, image2d.reshape(height * width, *image2d.shape[2:]).astype(np.float64)


def get_labelfunction_from_image(path_to_file, z_slice=0, data_name='label'):
    """data_io.py:256-269: z-slice of a 3-D label image as (Mesh, nodal label values)."""
    img = read_mha(path_to_file)
    a = img['array']
    sl = a[z_slice] if a.ndim >= 3 else a
    return image2fct2D(sl, img['origin'][:2], img['spacing'][:2])


def sample_image_at_points(image, points, order='nearest'):
    """Nearest-voxel lookup of an image dict (read_mha) at physical points -- how a label map defined on an image
    grid becomes a nodal label function on an arbitrary mesh."""
    a = image['array']
    d = len(image['spacing'])
    ijk = np.rint((np.asarray(points)[:, :d] - np.asarray(image['origin'])) / np.asarray(image['spacing'])).astype(np.int64)
    for k in range(d):
        ijk[:, k] = np.clip(ijk[:, k], 0, a.shape[d - 1 - k] - 1)
    return a[tuple(ijk[:, k] for k in reversed(range(d)))]


# ---- functions <-> images (data_io.py:101-227, 363-411) ------------------------------------------------------------------
class Image:
    """What the reference passes around as a SimpleITK image, reduced to the parts its converters touch: a voxel array
    in sitk.GetArrayFromImage order ([z,] y, x [, component]), origin, spacing.  Files: MetaImage (.mha)."""

    def __init__(self, array, origin=None, spacing=None, is_vector=False):
        self.array = np.asarray(array)
        self.is_vector = bool(is_vector)
        self.dim = self.array.ndim - (1 if self.is_vector else 0)
        self.origin = tuple(float(v) for v in (origin if origin is not None else [0.0] * self.dim))
        self.spacing = tuple(float(v) for v in (spacing if spacing is not None else [1.0] * self.dim))

    # SimpleITK-flavoured accessors used by the reference's helpers
    def GetOrigin(self):
        return self.origin

    def GetSpacing(self):
        return self.spacing

    def GetDimension(self):
        return self.dim

    def GetSize(self):
        return tuple(reversed(self.array.shape[:self.dim]))

    def GetNumberOfComponentsPerPixel(self):
        return self.array.shape[-1] if self.is_vector else 1

    def write(self, path, compressed=False):
        write_mha(path, self.array, list(self.origin), list(self.spacing), compressed, channels=self.is_vector)

    @staticmethod
    def read(path):
        d = read_mha(path)
        return Image(d['array'], d['origin'], d['spacing'], is_vector=d['array'].ndim > len(d['spacing']))


def compute_spacing(number_list):
    diff = np.diff(np.asarray(number_list, dtype=np.float64))
    if len(diff) and np.allclose(diff, diff[0], 1e-4):
        return float(diff[0])
    return 0.0


def get_measures_from_structured_mesh(mesh):
    """origin, size (points per axis), spacing, extent [2, dim], dim of a structured (box) mesh."""
    coords = mesh.points
    dim = coords.shape[1]
    size = np.zeros(dim, dtype=int)
    spacing = np.zeros(dim)
    extent = np.zeros((2, dim))
    for i in range(dim):
        u = np.unique(np.round(coords[:, i], 12))
        size[i], extent[0, i], extent[1, i], spacing[i] = len(u), u.min(), u.max(), compute_spacing(u)
    return extent[0].copy(), size, spacing, extent, dim


def get_measures_from_function(function):
    origin, size, spacing, extent, dim = get_measures_from_structured_mesh(function.mesh)
    v = function.values()
    return origin, size, spacing, extent, dim, (1 if v.ndim == 1 else v.shape[1])


def get_measures_from_image(image):
    origin, spacing, dim = np.array(image.GetOrigin()), np.array(image.GetSpacing()), image.GetDimension()
    size = np.array(image.GetSize(), dtype=int)
    extent = np.stack([origin, origin + spacing * (size - 1)])
    return origin, size, spacing, extent, dim, image.GetNumberOfComponentsPerPixel()


def create_image_from_fenics_function(function, size_new=None):
    """Samples a P1 function of a structured mesh on the regular grid of its vertices (or on `size_new` points per
    axis) -> Image.  On the mesh's own grid the samples are the nodal values (one sort, no point location)."""
    origin, size, spacing, extent, dim, vdim = get_measures_from_function(function)
    vals = function.values()
    if size_new is None or tuple(size_new) == tuple(size):
        ijk = np.rint((function.mesh.points - origin) / spacing).astype(np.int64)
        shape = tuple(reversed(size)) + ((vdim,) if vdim > 1 else ())
        arr = np.full(shape, np.nan)
        arr[tuple(ijk[:, k] for k in reversed(range(dim)))] = vals
        return Image(arr, origin, spacing, is_vector=vdim > 1)
    size_new = np.asarray(size_new, dtype=int)
    axes = [np.linspace(extent[0, i], extent[1, i], size_new[i]) for i in range(dim)]
    grid = np.stack(np.meshgrid(*axes, indexing='ij'), axis=-1).reshape(-1, dim)
    sampled = np.asarray(function(grid))                         # P1 evaluation (small grids only)
    arr = sampled.reshape(tuple(size_new) + ((vdim,) if vdim > 1 else ()))
    arr = np.swapaxes(arr, 0, dim - 1) if dim > 1 else arr
    return Image(arr, origin, [compute_spacing(a) for a in axes], is_vector=vdim > 1)


def create_fenics_function_from_image(image):
    """One vertex per voxel: Rectangle/BoxMesh with (size - 1) cells per axis over the image extent, voxel values as
    nodal values (scalar or vector)."""
    from ..fenics_local import Function
    from ..mesh import BoxMesh
    origin, size, spacing, extent, dim, vdim = get_measures_from_image(image)
    n = [int(s) - 1 for s in size]
    mesh = RectangleMesh(tuple(extent[0]), tuple(extent[1]), *n) if dim == 2 else \
        BoxMesh(tuple(extent[0]), tuple(extent[1]), *n)
    a = np.asarray(image.array, dtype=np.float64)
    vals = a.reshape(-1, vdim) if vdim > 1 else a.reshape(-1)     # [z,] y, x order == DOLFIN's vertex order of box meshes
    return Function(mesh, {None: vals})


create_fenics_function_from_image_quick = create_fenics_function_from_image


# ---- containers --------------------------------------------------------------------------------------------------------
def save_mesh_hdf5(mesh, path, subdomains=None, boundaries=None):
    """data_io.py:663-679 (container is .npz here)."""
    path = os.path.splitext(path)[0] + '.npz'
    extra = {}
    if subdomains is not None:
        extra['subdomains'] = np.asarray(subdomains.array() if hasattr(subdomains, 'array') else subdomains)
    if boundaries is not None:
        extra['boundaries'] = np.asarray(boundaries.array() if hasattr(boundaries, 'array') else boundaries)
    np.savez_compressed(path, points=mesh.points, cells=mesh.cells, **extra)
    return path


def read_mesh_hdf5(path):
    """data_io.py:681-713 -> (mesh, subdomains, boundaries)"""
    z = np.load(os.path.splitext(path)[0] + '.npz')
    return Mesh(z['points'], z['cells']), (z['subdomains'] if 'subdomains' in z else None), \
        (z['boundaries'] if 'boundaries' in z else None)


def merge_VTUs(directory, time_step=1, sim_time=None, remove=False, reference=None, name="solution"):
    """data_io.py:649-661: the per-step files stay as they are; a ParaView collection (.pvd) indexes them by time."""
    files = sorted(f for f in os.listdir(directory) if f.startswith(name + '_') and f.endswith('.vtu'))
    lines = ['<?xml version="1.0"?>', '<VTKFile type="Collection" version="0.1">', '<Collection>']
    for k, f in enumerate(files):
        lines.append('<DataSet timestep="%g" part="0" file="%s"/>' % (k * float(time_step), f))
    lines += ['</Collection>', '</VTKFile>']
    out = os.path.join(directory, name + '.pvd')
    with open(out, 'w') as fh:
        fh.write("\n".join(lines) + "\n")
    return out
