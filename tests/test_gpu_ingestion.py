"""
f3: mesh / label ingestion meets the device.  The reference converts a labelled .vtu into a DOLFIN mesh + cell
MeshFunction (utils/data_io.py:469-524, convert_vtk_mesh_to_fenics_hdf5.py:14-61) and turns a label image slice into a
label function (data_io.py:256-269); both products are then handed to TumorGrowthBrain.  Here: files are written,
read back through glimslib_amd.utils.data_io, and run on the HIP backend against the oracle.
"""
import os

import numpy as np
import pytest
from scipy.spatial import Delaunay

from glimslib_amd import fenics_local as fenics
from glimslib_amd.simulation import TumorGrowthBrain
from glimslib_amd.utils import data_io as dio
from oracle.glims_oracle import OracleTumorGrowth, rel_l2

pytestmark = pytest.mark.gpu

BRAIN = dict(E_GM=3000E-6, E_WM=3000E-6, E_CSF=1000E-6, E_VENT=1000E-6, nu_GM=0.45, nu_WM=0.45, nu_CSF=0.45,
             nu_VENT=0.3, D_GM=0.02, D_WM=0.1, rho_GM=0.05, rho_WM=0.05, coupling=0.1)
TISSUES = {1: 'CSF', 3: 'WM', 2: 'GM', 4: 'Ventricles'}


class _Hull(fenics.SubDomain):
    def inside(self, x, on_boundary):
        return on_boundary


def _oracle_tables(lab):
    t = lambda csf, gm, wm, vent: np.array([0.0, csf, gm, wm, vent])[lab]
    return (t(0.0, BRAIN['D_GM'], BRAIN['D_WM'], 0.0), t(0.0, BRAIN['rho_GM'], BRAIN['rho_WM'], 0.0), BRAIN['coupling'],
            t(BRAIN['E_CSF'], BRAIN['E_GM'], BRAIN['E_WM'], BRAIN['E_VENT']),
            t(BRAIN['nu_CSF'], BRAIN['nu_GM'], BRAIN['nu_WM'], BRAIN['nu_VENT']))


def test_unstructured_vtu_with_element_block_ids_runs_on_the_device(tmp_path):
    """A 3-D unstructured tetrahedral mesh with a cell array 'ElementBlockIds' and one ORPHANED vertex, written as
    .vtu, converted (orphan removed, labels kept) and run through TumorGrowthBrain for 3 steps."""
    rng = np.random.default_rng(11)
    pts = rng.random((1500, 3)) * np.array([24.0, 20.0, 16.0])
    cells = Delaunay(pts).simplices.astype(np.int32)
    X = pts[cells]
    vol = np.abs(np.linalg.det(X[:, 1:] - X[:, :1])) / 6.0
    cells = cells[vol > 1e-6 * vol.mean()]
    mid = pts[cells].mean(axis=1)
    r = np.linalg.norm((mid - np.array([12.0, 10.0, 8.0])) / np.array([9.0, 7.5, 6.0]), axis=1)
    blocks = np.where(r < 0.35, 4, np.where(r < 0.7, 3, np.where(r < 1.0, 2, 1)))       # ventricle, WM, GM, CSF
    pts_file = np.vstack([pts[:700], [[-5.0, -5.0, -5.0]], pts[700:]])                 # vertex 700 belongs to no cell
    cells_file = np.where(cells >= 700, cells + 1, cells)
    path = dio.write_vtu(os.path.join(str(tmp_path), "brain_like.vtu"), pts_file, cells_file,
                         cell_fields={'ElementBlockIds': blocks})
    assert list(dio.identify_orphaned_vertices(pts_file, cells_file)) == [700]
    mesh, sub = dio.convert_vtu_to_mesh(path)
    assert mesh.num_vertices() == len(pts) and np.array_equal(mesh.cells, cells) and np.array_equal(sub, blocks)
    assert np.allclose(mesh.points, pts, rtol=0, atol=1e-14)

    sim = TumorGrowthBrain(mesh)
    sim.setup_global_parameters(subdomains=sub, domain_names=TISSUES, boundaries={'hull': _Hull()},
                                dirichlet_bcs={'clamped': {'bc_value': fenics.Constant((0.0, 0.0, 0.0)),
                                                           'named_boundary': 'hull', 'subspace_id': 0}},
                                von_neumann_bcs={})
    iv = fenics.Expression('exp(-a*(pow(x[0]-12, 2) + pow(x[1]-10, 2) + pow(x[2]-4, 2)))', degree=1, a=0.08)
    sim.setup_model_parameters(iv_expression={0: fenics.Constant((0.0, 0.0, 0.0)), 1: iv},
                               sim_time=3, sim_time_step=1, **BRAIN)
    sol = sim.run(save_method=None, plot=False)
    lab = np.asarray(sub)
    D, rho, gam, E, nu = _oracle_tables(lab)
    f = mesh.facets()
    bn = np.unique(f['vertices'][f['exterior']])
    dofs = (bn[:, None] * 3 + np.arange(3)).ravel()
    o = OracleTumorGrowth(mesh.points, mesh.cells, D, rho, gam, E, nu, 1.0, dirichlet_u=(dofs, np.zeros(len(dofs))))
    uo, co = o.run(sim.params.create_initial_value_function().components[1], 3.0)
    print("ingested vtu: c %.2e, u %.2e" % (rel_l2(sol.components[1], co), rel_l2(sol.components[0].reshape(-1), uo)))
    assert rel_l2(sol.components[1], co) < 1e-9
    assert rel_l2(sol.components[0].reshape(-1), uo) < 1e-6       # Delaunay slivers: ill-conditioned K_el
    sim.close()


def test_label_image_slice_becomes_the_label_function_of_a_2d_run(tmp_path):
    """A 3-D label image (.mha, zlib) -> z-slice -> pixel mesh + nodal label function -> subdomains through the
    reference's rule -> 2-D TumorGrowthBrain run, against the oracle with the same cell labels."""
    z, ny, nx = 3, 33, 41
    yy, xx = np.meshgrid(np.arange(ny), np.arange(nx), indexing='ij')
    r = np.hypot((xx - 20) / 18.0, (yy - 16) / 14.0)
    sl = np.where(r < 0.3, 4, np.where(r < 0.65, 3, np.where(r < 0.95, 2, 1))).astype(np.int16)
    vol = np.stack([np.ones_like(sl), sl, np.ones_like(sl)])
    p = os.path.join(str(tmp_path), "labels.mha")
    dio.write_mha(p, vol, origin=[-20.0, -16.0, 0.0], spacing=[1.0, 1.0, 2.0], compressed=True)
    mesh, labelfun = dio.get_labelfunction_from_image(p, z_slice=1)
    assert mesh.num_vertices() == nx * ny
    sim = TumorGrowthBrain(mesh)
    sim.setup_global_parameters(label_function=labelfun, domain_names=TISSUES, boundaries={'hull': _Hull()},
                                dirichlet_bcs={'clamped': {'bc_value': fenics.Constant((0.0, 0.0)),
                                                           'named_boundary': 'hull', 'subspace_id': 0}},
                                von_neumann_bcs={})
    lab = np.asarray(sim.subdomains.subdomains.array())
    assert set(np.unique(lab)) == {1, 2, 3, 4}
    iv = fenics.Expression('exp(-a*(pow(x[0]-6, 2) + pow(x[1]+2, 2)))', degree=1, a=0.1)
    sim.setup_model_parameters(iv_expression={0: fenics.Constant((0.0, 0.0)), 1: iv}, sim_time=4, sim_time_step=1,
                               **BRAIN)
    sol = sim.run(save_method=None, plot=False)
    D, rho, gam, E, nu = _oracle_tables(lab)
    f = mesh.facets()
    bn = np.unique(f['vertices'][f['exterior']])
    dofs = (bn[:, None] * 2 + np.arange(2)).ravel()
    o = OracleTumorGrowth(mesh.points, mesh.cells, D, rho, gam, E, nu, 1.0, dirichlet_u=(dofs, np.zeros(len(dofs))))
    uo, co = o.run(sim.params.create_initial_value_function().components[1], 4.0)
    assert rel_l2(sol.components[1], co) < 1e-9 and rel_l2(sol.components[0].reshape(-1), uo) < 1e-8
    sim.close()
