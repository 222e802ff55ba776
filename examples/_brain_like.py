"""Synthetic stand-in for the atlas mesh the reference loads from brain_atlas_mesh_3d.hdf5 (a git-LFS stub)."""
import numpy as np

from glimslib_amd import fenics_local as fenics


def brain_like_mesh(n=24):
    mesh = fenics.BoxMesh(fenics.Point(0, -240, 0), fenics.Point(240, 0, 155), n, n, n)
    mid = mesh.cell_midpoints()
    r = np.sqrt(((mid[:, 0] - 120) / 115) ** 2 + ((mid[:, 1] + 120) / 115) ** 2 + ((mid[:, 2] - 77.5) / 72) ** 2)
    # Ventricles (4) inside white matter (3) inside grey matter (2) inside CSF (1)
    labels = np.where(r < 0.2, 4, np.where(r < 0.62, 3, np.where(r < 0.85, 2, 1)))
    return mesh, labels
