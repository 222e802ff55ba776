"""
Pins the CPU oracle (oracle/glims_oracle.py) with the known-answer tests K1-K8 of SURVEY.md section 8c and with
the one golden fixture that comes from the reference itself.  No GPU needed.
"""
import json
import os

import numpy as np
import pytest

from oracle.glims_oracle import (OracleTumorGrowth, box_mesh, rectangle_mesh, boundary_facets, compute_growth_logistic,
                                 assemble_mass, assemble_stiffness, p1_geometry, rel_l2, compute_mu, compute_lambda)

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_golden_logistic_from_reference():
    g = json.load(open(os.path.join(GOLD, "logistic_growth.json")))
    for row in g["scalar_cases"]:
        assert compute_growth_logistic(row["conc"], row["prolif_rate"], row["conc_max"]) == row["expected"]
    v = g["vector_case"]
    np.testing.assert_array_equal(compute_growth_logistic(np.array(v["conc"]), v["prolif_rate"], v["conc_max"]),
                                  np.array(v["expected"]))
    # the SURVEY's quoted sample
    np.testing.assert_allclose(compute_growth_logistic(np.array([0, .25, .5, 1]), 0.1, 1.0), [0, 0.01875, 0.025, 0])


def test_lame_constants():
    assert compute_mu(3.0, 0.5) == 1.0
    assert abs(compute_lambda(1.0, 0.25) - 0.4) < 1e-15


@pytest.mark.parametrize("dim", [2, 3])
def test_K1_uniform_field_follows_scalar_recurrence(dim):
    pts, cells = (rectangle_mesh((0, 0), (2, 1), 7, 5) if dim == 2 else box_mesh((0, 0, 0), (1, 2, 1), 4, 5, 3))
    rho, dt = 0.1, 1.0
    o = OracleTumorGrowth(pts, cells, D=0.3, rho=rho, gamma=0.1, E=1e-3, nu=0.4, dt=dt)
    c = np.full(len(pts), 0.3)
    cn = 0.3
    for _ in range(4):
        c, _ = o.rd_step(c)
        a = dt * rho
        cn = (-(1 - a) + np.sqrt((1 - a) ** 2 + 4 * a * cn)) / (2 * a)
        assert np.abs(c - cn).max() < 1e-13


@pytest.mark.parametrize("dim", [2, 3])
def test_K2_mass_conservation_without_proliferation(dim):
    pts, cells = (rectangle_mesh((0, 0), (2, 1), 9, 6) if dim == 2 else box_mesh((0, 0, 0), (1, 2, 1), 4, 5, 3))
    o = OracleTumorGrowth(pts, cells, D=0.05, rho=0.0, gamma=0.1, E=1e-3, nu=0.4, dt=0.5)
    c = np.exp(-4 * ((pts - pts.mean(0)) ** 2).sum(1))
    m0 = (o.M @ c).sum()
    for _ in range(3):
        c, _ = o.rd_step(c)
    assert abs((o.M @ c).sum() - m0) < 1e-13 * abs(m0)


def test_K3_one_diffusion_step_vs_dense_solve():
    pts, cells = box_mesh((0, 0, 0), (1, 1, 1), 3, 3, 3)
    D, dt = 0.2, 0.7
    o = OracleTumorGrowth(pts, cells, D=D, rho=0.0, gamma=0.0, E=1.0, nu=0.3, dt=dt)
    c0 = np.sin(3 * pts[:, 0]) + pts[:, 1] ** 2
    c1, _ = o.rd_step(c0)
    M = assemble_mass(pts, cells).toarray()
    K = assemble_stiffness(pts, cells, np.full(len(cells), D)).toarray()
    ref = np.linalg.solve(M + dt * K, M @ c0)
    assert rel_l2(c1, ref) < 1e-12


@pytest.mark.parametrize("dim", [2, 3])
def test_K4_K5_uniform_growth_strain_and_patch_test(dim):
    pts, cells = (rectangle_mesh((0, 0), (2, 1), 5, 4) if dim == 2 else box_mesh((0, 0, 0), (1, 2, 1), 3, 4, 3))
    bf, _ = boundary_facets(cells)
    bn = np.unique(bf)
    dofs = (bn[:, None] * dim + np.arange(dim)).ravel()
    gamma, cval = 0.2, 0.5
    uex = (gamma * cval * (pts - pts[0])).ravel()          # u = gamma c (x - x0): stress-free growth
    o = OracleTumorGrowth(pts, cells, 0.1, 0.1, gamma, 1e-3, 0.4, 1.0, dirichlet_u=(dofs, uex[dofs]))
    u = o.mech_solve(np.full(len(pts), cval))
    assert np.abs(u - uex).max() < 1e-13
    # K5: linear displacement field reproduced with c = 0
    A = np.arange(dim * dim).reshape(dim, dim) * 0.01 + 0.02 * np.eye(dim)
    ulin = (pts @ A.T).ravel()
    o = OracleTumorGrowth(pts, cells, 0.1, 0.1, gamma, 1e-3, 0.3, 1.0, dirichlet_u=(dofs, ulin[dofs]))
    u = o.mech_solve(np.zeros(len(pts)))
    assert np.abs(u - ulin).max() < 1e-13


def test_residual_identity_and_jacobian_fd():
    rng = np.random.default_rng(1)
    pts, cells = box_mesh((0, 0, 0), (1, 1, 1), 4, 4, 4)
    rho = rng.random(len(cells)) * 0.2
    o = OracleTumorGrowth(pts, cells, D=rng.random(len(cells)), rho=rho, gamma=0.1, E=1e-3, nu=0.4, dt=0.8)
    c, cp, v = rng.random(len(pts)), rng.random(len(pts)), rng.standard_normal(len(pts))
    J = o.rd_jacobian(c)
    R = o.rd_residual(c, cp)
    assert np.abs(R - (0.5 * (J @ c + o.S @ c) - o.M @ cp)).max() < 1e-15      # the identity the HIP kernel uses
    eps = 1e-6
    fd = (o.rd_residual(c + eps * v, cp) - o.rd_residual(c - eps * v, cp)) / (2 * eps)
    assert np.abs(J @ v - fd).max() < 1e-9 * np.abs(fd).max()
    assert abs(J - J.T).max() < 1e-15


def test_monolithic_newton_equals_split_solve():
    """The reference's monolithic SNES/LU iteration and 'Newton on c, then one elastic solve' share the fixed point."""
    pts, cells = box_mesh((0, 0, 0), (1, 1, 1), 5, 5, 5)
    bf, _ = boundary_facets(cells)
    bn = np.unique(bf)
    dofs = (bn[:, None] * 3 + np.arange(3)).ravel()
    lab = (pts[cells].mean(1)[:, 0] > 0.5).astype(int)
    o = OracleTumorGrowth(pts, cells, np.array([0.01, 0.03])[lab], np.array([0.1, 0.05])[lab], 0.2,
                          np.array([1e-3, 3e-3])[lab], np.array([0.4, 0.45])[lab], 1.0,
                          dirichlet_u=(dofs, np.zeros(len(dofs))))
    c0 = np.exp(-20 * ((pts - 0.5) ** 2).sum(1))
    u1, c1 = o.run(c0, 3.0)
    u2, c2 = o.run(c0, 3.0, monolithic=True)
    assert rel_l2(c1, c2) < 1e-11 and rel_l2(u1, u2) < 1e-10


def test_time_loop_guard():
    pts, cells = rectangle_mesh((0, 0), (1, 1), 3, 3)
    o = OracleTumorGrowth(pts, cells, 0.1, 0.1, 0.1, 1e-3, 0.4, 0.5)
    o.run(np.full(len(pts), 0.1), 2.0, mechanics=False)
    assert o.n_steps == 4          # t <= T - 1e-5 (simulation_base.py:277)
    o.run(np.full(len(pts), 0.1), 2.00001, mechanics=False)
    assert o.n_steps == 5


def test_K8_fisher_front_speed_sanity():
    """1-D-like strip: front speed tends to 2 sqrt(D rho) (loose tolerance, implicit Euler damps it)."""
    D, rho, dt = 1.0, 1.0, 0.05
    pts, cells = rectangle_mesh((0, 0), (60, 0.5), 480, 1)
    o = OracleTumorGrowth(pts, cells, D, rho, 0.0, 1.0, 0.3, dt)
    c = np.where(pts[:, 0] < 2.0, 1.0, 0.0)
    pos = []
    for k in range(400):
        c, _ = o.rd_step(c, rtol=1e-9)
        if k in (199, 399):
            x = pts[:, 0][(np.abs(pts[:, 1]) < 1e-12)]
            cc = c[(np.abs(pts[:, 1]) < 1e-12)]
            order = np.argsort(x)
            pos.append(np.interp(0.5, cc[order][::-1], x[order][::-1]))
    speed = (pos[1] - pos[0]) / (200 * dt)
    assert 1.6 < speed < 2.1


def test_golden_oracle_regression():
    from glimslib_amd import workloads
    w = workloads.config_c1()
    g = np.load(os.path.join(GOLD, "oracle_c1.npz"))
    dofs = (w.dirichlet_nodes[:, None] * 2 + np.arange(2)).ravel()
    o = OracleTumorGrowth(w.mesh.points, w.mesh.cells, w.per_cell('D'), w.per_cell('rho'), w.per_cell('gamma'),
                          w.per_cell('E'), w.per_cell('nu'), w.dt, dirichlet_u=(dofs, np.zeros(len(dofs))))
    u, c = o.run(w.c0, 3 * w.dt)
    u10, c10 = o.run(c * 0 + w.c0, w.n_steps * w.dt)
    assert rel_l2(c10, g['c']) < 1e-12 and rel_l2(u10, g['u']) < 1e-10
