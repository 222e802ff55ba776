"""The C/OpenMP restatement (oracle/glims_oracle_c.c) against the numpy oracle -- two independent CPU
implementations of the same scheme (different storage, different element-integral code path).  No GPU needed."""
import numpy as np
import pytest

from oracle.c_port import COracle
from oracle.glims_oracle import OracleTumorGrowth, box_mesh, rectangle_mesh, rel_l2


@pytest.mark.parametrize("dim", [2, 3])
def test_c_port_matches_numpy_oracle(dim):
    pts, cells = (rectangle_mesh((0, 0), (3, 2), 13, 9) if dim == 2 else box_mesh((0, 0, 0), (1, 1.2, 0.9), 7, 6, 5))
    rng = np.random.default_rng(0)
    m = len(cells)
    D, rho = rng.random(m) * 0.1, rng.random(m) * 0.1
    rho[: m // 5] = 0.0
    o = OracleTumorGrowth(pts, cells, D, rho, 0.0, 1.0, 0.3, 0.7)
    co = COracle(pts, cells, D, rho, 0.7)
    x = rng.standard_normal(len(pts))
    assert rel_l2(co.apply(2, x), o.M @ x) < 1e-14 and rel_l2(co.apply(1, x), o.S @ x) < 1e-14
    c0 = np.exp(-3 * ((pts - pts.mean(0)) ** 2).sum(1))
    load = 0.01 * rng.random(len(pts))
    o.rd_load = load
    c = c0.copy()
    for _ in range(3):
        c, _ = o.rd_step(c)
    cc = co.step(c0, 3, load=load)
    assert rel_l2(cc, c) < 1e-10
    assert rel_l2(co.apply(0, x), o.rd_jacobian(cc) @ x) < 1e-9      # A left at the last sweep point
    co.close()


def test_c_port_uniform_field_recurrence():
    pts, cells = box_mesh((0, 0, 0), (1, 1, 1), 5, 4, 3)
    co = COracle(pts, cells, 0.3, 0.1, 1.0)
    c = co.step(np.full(len(pts), 0.3), 3)
    cn = 0.3
    for _ in range(3):
        cn = (-(1 - .1) + np.sqrt((1 - .1) ** 2 + 4 * .1 * cn)) / (2 * .1)
    assert np.abs(c - cn).max() < 1e-12


def test_c_oracle_is_clean_under_address_and_ub_sanitizers():
    """`make -C oracle sanitize`: the C restatement on a small 3-D problem with
    -fsanitize=address,undefined; any report makes the run fail."""
    import os
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        import pytest
        pytest.skip("no gcc")
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    r = subprocess.run(["make", "-C", here, "sanitize"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "sanitize_main: status 0" in r.stdout and "ERROR" not in r.stderr and "runtime error" not in r.stderr
