"""The C-ABI library loads and exports every symbol include/glims_hip.h declares (no compute calls, no GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "glims_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(glims_[a-z_0-9]+)\s*\(", src)))


def test_header_declares_the_documented_surface():
    names = _declared()
    for must in ("glims_create", "glims_destroy", "glims_set_materials", "glims_setup", "glims_set_state",
                 "glims_step", "glims_solve_mechanics", "glims_get_state", "glims_apply", "glims_comm_init",
                 "glims_set_halo", "glims_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from glimslib_amd import _backend
    lib = _backend.load_library()
    for name in _declared():
        assert hasattr(lib, name), "libglimship.so does not export %s" % name
    assert set(_declared()) == set(_backend.SIGNATURES), "ctypes table and header disagree"
    assert lib.glims_abi_version() == _backend.ABI_VERSION == 6


def test_struct_layouts_match_header_field_order():
    from glimslib_amd import _backend
    src = open(os.path.join(ROOT, "include", "glims_hip.h")).read()
    body = re.search(r"typedef struct glims_options \{(.*?)\} glims_options;", src, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"(?:double|int)\s+([a-z_]+);", body)
    assert fields == [f for f, _ in _backend.Options._fields_]
    body = re.search(r"typedef struct glims_stats \{(.*?)\} glims_stats;", src, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"(?:double|int64_t)\s+([a-z_0-9]+);", body)
    assert fields == [f for f, _ in _backend.Stats._fields_]
    opt = _backend.Options()
    assert lib_default(opt) == 0 and opt.newton_rtol == 1e-10 and opt.check_every == 8


def lib_default(opt):
    from glimslib_amd import _backend
    return _backend.load_library().glims_options_default(ctypes.byref(opt))


def test_no_device_is_a_loud_error_not_a_fallback():
    """On a box without a GPU, glims_create must fail with GLIMS_E_NO_DEVICE; on a GPU box it must succeed."""
    import numpy as np
    from glimslib_amd import _backend
    pts = np.array([[0., 0.], [1., 0.], [0., 1.]])
    cells = np.array([[0, 1, 2]], dtype=np.int32)
    import torch
    if torch.cuda.is_available():
        h = _backend.Handle(pts, cells, np.zeros(1, dtype=np.int32))
        h.close()
    else:
        try:
            _backend.Handle(pts, cells, np.zeros(1, dtype=np.int32))
        except _backend.BackendError as e:
            assert e.code in (_backend.GLIMS_E_NO_DEVICE, _backend.GLIMS_E_HIP)
        else:
            raise AssertionError("glims_create succeeded without a GPU")


def test_bench_and_public_api_refuse_to_run_without_a_gpu(tmp_path):
    """No CPU fallback anywhere on the product path: on a box without a GPU bench.py exits with a message and a
    non-zero status, and a simulation's run() raises instead of computing something else."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("this box has a GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "c1", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "no GPU" in r.stderr and not r.stdout.strip()
    from glimslib_amd import fenics_local as fenics
    from glimslib_amd.simulation import TumorGrowth
    mesh = fenics.RectangleMesh(fenics.Point(0, 0), fenics.Point(1, 1), 4, 4)
    sim = TumorGrowth(mesh)
    sim.setup_global_parameters()
    sim.setup_model_parameters(iv_expression={0: fenics.Constant((0.0, 0.0)), 1: fenics.Constant(0.1)}, diffusion=0.1,
                               coupling=0.1, proliferation=0.1, E=1.0, poisson=0.3, sim_time=1, sim_time_step=1)
    with pytest.raises(Exception) as ei:
        sim.run(save_method=None, plot=False, output_dir=str(tmp_path))
    assert "HIP" in str(ei.value) or "device" in str(ei.value).lower()


def test_bench_gpus_n_without_a_launcher_starts_its_own_ranks():
    """`python bench.py --gpus 2` with WORLD_SIZE unset (how a driver that mirrors its 1-GPU command would call it) must not be
    a usage error: bench.py starts torch.distributed.run as a child process before anything touches the GPU and hands back the
    child's exit code.  Here (no GPU) both ranks must come up and refuse loudly: exit code 3 per rank -> non-zero overall,
    the message twice, nothing on stdout."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("this box has a GPU (covered by tests/test_gpu_multirank.py::test_bench_self_launch_two_ranks_on_one_device)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["GLIMS_FORCE_DEVICE"] = "0"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "c1", "--steps", "1",
                        "--warmup", "0"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0 and r.returncode != 2, (r.returncode, r.stderr[-2000:])
    assert "launching -m torch.distributed.run" in r.stderr and "--nproc-per-node 2" in r.stderr
    assert r.stderr.count("no GPU visible") == 2, r.stderr[-2000:]
    assert not r.stdout.strip()
