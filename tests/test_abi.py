"""The C-ABI library loads and exports every symbol include/glims_hip.h declares (no compute calls, no GPU)."""
import ctypes
import os
import re

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
    assert lib.glims_abi_version() == 1


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
