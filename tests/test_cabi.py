"""The C-ABI library loads on a GPU-less host and exports every function include/saa_hip.h declares
(no compute call is made here); host-only plan builder sanity."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import REPO
from synchronization_avoiding_algorithms_amd import _lib


def _declared_functions():
    text = open(os.path.join(REPO, "include", "saa_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(saa_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    names = _declared_functions()
    assert len(names) >= 20
    assert sorted(_lib.SIGNATURES) == names


def test_library_exports_every_declared_symbol():
    _lib.build_library()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared_functions():
        assert hasattr(lib, name), name
    assert _lib.load().saa_abi_version() == _lib.ABI_VERSION
    # the product library exports the header's entry points and nothing else of ours: diagnostics (ablated kernels,
    # stamps, saa_debug_*) live in the separate libsaa_hip_diag.so the tools build for themselves
    import subprocess

    syms = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in syms.splitlines() if " T " in ln and ln.split()[-1].startswith("saa_")}
    assert exported == set(_declared_functions()), exported ^ set(_declared_functions())


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_product_does_not_import_oracle():
    pkg = os.path.join(REPO, "synchronization_avoiding_algorithms_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_host_plan_properties(beam_coarse):
    from synchronization_avoiding_algorithms_amd import plan_host_stats
    from synchronization_avoiding_algorithms_amd.mesh import structured_beam

    st = plan_host_stats(beam_coarse.points, beam_coarse.tets)
    assert st["n_blocks"] == 1 and st["n_elem_copies"] == 256 and st["n_halo_total"] == 0
    mesh = structured_beam(6)
    for bn in (64, 200, 0):
        st = plan_host_stats(mesh.points, mesh.tets, bn)
        assert st["n_elem_copies"] >= len(mesh.tets)
        assert st["max_local"] >= st["max_owned"] and st["lds_bytes"] <= 160 * 1024
        assert 1.0 <= st["lds_conflict_factor"] < 4.0
    # malformed input is rejected with a message, not a crash
    bad = mesh.tets.copy()
    bad[0, 0] = len(mesh.points) + 5
    with pytest.raises(_lib.SaaError, match="outside"):
        plan_host_stats(mesh.points, bad)


def test_synthetic_beam_shapes():
    from synchronization_avoiding_algorithms_amd import fem_setup as fs
    from synchronization_avoiding_algorithms_amd.mesh import clamp_nodes, structured_beam

    m = structured_beam(2)
    assert len(m.tets) == 25 * 2 * 2 * 2 * 6 and len(m.points) == 51 * 3 * 3
    vol = fs.signed_volumes(m.points, m.tets)
    assert (vol > 0).all() and np.isclose(vol.sum(), 25.0)
    assert len(clamp_nodes(m)) == 9
    lumped, load = fs.lumped_mass_and_load(m.points, m.tets, 1.0, 0.5)
    assert np.isclose(lumped.sum() / 3, 25.0) and np.allclose(load.reshape(-1, 3).sum(0), [0, -12.5, -12.5])


def test_predictor_size_bounds_are_the_ones_the_kernels_index_with():
    """saa_predictor_create validates the shape before it touches a device: an input size whose 32-bit tile offsets would
    overflow is refused with a clear message, while the width of a large k-way partition (30 000 inputs = 10 000 shared
    nodes; round 3 refused everything above 23 170) passes the validation and only then fails for want of a GPU."""
    import ctypes as C

    from synchronization_avoiding_algorithms_amd import _lib

    lib = _lib.load()
    dummy = (C.c_float * 4)()
    ptrs = (C.POINTER(C.c_float) * 22)(*[C.cast(dummy, C.POINTER(C.c_float))] * 22)
    h = C.c_void_p()
    rc = lib.saa_predictor_create(0, 3_000_000, 50, 20, 20, 150, ptrs, 22, C.byref(h))
    assert rc == _lib.SAA_E_ARG and b"input_size too large" in lib.saa_last_error()
    rc = lib.saa_predictor_create(0, 5, 200, 20, 20, 150, ptrs, 22, C.byref(h))
    assert rc == _lib.SAA_E_ARG and b"hidden size above 128" in lib.saa_last_error()
    import torch

    if not torch.cuda.is_available():  # (on a GPU box this would go on to read 30 000-wide weights from the dummy)
        rc = lib.saa_predictor_create(0, 30_000, 50, 20, 20, 150, ptrs, 22, C.byref(h))
        assert rc == _lib.SAA_E_HIP, (rc, lib.saa_last_error())  # past the shape checks: only the device is missing
