"""ctypes binding of ``libsaa_hip.so`` (C ABI in ``include/saa_hip.h``) and its in-tree build.

There is no CPU fallback: if the library cannot be loaded every compute entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("SAA_LIB_PATH") or os.path.join(_HERE, "libsaa_hip.so")  # override: experiments only
#: diagnostic build (-DSAA_DIAGNOSTICS: ablated step kernels, in-kernel stamps, saa_debug_* exports) used by tools/
#: only; built on request (``build_library(diag=True)``), loaded through SAA_LIB_PATH, never by the package itself
DIAG_LIB_PATH = os.path.join(_HERE, "libsaa_hip_diag.so")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "saa_hip.h")
SOURCES = ["saa_plan.cpp", "saa_partition.cpp", "saa_kernels.hip", "saa_setup.hip", "saa_predictor.hip", "saa_topology.hip", "saa_api.cpp"]
HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics", "-ldl"]

ABI_VERSION = 10  # what saa_abi_version() of a matching library returns (include/saa_hip.h)
SAA_OK, SAA_E_ARG, SAA_E_HIP, SAA_E_STATE, SAA_E_CAPACITY = 0, -1, -2, -3, -4


class SaaError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"libsaa_hip error {code}: {message}")
        self.code = code


class Problem(C.Structure):
    _fields_ = [
        ("n_nodes", C.c_int32), ("n_elems", C.c_int32),
        ("xyz", C.POINTER(C.c_double)), ("tets", C.POINTER(C.c_int32)),
        ("lumped_mass", C.POINTER(C.c_double)), ("f_ext", C.POINTER(C.c_double)),
        ("dirichlet_dofs", C.POINTER(C.c_int32)), ("n_dirichlet", C.c_int32),
        ("shared_nodes", C.POINTER(C.c_int32)), ("shared_slots", C.POINTER(C.c_int32)),
        ("n_shared", C.c_int32), ("n_global_shared", C.c_int32),
        ("lambda_", C.c_double), ("mu", C.c_double), ("dt", C.c_double), ("alpha", C.c_double),
        ("ramp", C.c_int32), ("device", C.c_int32), ("block_nodes", C.c_int32), ("threads", C.c_int32),
    ]


class PartitionStats(C.Structure):
    _fields_ = [("face_cut", C.c_int64), ("min_part", C.c_int64), ("max_part", C.c_int64), ("interface_nodes", C.c_int32)]

    def as_dict(self):
        return {name: int(getattr(self, name)) for name, _ in self._fields_}


class PlanStats(C.Structure):
    _fields_ = [
        ("n_blocks", C.c_int32), ("max_owned", C.c_int32), ("max_local", C.c_int32),
        ("n_elem_copies", C.c_int64), ("n_halo_total", C.c_int64),
        ("lds_bytes", C.c_int32), ("threads", C.c_int32), ("lds_conflict_factor", C.c_double),
        ("lds_atomic_conflict_factor", C.c_double), ("n_items", C.c_int64), ("n_pairs", C.c_int64),
        ("n_by_construction", C.c_int64), ("n_renumbered", C.c_int32), ("reserved", C.c_int32),
    ]

    def as_dict(self):
        return {name: (float if name.endswith("conflict_factor") else int)(getattr(self, name))
                for name, _ in self._fields_ if name != "reserved"}


_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_H = C.c_void_p

#: name -> (restype, argtypes); must list every function declared in include/saa_hip.h
SIGNATURES = {
    "saa_last_error": (C.c_char_p, []),
    "saa_abi_version": (C.c_int32, []),
    "saa_part_mesh_kway": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _ip, _ip, C.POINTER(PartitionStats)]),
    "saa_setup_fields": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _dp, _ip, C.c_double, C.c_double, _dp, _dp, _dp]),
    "saa_create": (C.c_int, [C.POINTER(Problem), C.POINTER(_H)]),
    "saa_destroy": (C.c_int, [_H]),
    "saa_plan_stats_get": (C.c_int, [_H, C.POINTER(PlanStats)]),
    "saa_plan_host_stats": (C.c_int, [C.c_int32, C.c_int32, _dp, _ip, C.c_int32, C.POINTER(PlanStats)]),
    "saa_plan_host_check": (C.c_int, [C.c_int32, C.c_int32, _dp, _ip, C.c_int32, C.POINTER(C.c_int64)]),
    "saa_set_stream": (C.c_int, [_H, C.c_void_p]),
    "saa_set_state": (C.c_int, [_H, _dp, _dp, C.c_double]),
    "saa_get_state": (C.c_int, [_H, _dp, _dp, _dp]),
    "saa_get_state_device": (C.c_int, [_H, C.c_void_p, C.c_void_p]),
    "saa_set_loads": (C.c_int, [_H, _dp, _dp]),
    "saa_internal_force": (C.c_int, [_H, _dp, _dp]),
    "saa_internal_force_device": (C.c_int, [_H, C.c_void_p, C.c_void_p]),
    "saa_cd_update": (C.c_int, [_H, _dp, _dp, _dp, C.c_double, _dp]),
    "saa_step": (C.c_int, [_H, C.c_int32]),
    "saa_set_interface_buffer": (C.c_int, [_H, C.c_void_p]),
    "saa_step_begin": (C.c_int, [_H]),
    "saa_step_finish": (C.c_int, [_H, C.c_void_p, C.c_int64]),
    "saa_comm_unique_id": (C.c_int, [C.c_char_p, C.POINTER(C.c_uint8)]),
    "saa_comm_init": (C.c_int, [_H, C.c_char_p, C.POINTER(C.c_uint8), C.c_int32, C.c_int32]),
    "saa_step_synced": (C.c_int, [_H, C.c_int32, C.c_void_p, C.c_int64]),
    "saa_resident_kernel_info": (C.c_int, [_H, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "saa_set_recorder": (C.c_int, [_H, C.c_void_p, C.c_int64, C.c_int32, C.c_int64]),
    "saa_set_resident_kernel": (C.c_int, [_H, C.c_int32]),
    "saa_set_option": (C.c_int, [_H, C.c_char_p, C.c_double]),
    "saa_set_deterministic": (C.c_int, [_H, C.c_int32]),
    "saa_peer_export": (C.c_int, [_H, C.c_int32, C.POINTER(C.c_uint8), C.POINTER(C.c_int32)]),
    "saa_peer_attach": (C.c_int, [_H, C.c_int32, C.c_int32, C.POINTER(C.c_uint8), C.POINTER(C.c_int32),
                                  C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "saa_peer_selftest": (C.c_int, [_H, C.POINTER(C.c_int32)]),
    "saa_peer_attach_loopback": (C.c_int, [_H, C.c_int32]),
    "saa_step_peer": (C.c_int, [_H, C.c_int32, C.c_void_p, C.c_int64]),
    "saa_step_predicted": (C.c_int, [_H, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]),
    "saa_halo_gather": (C.c_int, [_H, C.c_void_p]),
    "saa_halo_scatter": (C.c_int, [_H, C.c_void_p]),
    "saa_synchronize": (C.c_int, [_H]),
    "saa_time_steps": (C.c_int, [_H, C.c_int32, _dp]),
    "saa_device_copy_bandwidth": (C.c_int, [C.c_int32, C.c_int64, C.c_int32, _dp]),
    "saa_predictor_create": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                       C.POINTER(C.POINTER(C.c_float)), C.c_int32, C.POINTER(_H)]),
    "saa_predictor_predict": (C.c_int, [_H, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_double,
                                        C.c_void_p, C.c_int64, C.c_void_p]),
    "saa_predictor_destroy": (C.c_int, [_H]),
    "saa_lstm_cell_forward": (C.c_int, [C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 7),
    "saa_lstm_cell_backward": (C.c_int, [C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 8),
    "saa_lstm_recurrence_forward": (C.c_int, [C.c_int32] * 5 + [C.c_void_p] * 9),
    "saa_lstm_recurrence_backward": (C.c_int, [C.c_int32] * 5 + [C.c_void_p] * 11),
    "saa_train_stats": (C.c_int, [C.c_int32, C.c_int64] + [C.c_void_p] * 5),
    "saa_topology_build": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _ip, _ip, C.c_int32, C.c_int32, _dp, C.c_int32, _ip,
                                     C.c_double, C.POINTER(_H)]),
    "saa_topology_sizes": (C.c_int, [_H, _ip]),
    "saa_topology_get": (C.c_int, [_H, _ip, _ip, _ip, _ip, _ip, _ip, _ip, _ip, _ip]),
    "saa_topology_destroy": (C.c_int, [_H]),
}

_lib = None


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(_CSRC, f) for f in os.listdir(_CSRC)] + [HEADER]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False, diag: bool = False, extra_flags=()) -> str:
    """Compile the HIP sources for gfx950 into ``libsaa_hip.so`` next to this file (in-tree).  ``diag=True`` builds
    the diagnostic variant ``libsaa_hip_diag.so`` instead (tools/ only)."""
    out = DIAG_LIB_PATH if diag else LIB_PATH
    if not force and not diag and not needs_build():
        return out
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: libsaa_hip.so cannot be built")
    flags = [*HIPCC_FLAGS, *(["-DSAA_DIAGNOSTICS"] if diag else []), *extra_flags]
    cmd = [hipcc, *flags, *SOURCES[:-1], "-x", "hip", SOURCES[-1], "-o", out + ".tmp"]
    res = subprocess.run(cmd, cwd=_CSRC, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + res.stderr)
    if verbose and res.stderr:
        print(res.stderr)
    os.replace(out + ".tmp", out)
    return out


def load():
    """Load the library (once) and attach the prototypes.  Raises if it is missing - no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(the HIP extension is mandatory; there is no CPU fallback)")
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64.so.  If libsaa_hip.so came first it would
    # pull in the system copy under the same soname and a later `import torch` would find "No HIP GPUs".  Loading
    # torch first makes both use torch's copy (the package needs torch anyway for streams and device buffers).
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = C.CDLL(LIB_PATH)
    lib.saa_abi_version.restype = C.c_int32
    old_ok = os.environ.get("SAA_ALLOW_OLD_ABI") == "1"  # set by tools/ab.py only: an older build next to the current one
    if old_ok and 6 <= lib.saa_abi_version() < ABI_VERSION:
        pass  # (newer entry points are missing there)
    elif lib.saa_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH} has ABI version {lib.saa_abi_version()}, this package needs {ABI_VERSION}: "
                           "rebuild it (`python -c 'import __graft_entry__ as g; g.build()'`)")
    for name, (res, args) in SIGNATURES.items():
        if old_ok and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code: int):
    if code != 0:
        msg = load().saa_last_error()
        raise SaaError(code, msg.decode() if msg else "")
