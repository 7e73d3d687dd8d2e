"""Tools that need the diagnostic build of the library (-DSAA_DIAGNOSTICS: ablated step kernels, in-kernel stamps,
saa_debug_* exports) import this module BEFORE the package: it builds ``libsaa_hip_diag.so`` when missing or stale and
points SAA_LIB_PATH at it.  The product library never contains any of that."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
_pkg = os.path.join(REPO, "synchronization_avoiding_algorithms_amd")
_diag = os.path.join(_pkg, "libsaa_hip_diag.so")


def diag_library_path():
    """Path of the diagnostic build, built first when missing or older than the sources (tests that need one of its
    switches run a child process with SAA_LIB_PATH pointing there)."""
    csrc = os.path.join(_pkg, "csrc")
    newest = max(os.path.getmtime(os.path.join(csrc, f)) for f in os.listdir(csrc))
    if not os.path.exists(_diag) or os.path.getmtime(_diag) < newest:
        import importlib.util

        spec = importlib.util.spec_from_file_location("_saa_lib_build", os.path.join(_pkg, "_lib.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build_library(diag=True)
    return _diag


def use_diag_library(extra_flags=()):
    if "SAA_LIB_PATH" in os.environ:
        return os.environ["SAA_LIB_PATH"]
    csrc = os.path.join(_pkg, "csrc")
    newest = max(os.path.getmtime(os.path.join(csrc, f)) for f in os.listdir(csrc))
    if extra_flags or not os.path.exists(_diag) or os.path.getmtime(_diag) < newest:
        import importlib.util

        spec = importlib.util.spec_from_file_location("_saa_lib_build", os.path.join(_pkg, "_lib.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build_library(diag=True, extra_flags=tuple(extra_flags))
    os.environ["SAA_LIB_PATH"] = _diag
    return _diag
