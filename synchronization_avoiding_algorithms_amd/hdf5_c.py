"""HDF5 through the C library itself (``libhdf5``, ctypes) where ``h5py`` is not importable.

The reference stores its trajectories with ``h5py`` (``Data_prepare.py:243-246``: ``create_dataset('Displacement',
data=d1_save, compression='gzip')``; ``Shared_extraction.py:38-40`` without compression; read back by
``DNN_tools.py:286`` / ``Shared_extraction.py:32-36``).  ``h5py`` is a binding of ``libhdf5``; where only the library is
installed (this image: HDF5 1.10 under ``/opt/conda/lib``, no ``h5py``) the same files are written and read by calling
it directly: a 2-D little-endian float64 dataset, and for ``compression='gzip'`` what ``h5py`` does with that argument -
chunked layout with its automatic chunk shape (``h5py/_hl/filters.py: guess_chunk``, restated below) and the deflate
filter at level 4, fill time "alloc".  Pinned against files ``h5py`` 3.3 itself wrote (``tests/golden/make_golden_hdf5.py``,
run with the image's Anaconda interpreter, which has it): byte for byte equal except the four bytes of the modification
time stamp.
"""
from __future__ import annotations

import ctypes as C
import ctypes.util
import os
import sys

import numpy as np

_lib = None
_hid = C.c_int64   # hid_t since HDF5 1.10
_hsize = C.c_uint64

H5F_ACC_RDONLY, H5F_ACC_TRUNC, H5P_DEFAULT, H5S_ALL = 0, 2, 0, 0
H5Z_FILTER_DEFLATE = 1


def _find():
    names = []
    found = ctypes.util.find_library("hdf5")
    if found:
        names.append(found)
    for d in (os.path.join(sys.prefix, "lib"), "/opt/conda/lib", "/usr/lib/x86_64-linux-gnu/hdf5/serial",
              "/usr/lib/x86_64-linux-gnu", "/usr/lib64", "/usr/local/lib"):
        if os.path.isdir(d):
            names += sorted(os.path.join(d, f) for f in os.listdir(d) if f.startswith("libhdf5.so"))
    for name in names:
        try:
            lib = C.CDLL(name)
        except OSError:
            continue
        if not hasattr(lib, "H5Dcreate2"):
            continue
        maj, mnr, rel = C.c_uint(), C.c_uint(), C.c_uint()
        lib.H5open()
        lib.H5get_libversion(C.byref(maj), C.byref(mnr), C.byref(rel))
        if (maj.value, mnr.value) < (1, 10):  # (hid_t is 32 bits before 1.10)
            continue
        return lib
    return None


def available() -> bool:
    return load() is not None


def load():
    """The library with prototypes attached, or None."""
    global _lib
    if _lib is not None:
        return _lib or None
    lib = _find()
    if lib is None:
        _lib = False
        return None
    sig = {
        "H5Fcreate": (_hid, [C.c_char_p, C.c_uint, _hid, _hid]), "H5Fopen": (_hid, [C.c_char_p, C.c_uint, _hid]),
        "H5Fclose": (C.c_int, [_hid]), "H5Screate_simple": (_hid, [C.c_int, C.POINTER(_hsize), C.POINTER(_hsize)]),
        "H5Sclose": (C.c_int, [_hid]), "H5Pcreate": (_hid, [_hid]), "H5Pclose": (C.c_int, [_hid]),
        "H5Pset_chunk": (C.c_int, [_hid, C.c_int, C.POINTER(_hsize)]), "H5Pset_deflate": (C.c_int, [_hid, C.c_uint]),
        "H5Pset_fill_time": (C.c_int, [_hid, C.c_int]),
        "H5Dcreate2": (_hid, [_hid, C.c_char_p, _hid, _hid, _hid, _hid, _hid]), "H5Dopen2": (_hid, [_hid, C.c_char_p, _hid]),
        "H5Dwrite": (C.c_int, [_hid, _hid, _hid, _hid, _hid, C.c_void_p]),
        "H5Dread": (C.c_int, [_hid, _hid, _hid, _hid, _hid, C.c_void_p]), "H5Dclose": (C.c_int, [_hid]),
        "H5Dget_space": (_hid, [_hid]), "H5Sget_simple_extent_ndims": (C.c_int, [_hid]),
        "H5Sget_simple_extent_dims": (C.c_int, [_hid, C.POINTER(_hsize), C.POINTER(_hsize)]),
        "H5Zfilter_avail": (C.c_int, [C.c_int]), "H5Eset_auto2": (C.c_int, [_hid, C.c_void_p, C.c_void_p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    lib.H5Eset_auto2(0, None, None)  # no error stack on stderr: failures become Python exceptions below
    _lib = lib
    return lib


def library_version():
    """(major, minor, release) of the HDF5 library in use, or None."""
    lib = load()
    if lib is None:
        return None
    maj, mnr, rel = C.c_uint(), C.c_uint(), C.c_uint()
    lib.H5get_libversion(C.byref(maj), C.byref(mnr), C.byref(rel))
    return maj.value, mnr.value, rel.value


def _const(lib, name):
    return _hid.in_dll(lib, name).value


def guess_chunk(shape, typesize=8):
    """``h5py``'s automatic chunk shape (h5py 3.x ``_hl/filters.py: guess_chunk``): chunks of about 16 KiB x 2^log10(MiB of
    data), between 8 KiB and 1 MiB, halving the axes in turn."""
    base, lo, hi = 16 * 1024, 8 * 1024, 1024 * 1024
    chunks = np.array([max(int(s), 1) for s in shape], dtype="=f8")
    dset_size = float(np.prod(chunks)) * typesize
    target = base * (2 ** np.log10(dset_size / (1024.0 * 1024.0)))
    target = min(max(target, lo), hi)
    idx = 0
    while True:
        chunk_bytes = float(np.prod(chunks)) * typesize
        if (chunk_bytes < target or abs(chunk_bytes - target) / target < 0.5) and chunk_bytes < hi:
            break
        if np.prod(chunks) == 1:
            break
        chunks[idx % len(chunks)] = np.ceil(chunks[idx % len(chunks)] / 2.0)
        idx += 1
    return tuple(int(x) for x in chunks)


def write_dataset(path, name, data, gzip=True, level=4):
    """``h5py.File(path, 'w').create_dataset(name, data=data, compression='gzip' if gzip else None)``."""
    lib = load()
    if lib is None:
        raise RuntimeError("libhdf5 not found")
    a = np.ascontiguousarray(data, dtype="<f8")
    dims = (_hsize * a.ndim)(*a.shape)
    fid = lib.H5Fcreate(os.fsencode(path), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
    if fid < 0:
        raise OSError(f"H5Fcreate failed for {path}")
    sid = did = pid = -1
    try:
        sid = lib.H5Screate_simple(a.ndim, dims, None)
        pid = lib.H5Pcreate(_const(lib, "H5P_CLS_DATASET_CREATE_ID_g"))
        if gzip and a.size > 0:
            if lib.H5Zfilter_avail(H5Z_FILTER_DEFLATE) <= 0:
                raise RuntimeError("this libhdf5 has no deflate filter")
            ch = (_hsize * a.ndim)(*guess_chunk(a.shape, 8))
            # (h5py also asks for the fill value to be written when a chunk is allocated: H5D_FILL_TIME_ALLOC = 0)
            if lib.H5Pset_chunk(pid, a.ndim, ch) < 0 or lib.H5Pset_deflate(pid, int(level)) < 0 or lib.H5Pset_fill_time(pid, 0) < 0:
                raise OSError("H5Pset_chunk / H5Pset_deflate / H5Pset_fill_time failed")
        did = lib.H5Dcreate2(fid, name.encode(), _const(lib, "H5T_IEEE_F64LE_g"), sid, H5P_DEFAULT, pid, H5P_DEFAULT)
        if did < 0:
            raise OSError(f"H5Dcreate2 failed for {name}")
        if a.size and lib.H5Dwrite(did, _const(lib, "H5T_NATIVE_DOUBLE_g"), H5S_ALL, H5S_ALL, H5P_DEFAULT,
                                   a.ctypes.data_as(C.c_void_p)) < 0:
            raise OSError("H5Dwrite failed")
    finally:
        for closer, h in ((lib.H5Dclose, did), (lib.H5Pclose, pid), (lib.H5Sclose, sid), (lib.H5Fclose, fid)):
            if h >= 0:
                closer(h)
    return path


def read_dataset(path, name):
    """``np.array(h5py.File(path, 'r')[name])`` for a float dataset of any rank."""
    lib = load()
    if lib is None:
        raise RuntimeError("libhdf5 not found")
    fid = lib.H5Fopen(os.fsencode(path), H5F_ACC_RDONLY, H5P_DEFAULT)
    if fid < 0:
        raise OSError(f"{path} is not an HDF5 file this library can open")
    did = sid = -1
    try:
        did = lib.H5Dopen2(fid, name.encode(), H5P_DEFAULT)
        if did < 0:
            raise KeyError(f"{path} has no dataset {name}")
        sid = lib.H5Dget_space(did)
        nd = lib.H5Sget_simple_extent_ndims(sid)
        dims = (_hsize * max(nd, 1))()
        lib.H5Sget_simple_extent_dims(sid, dims, None)
        out = np.empty(tuple(int(dims[i]) for i in range(nd)), dtype=np.float64)
        if out.size and lib.H5Dread(did, _const(lib, "H5T_NATIVE_DOUBLE_g"), H5S_ALL, H5S_ALL, H5P_DEFAULT,
                                    out.ctypes.data_as(C.c_void_p)) < 0:
            raise OSError("H5Dread failed")
        return out
    finally:
        for closer, h in ((lib.H5Sclose, sid), (lib.H5Dclose, did), (lib.H5Fclose, fid)):
            if h >= 0:
                closer(h)
