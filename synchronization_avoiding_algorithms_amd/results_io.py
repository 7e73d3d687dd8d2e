"""Result artefacts with the reference's names and logical layout.

The reference stores trajectories as HDF5 datasets ``Displacement`` of shape ``(n_dof_local, n_saved)``
(``Data_prepare.py:243-246``, ``Shared_extraction.py:38-40``, ``Online_predictor.py:321-324``).  ``h5py`` is
used when importable; else the HDF5 C library itself through ctypes (:mod:`hdf5_c`: the same container, chunked and
deflated the way ``h5py`` does it for ``compression='gzip'``); only where neither exists does the same array go to
``<name>.npz`` under the same key.  Readers accept whatever is there."""
from __future__ import annotations

import os

import numpy as np

DATASET = "Displacement"


def _h5py():
    try:
        import h5py  # noqa: PLC0415

        return h5py
    except ImportError:
        return None


def save_displacement(path_hdf5: str, data: np.ndarray, compress: bool = True) -> str:
    """Write ``data`` under ``Displacement``; returns the path actually written."""
    os.makedirs(os.path.dirname(path_hdf5) or ".", exist_ok=True)
    h5 = _h5py()
    if h5 is not None:
        with h5.File(path_hdf5, "w") as f:
            f.create_dataset(DATASET, data=data, compression="gzip" if compress else None)
        return path_hdf5
    from . import hdf5_c

    if hdf5_c.available():
        return hdf5_c.write_dataset(path_hdf5, DATASET, data, gzip=compress)
    alt = os.path.splitext(path_hdf5)[0] + ".npz"
    (np.savez_compressed if compress else np.savez)(alt, **{DATASET: data})
    return alt


def load_displacement(path_hdf5: str) -> np.ndarray:
    h5 = _h5py()
    if os.path.exists(path_hdf5) and h5 is not None:
        with h5.File(path_hdf5, "r") as f:
            return np.array(f[DATASET])
    if os.path.exists(path_hdf5):
        from . import hdf5_c

        if hdf5_c.available():
            return hdf5_c.read_dataset(path_hdf5, DATASET)
    alt = os.path.splitext(path_hdf5)[0] + ".npz"
    if os.path.exists(alt):
        with np.load(alt, allow_pickle=False) as z:
            return z[DATASET]
    raise FileNotFoundError(f"neither {path_hdf5} (needs h5py or libhdf5) nor {alt} exists")


def save_int_list(path: str, values) -> None:
    """``np.savetxt(..., fmt='%d')`` like ``Data_prepare.py:116-118,124``."""
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    np.savetxt(path, np.asarray(values, dtype=np.int64), delimiter=",", fmt="%d")


def load_int_list(path: str) -> np.ndarray:
    return np.atleast_1d(np.genfromtxt(path, delimiter=",")).astype(np.int64)
