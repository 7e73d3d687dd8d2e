#!/opt/conda/bin/python3.9
"""Golden `Displacement` containers written by h5py itself (h5py 3.3.0 / HDF5 1.10.6 in the image's Anaconda interpreter -
the interpreter of this repository has no h5py), exactly as the reference writes them:

    Data_prepare.py:243-246     hf.create_dataset('Displacement', data=d1_save, compression='gzip')
    Shared_extraction.py:38-40  hf.create_dataset('Displacement', data=d)

    /opt/conda/bin/python3.9 tests/golden/make_golden_hdf5.py

Writes tests/golden/h5py_gzip.hdf5 (a (24, 400) trajectory, 53 KB) and h5py_plain.hdf5 ((6, 400)); the data are
`numpy.random.default_rng(7).normal(size=(24, 400)) * 1e-3` and its rows 3..8."""
import os

import h5py
import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
a = np.random.default_rng(7).normal(size=(24, 400)) * 1e-3
with h5py.File(os.path.join(here, "h5py_gzip.hdf5"), "w") as hf:
    hf.create_dataset("Displacement", data=a, compression="gzip")
with h5py.File(os.path.join(here, "h5py_plain.hdf5"), "w") as hf:
    hf.create_dataset("Displacement", data=a[3:9])
print(h5py.__version__, h5py.version.hdf5_version, h5py.File(os.path.join(here, "h5py_gzip.hdf5"), "r")["Displacement"].chunks)
