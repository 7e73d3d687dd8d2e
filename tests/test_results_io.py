"""The `Displacement` container (Data_prepare.py:243-246, Shared_extraction.py:22-40, Online_predictor.py:321-324): real
HDF5 - through h5py where it is importable, else through the HDF5 C library itself (hdf5_c.py) - with the layout h5py gives
`create_dataset('Displacement', data=..., compression='gzip')`; `.npz` only where neither exists."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from synchronization_avoiding_algorithms_amd import hdf5_c
from synchronization_avoiding_algorithms_amd import results_io as rio


def _h5dump():
    for cand in (shutil.which("h5dump"), "/opt/conda/bin/h5dump"):
        if cand and os.path.exists(cand):
            return cand
    return None


@pytest.mark.skipif(not hdf5_c.available(), reason="no libhdf5 >= 1.10 on this host")
def test_displacement_files_are_hdf5_with_h5pys_layout(tmp_path):
    rng = np.random.default_rng(0)
    traj = rng.normal(size=(330, 1000)) * 1e-3                     # (n_dof_local, n_saved) like d1_save
    path = rio.save_displacement(str(tmp_path / "Local-rank-0.hdf5"), traj)
    assert path.endswith(".hdf5") and open(path, "rb").read(8) == b"\x89HDF\r\n\x1a\n"   # the HDF5 signature
    assert np.array_equal(rio.load_displacement(path), traj)       # bit for bit
    shared = traj[[3, 4, 5, 30, 31, 32]]                            # Shared_extraction.py:36-40: no compression
    p2 = rio.save_displacement(str(tmp_path / "rank=0-shared_dof.hdf5"), shared, compress=False)
    assert np.array_equal(rio.load_displacement(p2), shared)
    dump = _h5dump()
    if dump:  # the library's own tool on what was written: name, type, shape, chunked + deflate level 4 / contiguous
        out = subprocess.run([dump, "-H", "-p", path], capture_output=True, text=True).stdout
        assert 'DATASET "Displacement"' in out and "H5T_IEEE_F64LE" in out and "( 330, 1000 )" in out
        assert "CHUNKED ( 21, 125 )" in out and "DEFLATE { LEVEL 4 }" in out
        out2 = subprocess.run([dump, "-H", "-p", p2], capture_output=True, text=True).stdout
        assert "CONTIGUOUS" in out2 and "( 6, 1000 )" in out2 and "NONE" in out2


def test_chunk_shapes_follow_h5pys_rule():
    """h5py's automatic chunking (filters.guess_chunk): ~16 KiB x 2^log10(MiB), clamped to [8 KiB, 1 MiB], axes halved in turn."""
    for shape in ((330, 1000), (330, 100000), (24, 5), (571200, 300), (1, 1)):
        ch = hdf5_c.guess_chunk(shape)
        assert len(ch) == len(shape) and all(1 <= c <= s for c, s in zip(ch, shape))
        assert np.prod(ch) * 8 <= 1024 * 1024
    assert hdf5_c.guess_chunk((330, 1000)) == (21, 125) and hdf5_c.guess_chunk((24, 5)) == (24, 5)


def test_readers_accept_the_npz_stand_in(tmp_path):
    """Files written by earlier rounds (no HDF5 library found: `<name>.npz` under the same key) still load."""
    a = np.arange(12.0).reshape(3, 4)
    np.savez_compressed(tmp_path / "Local-rank-1.npz", Displacement=a)
    assert np.array_equal(rio.load_displacement(str(tmp_path / "Local-rank-1.hdf5")), a)


def _mask_mtime(raw):
    """The object-modification-time message (type 0x0012, 8 bytes: version 1, 3 reserved, 4 bytes of seconds) is the only
    thing that differs between two writes of the same data: its seconds are zeroed."""
    raw = bytearray(raw)
    key = bytes([0x12, 0x00, 0x08, 0x00, 0x00, 0x00, 0x00, 0x00, 0x01, 0x00, 0x00, 0x00])
    pos = raw.find(key)
    assert pos > 0 and raw.find(key, pos + 1) < 0  # exactly one such message (one dataset)
    raw[pos + len(key):pos + len(key) + 4] = b"\0\0\0\0"
    return bytes(raw)


@pytest.mark.skipif(not hdf5_c.available(), reason="no libhdf5 >= 1.10 on this host")
def test_files_equal_the_ones_h5py_wrote(tmp_path):
    """tests/golden/h5py_*.hdf5 were written by h5py 3.3 itself with the reference's two calls (make_golden_hdf5.py): this
    repository's writer produces the same bytes (but for the time stamp) and its reader returns the same array."""
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    a = np.random.default_rng(7).normal(size=(24, 400)) * 1e-3
    for name, data, gz in (("h5py_gzip.hdf5", a, True), ("h5py_plain.hdf5", a[3:9], False)):
        gold = os.path.join(here, name)
        assert np.array_equal(rio.load_displacement(gold), data)
        mine = rio.save_displacement(str(tmp_path / name), data, compress=gz)
        x, y = open(gold, "rb").read(), open(mine, "rb").read()
        if hdf5_c.library_version()[:2] == (1, 10):  # (another library version lays the file out differently)
            assert len(x) == len(y) and _mask_mtime(x) == _mask_mtime(y), name


def test_without_any_hdf5_library_the_npz_stand_in_is_written(tmp_path, monkeypatch):
    """No h5py and no libhdf5 (another host): the same array under the same key in `<name>.npz`, and the path says so."""
    monkeypatch.setattr(hdf5_c, "available", lambda: False)
    monkeypatch.setattr(rio, "_h5py", lambda: None)
    a = np.arange(24.0).reshape(6, 4)
    path = rio.save_displacement(str(tmp_path / "Modeled_Local-rank-0.hdf5"), a)
    assert path.endswith(".npz") and not os.path.exists(tmp_path / "Modeled_Local-rank-0.hdf5")
    assert np.array_equal(rio.load_displacement(str(tmp_path / "Modeled_Local-rank-0.hdf5")), a)
