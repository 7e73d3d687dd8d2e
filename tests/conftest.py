import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def beam_coarse():
    from synchronization_avoiding_algorithms_amd.mesh import Mesh

    g = load_golden("beam_coarse_mesh.npz")
    return Mesh(g["points"], {"tetra": g["tetra"], "triangle": g["triangle"]})


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    nb = np.linalg.norm(b)
    return np.linalg.norm(a - b) / nb if nb > 0 else np.linalg.norm(a)


def free_port():
    """A TCP port nobody listens on right now (for the rendezvous of spawned ranks)."""
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def oracle_hybrid_tworank(resync_every=None, resync_steps=None):
    """The oracle's re-enactment of Online_predictor.py:251-318 on the reference's own two-rank case
    (golden/hybrid_tworank.npz: mesh partition, per-rank LSTM weights and scaling constants), optionally with the
    re-synchronisation extension.  Returns (saved trajectories, shared-dof histories), one entry per rank."""
    import torch

    from oracle import fem_oracle as fo
    from oracle import lstm_oracle as lo

    g = load_golden("hybrid_tworank.npz")
    m = load_golden("beam_coarse_mesh.npz")
    T, n_p, n_f, n_s, hid = (int(g[k]) for k in ("test_num", "n_past", "n_future", "filter_size", "hidden_size"))
    ranks, dt, shared, _ = fo.setup_problem(m["points"], m["tetra"], m["triangle"], 2, g["epart"])
    loc = [fo.node_to_dof(fo.local_index(shared[r], ranks[r].nodes)) for r in range(2)]
    models = [lo.load_model(len(loc[r]), hid, {k[len(f"r{r}_w::"):]: torch.from_numpy(v) for k, v in g.items()
                                               if k.startswith(f"r{r}_w::")}) for r in range(2)]
    scales = [tuple(float(v) for v in g[f"r{r}_scale"]) for r in range(2)]

    def predictor(r, n, hist):
        return lo.predictor_table(n, models[r], n_p, n_f, n_s, len(loc[r]), hist, *scales[r])

    return fo.run_hybrid(ranks, dt, T, loc, predictor, n_p, n_f, n_s, resync_every=resync_every,
                         resync_steps=resync_steps)
