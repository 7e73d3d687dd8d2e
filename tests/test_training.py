"""LSTM training counterpart of Model_training.py: windowing/scaling equal the inference-side constants,
the loss goes down, and the saved state_dict feeds call_model / the predictor (CPU, tiny)."""
import os

import numpy as np
import torch

from synchronization_avoiding_algorithms_amd import drivers, predictor as pr, results_io as rio, training as tr


def _fake_shared_trajectory(n_in=6, n_steps=1200):
    t = np.arange(n_steps)[None, :]
    j = np.arange(n_in)[:, None]
    return 1e-2 * np.sin(0.01 * t + 0.5 * j) * (1 + 0.05 * j)


def test_windowing_and_scaling_match_inference_side():
    traj = _fake_shared_trajectory()
    X, Y = tr.windowed_dataset(traj, 10, 4, 3, 0.5)
    assert X.shape == (54, 4, 6) and Y.shape == (54, 3, 6) and X.dtype == torch.float32
    # window k: inputs = filtered samples k..k+3, targets k+4..k+6 (DNN_tools.py:303-307)
    filt = traj.T[:600][::10].astype(np.float32)
    assert np.array_equal(X[5].numpy(), filt[5:9]) and np.array_equal(Y[5].numpy(), filt[9:12])
    Xs, Ys, smax, smin = tr.scale_to_zero_one(X, Y)
    assert (smax, smin) == pr.scaling_constants(traj, 10, 4, 3, 0.5)
    assert float(Xs.max()) <= 0.0 and float(min(Xs.min(), Ys.min())) == -1.0


def test_training_reduces_loss_and_writes_reference_layout(tmp_path):
    out = str(tmp_path)
    traj = _fake_shared_trajectory()
    rio.save_int_list(os.path.join(out, drivers.PATHS["shared"].format(r=0)), [3, 9])
    rio.save_displacement(os.path.join(out, drivers.PATHS["shared_traj"].format(r=0)), traj, compress=False)
    path, train_loss, test_loss = tr.train_rank_model(out, 0, device="cpu", hidden_size=8, filter_size=10, n_past=4,
                                                      n_future=3, num_epochs=40, learning_rate=5e-3, seed=0)
    assert path.endswith("Distributed_save/Rank-0/nB-10-nH-8-Lr-0.005-filter=10/model.pth")
    assert train_loss[-1] < 0.2 * train_loss[0] and np.isfinite(test_loss).all()
    model = pr.call_model("cpu", 10, 6, 8, path)
    hist = torch.from_numpy(traj.T.copy())
    smax, smin = pr.scaling_constants(traj, 10, 4, 3, 0.5)
    table = pr.predict_table(model, 400, 4, 3, 10, hist, smax, smin)
    assert table.shape == (30, 6) and torch.isfinite(table).all()
    # a (briefly) trained model tracks the smooth signal far better than the mean would
    err = (table - hist[400:430]).abs().max().item()
    assert err < 0.5 * (smax - smin)


import pytest  # noqa: E402


@pytest.mark.gpu
def test_graph_captured_training_step_matches_eager_training(tmp_path, monkeypatch):
    """On the GPU the optimiser step is replayed as a HIP graph (GraphedTrainStep): same seeds, same batches, same
    learning-rate schedule as eager training -> the same losses up to fp32 run-to-run noise."""
    traj = _fake_shared_trajectory(n_in=12, n_steps=4000)
    losses, paths = [], []
    for mode in ("0", "1"):
        out = str(tmp_path / f"graph{mode}")
        rio.save_int_list(os.path.join(out, drivers.PATHS["shared"].format(r=0)), [3, 9, 11, 17])
        rio.save_displacement(os.path.join(out, drivers.PATHS["shared_traj"].format(r=0)), traj, compress=False)
        monkeypatch.setenv("SAA_TRAIN_GRAPH", mode)
        path, train_loss, test_loss = tr.train_rank_model(out, 0, device="cuda", hidden_size=8, filter_size=10, n_past=4,
                                                          n_future=3, num_epochs=12, learning_rate=5e-3, seed=0)
        losses.append((np.array(train_loss), np.array(test_loss)))
        paths.append(path)
    (tl0, vl0), (tl1, vl1) = losses
    assert tl1[-1] < 0.5 * tl1[0]
    assert np.allclose(tl1, tl0, rtol=2e-2) and np.allclose(vl1, vl0, rtol=2e-2)
    a, b = torch.load(paths[0], weights_only=True), torch.load(paths[1], weights_only=True)
    assert a.keys() == b.keys()
    for k in a:
        assert torch.allclose(a[k], b[k], rtol=0, atol=5e-3), k
