"""Product LSTM predictor (batched) against the reference-generated goldens, on CPU."""
import numpy as np
import torch

from conftest import load_golden
from synchronization_avoiding_algorithms_amd import predictor as pr


def _weights(g, prefix):
    return {k[len(prefix):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(prefix)}


def _table_against_reference(tmp_path, device, tol):
    g = load_golden("predictor_table.npz")
    n_in, hid = int(g["input_size"]), int(g["hidden_size"])
    path = tmp_path / "model.pth"
    torch.save(_weights(g, "w::"), path)  # same format as Model_training.py:179-180
    model = pr.call_model(device, int(g["n_s"]), n_in, hid, str(path))
    smax, smin = (float(v) for v in g["scale"])
    NF = pr.encoder_decoder_predictor(device, int(g["n"]), model, int(g["n_p"]), int(g["n_f"]), int(g["n_s"]),
                                      n_in, g["d_sol"], smax, smin)
    assert NF.dtype == np.float64 and NF.shape == g["NF"].shape
    # fp32 model: batched vs the reference's batch-1 passes differ by fp32 round-off only
    assert np.abs(NF - g["NF"]).max() <= tol * np.abs(g["NF"]).max()
    assert np.array_equal(NF, NF.astype(np.float32).astype(np.float64))  # fp32 values widened (:54)
    return g, model, (smax, smin)


def test_reference_state_dict_loads_and_table_matches(tmp_path):
    _table_against_reference(tmp_path, "cpu", 2e-5)


def test_scaling_constants_match_reference():
    g = load_golden("hybrid_tworank.npz")
    for r in range(2):
        smax, smin = pr.scaling_constants(g[f"r{r}_shared_traj"], int(g["filter_size"]), int(g["n_past"]),
                                          int(g["n_future"]), float(g["cut_off"]))
        assert (smax, smin) == tuple(float(v) for v in g[f"r{r}_scale"])


def test_model_predict_single_equals_batched():
    torch.manual_seed(0)
    model = pr.LSTM_encoder_decoder(9, 7)
    X = torch.randn(5, 6, 9)
    one = torch.stack([pr.model_predict("cpu", model, X[i], 4) for i in range(5)])
    assert torch.allclose(one, pr.model_predict("cpu", model, X, 4), atol=1e-6)


import pytest  # noqa: E402


@pytest.mark.gpu
def test_device_predictor_graph_replay_equals_eager_calls():
    """The PyTorch-ROCm route of DevicePredictor (backend="torch"; the default on a GPU is the library's own
    kernels, tests/test_gpu_predictor.py) replays the per-window prediction as a HIP graph after two eager calls; the
    window position is a device scalar, so one capture serves every window."""
    import torch

    from synchronization_avoiding_algorithms_amd import predictor as pr

    torch.manual_seed(3)
    dev = torch.device("cuda")
    n_p, n_f, n_s, insz = 4, 3, 10, 18
    model = pr.LSTM_encoder_decoder(insz, 8).to(dev).eval()
    hist = torch.randn(400, insz, dtype=torch.float64, device=dev) * 1e-3
    p = pr.DevicePredictor(model, n_p, n_f, n_s, 2e-3, -2e-3, backend="torch")
    assert p.backend == "PyTorch-ROCm, HIP graph"
    with torch.no_grad():
        for n in (40, 70, 100, 133, 260, 41):
            got = p(n, hist).clone()
            want = pr.predict_table(model, n, n_p, n_f, n_s, hist, 2e-3, -2e-3)
            assert torch.allclose(got, want, rtol=1e-6, atol=1e-12), n
        assert p._graph is not None  # the later calls were replays
        other = hist.clone()         # a different history tensor: captured again, still right
        for n in (50, 90, 120, 200):
            assert torch.allclose(p(n, other), pr.predict_table(model, n, n_p, n_f, n_s, other, 2e-3, -2e-3),
                                  rtol=1e-6, atol=1e-12)


@pytest.mark.gpu
def test_reference_shaped_table_on_the_gpu(tmp_path):
    """The fixture at the reference's real shape (24 inputs, H = 50, n_p = n_f = 20) through MIOpen / rocBLAS on the
    GPU, eager and as the replayed HIP graph of DevicePredictor's PyTorch-ROCm route (fp32: 1e-4 of the table's range)."""
    g, model, (smax, smin) = _table_against_reference(tmp_path, "cuda", 1e-4)
    dev = pr.DevicePredictor(model, int(g["n_p"]), int(g["n_f"]), int(g["n_s"]), smax, smin, backend="torch")
    hist = torch.from_numpy(g["d_sol"]).to("cuda")
    for _ in range(4):  # eager calls, capture, replay
        table = dev(int(g["n"]), hist)
    torch.cuda.synchronize()
    assert np.abs(table.cpu().numpy() - g["NF"]).max() <= 1e-4 * np.abs(g["NF"]).max()
