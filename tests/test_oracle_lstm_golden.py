"""Pin the LSTM / hybrid-loop oracle to vectors produced by the reference's own
``encoder_decoder_predictor`` and ``Online_predictor.py`` loop (tests/golden/make_golden.py)."""
import numpy as np
import torch

from conftest import load_golden, rel_l2
from oracle import fem_oracle as fo
from oracle import lstm_oracle as lo


def _weights(g, prefix):
    return {k[len(prefix):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(prefix)}


def test_predictor_table():
    g = load_golden("predictor_table.npz")
    assert bool(g["NF_is_fp32_exact"])
    model = lo.load_model(int(g["input_size"]), int(g["hidden_size"]), _weights(g, "w::"))
    smax, smin = (float(v) for v in g["scale"])
    NF = lo.predictor_table(int(g["n"]), model, int(g["n_p"]), int(g["n_f"]), int(g["n_s"]),
                            int(g["input_size"]), g["d_sol"], smax, smin)
    assert NF.shape == g["NF"].shape
    assert np.abs(NF - g["NF"]).max() <= 1e-6 * np.abs(g["NF"]).max()


def test_state_dict_layout():
    """Key names/shapes of DNN_tools.py:85-98."""
    g = load_golden("predictor_table.npz")
    w = _weights(g, "w::")
    H, n_in = int(g["hidden_size"]), int(g["input_size"])
    assert w["encoder.lstm_encoder.weight_ih_l0"].shape == (4 * H, n_in)
    assert w["encoder.lstm_encoder.weight_ih_l1_reverse"].shape == (4 * H, 2 * H)
    assert w["decoder.lstm_decoder.weight_ih_l0"].shape == (8 * H, n_in)
    assert w["decoder.lstm_decoder.weight_hh_l0"].shape == (8 * H, 2 * H)
    assert w["decoder.fc.weight"].shape == (n_in, 2 * H)
    assert len(w) == 16 + 4 + 2


def test_scaling_constants_and_hybrid_loop(beam_coarse):
    g = load_golden("hybrid_tworank.npz")
    T, n_p, n_f, n_s = (int(g[k]) for k in ("test_num", "n_past", "n_future", "filter_size"))
    hid = int(g["hidden_size"])
    ranks, dt, shared, _ = fo.setup_problem(beam_coarse.points, beam_coarse.tets, beam_coarse.triangles,
                                            2, g["epart"])
    assert dt == float(g["dt"])
    models, scales, loc = [], [], []
    for r in range(2):
        sd = fo.node_to_dof(fo.local_index(shared[r], ranks[r].nodes))
        assert np.array_equal(sd, g[f"r{r}_shared_dof"]) and np.array_equal(sd, g[f"r{r}_loc_dof_shared"])
        loc.append(sd)
        smax, smin = lo.scaling_constants(g[f"r{r}_shared_traj"], n_s, n_p, n_f, float(g["cut_off"]))
        assert (smax, smin) == tuple(float(v) for v in g[f"r{r}_scale"])
        scales.append((smax, smin))
        models.append(lo.load_model(len(sd), hid, _weights(g, f"r{r}_w::")))

    # ground truth part: the Shared_extraction.py output is rows shared_dof of the trajectory
    _, _, _, snaps = fo.run_ground_truth(ranks, dt, T, snapshots=(T,))
    for r in range(2):
        assert rel_l2(snaps[T][r], g[f"r{r}_truth_last"]) < 1e-13
        assert rel_l2(snaps[T][r][loc[r], 0], g[f"r{r}_shared_traj"][:, -1]) < 1e-13

    def predictor(r, n, hist):
        return lo.predictor_table(n, models[r], n_p, n_f, n_s, len(loc[r]), hist, *scales[r])

    save, hist = fo.run_hybrid(ranks, dt, T, loc, predictor, n_p, n_f, n_s)
    for r in range(2):
        ref = g[f"r{r}_modeled"]
        assert save[r].shape == ref.shape
        # the LSTM runs in fp32, so agreement is to fp32 round-off of the predictions
        assert rel_l2(save[r], ref) < 1e-6, rel_l2(save[r], ref)
        assert rel_l2(hist[r], g[f"r{r}_d_sol_shared"]) < 1e-6
        # synchronised warm-up part is fp64-exact
        i_cri = n_p * n_s - 1
        assert rel_l2(save[r][:, :i_cri + 1], ref[:, :i_cri + 1]) < 1e-13
