#!/usr/bin/env python3
"""Diagnostic: one prediction window (DNN_prediction.py:38-55, all 150 phases) on the library's own kernels against the
PyTorch-ROCm path replayed as a HIP graph - milliseconds per window, and the largest difference between the two tables.

    python tools/predictor_point.py [input_size ...]      (default: 24 9126)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from synchronization_avoiding_algorithms_amd import predictor as pr  # noqa: E402

native_only = "--native-only" in sys.argv  # (for counter passes: the PyTorch path is ~600 launches per window)
sizes = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [24, 9126]
n_p = n_f = 20
n_s, H = 150, 50
for I in sizes:
    torch.manual_seed(1)
    model = pr.LSTM_encoder_decoder(I, H).cuda().eval()
    hist = torch.cumsum(torch.randn(2 * n_p * n_s, I, device="cuda", dtype=torch.float64) * 1e-4, 0)
    smax, smin = float(hist.max()) * 1.05, float(hist.min()) * 1.05
    t0 = time.perf_counter()
    nat = pr.NativePredictor(model, n_p, n_f, n_s)
    t_create = time.perf_counter() - t0
    n = n_p * n_s + 17
    got = nat.predict(n, hist, smax, smin).clone()
    if native_only:
        for _ in range(5):
            nat.predict(n, hist, smax, smin, got)
        torch.cuda.synchronize()
        print(f"input_size {I:5d}: 6 windows on the native kernels", flush=True)
        nat.close()
        continue
    graph = pr.DevicePredictor(model, n_p, n_f, n_s, smax, smin, backend="torch")
    with torch.no_grad():
        for _ in range(4):
            want = graph(n, hist).clone()
    err = float((got - want).abs().max() / want.abs().max())
    # both against the same model evaluated in fp64 (weights widened, history not rounded to fp32)
    import copy

    m64 = copy.deepcopy(model).double()
    with torch.no_grad():
        past, fut = pr._phase_indices(n, n_p, n_f, n_s)
        import numpy as np

        X = pr.scale_forward(hist[torch.as_tensor(np.stack(past), device="cuda")], smax, smin)
        Y = pr.scale_it_back(pr.model_predict("cuda", m64, X, n_f), smax, smin)
        ref = torch.zeros_like(want)
        ref[torch.as_tensor(np.stack(fut), device="cuda").reshape(-1)] = Y.reshape(-1, I)
    e_nat = float((got - ref).abs().max() / ref.abs().max())
    e_pt = float((want - ref).abs().max() / ref.abs().max())
    print(f"input_size {I:5d}: against the fp64 evaluation of the model: native {e_nat:.2e}, PyTorch-ROCm fp32 {e_pt:.2e} "
          "(largest difference / range)", flush=True)

    def ms(fn, reps=50):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps

    with torch.no_grad():
        t_graph = ms(lambda: graph(n, hist))
    t_nat = ms(lambda: nat.predict(n, hist, smax, smin, got))
    print(f"input_size {I:5d}: native {t_nat:.3f} ms per window, PyTorch-ROCm (HIP graph) {t_graph:.3f} ms, "
          f"max difference {err:.2e} of the range; model set-up {t_create:.2f} s", flush=True)
    nat.close()
