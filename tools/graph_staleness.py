#!/usr/bin/env python3
"""Diagnostic: does a captured PyTorch HIP graph still return what it returned, after many other kernel launches?
(ROCm 7.2 / PyTorch 2.10: the validation pass of the training loop, captured as a graph, came back with corrupted sums
after ~10^4 launches - DESIGN section 7.)  Here: DevicePredictor's PyTorch-ROCm route (SAA_PREDICT_NATIVE=0), whose
window table is a graph replay."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SAA_PREDICT_NATIVE"] = "0"
import torch  # noqa: E402

from synchronization_avoiding_algorithms_amd import predictor as pr  # noqa: E402

torch.manual_seed(1)
I, H, n_p, n_f, n_s = 600, 50, 20, 20, 150
model = pr.LSTM_encoder_decoder(I, H).cuda().eval()
hist = torch.cumsum(torch.randn(2 * n_p * n_s, I, device="cuda", dtype=torch.float64) * 1e-4, 0)
smax, smin = float(hist.max()) * 1.05, float(hist.min()) * 1.05
p = pr.DevicePredictor(model, n_p, n_f, n_s, smax, smin)
n = n_p * n_s + 17
with torch.no_grad():
    want = pr.predict_table(model, n, n_p, n_f, n_s, hist, smax, smin)
    for _ in range(4):
        got = p(n, hist).clone()
    assert p._graph is not None
    print(f"right after capture: max difference graph - eager {float((got - want).abs().max()):.3e}", flush=True)
    z = torch.zeros(1000, device="cuda")
    done = 0
    for more in (5000, 5000, 10000, 20000, 60000):
        for _ in range(more):
            z.add_(1.0)
        done += more
        got = p(n, hist).clone()
        print(f"after {done} other launches: max difference graph - eager {float((got - want).abs().max()):.3e}", flush=True)
