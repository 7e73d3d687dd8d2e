#!/usr/bin/env python3
"""Diagnostic: the LSTM side of the 8-GPU sync-avoiding leg at its real width on ONE GPU - an interior rank of the
8-slab partition of the 8.2M-tet beam has 3042 shared nodes = 9126 inputs.  Synthetic smooth history of the bench's
shape (30000 steps), training for a bounded time (HIP-graph optimiser step), then the per-window predictor."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synchronization_avoiding_algorithms_amd import predictor as pr  # noqa: E402
from synchronization_avoiding_algorithms_amd import training as tr  # noqa: E402

width = int(sys.argv[1]) if len(sys.argv) > 1 else 9126
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
n_steps, n_p, n_f, n_s = 30000, 20, 20, 150
dev = torch.device("cuda")
t = torch.arange(n_steps, dtype=torch.float64, device=dev)[:, None] * 9.2e-6
j = torch.arange(width, dtype=torch.float64, device=dev)[None, :]
hist = 1e-3 * t.clamp(max=1.0) ** 2 * (1.0 + 0.3 * torch.sin(40.0 * t + 0.01 * j)) * (1.0 + 1e-4 * j)
torch.cuda.synchronize()
t0 = time.time()
mode = sys.argv[3] if len(sys.argv) > 3 else "graph"  # "eager": the optimiser step launched kernel by kernel
model, smax, smin, tl, vl = tr.train_on_history(hist, n_s, n_p, n_f, seed=0, hidden_size=50, max_seconds=seconds,
                                                graph=(mode != "eager"))
torch.cuda.synchronize()
t1 = time.time()
print(f"[{mode}, DEBUG_CLR_GRAPH_PACKET_CAPTURE={os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE', 'unset')}] "
      f"width {width}: {len(tl)} epochs in {t1 - t0:.1f} s ({(t1 - t0) / len(tl) * 1e3:.0f} ms per epoch of 13 batches), "
      f"train MSE {tl[0]:.3e} -> {tl[-1]:.3e}, validation {vl[-1]:.3e}; "
      f"parameters {sum(p.numel() for p in model.parameters())}, peak memory {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB")
predictor = pr.DevicePredictor(model, n_p, n_f, n_s, smax, smin)
start = n_steps - 3 * n_f * n_s
with torch.no_grad():
    for _ in range(3):
        predictor(start, hist)
    torch.cuda.synchronize()
    t0 = time.time()
    for k in range(10):
        table = predictor(start, hist)
    torch.cuda.synchronize()
    per = (time.time() - t0) / 10
truth = hist[start:start + n_f * n_s]
err = float((table - truth).norm() / truth.norm())
print(f"predictor: {per * 1e3:.2f} ms per window of {n_f * n_s} steps ({predictor.backend}), rel-L2 of the predicted window "
      f"against the history it was trained on {err:.3e}")
