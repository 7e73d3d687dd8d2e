"""Captured PyTorch HIP graphs on this ROCm: what makes their replays wrong, the switch that avoids it, and a canary.

Finding (ROCm 7.2 runtime under PyTorch 2.10, MI355X; ``tools/graph_staleness.py``, ``profiles/r04_graph_staleness.txt``,
one changed variable per run): a captured graph that contains one of ATen's MULTI-BLOCK reductions (``mean`` / ``sum`` /
``mse_loss`` over more elements than one workgroup reduces: partial results in a global staging buffer, a semaphore, the
last workgroup finishes) returns wrong values from its SECOND replay on when the HIP runtime replays graphs from
pre-recorded AQL packets (``DEBUG_CLR_GRAPH_PACKET_CAPTURE``, on by default).  The first replay is right; single-block
reductions, elementwise kernels, GEMMs and this library's own kernels are not affected; kernel-argument pool size, blit
options, the stream of the replay and the kind or number of launches in between change nothing; with
``DEBUG_CLR_GRAPH_PACKET_CAPTURE=0`` (the runtime enqueues the graph's nodes through its ordinary dispatch path) 60 replays
over 30 000 intervening launches stay bit-equal to eager execution, at the same replay speed (16 ms per training epoch at
9126 inputs either way, 30 ms eager).  That is what corrupted round 2's graph-replayed validation sums and the loss value
inside the training-step graph (the reduction of ``mse_loss``), while the weights - whose gradients need no such
reduction - stayed right.

So the package switches packet capture off before the HIP runtime starts (:func:`configure_runtime`, called when the
package is imported), and every user of a torch-captured graph first asks :func:`replays_are_trustworthy`, which
captures the structure that showed the fault, replays it four times and compares with eager execution; when that canary fails (the switch
came too late because the process had initialised HIP already, or another runtime version misbehaves differently) the
caller runs eagerly.
"""
from __future__ import annotations

import os
import warnings

_SWITCH = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"
_verdict = {}


def configure_runtime():
    """Ask the HIP runtime not to replay graphs from pre-recorded packets.  Effective only if no HIP call has been made in
    this process yet (the runtime reads its switches when it initialises); harmless otherwise - the canary decides."""
    os.environ.setdefault(_SWITCH, "0")


def replays_are_trustworthy(device=None) -> bool:
    """True iff a captured graph holding multi-block reductions replays correctly on ``device`` in THIS process (result
    cached per device).  Four replays of the known trigger, each compared bit for bit with eager execution."""
    import torch

    if not torch.cuda.is_available():
        return False
    dev = torch.device("cuda", torch.cuda.current_device() if device is None else
                       (device.index if isinstance(device, torch.device) else int(device)))
    if dev.index in _verdict:
        return _verdict[dev.index]
    ok = True
    try:
        with torch.no_grad():
            # the structure that showed the fault (round 2's validation pass without its model): five batches, per batch
            # three mean-square reductions over 10 x 20 x 1000 values and three in-place additions into fp64 running sums
            gen = torch.Generator(device=dev).manual_seed(11)
            batches = [(torch.rand(b, 20, 1000, device=dev, generator=gen) - 1.0,
                        torch.rand(b, 20, 1000, device=dev, generator=gen) - 1.0) for b in (10, 10, 10, 10, 1)]
            sums = torch.zeros(3, dtype=torch.float64, device=dev)
            mse = torch.nn.functional.mse_loss

            def body():
                sums.zero_()
                for X, Y in batches:
                    loss = mse(X * 0.5 + 0.125, Y)
                    sums[0] += loss.double()
                    sums[1] += (1.0 - loss / mse(Y, torch.mean(Y) + torch.zeros_like(Y))).double()
                    sums[2] += (1.0 - loss / mse(Y, torch.zeros_like(Y))).double()

            body()
            want = sums.clone()
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                body()
            torch.cuda.current_stream(dev).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                body()
            z = torch.zeros(1000, device=dev)
            for _ in range(4):  # (the first replay is right either way; the second was wrong in every failing run)
                for _ in range(20):
                    z.add_(1.0)
                graph.replay()
                torch.cuda.synchronize(dev)
                ok = ok and bool(torch.equal(sums, want))
            del graph
    except Exception as exc:  # noqa: BLE001 - a runtime that cannot capture at all is not trustworthy either
        warnings.warn(f"HIP graph canary could not run ({exc!r}): torch-captured graphs stay off")
        ok = False
    if not ok:
        warnings.warn("HIP graph canary failed: a captured graph with multi-block reductions replays wrongly in this process "
                      f"({_SWITCH}={os.environ.get(_SWITCH, 'unset')}; it must be 0 before the first HIP call) - "
                      "torch-captured graphs stay off, the steps run eagerly")
    _verdict[dev.index] = ok
    return ok
