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
captures and replays the known trigger on the device and compares with eager execution; when that canary fails (the switch
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
    cached per device).  Three replays on fresh inputs, each compared bit for bit with eager execution."""
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
            gen = torch.Generator(device=dev).manual_seed(11)
            X = torch.rand(10, 20, 1000, device=dev, generator=gen)
            Y = torch.rand(10, 20, 1000, device=dev, generator=gen)
            out = torch.zeros(3, dtype=torch.float64, device=dev)

            def body():
                loss = torch.nn.functional.mse_loss(X * 0.5 + 0.125, Y)            # multi-block reduction
                out[0] = loss.double()
                out[1] = (1.0 - loss / torch.nn.functional.mse_loss(Y, torch.mean(Y) + torch.zeros_like(Y))).double()
                out[2] = (X.double().sum() - Y.double().sum())

            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                body()
            torch.cuda.current_stream(dev).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                body()
            for _ in range(3):
                X.copy_(torch.rand(X.shape, device=dev, generator=gen))
                Y.copy_(torch.rand(Y.shape, device=dev, generator=gen))
                graph.replay()
                got = out.clone()
                body()
                torch.cuda.synchronize(dev)
                ok = ok and bool(torch.equal(got, out))
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
