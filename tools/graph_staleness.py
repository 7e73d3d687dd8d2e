#!/usr/bin/env python3
"""Diagnostic: what goes wrong in a captured PyTorch HIP graph after ~10^4 other kernel launches (ROCm 7.2 / PyTorch 2.10;
DESIGN.md section 7, profiles/r03_graph_staleness.txt case 2), one changed variable per run.

    python tools/graph_staleness.py [--pass full|mse|scalar] [--filler add|big|copy|graph] [--zero memset|mul|none]
                                    [--replay-stream same|side] [--step 500] [--max 30000]

The captured pass is round 2's validation loop (`model_test` over five fixed batches, three fp64 running sums formed inside
the graph).  Between replays `--step` filler kernels are launched eagerly; the first replay whose sums differ from the eager
pass is reported together with EVERY intermediate of the pass that the graph holds (per batch: decoded output checksum, loss,
the two denominators, the three fp64 addends), so that the first node that goes wrong is named.  HIP runtime switches are
taken from the environment of the run (HSA_KERNARG_POOL_SIZE, DEBUG_CLR_GRAPH_PACKET_CAPTURE, ...): the caller changes one.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

from synchronization_avoiding_algorithms_amd import predictor as pr  # noqa: E402
from synchronization_avoiding_algorithms_amd import training as tr  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--pass", dest="what", default="full", choices=["full", "mse", "scalar"])
ap.add_argument("--filler", default="add", choices=["add", "big", "copy", "graph"])
ap.add_argument("--zero", default="memset", choices=["memset", "mul", "none"])
ap.add_argument("--replay-stream", default="same", choices=["same", "side"])
ap.add_argument("--step", type=int, default=500)
ap.add_argument("--max", type=int, default=30000)
ap.add_argument("--width", type=int, default=1000)
args = ap.parse_args()

dev = torch.device("cuda", 0)
torch.manual_seed(3)
I, H, n_p, n_f = args.width, 50, 20, 20
model = pr.LSTM_encoder_decoder(I, H, 2, True, 0.0, 0.0).to(dev).eval()
crit = nn.MSELoss()
batches = [(torch.rand(b, n_p, I, device=dev) - 1.0, torch.rand(b, n_f, I, device=dev) - 1.0) for b in (10, 10, 10, 10, 1)]
sums = torch.zeros(3, dtype=torch.float64, device=dev)
keep = {}  # name -> tensor that lives in the graph's pool: the intermediates of the captured pass


def the_pass(record):
    if args.zero == "memset":
        sums.zero_()
    elif args.zero == "mul":
        sums.mul_(0.0)
    for k, (X, Y) in enumerate(batches):
        if args.what == "full":
            out = tr._decode(model, X, n_f)
        elif args.what == "mse":
            out = X * 0.5 + 0.125
        else:
            out = None
        if out is not None:
            loss = crit(out, Y)
            d1 = crit(Y, torch.mean(Y) + torch.zeros_like(Y))
            d2 = crit(Y, torch.zeros_like(Y))
        else:  # no reduction at all: scalars that are already there
            loss, d1, d2 = X[0, 0, 0] * 1.0, Y[0, 0, 0] * 1.0, Y[0, 0, 1] * 1.0
        a0, a1, a2 = loss.double(), (1.0 - loss / d1).double(), (1.0 - loss / d2).double()
        sums[0] += a0
        sums[1] += a1
        sums[2] += a2
        if record is not None:
            if out is not None:
                record[f"b{k}.out_sum"] = out.double().sum()
            for name, t in (("loss", loss), ("den_r2", d1), ("den_rel", d2), ("add0", a0), ("add1", a1), ("add2", a2)):
                record[f"b{k}.{name}"] = t


with torch.no_grad():
    eager = {}
    the_pass(eager)
    want = sums.clone()
    eager = {k: v.clone() for k, v in eager.items()}
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            the_pass(None)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        the_pass(keep)
    graph.replay()
    torch.cuda.synchronize()
    print(f"variant: pass={args.what} filler={args.filler} zero={args.zero} replay_stream={args.replay_stream} "
          f"env={ {k: v for k, v in os.environ.items() if k.startswith(('HSA_KERNARG', 'DEBUG_CLR', 'DEBUG_HIP', 'HIP_FORCE', 'ROC_'))} }")
    print(f"right after capture: sums equal eager: {torch.equal(sums, want)}  {sums.tolist()}", flush=True)

    z = torch.zeros(1000, device=dev)
    many = [torch.zeros(8, device=dev) for _ in range(64)]
    src = torch.zeros(1 << 16, device=dev)
    fgraph = None
    if args.filler == "graph":
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            z.add_(1.0)
        torch.cuda.current_stream().wait_stream(side)
        fgraph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(fgraph):
            for _ in range(100):
                z.add_(1.0)

    def filler(n):
        if args.filler == "add":       # one small elementwise kernel per launch (kernel arguments of ~100 bytes)
            for _ in range(n):
                z.add_(1.0)
        elif args.filler == "big":     # multi-tensor kernels: kernel arguments of a few kilobytes per launch
            for _ in range(n):
                torch._foreach_add_(many, 1.0)
        elif args.filler == "copy":    # no kernel arguments from the kernarg pool at all: device-to-device copies
            for _ in range(n):
                z.copy_(src[:1000], non_blocking=True)
        else:                          # launches from ANOTHER graph's replay (100 kernels per replay)
            for _ in range(max(1, n // 100)):
                fgraph.replay()

    done, bad_at = 0, None
    while done < args.max:
        filler(args.step)
        done += args.step
        if args.replay_stream == "side":
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                graph.replay()
            torch.cuda.current_stream().wait_stream(side)
        else:
            graph.replay()
        torch.cuda.synchronize()
        if not torch.equal(sums, want):
            bad_at = done
            break
    if bad_at is None:
        print(f"no difference after {done} filler launches ({done // args.step} replays)")
        sys.exit(0)
    print(f"FIRST DIFFERENCE after {bad_at} filler launches: sums {sums.tolist()}  eager {want.tolist()}")
    for name in sorted(keep, key=lambda s: (int(s[1:s.index('.')]), s)):
        g, e = keep[name], eager[name]
        ok = torch.equal(g, e)
        print(f"   {name:14s} {'ok ' if ok else 'BAD'} graph {g.flatten()[:1].tolist()} eager {e.flatten()[:1].tolist()}")
    graph.replay()
    torch.cuda.synchronize()
    print(f"one more replay right away: sums {sums.tolist()} (equal eager: {torch.equal(sums, want)})")
