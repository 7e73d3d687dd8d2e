#!/usr/bin/env python3
"""The reference's minimal working example (README.md:33-38) end to end on GPUs:

    Data_prepare -> Shared_extraction -> Model_training -> Online_predictor -> error report

    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 examples/beam_coarse_workflow.py \
        --mesh tests/golden/beam_coarse_mesh.npz --steps 100000 --epochs 400 --out /tmp/saa_run

One process per rank; on a one-GPU machine pass --same-device --backend gloo (ranks share cuda:0).
Writes the reference's artefact tree under --out and prints, per rank, the rel-L2 error of the
sync-avoiding trajectory against the synchronised one (what Results/plotter.py shows as a plot).
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mesh", default="tests/golden/beam_coarse_mesh.npz")
    ap.add_argument("--steps", type=int, default=100000)
    ap.add_argument("--epochs", type=int, default=400)
    ap.add_argument("--filter-size", type=int, default=150)
    ap.add_argument("--hidden-size", type=int, default=50)
    ap.add_argument("--out", default="/tmp/saa_run")
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--same-device", action="store_true")
    ap.add_argument("--sequential-training", action="store_true",
                    help="ranks sharing one GPU train one after the other (each with the graphed optimiser step) instead "
                         "of at the same time with eager launches: what a full-schedule run on a one-GPU box wants")
    ap.add_argument("--resync-every", default="",
                    help="comma-separated k: after the reference's loop, run the sync-avoiding part again with synchronised "
                         "steps (one window) after every k predicted windows - an extension, drivers.online_predictor")
    args = ap.parse_args()
    if args.epochs <= 0:  # the reference's schedule: until the decayed rate reaches lr_min (Model_training.py:65)
        args.epochs = None
    # ranks sharing one GPU: graph replays of two processes on one device get in each other's way (2x slower than eager
    # launches); with one GPU per rank the graphed optimiser step is 3.7x faster and stays on
    train_graph = False if (args.same_device and not args.sequential_training) else None

    import torch
    import torch.distributed as dist

    from synchronization_avoiding_algorithms_amd import drivers, training
    from synchronization_avoiding_algorithms_amd.mesh import Mesh, read_vtk

    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = 0 if args.same_device else int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group(args.backend)
    if args.mesh.endswith(".npz"):
        g = np.load(args.mesh)
        mesh = Mesh(g["points"], {"tetra": g["tetra"], "triangle": g["triangle"]})
    else:
        mesh = read_vtk(args.mesh)
    barrier = dist.barrier if world > 1 else (lambda: None)

    t0 = time.time()
    _, truth = drivers.data_prepare(mesh, args.steps, 1, args.out, rank, world, device=local)
    barrier()
    t1 = time.time()
    drivers.shared_extraction(args.out, rank)
    if world == 1:  # one partition: no shared nodes, nothing to predict (the reference's example runs on >= 2 ranks)
        print(f"[rank 0] steps {args.steps}: data_prepare {t1 - t0:.1f} s; a single partition has no shared nodes - run "
              "with --nproc-per-node 2 (add --same-device --backend gloo on a one-GPU machine) for the sync-avoiding part")
        return
    for turn in range(world if args.sequential_training else 1):
        if not args.sequential_training or turn == rank:
            path, tl, vl = training.train_rank_model(args.out, rank, device=f"cuda:{local}", hidden_size=args.hidden_size,
                                                     filter_size=args.filter_size, num_epochs=args.epochs, seed=rank,
                                                     verbose=True, graph=train_graph)
        barrier()
    t2 = time.time()
    _, modeled, _ = drivers.online_predictor(mesh, args.steps, 1, args.out, rank, world, device=local,
                                             filter_size=args.filter_size, hidden_size=args.hidden_size)
    t3 = time.time()
    i_cri = 20 * args.filter_size
    win = 20 * args.filter_size

    def errors(modeled):
        err_all = np.linalg.norm(modeled - truth) / np.linalg.norm(truth)
        err_pred = np.linalg.norm(modeled[:, i_cri:] - truth[:, i_cri:]) / np.linalg.norm(truth[:, i_cri:])
        # the field at the end of every window (n_future * filter_size steps each, Online_predictor.py:284)
        ends = list(range(i_cri + win - 1, modeled.shape[1], win))
        return err_all, err_pred, [float(np.linalg.norm(modeled[:, e] - truth[:, e]) / np.linalg.norm(truth[:, e])) for e in ends]

    err_all, err_pred, per_window = errors(modeled)
    print(f"[rank {rank}] rel-L2 of the rank's displacement field at the end of predicted window 1.."
          f"{len(per_window)}: " + " ".join(f"{v:.2e}" for v in per_window), flush=True)
    for k in [int(v) for v in args.resync_every.split(",") if v]:
        barrier()
        tk = time.time()
        _, again, _ = drivers.online_predictor(mesh, args.steps, 1, args.out, rank, world, device=local,
                                               filter_size=args.filter_size, hidden_size=args.hidden_size, resync_every=k)
        _, e_pred, e_win = errors(again)
        n_sync = sum(1 for w in range(len(e_win)) if w % (k + 1) == k)
        print(f"[rank {rank}] one synchronised window after every {k} predicted ones ({n_sync} of {len(e_win)} windows "
              f"synchronised, {time.time() - tk:.1f} s): {e_pred:.3e} over the phase after the warm-up; at the window ends: "
              + " ".join(f"{v:.2e}" for v in e_win), flush=True)
    print(f"[rank {rank}] steps {args.steps}: data_prepare {t1 - t0:.1f} s, extraction+training ({len(tl)} epochs, "
          f"final train/val MSE {tl[-1]:.3e}/{vl[-1]:.3e}) {t2 - t1:.1f} s, online_predictor {t3 - t2:.1f} s; "
          f"rel-L2(sync-avoiding vs synchronised) = {err_all:.3e} overall, {err_pred:.3e} over the predicted phase",
          flush=True)
    barrier()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
