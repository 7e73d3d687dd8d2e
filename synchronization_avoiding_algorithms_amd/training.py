"""Per-rank training of the shared-node LSTM on PyTorch-ROCm.

Counterpart of /root/reference ``Model_training.py`` + the training half of ``Tools/DNN_tools.py``
(``model_train`` ``:103-165``, ``model_test`` ``:170-207``, windowing ``:284-313``, scaling ``:259-269``):
one independent model per rank on that rank's shared-dof trajectory, Adam with exponentially decayed
learning rate down to ``lr_min``, mini-batches of ``n_B`` windows, MSE on the recursively decoded
``n_future`` steps in the [-1, 0] scaling, random 75/25 train/validation split, ``model.pth`` (a plain
``state_dict``) at the path ``Online_predictor.py:139-140`` expects.  No gradient exchange between ranks.
"""
from __future__ import annotations

import math
import os

import numpy as np
import torch
import torch.nn as nn

from . import results_io as rio
from .predictor import LSTM_encoder_decoder


def windowed_dataset(displacement_shared, filter_size, n_past, n_future, cut_off, device="cpu"):
    """``Dis_data_filtered_subset_coronary`` (``DNN_tools.py:284-313``) without the Python copy loop:
    ``(groups, n_past, in)`` inputs and ``(groups, n_future, in)`` targets, fp32, on ``device``."""
    data = np.asarray(displacement_shared).transpose()
    data = data[0:int(cut_off * len(data)), :][0::filter_size, :]
    series = torch.from_numpy(np.ascontiguousarray(data)).float().to(device)
    groups = series.shape[0] - n_future - n_past + 1
    if groups < 1:
        raise ValueError("trajectory too short for one (n_past, n_future) window")
    win = series.unfold(0, n_past + n_future, 1).permute(0, 2, 1)     # (groups, n_past+n_future, in)
    return win[:, :n_past, :].contiguous(), win[:, n_past:, :].contiguous()


def scale_to_zero_one(X, Y):
    """``Scale_to_zero_one`` (``DNN_tools.py:259-269``): joint max/min, values mapped to [-1, 0]."""
    smin, smax = min(X.min(), Y.min()), max(X.max(), Y.max())
    return (X - smax) / (-smin + smax), (Y - smax) / (-smin + smax), smax.item(), smin.item()


def _decode(model, X, n_future):
    """Recursive decoding (``DNN_tools.py:119-127,185-193``): the decoder's own output is its next input."""
    if X.is_cuda and _FOLDED and model.decoder.dropout.p == 0.0:
        return _decode_folded(model, X, n_future)
    h, c = model.encoder(X)
    inp = X[:, -1, :]
    outs = []
    for _ in range(n_future):
        inp, h, c = model.decoder(inp, h, c)
        outs.append(inp)
    return torch.stack(outs, dim=1)


#: GPU formulation of the training pass (tests flip them to compare with the literal form; the CPU path is always literal)
_FOLDED = True          # decoder feedback folded into the recurrent matrix (_decode_folded)
_FUSED_CELL = True      # pointwise part of an LSTM step as one library kernel (_FusedCell)


class _FusedCell(torch.autograd.Function):
    """The pointwise part of an LSTM step on the library's kernels (``saa_lstm_cell_forward`` / ``_backward``): one launch
    where the elementwise formulation takes ten, one for its backward where autograd takes twenty - the training step is
    launch-bound (~4 us per kernel inside the replayed HIP graph).  Capturable: the kernels go to torch's current stream."""

    @staticmethod
    def forward(ctx, gates, c_prev):
        from . import _lib

        lib = _lib.load()
        gates, c_prev = gates.contiguous(), c_prev.contiguous()
        B, D = c_prev.shape
        h, c, tanh_c, act = torch.empty_like(c_prev), torch.empty_like(c_prev), torch.empty_like(c_prev), torch.empty_like(gates)
        stream = torch.cuda.current_stream(gates.device).cuda_stream
        _lib.check(lib.saa_lstm_cell_forward(gates.device.index, B, D, gates.data_ptr(), c_prev.data_ptr(), h.data_ptr(),
                                             c.data_ptr(), act.data_ptr(), tanh_c.data_ptr(), stream))
        ctx.save_for_backward(act, tanh_c, c_prev)
        return h, c

    @staticmethod
    def backward(ctx, dh, dc):
        from . import _lib

        lib = _lib.load()
        act, tanh_c, c_prev = ctx.saved_tensors
        B, D = c_prev.shape
        dh = dh.contiguous() if dh is not None else None
        dc = dc.contiguous() if dc is not None else None
        dgates, dc_prev = torch.empty_like(act), torch.empty_like(c_prev)
        stream = torch.cuda.current_stream(act.device).cuda_stream
        _lib.check(lib.saa_lstm_cell_backward(act.device.index, B, D, act.data_ptr(), tanh_c.data_ptr(), c_prev.data_ptr(),
                                              dh.data_ptr() if dh is not None else None,
                                              dc.data_ptr() if dc is not None else None, dgates.data_ptr(),
                                              dc_prev.data_ptr(), stream))
        return dgates, dc_prev


_FUSED_RECURRENCE = True  # whole recurrences as single launches (_Recurrence)


class _Recurrence(torch.autograd.Function):
    """``h_t, c_t`` over all time steps of one LSTM recurrence (``gates_t = pre_t + h_{t-1} W^T``) in ONE launch of the
    library (``saa_lstm_recurrence_forward``), its backward in one more plus one product for the weight gradient -
    where the step-by-step formulation costs two launches per step forward and seven backward (decoder) or four per step,
    layer, direction and pass (MIOpen's encoder).  Widths 50 and 100."""

    @staticmethod
    def forward(ctx, pre, h0, c0, W, reverse):
        from . import _lib

        lib = _lib.load()
        pre, W = pre.contiguous(), W.contiguous()
        B, T, G = pre.shape
        H = G // 4
        h0c = h0.contiguous() if h0 is not None else None
        c0c = c0.contiguous() if c0 is not None else None
        Hall = torch.empty((B, T, H), dtype=pre.dtype, device=pre.device)
        c_all, tanhc, act = torch.empty_like(Hall), torch.empty_like(Hall), torch.empty_like(pre)
        stream = torch.cuda.current_stream(pre.device).cuda_stream
        _lib.check(lib.saa_lstm_recurrence_forward(pre.device.index, B, T, H, int(bool(reverse)), pre.data_ptr(),
                                                   h0c.data_ptr() if h0c is not None else None,
                                                   c0c.data_ptr() if c0c is not None else None, W.data_ptr(),
                                                   Hall.data_ptr(), c_all.data_ptr(), act.data_ptr(), tanhc.data_ptr(), stream))
        ctx.save_for_backward(Hall, c_all, act, tanhc, W, h0c if h0c is not None else torch.empty(0, device=pre.device),
                              c0c if c0c is not None else torch.empty(0, device=pre.device))
        ctx.reverse, ctx.has_h0, ctx.has_c0 = bool(reverse), h0c is not None, c0c is not None
        last = 0 if reverse else T - 1
        return Hall, c_all[:, last, :].clone()

    @staticmethod
    def backward(ctx, dH, dcT):
        from . import _lib

        lib = _lib.load()
        Hall, c_all, act, tanhc, W, h0, c0 = ctx.saved_tensors
        B, T, H = Hall.shape
        dH = dH.contiguous() if dH is not None else torch.zeros_like(Hall)
        dcT = dcT.contiguous() if dcT is not None else None
        dpre, dh0, dc0 = torch.empty_like(act), torch.empty((B, H), dtype=Hall.dtype, device=Hall.device), \
            torch.empty((B, H), dtype=Hall.dtype, device=Hall.device)
        stream = torch.cuda.current_stream(Hall.device).cuda_stream
        _lib.check(lib.saa_lstm_recurrence_backward(Hall.device.index, B, T, H, int(ctx.reverse), dH.data_ptr(),
                                                    dcT.data_ptr() if dcT is not None else None,
                                                    c0.data_ptr() if ctx.has_c0 else None, W.data_ptr(), c_all.data_ptr(),
                                                    act.data_ptr(), tanhc.data_ptr(), dpre.data_ptr(), dh0.data_ptr(),
                                                    dc0.data_ptr(), stream))
        # h_{t-1} of every step in processing order: the initial state, then the outputs shifted by one step
        first = h0.unsqueeze(1) if ctx.has_h0 else torch.zeros((B, 1, H), dtype=Hall.dtype, device=Hall.device)
        hprev = torch.cat((Hall[:, 1:, :], first), dim=1) if ctx.reverse else torch.cat((first, Hall[:, :-1, :]), dim=1)
        dW = dpre.reshape(B * T, 4 * H).t() @ hprev.reshape(B * T, H)
        return dpre, (dh0 if ctx.has_h0 else None), (dc0 if ctx.has_c0 else None), dW, None


def _recurrence_ok(t, width):
    return _FUSED_RECURRENCE and t.is_cuda and t.dtype == torch.float32 and width in (50, 100)


def _encode_fused(model, X):
    """``LSTM_Encoder.forward`` (``DNN_tools.py:35-59``) with every (layer, direction) recurrence as one launch
    (:class:`_Recurrence`): per layer and direction one product for the input projections of all time steps, the recurrence,
    and the layer's output as the concatenation of the two directions; returns the last layer's final forward | backward
    states like the module does."""
    enc = model.encoder
    lstm, H, L, dirs = enc.lstm_encoder, enc.hidden_size, enc.num_layers, enc.D
    B, T = X.shape[0], X.shape[1]
    inp = X
    for layer in range(L):
        outs, finals = [], []
        for d in range(dirs):
            sfx = f"_l{layer}" + ("_reverse" if d else "")
            w_ih, w_hh = getattr(lstm, "weight_ih" + sfx), getattr(lstm, "weight_hh" + sfx)
            bias = getattr(lstm, "bias_ih" + sfx) + getattr(lstm, "bias_hh" + sfx)
            pre = torch.addmm(bias, inp.reshape(B * T, -1), w_ih.t()).reshape(B, T, 4 * H)
            Hall, cT = _Recurrence.apply(pre, None, None, w_hh, bool(d))
            outs.append(Hall)
            finals.append((Hall[:, 0 if d else T - 1, :], cT))
        inp = torch.cat(outs, dim=2) if dirs == 2 else outs[0]
    if dirs == 2:
        return (torch.cat((finals[0][0], finals[1][0]), 1).unsqueeze(0), torch.cat((finals[0][1], finals[1][1]), 1).unsqueeze(0))
    return finals[0][0].unsqueeze(0), finals[0][1].unsqueeze(0)


def _cell(gates, c):
    """``c' = f c + i g``, ``h = o tanh(c')`` from the pre-activations ``(B, 4D)`` in PyTorch's gate order."""
    if gates.is_cuda and _FUSED_CELL and gates.dtype == torch.float32:
        return _FusedCell.apply(gates, c)
    gi, gf, gg, go = gates.chunk(4, dim=1)
    c = torch.sigmoid(gf) * c + torch.sigmoid(gi) * torch.tanh(gg)
    return torch.sigmoid(go) * torch.tanh(c), c


def _decode_folded(model, X, n_future):
    """The same function of the same parameters with the decoder's feedback folded (GPU only; the CPU path always
    keeps the literal form): from its second step on the decoder's input is
    ``fc(h) = h W_fc^T + b_fc``, so ``gates = fc(h) W_ih^T + h W_hh^T + b = h (W_ih W_fc + W_hh)^T + (W_ih b_fc + b)``.
    The folded matrix is formed once per pass (a 4D x I by I x D product) and every step is a D -> 4D product instead of
    an I -> 4D one plus an ``nn.LSTM`` call; the outputs of all steps come from one product at the end.  Autograd
    differentiates through the fold, so the gradients with respect to ``W_ih``, ``W_fc``, ``W_hh`` and the biases are those
    of the literal form up to fp32 round-off (what predictor kernels do for inference, ``csrc/saa_predictor.hip``)."""
    enc = model.encoder
    if _recurrence_ok(X, enc.hidden_size) and enc.lstm_encoder.dropout == 0.0:
        h, c = _encode_fused(model, X)
    else:
        h, c = model.encoder(X)
    h, c = h[0], c[0]
    lstm, fc = model.decoder.lstm_decoder, model.decoder.fc
    w_ih, w_hh, bias = lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0 + lstm.bias_hh_l0
    w_comb = torch.addmm(w_hh, w_ih, fc.weight)                       # (4D, D)
    b_comb = torch.addmv(bias, w_ih, fc.bias)
    gates = torch.addmm(bias, X[:, -1, :], w_ih.t()) + h @ w_hh.t()   # first step: the last history row (:224)
    if _recurrence_ok(X, model.decoder.hidden_size):
        # all steps in one launch: the first step's pre-activations as they stand, the bias of the folded form for the
        # others; the recurrent matrix of the kernel is the folded one, and its initial h is zero (h's part of the first
        # step is already inside `gates`)
        B = X.shape[0]
        pre = torch.cat((gates.unsqueeze(1), b_comb.expand(B, n_future - 1, -1)), dim=1) if n_future > 1 else gates.unsqueeze(1)
        H, _ = _Recurrence.apply(pre, None, c, w_comb, False)
        return torch.matmul(H, fc.weight.t()) + fc.bias
    hs = []
    for t in range(n_future):
        if t > 0:
            gates = torch.addmm(b_comb, h, w_comb.t())
        h, c = _cell(gates, c)
        hs.append(h)
    H = torch.stack(hs, dim=1)                                        # (B, n_future, D)
    return torch.matmul(H, fc.weight.t()) + fc.bias


class GraphedTrainStep:
    """One optimiser step (decode, loss, backward, Adam) captured as a HIP graph; the running sums of the epoch are formed
    outside it (:meth:`_stats`).

    The step is ~1300 kernels of a few microseconds each: launched one by one from Python the GPU idles most of the
    time (13 ms per step at input size 24), replayed as a graph it takes 4 ms.  The first ``warmup`` full batches of
    the first epoch run eagerly on a side stream (they are ordinary training steps), then the step is captured once
    and replayed for every later full batch; a trailing partial batch runs eagerly.  Needs an optimiser built with
    ``capturable=True`` and a tensor learning rate (so that the scheduler's updates reach the replays)."""

    def __init__(self, model, criterion, optimizer, n_future, batch_shape_x, batch_shape_y, device, warmup=3):
        self.model, self.criterion, self.optimizer, self.n_future = model, criterion, optimizer, n_future
        self.X = torch.zeros(batch_shape_x, dtype=torch.float32, device=device)
        self.Y = torch.zeros(batch_shape_y, dtype=torch.float32, device=device)
        self.sums = torch.zeros(3, dtype=torch.float64, device=device)
        self._scratch = torch.zeros(3, dtype=torch.float64, device=device)
        self.graph, self.seen, self.warmup = None, 0, warmup
        self.side = torch.cuda.Stream(device=device)

    def _step(self):
        self.optimizer.zero_grad(set_to_none=True)
        out = _decode(self.model, self.X, self.n_future)
        self.out = out.detach()  # (same storage, static inside the captured graph: read by _stats; no hold on the autograd graph)
        loss = self.criterion(out, self.Y)
        loss.backward()
        self.optimizer.step()

    def _stats(self):
        """The three running sums of ``model_train`` (``DNN_tools.py:144-155``) from the step's decoded output, computed
        OUTSIDE the captured graph by the library's two-launch kernel (fp64 sums).  (Round 3 moved them out because the
        loss value came back corrupted from replays; round 4 found the cause - multi-block reductions in graphs replayed
        from pre-recorded packets, ``hip_graphs.py`` - and the graph is only used where the canary passes.  The sums stay
        outside: one launch pair instead of a dozen tiny ATen kernels per step.)"""
        from . import _lib

        if not isinstance(self.criterion, nn.MSELoss) or self.out.dtype != torch.float32:
            with torch.no_grad():
                loss = self.criterion(self.out, self.Y)
                self.sums.add_(torch.stack([loss, 1.0 - loss / (self.Y - self.Y.mean()).square().mean(),
                                            1.0 - loss / self.Y.square().mean()]).double())
            return
        out = self.out if self.out.is_contiguous() else self.out.contiguous()
        _lib.check(_lib.load().saa_train_stats(out.device.index, out.numel(), out.data_ptr(), self.Y.data_ptr(),
                                               self._scratch.data_ptr(), self.sums.data_ptr(),
                                               torch.cuda.current_stream(out.device).cuda_stream))

    def run(self, X, Y):
        self.X.copy_(X)
        self.Y.copy_(Y)
        if self.graph is not None:
            self.graph.replay()
            self._stats()
            return
        if self.seen < self.warmup:  # eager, on a side stream as graph capture wants its warm-up
            self.side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.side):
                self._step()
            torch.cuda.current_stream().wait_stream(self.side)
            self._stats()
            self.seen += 1
            return
        graph = torch.cuda.CUDAGraph()
        self.optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(graph):
            self._step()
        self.graph = graph
        self.graph.replay()  # capture does not execute: this is the step for the batch just copied in
        self._stats()


class BatchedValidation:
    """The validation pass of one epoch (``model_test``, ``DNN_tools.py:170-207``, over the fixed, unshuffled validation
    batches) as ONE forward pass over all validation windows: the windows are independent rows of the batch, so the
    per-batch mean squares of the reference's loop are segment sums of the per-window squared errors; the denominators of
    the two accuracy figures depend on the targets only and are formed once.  ~200 launches and one transfer per epoch.

    (Round 2 replayed the batch loop as a HIP graph and got corrupted sums back - negative mean squares: its ``mse_loss``
    / ``mean`` nodes are multi-block reductions, which replay wrongly from pre-recorded packets on this ROCm,
    ``hip_graphs.py``.  One batched eager pass is cheaper than that graph was anyway.)"""

    def __init__(self, model, criterion, batches, n_future, device):
        self.model, self.n_future = model, n_future
        self.X = torch.cat([b[0] for b in batches])
        self.Y = torch.cat([b[1] for b in batches])
        sizes = [int(b[0].shape[0]) for b in batches]
        self.seg = torch.repeat_interleave(torch.arange(len(sizes), device=device), torch.tensor(sizes, device=device))
        per = float(self.Y[0].numel())
        self.count = torch.tensor(sizes, dtype=torch.float32, device=device) * per        # elements per batch
        with torch.no_grad():                                                             # DNN_tools.py:196-201
            self.den_r2 = torch.stack([criterion(b[1], torch.mean(b[1]) + torch.zeros_like(b[1])) for b in batches])
            self.den_rel = torch.stack([criterion(b[1], torch.zeros_like(b[1])) for b in batches])
        self.n_batches = len(sizes)

    def run(self):
        self.model.eval()
        with torch.no_grad():
            err = (_decode(self.model, self.X, self.n_future) - self.Y).square().flatten(1).sum(1)
            loss = torch.zeros(self.n_batches, dtype=torch.float32, device=err.device).index_add_(0, self.seg, err) / self.count
            sums = torch.stack([loss.double().sum(), (1.0 - loss / self.den_r2).double().sum(),
                                (1.0 - loss / self.den_rel).double().sum()])
        return tuple(sums.tolist())


def _decode_teacher_forced(model, X, Y, n_future, ratio):
    """'mtf' decoding of ``DNN_tools.py:131-142`` (not used in the paper): with probability ``ratio`` the next decoder
    input is the truth instead of the model's own output."""
    import random

    h, c = model.encoder(X)
    inp, outs = X[:, -1, :], []
    for i in range(n_future):
        out, h, c = model.decoder(inp, h, c)
        outs.append(out)
        inp = Y[:, i, :] if random.random() < ratio else out
    return torch.stack(outs, dim=1)


def model_train(device, model, batches, criterion, optimizer, n_future, training_method="recursive", ratio=0.5,
                graphed=None):
    """One epoch of training with the signature of ``DNN_tools.py:103`` (``batches``: any iterable of ``(X, Y)``
    mini-batches, e.g. a ``DataLoader`` over ``MyDataset``).  'recursive' decoding (the paper's method) may run as a
    replayed HIP graph (``graphed``); 'mtf' runs eagerly.
    Returns (sum of batch losses, sum of R2 accuracies, sum of relative accuracies, model)."""
    if training_method not in ("recursive", "mtf"):
        raise ValueError("training_method must be 'recursive' or 'mtf'")
    model.train()
    # the three running sums stay on the device (float64, like the reference's Python floats) and come back once per
    # epoch: a .item() per batch would stall the launch-bound GPU three times per optimiser step
    sums = torch.zeros(3, dtype=torch.float64, device=device)
    if graphed is not None:
        graphed.sums.zero_()
    for X, Y in batches:
        if graphed is not None and training_method == "recursive" and X.shape == graphed.X.shape:
            graphed.run(X, Y)
            continue
        optimizer.zero_grad()
        if training_method == "recursive":
            out = _decode(model, X, n_future)
        else:
            out = _decode_teacher_forced(model, X, Y, n_future, ratio)
            ratio = ratio - 0.005 if ratio > 0.005 else ratio  # DNN_tools.py:160-163
        loss = criterion(out, Y)
        with torch.no_grad():
            sums[1] += (1.0 - loss / criterion(Y, torch.mean(Y) + torch.zeros_like(Y))).double()
            sums[2] += (1.0 - loss / criterion(Y, torch.zeros_like(Y))).double()
            sums[0] += loss.detach().double()
        loss.backward()
        optimizer.step()
    if graphed is not None:
        sums += graphed.sums
    loss_sum, r2_sum, rel_sum = sums.tolist()
    return loss_sum, r2_sum, rel_sum, model


def model_test(device, model, batches, criterion, n_future):
    """Validation pass (``DNN_tools.py:170-207``)."""
    model.eval()
    sums = torch.zeros(3, dtype=torch.float64, device=device)
    with torch.no_grad():
        for X, Y in batches:
            loss = criterion(_decode(model, X, n_future), Y)
            sums[0] += loss.double()
            sums[1] += (1.0 - loss / criterion(Y, torch.mean(Y) + torch.zeros_like(Y))).double()
            sums[2] += (1.0 - loss / criterion(Y, torch.zeros_like(Y))).double()
    loss_sum, r2_sum, rel_sum = sums.tolist()
    return loss_sum, r2_sum, rel_sum


def _batches(X, Y, batch_size, shuffle, generator=None):
    idx = torch.randperm(X.shape[0], generator=generator, device="cpu") if shuffle else torch.arange(X.shape[0])
    idx = idx.to(X.device)
    return [(X[idx[i:i + batch_size]], Y[idx[i:i + batch_size]]) for i in range(0, X.shape[0], batch_size)]


def windows_from_history(hist, filter_size, n_past, n_future, cut_off=1.0):
    """The windows of :func:`windowed_dataset` cut from a device-resident shared-dof history ``(n_steps, in)`` - the
    layout of ``d_sol_shared`` (``Online_predictor.py:260``), i.e. the transpose of the ``Displacement`` dataset
    ``Shared_extraction.py`` stores - without a round trip through the host."""
    series = hist[0:int(cut_off * hist.shape[0])][0::filter_size].float()
    groups = series.shape[0] - n_future - n_past + 1
    if groups < 1:
        raise ValueError("history too short for one (n_past, n_future) window")
    win = series.unfold(0, n_past + n_future, 1).permute(0, 2, 1)
    return win[:, :n_past, :].contiguous(), win[:, n_past:, :].contiguous()


def fit_windows(X, Y, hidden_size=50, batch_size=10, learning_rate=5e-4, decay=0.998, lr_min=5e-7, T_portion=0.75,
                num_epochs=None, max_seconds=None, generator=None, verbose=False, rank=0, log=None, graph=None):
    """The training loop of ``Model_training.py:60-139`` on already scaled windows ``X (groups, n_past, in)``,
    ``Y (groups, n_future, in)``: Adam, learning rate ``decay**epoch``, random ``T_portion`` training split, shuffled
    mini-batches, validation on the rest; on a GPU the optimiser step is replayed as a HIP graph.  Stops after
    ``num_epochs`` (default: until the rate reaches ``lr_min``, ``:65``) or ``max_seconds`` of wall time.  ``verbose``:
    the reference's progress line every 50 epochs (rank 0), to ``log`` (default: stdout).
    Returns ``(model, train_loss, validation_loss)`` (per-epoch means)."""
    import time

    device = X.device
    n_future = Y.shape[1]
    model = LSTM_encoder_decoder(X.shape[2], hidden_size, 2, True, 0.0, 0.0).to(device)
    criterion = nn.MSELoss()
    # the optimiser step as a replayed HIP graph - only where replays are proven right in this process (hip_graphs.py:
    # the loss reduction inside the step is exactly the kind of node that replays wrongly from pre-recorded packets)
    use_graph = device.type == "cuda" and (graph is None or bool(graph))
    if use_graph:
        from . import hip_graphs

        use_graph = hip_graphs.replays_are_trustworthy(device)
    if use_graph:  # capturable Adam with a tensor learning rate: the scheduler's updates reach the graph replays
        optimizer = torch.optim.Adam(model.parameters(), lr=torch.tensor(learning_rate, device=device), capturable=True,
                                     fused=True)  # one kernel, one pass over the parameters
    else:
        optimizer = torch.optim.Adam(model.parameters(), lr=learning_rate)
    scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda epoch: decay ** epoch)
    if num_epochs is None:
        num_epochs = int(math.log(lr_min / learning_rate, decay))     # Model_training.py:65
    n = X.shape[0]
    train_idx = np.random.choice(n, size=int(T_portion * n), replace=False)
    test_idx = np.setdiff1d(np.arange(n), train_idx)
    Xtr, Ytr = X[torch.as_tensor(train_idx, device=device)], Y[torch.as_tensor(train_idx, device=device)]
    Xte, Yte = X[torch.as_tensor(test_idx, device=device)], Y[torch.as_tensor(test_idx, device=device)]
    train_loss, test_loss = [], []
    graphed = None
    if use_graph and Xtr.shape[0] >= batch_size:
        graphed = GraphedTrainStep(model, criterion, optimizer, n_future, (batch_size,) + tuple(Xtr.shape[1:]),
                                   (batch_size,) + tuple(Ytr.shape[1:]), device)
    vb = _batches(Xte, Yte, batch_size, False)  # (fixed: the validation split is not shuffled)
    gval = BatchedValidation(model, criterion, vb, n_future, device) if device.type == "cuda" and vb else None
    t0 = time.time()
    for epoch in range(num_epochs):
        tb = _batches(Xtr, Ytr, batch_size, True, generator)
        lt, r2, _, model = model_train(device, model, tb, criterion, optimizer, n_future, graphed=graphed)
        if gval is not None:
            lv, _, _ = gval.run()
        else:
            lv, _, _ = model_test(device, model, vb, criterion, n_future) if vb else (float("nan"), 0, 0)
        train_loss.append(lt / len(tb))
        test_loss.append(lv / max(len(vb), 1))
        if verbose and rank == 0 and epoch % 50 == 0:
            print("Epoch: %d, mse training loss: %1.5e, R2 accuracy: %.3f, lr=%g"
                  % (epoch, train_loss[-1], r2 / len(tb), float(optimizer.param_groups[0]["lr"])), file=log, flush=True)
        scheduler.step()
        if max_seconds is not None and time.time() - t0 > max_seconds:
            break
    return model, train_loss, test_loss


def train_on_history(hist, filter_size, n_past, n_future, cut_off=1.0, seed=None, **fit_kw):
    """Train one rank's model straight from the shared-dof history a synchronised run left on the device
    (``PartitionedSolver.step_synced(n, hist, 0)``): windowing, [-1, 0] scaling, :func:`fit_windows`.
    Returns ``(model, scale_max, scale_min, train_loss, validation_loss)`` - what ``DevicePredictor`` needs."""
    gen = None
    if seed is not None:
        torch.manual_seed(seed)
        np.random.seed(seed)
        gen = torch.Generator().manual_seed(seed)
    X, Y = windows_from_history(hist, filter_size, n_past, n_future, cut_off)
    X, Y, smax, smin = scale_to_zero_one(X, Y)
    model, train_loss, test_loss = fit_windows(X, Y, generator=gen, **fit_kw)
    return model.eval(), smax, smin, train_loss, test_loss


def train_rank_model(out_dir=".", rank=0, device=None, batch_size=10, learning_rate=5e-4, hidden_size=50,
                     filter_size=150, cut_off=0.5, lr_min=5e-7, decay=0.998, T_portion=0.75, n_future=20, n_past=20,
                     num_epochs=None, seed=None, verbose=False, graph=None):
    """``Model_training.py:17-181`` for one rank; returns ``(model_path, train_loss, validation_loss)``.  ``graph``: on a
    GPU, replay the optimiser step as a HIP graph (default) or launch it eagerly (False)."""
    from .drivers import PATHS

    device = torch.device(device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu"))
    gen = None
    if seed is not None:
        torch.manual_seed(seed)
        np.random.seed(seed)
        gen = torch.Generator().manual_seed(seed)
    shared = rio.load_int_list(os.path.join(out_dir, PATHS["shared"].format(r=rank)))
    traj = rio.load_displacement(os.path.join(out_dir, PATHS["shared_traj"].format(r=rank)))
    if len(shared) == 0:
        raise ValueError(f"rank {rank} has no shared nodes: nothing to train (a one-partition run)")
    if traj.shape[0] != 3 * len(shared):
        raise ValueError("shared-dof trajectory and shared-node list disagree")
    X, Y = windowed_dataset(traj, filter_size, n_past, n_future, cut_off, device)
    X, Y, _, _ = scale_to_zero_one(X, Y)
    model, train_loss, test_loss = fit_windows(X, Y, hidden_size, batch_size, learning_rate, decay, lr_min, T_portion,
                                               num_epochs, None, gen, verbose, rank, graph=graph)
    path = os.path.join(out_dir, PATHS["model"].format(r=rank, nB=batch_size, nH=hidden_size, lr=learning_rate,
                                                       ns=filter_size))
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savetxt(os.path.join(os.path.dirname(path), "train_loss.csv"), train_loss, delimiter=",")
    np.savetxt(os.path.join(os.path.dirname(path), "test_loss.csv"), test_loss, delimiter=",")
    torch.save(model.state_dict(), path)
    return path, train_loss, test_loss
