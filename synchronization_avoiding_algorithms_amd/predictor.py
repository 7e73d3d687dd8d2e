"""LSTM encoder-decoder shared-node predictor on PyTorch-ROCm.

Same architecture, parameter names and numerics as the reference's per-rank model
(/root/reference ``Tools/DNN_tools.py:16-98``: 2-layer bidirectional LSTM encoder whose last layer's
forward|backward ``(h, c)`` seed a 1-layer LSTM decoder of width ``2H`` followed by ``Linear(2H -> in)``)
so that a ``model.pth`` written by the reference's ``Model_training.py:179-180`` loads unchanged.

What differs is the execution: the reference runs ``filter_size`` (=150) sequential batch-1 passes on
the CPU per prediction window (``Tools/DNN_prediction.py:38-55``); here the phase offsets form ONE batch
on the GPU, the shared-dof history never leaves the device, and the decoder's ``n_future`` steps reuse
pre-allocated buffers.  Arithmetic stays fp32 like the reference (``.float()`` at ``DNN_prediction.py:49``).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn


class LSTM_Encoder(nn.Module):
    """``DNN_tools.py:16-59``; returns the last layer's states as ``(1, N, D*H)``."""

    def __init__(self, input_size, hidden_size, num_layers=2, Bi_dir=True, dp=0.0):
        super().__init__()
        self.input_size, self.hidden_size, self.num_layers = input_size, hidden_size, num_layers
        self.D = 2 if Bi_dir else 1
        self.lstm_encoder = nn.LSTM(input_size=input_size, hidden_size=hidden_size, num_layers=num_layers,
                                    batch_first=True, dropout=dp, bidirectional=bool(Bi_dir))

    def forward(self, x):
        _, (hn, cn) = self.lstm_encoder(x)
        n = x.shape[0]
        hn = hn.view(self.num_layers, self.D, n, self.hidden_size)[-1]
        cn = cn.view(self.num_layers, self.D, n, self.hidden_size)[-1]
        if self.D == 2:
            return torch.cat((hn[0], hn[1]), 1).unsqueeze(0), torch.cat((cn[0], cn[1]), 1).unsqueeze(0)
        return hn, cn


class LSTM_Decoder(nn.Module):
    """``DNN_tools.py:63-80``: one recursive step per call."""

    def __init__(self, input_size, hidden_size, Bi_dir=True, dp=0.0):
        super().__init__()
        self.input_size = input_size
        self.hidden_size = hidden_size * 2 if Bi_dir else hidden_size
        self.lstm_decoder = nn.LSTM(input_size=input_size, hidden_size=self.hidden_size, num_layers=1,
                                    batch_first=True, bidirectional=False)
        self.fc = nn.Linear(self.hidden_size, input_size)
        self.dropout = nn.Dropout(dp)

    def forward(self, x, encoded_hn, encoded_cn):
        out, (hn, cn) = self.lstm_decoder(x.unsqueeze(1), (encoded_hn, encoded_cn))
        return self.fc(self.dropout(out.squeeze(1))), hn, cn


class LSTM_encoder_decoder(nn.Module):
    """``DNN_tools.py:85-98``; state_dict keys ``encoder.lstm_encoder.*``, ``decoder.lstm_decoder.*``,
    ``decoder.fc.*``."""

    def __init__(self, input_size, hidden_size, num_layers_encoder=2, Bi_dir_encoder=True, dp_encoder=0.0,
                 dp_decoder=0.0):
        super().__init__()
        self.input_size, self.hidden_size = input_size, hidden_size
        self.encoder = LSTM_Encoder(input_size, hidden_size, num_layers_encoder, Bi_dir_encoder, dp_encoder)
        self.decoder = LSTM_Decoder(input_size, hidden_size, Bi_dir_encoder, dp_decoder)


def call_model(device, filter_size, input_size, hidden_size, model_path):
    """``DNN_prediction.py:18-34``: 2 encoder layers, bidirectional, no dropout; weights from a local
    ``state_dict`` file (loaded with ``weights_only=True`` - nothing in the file is executed)."""
    model = LSTM_encoder_decoder(input_size, hidden_size, 2, True, 0.0, 0.0)
    model.load_state_dict(torch.load(model_path, map_location=device, weights_only=True))
    return model.to(device)


def scale_forward(X, scale_max, scale_min):
    """``DNN_tools.py:272-275``: maps the training range to [-1, 0]."""
    return (X - scale_max) / (-scale_min + scale_max)


def scale_it_back(X, scale_max, scale_min):
    """``DNN_tools.py:277-279``."""
    return X * (scale_max - scale_min) + scale_max


def model_predict(device, model, X, n_future):
    """``DNN_tools.py:212-234`` for one window ``X (n_past, in)`` or a batch ``(B, n_past, in)``."""
    model.eval()
    single = X.dim() == 2
    if single:
        X = X.unsqueeze(0)
    with torch.no_grad():
        h, c = model.encoder(X)
        out = torch.empty((X.shape[0], n_future, X.shape[2]), device=X.device, dtype=X.dtype)
        inp = X[:, -1, :]
        for i in range(n_future):
            inp, h, c = model.decoder(inp, h, c)
            out[:, i, :] = inp
    return out[0] if single else out


def _phase_indices(n, n_p, n_f, n_s):
    """History / future row indices of every phase offset (``DNN_prediction.py:44-45``)."""
    past = [np.arange(i + n - n_p * n_s, i + n - 1, n_s) for i in range(n_s)]
    fut = [np.arange(i + n, n + i + n_f * n_s - 1, n_s) - n for i in range(n_s)]
    return past, fut


def predict_table(model, n, n_p, n_f, n_s, hist, scale_max, scale_min):
    """Batched ``encoder_decoder_predictor`` on whatever device ``hist`` (float64, ``(steps, in)``) lives on.

    Returns the float64 ``(n_s*n_f, in)`` table whose row ``k`` is the prediction for step ``n + k``.
    """
    past, fut = _phase_indices(n, n_p, n_f, n_s)
    if len({len(p) for p in past}) != 1 or len({len(f) for f in fut}) != 1:
        raise ValueError("phase windows of unequal length (n_s == 1?) cannot be batched")
    dev = hist.device
    pidx = torch.as_tensor(np.stack(past), device=dev)                      # (n_s, n_p)
    X = scale_forward(hist[pidx], scale_max, scale_min).float()            # fp64 scaling, then .float() (:48-49)
    Y = model_predict(dev, model, X, len(fut[0]))                          # (n_s, n_f, in) fp32
    Y = scale_it_back(Y, scale_max, scale_min)                             # fp32 like the reference (:51)
    table = torch.zeros((n_s * n_f, hist.shape[1]), dtype=torch.float64, device=dev)
    fidx = torch.as_tensor(np.stack(fut), device=dev)                      # (n_s, n_f)
    table[fidx.reshape(-1)] = Y.reshape(-1, hist.shape[1]).double()
    return table


def encoder_decoder_predictor(device, n, model, n_p, n_f, n_s, input_size, d_sol, scale_max, scale_min):
    """Drop-in for ``DNN_prediction.py:38-55``: NumPy history in, float64 NumPy table out."""
    lo = n - n_p * n_s
    hist = torch.zeros((n, input_size), dtype=torch.float64, device=device)
    hist[lo:n] = torch.from_numpy(np.ascontiguousarray(d_sol[lo:n, :])).to(device)
    return predict_table(model, n, n_p, n_f, n_s, hist, scale_max, scale_min).cpu().numpy()


def scaling_constants(displacement_shared, filter_size, n_past, n_future, cut_off):
    """``scale_max, scale_min`` of ``Online_predictor.py:130-136``.

    ``displacement_shared`` is the ``(input_size, n_steps)`` array ``Shared_extraction.py:36-39`` stores.
    The reference windows the first ``cut_off`` share of the ``filter_size``-subsampled series
    (``DNN_tools.py:284-313``) and takes max/min over all windows in fp32 (``:259-269``); the windows
    jointly cover every retained sample, hence max/min of the fp32 series.
    """
    data = np.asarray(displacement_shared).transpose()
    data = data[0:int(cut_off * len(data)), :][0::filter_size, :]
    if data.shape[0] < n_past + n_future:
        raise ValueError("trajectory too short for one (n_past, n_future) window")
    data32 = torch.from_numpy(np.ascontiguousarray(data)).float()
    return data32.max().item(), data32.min().item()


#: the reference's state_dict keys in its own order (``Model_training.py:179-180`` saves them) - the
#: order ``saa_predictor_create`` takes the tensors in
STATE_KEYS = tuple(
    [f"encoder.lstm_encoder.{name}_l{layer}{sfx}" for layer in (0, 1) for sfx in ("", "_reverse")
     for name in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    + [f"decoder.lstm_decoder.{name}_l0" for name in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    + ["decoder.fc.weight", "decoder.fc.bias"])


class NativePredictor:
    """The model on the HIP library's own kernels (``csrc/saa_predictor.hip``, C ABI ``saa_predictor_*``): the window's
    3000 history rows go through ONE f32 matrix-core GEMM (the fp64 scaling to [-1, 0] fused into its loads), the
    recurrences run one workgroup per phase with the decoder's output layer folded into its recurrent matrix, and a second
    GEMM scales back and writes the fp64 table - four launches where the PyTorch path takes ~600 (4.7 ms -> 0.46 ms per
    window at 9126 inputs).  fp32 like the reference; another summation order than ATen's (round-off level differences)."""

    def __init__(self, model, n_past, n_future, filter_size, device_index=0):
        import ctypes as C

        from . import _lib

        sd = model.state_dict()
        missing = [k for k in STATE_KEYS if k not in sd]
        if missing:
            raise ValueError(f"not the reference's encoder-decoder: state_dict lacks {missing[:3]}")
        self._lib, self._h = _lib.load(), C.c_void_p()
        self.input_size, hidden = int(sd[STATE_KEYS[0]].shape[1]), int(sd[STATE_KEYS[1]].shape[1])
        want = {"encoder.lstm_encoder.weight_ih_l1": (4 * hidden, 2 * hidden),
                "decoder.lstm_decoder.weight_ih_l0": (8 * hidden, self.input_size),
                "decoder.lstm_decoder.weight_hh_l0": (8 * hidden, 2 * hidden),
                "decoder.fc.weight": (self.input_size, 2 * hidden)}
        for k, shape in want.items():
            if tuple(sd[k].shape) != shape:
                raise ValueError(f"{k} has shape {tuple(sd[k].shape)}, the reference's architecture needs {shape}")
        host = [sd[k].detach().to("cpu", torch.float32).contiguous() for k in STATE_KEYS]
        ptrs = (C.POINTER(C.c_float) * len(host))(*[C.cast(t.data_ptr(), C.POINTER(C.c_float)) for t in host])
        self.n_p, self.n_f, self.n_s, self.device_index = int(n_past), int(n_future), int(filter_size), int(device_index)
        _lib.check(self._lib.saa_predictor_create(self.device_index, self.input_size, hidden, self.n_p, self.n_f, self.n_s,
                                                  ptrs, len(host), C.byref(self._h)))

    def predict(self, n, hist, scale_max, scale_min, table=None):
        """``hist``: fp64 CUDA tensor ``(rows, input_size)`` (unit stride along the inputs); returns the fp64 table
        ``(n_s*n_f, input_size)`` on the same device, enqueued on torch's current stream."""
        from . import _lib

        if not (hist.is_cuda and hist.dtype == torch.float64 and hist.dim() == 2 and hist.stride(1) == 1
                and hist.shape[1] == self.input_size and hist.device.index == self.device_index):
            raise ValueError("history must be a float64 (rows, input_size) tensor on the predictor's GPU")
        if table is None:
            table = torch.empty((self.n_s * self.n_f, self.input_size), dtype=torch.float64, device=hist.device)
        stream = torch.cuda.current_stream(hist.device).cuda_stream
        _lib.check(self._lib.saa_predictor_predict(self._h, hist.data_ptr(), hist.shape[0], hist.stride(0), int(n),
                                                   float(scale_max), float(scale_min), table.data_ptr(), table.stride(0),
                                                   stream))
        return table

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.saa_predictor_destroy(self._h)
            self._h.value = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _lib_error():
    from . import _lib

    return _lib.SaaError


class DevicePredictor:
    """Callable for :func:`distributed.run_hybrid`: keeps model and scaling on the solver's device.

    On a GPU the window's table comes from :class:`NativePredictor` (the library's own kernels; four launches).
    ``backend="torch"`` selects the PyTorch-ROCm path instead (MIOpen / rocBLAS): one call is ~600 launches of
    microsecond kernels (7 ms from Python for 1.5 ms of GPU work, against ~33 ms of time stepping per window), so it is
    captured as a HIP graph after ``warmup`` eager calls and replayed: the window position ``n`` lives in a device scalar,
    the history tensor and the output table are static (``run_hybrid`` allocates the history once and consumes the table
    before the next call).  On the CPU (the drop-in's ``device='cpu'``) the call is the eager batched PyTorch one."""

    def __init__(self, model, n_past, n_future, filter_size, scale_max, scale_min, warmup=2, backend="native",
                 graph=True):
        if backend not in ("native", "torch"):
            raise ValueError("backend must be 'native' (the library's predictor kernels) or 'torch' (PyTorch-ROCm)")
        self.model = model.eval()
        self.n_p, self.n_f, self.n_s = n_past, n_future, filter_size
        self.scale_max, self.scale_min = float(scale_max), float(scale_min)
        self._graph, self._key, self._calls, self._warmup = None, None, 0, warmup
        self._use_graph = bool(graph)
        self._use_native = backend == "native" and filter_size >= 2
        self._native, self._native_table = None, None

    def _eager(self, n, hist):
        return predict_table(self.model, n, self.n_p, self.n_f, self.n_s, hist, self.scale_max, self.scale_min)

    def _capture(self, hist):
        past, fut = _phase_indices(0, self.n_p, self.n_f, self.n_s)
        if len({len(p) for p in past}) != 1 or len({len(f) for f in fut}) != 1:
            return False
        dev = hist.device
        self._n = torch.zeros((), dtype=torch.int64, device=dev)
        self._pidx0 = torch.as_tensor(np.stack(past), device=dev)           # row offsets relative to n (negative)
        self._fidx = torch.as_tensor(np.stack(fut), device=dev).reshape(-1)
        self._table = torch.zeros((self.n_s * self.n_f, hist.shape[1]), dtype=torch.float64, device=dev)
        n_fut = len(fut[0])

        def body():
            X = scale_forward(hist[self._pidx0 + self._n], self.scale_max, self.scale_min).float()
            Y = scale_it_back(model_predict(dev, self.model, X, n_fut), self.scale_max, self.scale_min)
            self._table[self._fidx] = Y.reshape(-1, hist.shape[1]).double()

        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):  # capture wants the ops to have run once on a side stream
            self._n.fill_(self.n_p * self.n_s)
            body()
        torch.cuda.current_stream(dev).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            body()
        self._graph, self._key = graph, (hist.data_ptr(), tuple(hist.shape))
        return True

    @property
    def backend(self):
        return "native HIP" if self._use_native else ("PyTorch-ROCm, HIP graph" if self._use_graph else "PyTorch-ROCm")

    def __call__(self, n, hist):
        if self._use_native and hist.is_cuda:
            if self._native is None or self._native.device_index != hist.device.index:
                try:
                    self._native = NativePredictor(self.model, self.n_p, self.n_f, self.n_s, hist.device.index)
                except (_lib_error(), ValueError) as exc:
                    # a shape the library's kernels do not take (hidden size above 128, n_past * hidden beyond one
                    # workgroup's LDS, ...): said once, then the PyTorch-ROCm route, which takes any shape
                    import warnings

                    warnings.warn(f"native predictor refused this model ({exc}); using the PyTorch-ROCm route")
                    self._use_native = False
                    return self(n, hist)
                self._native_table = torch.empty((self.n_s * self.n_f, hist.shape[1]), dtype=torch.float64,
                                                 device=hist.device)
            return self._native.predict(n, hist, self.scale_max, self.scale_min, self._native_table)
        if self._use_graph and hist.is_cuda and self._graph is None:
            from . import hip_graphs

            self._use_graph = hip_graphs.replays_are_trustworthy(hist.device)  # (cached per device)
        if not (self._use_graph and hist.is_cuda):
            return self._eager(n, hist)
        if self._graph is not None and self._key != (hist.data_ptr(), tuple(hist.shape)):
            self._graph = None  # another history tensor: capture again
            self._calls = 0
        if self._graph is None:
            self._calls += 1
            if self._calls <= self._warmup or n < self.n_p * self.n_s or not self._capture(hist):
                return self._eager(n, hist)
        self._n.fill_(int(n))
        self._graph.replay()
        return self._table
