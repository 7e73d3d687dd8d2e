"""ORACLE (test infrastructure, NOT product code) - CPU restatement of the shared-node predictor.

Follows /root/reference ``Tools/DNN_tools.py:16-98`` (encoder/decoder modules and the state_dict
key layout), ``:212-234`` (``model_predict``: batch-1 encode + recursive decode), ``:259-279``
(scaling to [-1,0] and back) and ``Tools/DNN_prediction.py:18-55`` (``call_model``,
``encoder_decoder_predictor``: one sequential batch-1 pass per phase offset, fp32 on the CPU).

Deliberately kept sequential/batch-1 like the reference so that it checks the product's batched
GPU path rather than sharing its structure.  Pinned by ``tests/golden/predictor_*.npz`` (written by
the real reference functions with seeded weights).  The third-party arithmetic underneath is
``torch.nn.LSTM`` / ``torch.nn.Linear`` of the installed torch, as in the reference.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn


class _Encoder(nn.Module):
    def __init__(self, input_size, hidden_size):
        super().__init__()
        self.hidden_size = hidden_size
        # DNN_prediction.py:21-24: always 2 layers, bidirectional, dropout 0
        self.lstm_encoder = nn.LSTM(input_size=input_size, hidden_size=hidden_size, num_layers=2,
                                    batch_first=True, dropout=0.0, bidirectional=True)

    def forward(self, x):
        _, (hn, cn) = self.lstm_encoder(x)
        n = x.shape[0]
        hn = hn.view(2, 2, n, self.hidden_size)[-1]  # last layer: (direction, N, H)
        cn = cn.view(2, 2, n, self.hidden_size)[-1]
        return (torch.cat((hn[0], hn[1]), 1).unsqueeze(0),
                torch.cat((cn[0], cn[1]), 1).unsqueeze(0))


class _Decoder(nn.Module):
    def __init__(self, input_size, hidden_size):
        super().__init__()
        self.lstm_decoder = nn.LSTM(input_size=input_size, hidden_size=2 * hidden_size, num_layers=1,
                                    batch_first=True, bidirectional=False)
        self.fc = nn.Linear(2 * hidden_size, input_size)

    def forward(self, x, h, c):
        out, (h, c) = self.lstm_decoder(x.unsqueeze(1), (h, c))
        return self.fc(out.squeeze(1)), h, c


class OracleSeq2Seq(nn.Module):
    """Same parameter names as ``LSTM_encoder_decoder`` (DNN_tools.py:85-98)."""

    def __init__(self, input_size, hidden_size):
        super().__init__()
        self.encoder = _Encoder(input_size, hidden_size)
        self.decoder = _Decoder(input_size, hidden_size)


def load_model(input_size, hidden_size, state_dict):
    model = OracleSeq2Seq(input_size, hidden_size)
    model.load_state_dict(state_dict)
    return model.eval()


def predict_one(model, X, n_future):
    """``model_predict`` (DNN_tools.py:212-234): X (n_past, in) fp32 -> (n_future, in)."""
    with torch.no_grad():
        X = X.unsqueeze(0)
        h, c = model.encoder(X)
        out = torch.zeros((n_future, X.shape[2]))
        inp = X[:, -1, :]
        for i in range(n_future):
            y, h, c = model.decoder(inp, h, c)
            out[i, :] = y
            inp = y
    return out


def predictor_table(n, model, n_p, n_f, n_s, input_size, d_sol, scale_max, scale_min):
    """``encoder_decoder_predictor`` (DNN_prediction.py:38-55)."""
    NF = np.zeros((n_s * n_f, input_size))
    for i in range(n_s):
        past = np.arange(i + n - n_p * n_s, i + n - 1, n_s)
        future = np.arange(i + n, n + i + n_f * n_s - 1, n_s)
        X = (d_sol[past, :] - scale_max) / (-scale_min + scale_max)
        Y = predict_one(model, torch.from_numpy(X).float(), n_f)
        Y = Y * (scale_max - scale_min) + scale_max
        NF[future - n, :] = Y.numpy()
    return NF


def scaling_constants(displacement_shared, filter_size, n_past, n_future, cut_off):
    """Online_predictor.py:130-136 via DNN_tools.py:284-313 and :259-269.

    ``displacement_shared``: (input_size, n_steps) as stored by Shared_extraction.py:22-40.
    The windows together cover every retained sample, so max/min over (X, Y) equals max/min over
    the filtered fp32 series; the windowing is still done to mirror the reference exactly.
    """
    data = np.asarray(displacement_shared).transpose()
    data = data[0:int(cut_off * len(data)), :][0::filter_size, :]
    data = torch.from_numpy(np.ascontiguousarray(data)).float()
    groups = data.shape[0] - n_future - n_past + 1
    X = torch.zeros((groups, n_past, data.shape[1]))
    Y = torch.zeros((groups, n_future, data.shape[1]))
    c = 0
    for i in range(n_past, data.shape[0] - n_future + 1):
        X[c] = data[i - n_past:i, :]
        Y[c] = data[i:i + n_future, :]
        c += 1
    smin = min(X.min(), Y.min())
    smax = max(X.max(), Y.max())
    return smax.item(), smin.item()
