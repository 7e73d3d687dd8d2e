"""Counterpart of the reference's ``Tools/DNN_tools.py``: the LSTM encoder-decoder, its recursive prediction and
scaling helpers (inference side, ``predictor.py``) and the data preparation / training functions the reference's
``Model_training.py`` and ``Online_predictor.py:129-136`` import from this module (``training.py``)."""
from __future__ import annotations

import torch
from torch.utils.data import Dataset

from .. import results_io as _rio
from ..predictor import (LSTM_Decoder, LSTM_Encoder, LSTM_encoder_decoder, model_predict,  # noqa: F401
                         scale_forward, scale_it_back, scaling_constants)
from ..training import model_test, model_train, windowed_dataset  # noqa: F401


class MyDataset(Dataset):
    """``(x[i], y[i])`` pairs for a ``DataLoader`` (``DNN_tools.py:239-255``); first dimension = sample."""

    def __init__(self, x, y):
        super().__init__()
        if x.shape[0] != y.shape[0]:
            raise AssertionError("x and y hold different numbers of samples")
        self.x, self.y = x, y

    def __len__(self):
        return self.y.shape[0]

    def __getitem__(self, index):
        return self.x[index], self.y[index]


def Scale_to_zero_one(X, Y):
    """Joint affine map of inputs and targets to [-1, 0] (``DNN_tools.py:259-269``).  Returns
    ``(X, Y, scale_max, scale_min)`` with the two constants as 0-dim tensors, like the reference (its callers take
    ``.item()``, ``Online_predictor.py:135-136``)."""
    scale_min, scale_max = torch.minimum(X.min(), Y.min()), torch.maximum(X.max(), Y.max())
    return (X - scale_max) / (-scale_min + scale_max), (Y - scale_max) / (-scale_min + scale_max), scale_max, scale_min


def Dis_data_filtered_subset_coronary(device, input_size, filter_size, n_past, n_future, Path, cut_off):
    """``DNN_tools.py:284-313``: the ``Displacement`` dataset ``(input_size, n_steps)`` at ``Path`` (HDF5, or the
    ``.npz`` stand-in ``results_io`` writes where h5py is absent) -> first ``cut_off`` share of the steps, every
    ``filter_size``-th of them, cut into all windows of ``n_past`` inputs + ``n_future`` targets; fp32 tensors
    ``(groups, n_past, input_size)``, ``(groups, n_future, input_size)`` on ``device``."""
    data = _rio.load_displacement(Path)
    if data.shape[0] != input_size:
        raise ValueError(f"{Path}: {data.shape[0]} shared dofs, expected input_size = {input_size}")
    return windowed_dataset(data, filter_size, n_past, n_future, cut_off, device)
