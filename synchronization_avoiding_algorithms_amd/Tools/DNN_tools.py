"""Counterpart of the reference's ``Tools/DNN_tools.py`` (inference side)."""
from ..predictor import (LSTM_Decoder, LSTM_Encoder, LSTM_encoder_decoder, model_predict,  # noqa: F401
                         scale_forward, scale_it_back, scaling_constants)
