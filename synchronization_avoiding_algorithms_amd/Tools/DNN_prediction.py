"""Counterpart of the reference's ``Tools/DNN_prediction.py``."""
from ..predictor import call_model, encoder_decoder_predictor  # noqa: F401
