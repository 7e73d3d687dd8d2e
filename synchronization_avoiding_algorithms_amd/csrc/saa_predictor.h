// Shared-node predictor (saa_predictor.hip): the reference's LSTM encoder-decoder (Tools/DNN_tools.py:16-98) evaluated
// for all phase offsets of one prediction window (Tools/DNN_prediction.py:38-55) in four launches.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

namespace saa {

struct PredictorShape {
  int input_size;  // 3 * shared nodes of the rank (Online_predictor.py:126-129)
  int hidden;      // encoder width H; the decoder is 2H wide (DNN_tools.py:66)
  int n_past, n_future, filter;  // n_p, n_f, n_s of DNN_prediction.py:38
};

constexpr int kPredictorWeights = 22;  // tensors of the reference's state_dict, in its order (saa_hip.h)

struct Predictor;

// `weights`: host pointers, fp32, the state_dict's tensors in its own order.  Returns hipSuccess and *out, or an error
// (err filled when the arguments are at fault: hipErrorInvalidValue).
hipError_t predictor_create(int device, const PredictorShape &shape, const float *const *weights, Predictor **out,
                            std::string &err);
void predictor_destroy(Predictor *p);
const PredictorShape &predictor_shape(const Predictor *p);
int predictor_device(const Predictor *p);

// Rows [n - n_p*n_s, n) of the device history `hist` (row stride ld_hist doubles) -> the (n_s*n_f) x input_size table
// (row stride ld_table doubles) whose row k is the prediction for step n + k.  Enqueued on `st`; no host synchronisation.
hipError_t predictor_predict(Predictor *p, const double *hist, int64_t ld_hist, int64_t n, double scale_max, double scale_min,
                             double *table, int64_t ld_table, hipStream_t st);

// Pointwise part of one LSTM step and its backward (device pointers, fp32, gates in PyTorch's order i, f, g, o), enqueued
// on `st` - for the training pass of training.py.
hipError_t lstm_cell_forward(int32_t B, int32_t D, const float *gates, const float *c_prev, float *h, float *c, float *act,
                             float *tanh_c, hipStream_t st);
hipError_t lstm_cell_backward(int32_t B, int32_t D, const float *act, const float *tanh_c, const float *c_prev, const float *dh,
                              const float *dc_next, float *dgates, float *dc_prev, hipStream_t st);

// A whole LSTM recurrence of the training pass and its backward, one launch each (width 50 or 100; else
// hipErrorInvalidValue): see saa_hip.h.
hipError_t lstm_rec_forward(int32_t B, int32_t T, int32_t HP, int32_t reverse, const float *pre, const float *h0, const float *c0,
                            const float *W, float *H, float *c_all, float *act, float *tanhc, hipStream_t st);
hipError_t lstm_rec_backward(int32_t B, int32_t T, int32_t HP, int32_t reverse, const float *dH, const float *dcT, const float *c0,
                             const float *W, const float *c_all, const float *act, const float *tanhc, float *dpre, float *dh0,
                             float *dc0, hipStream_t st);

// sums[0..2] += (mse, 1 - mse / var(y), 1 - mse / mean(y^2)) of `out` against `y` (n fp32 elements each); `part` = three
// doubles of scratch that are zero on entry and left zero.
hipError_t train_stats(int64_t n, const float *out, const float *y, double *part, double *sums, hipStream_t st);

}  // namespace saa
