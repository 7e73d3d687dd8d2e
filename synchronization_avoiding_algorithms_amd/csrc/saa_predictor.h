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

}  // namespace saa
