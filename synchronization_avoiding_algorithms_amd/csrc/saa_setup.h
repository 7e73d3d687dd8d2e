// Set-up kernels (saa_setup.hip): lumped mass, pre-assembled load and shortest edge of a set of elements on the GPU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace saa {

// Host in, host out (set-up runs once); any output may be null.  Works on the current device, null stream.
hipError_t setup_fields(int32_t n_nodes, int32_t n_elems, const double *xyz_host, const int32_t *tets_host, double rho, double fz,
                        double *lumped_host, double *fpre_host, double *min_edge_host);

// Device-to-device copy rate (read + written bytes per second) of a 16-byte-per-lane copy kernel over two buffers of
// n_bytes each, `reps` timed launches after two warm ones; current device, null stream.
hipError_t copy_bandwidth(int device, int64_t n_bytes, int reps, double *bytes_per_s);

}  // namespace saa
