// Set-up kernels (saa_setup.hip): lumped mass, pre-assembled load and shortest edge of a set of elements on the GPU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace saa {

// Host in, host out (set-up runs once); any output may be null.  Works on the current device, null stream.
hipError_t setup_fields(int32_t n_nodes, int32_t n_elems, const double *xyz_host, const int32_t *tets_host, double rho, double fz,
                        double *lumped_host, double *fpre_host, double *min_edge_host);

}  // namespace saa
