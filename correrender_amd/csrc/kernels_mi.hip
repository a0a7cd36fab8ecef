// kernels_mi.hip -- placeholder until the mutual-information estimators land.
#include "crf_internal.h"
namespace crf {
hipError_t launch_mi_binned(const float* const*, int, size_t, const float*, const BinnedArgs&, float*, float*,
                            hipStream_t, hipEvent_t, hipEvent_t, LaunchInfo*) { return hipErrorNotSupported; }
hipError_t launch_mi_kraskov(const float* const*, int, size_t, const float*, const KraskovArgs&, float*, float*,
                             hipStream_t, hipEvent_t, hipEvent_t, LaunchInfo*) { return hipErrorNotSupported; }
}
