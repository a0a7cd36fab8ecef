// kernels_rank.hip -- placeholder until the sort-based estimators land (Spearman, Kendall).
#include "crf_internal.h"
namespace crf {
hipError_t launch_spearman(const float* const*, int, size_t, const float*, float*, float*, hipStream_t, hipEvent_t,
                           hipEvent_t, LaunchInfo*) { return hipErrorNotSupported; }
hipError_t launch_kendall(const float* const*, int, size_t, const float*, float*, float*, hipStream_t, hipEvent_t,
                          hipEvent_t, LaunchInfo*) { return hipErrorNotSupported; }
}
