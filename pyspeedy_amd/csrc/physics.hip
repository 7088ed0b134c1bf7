#include <hip/hip_runtime.h>
#include "../../include/pyspeedy_amd.h"
#include "device_tables.hpp"
namespace spd {
hipError_t run_physics(const DeviceTables &T, const spd_physics_args &a, int nmembers, hipStream_t s) {
    return hipErrorNotSupported;
}
}
