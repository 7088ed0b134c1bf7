// HIP side of driver_backend.hpp.
#include <hip/hip_runtime.h>

#include "driver_backend.hpp"
#include "stream_apart.hpp"

namespace drvdev {
int device_count() {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}
bool get_device(int *device) { return hipGetDevice(device) == hipSuccess; }
bool set_device(int device) { return hipSetDevice(device) == hipSuccess; }
bool stream_create_apart(void **stream, void *const *others, int n_others) {
    hipStream_t s = nullptr;
    if (spd::create_stream_apart(&s, reinterpret_cast<const hipStream_t *>(others), n_others, hipStreamDefault) != hipSuccess) return false;
    *stream = s;
    return true;
}
void stream_destroy(void *stream) {
    if (stream) (void)hipStreamDestroy(static_cast<hipStream_t>(stream));
}
bool device_synchronize() { return hipDeviceSynchronize() == hipSuccess; }
bool null_stream_synchronize() { return hipStreamSynchronize(nullptr) == hipSuccess; }
}  // namespace drvdev
