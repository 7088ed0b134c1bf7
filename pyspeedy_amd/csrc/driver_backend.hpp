// What the outer boundary (driver.cpp) needs from the GPU runtime, and nothing else: driver.cpp is plain C++ -- containers,
// registry, grouping, pending-step tokens -- over the batched device model (spd_model_*, include/pyspeedy_amd.h) and these few
// device calls.  The product links driver_backend_hip.hip; tests/sanitize/ links a stub of this header and of the spd_model_*
// functions instead, so that the same driver.cpp runs under AddressSanitizer / UndefinedBehaviorSanitizer / ThreadSanitizer on
// a machine without a GPU.
#pragma once
#include <string>

namespace drvdev {
int device_count();
bool get_device(int *device);
bool set_device(int device);
// a blocking stream on the current device that runs side by side with each of `others` (streams of the same device, idle now):
// not on a hardware queue one of them is on (stream_apart.hpp)
bool stream_create_apart(void **stream, void *const *others, int n_others);
void stream_destroy(void *stream);
bool device_synchronize();          // the current device
bool null_stream_synchronize();

// The driver switches the calling thread's HIP device whenever it touches a container that lives elsewhere (one process over
// several GPUs).  Every entry point that may do so holds one of these: the device that was current when the caller came in
// is current again when it leaves (a host that allocates its own buffers with "the current device", torch included, must not
// find them on another GPU because it stepped an ensemble in between).
struct DeviceGuard {
    int saved = -1;
    DeviceGuard() {
        if (!get_device(&saved)) saved = -1;
    }
    ~DeviceGuard() {
        int now = -1;
        if (saved >= 0 && get_device(&now) && now != saved) (void)set_device(saved);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};
}  // namespace drvdev

// records the message returned by spd_last_error() (thread-local) and returns `code` (capi.hip)
int spd_set_error(int code, const std::string &msg);
