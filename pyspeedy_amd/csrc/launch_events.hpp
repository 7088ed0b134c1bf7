// Kernel launch with optional profiling events attached to the dispatch itself.
// spd_model_profile wants the duration of single kernels.  A pair of hipEventRecord calls around a launch puts two marker
// packets into the queue and reads 4-7 us more than the kernel takes (and keeps the next kernel from starting back to back);
// events handed to hipExtLaunchKernel take the begin / end time stamps of the kernel's own dispatch packet instead -- the
// numbers then agree with rocprofv3's kernel trace and the launch sequence is not disturbed.  The model announces the events of
// the NEXT launch of the calling thread (announce_launch_events); the launch sites of the step kernels go through
// spd::launch, which picks them up.
#pragma once
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

namespace spd {

struct LaunchEvents {
    hipEvent_t start = nullptr, stop = nullptr;
};
LaunchEvents &pending_launch_events();  // thread-local slot (model.hip)

template <typename K, typename... A>
inline void launch(K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t s, A... args) {
    LaunchEvents &ev = pending_launch_events();
    if (ev.start && ev.stop) {
        hipExtLaunchKernelGGL(kernel, grid, block, lds, s, ev.start, ev.stop, 0, args...);
        ev = LaunchEvents{};
    } else {
        hipLaunchKernelGGL(kernel, grid, block, lds, s, args...);
    }
}

}  // namespace spd
