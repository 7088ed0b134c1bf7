// A call that may never return, with a bound on how long the caller waits for it.
//
// The one-process ensemble hands the shared boundary fields from GPU to GPU with an RCCL broadcast (model.hip:
// spd_model_broadcast_vars).  RCCL's single-process initialisation (ncclCommInitAll) and its group call talk to every GPU of
// the node; on a node where one link or one device does not answer they block for good, and a host that made the call on its
// own thread would block with them.  run_bounded makes the call on a thread of its own and waits for it for `seconds` only:
// when the time is up the caller goes on (and takes the point-to-point path); the thread is left behind, detached, with
// everything it touches kept alive by the shared state it owns -- so `fn` must capture by VALUE.
//
// Host-only C++, no HIP: tests/sanitize/driver_sanitize.cpp runs it under ASan / UBSan / TSan with a call that never returns.
#pragma once
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

namespace spd {

struct BoundedResult {
    bool finished = false;  // the call returned inside the bound
    int rc = 0;             // ... with this value
};

inline BoundedResult run_bounded(std::function<int()> fn, double seconds) {
    struct Shared {
        std::mutex m;
        std::condition_variable cv;
        bool done = false;
        int rc = 0;
        std::function<int()> fn;
    };
    auto shared = std::make_shared<Shared>();
    shared->fn = std::move(fn);
    std::thread([shared] {
        const int rc = shared->fn();
        std::lock_guard<std::mutex> lock(shared->m);
        shared->rc = rc;
        shared->done = true;
        shared->cv.notify_all();
    }).detach();
    BoundedResult out;
    std::unique_lock<std::mutex> lock(shared->m);
    out.finished = shared->cv.wait_for(lock, std::chrono::duration<double>(seconds), [&] { return shared->done; });
    if (out.finished) out.rc = shared->rc;
    return out;
}

}  // namespace spd
