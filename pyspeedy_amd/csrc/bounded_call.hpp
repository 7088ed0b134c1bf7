// A call that may never return, with a bound on how long the caller waits for it.
//
// The one-process ensemble hands the shared boundary fields from GPU to GPU with an RCCL broadcast (model.hip:
// spd_model_broadcast_vars).  RCCL's single-process initialisation (ncclCommInitAll) and its group call talk to every GPU of
// the node; on a node where one link or one device does not answer they block for good, and a host that made the call on its
// own thread would block with them.  run_bounded makes the call on a thread of its own and waits for it for `seconds` only:
// when the time is up the caller goes on (and takes the point-to-point path); the thread is left behind, detached, with
// everything it touches kept alive by the shared state it owns -- so `fn` must capture by VALUE.
//
// A thread that was left behind is still somewhere inside the call when the process ends, and exit() then runs the static
// destructors of every library in the process -- RCCL's and the HIP runtime's among them -- under its feet.  From the first call
// that ran out of time the process therefore leaves through _exit() once exit() is under way: an on_exit handler (registered then,
// so it runs before the destructors of everything constructed earlier) flushes the C streams and ends the process with the
// status exit() was given, without further teardown.  What Python or the host program do before they call exit() has happened.
//
// Host-only C++, no HIP: tests/sanitize/driver_sanitize.cpp runs it under ASan / UBSan / TSan with a call that never returns.
#pragma once
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <unistd.h>

namespace spd {

inline std::atomic<int> &abandoned_calls() {  // calls that ran out of time and whose threads were left behind
    static std::atomic<int> n{0};
    return n;
}

inline void leave_without_teardown_at_exit() {
    static std::once_flag once;
    std::call_once(once, [] {
        (void)on_exit([](int status, void *) {
            if (abandoned_calls().load() > 0) {
                std::fflush(nullptr);
                _exit(status);
            }
        }, nullptr);
    });
}

struct BoundedResult {
    bool finished = false;  // the call returned inside the bound
    int rc = 0;             // ... with this value
};

// (The caller watches an atomic flag with short sleeps instead of waiting on a condition variable: the wait is once per process,
// a fraction of a millisecond of latency does not matter to it, and a timed wait on the steady clock is pthread_cond_clockwait,
// which the ThreadSanitizer of this toolchain does not know -- it reports the worker's lock of the mutex as a double lock.)
inline BoundedResult run_bounded(std::function<int()> fn, double seconds) {
    struct Shared {
        std::atomic<bool> done{false};
        int rc = 0;
        std::function<int()> fn;
    };
    auto shared = std::make_shared<Shared>();
    shared->fn = std::move(fn);
    std::thread([shared] {
        shared->rc = shared->fn();
        shared->done.store(true, std::memory_order_release);
    }).detach();
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(seconds);
    BoundedResult out;
    while (!(out.finished = shared->done.load(std::memory_order_acquire)) && std::chrono::steady_clock::now() < deadline)
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    if (out.finished) {
        out.rc = shared->rc;
    } else {
        abandoned_calls().fetch_add(1);
        leave_without_teardown_at_exit();
    }
    return out;
}

}  // namespace spd
