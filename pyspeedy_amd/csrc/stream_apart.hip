// See stream_apart.hpp.
#include "stream_apart.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace spd {
namespace {
__global__ void sleep_kernel(int rounds) {
    for (int i = 0; i < rounds; ++i) __builtin_amdgcn_s_sleep(127);  // 127 x 64 cycles
}
constexpr int kRounds = 18;  // about 60 us at 2.4 GHz

// Time from the start of the sleeping kernel on `a` to the end of the later of the two kernels (on `a` and, at the same time, on
// `b`), by HIP events on the streams themselves -- the host's clock would add its own launch and wake-up jitter to a 60 us
// measurement that is compared against a factor of 1.5.  The smallest of three.
struct SleepEvents {
    hipEvent_t start = nullptr, end_a = nullptr, end_b = nullptr;
    hipError_t create() {
        hipError_t e = hipEventCreate(&start);
        if (e == hipSuccess) e = hipEventCreate(&end_a);
        if (e == hipSuccess) e = hipEventCreate(&end_b);
        return e;
    }
    ~SleepEvents() {
        if (start) (void)hipEventDestroy(start);
        if (end_a) (void)hipEventDestroy(end_a);
        if (end_b) (void)hipEventDestroy(end_b);
    }
};

hipError_t sleep_time(SleepEvents &ev, hipStream_t a, hipStream_t b, double *seconds) {
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        hipError_t e = hipEventRecord(ev.start, a);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(sleep_kernel, dim3(1), dim3(64), 0, a, kRounds);
        if (b) hipLaunchKernelGGL(sleep_kernel, dim3(1), dim3(64), 0, b, kRounds);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipEventRecord(ev.end_a, a);
        if (e == hipSuccess && b) e = hipEventRecord(ev.end_b, b);
        if (e == hipSuccess) e = hipStreamSynchronize(a);
        if (e == hipSuccess && b) e = hipStreamSynchronize(b);
        float ms_a = 0.0f, ms_b = 0.0f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms_a, ev.start, ev.end_a);
        if (e == hipSuccess && b) e = hipEventElapsedTime(&ms_b, ev.start, ev.end_b);
        if (e != hipSuccess) return e;
        best = std::min(best, 1e-3 * static_cast<double>(std::max(ms_a, ms_b)));
    }
    *seconds = best;
    return hipSuccess;
}
}  // namespace

hipError_t create_stream_apart(hipStream_t *out, const hipStream_t *others, int n_others, unsigned flags, bool *apart) {
    static const int mode = getenv("PYSPEEDY_AMD_STREAMS_APART") ? atoi(getenv("PYSPEEDY_AMD_STREAMS_APART")) : 1;  // 0: off, 2: report
    const bool measure = mode != 0;
    hipStream_t cand = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&cand, flags);
    if (e != hipSuccess) return e;
    bool ok = true, all_measured = true;
    hipStream_t rejected[kApartTries];
    int n_rejected = 0;
    SleepEvents ev;
    if (measure && n_others > 0) e = ev.create();
    if (measure && n_others > 0 && e == hipSuccess) {
        for (int attempt = 0;; ++attempt) {
            double alone = 0.0;
            e = sleep_time(ev, cand, nullptr, &alone);  // (also the first launch on the stream: its queue exists from here on)
            if (e == hipSuccess) e = sleep_time(ev, cand, nullptr, &alone);
            ok = true;
            all_measured = true;
            for (int i = 0; i < n_others && ok && e == hipSuccess; ++i) {
                if (!others[i] || others[i] == cand) continue;
                if (hipStreamQuery(others[i]) != hipSuccess) {  // busy: it cannot be measured now (and is not made to wait)
                    (void)hipGetLastError();
                    all_measured = false;  // ... so nothing is claimed about it: the verdict below is "not known to be apart"
                    if (mode == 2)
                        std::fprintf(stderr, "create_stream_apart: attempt %d, against stream %d of %d: busy, not measured\n", attempt, i, n_others);
                    continue;
                }
                double both = 0.0;
                e = sleep_time(ev, cand, others[i], &both);
                ok = both < 1.5 * alone;
                if (mode == 2)
                    std::fprintf(stderr, "create_stream_apart: attempt %d, against stream %d of %d: alone %.1f us, both %.1f us -> %s\n", attempt, i,
                                 n_others, alone * 1e6, both * 1e6, ok ? "side by side" : "one queue");
            }
            if (e != hipSuccess || ok || attempt + 1 == kApartTries) break;
            // The replacement is created while every candidate rejected so far still holds its place: HIP puts a new stream on
            // the queue with the fewest streams, and without the place-holders it alternates between the queues of two of
            // the `others` (seen with three member groups behind one idle stream).
            rejected[n_rejected++] = cand;
            cand = nullptr;
            e = hipStreamCreateWithFlags(&cand, flags);
            if (e != hipSuccess) break;
        }
    }
    for (int i = 0; i < n_rejected; ++i) (void)hipStreamDestroy(rejected[i]);
    if (e != hipSuccess) {
        if (cand) (void)hipStreamDestroy(cand);
        return e;
    }
    if (apart) *apart = ok && all_measured && (measure || n_others == 0);  // (side by side with ALL of them, and measured to be)
    *out = cand;
    return hipSuccess;
}
}  // namespace spd
