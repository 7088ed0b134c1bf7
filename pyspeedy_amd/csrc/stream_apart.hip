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

// wall time of the sleeping kernel on `a` (and, at the same time, on `b`): the smallest of three
hipError_t sleep_time(hipStream_t a, hipStream_t b, double *seconds) {
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        const auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(sleep_kernel, dim3(1), dim3(64), 0, a, kRounds);
        if (b) hipLaunchKernelGGL(sleep_kernel, dim3(1), dim3(64), 0, b, kRounds);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(a);
        if (e == hipSuccess && b) e = hipStreamSynchronize(b);
        if (e != hipSuccess) return e;
        best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
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
    bool ok = true;
    hipStream_t rejected[kApartTries];
    int n_rejected = 0;
    if (measure && n_others > 0) {
        for (int attempt = 0;; ++attempt) {
            double alone = 0.0;
            e = sleep_time(cand, nullptr, &alone);  // (also the first launch on the stream: its queue exists from here on)
            if (e == hipSuccess) e = sleep_time(cand, nullptr, &alone);
            ok = true;
            for (int i = 0; i < n_others && ok && e == hipSuccess; ++i) {
                if (!others[i] || others[i] == cand) continue;
                if (hipStreamQuery(others[i]) != hipSuccess) {  // busy: it cannot be measured now (and is not made to wait)
                    (void)hipGetLastError();
                    continue;
                }
                double both = 0.0;
                e = sleep_time(cand, others[i], &both);
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
    if (apart) *apart = ok;
    *out = cand;
    return hipSuccess;
}
}  // namespace spd
