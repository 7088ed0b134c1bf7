// Streams that really run side by side.
// HIP hands the streams of a process out over a few hardware queues (GPU_MAX_HW_QUEUES, default 4): a new stream goes to the
// queue with the fewest streams on it, ties broken arbitrarily, idle streams counted.  Two streams on one hardware queue are
// one queue: their kernels run strictly one after the other.  For the member groups of a model step, and for the device models
// a parallel_step drives side by side, that is the difference between 0.25 and 0.32 ms per step at 64 members -- and which it is
// depends on what else the process has created before (two idle streams were enough: tools/experiments/r04_idle_streams.py; a
// host's own stream pools and a communication library's streams are the usual case).  Streams of the high-priority class or
// with a CU mask get queues of their own but overlap less (0.274); the pool's queues are the ones to be on -- different ones.
// create_stream_apart creates a stream and MEASURES whether it overlaps with each of `others` (a 60 us sleeping kernel on
// both: together they take as long as one, or as long as two); a stream that does not is replaced -- the replacement is created
// while the rejected ones still hold their places, so it lands on another queue -- at most kApartTries times.
#pragma once
#include <hip/hip_runtime.h>

namespace spd {
constexpr int kApartTries = 4;
// flags: hipStreamDefault or hipStreamNonBlocking.  *apart (may be null): whether the stream was MEASURED to overlap with every one
// of `others` in the end -- false also when one of them was busy and could not be measured against (it is not made to wait),
// or when the measurement is switched off (PYSPEEDY_AMD_STREAMS_APART=0): nothing is claimed that was not seen.  The current
// device must be the streams' device.  Durations come from HIP events on the streams, not from the host's clock.
hipError_t create_stream_apart(hipStream_t *out, const hipStream_t *others, int n_others, unsigned flags, bool *apart = nullptr);
}  // namespace spd
