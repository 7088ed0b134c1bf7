// Streaming-rate probe of the device (spd_stream_probe, include/pyspeedy_amd.h): what a kernel of a given SHAPE can move
// through HBM on this box, measured with kernels of this library -- not a runtime blit, not a framework's elementwise op.
//
// The step's kernels are priced against 8 TB/s (the contract's peak); how much of the distance to that figure is the
// kernels' own and how much is the memory system's can only be read against a kernel that does nothing but move bytes in the
// same shape.  The shapes that matter here:
//   * the mix of input and output streams (a copy is 1 : 1, the column kernel 2 : 1, a transform S + G in either direction),
//   * bytes per lane and access (the column kernel: one fp64 per lane, 512 B per wave-instruction; the transforms: 16 B),
//   * how many wavefronts a SIMD holds (the column kernel: two, at 256 VGPRs; a plain copy: eight),
//   * how much a wavefront has in flight before it first has to wait (loads issued back to back before the first store),
//   * how long a wavefront lives (one row and out, or the column kernel's 243 rows),
//   * the non-temporal hint on both sides.
// One kernel template covers them; the probe allocates its own buffers, times every launch by the time stamps of its dispatch
// packet (hipExtLaunchKernel events, as spd_model_profile does) and returns mean and minimum.
//
// No counterpart in the reference: measurement infrastructure ("report measured copy / triad bandwidth
// on the box and use the spec peak for the contract fraction").
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <mutex>
#include <string>
#include <vector>

#include "../../include/pyspeedy_amd.h"
#include "context.hpp"

namespace spd {
namespace {

constexpr int kLanes = 64;
typedef double double_x2 __attribute__((ext_vector_type(2)));  // 16 B per lane: one global_load_dwordx4

template <typename V> __device__ __forceinline__ V probe_add(V a, V b);
template <> __device__ __forceinline__ double probe_add(double a, double b) { return a + b; }
template <> __device__ __forceinline__ double_x2 probe_add(double_x2 a, double_x2 b) { return a + b; }
template <typename V> __device__ __forceinline__ V probe_value(size_t i);
template <> __device__ __forceinline__ double probe_value(size_t i) { return static_cast<double>(i); }
template <> __device__ __forceinline__ double_x2 probe_value(size_t i) { return double_x2{static_cast<double>(i), 1.0}; }
__device__ __forceinline__ bool probe_is(double v, double x) { return v == x; }
__device__ __forceinline__ bool probe_is(double_x2 v, double x) { return v.x == x && v.y == x; }

// One wavefront per workgroup.  A wavefront owns `iters` x U rows (a row = 64 lanes x sizeof(V), contiguous) of every stream;
// per iteration it requests the U x NR input rows back to back, then combines and stores U x NW output rows.  The input
// streams lie `stream_len` elements apart in src, the output streams in dst.  Where a wavefront's rows lie inside a stream is
// the layout: its rows `row_stride` elements apart, the first rows of consecutive wavefronts `wave_stride` apart --
//   layout 0 (a wavefront walks a chunk of its own): row_stride = 64, wave_stride = rows x 64;
//   layout 1 (the column kernel: row r of every wavefront lies in array r, consecutive wavefronts side by side in each array):
//            row_stride = wavefronts x 64, wave_stride = 64.
// Dynamic LDS is the occupancy limiter only (20 KB per workgroup: 8 wavefronts per CU).
template <typename V, int NR, int NW, int U, bool NT>
__global__ __launch_bounds__(kLanes) void stream_probe_kernel(const V *__restrict__ src, V *__restrict__ dst, size_t stream_len,
                                                             size_t row_stride, size_t wave_stride, int iters, double never) {
    const size_t first = static_cast<size_t>(blockIdx.x) * wave_stride + threadIdx.x;
    V sink = probe_value<V>(0);
    for (int it = 0; it < iters; ++it) {
        const size_t at = first + static_cast<size_t>(it) * U * row_stride;
        V r[U][NR > 0 ? NR : 1];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int s = 0; s < NR; ++s) {
                const V *p = src + s * stream_len + at + u * row_stride;
                r[u][s] = NT ? __builtin_nontemporal_load(p) : *p;
            }
        __builtin_amdgcn_sched_barrier(0);  // every load of the batch is in flight before the first use
#pragma unroll
        for (int u = 0; u < U; ++u) {
            V acc = NR > 0 ? r[u][0] : probe_value<V>(at + u * row_stride);
#pragma unroll
            for (int s = 1; s < NR; ++s) acc = probe_add(acc, r[u][s]);
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                V *p = dst + w * stream_len + at + u * row_stride;
                if (NT) __builtin_nontemporal_store(acc, p);
                else *p = acc;
            }
            if (NW == 0) sink = probe_add(sink, acc);
        }
    }
    if (NW == 0 && probe_is(sink, never)) dst[threadIdx.x] = sink;  // (read-only shape: keeps the loads alive, never stores)
}

template <typename V, int NR, int NW, int U, bool NT>
hipError_t probe_launch(const void *src, void *dst, size_t stream_len, size_t row_stride, size_t wave_stride, int iters,
                        unsigned blocks, unsigned lds, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    hipExtLaunchKernelGGL((stream_probe_kernel<V, NR, NW, U, NT>), dim3(blocks), dim3(kLanes), lds, s, e0, e1, 0,
                          static_cast<const V *>(src), static_cast<V *>(dst), stream_len, row_stride, wave_stride, iters, -1.0);
    return hipGetLastError();
}

using Launcher = hipError_t (*)(const void *, void *, size_t, size_t, size_t, int, unsigned, unsigned, hipStream_t, hipEvent_t,
                                hipEvent_t);

template <typename V, int NR, int NW, int U>
Launcher pick_nt(int nt) {
    return nt ? probe_launch<V, NR, NW, U, true> : probe_launch<V, NR, NW, U, false>;
}
template <typename V, int NR, int NW>
Launcher pick_u(int u, int nt) {
    switch (u) {
        case 1: return pick_nt<V, NR, NW, 1>(nt);
        case 2: return pick_nt<V, NR, NW, 2>(nt);
        case 4: return pick_nt<V, NR, NW, 4>(nt);
        case 8: return pick_nt<V, NR, NW, 8>(nt);
        case 16: return pick_nt<V, NR, NW, 16>(nt);
        default: return nullptr;
    }
}
template <typename V>
Launcher pick_mix(int nr, int nw, int u, int nt) {
    switch (nr * 10 + nw) {
        case 11: return pick_u<V, 1, 1>(u, nt);  // copy
        case 21: return pick_u<V, 2, 1>(u, nt);  // the column kernel's mix
        case 32: return pick_u<V, 3, 2>(u, nt);
        case 10: return pick_u<V, 1, 0>(u, nt);  // read only
        case 1: return pick_u<V, 0, 1>(u, nt);   // write only
        default: return nullptr;
    }
}

}  // namespace
}  // namespace spd

extern "C" int spd_stream_probe(spd_handle h, const spd_stream_probe_args *a, double *mean_us, double *min_us,
                                uint64_t *bytes_moved, uint64_t *workgroups) {
    using namespace spd;
    if (!h || !a || !mean_us || !min_us) return spd_set_error(SPD_E_ARG, "spd_stream_probe: null argument");
    if (a->layout != 0 && a->layout != 1) return spd_set_error(SPD_E_ARG, "spd_stream_probe: layout is 0 or 1");
    if (a->lane_bytes != 8 && a->lane_bytes != 16) return spd_set_error(SPD_E_ARG, "spd_stream_probe: lane_bytes is 8 or 16");
    if (a->waves_per_simd < 1 || a->waves_per_simd > 8)
        return spd_set_error(SPD_E_ARG, "spd_stream_probe: waves_per_simd is 1 ... 8");
    if (a->rows_per_wave < 1 || a->reps < 1 || a->reps > 1000 || a->total_bytes == 0 || a->total_bytes > (64ull << 30))
        return spd_set_error(SPD_E_ARG, "spd_stream_probe: rows_per_wave, reps (1 ... 1000) and total_bytes (up to 64 GiB) must be positive");
    const Launcher go = a->lane_bytes == 8 ? pick_mix<double>(a->reads, a->writes, a->in_flight, a->nontemporal)
                                           : pick_mix<double_x2>(a->reads, a->writes, a->in_flight, a->nontemporal);
    if (!go)
        return spd_set_error(SPD_E_ARG, "spd_stream_probe: reads : writes is one of 1:1, 2:1, 3:2, 1:0, 0:1 and in_flight one of 1, 2, 4, 8, 16");
    const int streams = a->reads + a->writes;
    const size_t row_bytes = static_cast<size_t>(kLanes) * a->lane_bytes;
    // rows of one stream a wavefront works through: rows_per_wave counts the rows of ALL its streams (the column kernel: 243)
    int iters = a->rows_per_wave / (streams * a->in_flight);
    if (iters < 1) iters = 1;
    const size_t rows_per_wave_stream = static_cast<size_t>(iters) * a->in_flight;
    size_t waves = a->total_bytes / (row_bytes * streams * rows_per_wave_stream);
    if (waves < 1) waves = 1;
    if (waves > 0x7fffffffull) return spd_set_error(SPD_E_ARG, "spd_stream_probe: too many workgroups");
    const size_t stream_len = waves * rows_per_wave_stream * kLanes;  // elements of V per stream
    const size_t row_stride = a->layout ? waves * kLanes : kLanes, wave_stride = a->layout ? kLanes : rows_per_wave_stream * kLanes;
    const size_t stream_bytes = stream_len * a->lane_bytes;
    // occupancy: 160 KiB of LDS per CU shared by the resident workgroups (one wavefront each, four SIMDs per CU)
    const unsigned lds = a->waves_per_simd >= 8 ? 0u : static_cast<unsigned>((160u * 1024u) / (4u * a->waves_per_simd)) & ~255u;

    int dev_now = 0;
    if (hipGetDevice(&dev_now) != hipSuccess || hipSetDevice(spd_device(h)) != hipSuccess)
        return spd_set_error(SPD_E_DEVICE, "spd_stream_probe: cannot select the handle's device");
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipStream_t s = nullptr;
    int rc = SPD_OK;
    std::string what;
    auto check = [&](hipError_t e, const std::string &where) {
        if (e != hipSuccess && rc == SPD_OK) {
            (void)hipGetLastError();
            rc = SPD_E_DEVICE;
            what = "spd_stream_probe: " + where + ": " + hipGetErrorString(e);
        }
        return e == hipSuccess;
    };
    // one buffer per context, grown on demand and kept (a 4 GB hipMalloc + hipFree per shape is slower than the probe itself);
    // input streams first, output streams behind them, each part on a 4 KiB boundary
    const size_t src_bytes = (a->reads * stream_bytes + 4095) & ~static_cast<size_t>(4095), dst_bytes = a->writes * stream_bytes + 4096;
    std::lock_guard<std::mutex> lock(h->probe_mutex);
    if (h->probe_bytes < src_bytes + dst_bytes) {
        if (h->probe_buf) (void)hipFree(h->probe_buf);
        h->probe_buf = nullptr;
        h->probe_bytes = 0;
        if (check(hipMalloc(&h->probe_buf, src_bytes + dst_bytes), "hipMalloc of " + std::to_string(src_bytes + dst_bytes) + " bytes") &&
            check(hipMemset(h->probe_buf, 0, src_bytes + dst_bytes), "hipMemset"))
            h->probe_bytes = src_bytes + dst_bytes;
    }
    char *src = static_cast<char *>(h->probe_buf), *dst = src + src_bytes;
    double sum = 0.0, best = 1e300;
    if (rc == SPD_OK && check(hipEventCreate(&e0), "hipEventCreate") && check(hipEventCreate(&e1), "hipEventCreate") &&
        check(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreate")) {
        for (int rep = -2; rep < a->reps && rc == SPD_OK; ++rep) {  // two untimed launches first
            if (!check(go(src, dst, stream_len, row_stride, wave_stride, iters, static_cast<unsigned>(waves), lds, s, e0, e1), "launch")) break;
            if (!check(hipEventSynchronize(e1), "hipEventSynchronize")) break;
            float ms = 0.f;
            if (!check(hipEventElapsedTime(&ms, e0, e1), "hipEventElapsedTime")) break;
            if (rep >= 0) {
                sum += ms * 1e3;
                if (ms * 1e3 < best) best = ms * 1e3;
            }
        }
    }
    if (s) (void)hipStreamDestroy(s);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipSetDevice(dev_now);
    if (rc != SPD_OK) return spd_set_error(rc, what);
    *mean_us = sum / a->reps;
    *min_us = best;
    if (bytes_moved) *bytes_moved = static_cast<uint64_t>(stream_bytes) * streams;
    if (workgroups) *workgroups = static_cast<uint64_t>(waves);
    return SPD_OK;
}
