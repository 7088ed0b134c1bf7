// TEST INFRASTRUCTURE: a stand-in for the GPU side of the outer boundary, so that pyspeedy_amd/csrc/driver.cpp -- the product's
// own file, unchanged -- runs under AddressSanitizer / UndefinedBehaviorSanitizer / ThreadSanitizer on a machine without a GPU.
// It implements (a) csrc/driver_backend.hpp over a number of pretend devices and (b) the spd_model_* functions driver.cpp calls
// over a toy model whose "state" is whatever was stored with spd_model_set plus two numbers per member that a step advances:
//     olr[0]  steps taken,    olr[1]  a fingerprint  f <- f * A + (step counter of the model before the step) + C
// Both depend on nothing but the member's own history, so a container's values after n steps are predictable whatever the
// driver did to it in between (gathered, split, copied, stepped in another model) -- that is what driver_sanitize.cpp checks.
// olr[2] != 0 makes the range check of that member fail (-2).  Nothing here is numerics; nothing of it is linked into the product.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pyspeedy_amd.h"
#include "../../pyspeedy_amd/csrc/driver_backend.hpp"
#include "driver_stub.hpp"

// ---------------------------------------------------------------------------------------------------------------------
// pretend devices
// ---------------------------------------------------------------------------------------------------------------------
namespace {
std::atomic<int> g_devices{2};
std::atomic<int> g_check_delay_us{0};
std::atomic<int> g_fail_next_check_begin{0};
std::atomic<int> g_fail_next_steps_begin{0};
std::atomic<long> g_device_syncs{0}, g_peer_copies{0}, g_local_copies{0}, g_collectives{0};
std::atomic<int> g_collective_available{1};
std::atomic<int> g_fail_model_create_in{0};  // n > 0: the n-th spd_model_create from now fails as a hipMalloc out of memory does
thread_local int t_device = 0;
thread_local std::string t_error;
}  // namespace

void stub_set_device_count(int n) { g_devices = n; }
void stub_set_check_delay_us(int us) { g_check_delay_us = us; }
void stub_fail_next_check_begin(int count) { g_fail_next_check_begin = count; }
void stub_fail_next_steps_begin(int count) { g_fail_next_steps_begin = count; }
long stub_device_syncs() { return g_device_syncs.load(); }
long stub_peer_copies() { return g_peer_copies.load(); }
long stub_local_copies() { return g_local_copies.load(); }
long stub_collectives() { return g_collectives.load(); }
void stub_set_collective_available(int yes) { g_collective_available = yes; }
void stub_fail_model_create_in(int n) { g_fail_model_create_in = n; }
int stub_current_device() { return t_device; }
void stub_set_current_device(int d) { t_device = d; }

namespace drvdev {
int device_count() { return g_devices.load(); }
bool get_device(int *device) {
    if (g_devices.load() <= 0) return false;
    *device = t_device;
    return true;
}
bool set_device(int device) {
    if (device < 0 || device >= g_devices.load()) return false;
    t_device = device;
    return true;
}
bool stream_create_apart(void **stream, void *const *others, int n_others) {
    for (int i = 0; i < n_others; ++i)
        if (others[i] && *static_cast<int *>(others[i]) != t_device) std::abort();  // (only streams of the current device are compared)
    *stream = new int(t_device);
    return true;
}
void stream_destroy(void *stream) { delete static_cast<int *>(stream); }
bool device_synchronize() {
    ++g_device_syncs;
    return true;
}
bool null_stream_synchronize() { return true; }
}  // namespace drvdev

int spd_set_error(int code, const std::string &msg) {
    t_error = msg;
    return code;
}

// ---------------------------------------------------------------------------------------------------------------------
// toy context and model
// ---------------------------------------------------------------------------------------------------------------------
struct spd_context {
    int device;
};

struct spd_model {
    int device = 0, M = 0;
    spd_model_control ctl{};
    bool initialized = false;
    std::vector<std::map<std::string, std::vector<char>>> vars;  // per member
    std::atomic<bool> slot_busy[2] = {{false}, {false}};
    // a check that was put off and not yet settled: ending it would have to LAUNCH it (the real model does), which is work inside
    // the model; a settled one is a pure wait
    std::atomic<bool> unsettled[2] = {{false}, {false}};
    int next_slot = 0;
    std::vector<int32_t> slot_codes[2];
    int member_groups = 2;              // spd_model_get_option / _set_option
    int groups_of_last_checked_call = 0;  // ... as it stood when the last checked multi-step call was issued
    std::atomic<int> steps_pending{0};  // spd_model_step_checked_begin / _end
    std::vector<int32_t> steps_failed, steps_accepted;
    // a model is driven by one host thread at a time (the contract of the boundary): two threads inside one model are a bug of
    // the DRIVER, and this counter catches it even where ThreadSanitizer's happens-before would not
    std::atomic<int> inside{0};
};

namespace {
struct Inside {
    spd_model *m;
    explicit Inside(spd_model *m_) : m(m_) {
        if (m->inside.fetch_add(1) != 0) {
            std::fprintf(stderr, "driver_stub: two host threads inside one device model\n");
            std::abort();
        }
        m->unsettled[0] = m->unsettled[1] = false;  // (whatever enters the model sends a check that was put off out first)
    }
    ~Inside() { m->inside.fetch_sub(1); }
};

std::vector<char> &var(spd_model *m, int member, const std::string &name, size_t bytes) {
    std::vector<char> &v = m->vars[member][name];
    if (v.size() != bytes) v.assign(bytes, 0);
    return v;
}

constexpr size_t kOlrBytes = 96 * 48 * sizeof(double);

void advance_date(spd_model_control &c) {  // 40 minutes; 365-day calendar (stub_calendar in driver_stub.hpp mirrors it)
    stub_advance(c.year, c.month, c.day, c.hour, c.minute, c.month_idx);
}
}  // namespace

extern "C" {

const char *spd_last_error(void) { return t_error.c_str(); }

int spd_create(spd_handle *out, int device) {
    if (!out) return spd_set_error(SPD_E_ARG, "spd_create: null argument");
    if (device < 0 || device >= g_devices.load()) return spd_set_error(SPD_E_DEVICE, "spd_create: no such device");
    static spd_context contexts[64];  // (alive for the life of the process, like the driver's contexts)
    if (device >= 64) return spd_set_error(SPD_E_DEVICE, "spd_create: no such device");
    contexts[device].device = device;
    *out = &contexts[device];
    return SPD_OK;
}

long spd_get_table_host(spd_handle, const char *name, double *buf, size_t n) {
    const std::string s(name ? name : "");
    const long len = s == "radang" ? 48 : (s == "fsg" ? 8 : (s == "hsg" ? 9 : (s == "fband" ? 1204 : -1)));
    if (len < 0) return spd_set_error(SPD_E_ARG, "spd_get_table_host: unknown table");
    if (buf)
        for (long i = 0; i < len && i < static_cast<long>(n); ++i) buf[i] = 0.1 + 0.01 * static_cast<double>(i);
    return len;
}

int spd_model_create(spd_handle h, int nmembers, spd_model_handle *out) {
    if (!h || !out || nmembers < 1) return spd_set_error(SPD_E_ARG, "spd_model_create: bad argument");
    if (g_fail_model_create_in.load() > 0 && g_fail_model_create_in.fetch_sub(1) == 1)
        return spd_set_error(SPD_E_DEVICE, "hipMalloc(&p, size): out of memory (injected)");
    spd_model *m = new spd_model();
    m->device = h->device;
    m->M = nmembers;
    m->vars.resize(nmembers);
    m->ctl.land_coupling_flag = m->ctl.sst_anomaly_coupling_flag = 1;
    m->ctl.month_idx = 1;
    m->ctl.month = m->ctl.day = 1;
    m->ctl.air_absortivity_co2 = m->ctl.ablco2_ref = 6.0;
    *out = m;
    return SPD_OK;
}

int spd_model_destroy(spd_model_handle m) {
    if (m) {
        Inside in(m);
        if (t_device != m->device) {
            std::fprintf(stderr, "driver_stub: a device model destroyed while another device is current\n");
            std::abort();
        }
    }
    delete m;
    return SPD_OK;
}

int spd_model_get_control(spd_model_handle m, spd_model_control *out) {
    if (!m || !out) return spd_set_error(SPD_E_ARG, "spd_model_get_control: null argument");
    *out = m->ctl;
    return SPD_OK;
}

int spd_model_set_control(spd_model_handle m, const spd_model_control *in) {
    if (!m || !in) return spd_set_error(SPD_E_ARG, "spd_model_set_control: null argument");
    Inside guard(m);
    m->ctl = *in;
    m->initialized = true;
    return SPD_OK;
}

int spd_model_init_sst_anom(spd_model_handle m, int n_months) {
    if (!m || n_months < 1) return spd_set_error(SPD_E_ARG, "spd_model_init_sst_anom: bad argument");
    Inside guard(m);
    for (int i = 0; i < m->M; ++i) var(m, i, "sst_anom", static_cast<size_t>(n_months + 2) * 96 * 48 * sizeof(double));
    return SPD_OK;
}

int spd_model_copy_member(spd_model_handle dst, int di, spd_model_handle src, int si, void *) {
    if (!dst || !src || di < 0 || di >= dst->M || si < 0 || si >= src->M) return spd_set_error(SPD_E_ARG, "spd_model_copy_member: bad argument");
    if (dst->device != src->device) return spd_set_error(SPD_E_ARG, "spd_model_copy_member: models live on different devices");
    Inside a(dst);
    dst->vars[di] = src->vars[si];
    return SPD_OK;
}

int spd_model_copy_vars_enqueue(spd_model_handle dst, int di, spd_model_handle src, int si, const char *const *names, int n, void *) {
    if (!dst || !src || di < 0 || di >= dst->M || si < 0 || si >= src->M) return spd_set_error(SPD_E_ARG, "spd_model_copy_vars: bad argument");
    Inside guard(dst);
    t_device = dst->device;
    for (int k = 0; k < n; ++k) {
        auto it = src->vars[si].find(names[k]);
        if (it == src->vars[si].end()) continue;  // (never set: nothing to hand over)
        if (dst == src && di == si) continue;
        dst->vars[di][names[k]] = it->second;
    }
    ++(dst->device != src->device ? g_peer_copies : g_local_copies);
    return SPD_OK;
}

int spd_model_broadcast_vars(const spd_model_handle *models, const int *members, int n, int root, const char *const *names, int nn) {
    // modes: 0 = no collective library, 1 = works, 2 = its initialisation did not come back inside the bound (nothing enqueued),
    //        3 = the broadcast was enqueued and did not complete inside the bound
    if (!g_collective_available.load()) return spd_set_error(SPD_E_DEVICE, "spd_model_broadcast_vars: RCCL is not available (stub)");
    if (g_collective_available.load() == 2)
        return spd_set_error(SPD_E_DEVICE, "spd_model_broadcast_vars: ncclCommInitAll over 2 devices did not return within 30 s (stub)");
    if (g_collective_available.load() == 3)
        return spd_set_error(SPD_E_TIMEOUT, "spd_model_broadcast_vars: the broadcast did not complete on 1 of 2 devices within 30 s (stub)");
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j)
            if (models[i]->device == models[j]->device) return spd_set_error(SPD_E_ARG, "spd_model_broadcast_vars: one model per GPU");
    for (int i = 0; i < n; ++i) {
        if (i == root) continue;
        Inside guard(models[i]);
        for (int k = 0; k < nn; ++k) {
            auto it = models[root]->vars[members[root]].find(names[k]);
            if (it != models[root]->vars[members[root]].end()) models[i]->vars[members[i]][names[k]] = it->second;
        }
    }
    ++g_collectives;
    return SPD_OK;
}

int spd_model_set_time_step(spd_model_handle m, double) { return m ? SPD_OK : spd_set_error(SPD_E_ARG, "null model"); }

int spd_model_set_flags(spd_model_handle m, int land, int sst, int co2) {
    if (!m) return spd_set_error(SPD_E_ARG, "spd_model_set_flags: null model");
    Inside guard(m);
    m->ctl.land_coupling_flag = land ? 1 : 0;
    m->ctl.sst_anomaly_coupling_flag = sst ? 1 : 0;
    m->ctl.increase_co2 = co2 ? 1 : 0;
    return SPD_OK;
}

int spd_model_set_co2(spd_model_handle m, double v) {
    if (!m) return spd_set_error(SPD_E_ARG, "spd_model_set_co2: null model");
    m->ctl.air_absortivity_co2 = v;
    return SPD_OK;
}

int spd_model_set_sppt(spd_model_handle m, int on, uint64_t seed, int64_t first) {
    if (!m) return spd_set_error(SPD_E_ARG, "spd_model_set_sppt: null model");
    m->ctl.sppt_on = on ? 1 : 0;
    m->ctl.sppt_seed = seed;
    m->ctl.sppt_first_member_id = first;
    return SPD_OK;
}

int spd_model_init(spd_model_handle m, int y, int mo, int d, int h, int mi, void *) {
    if (!m) return spd_set_error(SPD_E_ARG, "spd_model_init: null model");
    Inside guard(m);
    if (t_device != m->device) return spd_set_error(SPD_E_DEVICE, "spd_model_init: wrong current device");
    m->ctl.year = y; m->ctl.month = mo; m->ctl.day = d; m->ctl.hour = h; m->ctl.minute = mi;
    m->ctl.month_idx = 1;
    m->ctl.current_step = 0;
    m->ctl.ablco2_ref = m->ctl.air_absortivity_co2;
    m->initialized = true;
    for (int i = 0; i < m->M; ++i) {  // the state starts from its member's boundary fields: the fingerprint from orog[0]
        double *olr = reinterpret_cast<double *>(var(m, i, "olr", kOlrBytes).data());
        auto it = m->vars[i].find("orog");
        olr[0] = 0.0;
        olr[1] = it == m->vars[i].end() ? 0.0 : *reinterpret_cast<const double *>(it->second.data());
        olr[2] = 0.0;
    }
    return SPD_OK;
}

int spd_model_mark_initialized(spd_model_handle m, int step, int y, int mo, int d, int h, int mi) {
    if (!m) return spd_set_error(SPD_E_ARG, "spd_model_mark_initialized: null model");
    Inside guard(m);
    m->ctl.year = y; m->ctl.month = mo; m->ctl.day = d; m->ctl.hour = h; m->ctl.minute = mi;
    m->ctl.current_step = step;
    m->ctl.ablco2_ref = m->ctl.air_absortivity_co2;
    m->initialized = true;
    return SPD_OK;
}

int spd_model_step(spd_model_handle m, int nsteps, void *) {
    if (!m) return spd_set_error(SPD_E_ARG, "spd_model_step: null model");
    if (!m->initialized) return spd_set_error(SPD_E_ARG, "spd_model_step: model state not initialized");
    Inside guard(m);
    if (t_device != m->device) return spd_set_error(SPD_E_DEVICE, "spd_model_step: wrong current device");
    for (int it = 0; it < nsteps; ++it) {
        for (int i = 0; i < m->M; ++i) {
            double *olr = reinterpret_cast<double *>(var(m, i, "olr", kOlrBytes).data());
            olr[0] += 1.0;
            olr[1] = stub_fingerprint(olr[1], m->ctl.current_step);
        }
        m->ctl.current_step += 1;
        advance_date(m->ctl);
    }
    return SPD_OK;
}

// k steps with the range check of every step: the stub evaluates the check after each step (a member whose olr[2] is set fails
// from the step on at which olr[3] says so: olr[3] = n means "fails once its step counter reaches n", 0 = at once)
int spd_model_step_checked_begin(spd_model_handle m, int nsteps, void *) {
    if (!m) return spd_set_error(SPD_E_ARG, "spd_model_step_checked_begin: null model");
    if (nsteps < 1 || nsteps > 4096) return spd_set_error(SPD_E_ARG, "spd_model_step_checked_begin: 1 ... 4096 steps per call");
    if (!m->initialized) return spd_set_error(SPD_E_ARG, "spd_model_step_checked_begin: model state not initialized");
    if (m->steps_pending) return spd_set_error(SPD_E_ARG, "spd_model_step_checked_begin: a checked multi-step call is in flight already");
    Inside guard(m);
    if (t_device != m->device) return spd_set_error(SPD_E_DEVICE, "spd_model_step_checked_begin: wrong current device");
    if (g_fail_next_steps_begin.load() > 0 && g_fail_next_steps_begin.fetch_sub(1) > 0)
        return spd_set_error(SPD_E_DEVICE, "spd_model_step_checked_begin: injected device error");
    m->groups_of_last_checked_call = m->member_groups;
    m->steps_failed.assign(m->M, -1);
    m->steps_accepted.assign(static_cast<size_t>(m->M) * 7, 0);
    auto note = [&](int i) {
        int32_t *a = m->steps_accepted.data() + 7 * static_cast<size_t>(i);
        a[0] = m->ctl.current_step; a[1] = m->ctl.year; a[2] = m->ctl.month; a[3] = m->ctl.day; a[4] = m->ctl.hour; a[5] = m->ctl.minute;
        a[6] = m->ctl.month_idx;
    };
    for (int i = 0; i < m->M; ++i) note(i);
    for (int it = 0; it < nsteps; ++it) {
        for (int i = 0; i < m->M; ++i) {
            double *olr = reinterpret_cast<double *>(var(m, i, "olr", kOlrBytes).data());
            olr[0] += 1.0;
            olr[1] = stub_fingerprint(olr[1], m->ctl.current_step);
        }
        m->ctl.current_step += 1;
        advance_date(m->ctl);
        for (int i = 0; i < m->M; ++i) {
            const double *olr = reinterpret_cast<const double *>(var(m, i, "olr", kOlrBytes).data());
            if (m->steps_failed[i] < 0 && olr[2] != 0.0 && m->ctl.current_step >= static_cast<int>(olr[3])) m->steps_failed[i] = it;
            if (m->steps_failed[i] < 0) note(i);
        }
    }
    m->steps_pending = nsteps;
    return SPD_OK;
}

// (waits and reads only: outside the model, like spd_model_check_end of a check that is out)
int spd_model_step_checked_end(spd_model_handle m, int32_t *first_failed_step, int32_t *accepted) {
    if (!m || !first_failed_step) return spd_set_error(SPD_E_ARG, "spd_model_step_checked_end: null argument");
    if (!m->steps_pending) return spd_set_error(SPD_E_ARG, "spd_model_step_checked_end: no checked multi-step call is in flight");
    if (const int us = g_check_delay_us.load()) std::this_thread::sleep_for(std::chrono::microseconds(us));
    m->steps_pending = 0;
    std::memcpy(first_failed_step, m->steps_failed.data(), sizeof(int32_t) * m->M);
    if (accepted) std::memcpy(accepted, m->steps_accepted.data(), sizeof(int32_t) * 7 * m->M);
    return SPD_OK;
}

// (the launch plan is the model's own business: the stub keeps the one switch the driver touches)
int spd_model_get_option(spd_model_handle m, const char *name, int32_t *value) {
    if (!m || !name || !value) return spd_set_error(SPD_E_ARG, "spd_model_get_option: null argument");
    if (std::string(name) != "member_groups") return spd_set_error(SPD_E_ARG, "spd_model_get_option: unknown option");
    *value = m->member_groups;
    return SPD_OK;
}
int spd_model_set_option(spd_model_handle m, const char *name, int32_t value) {
    if (!m || !name) return spd_set_error(SPD_E_ARG, "spd_model_set_option: null argument");
    if (std::string(name) == "prepare_multi_step" && value == 1) return SPD_OK;
    if (std::string(name) != "member_groups" || value < 1 || value > 4) return spd_set_error(SPD_E_ARG, "spd_model_set_option: unknown option");
    m->member_groups = value;
    return SPD_OK;
}

int spd_model_checks_in_flight(spd_model_handle m) { return m ? (m->slot_busy[0] ? 1 : 0) + (m->slot_busy[1] ? 1 : 0) : SPD_E_ARG; }

static void range_check(spd_model *m, std::vector<int32_t> &codes) {
    codes.assign(m->M, 0);
    for (int i = 0; i < m->M; ++i) {
        const double *olr = reinterpret_cast<const double *>(var(m, i, "olr", kOlrBytes).data());
        if (olr[2] != 0.0) codes[i] = -2;
    }
}

int spd_model_check_begin(spd_model_handle m, int, void *) {
    if (!m) return spd_set_error(SPD_E_ARG, "spd_model_check_begin: null model");
    Inside guard(m);
    if (g_fail_next_check_begin.load() > 0 && g_fail_next_check_begin.fetch_sub(1) > 0)
        return spd_set_error(SPD_E_DEVICE, "spd_model_check_begin: injected device error");
    const int slot = m->slot_busy[m->next_slot] ? 1 - m->next_slot : m->next_slot;
    if (m->slot_busy[slot]) return spd_set_error(SPD_E_ARG, "spd_model_check_begin: two checks are in flight already");
    range_check(m, m->slot_codes[slot]);
    m->slot_busy[slot] = true;
    m->next_slot = 1 - slot;
    return slot;
}

// (the stub has no launches to save: the check is evaluated where it is put off, on the state as it is then -- which is what the
// real one guarantees to look at)
int spd_model_check_defer(spd_model_handle m, int time_level, void *stream) {
    const int slot = spd_model_check_begin(m, time_level, stream);
    if (slot >= 0) m->unsettled[slot] = true;  // (after the guard of check_begin has gone)
    return slot;
}

int spd_model_check_settle(spd_model_handle m, int slot) {
    if (!m || slot > 1) return spd_set_error(SPD_E_ARG, "spd_model_check_settle: bad argument");
    if (slot >= 0 && !m->unsettled[slot].load()) return SPD_OK;  // (out already: nothing to do inside the model)
    const bool other = slot >= 0 && m->unsettled[1 - slot].load();
    Inside guard(m);  // (entering the model settles both in the stub; the real one leaves the other slot's check waiting)
    if (other) m->unsettled[1 - slot] = true;
    return SPD_OK;
}

// The wait for a check that is out already happens OUTSIDE the model (the driver makes this call without its lock, beside
// whatever another host thread does to the model under the lock); ending a check that still has to be launched is work inside
// it, and the Inside guard aborts when the driver lets that happen beside another thread.
int spd_model_check_end(spd_model_handle m, int slot, int32_t *codes) {
    if (!m || !codes || slot < 0 || slot > 1 || !m->slot_busy[slot]) return spd_set_error(SPD_E_ARG, "spd_model_check_end: no check in this slot");
    if (m->unsettled[slot].exchange(false)) {
        Inside guard(m);
        if (const int us = g_check_delay_us.load()) std::this_thread::sleep_for(std::chrono::microseconds(us));
    } else if (const int us = g_check_delay_us.load()) {
        std::this_thread::sleep_for(std::chrono::microseconds(us));
    }
    std::memcpy(codes, m->slot_codes[slot].data(), sizeof(int32_t) * m->M);
    m->slot_busy[slot] = false;
    return SPD_OK;
}

int spd_model_check(spd_model_handle m, int, int32_t *codes, double *, void *) {
    if (!m || !codes) return spd_set_error(SPD_E_ARG, "spd_model_check: null argument");
    Inside guard(m);
    std::vector<int32_t> c;
    range_check(m, c);
    std::memcpy(codes, c.data(), sizeof(int32_t) * m->M);
    return SPD_OK;
}

int spd_model_current_step(spd_model_handle m) { return m ? m->ctl.current_step : SPD_E_ARG; }
int spd_model_spectral2grid(spd_model_handle m, int, int, void *) { return m ? SPD_OK : SPD_E_ARG; }
int spd_model_grid2spectral(spd_model_handle m, int, int, void *) { return m ? SPD_OK : SPD_E_ARG; }
int spd_model_grid_filter(spd_model_handle m, int, int, void *) { return m ? SPD_OK : SPD_E_ARG; }

int spd_model_set(spd_model_handle m, const char *name, int member, const void *host, size_t bytes) {
    if (!m || !name || !host || member < -1 || member >= m->M) return spd_set_error(SPD_E_ARG, "spd_model_set: bad argument");
    Inside guard(m);
    for (int i = member < 0 ? 0 : member; i <= (member < 0 ? m->M - 1 : member); ++i)
        std::memcpy(var(m, i, name, bytes).data(), host, bytes);
    return SPD_OK;
}

int spd_model_get(spd_model_handle m, const char *name, int member, void *host, size_t bytes) {
    if (!m || !name || !host || member < 0 || member >= m->M) return spd_set_error(SPD_E_ARG, "spd_model_get: bad argument");
    Inside guard(m);
    std::memcpy(host, var(m, member, name, bytes).data(), bytes);
    return SPD_OK;
}

}  // extern "C"
