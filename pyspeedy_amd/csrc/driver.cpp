// Outer boundary (include/pyspeedy_amd_driver.h): the procedures of the reference's f2py module speedy_driver
// (registry/templates/speedy_driver.f90.j2) on top of the batched device model (spd_model_*, model.hip).  Host code only:
// containers, the name-driven registry, and the gathering of independent one-member models into one batched model so that
// parallel_step is one set of kernel launches for the whole ensemble.  Plain C++: the GPU runtime is reached through the
// spd_model_* functions and the few calls of driver_backend.hpp, so that this file also builds against a stub of both and runs
// under the sanitizers on a machine without a GPU (tests/test_sanitizers.py, tests/sanitize/driver_stub.cpp).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pyspeedy_amd.h"
#include "../../include/pyspeedy_amd_driver.h"
#include "driver_backend.hpp"

namespace {

constexpr int IXc = 96, ILc = 48, KXc = 8, MXc = 31, NXc = 32;
constexpr double kDelt = 86400.0 / 36;  // params.f90:33

// ---------------------------------------------------------------------------------------------------------------------
// registry (registry/model_state_def.py:121-495): name -> element type, shape, where the value lives
// ---------------------------------------------------------------------------------------------------------------------
enum Where { Device, Table, Scalar, HostOnly };
struct RegVar {
    const char *name;
    int dtype;
    int ndim;
    int shape[5];  // -1 = n_months + 2
    Where where;
};

#define G2 2, {IXc, ILc, 0, 0, 0}
#define G3 3, {IXc, ILc, KXc, 0, 0}
#define GA 3, {IXc, ILc, 3, 0, 0}
#define G12 3, {IXc, ILc, 12, 0, 0}
const RegVar kRegistry[] = {
    {"vor", SPD_T_COMPLEX128, 4, {MXc, NXc, KXc, 2, 0}, Device}, {"div", SPD_T_COMPLEX128, 4, {MXc, NXc, KXc, 2, 0}, Device},
    {"t", SPD_T_COMPLEX128, 4, {MXc, NXc, KXc, 2, 0}, Device}, {"tr", SPD_T_COMPLEX128, 4, {MXc, NXc, KXc, 2, 0}, Device},
    {"ps", SPD_T_COMPLEX128, 3, {MXc, NXc, 2, 0, 0}, Device}, {"phi", SPD_T_COMPLEX128, 3, {MXc, NXc, KXc, 0, 0}, Device},
    {"phis", SPD_T_COMPLEX128, 2, {MXc, NXc, 0, 0, 0}, Device}, {"tcorh", SPD_T_COMPLEX128, 2, {MXc, NXc, 0, 0, 0}, Device},
    {"qcorh", SPD_T_COMPLEX128, 2, {MXc, NXc, 0, 0, 0}, Device},
    {"u_grid", SPD_T_FLOAT64, G3, Device}, {"v_grid", SPD_T_FLOAT64, G3, Device}, {"t_grid", SPD_T_FLOAT64, G3, Device},
    {"q_grid", SPD_T_FLOAT64, G3, Device}, {"phi_grid", SPD_T_FLOAT64, G3, Device}, {"ps_grid", SPD_T_FLOAT64, G2, Device},
    {"rad_st4a", SPD_T_FLOAT64, 4, {IXc, ILc, KXc, 2, 0}, Device}, {"rad_flux", SPD_T_FLOAT64, 3, {IXc, ILc, 4, 0, 0}, Device},
    {"tt_rsw", SPD_T_FLOAT64, G3, Device}, {"rad_tau2", SPD_T_FLOAT64, 4, {IXc, ILc, KXc, 4, 0}, Device},
    {"rad_strat_corr", SPD_T_FLOAT64, 3, {IXc, ILc, 2, 0, 0}, Device},
    {"fmask_land", SPD_T_FLOAT64, G2, Device}, {"phis0", SPD_T_FLOAT64, G2, Device}, {"forog", SPD_T_FLOAT64, G2, Device},
    {"sst_am", SPD_T_FLOAT64, G2, Device}, {"alb_land", SPD_T_FLOAT64, G2, Device}, {"alb_sea", SPD_T_FLOAT64, G2, Device},
    {"snowc", SPD_T_FLOAT64, G2, Device}, {"land_temp", SPD_T_FLOAT64, G2, Device},
    {"soil_avail_water", SPD_T_FLOAT64, G2, Device}, {"flux_solar_in", SPD_T_FLOAT64, G2, Device},
    {"flux_ozone_upper", SPD_T_FLOAT64, G2, Device}, {"flux_ozone_lower", SPD_T_FLOAT64, G2, Device},
    {"zenit_correction", SPD_T_FLOAT64, G2, Device}, {"stratospheric_correction", SPD_T_FLOAT64, G2, Device},
    {"alb_surface", SPD_T_FLOAT64, G2, Device}, {"precnv", SPD_T_FLOAT64, G2, Device}, {"precls", SPD_T_FLOAT64, G2, Device},
    {"cbmf", SPD_T_FLOAT64, G2, Device}, {"slrd", SPD_T_FLOAT64, G2, Device}, {"slr", SPD_T_FLOAT64, G2, Device},
    {"olr", SPD_T_FLOAT64, G2, Device}, {"tsr", SPD_T_FLOAT64, G2, Device}, {"ssrd", SPD_T_FLOAT64, G2, Device},
    {"ssr", SPD_T_FLOAT64, G2, Device}, {"qcloud_equiv", SPD_T_FLOAT64, G2, Device},
    {"slru", SPD_T_FLOAT64, GA, Device}, {"ustr", SPD_T_FLOAT64, GA, Device}, {"vstr", SPD_T_FLOAT64, GA, Device},
    {"shf", SPD_T_FLOAT64, GA, Device}, {"evap", SPD_T_FLOAT64, GA, Device}, {"hfluxn", SPD_T_FLOAT64, GA, Device},
    {"stl12", SPD_T_FLOAT64, G12, Device}, {"snowd12", SPD_T_FLOAT64, G12, Device}, {"soilw12", SPD_T_FLOAT64, G12, Device},
    {"sst12", SPD_T_FLOAT64, G12, Device}, {"sea_ice_frac12", SPD_T_FLOAT64, G12, Device},
    {"soil_wc_l1", SPD_T_FLOAT64, G12, Device}, {"soil_wc_l2", SPD_T_FLOAT64, G12, Device},
    {"soil_wc_l3", SPD_T_FLOAT64, G12, Device}, {"sst_anom", SPD_T_FLOAT64, 3, {IXc, ILc, -1, 0, 0}, Device},
    {"stlcl_obs", SPD_T_FLOAT64, G2, Device}, {"snowdcl_obs", SPD_T_FLOAT64, G2, Device},
    {"soilwcl_obs", SPD_T_FLOAT64, G2, Device}, {"stl_lm", SPD_T_FLOAT64, G2, Device}, {"snow_depth", SPD_T_FLOAT64, G2, Device},
    {"cdland", SPD_T_FLOAT64, G2, Device}, {"rhcapl", SPD_T_FLOAT64, G2, Device}, {"sstcl_ob", SPD_T_FLOAT64, G2, Device},
    {"sicecl_ob", SPD_T_FLOAT64, G2, Device}, {"ticecl_ob", SPD_T_FLOAT64, G2, Device}, {"sstan_ob", SPD_T_FLOAT64, G2, Device},
    {"sst_om", SPD_T_FLOAT64, G2, Device}, {"tice_om", SPD_T_FLOAT64, G2, Device}, {"sice_om", SPD_T_FLOAT64, G2, Device},
    {"sstan_am", SPD_T_FLOAT64, G2, Device}, {"sice_am", SPD_T_FLOAT64, G2, Device}, {"tice_am", SPD_T_FLOAT64, G2, Device},
    {"ssti_om", SPD_T_FLOAT64, G2, Device}, {"cdsea", SPD_T_FLOAT64, G2, Device}, {"cdice", SPD_T_FLOAT64, G2, Device},
    {"rhcaps", SPD_T_FLOAT64, G2, Device}, {"rhcapi", SPD_T_FLOAT64, G2, Device}, {"hfseacl", SPD_T_FLOAT64, G2, Device},
    {"fmask_sea", SPD_T_FLOAT64, G2, Device}, {"alb0", SPD_T_FLOAT64, G2, Device}, {"orog", SPD_T_FLOAT64, G2, Device},
    {"phi0", SPD_T_FLOAT64, G2, Device}, {"fmask_orig", SPD_T_FLOAT64, G2, Device}, {"veg_high", SPD_T_FLOAT64, G2, Device},
    {"veg_low", SPD_T_FLOAT64, G2, Device}, {"bmask_land", SPD_T_FLOAT64, G2, Device}, {"bmask_sea", SPD_T_FLOAT64, G2, Device},
    // allocated by the reference, never read by its time step: plain host arrays of the container
    {"snowcv", SPD_T_FLOAT64, G2, HostOnly}, {"snowls", SPD_T_FLOAT64, G2, HostOnly}, {"sstcl_om", SPD_T_FLOAT64, G2, HostOnly},
    {"wsst_ob", SPD_T_FLOAT64, G2, HostOnly}, {"sstom12", SPD_T_FLOAT64, G12, HostOnly},
    // read-only tables of the context
    {"lon", SPD_T_FLOAT32, 1, {IXc, 0, 0, 0, 0}, Table}, {"lat", SPD_T_FLOAT32, 1, {ILc, 0, 0, 0, 0}, Table},
    {"lev", SPD_T_FLOAT32, 1, {KXc, 0, 0, 0, 0}, Table}, {"deglat_s", SPD_T_FLOAT64, 1, {ILc, 0, 0, 0, 0}, Table},
    {"fband", SPD_T_FLOAT64, 2, {301, 4, 0, 0, 0}, Table}, {"xgeop1", SPD_T_FLOAT64, 1, {KXc, 0, 0, 0, 0}, Table},
    {"xgeop2", SPD_T_FLOAT64, 1, {KXc, 0, 0, 0, 0}, Table},
    // scalars
    {"current_step", SPD_T_INT32, 0, {0, 0, 0, 0, 0}, Scalar}, {"increase_co2", SPD_T_LOGICAL, 0, {0, 0, 0, 0, 0}, Scalar},
    {"compute_shortwave", SPD_T_LOGICAL, 0, {0, 0, 0, 0, 0}, Scalar},
    {"air_absortivity_co2", SPD_T_FLOAT64, 0, {0, 0, 0, 0, 0}, Scalar},
    {"land_coupling_flag", SPD_T_LOGICAL, 0, {0, 0, 0, 0, 0}, Scalar},
    {"sst_anomaly_coupling_flag", SPD_T_LOGICAL, 0, {0, 0, 0, 0, 0}, Scalar},
    {"ablco2_ref", SPD_T_FLOAT64, 0, {0, 0, 0, 0, 0}, Scalar},
};
#undef G2
#undef G3
#undef GA
#undef G12
constexpr int kRegistryCount = sizeof(kRegistry) / sizeof(kRegistry[0]);

const RegVar *find_var(const char *name) {
    if (!name) return nullptr;
    for (const RegVar &v : kRegistry)
        if (std::strcmp(v.name, name) == 0) return &v;
    return nullptr;
}
size_t elem_bytes(int dtype) { return dtype == SPD_T_COMPLEX128 ? 16 : (dtype == SPD_T_FLOAT64 ? 8 : 4); }

// ---------------------------------------------------------------------------------------------------------------------
// containers
// ---------------------------------------------------------------------------------------------------------------------
std::atomic<int> g_models_alive{0};

struct Batch {  // one device model shared by the containers of its members
    Batch() { ++g_models_alive; }
    Batch(const Batch &) = delete;
    spd_handle ctx = nullptr;
    spd_model_handle model = nullptr;
    int device = 0, members = 0, n_months = 1;
    bool sst_anom_allocated = false;
    std::vector<char> initialized;
    // The steps and range checks of this model are issued on a stream of its own, so that the models of one parallel_step --
    // on different devices, or several on one device -- run side by side.  A blocking stream: everything else the driver does
    // (initialisation, copies, transforms) stays on the null stream, which orders itself against it.  Created when the model is
    // first stepped (stream_for): creating and destroying a stream costs a millisecond each, and the one-member models of a host
    // with the reference's call sequence are gathered into batched models before any of them is stepped on its own.  Destroyed
    // with the model, not kept for the next one: an idle stream still holds its place among the device's few hardware queues,
    // and streams created after it double up on the others (measured: two idle streams of this kind made a model that steps
    // three member groups on streams of its own 33 % slower, tools/experiments/r04_idle_streams.py).
    void *stream = nullptr;
    // A step was enqueued and its range check (or anything else that had to follow it) could not be: the device state has
    // moved on while date and codes say it has not.  Nothing steps such a model again until it is initialised anew.
    bool advanced_without_check = false;
    // step counter of the last range failure each member was told about on stderr (-1: none); see report_out_of_range
    std::vector<int32_t> failed_step;
    bool merged = false;  // made by regroup() out of several device models for a multi-step call (spd_parallel_steps_begin)
    ~Batch() {
        --g_models_alive;
        drvdev::DeviceGuard guard;  // (may run from any entry point that drops the last reference, or from a host's garbage collector)
        (void)drvdev::set_device(device);
        if (model) (void)spd_model_destroy(model);
        if (stream) drvdev::stream_destroy(stream);
    }
};
struct State {
    std::shared_ptr<Batch> batch;
    int member = 0;
    std::map<std::string, std::vector<double>> host;  // HostOnly arrays
    bool compute_shortwave = true;                      // model_state_def.py:312-318 default
};
struct Date {
    int32_t ymdhm[5];
};
struct Control {  // ControlParams_t: start / end and the running model date with its month index
    Date start, end, now;
    int32_t month_idx = 1;
};

std::recursive_mutex g_mutex;
int64_t g_next = 1;
std::map<int64_t, std::shared_ptr<State>> g_states;
std::map<int64_t, Date> g_dates;
std::map<int64_t, Control> g_controls;
std::map<int, spd_handle> g_contexts;  // one context per device, alive for the life of the process
// Counts every event after which the grouping of an argument list may come out differently: containers created or closed,
// batches gathered or split, a member initialised, a step that left members with different dates.  parallel_step keeps the
// plan of its last argument list and re-uses it while this has not moved (plan_step).
uint64_t g_epoch = 1;
struct BroadcastStats {
    int peer_copies, local_copies, collective_devices;
} g_broadcast_stats{0, 0, 0};  // of the last spd_broadcast_boundary
std::string g_broadcast_note;    // ... and its transport in words, with the reason when the collective was not used

// The plan of an argument list: which of its containers are (all) the members of which device model.  Made by plan_step,
// never changed afterwards, shared by the calls that use it.
struct GroupPlan {
    std::shared_ptr<Batch> batch;
    std::vector<int> positions;        // indices into the argument list
    std::vector<int> members;          // member index of each position
    std::vector<int64_t> control_ids;  // the control containers, looked up again whenever the lock was given up
};
struct Plan {
    std::vector<int64_t> states, controls;  // the argument list it was made for
    std::vector<GroupPlan> groups;
    uint64_t epoch = 0;
    bool stretch = false;  // made for a multi-step call (the device models of a device merged into one) or for single steps
};
// A host with the reference's loop hands parallel_step the same two lists at every model step: the plans of the last few
// argument lists are kept (several host threads may each step a list of their own) and used again while nothing has happened
// that could change them (g_epoch) -- the per-step host work of the call is then a comparison of the lists and one
// control-container look-up per device model, whatever the number of containers.
constexpr size_t kKeptPlans = 8;
std::vector<std::shared_ptr<const Plan>> g_plans;  // most recently used first

void regrouped() {  // (lock held) something happened after which an argument list may group differently
    ++g_epoch;
    g_plans.clear();  // (a plan holds references to device models: they must be free to die with their containers)
}


// The stream a device model is stepped on, created when it is first needed (lock held, calling thread; the model's device is
// current) -- on a hardware queue of its own among the models of its device: the device models of one parallel_step are there to
// run side by side, and two streams that HIP has put on one hardware queue do not (stream_apart.hpp).
bool stream_for(Batch &b) {
    if (b.stream) return true;
    std::vector<void *> others;
    for (auto &kv : g_states) {
        const Batch &o = *kv.second->batch;
        if (&o != &b && o.device == b.device && o.stream && std::find(others.begin(), others.end(), o.stream) == others.end())
            others.push_back(o.stream);
    }
    return drvdev::stream_create_apart(&b.stream, others.data(), static_cast<int>(others.size()));
}

int fail(int code, const std::string &msg) { return spd_set_error(code, msg); }

int context_for_device(int dev, spd_handle *out) {
    auto it = g_contexts.find(dev);
    if (it == g_contexts.end()) {
        spd_handle h = nullptr;
        if (int rc = spd_create(&h, dev)) return rc;
        it = g_contexts.emplace(dev, h).first;
    }
    *out = it->second;
    return SPD_OK;
}

// Where a new state container lives.  Default: the HIP device that is current in the calling thread (one process per GPU,
// the torch.distributed layout).  spd_set_device_placement(k) or PYSPEEDY_AMD_DEVICES=k|all makes ONE process spread its
// containers over devices 0 .. k-1: single containers round-robin in creation order, the members of
// spd_modelstate_init_ensemble in blocks (member e of n on device e k / n).  parallel_step then drives all
// devices from the one call (the reference's one-process ensemble, speedy_driver.f90.j2:58-79).
int g_place_ndev = -1;  // -1: not decided yet (environment), 0: current device, k > 0: devices 0 .. k-1
long g_place_counter = 0;

int device_count() { return drvdev::device_count(); }

int placement_devices() {
    if (g_place_ndev < 0) {
        g_place_ndev = 0;
        if (const char *e = getenv("PYSPEEDY_AMD_DEVICES")) {
            const int have = device_count();
            int want = std::strcmp(e, "all") == 0 ? have : atoi(e);
            if (want > have) want = have;
            g_place_ndev = want > 0 ? want : 0;
        }
    }
    return g_place_ndev;
}

int current_device(int *dev) {
    if (!drvdev::get_device(dev)) return fail(SPD_E_DEVICE, "speedy driver: no HIP device (there is no CPU fallback)");
    return SPD_OK;
}

int new_batch(int members, int device, std::shared_ptr<Batch> *out) {
    auto b = std::make_shared<Batch>();
    b->device = device;
    if (!drvdev::set_device(device)) return fail(SPD_E_DEVICE, "speedy driver: hipSetDevice(" + std::to_string(device) + ") failed");
    if (int rc = context_for_device(device, &b->ctx)) return rc;
    if (int rc = spd_model_create(b->ctx, members, &b->model)) return rc;
    b->members = members;
    b->initialized.assign(members, 0);
    *out = b;
    return SPD_OK;
}


std::shared_ptr<State> state_of(int64_t cnt) {
    auto it = g_states.find(cnt);
    return it == g_states.end() ? nullptr : it->second;
}

bool same_date(const Control &a, const Control &b) {
    return std::memcmp(a.now.ymdhm, b.now.ymdhm, sizeof(a.now.ymdhm)) == 0 && a.month_idx == b.month_idx;
}

// the model's host-side control block with the date of a control container in it
int push_date(Batch &b, const Control &c) {
    spd_model_control mc;
    if (int rc = spd_model_get_control(b.model, &mc)) return rc;
    // A host with the reference's loop hands the model back the date the model gave it: nothing to do then (setting the
    // control block drops the day's interpolated climatologies, and every step would interpolate them again).
    if (mc.year == c.now.ymdhm[0] && mc.month == c.now.ymdhm[1] && mc.day == c.now.ymdhm[2] && mc.hour == c.now.ymdhm[3] &&
        mc.minute == c.now.ymdhm[4] && mc.month_idx == c.month_idx)
        return SPD_OK;
    mc.year = c.now.ymdhm[0]; mc.month = c.now.ymdhm[1]; mc.day = c.now.ymdhm[2]; mc.hour = c.now.ymdhm[3];
    mc.minute = c.now.ymdhm[4];
    mc.month_idx = c.month_idx;
    return spd_model_set_control(b.model, &mc);
}
int pull_date(Batch &b, Control &c, int32_t *current_step = nullptr) {
    spd_model_control mc;
    if (int rc = spd_model_get_control(b.model, &mc)) return rc;
    if (current_step) *current_step = mc.current_step;
    c.now.ymdhm[0] = mc.year; c.now.ymdhm[1] = mc.month; c.now.ymdhm[2] = mc.day; c.now.ymdhm[3] = mc.hour;
    c.now.ymdhm[4] = mc.minute;
    c.month_idx = mc.month_idx;
    return SPD_OK;
}

// every container that is a member of `b`
std::vector<std::shared_ptr<State>> members_of(const std::shared_ptr<Batch> &b) {
    std::vector<std::shared_ptr<State>> out;
    for (auto &kv : g_states)
        if (kv.second->batch == b) out.push_back(kv.second);
    return out;
}

// Take a batched model apart: every container bound to it gets a one-member model of its own (device-to-device copies).
int split_batch(const std::shared_ptr<Batch> &b) {
    if (b->members == 1) return SPD_OK;
    spd_model_control mc;
    if (int rc = spd_model_get_control(b->model, &mc)) return rc;
    regrouped();  // (first: a copy that fails below leaves some containers moved already, and no kept plan may outlive that)
    for (auto &st : members_of(b)) {
        std::shared_ptr<Batch> single;
        if (int rc = new_batch(1, b->device, &single)) return rc;
        if (b->sst_anom_allocated) {
            if (int rc = spd_model_init_sst_anom(single->model, b->n_months)) return rc;
            single->n_months = b->n_months;
            single->sst_anom_allocated = true;
        }
        spd_model_control mine = mc;
        if (mc.sppt_on) {  // the generator is keyed by global member ids: the single keeps the id its member had in the batch
            mine.sppt_first_member_id = mc.sppt_first_member_id + st->member;
            if (int rc = spd_model_set_sppt(single->model, 1, mc.sppt_seed, mine.sppt_first_member_id)) return rc;
        }
        if (int rc = spd_model_copy_member(single->model, 0, b->model, st->member, nullptr)) return rc;
        single->initialized[0] = b->initialized[st->member];
        single->advanced_without_check = b->advanced_without_check;
        if (single->initialized[0]) {
            if (int rc = spd_model_set_control(single->model, &mine)) return rc;
            if (int rc = spd_model_set_time_step(single->model, 2 * kDelt)) return rc;
        } else {
            (void)spd_model_set_flags(single->model, mc.land_coupling_flag, mc.sst_anomaly_coupling_flag, mc.increase_co2);
            (void)spd_model_set_co2(single->model, mc.air_absortivity_co2);
        }
        st->batch = single;
        st->member = 0;
    }
    if (!drvdev::device_synchronize()) return fail(SPD_E_DEVICE, "speedy driver: device error while splitting a batch");
    return SPD_OK;  // (`b` dies with the caller's reference)
}

// How many device models the n members that share a device (and a date, and their control flags) are kept in.  From 32
// members up: two.  parallel_step enqueues the step and range check of every device model before it waits for any, so the
// kernels of the two share the GPU and fill each other's tails -- the member groups of spd_model_step one level up, where no
// stream has to be forked and joined around every step of a host that calls once per model step (measured at 64 members:
// two models 0.28 ms per step through spd_parallel_step_begin / _end, one model 0.30, one model stepped as two stream groups
// 0.34).  PYSPEEDY_AMD_DRIVER_SPLIT=0 keeps one model; =n splits from n members.
size_t device_model_parts(size_t n) {
    static const int split_from = [] {
        const char *e = getenv("PYSPEEDY_AMD_DRIVER_SPLIT");
        return e ? atoi(e) : 32;
    }();
    return (split_from > 0 && n >= static_cast<size_t>(split_from)) ? 2 : 1;
}

// Gather independent, initialised one-member models of ONE device that agree in their control blocks (the caller made the
// class: same device, same n_months, same control container date) into one batched model.  *done = false when they turn out
// not to be gatherable after all (then they are stepped as they are).
int gather(const std::vector<std::shared_ptr<State>> &states, bool *done) {
    *done = false;
    const int n = static_cast<int>(states.size());
    spd_model_control first{};
    for (int i = 0; i < n; ++i) {
        const Batch &b = *states[i]->batch;
        if (b.members != 1 || !b.initialized[0]) return SPD_OK;
        if (b.device != states[0]->batch->device || b.n_months != states[0]->batch->n_months ||
            b.sst_anom_allocated != states[0]->batch->sst_anom_allocated)
            return SPD_OK;
        spd_model_control mc;
        if (int rc = spd_model_get_control(b.model, &mc)) return rc;
        if (i == 0) first = mc;
        else if (std::memcmp(&mc, &first, sizeof(mc)) != 0) return SPD_OK;
    }
    if (first.sppt_on) return SPD_OK;  // (the SPPT generator is keyed by member ids of the model it was set up for)
    std::shared_ptr<Batch> big;
    if (int rc = new_batch(n, states[0]->batch->device, &big)) return rc;
    if (states[0]->batch->sst_anom_allocated) {
        if (int rc = spd_model_init_sst_anom(big->model, states[0]->batch->n_months)) return rc;
        big->n_months = states[0]->batch->n_months;
        big->sst_anom_allocated = true;
    }
    for (int i = 0; i < n; ++i)
        if (int rc = spd_model_copy_member(big->model, i, states[i]->batch->model, 0, nullptr)) return rc;
    if (!drvdev::device_synchronize()) return fail(SPD_E_DEVICE, "speedy driver: device error while gathering a batch");
    if (int rc = spd_model_set_control(big->model, &first)) return rc;
    if (int rc = spd_model_set_time_step(big->model, 2 * kDelt)) return rc;
    for (int i = 0; i < n; ++i) {
        states[i]->batch = big;  // (the one-member model dies with its last reference)
        states[i]->member = i;
        big->initialized[i] = 1;
    }
    regrouped();
    *done = true;
    return SPD_OK;
}

// Re-cut the device models of ONE device: `states` (in this order; all the members of the models they come from, initialised,
// with equal control blocks) become `parts` new batched models, device to device.  The outer boundary keeps 32 or more containers
// of a device in TWO models because a host that calls once per step gets the overlap of two member groups that way without any
// stream hand-over per step (device_model_parts); a host that hands over MANY steps at once (spd_parallel_steps_begin: the
// stretches of Speedy.run / SpeedyEns.run) is better served by ONE model, whose own multi-step plan then forms the groups, offsets
// them against each other and takes a large ensemble in rounds (measured at 64 members: 0.256 ms per step against 0.294 with the
// two halves side by side, profiles/r06_facade_plans.txt).  So the first multi-step call over the two halves merges them (parts =
// 1), and a single step over a model that was merged halves it again (parts = 2): a few milliseconds, once per change of habit.
// *done = false when the models turn out not to be re-cuttable (SPPT: its generator is keyed by the member ids of the model it
// was set up for; unequal control blocks; a model that is marked) -- they are then stepped as they are.
int regroup(const std::vector<std::shared_ptr<State>> &states, int parts, bool *done) {
    *done = false;
    const int n = static_cast<int>(states.size());
    if (n < 2 || parts < 1 || parts > n) return SPD_OK;
    const Batch &b0 = *states[0]->batch;
    spd_model_control first{};
    std::vector<const Batch *> seen;
    for (int i = 0; i < n; ++i) {
        const Batch &b = *states[i]->batch;
        if (!b.initialized[states[i]->member] || b.advanced_without_check) return SPD_OK;
        if (b.device != b0.device || b.n_months != b0.n_months || b.sst_anom_allocated != b0.sst_anom_allocated) return SPD_OK;
        if (std::find(seen.begin(), seen.end(), &b) != seen.end()) continue;
        seen.push_back(&b);
        if (spd_model_checks_in_flight(b.model) != 0) return SPD_OK;
        spd_model_control mc;
        if (int rc = spd_model_get_control(b.model, &mc)) return rc;
        if (seen.size() == 1) first = mc;
        else if (std::memcmp(&mc, &first, sizeof(mc)) != 0) return SPD_OK;
    }
    if (first.sppt_on) return SPD_OK;
    regrouped();  // (first: a copy that fails below leaves some containers moved already, and no kept plan may outlive that)
    std::vector<std::shared_ptr<Batch>> made;
    for (int part = 0, at = 0; part < parts; ++part) {
        const int count = n / parts + (part < n % parts ? 1 : 0);
        std::shared_ptr<Batch> big;
        if (int rc = new_batch(count, b0.device, &big)) return rc;
        if (b0.sst_anom_allocated) {
            if (int rc = spd_model_init_sst_anom(big->model, b0.n_months)) return rc;
            big->n_months = b0.n_months;
            big->sst_anom_allocated = true;
        }
        for (int i = 0; i < count; ++i)
            if (int rc = spd_model_copy_member(big->model, i, states[at + i]->batch->model, states[at + i]->member, nullptr)) return rc;
        big->merged = parts == 1;
        if (big->merged) (void)spd_model_set_option(big->model, "prepare_multi_step", 1);
        made.push_back(big);
        at += count;
    }
    if (!drvdev::device_synchronize()) return fail(SPD_E_DEVICE, "speedy driver: device error while regrouping device models");
    for (auto &big : made) {
        if (int rc = spd_model_set_control(big->model, &first)) return rc;
        if (int rc = spd_model_set_time_step(big->model, 2 * kDelt)) return rc;
    }
    for (int part = 0, at = 0; part < parts; ++part) {
        const int count = made[part]->members;
        for (int i = 0; i < count; ++i) {
            states[at + i]->batch = made[part];  // (a model none of whose members is left dies with its last reference)
            states[at + i]->member = i;
            made[part]->initialized[i] = 1;
        }
        at += count;
    }
    *done = true;
    return SPD_OK;
}

int table_values(const RegVar &v, spd_handle ctx, std::vector<double> &out64, std::vector<float> &out32) {
    auto tab = [&](const char *name, std::vector<double> &dst) -> int {
        const long n = spd_get_table_host(ctx, name, nullptr, 0);
        if (n < 0) return static_cast<int>(n);
        dst.resize(n);
        return spd_get_table_host(ctx, name, dst.data(), dst.size()) < 0 ? SPD_E_ARG : SPD_OK;
    };
    const std::string s(v.name);
    std::vector<double> a, b;
    if (s == "lon") {  // initialization.f90:86
        for (int i = 0; i < IXc; ++i) out32.push_back(3.75f * static_cast<float>(i));
    } else if (s == "lat" || s == "deglat_s") {  // initialization.f90:87 (default real) / sea_model.f90 (real(p))
        if (int rc = tab("radang", a)) return rc;
        for (int j = 0; j < ILc; ++j) {
            if (s == "lat") out32.push_back(static_cast<float>(a[j]) * 90.0f / std::asin(1.0f));
            else out64.push_back(a[j] * 90.0 / std::asin(1.0));
        }
    } else if (s == "lev") {  // initialization.f90:85
        if (int rc = tab("fsg", a)) return rc;
        for (int k = 0; k < KXc; ++k) out32.push_back(static_cast<float>(a[k]));
    } else if (s == "fband") {
        if (int rc = tab("fband", out64)) return rc;
    } else {  // xgeop1 / xgeop2, geopotential.f90:16-31
        if (int rc = tab("hsg", a)) return rc;
        if (int rc = tab("fsg", b)) return rc;
        const double rgas = static_cast<double>(2.0f / 7.0f) * 1004.0;
        out64.assign(KXc, 0.0);
        for (int k = 0; k < KXc; ++k) {
            if (s == "xgeop1") out64[k] = rgas * std::log(a[k + 1] / b[k]);
            else if (k > 0) out64[k] = rgas * std::log(b[k] / a[k]);
        }
    }
    return SPD_OK;
}

size_t var_bytes(const RegVar &v, const Batch &b) {
    size_t n = elem_bytes(v.dtype);
    for (int d = 0; d < v.ndim; ++d) n *= static_cast<size_t>(v.shape[d] < 0 ? b.n_months + 2 : v.shape[d]);
    return n;
}

}  // namespace

#define LOCK std::lock_guard<std::recursive_mutex> lock_(g_mutex)

extern "C" {

// ---------------------------------------------------------------------------------------------------------------------
// ModelState
// ---------------------------------------------------------------------------------------------------------------------
// the n members that share a device live in one device model up to 31 members and in two from 32 up (plan_step has the why)
static int make_device_containers(int64_t *state_cnts, int n, int device, bool whole);

static int make_containers(int64_t *state_cnts, int n, int device, bool whole = false) {
    std::shared_ptr<Batch> b;
    if (int rc = new_batch(n, device, &b)) return rc;
    b->merged = whole;  // (one model for a host that steps many steps at once: halved when it turns out to step one by one)
    if (whole) (void)spd_model_set_option(b->model, "prepare_multi_step", 1);  // (its group streams now, not inside the first stretch)
    for (int i = 0; i < n; ++i) {
        auto st = std::make_shared<State>();
        st->batch = b;
        st->member = i;
        state_cnts[i] = g_next++;
        g_states[state_cnts[i]] = st;
    }
    regrouped();
    return SPD_OK;
}

static int make_device_containers(int64_t *state_cnts, int n, int device, bool whole) {
    if (whole) return make_containers(state_cnts, n, device, true);
    const int parts = static_cast<int>(device_model_parts(static_cast<size_t>(n)));
    for (int part = 0, first = 0; part < parts; ++part) {
        const int count = n / parts + (part < n % parts ? 1 : 0);
        if (int rc = make_containers(state_cnts + first, count, device)) return rc;
        first += count;
    }
    return SPD_OK;
}

int spd_modelstate_init(int64_t *state_cnt) {
    if (!state_cnt) return fail(SPD_E_ARG, "spd_modelstate_init: null argument");
    drvdev::DeviceGuard guard;
    LOCK;
    int dev = 0;
    const int k = placement_devices();
    if (k > 0) dev = static_cast<int>(g_place_counter++ % k);  // one process over k devices: round-robin in creation order
    else if (int rc = current_device(&dev)) return rc;
    return make_containers(state_cnt, 1, dev);
}

int spd_modelstate_init_on(int64_t *state_cnt, int32_t device) {
    if (!state_cnt) return fail(SPD_E_ARG, "spd_modelstate_init_on: null argument");
    if (device < 0 || device >= device_count()) return fail(SPD_E_ARG, "spd_modelstate_init_on: no such HIP device");
    drvdev::DeviceGuard guard;
    LOCK;
    return make_containers(state_cnt, 1, device);
}

// n containers batched from the start over k devices (k = 0: the current device); (lock held)
static int place_ensemble(int64_t *state_cnts, int32_t n_members, int k, bool whole);

// All n containers, or none: an ensemble is several device models (two per device from 32 members up, one or two on every
// device of a one-process ensemble), and creating one of the later ones may fail -- a GPU out of memory, a device that does
// not answer.  The containers of the models made before it are closed again (their device models go with them, the memory
// returns to the context), every entry of state_cnts is 0, and the caller gets the error of the model that failed.
static int init_ensemble(int64_t *state_cnts, int32_t n_members, int k, bool whole = false) {
    for (int i = 0; i < n_members; ++i) state_cnts[i] = 0;
    const int rc = place_ensemble(state_cnts, n_members, k, whole);
    if (rc == SPD_OK) return rc;
    const std::string why = spd_last_error();
    regrouped();  // (no kept plan may hold on to a model that is about to go)
    for (int i = 0; i < n_members; ++i) {
        if (state_cnts[i]) g_states.erase(state_cnts[i]);
        state_cnts[i] = 0;
    }
    return fail(rc, why);
}

static int place_ensemble(int64_t *state_cnts, int32_t n_members, int k, bool whole) {
    if (k <= 1) {
        int dev = 0;
        if (k == 0) {
            if (int rc = current_device(&dev)) return rc;
        }
        return make_device_containers(state_cnts, n_members, dev, whole);
    }
    // block partition: member e on device e k / n, one batched model per device
    for (int d = 0, first = 0; d < k; ++d) {
        int last = first;
        while (last < n_members && static_cast<long>(last) * k / n_members == d) ++last;
        if (last > first)
            if (int rc = make_device_containers(state_cnts + first, last - first, d, whole)) return rc;
        first = last;
    }
    return SPD_OK;
}

int spd_modelstate_init_ensemble(int64_t *state_cnts, int32_t n_members) {
    if (!state_cnts || n_members < 1) return fail(SPD_E_ARG, "spd_modelstate_init_ensemble: bad argument");
    drvdev::DeviceGuard guard;
    LOCK;
    return init_ensemble(state_cnts, n_members, placement_devices());
}

// ONE device model per device whatever the size: for hosts that hand over many steps at once (spd_parallel_steps_begin).
// n_devices < 0: the process-wide placement.
int spd_modelstate_init_ensemble_whole(int64_t *state_cnts, int32_t n_members, int32_t n_devices) {
    if (!state_cnts || n_members < 1) return fail(SPD_E_ARG, "spd_modelstate_init_ensemble_whole: bad argument");
    if (n_devices > device_count()) return fail(SPD_E_ARG, "spd_modelstate_init_ensemble_whole: more devices than the process can see");
    drvdev::DeviceGuard guard;
    LOCK;
    return init_ensemble(state_cnts, n_members, n_devices < 0 ? placement_devices() : n_devices, true);
}

// the same with the number of devices as an argument: the process-wide placement is neither read nor changed
int spd_modelstate_init_ensemble_on(int64_t *state_cnts, int32_t n_members, int32_t n_devices) {
    if (!state_cnts || n_members < 1) return fail(SPD_E_ARG, "spd_modelstate_init_ensemble_on: bad argument");
    if (n_devices < 0 || n_devices > device_count())
        return fail(SPD_E_ARG, "spd_modelstate_init_ensemble_on: more devices than the process can see");
    drvdev::DeviceGuard guard;
    LOCK;
    return init_ensemble(state_cnts, n_members, n_devices);
}

int spd_device_count(int32_t *n_devices) {
    if (!n_devices) return fail(SPD_E_ARG, "spd_device_count: null argument");
    *n_devices = device_count();
    return SPD_OK;
}

int spd_set_device_placement(int32_t n_devices) {
    if (n_devices < 0 || n_devices > device_count()) return fail(SPD_E_ARG, "spd_set_device_placement: more devices than the process can see");
    LOCK;
    g_place_ndev = n_devices;
    g_place_counter = 0;
    return SPD_OK;
}

int spd_modelstate_device(int64_t state_cnt, int32_t *device) {
    LOCK;
    auto st = state_of(state_cnt);
    if (!st || !device) return fail(SPD_E_ARG, "spd_modelstate_device: not a live state container");
    *device = st->batch->device;
    return SPD_OK;
}

int spd_modelstate_init_sst_anom(int64_t state_cnt, int32_t n_months) {
    drvdev::DeviceGuard guard;
    LOCK;
    auto st = state_of(state_cnt);
    if (!st) return fail(SPD_E_ARG, "spd_modelstate_init_sst_anom: not a live state container");
    Batch &b = *st->batch;
    if (b.sst_anom_allocated && b.n_months == n_months) return SPD_OK;
    if (int rc = spd_model_init_sst_anom(b.model, n_months)) return rc;
    b.n_months = n_months;
    b.sst_anom_allocated = true;
    regrouped();
    return SPD_OK;
}

int spd_modelstate_close(int64_t state_cnt) {
    drvdev::DeviceGuard guard;  // (the device model's destructor switches to its device)
    LOCK;
    auto it = g_states.find(state_cnt);
    if (it == g_states.end()) return SPD_OK;
    regrouped();         // (drops the kept plan's references first)
    g_states.erase(it);  // the device model goes with its last container
    return SPD_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Datetime, ControlParams
// ---------------------------------------------------------------------------------------------------------------------
int spd_create_datetime(int32_t year, int32_t month, int32_t day, int32_t hour, int32_t minute, int64_t *datetime_cnt) {
    if (!datetime_cnt) return fail(SPD_E_ARG, "spd_create_datetime: null argument");
    LOCK;
    *datetime_cnt = g_next++;
    g_dates[*datetime_cnt] = Date{{year, month, day, hour, minute}};
    return SPD_OK;
}

int spd_get_datetime(int64_t cnt, int32_t *year, int32_t *month, int32_t *day, int32_t *hour, int32_t *minute) {
    LOCK;
    auto it = g_dates.find(cnt);
    if (it == g_dates.end() || !year || !month || !day || !hour || !minute)
        return fail(SPD_E_ARG, "spd_get_datetime: not a live datetime container");
    *year = it->second.ymdhm[0]; *month = it->second.ymdhm[1]; *day = it->second.ymdhm[2];
    *hour = it->second.ymdhm[3]; *minute = it->second.ymdhm[4];
    return SPD_OK;
}

int spd_close_datetime(int64_t cnt) {
    LOCK;
    g_dates.erase(cnt);
    return SPD_OK;
}

int spd_controlparams_init(int64_t *control_cnt, int64_t start_cnt, int64_t end_cnt) {
    if (!control_cnt) return fail(SPD_E_ARG, "spd_controlparams_init: null argument");
    LOCK;
    auto a = g_dates.find(start_cnt), b = g_dates.find(end_cnt);
    if (a == g_dates.end() || b == g_dates.end()) return fail(SPD_E_ARG, "spd_controlparams_init: not a live datetime container");
    const Date &s = a->second;
    if (s.ymdhm[1] < 1 || s.ymdhm[1] > 12 || s.ymdhm[2] < 1 || s.ymdhm[2] > 31)
        return fail(SPD_E_ARG, "spd_controlparams_init: bad start date");
    Control c;
    c.start = s;
    c.end = b->second;
    c.now = s;  // initialize_control, model_control.f90:91: the model datetime starts at the start datetime
    c.month_idx = 1;
    *control_cnt = g_next++;
    g_controls[*control_cnt] = c;
    return SPD_OK;
}

int spd_controlparams_close(int64_t cnt) {
    LOCK;
    if (g_controls.erase(cnt)) regrouped();
    return SPD_OK;
}

int spd_controlparams_get_model_datetime(int64_t cnt, int32_t *ymdhm, int32_t *month_idx) {
    LOCK;
    auto it = g_controls.find(cnt);
    if (it == g_controls.end() || !ymdhm) return fail(SPD_E_ARG, "spd_controlparams_get_model_datetime: not a live control container");
    std::memcpy(ymdhm, it->second.now.ymdhm, sizeof(it->second.now.ymdhm));
    if (month_idx) *month_idx = it->second.month_idx;
    return SPD_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// init / step / parallel_step / check / transforms
// ---------------------------------------------------------------------------------------------------------------------
int spd_init(int64_t state_cnt, int64_t control_cnt, int32_t *error_code) {
    if (!error_code) return fail(SPD_E_ARG, "spd_init: null argument");
    drvdev::DeviceGuard guard;
    LOCK;
    auto st = state_of(state_cnt);
    auto ci = g_controls.find(control_cnt);
    if (!st || ci == g_controls.end()) return fail(SPD_E_ARG, "spd_init: not a live state / control container");
    Control &c = ci->second;
    std::shared_ptr<Batch> b = st->batch;
    if (!drvdev::set_device(b->device)) return fail(SPD_E_DEVICE, "spd_init: hipSetDevice failed");
    const int32_t *d = c.start.ymdhm;
    if (b->members > 1) {
        // A member of a batched model.  The batch has ONE date and step counter: it takes them when its last member has been
        // initialised, and all members must have been given the same start date.  A member that is initialised again after
        // the batch is complete, or with another start date than the members before it, leaves the batch first.
        bool any = false, all = true;
        for (int i = 0; i < b->members; ++i) {
            any = any || b->initialized[i];
            all = all && b->initialized[i];
        }
        spd_model_control mc;
        if (int rc = spd_model_get_control(b->model, &mc)) return rc;
        const bool same_start = !any || (mc.year == d[0] && mc.month == d[1] && mc.day == d[2] && mc.hour == d[3] && mc.minute == d[4] &&
                                         mc.current_step == 0);
        if (all || !same_start) {
            if (int rc = split_batch(b)) return rc;
            b = st->batch;
        }
    }
    regrouped();
    if (b->members == 1) {
        if (int rc = spd_model_init(b->model, d[0], d[1], d[2], d[3], d[4], nullptr)) return rc;
        b->advanced_without_check = false;
    } else {
        // initialise a scratch one-member model from this member's boundary fields and copy the resulting state into the
        // member's slot
        std::shared_ptr<Batch> scratch;
        if (int rc = new_batch(1, b->device, &scratch)) return rc;
        int rc = SPD_OK;
        if (b->sst_anom_allocated) rc = spd_model_init_sst_anom(scratch->model, b->n_months);
        spd_model_control mc;
        if (rc == SPD_OK) rc = spd_model_get_control(b->model, &mc);
        if (rc == SPD_OK) rc = spd_model_copy_member(scratch->model, 0, b->model, st->member, nullptr);
        if (rc == SPD_OK) rc = spd_model_set_flags(scratch->model, mc.land_coupling_flag, mc.sst_anomaly_coupling_flag, mc.increase_co2);
        if (rc == SPD_OK) rc = spd_model_set_co2(scratch->model, mc.air_absortivity_co2);
        if (rc == SPD_OK) rc = spd_model_init(scratch->model, d[0], d[1], d[2], d[3], d[4], nullptr);
        if (rc == SPD_OK) rc = spd_model_copy_member(b->model, st->member, scratch->model, 0, nullptr);
        if (rc == SPD_OK && !drvdev::device_synchronize()) rc = fail(SPD_E_DEVICE, "spd_init: device error");
        // (date, step counter 0 and the CO2 reference: the same values for every member of the batch, see above)
        if (rc == SPD_OK) rc = spd_model_mark_initialized(b->model, 0, d[0], d[1], d[2], d[3], d[4]);
        if (rc == SPD_OK) rc = spd_model_set_time_step(b->model, 2 * kDelt);
        if (rc != SPD_OK) return rc;  // (the scratch model dies at the end of this block)
    }
    b->initialized[st->member] = 1;
    c.now = c.start;
    c.month_idx = 1;
    *error_code = 0;
    return SPD_OK;
}

// init for a whole ensemble: the same as spd_init for every container in turn, except that containers which are -- all of them, in
// this list, with one start date -- the members of a device model nobody has initialised yet are initialised by ONE
// initialisation of that model (one pass over its members on the device, identical boundary sets preprocessed once) instead of
// member by member through a scratch model: 256 members in 20 ms instead of 2 s.  Same states, bit for bit.
int spd_init_ensemble(const int64_t *state_cnts, const int64_t *control_cnts, int32_t *error_codes, int32_t n) {
    if (n < 0 || (n > 0 && (!state_cnts || !control_cnts || !error_codes))) return fail(SPD_E_ARG, "spd_init_ensemble: bad argument");
    drvdev::DeviceGuard guard;
    LOCK;
    std::vector<char> done(n, 0);
    std::map<const Batch *, std::vector<int>> positions_of;
    for (int i = 0; i < n; ++i) {
        auto st = state_of(state_cnts[i]);
        if (!st || g_controls.find(control_cnts[i]) == g_controls.end())
            return fail(SPD_E_ARG, "spd_init_ensemble: not a live state / control container");
        positions_of[st->batch.get()].push_back(i);
    }
    for (auto &kv : positions_of) {
        const std::vector<int> &mine = kv.second;
        std::shared_ptr<Batch> b = state_of(state_cnts[mine[0]])->batch;
        if (b->members == 1 || static_cast<int>(mine.size()) != b->members) continue;
        bool fresh = true, one_start = true;
        for (int k = 0; k < b->members; ++k) fresh = fresh && !b->initialized[k];
        const Control &c0 = g_controls[control_cnts[mine[0]]];
        std::vector<char> seen(b->members, 0);
        for (int i : mine) {
            one_start = one_start && std::memcmp(g_controls[control_cnts[i]].start.ymdhm, c0.start.ymdhm, sizeof(c0.start.ymdhm)) == 0;
            seen[state_of(state_cnts[i])->member] = 1;
        }
        for (char s : seen) fresh = fresh && s;  // (every member exactly once)
        if (!fresh || !one_start) continue;
        const int32_t *d = c0.start.ymdhm;
        if (!drvdev::set_device(b->device)) return fail(SPD_E_DEVICE, "spd_init_ensemble: hipSetDevice failed");
        if (int rc = spd_model_init(b->model, d[0], d[1], d[2], d[3], d[4], nullptr)) return rc;
        if (!drvdev::null_stream_synchronize()) return fail(SPD_E_DEVICE, "spd_init_ensemble: device error");
        b->advanced_without_check = false;
        for (int i : mine) {
            b->initialized[state_of(state_cnts[i])->member] = 1;
            Control &c = g_controls[control_cnts[i]];
            c.now = c.start;
            c.month_idx = 1;
            error_codes[i] = 0;
            done[i] = 1;
        }
        regrouped();
    }
    for (int i = 0; i < n; ++i)
        if (!done[i])
            if (int rc = spd_init(state_cnts[i], control_cnts[i], &error_codes[i])) return rc;
    return SPD_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// parallel_step
// ---------------------------------------------------------------------------------------------------------------------
// what ONE call does with one group of its plan
struct GroupRun {
    Control before;    // model date before the step
    Control advanced;  // ... and after it (valid when slot >= 0)
    int slot = -1;     // pending check; -1: the members were not initialised; -2: the step could not be issued
    int32_t step = 0;  // the model's step counter after the step: what the range check reports on stderr
    bool behind = false;  // enqueued while the check of the step before it was still out
    int nsteps = 1;    // > 1 or `checked`: a multi-step call with the check of every step recorded on the device (spd_parallel_steps_begin)
    bool checked = false;
    int rc = SPD_OK;   // status of this group's device calls
    std::string error; // ... and its message
};
struct PendingStep {
    std::shared_ptr<const Plan> plan;
    std::vector<GroupRun> run;
    int nsteps = 0;  // 0: spd_parallel_step_begin; k >= 1: spd_parallel_steps_begin of k steps
};
std::map<int64_t, PendingStep> g_pending;
// Host-side order of the device work of the multi-group paths, for tests: (kind, group) pairs, kind 1 = step + check
// enqueued, 2 = waiting for the check of that group started, 3 = finished.  Off unless spd_driver_trace(1) was called.
std::atomic<bool> g_trace_on{false};
std::mutex g_trace_mutex;  // (its own lock: issue threads record while the calling thread holds the library's)
std::vector<int32_t> g_trace;
static void trace(int kind, int group) {
    if (!g_trace_on.load(std::memory_order_relaxed)) return;
    std::lock_guard<std::mutex> lock(g_trace_mutex);
    g_trace.push_back(kind);
    g_trace.push_back(group);
}

// Resolve the containers; gather independent one-member models into batched models -- per device and per set of members that
// agree in date, control flags and anomaly length, so one odd member or a second device never de-batches the rest --; split
// batches that are asked for in a different grouping; make the groups to step.
static int make_plan(const int64_t *state_cnts, const int64_t *control_cnts, int n, bool stretch, std::shared_ptr<const Plan> &out,
                     const char *who) {
    std::vector<std::shared_ptr<State>> states(n);
    std::vector<Control *> controls(n);
    {
        std::vector<int64_t> seen(state_cnts, state_cnts + n);
        std::sort(seen.begin(), seen.end());
        if (std::adjacent_find(seen.begin(), seen.end()) != seen.end())
            return fail(SPD_E_ARG, std::string(who) + ": the same state container twice");
    }
    for (int i = 0; i < n; ++i) {
        states[i] = state_of(state_cnts[i]);
        auto ci = g_controls.find(control_cnts[i]);
        if (!states[i] || ci == g_controls.end()) return fail(SPD_E_ARG, std::string(who) + ": not a live state / control container");
        controls[i] = &ci->second;
    }
    // independent one-member models -> batched models (once; later calls find them batched)
    std::vector<char> classed(n, 1);
    std::vector<spd_model_control> own(n);  // the control block of every initialised one-member model, read once
    for (int i = 0; i < n; ++i) {
        const Batch &b = *states[i]->batch;
        if (b.members != 1 || !b.initialized[0] || b.advanced_without_check) continue;
        if (int rc = spd_model_get_control(b.model, &own[i])) return rc;
        classed[i] = own[i].sppt_on ? 1 : 0;  // (an SPPT member keeps its own model: gather() has the why)
    }
    for (int i = 0; i < n; ++i) {
        if (classed[i]) continue;
        std::vector<std::shared_ptr<State>> cls;
        for (int j = i; j < n; ++j) {
            const Batch &bi = *states[i]->batch, &bj = *states[j]->batch;
            if (classed[j] || bj.device != bi.device || bj.n_months != bi.n_months || bj.sst_anom_allocated != bi.sst_anom_allocated ||
                !same_date(*controls[i], *controls[j]) || std::memcmp(&own[i], &own[j], sizeof(own[i])) != 0)
                continue;
            classed[j] = 1;
            cls.push_back(states[j]);
        }
        // From 32 members up a class becomes TWO device models: the call enqueues both before it waits for either (below), so
        // their kernels share the GPU and fill each other's tails -- the member groups of spd_model_step, one level up,
        // where no stream has to be forked and joined at every step of a host that calls once per model step.
        const size_t parts = device_model_parts(cls.size());
        for (size_t part = 0, first = 0; part < parts; ++part) {
            const size_t count = cls.size() / parts + (part < cls.size() % parts ? 1 : 0);
            std::vector<std::shared_ptr<State>> sub(cls.begin() + first, cls.begin() + first + count);
            first += count;
            if (sub.size() > 1) {
                bool done = false;
                if (int rc = gather(sub, &done)) return rc;
            }
        }
    }
    // Device models that are whole in the list: for a multi-step call the models of a device that agree in everything become ONE
    // (regroup has the why); for single steps a model that was merged that way is halved again.
    {
        static const size_t merge_up_to = [] {  // (the merged model exists beside its parts while the members are copied)
            const char *e = getenv("PYSPEEDY_AMD_DRIVER_MERGE");
            return static_cast<size_t>(e ? atoi(e) : 1024);
        }();
        std::map<const Batch *, std::vector<int>> of;  // model -> its positions in the argument list, in order
        std::vector<const Batch *> order;
        for (int i = 0; i < n; ++i) {
            const Batch *b = states[i]->batch.get();
            if (of.find(b) == of.end()) order.push_back(b);
            of[b].push_back(i);
        }
        auto whole = [&](const Batch *b) {
            const std::vector<int> &mine = of[b];
            if (static_cast<int>(mine.size()) != b->members) return false;
            for (int j : mine)
                if (!same_date(*controls[mine[0]], *controls[j])) return false;
            return true;
        };
        if (stretch) {
            std::map<int, std::vector<const Batch *>> by_device;
            for (const Batch *b : order)
                if (whole(b)) by_device[b->device].push_back(b);
            for (auto &kv : by_device) {
                // (models that agree in the date of their containers: the first such class of the device)
                std::vector<std::shared_ptr<State>> all;
                const Batch *lead = kv.second[0];
                int models = 0;
                for (const Batch *b : kv.second) {
                    if (!same_date(*controls[of[lead][0]], *controls[of[b][0]])) continue;
                    ++models;
                    for (int j : of[b]) all.push_back(states[j]);
                }
                if (models < 2 || all.size() > merge_up_to) continue;
                bool done = false;
                if (int rc = regroup(all, 1, &done)) return rc;
            }
        } else {
            for (const Batch *b : order) {
                if (!b->merged || !whole(b) || device_model_parts(static_cast<size_t>(b->members)) < 2) continue;
                std::vector<std::shared_ptr<State>> all;
                for (int j : of[b]) all.push_back(states[j]);
                bool done = false;
                if (int rc = regroup(all, 2, &done)) return rc;
            }
        }
    }
    auto plan = std::make_shared<Plan>();
    plan->stretch = stretch;
    plan->states.assign(state_cnts, state_cnts + n);
    plan->controls.assign(control_cnts, control_cnts + n);
    // (containers of one device model are found through a map from the model to its group: linear in n)
    std::map<const Batch *, std::vector<int>> positions_of;
    for (int i = 0; i < n; ++i) positions_of[states[i]->batch.get()].push_back(i);
    std::vector<char> handled(n, 0);
    for (int i = 0; i < n; ++i) {
        if (handled[i]) continue;
        std::shared_ptr<Batch> b = states[i]->batch;
        std::vector<int> mine = positions_of[b.get()];  // positions of the argument list that belong to this model
        bool whole = static_cast<int>(mine.size()) == b->members;
        for (int j : mine) whole = whole && same_date(*controls[i], *controls[j]);
        if (!whole) {  // a different grouping than the batch: take it apart and step this container on its own
            if (int rc = split_batch(b)) return rc;
            for (int j : mine) positions_of[states[j]->batch.get()].assign(1, j);
            b = states[i]->batch;
            mine.assign(1, i);
        }
        GroupPlan g;
        g.batch = b;
        g.positions = mine;
        for (int j : mine) {
            g.members.push_back(states[j]->member);
            g.control_ids.push_back(control_cnts[j]);
            handled[j] = 1;
        }
        plan->groups.push_back(std::move(g));
    }
    plan->epoch = g_epoch;  // (after the gathers and splits above, which moved it)
    out = plan;
    return SPD_OK;
}

// The plan of this argument list (the last call's, when it still holds) and a fresh run record per group.  (lock held)
static int plan_step(const int64_t *state_cnts, const int64_t *control_cnts, int n, std::shared_ptr<const Plan> &plan,
                     std::vector<GroupRun> &run, const char *who, bool stretch = false) {
    const size_t bytes = static_cast<size_t>(n) * sizeof(int64_t);
    plan.reset();
    for (size_t k = 0; k < g_plans.size(); ++k) {
        const Plan &p = *g_plans[k];
        if (p.epoch == g_epoch && p.stretch == stretch && static_cast<int>(p.states.size()) == n &&
            (n == 0 || (std::memcmp(p.states.data(), state_cnts, bytes) == 0 && std::memcmp(p.controls.data(), control_cnts, bytes) == 0))) {
            plan = g_plans[k];
            if (k > 0) std::rotate(g_plans.begin(), g_plans.begin() + k, g_plans.begin() + k + 1);  // most recently used first
            break;
        }
    }
    if (!plan) {
        if (int rc = make_plan(state_cnts, control_cnts, n, stretch, plan, who)) return rc;
        // (making the plan may have regrouped containers and emptied the list)
        if (g_plans.size() >= kKeptPlans) g_plans.pop_back();
        g_plans.insert(g_plans.begin(), plan);
    }
    run.assign(plan->groups.size(), GroupRun{});
    for (size_t i = 0; i < run.size(); ++i) {
        auto ci = g_controls.find(plan->groups[i].control_ids[0]);
        if (ci == g_controls.end()) return fail(SPD_E_ARG, std::string(who) + ": not a live control container");
        run[i].before = run[i].advanced = ci->second;
    }
    return SPD_OK;
}

static bool all_initialized(const Batch &b) {
    for (int i = 0; i < b.members; ++i)
        if (!b.initialized[i]) return false;
    return true;
}

// Model error code of a member whose step could not be issued or checked at all (a device or argument error of the call,
// not one of the reference's three codes): error_codes.f90 stops at -2.
constexpr int32_t kStepFailed = -3;

// Enqueue the step and the range check of one group on its model's stream; nothing waits.  (lock held)
// Everything that can refuse is asked BEFORE the step is enqueued (a free check slot, the control block); what fails from the
// step on is a device error, and it leaves the model marked: its state has moved while its date and codes say it has not, and
// it is not stepped again until its members are initialised anew.
static void issue_group(const GroupPlan &g, GroupRun &r, bool defer_check, int nsteps_checked = 0, bool shares_device = false) {
    Batch &b = *g.batch;
    if (!all_initialized(b)) return;  // slot stays -1: E_STATE_NOT_INITIALIZED
    int rc = SPD_OK;
    bool step_enqueued = false;
    if (nsteps_checked > 0) {  // k steps as ONE device call, every step's range check recorded by the device
        r.nsteps = nsteps_checked;
        r.checked = true;
        if (b.advanced_without_check)
            rc = fail(SPD_E_ARG, "speedy driver: an earlier step of this device model was enqueued but could not be checked; initialise its members again");
        if (rc == SPD_OK && !drvdev::set_device(b.device)) rc = fail(SPD_E_DEVICE, "speedy driver: hipSetDevice failed");
        if (rc == SPD_OK && !b.stream) rc = fail(SPD_E_DEVICE, "speedy driver: hipStreamCreate failed");
        if (rc == SPD_OK && spd_model_checks_in_flight(b.model) != 0)
            rc = fail(SPD_E_ARG, "speedy driver: a step of this device model is in flight; end it with spd_parallel_step_end first");
        if (rc == SPD_OK) rc = push_date(b, r.before);
        if (rc == SPD_OK) {
            // Several device models of ONE device in the call (the two halves an ensemble of 32 or more containers is kept in): they
            // are each other's member groups already -- each is stepped as ONE group on its own stream (in rounds of its own when it
            // is large), instead of splitting every half again into groups that would share the GPU six or eight ways.
            int32_t groups = 0;
            const bool one_group = shares_device && spd_model_get_option(b.model, "member_groups", &groups) == SPD_OK && groups > 1;
            if (one_group) (void)spd_model_set_option(b.model, "member_groups", 1);
            rc = spd_model_step_checked_begin(b.model, nsteps_checked, b.stream);
            if (one_group) (void)spd_model_set_option(b.model, "member_groups", groups);
            // (a refusal leaves the model as it was; a device error in the middle marks the model itself: spd_model_step)
            step_enqueued = rc == SPD_OK;
        }
        if (rc == SPD_OK) rc = pull_date(b, r.advanced, &r.step);
        if (rc == SPD_OK) {
            r.slot = 0;  // (no check slot: the codes wait in the model until spd_model_step_checked_end)
        } else {
            r.rc = rc;
            r.error = spd_last_error();
            r.slot = -2;
            if (step_enqueued) b.advanced_without_check = true;
        }
        return;
    }
    if (b.advanced_without_check)
        rc = fail(SPD_E_ARG, "speedy driver: an earlier step of this device model was enqueued but could not be checked; initialise its members again");
    if (rc == SPD_OK && !drvdev::set_device(b.device)) rc = fail(SPD_E_DEVICE, "speedy driver: hipSetDevice failed");
    if (rc == SPD_OK && !b.stream) rc = fail(SPD_E_DEVICE, "speedy driver: hipStreamCreate failed");  // (issue_all made it)
    if (rc == SPD_OK && spd_model_checks_in_flight(b.model) >= 2)  // (a step without its check is no step)
        rc = fail(SPD_E_ARG, "speedy driver: two steps of this device model are in flight already; end one with spd_parallel_step_end first");
    if (rc == SPD_OK) r.behind = spd_model_checks_in_flight(b.model) >= 1;
    if (rc == SPD_OK) rc = push_date(b, r.before);
    if (rc == SPD_OK) {
        step_enqueued = true;
        rc = spd_model_step(b.model, 1, b.stream);
    }
    if (rc == SPD_OK) {
        // (the overlapped form collects the check after the NEXT step has been enqueued: that step's first launch carries it)
        r.slot = defer_check ? spd_model_check_defer(b.model, 2, b.stream) : spd_model_check_begin(b.model, 2, b.stream);
        if (r.slot < 0) rc = r.slot;
    }
    if (rc == SPD_OK) rc = pull_date(b, r.advanced, &r.step);
    if (rc != SPD_OK) {
        r.rc = rc;
        r.error = spd_last_error();
        if (r.slot >= 0) {  // (the check is in flight: take it back so that the slot is free again)
            std::vector<int32_t> scratch(b.members);
            (void)spd_model_check_end(b.model, r.slot, scratch.data());
        }
        r.slot = -2;
        if (step_enqueued) b.advanced_without_check = true;
    }
}

// One host thread per GPU for the enqueue of a step.  A model step is six launches per device model; a host that drives N GPUs
// from one process -- the reference's own shape -- would issue 12 N launches per step from one thread, and from about four GPUs
// on the GPUs would wait for it (one launch costs the host 3-4 us, a 64-member step of one GPU lasts 270).  The calling thread
// issues the groups of the first device itself and hands the groups of every other device to that device's worker; it holds
// the library's lock all the while and waits for the workers before it goes on, so nothing else changes: a worker touches
// nothing but the device models it was handed.  Workers are created on first use and live for the life of the process
// (detached, asleep between calls).  PYSPEEDY_AMD_ISSUE_THREADS=0: everything from the calling thread; =2: a worker per device
// MODEL even on one device (how the path is rehearsed on a one-GPU box).
struct IssueWorker {
    std::mutex m;
    std::condition_variable cv;
    std::function<void()> job;
    bool busy = false;
    void loop() {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [&] { return static_cast<bool>(job); });
            std::function<void()> j = std::move(job);
            job = nullptr;
            lk.unlock();
            j();
            lk.lock();
            busy = false;
            cv.notify_all();
        }
    }
    void submit(std::function<void()> j) {
        std::lock_guard<std::mutex> lk(m);
        job = std::move(j);
        busy = true;
        cv.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return !busy; });
    }
};
std::map<int, IssueWorker *> g_issue_workers;  // by worker key (device, or device model in the rehearsal mode); under g_mutex

static IssueWorker *issue_worker(int key) {
    auto it = g_issue_workers.find(key);
    if (it != g_issue_workers.end()) return it->second;
    IssueWorker *w = new IssueWorker();  // (never destroyed: its thread may outlive every static of this library)
    std::thread([w] { w->loop(); }).detach();
    g_issue_workers[key] = w;
    return w;
}

// Enqueue the step and check of every group; after_issue(i) runs on the CALLING thread for every group, in order.  (lock held)
static void issue_all(const std::vector<GroupPlan> &groups, std::vector<GroupRun> &run, bool defer_check, int nsteps_checked = 0) {
    static const int mode = [] {
        const char *e = getenv("PYSPEEDY_AMD_ISSUE_THREADS");
        return e ? atoi(e) : 1;
    }();
    for (const GroupPlan &g : groups) {  // (streams are made here, by the calling thread, not by the device's issue thread)
        Batch &b = *g.batch;
        if (!b.stream && all_initialized(b) && !b.advanced_without_check && drvdev::set_device(b.device)) (void)stream_for(b);
    }
    std::map<int, std::vector<size_t>> by_key;  // worker key -> groups, in argument order
    for (size_t i = 0; i < groups.size(); ++i) by_key[mode == 2 ? static_cast<int>(i) : groups[i].batch->device].push_back(i);
    std::map<int, int> models_on;  // device -> device models of this call that live there
    for (const GroupPlan &g : groups) ++models_on[g.batch->device];
    auto shares = [&](size_t i) { return models_on[groups[i].batch->device] > 1; };
    if (mode == 0 || by_key.size() < 2) {
        for (size_t i = 0; i < groups.size(); ++i) {
            issue_group(groups[i], run[i], defer_check, nsteps_checked, shares(i));
            trace(1, static_cast<int>(i));
        }
        return;
    }
    std::vector<IssueWorker *> busy;
    const int mine = by_key.begin()->first;
    for (auto &kv : by_key) {
        if (kv.first == mine) continue;
        IssueWorker *w = issue_worker(kv.first);
        const std::vector<size_t> *list = &kv.second;
        w->submit([list, &groups, &run, defer_check, nsteps_checked, &models_on] {
            for (size_t i : *list) {
                issue_group(groups[i], run[i], defer_check, nsteps_checked, models_on.at(groups[i].batch->device) > 1);
                trace(1, static_cast<int>(i));
            }
        });
        busy.push_back(w);
    }
    for (size_t i : by_key[mine]) {
        issue_group(groups[i], run[i], defer_check, nsteps_checked, shares(i));
        trace(1, static_cast<int>(i));
    }
    for (IssueWorker *w : busy) w->wait();
}

// A check that was put off (the begin / end form: it rides in the next step's first launch) and has found no step to ride in
// is launched here, WITH the lock held: spd_model_check_end would otherwise launch it from the unlocked wait below -- a kernel
// launch and a change of the model's check bookkeeping beside whatever another host thread does to the same model under the
// lock (spd_set, a regrouping).  After this, collect_group only waits and reads.  (lock held)
static void settle_deferred(const GroupPlan &g, GroupRun &r) {
    if (r.slot < 0) return;
    Batch &b = *g.batch;
    int rc = SPD_OK;
    if (!drvdev::set_device(b.device)) rc = fail(SPD_E_DEVICE, "speedy driver: hipSetDevice failed");
    if (rc == SPD_OK) rc = spd_model_check_settle(b.model, r.slot);  // (only THIS step's check: the next one's waits for its step)
    if (rc != SPD_OK) {  // (collect_group still ends the slot; its codes no longer count)
        r.rc = rc;
        r.error = spd_last_error();
    }
}

// Wait for the check of one group.  (lock NOT held: other host threads may work on other containers meanwhile)
static void collect_group(const GroupPlan &g, GroupRun &r, std::vector<int32_t> &codes) {
    Batch &b = *g.batch;
    codes.assign(b.members, r.slot == -2 ? kStepFailed : -1);
    if (r.slot < 0) return;
    int rc = SPD_OK;
    if (!drvdev::set_device(b.device)) rc = fail(SPD_E_DEVICE, "speedy driver: hipSetDevice failed");
    if (rc == SPD_OK) rc = spd_model_check_end(b.model, r.slot, codes.data());
    if (rc != SPD_OK && r.rc == SPD_OK) {
        r.rc = rc;
        r.error = spd_last_error();
    }
    if (r.rc != SPD_OK) codes.assign(b.members, kStepFailed);
}

// What check_diagnostics writes to unit 0 for a state out of range (diagnostics.f90:69-70: `write(0, *) "Model variables out of
// accepted range"; write(0, *) "step =", state%current_step`), once per failing state.  List-directed output as gfortran -- the
// reference's compiler -- formats it: a leading blank, a default integer in 12 columns (the flang-built oracle library prints
// ` step = 36`).  One fwrite per message, so that the lines of concurrent host threads do not interleave inside one.
static void report_out_of_range(int32_t step) {
    char text[96];
    const int n = std::snprintf(text, sizeof(text), " Model variables out of accepted range\n step =%12d\n", static_cast<int>(step));
    if (n > 0) std::fwrite(text, 1, static_cast<size_t>(n), stderr);
}

// Hand the codes out and settle the dates: speedy.f90:57-71 advances the date only after a successful check.  (lock held)
// dates_ran_ahead: the begin / end form moved the dates at _begin already (and a later _begin may have moved them again): only
// a member whose check failed gets the date from before its step back.
static void settle_group(const GroupPlan &g, const GroupRun &r, const std::vector<int32_t> &codes, int32_t *error_codes,
                         bool dates_ran_ahead) {
    bool any_failed = false;
    for (size_t k = 0; k < g.positions.size(); ++k) {
        const int32_t code = codes[g.members[k]];
        error_codes[g.positions[k]] = code;
        any_failed = any_failed || code != 0;
        if (code == -2) {
            // The overlapped form enqueues step k + 1 before it has seen the check of step k.  When step k failed, the failure of
            // step k + 1 is one the reference's loop cannot produce (it stops at the first code, speedy.py:398-405): not reported.
            std::vector<int32_t> &told = g.batch->failed_step;
            if (told.size() != static_cast<size_t>(g.batch->members)) told.assign(g.batch->members, -1);
            const bool consequence = dates_ran_ahead && r.behind && told[g.members[k]] == r.step - 1;
            if (!consequence) report_out_of_range(r.step);
            told[g.members[k]] = r.step;
        }
        auto ci = g_controls.find(g.control_ids[k]);
        if (ci == g_controls.end() || (code == 0 && dates_ran_ahead)) continue;
        const Control &to = code == 0 ? r.advanced : r.before;
        ci->second.now = to.now;
        ci->second.month_idx = to.month_idx;
    }
    if (r.rc != SPD_OK && r.slot >= 0) g.batch->advanced_without_check = true;  // (stepped, and the check could not be collected)
    if (any_failed) regrouped();  // the members of a model may now disagree about the date
}

// the status of the call: the first group that failed (its message becomes spd_last_error again); the others were stepped
static int first_failure(const std::vector<GroupRun> &run) {
    for (const GroupRun &r : run)
        if (r.rc != SPD_OK) return fail(r.rc, r.error);
    return SPD_OK;
}

int spd_parallel_step(const int64_t *state_cnts, const int64_t *control_cnts, int32_t *error_codes, int32_t n) {
    if (n < 0 || (n > 0 && (!state_cnts || !control_cnts || !error_codes))) return fail(SPD_E_ARG, "spd_parallel_step: bad argument");
    drvdev::DeviceGuard guard;
    std::unique_lock<std::recursive_mutex> lock(g_mutex);
    std::shared_ptr<const Plan> plan;
    std::vector<GroupRun> run;
    if (int rc = plan_step(state_cnts, control_cnts, n, plan, run, "spd_parallel_step")) return rc;
    const std::vector<GroupPlan> &groups = plan->groups;
    // One device model, or several -- one per GPU of a one-process ensemble, the two models 32 or more containers of a device
    // are kept in, members that could not be batched --: every model's step and check are enqueued before the host waits for
    // any of them, so the devices (and the models that share one) work side by side; a model that fails does not keep the
    // others from being stepped; and the lock is given up while the host waits, so that other host threads can step THEIR
    // containers meanwhile (the reference's parallel_step is `!f2py threadsafe`).
    issue_all(groups, run, false);
    for (size_t i = 0; i < groups.size(); ++i) settle_deferred(groups[i], run[i]);  // (the synchronous form defers nothing: a no-op)
    lock.unlock();
    std::vector<std::vector<int32_t>> codes(groups.size());
    for (size_t i = 0; i < groups.size(); ++i) {
        trace(2, static_cast<int>(i));
        collect_group(groups[i], run[i], codes[i]);
        trace(3, static_cast<int>(i));
    }
    lock.lock();
    for (size_t i = 0; i < groups.size(); ++i) settle_group(groups[i], run[i], codes[i], error_codes, false);
    return first_failure(run);
}

int spd_parallel_step_begin(const int64_t *state_cnts, const int64_t *control_cnts, int32_t n, int64_t *token) {
    if (n < 0 || !token || (n > 0 && (!state_cnts || !control_cnts))) return fail(SPD_E_ARG, "spd_parallel_step_begin: bad argument");
    drvdev::DeviceGuard guard;
    LOCK;
    PendingStep p;
    if (int rc = plan_step(state_cnts, control_cnts, n, p.plan, p.run, "spd_parallel_step_begin")) return rc;
    // (the range check of this step is put off: the next _begin's first launch carries it.  PYSPEEDY_AMD_DEFER_CHECK=0: a launch of
    // its own behind the step, as in the synchronous form)
    static const bool defer = !(getenv("PYSPEEDY_AMD_DEFER_CHECK") && atoi(getenv("PYSPEEDY_AMD_DEFER_CHECK")) == 0);
    issue_all(p.plan->groups, p.run, defer);  // (a group that cannot be issued reports at _end; the others go ahead)
    for (size_t i = 0; i < p.run.size(); ++i) {
        const GroupPlan &g = p.plan->groups[i];
        GroupRun &r = p.run[i];
        if (r.slot < 0) continue;
        for (int64_t id : g.control_ids) {  // the dates run ahead of the check; _end puts a failed member's date back
            auto ci = g_controls.find(id);
            if (ci == g_controls.end()) continue;
            ci->second.now = r.advanced.now;
            ci->second.month_idx = r.advanced.month_idx;
        }
    }
    *token = g_next++;
    g_pending[*token] = std::move(p);
    return SPD_OK;
}

int spd_parallel_step_end(int64_t token, int32_t *error_codes) {
    drvdev::DeviceGuard guard;
    std::unique_lock<std::recursive_mutex> lock(g_mutex);
    auto it = g_pending.find(token);
    if (it == g_pending.end() || !error_codes || it->second.nsteps != 0) return fail(SPD_E_ARG, "spd_parallel_step_end: not a pending step");
    PendingStep p = std::move(it->second);
    g_pending.erase(it);
    for (size_t i = 0; i < p.plan->groups.size(); ++i) settle_deferred(p.plan->groups[i], p.run[i]);
    lock.unlock();
    const std::vector<GroupPlan> &groups = p.plan->groups;
    std::vector<std::vector<int32_t>> codes(groups.size());
    for (size_t i = 0; i < groups.size(); ++i) {
        trace(2, static_cast<int>(i));
        collect_group(groups[i], p.run[i], codes[i]);
        trace(3, static_cast<int>(i));
    }
    lock.lock();
    for (size_t i = 0; i < groups.size(); ++i) {
        const GroupPlan &g = groups[i];
        if (p.run[i].slot == -1) {  // not initialised: the dates were never touched
            for (size_t k = 0; k < g.positions.size(); ++k) error_codes[g.positions[k]] = -1;
            continue;
        }
        settle_group(g, p.run[i], codes[i], error_codes, true);
    }
    return first_failure(p.run);
}

// k steps of the same containers as ONE device call per device model, with the range check of EVERY step recorded on the device
// (spd_model_step_checked_begin): what a host with the reference's time loop (pyspeedy/speedy.py:396-405, 572-586) may do for the
// steps between two due callbacks -- nothing looks at the state in between.  The dates in the control containers move k steps at
// _begin; _end waits, hands out per member the code of the FIRST step whose check failed (the reference's loop stops there) and
// how many steps the member completed before it, writes the reference's stderr text for that step and puts the member's date
// back to the one after its last accepted step (speedy.f90:57-71 returns before advance_date).
int spd_parallel_steps_begin(const int64_t *state_cnts, const int64_t *control_cnts, int32_t n, int32_t n_steps, int64_t *token) {
    if (n < 0 || !token || (n > 0 && (!state_cnts || !control_cnts))) return fail(SPD_E_ARG, "spd_parallel_steps_begin: bad argument");
    if (n_steps < 1 || n_steps > 4096) return fail(SPD_E_ARG, "spd_parallel_steps_begin: 1 ... 4096 steps per call");
    drvdev::DeviceGuard guard;
    LOCK;
    PendingStep p;
    p.nsteps = n_steps;
    if (int rc = plan_step(state_cnts, control_cnts, n, p.plan, p.run, "spd_parallel_steps_begin", true)) return rc;
    issue_all(p.plan->groups, p.run, false, n_steps);  // (a group that cannot be issued reports at _end; the others go ahead)
    for (size_t i = 0; i < p.run.size(); ++i) {
        const GroupPlan &g = p.plan->groups[i];
        GroupRun &r = p.run[i];
        if (r.slot < 0) continue;
        for (int64_t id : g.control_ids) {  // the dates run ahead of the checks; _end puts a failed member's date back
            auto ci = g_controls.find(id);
            if (ci == g_controls.end()) continue;
            ci->second.now = r.advanced.now;
            ci->second.month_idx = r.advanced.month_idx;
        }
    }
    *token = g_next++;
    g_pending[*token] = std::move(p);
    return SPD_OK;
}

int spd_parallel_steps_end(int64_t token, int32_t *error_codes, int32_t *steps_done) {
    drvdev::DeviceGuard guard;
    std::unique_lock<std::recursive_mutex> lock(g_mutex);
    auto it = g_pending.find(token);
    if (it == g_pending.end() || !error_codes || it->second.nsteps < 1) return fail(SPD_E_ARG, "spd_parallel_steps_end: not a pending multi-step call");
    PendingStep p = std::move(it->second);
    g_pending.erase(it);
    lock.unlock();  // (other host threads may step THEIR containers while this one waits)
    const std::vector<GroupPlan> &groups = p.plan->groups;
    std::vector<std::vector<int32_t>> failed(groups.size()), accepted(groups.size());
    for (size_t i = 0; i < groups.size(); ++i) {
        GroupRun &r = p.run[i];
        Batch &b = *groups[i].batch;
        if (r.slot < 0) continue;
        trace(2, static_cast<int>(i));
        failed[i].assign(b.members, -1);
        accepted[i].assign(static_cast<size_t>(b.members) * 7, 0);
        int rc = SPD_OK;
        if (!drvdev::set_device(b.device)) rc = fail(SPD_E_DEVICE, "speedy driver: hipSetDevice failed");
        if (rc == SPD_OK) rc = spd_model_step_checked_end(b.model, failed[i].data(), accepted[i].data());
        if (rc != SPD_OK && r.rc == SPD_OK) {
            r.rc = rc;
            r.error = spd_last_error();
        }
        trace(3, static_cast<int>(i));
    }
    lock.lock();
    bool any_failed = false;
    for (size_t i = 0; i < groups.size(); ++i) {
        const GroupPlan &g = groups[i];
        const GroupRun &r = p.run[i];
        for (size_t k = 0; k < g.positions.size(); ++k) {
            const int pos = g.positions[k], mem = g.members[k];
            int32_t code = 0, done = p.nsteps;
            if (r.slot == -1) {  // not initialised: nothing was touched
                code = -1;
                done = 0;
            } else if (r.slot == -2 || r.rc != SPD_OK) {  // could not be issued, or its codes could not be collected
                code = kStepFailed;
                done = 0;
            } else if (failed[i][mem] >= 0) {
                code = -2;
                done = failed[i][mem];
            }
            error_codes[pos] = code;
            if (steps_done) steps_done[pos] = done;
            any_failed = any_failed || code != 0;
            auto ci = g_controls.find(g.control_ids[k]);
            if (code == -2) {
                const int32_t *a = accepted[i].data() + 7 * static_cast<size_t>(mem);
                report_out_of_range(a[0] + 1);  // the step counter of the step that failed (diagnostics.f90:70)
                std::vector<int32_t> &told = g.batch->failed_step;
                if (told.size() != static_cast<size_t>(g.batch->members)) told.assign(g.batch->members, -1);
                told[mem] = a[0] + 1;
                if (ci != g_controls.end()) {  // the date after the last accepted step (speedy.f90:57-71 returns before advance_date)
                    for (int d = 0; d < 5; ++d) ci->second.now.ymdhm[d] = a[1 + d];
                    ci->second.month_idx = a[6];
                }
            } else if (code == kStepFailed && r.slot != -1 && ci != g_controls.end()) {  // the dates stay (see the header)
                ci->second.now = r.before.now;
                ci->second.month_idx = r.before.month_idx;
            }
        }
        if (r.rc != SPD_OK && r.slot >= 0) g.batch->advanced_without_check = true;  // (stepped, and the checks could not be collected)
    }
    if (any_failed) regrouped();  // the members of a model may now disagree about the date
    return first_failure(p.run);
}

// The shared boundary fields (orography, masks, albedo, vegetation, the monthly climatologies and -- when both have
// the same length -- the SST anomalies) of container `root` into every other container of the list, device to device: the
// one exchange of a sharded ensemble, for a host that keeps all members in one process.  (One process per GPU: the same
// broadcast is ensemble.broadcast_boundary_conditions over RCCL through torch.distributed.)  The fields cross to another GPU
// ONCE: into the first container of the list that lives there -- all GPUs at once with one RCCL broadcast over xGMI
// (spd_model_broadcast_vars; PYSPEEDY_AMD_BROADCAST=peer, or an RCCL that cannot be loaded: one hipMemcpyPeerAsync per GPU
// instead, and spd_broadcast_boundary_stats says which it was) --; the other containers of that GPU take them from it with local
// copies queued behind on the same stream.  Devices are synchronised once before (the copies run on the null stream of the
// destination device, which must not overtake what the models' own streams still hold) and once after the call.
int spd_broadcast_boundary(const int64_t *state_cnts, int32_t n, int32_t root) {
    if (n < 1 || !state_cnts || root < 0 || root >= n) return fail(SPD_E_ARG, "spd_broadcast_boundary: bad argument");
    static const char *const kBoundary[] = {"orog", "fmask_orig", "alb0", "veg_high", "veg_low", "stl12", "snowd12", "soil_wc_l1",
                                            "soil_wc_l2", "soil_wc_l3", "sst12", "sea_ice_frac12", "sst_anom"};
    drvdev::DeviceGuard guard;
    LOCK;
    auto src = state_of(state_cnts[root]);
    if (!src) return fail(SPD_E_ARG, "spd_broadcast_boundary: not a live state container");
    std::vector<std::shared_ptr<State>> dst(n);
    std::map<int, std::vector<int>> on_device;  // device -> positions of the list (root excluded), in list order
    for (int i = 0; i < n; ++i) {
        if (i == root) continue;
        dst[i] = state_of(state_cnts[i]);
        if (!dst[i]) return fail(SPD_E_ARG, "spd_broadcast_boundary: not a live state container");
        on_device[dst[i]->batch->device].push_back(i);
    }
    const int src_device = src->batch->device;
    auto sync_device = [](int d) { return drvdev::set_device(d) && drvdev::device_synchronize(); };
    if (!sync_device(src_device)) return fail(SPD_E_DEVICE, "spd_broadcast_boundary: device error");
    for (auto &kv : on_device)
        if (kv.first != src_device && !sync_device(kv.first)) return fail(SPD_E_DEVICE, "spd_broadcast_boundary: device error");
    g_broadcast_stats = {0, 0, 0};
    g_broadcast_note = "local copies only (one device)";
    auto with_anomalies = [&](int i) {
        const Batch &d = *dst[i]->batch, &s = *src->batch;
        return d.sst_anom_allocated == s.sst_anom_allocated && d.n_months == s.n_months;
    };
    // ---- across GPUs: the first container of every other device receives from the root, all of them in one collective
    std::vector<char> filled(n, 0);  // 1: has the 12 fixed fields; 2: and the anomalies
    const char *transport = getenv("PYSPEEDY_AMD_BROADCAST");
    std::vector<int> receivers;
    for (auto &kv : on_device)
        if (kv.first != src_device) receivers.push_back(kv.second.front());
    if (!receivers.empty() && transport && std::strcmp(transport, "peer") == 0) g_broadcast_note = "peer copies (PYSPEEDY_AMD_BROADCAST=peer)";
    if (!receivers.empty() && !(transport && std::strcmp(transport, "peer") == 0)) {
        bool all_anom = true;
        for (int i : receivers) all_anom = all_anom && with_anomalies(i);
        std::vector<spd_model_handle> models{src->batch->model};
        std::vector<int> members{src->member};
        for (int i : receivers) {
            models.push_back(dst[i]->batch->model);
            members.push_back(dst[i]->member);
        }
        const int brc = spd_model_broadcast_vars(models.data(), members.data(), static_cast<int>(models.size()), 0, kBoundary, all_anom ? 13 : 12);
        if (brc == SPD_OK) {
            for (int i : receivers) filled[i] = all_anom ? 2 : 1;
            g_broadcast_stats.collective_devices = static_cast<int>(receivers.size());
            g_broadcast_note = "one RCCL broadcast to " + std::to_string(receivers.size()) + " other device(s)";
        } else if (brc == SPD_E_TIMEOUT) {
            // the collective was enqueued and did not complete inside its bound: whatever would be queued behind it on those
            // devices' null streams would wait with it.  No fallback; the caller gets the reason.
            g_broadcast_note = std::string("failed: ") + spd_last_error();
            return fail(SPD_E_TIMEOUT, g_broadcast_note);
        } else {
            // (RCCL cannot be loaded, refused to initialise, or its initialisation did not come back in time: nothing was
            // enqueued, the fields go point to point below and the stats show that no device was reached collectively)
            g_broadcast_note = std::string("peer copies, because: ") + spd_last_error();
        }
    }
    int rc = SPD_OK;
    for (auto &kv : on_device) {
        // the source of this device's copies: the root itself on its own device, elsewhere the first container that has
        // received everything (a container whose anomaly length differs from the root's receives 12 fields and cannot pass 13 on)
        std::shared_ptr<State> local = kv.first == src_device ? src : nullptr;
        for (int i : kv.second) {
            const Batch &d = *dst[i]->batch;
            const bool anom = with_anomalies(i);
            if (filled[i] == 0 || (filled[i] == 1 && anom)) {
                const std::shared_ptr<State> &from = local ? local : src;
                const bool crosses = from->batch->device != d.device;
                const char *const *names = filled[i] == 1 ? kBoundary + 12 : kBoundary;  // (only the anomalies are missing)
                const int count = filled[i] == 1 ? 1 : (anom ? 13 : 12);
                rc = spd_model_copy_vars_enqueue(d.model, dst[i]->member, from->batch->model, from->member, names, count, nullptr);
                if (rc != SPD_OK) break;
                ++(crosses ? g_broadcast_stats.peer_copies : g_broadcast_stats.local_copies);
            }
            if (!local && anom) local = dst[i];
        }
        if (rc != SPD_OK) break;
    }
    // (the call is synchronous, like every call of this interface)
    for (auto &kv : on_device)
        if (!sync_device(kv.first) && rc == SPD_OK) rc = fail(SPD_E_DEVICE, "spd_broadcast_boundary: device error");
    if (!receivers.empty() && !sync_device(src_device) && rc == SPD_OK) rc = fail(SPD_E_DEVICE, "spd_broadcast_boundary: device error");
    return rc;
}

int spd_broadcast_boundary_stats(int32_t *peer_copies, int32_t *local_copies, int32_t *collective_devices) {
    LOCK;
    if (peer_copies) *peer_copies = g_broadcast_stats.peer_copies;
    if (local_copies) *local_copies = g_broadcast_stats.local_copies;
    if (collective_devices) *collective_devices = g_broadcast_stats.collective_devices;
    return SPD_OK;
}

const char *spd_broadcast_boundary_note(void) {
    LOCK;
    static thread_local std::string copy;
    copy = g_broadcast_note;
    return copy.c_str();
}

int spd_driver_trace(int32_t on) {
    std::lock_guard<std::mutex> lock(g_trace_mutex);
    g_trace_on = on != 0;
    g_trace.clear();
    return SPD_OK;
}

int spd_driver_trace_read(int32_t *pairs, int32_t capacity) {
    std::lock_guard<std::mutex> lock(g_trace_mutex);
    const int n = static_cast<int>(g_trace.size() / 2);
    if (pairs)
        for (int i = 0; i < n && i < capacity; ++i) {
            pairs[2 * i] = g_trace[2 * i];
            pairs[2 * i + 1] = g_trace[2 * i + 1];
        }
    return n;
}

int spd_step(int64_t state_cnt, int64_t control_cnt, int32_t *error_code) {
    if (!error_code) return fail(SPD_E_ARG, "spd_step: null argument");
    return spd_parallel_step(&state_cnt, &control_cnt, error_code, 1);
}

int spd_check(int64_t state_cnt, int32_t *error_code) {
    if (!error_code) return fail(SPD_E_ARG, "spd_check: null argument");
    drvdev::DeviceGuard guard;
    LOCK;
    auto st = state_of(state_cnt);
    if (!st) return fail(SPD_E_ARG, "spd_check: not a live state container");
    Batch &b = *st->batch;
    if (!b.initialized[st->member]) {
        *error_code = -1;
        return SPD_OK;
    }
    if (!drvdev::set_device(b.device)) return fail(SPD_E_DEVICE, "spd_check: hipSetDevice failed");
    std::vector<int32_t> codes(b.members, 0);
    if (int rc = spd_model_check(b.model, 1, codes.data(), nullptr, nullptr)) return rc;
    *error_code = codes[st->member];
    if (*error_code == -2) report_out_of_range(spd_model_current_step(b.model));
    return SPD_OK;
}

static int transform(int64_t state_cnt, int which, const char *who) {
    drvdev::DeviceGuard guard;
    LOCK;
    auto st = state_of(state_cnt);
    if (!st) return fail(SPD_E_ARG, std::string(who) + ": not a live state container");
    Batch &b = *st->batch;
    if (!drvdev::set_device(b.device)) return fail(SPD_E_DEVICE, std::string(who) + ": hipSetDevice failed");
    int rc;
    if (which == 0) rc = spd_model_spectral2grid(b.model, st->member, 1, nullptr);
    else if (which == 1) rc = spd_model_grid2spectral(b.model, st->member, 1, nullptr);
    else rc = spd_model_grid_filter(b.model, st->member, 1, nullptr);
    if (rc == SPD_OK && !drvdev::null_stream_synchronize()) rc = fail(SPD_E_DEVICE, std::string(who) + ": device error");
    return rc;
}
int spd_transform_spectral2grid(int64_t state_cnt) { return transform(state_cnt, 0, "spd_transform_spectral2grid"); }
int spd_transform_grid2spectral(int64_t state_cnt) { return transform(state_cnt, 1, "spd_transform_grid2spectral"); }
int spd_apply_grid_filter(int64_t state_cnt) { return transform(state_cnt, 2, "spd_apply_grid_filter"); }

// ---------------------------------------------------------------------------------------------------------------------
// registry access
// ---------------------------------------------------------------------------------------------------------------------
static int access(int64_t state_cnt, const char *name, void *buf, size_t bytes, bool set) {
    const char *who = set ? "spd_set" : "spd_get";
    drvdev::DeviceGuard guard;
    LOCK;
    auto st = state_of(state_cnt);
    const RegVar *v = find_var(name);
    if (!st || !buf) return fail(SPD_E_ARG, std::string(who) + ": not a live state container / null buffer");
    if (!v) return fail(SPD_E_ARG, std::string(who) + ": unknown variable '" + (name ? name : "") + "'");
    std::shared_ptr<Batch> b = st->batch;
    const size_t need = var_bytes(*v, *b);
    if (bytes != need)
        return fail(SPD_E_SIZE, std::string(who) + ": '" + name + "' is " + std::to_string(need) + " bytes (Array shape missmatch)");
    if (!drvdev::set_device(b->device)) return fail(SPD_E_DEVICE, std::string(who) + ": hipSetDevice failed");
    switch (v->where) {
        case Device:
            return set ? spd_model_set(b->model, name, st->member, buf, bytes) : spd_model_get(b->model, name, st->member, buf, bytes);
        case HostOnly: {
            std::vector<double> &h = st->host[name];
            h.resize(need / sizeof(double), 0.0);
            if (set) std::memcpy(h.data(), buf, need);
            else std::memcpy(buf, h.data(), need);
            return SPD_OK;
        }
        case Table: {
            if (set) return fail(SPD_E_ARG, std::string("spd_set: '") + name + "' is a read-only table of the device context");
            std::vector<double> t64;
            std::vector<float> t32;
            if (int rc = table_values(*v, b->ctx, t64, t32)) return rc;
            if (v->dtype == SPD_T_FLOAT32) std::memcpy(buf, t32.data(), need);
            else std::memcpy(buf, t64.data(), need);
            return SPD_OK;
        }
        case Scalar: break;
    }
    spd_model_control mc;
    if (int rc = spd_model_get_control(b->model, &mc)) return rc;
    const std::string s(name);
    if (!set) {
        if (s == "current_step") *static_cast<int32_t *>(buf) = mc.current_step;
        else if (s == "increase_co2") *static_cast<int32_t *>(buf) = mc.increase_co2;
        else if (s == "compute_shortwave") *static_cast<int32_t *>(buf) = st->compute_shortwave ? 1 : 0;
        else if (s == "land_coupling_flag") *static_cast<int32_t *>(buf) = mc.land_coupling_flag;
        else if (s == "sst_anomaly_coupling_flag") *static_cast<int32_t *>(buf) = mc.sst_anomaly_coupling_flag;
        else if (s == "air_absortivity_co2") *static_cast<double *>(buf) = mc.air_absortivity_co2;
        else *static_cast<double *>(buf) = mc.ablco2_ref;
        return SPD_OK;
    }
    if (s == "current_step") return fail(SPD_E_ARG, "spd_set: current_step is advanced by the model");
    if (s == "compute_shortwave") {  // recomputed from the step counter by every step (speedy.f90:53); kept as a mirror
        st->compute_shortwave = *static_cast<const int32_t *>(buf) != 0;
        return SPD_OK;
    }
    spd_model_control want = mc;
    if (s == "increase_co2") want.increase_co2 = *static_cast<const int32_t *>(buf) != 0;
    else if (s == "land_coupling_flag") want.land_coupling_flag = *static_cast<const int32_t *>(buf) != 0;
    else if (s == "sst_anomaly_coupling_flag") want.sst_anomaly_coupling_flag = *static_cast<const int32_t *>(buf) != 0;
    else if (s == "air_absortivity_co2") want.air_absortivity_co2 = *static_cast<const double *>(buf);
    else want.ablco2_ref = *static_cast<const double *>(buf);
    if (std::memcmp(&want, &mc, sizeof(mc)) == 0) return SPD_OK;
    if (b->members > 1) {  // the scalars are per model: a member that wants its own leaves the batch
        if (int rc = split_batch(b)) return rc;
        b = st->batch;
    }
    if (b->initialized[0]) return spd_model_set_control(b->model, &want);
    if (int rc = spd_model_set_flags(b->model, want.land_coupling_flag, want.sst_anomaly_coupling_flag, want.increase_co2)) return rc;
    return spd_model_set_co2(b->model, want.air_absortivity_co2);
}

int spd_get(int64_t state_cnt, const char *name, void *buf, size_t bytes) { return access(state_cnt, name, buf, bytes, false); }
int spd_set(int64_t state_cnt, const char *name, const void *buf, size_t bytes) {
    return access(state_cnt, name, const_cast<void *>(buf), bytes, true);
}

int spd_get_shape(int64_t state_cnt, const char *name, int32_t *shape, int32_t *ndim) {
    LOCK;
    auto st = state_of(state_cnt);
    const RegVar *v = find_var(name);
    if (!st || !shape || !ndim) return fail(SPD_E_ARG, "spd_get_shape: not a live state container / null argument");
    if (!v) return fail(SPD_E_ARG, std::string("spd_get_shape: unknown variable '") + (name ? name : "") + "'");
    *ndim = v->ndim;
    for (int d = 0; d < v->ndim; ++d) shape[d] = v->shape[d] < 0 ? st->batch->n_months + 2 : v->shape[d];
    if (std::strcmp(v->name, "sst_anom") == 0 && !st->batch->sst_anom_allocated)
        for (int d = 0; d < v->ndim; ++d) shape[d] = 0;  // get_<v>_shape of an unallocated array, speedy_driver.f90.j2:277-281
    return SPD_OK;
}

int spd_is_array(const char *name, int32_t *is_array) {
    const RegVar *v = find_var(name);
    if (!v || !is_array) return fail(SPD_E_ARG, std::string("spd_is_array: unknown variable '") + (name ? name : "") + "'");
    *is_array = v->where == Scalar ? 0 : 1;
    return SPD_OK;
}

int spd_registry_entry(int32_t index, char *name, int32_t *dtype, int32_t *ndim, int32_t *shape, int32_t *is_read_only) {
    if (index < 0 || index >= kRegistryCount) return kRegistryCount;
    const RegVar &v = kRegistry[index];
    if (name) {
        std::strncpy(name, v.name, 31);
        name[31] = '\0';
    }
    if (dtype) *dtype = v.dtype;
    if (ndim) *ndim = v.ndim;
    if (shape)
        for (int d = 0; d < 5; ++d) shape[d] = d < v.ndim ? v.shape[d] : 0;
    if (is_read_only) *is_read_only = v.where == Table ? 1 : 0;
    return kRegistryCount;
}

int spd_driver_model(int64_t state_cnt, void **model, int32_t *member, int32_t *members_in_model) {
    LOCK;
    auto st = state_of(state_cnt);
    if (!st || !model) return fail(SPD_E_ARG, "spd_driver_model: not a live state container");
    *model = st->batch->model;
    if (member) *member = st->member;
    if (members_in_model) *members_in_model = st->batch->members;
    return SPD_OK;
}

int spd_driver_stats(int64_t state_cnt, int32_t *models_alive, int32_t *members_in_model) {
    LOCK;
    if (models_alive) *models_alive = g_models_alive.load();
    if (members_in_model) {
        auto st = state_of(state_cnt);
        *members_in_model = st ? st->batch->members : 0;
    }
    return SPD_OK;
}

}  // extern "C"
