// Ensemble model object: device-resident state of M members + the step driver (time_stepping.f90:38-147 `step`,
// tendencies.f90:11-39 `get_tendencies`) built from the hot-path kernels and the dynamics kernels.
// C ABI: the spd_model_* functions of include/pyspeedy_amd.h.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/pyspeedy_amd.h"
#include "context.hpp"
#include "model.hpp"
#include "coupler_point.hpp"
#include "diagnostics_block.hpp"
#include "bounded_call.hpp"
#include "launch_events.hpp"
#include "sppt_point.hpp"
#include "stream_apart.hpp"
#include "surface.hpp"

namespace spd {
hipError_t run_spec2grid_table(const DeviceTables &T, const FieldDesc *table, int nfields, hipStream_t st);
hipError_t run_grid2spec_table(const DeviceTables &T, const FieldDesc *table, int nfields, hipStream_t st);
hipError_t run_physics(const DeviceTables &T, const spd_physics_args &a, int nmembers, int fp32, hipStream_t s);
hipError_t run_dyn_physics(const ModelPtrs &P, const DynDeviceTables &D, const DeviceTables &T, const spd_physics_args &a,
                           int first, int nmembers, int fp32, int store32, int diag, hipStream_t s);
hipError_t run_geopotential(const ModelPtrs &P, const DynDeviceTables &D, int first, int count, int tl, const SpptArgs *sppt,
                            hipStream_t s);
hipError_t run_dyn_grid(const ModelPtrs &P, const DynDeviceTables &D, int M, hipStream_t s);
hipError_t run_spectral_step(const ModelPtrs &P, const DeviceTables &T, const DynDeviceTables &D, int M, int first, int count,
                             int j1, double dt, double eps, const CouplerArgs *cpl, bool early, hipStream_t s);
hipError_t run_diagnostics(const ModelPtrs &P, const DeviceTables &T, int M, int tl, int *err, double *diag, int ticket,
                           hipStream_t s);
hipError_t run_diagnostics_range(const ModelPtrs &P, const DeviceTables &T, int first, int count, int tl, int *err, double *diag,
                                 int ticket, hipStream_t s);
hipError_t run_coupler(const SurfacePtrs &S, int first, int count, const TimeInterp &w, int day, int land_coupling,
                       int sst_anomaly, int anom_planes, int fresh, hipStream_t s);
hipError_t run_forcing(const SurfacePtrs &S, int first, int count, const ZonalDevice &Z, double gamlat, double *corh_t,
                       double *corh_q, hipStream_t s);
hipError_t run_rest_state(const RestPtrs &R, int M, const RestConsts &c, hipStream_t s);
hipError_t run_scale_orog(const double *orog, double *phi0, long n, hipStream_t s);
hipError_t run_change_storage(double *array, long n, bool to_float, void *scratch, hipStream_t s);
hipError_t run_land_sea_init(const LandSeaPtrs &P, const LandSeaConsts &K, int first, int count, double *rows, hipStream_t s);
hipError_t run_multi_copy(const CopyList &L, hipStream_t s);
hipError_t run_spec2grid_table_check(const DeviceTables &T, const FieldDesc *table, int nfields, const CheckArgs &check, int members,
                                     hipStream_t st);
hipError_t run_copy_from_first(double *v, long n, int M, const int *flags, hipStream_t s);
hipError_t run_rest_surface(const double *phis0, double *forog, double *surf_ps, double *surf_q, const RestConsts &c, long n,
                            hipStream_t s);
hipError_t run_grid2spec(const DeviceTables &T, int stage, const double *src, double *dst, int prescale, int nfields,
                         hipStream_t stream);
hipError_t run_spec2grid(const DeviceTables &T, int stage, const double *src, double *dst, int kcos, int nfields,
                         hipStream_t stream);
hipError_t run_scale(const double *in, double *out, const double *table, double sign, int nfields, hipStream_t s);
SpptArgs sppt_args(double *spec, const DeviceTables &T, int M, unsigned long long seed, long long member_base, long long step,
                   int first);
hipError_t run_sppt_update(const SpptArgs &a, hipStream_t s);
hipError_t run_export_units(double *q, double *phi, double *ps, long n2d, hipStream_t s);
hipError_t run_export_pack(const void *src, bool src_is_float, void *dst, int levels, int count, hipStream_t s);
hipError_t run_log_ps(const double *ps_grid, double *out, long n2d, hipStream_t s);
hipError_t run_vort2vel(const DeviceTables &T, const double *vor, const double *div, double *ucos, double *vcos, int nfields,
                        hipStream_t s);
hipError_t run_vel2vort(const DeviceTables &T, const double *ucos, const double *vcos, double *vor, double *div, int nfields,
                        hipStream_t s);
hipError_t run_export_spec_units(double *tr, double *phi, long ncomplex, hipStream_t s);
}  // namespace spd

using namespace spd;

namespace {
constexpr int NG = IX * IL;
constexpr size_t C = 2;  // doubles per complex

struct RegEntry {
    void *ptr;            // device base
    size_t bytes_member;  // bytes per member (as fp64: the size the registry and the C boundary speak of)
    bool f32 = false;     // stored as fp32 (in the first half of the allocation) while the model's physics precision is fp32
};
}  // namespace

struct spd_model {
    spd_context *ctx = nullptr;
    int M = 0;
    ModelPtrs P{};
    const spd_dyn_tables *dyn = nullptr;            // the context's tables of the current time step (nullptr: none set yet)
    std::unique_ptr<spd_dyn_tables> dyn_private;    // only when the context already holds kMaxDynSteps other time steps
    DynDeviceTables D{};
    spd_physics_args pa{};
    // Device memory of the model: a few large zero-filled blocks the arrays are carved from (arena_alloc).  A model has some 170
    // arrays and tables; one hipMalloc + hipMemset + hipFree each made creating and closing a state container the most
    // expensive calls of a host that follows the reference's sequence (1.3 ms and 1.8 ms per one-member model).
    struct Block {
        char *base;
        size_t size, used;
    };
    std::vector<Block> blocks;
    std::map<std::string, RegEntry> reg;
    FieldDesc *inv_table[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};  // [dynamics time level j2 (0-based)][phi buffer]
    FieldDesc *inv_table_sppt[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};  // the same + 8 SPPT pattern transforms per member
    // ... and both with the physics-only outputs (time-level-1 T, q, phi, ln ps, lowest-level u, v) stored as fp32 (cfg 5)
    FieldDesc *inv_table32[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}}, *inv_table_sppt32[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    FieldDesc *fwd_table = nullptr;
    // Geopotential, double-buffered.  spectral_step_kernel ends by computing the geopotential the NEXT step needs (from the
    // temperature it has just advanced) into the buffer that is not in use; the next step switches to it instead of running
    // geopotential_kernel.  The registry's "phi" is always the buffer the last step USED (the reference's state%phi after a
    // step).  phi_ahead is dropped whenever something outside the step may have changed the temperature.
    double *phi_buf[2] = {nullptr, nullptr};
    int phi_cur = 0;
    // fold_geo: on by default for small ensembles (<= 8 members), where the step is bound by launch and dependent-latency
    // chains and one launch less is worth 1-3 %; at 64 members the longer spectral_step_kernel costs 2 % more than the
    // geopotential launch it saves (A/B in one session, profiles/).  PYSPEEDY_AMD_FOLD_GEO=0 / 1 overrides.
    bool phi_ahead = false, fold_geo = true;
    bool groups_apart = true;  // every group stream created so far was measured to run side by side with the others
    int *d_err = nullptr;
    double *d_diag = nullptr;
    // asynchronous range check (spd_model_check_begin / _end): two pinned result slots with their events
    int *h_err[2] = {nullptr, nullptr}, *h_err_sync = nullptr;  // (h_err_sync: pinned staging of the synchronous check)
    hipEvent_t err_event[2] = {nullptr, nullptr};
    int next_slot = 0;
    bool slot_busy[2] = {false, false};  // begun and not yet ended
    int check_ticket = 0, slot_ticket[2] = {0, 0};  // every range-check launch publishes its codes under a ticket of its own
    // A check whose launch is put off until the next step (spd_model_check_defer): it then rides in that step's spectral -> grid
    // launch.  Launched on its own as soon as anything else would look at or change the state first (settle_deferred_check).
    struct DeferredCheck {
        bool active = false;
        int slot = -1, time_level = 2;
        hipStream_t stream = nullptr;
    } deferred;
    hipStream_t slot_stream[2] = {nullptr, nullptr};  // the stream a slot's launch went out on
    bool slot_rode[2] = {false, false};               // ... inside a step's launch (no completion event of its own)
    int checks_alone = 0, checks_rode = 0;            // range checks launched on their own / carried by a step's launch
    double air_absortivity_co2 = 6.0;  // model_state_def.py:320 default
    // device copies of the dt-dependent tables (re-uploaded by set_time_step)
    // surface / coupler state, calendar and run control (do_single_step, speedy.f90:20-74)
    SurfacePtrs S{};
    Calendar cal;
    int current_step = 0;
    bool initialized = false;
    // SPPT (csrc/sppt.hip): AR(1) spectral pattern [M][8][992] complex, its grid-space image [M][8][NG]
    bool sppt_on = false, sppt_first = true;
    unsigned long long sppt_seed = 0;
    long long sppt_member_base = 0, sppt_step = 0;
    double *sppt_spec = nullptr, *sppt_grid = nullptr;
    // Members are stepped in `nchunks` groups on separate HIP streams (spd_model_step): a group's kernels overlap with
    // the other groups' (different kernels, complementary resources, no idle tail between dependent launches).
    // spectral -> grid transforms per member and step: 77 = the reference's 91 minus the 14 whose results nothing reads (u, v
    // above the lowest level at the physics' time level: physics.f90:93-94 computes them, get_surface_fluxes only uses level
    // kx; every registry variable stays bitwise identical, tests/test_run_gpu.py).  PYSPEEDY_AMD_PRUNE_DEAD=0 restores all 91.
    int inv_per_member = 77;
    int nchunks = 1;
    // Large ensembles in multi-step calls: from 4 x `block_members` members up, spd_model_step(m, n) takes the members in ROUNDS of
    // nchunks x block_members -- a round through ALL n steps before the next round starts (members never exchange data).  A group's
    // spectral step is then followed on its stream by the spectral -> grid launch of its own next step, which reads what was just
    // written while it is still in the 256 MB Infinity Cache, as in a 64-member ensemble; with 128 members per group it is not,
    // and a member-step costs 5-10 % more (profiles/r05_members_per_gpu.txt).  The host-side state of the step (calendar, step
    // counter, geopotential buffer, SPPT counter, CO2) is rewound for every round.  0: off (PYSPEEDY_AMD_BLOCK_MEMBERS, option
    // "block_members").
    int block_members = 32;
    hipStream_t cstream[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t cev[4] = {nullptr, nullptr, nullptr, nullptr}, ev_start = nullptr, ev_offset = nullptr;
    bool split_dyn_physics = false;  // PYSPEEDY_AMD_SPLIT_DYN=1: separate dynamics and physics launches (for measurements)
    // spd_model_set_physics_precision (BASELINE cfg 5): column physics arithmetic in fp32 AND fp32 storage of what only the
    // column physics reads back (RegEntry::f32: its time-level-1 inputs, the persisted radiation state, diagnostics-only outputs)
    int phys_fp32 = 0;
    bool phys_store32 = true;  // option physics_storage32 / PYSPEEDY_AMD_PHYS_STORE32: 0 keeps fp64 storage under the fp32 physics
    bool stored32 = false;     // how the RegEntry::f32 arrays are stored right now (= phys_fp32 && phys_store32)
    // A change of that storage converts the arrays in place, one by one; a device error in the middle leaves some of them
    // converted and `stored32` unable to say which.  The model then refuses every call that would read or advance its state.
    std::string poisoned;
    // ... and a device error in the middle of a step (some launches of it out, others not; or one member group a step ahead of
    // another): the STATE is then inconsistent, not the storage -- spd_model_init, which rebuilds every array from the boundary
    // fields, makes the model usable again; nothing else does.
    std::string step_poison;
    int fail_launch_after = -1;  // fault injection for tests (option "fail_launch_after"): the n-th step_range of the next call fails
    // spd_model_step_checked_begin / _end: the range check of EVERY step of a multi-step call, recorded by the device into pinned
    // host memory [steps][M] (4 * ticket + flag, as the single checks do) by check blocks that ride in the next step's
    // spectral -> grid launch; the last step's check is a launch of its own behind the call.
    int *h_steps_err = nullptr;
    int steps_cap = 0, steps_pending = 0, steps_ticket = 0;
    hipEvent_t steps_event = nullptr;
    std::vector<int32_t> steps_accepted;  // [steps + 1][7]: step counter, y, m, d, h, min, month_idx before the call and after each step
    // Dead-store elimination inside multi-step calls (PYSPEEDY_AMD_DIAG_EVERY_STEP=1 switches it off): only the LAST step
    // of a spd_model_step call stores the physics outputs that no later kernel reads -- the host can only look at the
    // state between calls, and every earlier value would be overwritten before that.
    bool diag_every_step = false;
    // The coupler's climatology interpolation is valid for a day (surface.hip): true after a coupling, false after anything
    // wrote to the state from outside the step
    bool surf_cache_valid = false;
    // The coupling of the step rides in the launch of spectral_step_kernel (tail blocks, dynamics.hip) instead of being a
    // launch of its own: PYSPEEDY_AMD_COUPLER_IN_SPECTRAL=0 / 1
    bool coupler_in_spectral = true;
    int spectral_early = -1;  // spectral_step_kernel with all loads up front: -1 = for launches of up to 8 members, 0 / 1 = never / always
    int land_coupling_flag = 1, sst_anomaly_flag = 1, increase_co2 = 0, anom_planes = 3;
    double ablco2_ref = 6.0;
    double *corh_t = nullptr, *corh_q = nullptr, *scratch_spec = nullptr;  // [M][NG], [M][NG], [2][M][992] complex
    double *orog = nullptr, *phi0 = nullptr, *fmask_orig = nullptr, *veg_high = nullptr, *veg_low = nullptr,
           *soil_wc_l1 = nullptr, *soil_wc_l2 = nullptr, *soil_wc_l3 = nullptr, *bmask_land = nullptr, *bmask_sea = nullptr;
    // optional profiling with HIP events on the launch stream: level 1 brackets the dominant kernel (the spec2grid table
    // launch) only, level 2 every kernel of the step (spd_model_profile; kernel ids SPD_K_* of pyspeedy_amd.h)
    int profile = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
    std::vector<int> prof_fields;  // fields of each profiled launch
    std::vector<int> prof_kernel;  // kernel id of each profiled launch
    size_t prof_used = 0;
    // grid-space copies of the prognostic variables in output units (prognostics.f90:125-219) and their transform tables
    double *u_grid = nullptr, *v_grid = nullptr, *t_grid = nullptr, *q_grid = nullptr, *phi_grid = nullptr, *ps_grid = nullptr;
    FieldDesc *exp_inv_table[2] = {nullptr, nullptr}, *exp_fwd_table[2] = {nullptr, nullptr};  // 41 / 40 per member; [phi buffer]
};

namespace spd {
LaunchEvents &pending_launch_events() {
    static thread_local LaunchEvents ev;
    return ev;
}
}  // namespace spd

static int m_fail(int code, const std::string &msg) { return spd_set_error(code, msg); }
static int usable(const spd_model *m, const char *who, bool about_to_init = false) {
    if (!m->poisoned.empty()) return m_fail(SPD_E_DEVICE, std::string(who) + ": this model is unusable: " + m->poisoned);
    if (!about_to_init && !m->step_poison.empty())
        return m_fail(SPD_E_ARG, std::string(who) + ": this model is unusable until it is initialised again (spd_model_init): " + m->step_poison);
    return SPD_OK;
}
static int apply_storage(spd_model *m, bool want32);  // (with spd_model_set_physics_precision)
static int settle_deferred_check(spd_model *m);        // (with spd_model_check_defer)
static int ensure_group_streams(spd_model *m, int G);  // (with spd_model_step)
static int ensure_steps_record(spd_model *m, int nsteps);

// (A failed runtime call also leaves its code behind as the thread's "last error", and the launch wrappers of the kernels report
// hipGetLastError(): a hipMalloc that ran out of memory would come back as the "failure" of the next launch of an unrelated model.
// The code is reported HERE, once, and cleared.)
#define M_HIP(call)                                                                   \
    do {                                                                              \
        hipError_t e_ = (call);                                                       \
        if (e_ != hipSuccess) {                                                       \
            (void)hipGetLastError();                                                  \
            return m_fail(SPD_E_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_)); \
        }                                                                             \
    } while (0)

// `bytes` of zero-filled device memory that lives as long as the model.  Every array starts on a 256-byte boundary.  The first
// block is sized for everything spd_model_create asks for (18.5 MB per member); whatever comes later -- SST anomalies of a
// longer period, the tables of another configuration -- opens further blocks.
static int arena_alloc(spd_model *m, size_t bytes, void **out) {
    constexpr size_t kAlign = 256;
    bytes = (bytes + kAlign - 1) / kAlign * kAlign;
    if (m->blocks.empty() || m->blocks.back().used + bytes > m->blocks.back().size) {
        const size_t M = static_cast<size_t>(m->M);
        const size_t want = m->blocks.empty() ? M * (19u << 20) + (1u << 20) : M * (2u << 20) + (1u << 20);
        const size_t size = bytes > want ? bytes : want;
        void *p = nullptr;
        M_HIP(hipSetDevice(m->ctx->device));
        {  // a block of this size that a dead model of this context left behind?
            std::lock_guard<std::mutex> lock(m->ctx->idle_mutex);
            auto &idle = m->ctx->idle_blocks;
            for (size_t i = 0; i < idle.size() && !p; ++i)
                if (idle[i].size == size) {
                    p = idle[i].base;
                    m->ctx->idle_bytes -= size;
                    idle.erase(idle.begin() + static_cast<long>(i));
                }
        }
        if (!p) M_HIP(hipMalloc(&p, size));
        m->blocks.push_back({static_cast<char *>(p), size, 0});
        M_HIP(hipMemset(p, 0, size));
    }
    spd_model::Block &b = m->blocks.back();
    *out = b.base + b.used;
    b.used += bytes;
    return SPD_OK;
}

static int dalloc(spd_model *m, size_t doubles, double **out, const char *name = nullptr, size_t bytes_member = 0) {
    void *p = nullptr;
    if (int rc = arena_alloc(m, doubles * sizeof(double), &p)) return rc;
    *out = static_cast<double *>(p);
    if (name) m->reg[name] = RegEntry{p, bytes_member};
    return SPD_OK;
}

static int upload_const(spd_model *m, const double *src, size_t n, const double **dst) {
    double *p = nullptr;
    if (int rc = dalloc(m, n, &p)) return rc;
    M_HIP(hipMemcpy(p, src, n * sizeof(double), hipMemcpyHostToDevice));
    *dst = p;
    return SPD_OK;
}

// Descriptor tables built on the host and uploaded TOGETHER: one allocation, one host-to-device copy for all the tables of a
// build (creating a model builds nine; one blocking copy each was a quarter of what creating a one-member model cost).
struct TableBatch {
    std::vector<std::vector<FieldDesc>> tables;
    std::vector<FieldDesc **> outs;
    void add(std::vector<FieldDesc> &&t, FieldDesc **out) {
        tables.push_back(std::move(t));
        outs.push_back(out);
    }
    int upload(spd_model *m) {
        constexpr size_t kAlign = 256;
        std::vector<size_t> at(tables.size());
        size_t total = 0;
        for (size_t i = 0; i < tables.size(); ++i) {
            at[i] = total;
            total += (tables[i].size() * sizeof(FieldDesc) + kAlign - 1) / kAlign * kAlign;
        }
        if (total == 0) return SPD_OK;
        void *d = nullptr;
        if (int rc = arena_alloc(m, total, &d)) return rc;
        std::vector<char> packed(total, 0);
        for (size_t i = 0; i < tables.size(); ++i) std::memcpy(packed.data() + at[i], tables[i].data(), tables[i].size() * sizeof(FieldDesc));
        M_HIP(hipMemcpy(d, packed.data(), total, hipMemcpyHostToDevice));
        for (size_t i = 0; i < tables.size(); ++i) *outs[i] = reinterpret_cast<FieldDesc *>(static_cast<char *>(d) + at[i]);
        tables.clear();
        outs.clear();
        return SPD_OK;
    }
};

// The four spectral -> grid descriptor tables ([dynamics time level][phi buffer]) of the step.  with_sppt: every member's
// entries are followed by the 8 transforms of its SPPT pattern (spectral AR(1) state -> grid, kcos = 1), so that they ride
// in the same launch instead of being a launch of their own.
static void build_inverse_tables(spd_model *m, bool with_sppt, bool phys_as_float, FieldDesc *(&out)[2][2], TableBatch &batch) {
    const int M = m->M;
    const ModelPtrs &P = m->P;
    const spd_physics_args &pa = m->pa;
    auto spec = [](double *base, size_t field) { return base + field * NSPEC * C; };
    auto grid = [](double *base, size_t field) { return base + field * NG; };
    // an output of the physics' time level: fp32 in the first half of its array when the column physics wants it so
    const int pf = phys_as_float ? kGridAsFloat : 0;
    auto pgrid = [&](const double *base, size_t field) {
        double *b = const_cast<double *>(base);
        return phys_as_float ? reinterpret_cast<double *>(reinterpret_cast<float *>(b) + field * NG) : b + field * NG;
    };
    for (int j2 = 0; j2 < 4; ++j2) {
        const int par = j2 >> 1;  // (j2 & 1) = dynamics time level, par = phi buffer
        std::vector<FieldDesc> t;
        t.reserve(static_cast<size_t>(M) * 99);
        for (int i = 0; i < M; ++i) {
            const size_t w = static_cast<size_t>(i) * 8, st = (static_cast<size_t>(i) * 2 + (j2 & 1)) * 8, s1 = static_cast<size_t>(i) * 2 * 8;
            // Entry order inside a member is variable-major, level-minor: workgroups are handed to the 8 XCDs round-robin by
            // workgroup id, so all entries of level k of a member land on the same XCD and the four transforms that read
            // vor_k / div_k (vorticity, divergence, u, v) share them through that XCD's L2 instead of fetching them four times.
            FieldDesc e[11][8];
            for (int k = 0; k < 8; ++k) {
                e[0][k] = {spec(P.vor, st + k), grid(P.vorg, w + k), 1, 0};
                e[1][k] = {spec(P.div, st + k), grid(P.divg, w + k), 1, 0};
                // u, v: vort2vel applied while the coefficients are staged (FieldDesc::mode 1 / 2), at the dynamics' time
                // level and at time level 1 for the physics (tendencies.f90:109-118, physics.f90:89-94)
                e[2][k] = {spec(P.vor, st + k), grid(P.ug2, w + k), 2, 1, spec(P.div, st + k)};
                e[3][k] = {spec(P.vor, st + k), grid(P.vg2, w + k), 2, 2, spec(P.div, st + k)};
                e[4][k] = {spec(P.vor, s1 + k), pgrid(pa.ug, w + k), 2 | pf, 1, spec(P.div, s1 + k)};
                e[5][k] = {spec(P.vor, s1 + k), pgrid(pa.vg, w + k), 2 | pf, 2, spec(P.div, s1 + k)};
                e[6][k] = {spec(P.t, st + k), grid(P.tg2, w + k), 1, 0};
                e[7][k] = {spec(P.tr, st + k), grid(P.trg2, w + k), 1, 0};
                e[8][k] = {spec(P.t, s1 + k), pgrid(pa.tg, w + k), 1 | pf, 0};
                e[9][k] = {spec(P.tr, s1 + k), pgrid(pa.qg, w + k), 1 | pf, 0};
                e[10][k] = {spec(m->phi_buf[par], w + k), pgrid(pa.phig, w + k), 1 | pf, 0};
            }
            const bool prune = m->inv_per_member == 77;
            for (int v = 0; v < 11; ++v)
                for (int k = 0; k < 8; ++k)
                    if (!(prune && (v == 4 || v == 5) && k < 7)) t.push_back(e[v][k]);
            // grad ln ps at the dynamics' time level (tendencies.f90:144-146): gradient applied while staging (mode 3 / 4)
            t.push_back({spec(P.ps, static_cast<size_t>(i) * 2 + (j2 & 1)), grid(P.px, i), 2, 3, nullptr});
            t.push_back({spec(P.ps, static_cast<size_t>(i) * 2 + (j2 & 1)), grid(P.py, i), 2, 4, nullptr});
            t.push_back({spec(P.ps, static_cast<size_t>(i) * 2), pgrid(pa.pslg, i), 1 | pf, 0});
            if (with_sppt)
                for (int k = 0; k < 8; ++k) t.push_back({spec(m->sppt_spec, w + k), grid(m->sppt_grid, w + k), 1, 0});
        }
        batch.add(std::move(t), &out[j2 & 1][par]);
    }
}

// the descriptor tables of the cfg 5 step (physics-only outputs as fp32), built when they are first needed
static int ensure_tables32(spd_model *m) {
    TableBatch batch;
    if (!m->inv_table32[0][0]) build_inverse_tables(m, false, true, m->inv_table32, batch);
    if (m->sppt_spec && !m->inv_table_sppt32[0][0]) build_inverse_tables(m, true, true, m->inv_table_sppt32, batch);
    return batch.upload(m);
}

static int build_tables(spd_model *m) {
    const int M = m->M;
    const ModelPtrs &P = m->P;
    auto spec = [](double *base, size_t field) { return base + field * NSPEC * C; };
    auto grid = [](double *base, size_t field) { return base + field * NG; };
    TableBatch batch;
    build_inverse_tables(m, false, false, m->inv_table, batch);
    std::vector<FieldDesc> t;
    t.reserve(static_cast<size_t>(M) * 73);
    const size_t pair = static_cast<size_t>(M) * 8;
    for (int i = 0; i < M; ++i) {
        const size_t w = static_cast<size_t>(i) * 8;
        for (int k = 0; k < 8; ++k) {
            // grid_vel2vort(..., kcos = 2): rows pre-multiplied by cosgr (spectral.f90:229-235) -> flag 1
            t.push_back({grid(P.utend, w + k), spec(P.specu, w + k), 1, 0});
            t.push_back({grid(P.vtend, w + k), spec(P.specv, w + k), 1, 0});
            t.push_back({grid(P.utg, w + k), spec(P.specu, pair + w + k), 1, 0});
            t.push_back({grid(P.vtg, w + k), spec(P.specv, pair + w + k), 1, 0});
            t.push_back({grid(P.uqg, w + k), spec(P.specu, 2 * pair + w + k), 1, 0});
            t.push_back({grid(P.vqg, w + k), spec(P.specv, 2 * pair + w + k), 1, 0});
            t.push_back({grid(P.ttend, w + k), spec(P.spec_tt, w + k), 0, 0});
            t.push_back({grid(P.trtend, w + k), spec(P.spec_tr, w + k), 0, 0});
            t.push_back({grid(P.keg, w + k), spec(P.spec_ke, w + k), 0, 0});
        }
        t.push_back({grid(P.psdtg, i), spec(P.spec_ps, i), 0, 0});
    }
    batch.add(std::move(t), &m->fwd_table);
    // export tables (prognostics.f90:125-219), time level 1.  sv holds ucos | vcos in the [M][2][8] layout of vor / div.
    for (int par = 0; par < 2; ++par) {
        std::vector<FieldDesc> ti, tf;
        const size_t half = static_cast<size_t>(M) * 16;  // fields in the ucos block of sv
        for (int i = 0; i < M; ++i) {
            const size_t w = static_cast<size_t>(i) * 8, s1 = static_cast<size_t>(i) * 16;
            for (int k = 0; k < 8; ++k) {
                ti.push_back({spec(P.sv, s1 + k), grid(m->u_grid, w + k), 2, 0});
                ti.push_back({spec(P.sv, half + s1 + k), grid(m->v_grid, w + k), 2, 0});
                ti.push_back({spec(P.t, s1 + k), grid(m->t_grid, w + k), 1, 0});
                ti.push_back({spec(P.tr, s1 + k), grid(m->q_grid, w + k), 1, 0});
                ti.push_back({spec(m->phi_buf[par], w + k), grid(m->phi_grid, w + k), 1, 0});
                tf.push_back({grid(m->u_grid, w + k), spec(P.sv, s1 + k), 1, 0});  // kcos = 2: rows times cosgr
                tf.push_back({grid(m->v_grid, w + k), spec(P.sv, half + s1 + k), 1, 0});
                tf.push_back({grid(m->t_grid, w + k), spec(P.t, s1 + k), 0, 0});
                tf.push_back({grid(m->q_grid, w + k), spec(P.tr, s1 + k), 0, 0});
                tf.push_back({grid(m->phi_grid, w + k), spec(m->phi_buf[par], w + k), 0, 0});
            }
            ti.push_back({spec(P.ps, static_cast<size_t>(i) * 2), grid(m->ps_grid, i), 1, 0});
        }
        batch.add(std::move(ti), &m->exp_inv_table[par]);
        batch.add(std::move(tf), &m->exp_fwd_table[par]);
    }
    return batch.upload(m);
}

extern "C" {

// The context's dynamics tables for time step `dt` (0: the dt-independent ones only), made on first demand.  `owner`: the model
// that asks -- when the context is full (a host that keeps inventing time steps) it gets a set of its own instead, rebuilt and
// uploaded in place at every change as every model's was before the tables moved to the context.
constexpr size_t kMaxDynSteps = 64;
static int dyn_upload(spd_context *ctx, spd_dyn_tables &set, const spd_dyn_tables *base, double *slab) {
    const DynHostTables &dh = set.host;
    DynDeviceTables &D = set.dev;
    std::vector<double> packed;
    auto put = [&](const std::vector<double> &v) {
        const double *at = slab + packed.size();
        packed.insert(packed.end(), v.begin(), v.end());
        return at;
    };
    if (!base) {  // dt-independent: horizontal diffusion coefficients and the Coriolis parameter
        D.dmp = put(dh.dmp); D.dmpd = put(dh.dmpd); D.dmps = put(dh.dmps);
        D.coriol = put(std::vector<double>(ctx->host.coriol.begin(), ctx->host.coriol.end()));
        for (int k = 0; k < 8; ++k) {
            D.tcorv[k] = dh.tcorv[k]; D.qcorv[k] = dh.qcorv[k]; D.tref[k] = dh.tref[k]; D.tref2[k] = dh.tref2[k];
            D.tref3[k] = dh.tref3[k]; D.xgeop1[k] = dh.xgeop1[k]; D.xgeop2[k] = dh.xgeop2[k]; D.geo_corf[k] = dh.geo_corf[k];
            D.dhs[k] = ctx->host.dhs[k]; D.dhsr[k] = ctx->host.dhsr[k]; D.fsgr[k] = ctx->host.fsgr[k];
        }
    } else {
        D = base->dev;
        D.dmp1 = put(dh.dmp1); D.dmp1d = put(dh.dmp1d); D.dmp1s = put(dh.dmp1s); D.elz = put(dh.elz); D.xj = put(dh.xj);
        D.xc = put(std::vector<double>(dh.xc.begin(), dh.xc.end()));
        D.xd = put(std::vector<double>(dh.xd.begin(), dh.xd.end()));
        for (int k = 0; k < 8; ++k) D.dhsx[k] = dh.dhsx[k];
    }
    M_HIP(hipMemcpy(slab, packed.data(), packed.size() * sizeof(double), hipMemcpyHostToDevice));
    return SPD_OK;
}
constexpr size_t kDynSlabDoubles = 4 * NSPEC + 8 * 8 * (MX + NX + 1) + 128;  // the larger of the two sets

static int dyn_tables(spd_context *ctx, double dt, spd_model *owner, const spd_dyn_tables **out) {
    std::lock_guard<std::mutex> lock(ctx->dyn_mutex);
    M_HIP(hipSetDevice(ctx->device));
    auto slab = [&](double **p) -> int {
        void *d = nullptr;
        M_HIP(hipMalloc(&d, kDynSlabDoubles * sizeof(double)));
        ctx->allocations.push_back(d);
        *p = static_cast<double *>(d);
        return SPD_OK;
    };
    if (!ctx->dyn_base) {
        auto base = std::make_unique<spd_dyn_tables>(DynHostTables(ctx->host));
        double *d = nullptr;
        if (int rc = slab(&d)) return rc;
        if (int rc = dyn_upload(ctx, *base, nullptr, d)) return rc;
        ctx->dyn_base = std::move(base);
    }
    if (dt == 0.0) {
        *out = ctx->dyn_base.get();
        return SPD_OK;
    }
    auto it = ctx->dyn_by_step.find(dt);
    if (it != ctx->dyn_by_step.end()) {
        *out = it->second.get();
        return SPD_OK;
    }
    auto set = std::make_unique<spd_dyn_tables>(ctx->dyn_base->host);
    set->host.set_time_step(ctx->host, dt);
    if (ctx->dyn_by_step.size() < kMaxDynSteps) {
        double *d = nullptr;
        if (int rc = slab(&d)) return rc;
        if (int rc = dyn_upload(ctx, *set, ctx->dyn_base.get(), d)) return rc;
        *out = set.get();
        ctx->dyn_by_step.emplace(dt, std::move(set));
        return SPD_OK;
    }
    if (!owner) return m_fail(SPD_E_ARG, "dyn_tables: the context holds its maximum of time steps");
    double *d = owner->dyn_private ? const_cast<double *>(owner->dyn_private->dev.dmp1) : nullptr;
    if (!d) {
        void *p = nullptr;
        if (int rc = arena_alloc(owner, kDynSlabDoubles * sizeof(double), &p)) return rc;
        d = static_cast<double *>(p);
    }
    M_HIP(hipDeviceSynchronize());  // kernels in flight may still read the set this one replaces
    if (int rc = dyn_upload(ctx, *set, ctx->dyn_base.get(), d)) return rc;
    owner->dyn_private = std::move(set);
    *out = owner->dyn_private.get();
    return SPD_OK;
}

int spd_model_create(spd_handle h, int nmembers, spd_model_handle *out) {
    if (!h || !out) return m_fail(SPD_E_ARG, "spd_model_create: null argument");
    if (nmembers <= 0) return m_fail(SPD_E_ARG, "spd_model_create: nmembers must be positive");
    *out = nullptr;
    M_HIP(hipSetDevice(h->device));
    spd_model *m = new spd_model();
    m->ctx = h;
    m->M = nmembers;
    if (const char *env = getenv("PYSPEEDY_AMD_SPLIT_DYN")) m->split_dyn_physics = atoi(env) != 0;
    m->inv_per_member = 77;
    if (const char *env = getenv("PYSPEEDY_AMD_PRUNE_DEAD")) m->inv_per_member = atoi(env) != 0 ? 77 : 91;
    if (const char *env = getenv("PYSPEEDY_AMD_DIAG_EVERY_STEP")) m->diag_every_step = atoi(env) != 0;
    m->fold_geo = nmembers <= 8;
    if (const char *env = getenv("PYSPEEDY_AMD_COUPLER_IN_SPECTRAL")) m->coupler_in_spectral = atoi(env) != 0;
    if (const char *env = getenv("PYSPEEDY_AMD_SPECTRAL_EARLY")) m->spectral_early = atoi(env);
    if (const char *env = getenv("PYSPEEDY_AMD_FOLD_GEO")) m->fold_geo = atoi(env) != 0;
    if (const char *env = getenv("PYSPEEDY_AMD_PHYS_STORE32")) m->phys_store32 = atoi(env) != 0;
    // Member groups on separate streams (spd_model_step).  Measured per step against one group (profiles/r03_member_groups.txt):
    // 16 members and fewer: nothing to fill, 0 ... +10 %; 20 members: 2 groups -3 %; 24 ... 48: 3 groups -9 ... -12 % (2 groups
    // -7 ... -10 %); 64: 2 groups -10 %, 3 groups -9.5 %; 96 / 128: -4 % / -3 % either way; 4 groups are slower everywhere.
    // PYSPEEDY_AMD_CHUNKS = 1 ... 4 or spd_model_set_option("member_groups") override.  While spd_model_profile is on the step
    // is issued as ONE group on the caller's stream: with overlapping launches the duration of a kernel is not its own, and
    // per-kernel durations are what the profile is for.
    m->nchunks = nmembers >= 64 ? 2 : (nmembers >= 24 ? 3 : (nmembers >= 20 ? 2 : 1));
    if (const char *env = getenv("PYSPEEDY_AMD_CHUNKS")) m->nchunks = atoi(env);
    if (m->nchunks < 1) m->nchunks = 1;
    if (m->nchunks > 4) m->nchunks = 4;
    if (m->nchunks > nmembers) m->nchunks = nmembers;
    if (const char *env = getenv("PYSPEEDY_AMD_BLOCK_MEMBERS")) m->block_members = atoi(env) > 0 ? atoi(env) : 0;
    const size_t M = nmembers, S = NSPEC * C, G3 = static_cast<size_t>(8) * NG;
    ModelPtrs &P = m->P;
    spd_physics_args &pa = m->pa;
    int rc = SPD_OK;
#define A(ptr, doubles, name, per)                      \
    if (rc == SPD_OK) rc = dalloc(m, (doubles), &(ptr), name, (per) * sizeof(double))
    // prognostic state (registry names of model_state_def.py:129-153)
    A(P.vor, M * 2 * 8 * S, "vor", 2 * 8 * S);
    A(P.div, M * 2 * 8 * S, "div", 2 * 8 * S);
    A(P.t, M * 2 * 8 * S, "t", 2 * 8 * S);
    A(P.tr, M * 2 * 8 * S, "tr", 2 * 8 * S);
    A(P.ps, M * 2 * S, "ps", 2 * S);
    A(P.phi, M * 8 * S, "phi", 8 * S);
    A(m->phi_buf[1], M * 8 * S, nullptr, 0);
    m->phi_buf[0] = P.phi;
    A(P.phis, M * S, "phis", S);
    A(P.tcorh, M * S, "tcorh", S);
    A(P.qcorh, M * S, "qcorh", S);
    A(P.sv, 4 * M * 8 * S, nullptr, 0);  // ucos | vcos, [M][2][8] each: work space of spectral2grid / grid2spectral
    A(P.vorg, M * G3, nullptr, 0); A(P.divg, M * G3, nullptr, 0); A(P.tg2, M * G3, nullptr, 0);
    A(P.trg2, M * G3, nullptr, 0); A(P.ug2, M * G3, nullptr, 0); A(P.vg2, M * G3, nullptr, 0);
    A(P.px, M * NG, nullptr, 0); A(P.py, M * NG, nullptr, 0);
    A(P.utend, M * G3, nullptr, 0); A(P.vtend, M * G3, nullptr, 0); A(P.ttend, M * G3, nullptr, 0);
    A(P.trtend, M * G3, nullptr, 0); A(P.keg, M * G3, nullptr, 0); A(P.utg, M * G3, nullptr, 0);
    A(P.vtg, M * G3, nullptr, 0); A(P.uqg, M * G3, nullptr, 0); A(P.vqg, M * G3, nullptr, 0);
    A(P.psdtg, M * NG, nullptr, 0);
    A(P.specu, 3 * M * 8 * S, nullptr, 0); A(P.specv, 3 * M * 8 * S, nullptr, 0);
    A(P.spec_tt, M * 8 * S, nullptr, 0); A(P.spec_tr, M * 8 * S, nullptr, 0); A(P.spec_ke, M * 8 * S, nullptr, 0);
    A(P.spec_ps, M * S, nullptr, 0);
    // physics: grid-point inputs (work) ...
    double *tmp = nullptr;
#define PA_IN(field, doubles, name, per)                            \
    A(tmp, doubles, name, per);                                     \
    pa.field = tmp
    PA_IN(ug, M * G3, "u_grid_phys", G3); PA_IN(vg, M * G3, "v_grid_phys", G3); PA_IN(tg, M * G3, "t_grid_phys", G3);
    PA_IN(qg, M * G3, "q_grid_phys", G3); PA_IN(phig, M * G3, "phi_grid_phys", G3); PA_IN(pslg, M * NG, "pslg_phys", NG);
    pa.utend = P.utend; pa.vtend = P.vtend; pa.ttend = P.ttend; pa.qtend = P.trtend;
    // ... surface / forcing fields and outputs under their registry names (model_state_def.py:202-457)
#define PA2(field) PA_IN(field, M * NG, #field, NG)
    PA2(fmask_land); PA2(phis0); PA2(forog); PA2(sst_am); PA2(alb_land); PA2(alb_sea); PA2(snowc); PA2(land_temp);
    PA2(soil_avail_water); PA2(flux_solar_in); PA2(flux_ozone_upper); PA2(flux_ozone_lower); PA2(zenit_correction);
    PA2(stratospheric_correction); PA2(alb_surface);
    PA2(precnv); PA2(precls); PA2(cbmf); PA2(slrd); PA2(slr); PA2(olr); PA2(tsr); PA2(ssrd); PA2(ssr); PA2(qcloud_equiv);
#define PA3(field) PA_IN(field, M * 3 * NG, #field, 3 * NG)
    PA3(slru); PA3(ustr); PA3(vstr); PA3(shf); PA3(evap); PA3(hfluxn);
    PA_IN(rad_st4a, M * 2 * G3, "rad_st4a", 2 * G3);
    PA_IN(rad_flux, M * 4 * NG, "rad_flux", 4 * NG);
    PA_IN(tt_rsw, M * G3, "tt_rsw", G3);
    PA_IN(rad_tau2, M * 4 * G3, "rad_tau2", 4 * G3);
    PA_IN(rad_strat_corr, M * 2 * NG, "rad_strat_corr", 2 * NG);
    // surface / coupler arrays under their registry names (model_state_def.py:250-410)
    SurfacePtrs &SF = m->S;
    const size_t G12 = static_cast<size_t>(12) * NG;
    A(SF.stl12, M * G12, "stl12", G12); A(SF.snowd12, M * G12, "snowd12", G12); A(SF.soilw12, M * G12, "soilw12", G12);
    A(SF.sst12, M * G12, "sst12", G12); A(SF.sea_ice_frac12, M * G12, "sea_ice_frac12", G12);
    A(SF.sst_anom, M * 3 * NG, "sst_anom", 3 * NG);
    A(m->soil_wc_l1, M * G12, "soil_wc_l1", G12); A(m->soil_wc_l2, M * G12, "soil_wc_l2", G12);
    A(m->soil_wc_l3, M * G12, "soil_wc_l3", G12);
#define S2(field) A(SF.field, M * NG, #field, NG)
    S2(stlcl_obs); S2(snowdcl_obs); S2(soilwcl_obs); S2(stl_lm); S2(snow_depth); S2(cdland); S2(rhcapl);
    S2(sstcl_ob); S2(sicecl_ob); S2(ticecl_ob); S2(sstan_ob); S2(sst_om); S2(tice_om); S2(sice_om); S2(sstan_am);
    S2(sice_am); S2(tice_am); S2(ssti_om); S2(cdsea); S2(cdice); S2(rhcaps); S2(rhcapi); S2(hfseacl); S2(fmask_sea);
#undef S2
    A(tmp, M * NG, "alb0", NG); SF.alb0 = tmp;
    A(m->orog, M * NG, "orog", NG); A(m->phi0, M * NG, "phi0", NG); A(m->fmask_orig, M * NG, "fmask_orig", NG);
    A(m->veg_high, M * NG, "veg_high", NG); A(m->veg_low, M * NG, "veg_low", NG);
    A(m->bmask_land, M * NG, "bmask_land", NG); A(m->bmask_sea, M * NG, "bmask_sea", NG);
    A(m->u_grid, M * G3, "u_grid", G3); A(m->v_grid, M * G3, "v_grid", G3); A(m->t_grid, M * G3, "t_grid", G3);
    A(m->q_grid, M * G3, "q_grid", G3); A(m->phi_grid, M * G3, "phi_grid", G3); A(m->ps_grid, M * NG, "ps_grid", NG);
    A(m->corh_t, M * NG, nullptr, 0); A(m->corh_q, M * NG, nullptr, 0); A(m->scratch_spec, 2 * M * S, nullptr, 0);
    SF.land_temp = const_cast<double *>(pa.land_temp); SF.soil_avail_water = const_cast<double *>(pa.soil_avail_water);
    SF.sst_am = const_cast<double *>(pa.sst_am); SF.hfluxn = pa.hfluxn; SF.shf = pa.shf; SF.evap = pa.evap; SF.ssrd = pa.ssrd;
    SF.flux_solar_in = const_cast<double *>(pa.flux_solar_in); SF.flux_ozone_upper = const_cast<double *>(pa.flux_ozone_upper);
    SF.flux_ozone_lower = const_cast<double *>(pa.flux_ozone_lower); SF.zenit_correction = const_cast<double *>(pa.zenit_correction);
    SF.stratospheric_correction = const_cast<double *>(pa.stratospheric_correction);
    SF.snowc = const_cast<double *>(pa.snowc); SF.alb_land = const_cast<double *>(pa.alb_land);
    SF.alb_sea = const_cast<double *>(pa.alb_sea); SF.alb_surface = const_cast<double *>(pa.alb_surface);
    SF.fmask_land = pa.fmask_land; SF.phis0 = pa.phis0;
#undef PA3
#undef PA2
#undef PA_IN
#undef A
    if (rc != SPD_OK) {
        spd_model_destroy(m);
        return rc;
    }
    for (const char *name : {"t_grid_phys", "q_grid_phys", "phi_grid_phys", "pslg_phys", "u_grid_phys", "v_grid_phys",  // inputs
                             "tt_rsw", "rad_tau2", "rad_strat_corr",                                                 // persisted
                             "rad_st4a", "rad_flux", "precnv", "precls", "cbmf", "slrd", "slr", "olr", "slru", "ustr", "vstr"})
        m->reg[name].f32 = true;
    // dynamics tables: the context's (dyn_tables); the time-step dependent ones arrive with spd_model_set_time_step
    if (rc == SPD_OK) {
        const spd_dyn_tables *base = nullptr;
        rc = dyn_tables(h, 0.0, nullptr, &base);
        if (rc == SPD_OK) m->D = base->dev;
    }
    if (rc == SPD_OK) {
        void *p = nullptr;
        rc = arena_alloc(m, sizeof(int) * M, &p);
        m->d_err = static_cast<int *>(p);
        if (rc == SPD_OK) rc = arena_alloc(m, sizeof(double) * M * 24, &p);
        m->d_diag = static_cast<double *>(p);
    }
    if (rc == SPD_OK) rc = build_tables(m);
    if (rc != SPD_OK) {
        spd_model_destroy(m);
        return rc;
    }
    *out = m;
    return SPD_OK;
}

int spd_model_destroy(spd_model_handle m) {
    if (!m) return SPD_OK;
    (void)hipSetDevice(m->ctx->device);
    // The first block (everything spd_model_create allocated) is kept for the next model of this size, up to kIdleBytes per
    // context; what hipFree would have waited for is waited for here.
    constexpr size_t kIdleBytes = static_cast<size_t>(1) << 30;
    bool keep_first = false;
    if (!m->blocks.empty() && hipDeviceSynchronize() == hipSuccess) {
        std::lock_guard<std::mutex> lock(m->ctx->idle_mutex);
        if (m->ctx->idle_bytes + m->blocks[0].size <= kIdleBytes) {
            m->ctx->idle_blocks.push_back({m->blocks[0].base, m->blocks[0].size});
            m->ctx->idle_bytes += m->blocks[0].size;
            keep_first = true;
        }
    }
    for (size_t i = keep_first ? 1 : 0; i < m->blocks.size(); ++i) (void)hipFree(m->blocks[i].base);
    for (int i = 0; i < 4; ++i) {
        if (m->cstream[i]) (void)hipStreamDestroy(m->cstream[i]);
        if (m->cev[i]) (void)hipEventDestroy(m->cev[i]);
    }
    if (m->ev_start) (void)hipEventDestroy(m->ev_start);
    if (m->ev_offset) (void)hipEventDestroy(m->ev_offset);
    if (m->h_err_sync) (void)hipHostFree(m->h_err_sync);
    if (m->h_steps_err) (void)hipHostFree(m->h_steps_err);
    if (m->steps_event) (void)hipEventDestroy(m->steps_event);
    for (int i = 0; i < 2; ++i) {
        if (m->h_err[i]) (void)hipHostFree(m->h_err[i]);
        if (m->err_event[i]) (void)hipEventDestroy(m->err_event[i]);
    }
    for (auto &pr : m->prof_events) {
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    delete m;
    return SPD_OK;
}

int spd_model_members(spd_model_handle m) { return m ? m->M : SPD_E_ARG; }

int spd_model_memory(spd_model_handle m, size_t *bytes_reserved, size_t *bytes_used) {
    if (!m) return m_fail(SPD_E_ARG, "spd_model_memory: null model");
    size_t reserved = 0, used = 0;
    for (const spd_model::Block &b : m->blocks) {
        reserved += b.size;
        used += b.used;
    }
    if (bytes_reserved) *bytes_reserved = reserved;
    if (bytes_used) *bytes_used = used;
    return SPD_OK;
}

long spd_model_var_bytes(spd_model_handle m, const char *name) {
    if (!m || !name) return m_fail(SPD_E_ARG, "spd_model_var_bytes: null argument");
    auto it = m->reg.find(name);
    if (it == m->reg.end()) return m_fail(SPD_E_ARG, std::string("spd_model_var_bytes: unknown variable '") + name + "'");
    return static_cast<long>(it->second.bytes_member);
}

// bytes of one real element of the variable as it is stored on the device NOW: 8, or 4 for the arrays only the column physics
// reads back while the model's physics precision is fp32 (their values then occupy the first half of the allocation)
int spd_model_var_storage(spd_model_handle m, const char *name) {
    if (!m || !name) return m_fail(SPD_E_ARG, "spd_model_var_storage: null argument");
    auto it = m->reg.find(name);
    if (it == m->reg.end()) return m_fail(SPD_E_ARG, std::string("spd_model_var_storage: unknown variable '") + name + "'");
    return (it->second.f32 && m->stored32) ? 4 : 8;
}

static int xfer(spd_model_handle m, const char *name, int member, void *host, size_t bytes, bool to_device) {
    if (!m || !name || !host) return m_fail(SPD_E_ARG, "spd_model_get/set: null argument");
    auto it = m->reg.find(name);
    if (it == m->reg.end()) return m_fail(SPD_E_ARG, std::string("spd_model_get/set: unknown variable '") + name + "'");
    const RegEntry &e = it->second;
    if (bytes != e.bytes_member)
        return m_fail(SPD_E_SIZE, std::string("spd_model_get/set: '") + name + "' needs exactly " + std::to_string(e.bytes_member) + " bytes per member");
    if (member < -1 || member >= m->M) return m_fail(SPD_E_ARG, "spd_model_get/set: member index out of range");
    if (int rc = usable(m, "spd_model_get/set")) return rc;
    if (member == -1 && !to_device) return m_fail(SPD_E_ARG, "spd_model_get: member = -1 (broadcast) is only valid for set");
    M_HIP(hipSetDevice(m->ctx->device));
    if (to_device)  // (a range check that was put off looks at the state as it is NOW)
        if (int rc = settle_deferred_check(m)) return rc;
    // the copies below are blocking copies on the null stream, which does not order against the (non-blocking) streams the
    // model's kernels were issued on: wait for everything in flight on the device first
    M_HIP(hipDeviceSynchronize());
    if (to_device) m->surf_cache_valid = m->phi_ahead = false;
    if (member < 0 && m->M > 1 && !(e.f32 && m->stored32)) {
        // the same values for every member: ONE copy from the host into member 0, handed to the others on the device (a
        // 256-member model set 12 boundary fields with 3072 blocking copies before)
        M_HIP(hipMemcpy(e.ptr, host, bytes, hipMemcpyHostToDevice));
        M_HIP(hipMemsetAsync(m->d_err, 0, sizeof(int) * m->M, nullptr));  // (flags of copy_from_first: nobody differs)
        M_HIP(run_copy_from_first(static_cast<double *>(e.ptr), static_cast<long>(bytes / sizeof(double)), m->M, m->d_err, nullptr));
        M_HIP(hipStreamSynchronize(nullptr));
        return SPD_OK;
    }
    const int first = member < 0 ? 0 : member, last = member < 0 ? m->M - 1 : member;
    if (e.f32 && m->stored32) {  // stored as fp32 (the first half of the allocation): the boundary speaks fp64
        const size_t n = bytes / sizeof(double);
        std::vector<float> narrow(n);
        double *wide = static_cast<double *>(host);
        if (to_device)
            for (size_t k = 0; k < n; ++k) narrow[k] = static_cast<float>(wide[k]);
        for (int i = first; i <= last; ++i) {
            float *dev = static_cast<float *>(e.ptr) + static_cast<size_t>(i) * n;
            if (to_device) {
                M_HIP(hipMemcpy(dev, narrow.data(), n * sizeof(float), hipMemcpyHostToDevice));
            } else {
                M_HIP(hipMemcpy(narrow.data(), dev, n * sizeof(float), hipMemcpyDeviceToHost));
                for (size_t k = 0; k < n; ++k) wide[k] = static_cast<double>(narrow[k]);
            }
        }
        return SPD_OK;
    }
    for (int i = first; i <= last; ++i) {
        char *dev = static_cast<char *>(e.ptr) + static_cast<size_t>(i) * e.bytes_member;
        if (to_device)
            M_HIP(hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice));
        else
            M_HIP(hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost));
    }
    return SPD_OK;
}

int spd_model_set(spd_model_handle m, const char *name, int member, const void *host, size_t bytes) {
    return xfer(m, name, member, const_cast<void *>(host), bytes, true);
}
int spd_model_get(spd_model_handle m, const char *name, int member, void *host, size_t bytes) {
    return xfer(m, name, member, host, bytes, false);
}

// The address is that of the array as it stands NOW and stays valid for the life of the model.  Two things follow for a caller
// that keeps it across steps (include/pyspeedy_amd.h has the contract):
//  * "phi": with the geopotential fold (launches of up to 8 members) the step alternates between two buffers; handing out the
//    address pins the geopotential to the buffer in use -- the fold is switched off for this model from here on -- so that
//    the address keeps showing what the registry's phi is after every later step;
//  * the model caches what it derived from the state (the look-ahead geopotential, the day's interpolated climatologies):
//    they are dropped here, and a caller that writes through a pointer it took EARLIER must call spd_model_invalidate.
void *spd_model_device_ptr(spd_model_handle m, const char *name) {
    if (!m || !name) return nullptr;
    if (usable(m, "spd_model_device_ptr") != SPD_OK) return nullptr;  // (spd_last_error says why)
    auto it = m->reg.find(name);
    if (it == m->reg.end()) return nullptr;
    if (settle_deferred_check(m) != SPD_OK) return nullptr;
    m->surf_cache_valid = m->phi_ahead = false;  // the caller may write through the pointer
    if (it->first == "phi") m->fold_geo = false;
    return it->second.ptr;
}

int spd_model_invalidate(spd_model_handle m) {
    if (!m) return m_fail(SPD_E_ARG, "spd_model_invalidate: null model");
    if (int rc = settle_deferred_check(m)) return rc;
    m->surf_cache_valid = m->phi_ahead = false;
    return SPD_OK;
}

int spd_model_set_co2(spd_model_handle m, double air_absortivity_co2) {
    if (!m) return m_fail(SPD_E_ARG, "spd_model_set_co2: null model");
    m->air_absortivity_co2 = air_absortivity_co2;
    return SPD_OK;
}

double spd_model_co2(spd_model_handle m) { return m ? m->air_absortivity_co2 : 0.0; }

int spd_model_set_time_step(spd_model_handle m, double dt) {
    if (!m) return m_fail(SPD_E_ARG, "spd_model_set_time_step: null model");
    if (!(dt > 0.0)) return m_fail(SPD_E_ARG, "spd_model_set_time_step: the time step must be positive");
    const spd_dyn_tables *set = nullptr;
    if (int rc = dyn_tables(m->ctx, dt, m, &set)) return rc;
    m->dyn = set;
    m->D = set->dev;
    return SPD_OK;
}

// HIP-event bracket around one launch (or a short group of launches) of the step when profiling asks for it
// attach = true (a scope around exactly ONE launch of a step kernel): the events are not recorded here but announced to the
// launch (launch_events.hpp), which attaches them to its dispatch packet: the pair then holds the kernel's own begin / end time
// stamps.  attach = false: recorded around whatever the scope holds (the daily forcing: three launches).
struct ProfScope {
    spd_model *m;
    hipStream_t s;
    hipEvent_t start = nullptr, stop = nullptr;
    bool attach;
    ProfScope(spd_model *m_, int kernel, int fields, hipStream_t s_, bool attach_ = true) : m(m_), s(s_), attach(attach_) {
        if (m->profile == 0 || (m->profile == 1 && kernel != SPD_K_SPEC2GRID)) return;
        if (m->prof_used == m->prof_events.size()) {
            hipEvent_t a = nullptr, b = nullptr;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
            m->prof_events.emplace_back(a, b);
            m->prof_fields.push_back(0);
            m->prof_kernel.push_back(0);
        }
        const size_t i = m->prof_used++;
        m->prof_fields[i] = fields;
        m->prof_kernel[i] = kernel;
        start = m->prof_events[i].first;
        stop = m->prof_events[i].second;
        if (attach) pending_launch_events() = LaunchEvents{start, stop};
        else (void)hipEventRecord(start, s);
    }
    ~ProfScope() {
        if (!stop) return;
        if (!attach) {
            (void)hipEventRecord(stop, s);
        } else if (pending_launch_events().start == start) {  // no launch picked the events up (nothing was launched): record
            pending_launch_events() = LaunchEvents{};        // them, so that reading the pair does not wait for ever
            (void)hipEventRecord(start, s);
            (void)hipEventRecord(stop, s);
        }
    }
};

// one `step(state, j1, j2, dt)` of time_stepping.f90 for the members [first, first + count) on stream s
// Which geopotential buffer the step that is about to be issued uses: the look-ahead of the previous spectral_step_kernel if
// there is one (then no geopotential launch is needed), otherwise the current buffer, to be filled by geopotential_kernel.
// Returns whether the stand-alone kernel has to run.  Called once per step (not per member group).
static bool begin_step_geopotential(spd_model *m) {
    const bool ahead = m->fold_geo && m->phi_ahead;
    if (ahead) m->phi_cur ^= 1;
    m->P.phi = m->phi_buf[m->phi_cur];
    m->P.phi_next = m->fold_geo ? m->phi_buf[m->phi_cur ^ 1] : nullptr;
    m->reg["phi"].ptr = m->P.phi;
    m->phi_ahead = m->fold_geo;  // true once the spectral step of this step has been issued
    return !ahead;
}

// the SPPT generator moves on once per model step (not per member group)
static void sppt_advance(spd_model *m) {
    if (!m->sppt_on) return;
    m->sppt_first = false;
    m->sppt_step += 1;
}

// cpl != nullptr: the coupling that follows the step is part of the last launch
// ride != nullptr: the range check of the PREVIOUS step of these members rides in this step's spectral -> grid launch
static hipError_t step_range(spd_model *m, int j1, int j2, double dt, int compute_shortwave, int first, int count, int diag,
                             bool run_geo, const CouplerArgs *cpl, hipStream_t s, hipEvent_t after_grid2spec = nullptr,
                             const CheckArgs *ride = nullptr) {
    const DeviceTables &T = m->ctx->dev;
    const int M = m->M;
    hipError_t e = hipSuccess;
    if (m->fail_launch_after >= 0 && m->fail_launch_after-- == 0) return hipErrorLaunchFailure;  // (fault injection, tests only)
    // physics.f90:234-236: a new SPPT pattern for every call of the physics, here for the members of this launch (the generator
    // is keyed by the global member id, so a group of members advances exactly its own part of the pattern; the caller moves
    // the generator's step counter once per model step, after all groups have been issued).  The AR(1) update rides in the
    // geopotential launch when there is one (both open the step, neither needs the other)
    SpptArgs sp{};
    if (m->sppt_on)
        sp = sppt_args(m->sppt_spec + static_cast<size_t>(first) * 8 * NSPEC * C, T, count, m->sppt_seed, m->sppt_member_base + first,
                       m->sppt_step, m->sppt_first ? 1 : 0);
    if (run_geo) {
        ProfScope ps(m, SPD_K_GEOPOTENTIAL, count, s);
        e = run_geopotential(m->P, m->D, first, count, 0, m->sppt_on ? &sp : nullptr, s);  // tendencies.f90:229
    } else if (m->sppt_on) {
        ProfScope ps(m, SPD_K_SPPT, 8 * count, s);
        e = run_sppt_update(sp, s);
    }
    spd_physics_args pa = m->pa;
    pa.compute_shortwave = compute_shortwave ? 1 : 0;
    pa.air_absortivity_co2 = m->air_absortivity_co2;
    pa.sppt_pattern = m->sppt_on ? m->sppt_grid : nullptr;  // its 8 transforms per member ride in the spectral -> grid launch below
    if (e == hipSuccess) {                                                                // :109-146, physics.f90:89-101
        const int per = m->inv_per_member + (m->sppt_on ? 8 : 0);
        FieldDesc *table = (m->stored32 ? (m->sppt_on ? m->inv_table_sppt32 : m->inv_table32)
                                         : (m->sppt_on ? m->inv_table_sppt : m->inv_table))[j2 - 1][m->phi_cur];
        ProfScope ps(m, SPD_K_SPEC2GRID, per * count, s);
        if (m->deferred.active && first == 0 && count == M && s == m->deferred.stream) {
            // the range check a host put off at the previous step (spd_model_check_defer): `M` more workgroups of this launch,
            // on the state as that step left it -- nothing of this step has written to it yet
            const int slot = m->deferred.slot;
            const CheckArgs chk{m->P.vor, m->P.div, m->P.t, m->deferred.time_level - 1, m->h_err[slot], m->d_diag, m->slot_ticket[slot]};
            e = run_spec2grid_table_check(T, table, per * count, chk, M, s);
            m->deferred.active = false;
            m->slot_rode[slot] = true;
            ++m->checks_rode;
        } else if (ride) {
            e = run_spec2grid_table_check(T, table + static_cast<size_t>(first) * per, per * count, *ride, count, s);
            ++m->checks_rode;
        } else {
            e = run_spec2grid_table(T, table + static_cast<size_t>(first) * per, per * count, s);
        }
    }
    if (e == hipSuccess) {
        if (m->split_dyn_physics && !m->phys_fp32) {  // whole model, fp64 only; the default is the fused launch (with SPPT: KEEP)
            {
                ProfScope ps(m, SPD_K_DYN_GRID, M, s);
                e = run_dyn_grid(m->P, m->D, M, s);                                       // :151-224
            }
            if (e == hipSuccess) {
                ProfScope ps(m, compute_shortwave ? SPD_K_PHYSICS_SW : SPD_K_PHYSICS, M, s);
                e = run_physics(T, pa, M, m->phys_fp32, s);                               // :231
            }
        } else {
            ProfScope ps(m, compute_shortwave ? SPD_K_COLUMN_SW : SPD_K_COLUMN, count, s);
            e = run_dyn_physics(m->P, m->D, T, pa, first, count, m->phys_fp32, m->stored32 ? 1 : 0, diag, s);  // both in one launch
        }
    }
    if (e == hipSuccess) {                                                                // :238-268
        ProfScope ps(m, SPD_K_GRID2SPEC, 73 * count, s);
        e = run_grid2spec_table(T, m->fwd_table + static_cast<size_t>(first) * 73, 73 * count, s);
    }
    if (after_grid2spec && e == hipSuccess) e = hipEventRecord(after_grid2spec, s);
    const double eps = (j1 == 1) ? 0.0 : static_cast<double>(0.05f);                      // rob, time_stepping.f90:130-134
    if (e == hipSuccess) {
        ProfScope ps(m, SPD_K_SPECTRAL_STEP, count, s);
        const bool early = m->spectral_early < 0 ? count <= 8 : m->spectral_early != 0;
        e = run_spectral_step(m->P, T, m->D, M, first, count, j1 - 1, dt, eps, cpl, early, s);
    }
    return e;
}

// time_stepping.f90 `step(state, j1, j2, dt)`; j1, j2 are the reference's 1-based time-level indices.
int spd_model_step_dynamics(spd_model_handle m, int j1, int j2, double dt, int compute_shortwave, void *stream) {
    if (!m) return m_fail(SPD_E_ARG, "spd_model_step_dynamics: null model");
    if (j1 < 1 || j1 > 2 || j2 < 1 || j2 > 2) return m_fail(SPD_E_ARG, "spd_model_step_dynamics: time levels are 1 or 2");
    if (!m->dyn) return m_fail(SPD_E_ARG, "spd_model_step_dynamics: call spd_model_set_time_step first");
    if (int rc = settle_deferred_check(m)) return rc;
    const bool run_geo = begin_step_geopotential(m);
    const hipError_t e = step_range(m, j1, j2, dt, compute_shortwave, 0, m->M, 1, run_geo, nullptr, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return m_fail(SPD_E_DEVICE, std::string("spd_model_step_dynamics: ") + hipGetErrorString(e));
    sppt_advance(m);
    return SPD_OK;
}

static int next_ticket(spd_model *m) {
    m->check_ticket = m->check_ticket % 0x0fffffff + 1;  // 1 ... 2^28 - 1: never 0 (fresh pinned memory), 4 * ticket + 1 fits an int
    return m->check_ticket;
}

// Wait for the codes of the range-check launch that carries `ticket` by watching the pinned memory it writes them to: the host
// sees each code the moment its store lands (a system-scope release store of 4 * ticket + flag), a few microseconds before a
// completion event behind the kernel would have been signalled and noticed -- for a host that makes one synchronous call per
// model step that wait is on the critical path of every step (PYSPEEDY_AMD_POLL_CODES=0: wait for the event / the stream
// instead).  The event (or the stream) is still asked every few thousand looks, so that a device fault ends the wait.
static int wait_codes(spd_model *m, const int *pinned, int ticket, hipEvent_t ev, hipStream_t s, int32_t *out, const char *who) {
    static const bool poll = !(getenv("PYSPEEDY_AMD_POLL_CODES") && atoi(getenv("PYSPEEDY_AMD_POLL_CODES")) == 0);
    const volatile int *codes = pinned;
    const int M = m->M;
    auto all_there = [&]() {
        for (int i = 0; i < M; ++i)
            if ((codes[i] >> 2) != ticket) return false;
        return true;
    };
    bool there = false;
    if (poll) {
        for (unsigned spin = 1; !(there = all_there()); ++spin) {
            if ((spin & 0xfff) == 0) {
                const hipError_t q = ev ? hipEventQuery(ev) : hipStreamQuery(s);
                if (q == hipSuccess) break;  // the launch is over: its stores are visible now if they ever will be
                if (q != hipErrorNotReady) return m_fail(SPD_E_DEVICE, std::string(who) + ": " + hipGetErrorString(q));
            }
            __builtin_ia32_pause();
        }
    } else {
        if (ev) M_HIP(hipEventSynchronize(ev));
        else M_HIP(hipStreamSynchronize(s));
    }
    if (!there && !all_there()) return m_fail(SPD_E_DEVICE, std::string(who) + ": the range check finished without publishing its codes");
    std::atomic_thread_fence(std::memory_order_acquire);
    for (int i = 0; i < M; ++i) out[i] = (codes[i] & 1) ? -2 : 0;
    return SPD_OK;
}

// diagnostics.f90 check_diagnostics for every member; synchronises the stream and returns the reference's codes.
int spd_model_check(spd_model_handle m, int time_level, int32_t *error_codes_host, double *diag_host, void *stream) {
    if (!m || !error_codes_host) return m_fail(SPD_E_ARG, "spd_model_check: null argument");
    if (int rc = usable(m, "spd_model_check")) return rc;
    if (time_level < 1 || time_level > 2) return m_fail(SPD_E_ARG, "spd_model_check: time level is 1 or 2");
    hipStream_t s = static_cast<hipStream_t>(stream);
    // The kernel writes the codes straight into pinned, coherent host memory (one 4-byte store per member over the fabric): a
    // device-to-host copy behind the kernel is a second operation on the stream -- several microseconds for a host that makes
    // this synchronous call once per model step -- and a copy into the caller's pageable buffer would be staged on top.
    if (!m->h_err_sync) {
        void *p = nullptr;
        M_HIP(hipHostMalloc(&p, sizeof(int) * m->M, hipHostMallocCoherent));
        m->h_err_sync = static_cast<int *>(p);
    }
    const int ticket = next_ticket(m);
    hipError_t e = run_diagnostics(m->P, m->ctx->dev, m->M, time_level - 1, m->h_err_sync, m->d_diag, ticket, s);
    if (e != hipSuccess) return m_fail(SPD_E_DEVICE, std::string("spd_model_check: ") + hipGetErrorString(e));
    if (diag_host) {
        M_HIP(hipMemcpyAsync(diag_host, m->d_diag, sizeof(double) * m->M * 24, hipMemcpyDeviceToHost, s));
        M_HIP(hipStreamSynchronize(s));
    }
    return wait_codes(m, m->h_err_sync, ticket, nullptr, s, error_codes_host, "spd_model_check");
}


// The same range check without stalling the launch pipeline: _begin enqueues the diagnostics of the current state and an
// asynchronous copy of the codes into pinned memory and returns a slot (0 or 1; at most two checks may be in flight);
// _end waits for that slot only and hands out the codes.  A host loop that begins the check of step k, launches step k + 1
// and only then ends the check of step k keeps the GPU busy while still seeing every code (one step late).
// a free slot with its pinned memory and event, a ticket for it; -> slot or a negative error
static int reserve_check_slot(spd_model *m, const char *who) {
    // any free slot (alternating while both are free): the condition for refusing is exactly "two in flight", which is what
    // spd_model_checks_in_flight lets a caller ask BEFORE it enqueues the step this check belongs to
    const int slot = m->slot_busy[m->next_slot] ? 1 - m->next_slot : m->next_slot;
    if (m->slot_busy[slot])
        return m_fail(SPD_E_ARG, std::string(who) + ": two checks are in flight already; end one with spd_model_check_end first");
    if (!m->h_err[slot]) {  // (pinned, coherent: the kernel stores the codes there itself, see spd_model_check)
        void *p = nullptr;
        M_HIP(hipHostMalloc(&p, sizeof(int) * m->M, hipHostMallocCoherent));
        m->h_err[slot] = static_cast<int *>(p);
        M_HIP(hipEventCreateWithFlags(&m->err_event[slot], hipEventDisableTiming));
    }
    m->slot_ticket[slot] = next_ticket(m);
    m->slot_busy[slot] = true;
    m->slot_rode[slot] = false;
    m->next_slot = 1 - slot;
    return slot;
}

static int launch_check(spd_model *m, int slot, int time_level, hipStream_t s, const char *who) {
    hipError_t e = run_diagnostics(m->P, m->ctx->dev, m->M, time_level - 1, m->h_err[slot], m->d_diag, m->slot_ticket[slot], s);
    if (e == hipSuccess) e = hipEventRecord(m->err_event[slot], s);
    if (e != hipSuccess) {
        m->slot_busy[slot] = false;
        return m_fail(SPD_E_DEVICE, std::string(who) + ": " + hipGetErrorString(e));
    }
    m->slot_stream[slot] = s;
    ++m->checks_alone;
    return SPD_OK;
}

// A deferred check that has not found a step to ride in is launched on its own, now, on the stream it was deferred on.  Called
// by everything that is about to read or change what the check looks at in another way than the next step of that stream does.
static int settle_deferred_check(spd_model *m) {
    if (!m->deferred.active) return SPD_OK;
    m->deferred.active = false;
    M_HIP(hipSetDevice(m->ctx->device));
    return launch_check(m, m->deferred.slot, m->deferred.time_level, m->deferred.stream, "deferred range check");
}

int spd_model_check_begin(spd_model_handle m, int time_level, void *stream) {
    if (!m) return m_fail(SPD_E_ARG, "spd_model_check_begin: null model");
    if (int rc = usable(m, "spd_model_check_begin")) return rc;
    if (time_level < 1 || time_level > 2) return m_fail(SPD_E_ARG, "spd_model_check_begin: time level is 1 or 2");
    if (int rc = settle_deferred_check(m)) return rc;
    const int slot = reserve_check_slot(m, "spd_model_check_begin");
    if (slot < 0) return slot;
    if (int rc = launch_check(m, slot, time_level, static_cast<hipStream_t>(stream), "spd_model_check_begin")) return rc;
    return slot;
}

// The same, except that nothing is launched now: the check is carried by the spectral -> grid launch of the NEXT spd_model_step
// call of ONE step on this stream (its first launch, `members` more workgroups: no launch of its own, no time on the step's
// stream), on the state exactly as it is now -- anything else that would read or write the state first (spd_model_set, the
// export transforms, member copies, a multi-step call, spd_model_check_end itself) launches it on its own before it goes on.
// For hosts that collect the check of step k after they have enqueued step k + 1 (spd_parallel_step_begin / _end).
int spd_model_check_defer(spd_model_handle m, int time_level, void *stream) {
    if (!m) return m_fail(SPD_E_ARG, "spd_model_check_defer: null model");
    if (int rc = usable(m, "spd_model_check_defer")) return rc;
    if (time_level < 1 || time_level > 2) return m_fail(SPD_E_ARG, "spd_model_check_defer: time level is 1 or 2");
    if (int rc = settle_deferred_check(m)) return rc;
    const int slot = reserve_check_slot(m, "spd_model_check_defer");
    if (slot < 0) return slot;
    m->deferred.active = true;
    m->deferred.slot = slot;
    m->deferred.time_level = time_level;
    m->deferred.stream = static_cast<hipStream_t>(stream);
    m->slot_stream[slot] = m->deferred.stream;
    return slot;
}

// Launch the check that spd_model_check_defer put off under `slot` (-1: whichever is waiting), now, if it is still waiting for a
// step to ride in -- a check put off under ANOTHER slot is left waiting for its step.  After this call
// spd_model_check_end of that slot only waits and reads: it changes nothing another host thread could be looking at, so a host
// may call it without the lock it serialises its other calls on this model with (csrc/driver.cpp does).
int spd_model_check_settle(spd_model_handle m, int slot) {
    if (!m) return m_fail(SPD_E_ARG, "spd_model_check_settle: null model");
    if (slot >= 0 && !(m->deferred.active && m->deferred.slot == slot)) return SPD_OK;  // (that one is out already, or rides)
    return settle_deferred_check(m);
}

int spd_model_check_counts(spd_model_handle m, int32_t *alone, int32_t *rode) {
    if (!m) return m_fail(SPD_E_ARG, "spd_model_check_counts: null model");
    if (alone) *alone = m->checks_alone;
    if (rode) *rode = m->checks_rode;
    return SPD_OK;
}

int spd_model_checks_in_flight(spd_model_handle m) {
    if (!m) return m_fail(SPD_E_ARG, "spd_model_checks_in_flight: null model");
    return (m->slot_busy[0] ? 1 : 0) + (m->slot_busy[1] ? 1 : 0);
}

int spd_model_check_end(spd_model_handle m, int slot, int32_t *error_codes_host) {
    if (!m || !error_codes_host) return m_fail(SPD_E_ARG, "spd_model_check_end: null argument");
    if (slot < 0 || slot > 1 || !m->slot_busy[slot]) return m_fail(SPD_E_ARG, "spd_model_check_end: no check was begun in this slot");
    if (m->deferred.active && m->deferred.slot == slot)  // no step came to carry it
        if (int rc = settle_deferred_check(m)) return rc;
    m->slot_busy[slot] = false;
    // (a check that rode in a step's launch has no completion event of its own: its stream is asked instead)
    return wait_codes(m, m->h_err[slot], m->slot_ticket[slot], m->slot_rode[slot] ? nullptr : m->err_event[slot], m->slot_stream[slot],
                      error_codes_host, "spd_model_check_end");
}


// ---------------------------------------------------------------------------------------------------------------
// run control: initialize_state (initialization.f90:13-91) and do_single_step (speedy.f90:20-74)
// ---------------------------------------------------------------------------------------------------------------
// set_forcing (forcing.f90:15-102): the host part (zonal forcing of the day, CO2 trend) ...
static ZonalDevice forcing_host(spd_model *m, int imode) {
    if (imode == 0) m->ablco2_ref = m->air_absortivity_co2;  // radset / forog are handled at initialisation
    const ZonalForcing z = zonal_average_fields(m->ctx->host, m->cal.tyear);
    ZonalDevice zd;
    for (int j = 0; j < 48; ++j) {
        zd.v[0][j] = z.flux_solar_in[j]; zd.v[1][j] = z.flux_ozone_upper[j]; zd.v[2][j] = z.flux_ozone_lower[j];
        zd.v[3][j] = z.zenit_correction[j]; zd.v[4][j] = z.stratospheric_correction[j];
    }
    if (m->increase_co2) {
        const double del_co2 = 0.005f;
        m->air_absortivity_co2 = m->ablco2_ref * std::exp(del_co2 * (m->cal.year + m->cal.tyear - 1950));
    }
    return zd;
}

// ... and the device part for the members [first, first + count)
static int forcing_range(spd_model *m, const ZonalDevice &zd, int first, int count, hipStream_t s) {
    const double gamlat = static_cast<double>(6.0f) / (1000.f * static_cast<double>(9.81f));  // setgam, forcing.f90:105-117
    const size_t og = static_cast<size_t>(first) * NG, os = static_cast<size_t>(first) * NSPEC * C;
    ProfScope ps(m, SPD_K_FORCING, count, s, false);
    hipError_t e = run_forcing(m->S, first, count, zd, gamlat, m->corh_t, m->corh_q, s);
    if (e == hipSuccess) e = run_grid2spec(m->ctx->dev, 0, m->corh_t + og, m->P.tcorh + os, 0, count, s);
    if (e == hipSuccess) e = run_grid2spec(m->ctx->dev, 0, m->corh_q + og, m->P.qcorh + os, 0, count, s);
    if (e != hipSuccess) return m_fail(SPD_E_DEVICE, std::string("set_forcing: ") + hipGetErrorString(e));
    return SPD_OK;
}

static int set_forcing(spd_model *m, int imode, hipStream_t s) { return forcing_range(m, forcing_host(m, imode), 0, m->M, s); }

static int couple_range(spd_model *m, int day, int first, int count, int fresh, hipStream_t s) {  // couple_sea_land, coupler.f90:35-48
    const TimeInterp w = time_interp(m->cal);
    if (m->sst_anomaly_flag && (w.a0 < 0 || w.a1 < 0 || w.a0 >= m->anom_planes || w.a1 >= m->anom_planes))
        return m_fail(SPD_E_ARG, "SST anomaly planes do not cover the simulated period (speedy.py:338-372)");
    hipError_t e;
    {
        ProfScope ps(m, SPD_K_COUPLER, count, s);
        e = run_coupler(m->S, first, count, w, day, m->land_coupling_flag, m->sst_anomaly_flag, m->anom_planes, fresh, s);
    }
    if (e != hipSuccess) return m_fail(SPD_E_DEVICE, std::string("couple_sea_land: ") + hipGetErrorString(e));
    return SPD_OK;
}

static int couple(spd_model *m, int day, hipStream_t s) { return couple_range(m, day, 0, m->M, 1, s); }

// initialize_state for every member from the boundary fields previously stored with spd_model_set:
// orog, fmask_orig, alb0, veg_high, veg_low, stl12, snowd12, soil_wc_l1, soil_wc_l2, sst12, sea_ice_frac12 [, sst_anom].
int spd_model_init(spd_model_handle m, int year, int month, int day, int hour, int minute, void *stream) {
    if (!m) return m_fail(SPD_E_ARG, "spd_model_init: null model");
    if (int rc = usable(m, "spd_model_init", true)) return rc;
    if (month < 1 || month > 12 || day < 1 || day > 31) return m_fail(SPD_E_ARG, "spd_model_init: bad start date");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int M = m->M;
    const DeviceTables &T = m->ctx->dev;
    M_HIP(hipSetDevice(m->ctx->device));
    if (int rc = settle_deferred_check(m)) return rc;
    M_HIP(hipDeviceSynchronize());  // (the boundary fields came through the null stream, see xfer; `stream` may be any)
    m->cal.set(year, month, day, hour, minute);
    m->current_step = 0;
    m->surf_cache_valid = m->phi_ahead = false;
    m->step_poison.clear();  // (every array of the state is rebuilt below)
    m->steps_pending = 0;
    m->fail_launch_after = -1;
    // ---- land_model_init / sea_model_init: every member's boundary fields preprocessed where they lie (surface.hip)
    {
        LandSeaPtrs L{};
        L.fmask_orig = m->fmask_orig; L.alb0 = m->S.alb0; L.veg_high = m->veg_high; L.veg_low = m->veg_low;
        L.soil_wc_l1 = m->soil_wc_l1; L.soil_wc_l2 = m->soil_wc_l2;
        L.stl12 = m->S.stl12; L.snowd12 = m->S.snowd12; L.sst12 = m->S.sst12; L.sea_ice_frac12 = m->S.sea_ice_frac12;
        L.sst_anom = m->S.sst_anom;
        L.soilw12 = m->S.soilw12; L.fmask_land = const_cast<double *>(m->pa.fmask_land); L.bmask_land = m->bmask_land;
        L.fmask_sea = m->S.fmask_sea; L.bmask_sea = m->bmask_sea; L.rhcapl = m->S.rhcapl; L.cdland = m->S.cdland;
        L.rhcaps = m->S.rhcaps; L.rhcapi = m->S.rhcapi; L.cdsea = m->S.cdsea; L.cdice = m->S.cdice;
        L.anom_planes = m->anom_planes;
        // (row statistics of the 24 planes: 2304 doubles per member in the spectral scratch, which holds 3968 per member)
        static_assert(2 * 24 * IL <= 2 * NSPEC * C, "scratch_spec holds the row statistics of land_sea_init");
        M_HIP(run_land_sea_init(L, land_sea_consts(m->ctx->host), 0, M, m->scratch_spec, s));
    }
    // ---- initialize_boundaries (boundaries.f90:22-37): phi0 = g * orog, phis0 = spectrally truncated phi0
    double *phis0 = const_cast<double *>(m->pa.phis0);
    hipError_t e = run_scale_orog(m->orog, m->phi0, static_cast<long>(M) * NG, s);
    if (e == hipSuccess) e = run_grid2spec(T, 0, m->phi0, m->scratch_spec, 0, M, s);
    if (e == hipSuccess) e = run_scale(m->scratch_spec, m->scratch_spec, T.trfilt, 1.0, M, s);
    if (e == hipSuccess) e = run_spec2grid(T, 0, m->scratch_spec, phis0, 1, M, s);
    // ---- initialize_from_rest_state (prognostics.f90:29-120)
    RestConsts rc{};
    rc.gam1 = static_cast<double>(6.0f) / (1000.0f * static_cast<double>(9.81f));
    rc.tref = 288.0f;
    rc.ttop = 216.0f;
    rc.sqrt2 = static_cast<double>(std::sqrt(2.0f));
    {
        const double rgam = phc::rgas * rc.gam1, qexp = static_cast<double>(7.5f) / static_cast<double>(2.5f);
        for (int k = 0; k < 8; ++k) {
            rc.fsg_rgam[k] = std::pow(m->ctx->host.fsg[k], rgam);
            rc.fsg_qexp[k] = std::pow(m->ctx->host.fsg[k], qexp);
        }
    }
    if (e == hipSuccess) e = run_grid2spec(T, 0, phis0, m->P.phis, 0, M, s);
    if (e == hipSuccess)
        e = run_rest_surface(phis0, const_cast<double *>(m->pa.forog), m->corh_t, m->corh_q, rc, static_cast<long>(M) * NG, s);
    if (e == hipSuccess) e = run_grid2spec(T, 0, m->corh_t, m->scratch_spec, 0, M, s);                       // ln ps
    if (e == hipSuccess) e = run_grid2spec(T, 0, m->corh_q, m->scratch_spec + static_cast<size_t>(M) * NSPEC * C, 0, M, s);  // q_sfc
    if (e == hipSuccess) {
        M_HIP(hipMemsetAsync(m->P.vor, 0, static_cast<size_t>(M) * 16 * NSPEC * C * sizeof(double), s));
        M_HIP(hipMemsetAsync(m->P.div, 0, static_cast<size_t>(M) * 16 * NSPEC * C * sizeof(double), s));
        M_HIP(hipMemsetAsync(m->P.t, 0, static_cast<size_t>(M) * 16 * NSPEC * C * sizeof(double), s));
        M_HIP(hipMemsetAsync(m->P.tr, 0, static_cast<size_t>(M) * 16 * NSPEC * C * sizeof(double), s));
        M_HIP(hipMemsetAsync(m->P.ps, 0, static_cast<size_t>(M) * 2 * NSPEC * C * sizeof(double), s));
        RestPtrs R{m->P.vor, m->P.div, m->P.t, m->P.tr, m->P.ps, m->P.phis, m->scratch_spec,
                   m->scratch_spec + static_cast<size_t>(M) * NSPEC * C, T.trfilt};
        e = run_rest_state(R, M, rc, s);
    }
    if (e != hipSuccess) return m_fail(SPD_E_DEVICE, std::string("spd_model_init: ") + hipGetErrorString(e));
    // ---- initialize_coupler (day 0), set_forcing(imode = 0), first_step (time_stepping.f90:13-27)
    if (int rc2 = couple(m, 0, s)) return rc2;
    if (int rc2 = set_forcing(m, 0, s)) return rc2;
    const double delt = 86400.0 / 36;
    int rc2 = spd_model_set_time_step(m, 0.5 * delt);
    if (rc2 == SPD_OK) rc2 = spd_model_step_dynamics(m, 1, 1, 0.5 * delt, 1, stream);
    if (rc2 == SPD_OK) rc2 = spd_model_set_time_step(m, delt);
    if (rc2 == SPD_OK) rc2 = spd_model_step_dynamics(m, 1, 2, delt, 1, stream);
    if (rc2 == SPD_OK) rc2 = spd_model_set_time_step(m, 2 * delt);
    if (rc2 != SPD_OK) return rc2;
    m->initialized = true;
    return SPD_OK;
}

// the pinned [steps][members] array the range checks of a checked multi-step call publish their codes in, and the event behind the call
static int ensure_steps_record(spd_model *m, int nsteps) {
    M_HIP(hipSetDevice(m->ctx->device));
    if (m->steps_cap < nsteps) {
        if (m->h_steps_err) M_HIP(hipHostFree(m->h_steps_err));
        m->h_steps_err = nullptr;
        m->steps_cap = 0;
        void *p = nullptr;
        const int cap = nsteps < 64 ? 64 : nsteps;
        M_HIP(hipHostMalloc(&p, sizeof(int) * static_cast<size_t>(cap) * m->M, hipHostMallocCoherent));
        m->h_steps_err = static_cast<int *>(p);
        m->steps_cap = cap;
        std::memset(m->h_steps_err, 0, sizeof(int) * static_cast<size_t>(cap) * m->M);
    }
    if (!m->steps_event) M_HIP(hipEventCreateWithFlags(&m->steps_event, hipEventDisableTiming));
    return SPD_OK;
}

// The streams the member groups of multi-step calls are issued on (and their events), made once per model: at its first multi-step
// call, or before when a host that knows it will make such calls says so (option "prepare_multi_step": a millisecond per stream
// and the measurement that it runs side by side with the others then belong to setting the model up, not to the first stretch of
// its time loop).  Not for every model: an idle stream holds its place among the device's few hardware queues.
static int ensure_group_streams(spd_model *m, int G) {
    M_HIP(hipSetDevice(m->ctx->device));
    if (!m->ev_start) M_HIP(hipEventCreateWithFlags(&m->ev_start, hipEventDisableTiming));
    for (int g = 0; g < G && g < 4; ++g) {
        if (m->cstream[g]) continue;
        // on a hardware queue none of the groups before it is on (stream_apart.hpp: measured, not assumed)
        bool apart = true;
        M_HIP(create_stream_apart(&m->cstream[g], m->cstream, g, hipStreamNonBlocking, &apart));
        m->groups_apart = m->groups_apart && apart;
        M_HIP(hipEventCreateWithFlags(&m->cev[g], hipEventDisableTiming));
    }
    return SPD_OK;
}

// do_single_step (speedy.f90:20-74) `nsteps` times for all members.  Nothing synchronises; the range check of
// diagnostics.f90 is available separately through spd_model_check (the reference runs it after every step).
// record: the range check of every step is left in m->h_steps_err[step][member] (spd_model_step_checked_begin) -- the check of step
// k rides in the spectral -> grid launch of step k + 1 of the same members, the last one is a launch of its own behind the call.
static int step_impl(spd_model *m, int nsteps, void *stream, bool record, const char *who) {
    if (!m) return m_fail(SPD_E_ARG, std::string(who) + ": null model");
    if (int rc = usable(m, who)) return rc;
    if (!m->initialized) return m_fail(SPD_E_ARG, std::string(who) + ": model state not initialized (error code -1 of the reference)");
    if (!m->dyn) return m_fail(SPD_E_ARG, std::string(who) + ": call spd_model_set_time_step first");
    if (m->steps_pending) return m_fail(SPD_E_ARG, std::string(who) + ": a checked multi-step call is in flight; end it with spd_model_step_checked_end first");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const double delt = 86400.0 / 36;
    // Members never exchange data, so the step is issued group by group on separate streams: every group runs the same
    // six launches on its own members, and the groups' kernels overlap on the GPU.  The caller's stream orders the whole call:
    // the group streams start after everything already enqueued on it, and it continues after all of them.
    // (a call of ONE step forks and joins the group streams around that step -- two dependent cross-stream hand-overs per
    // step cost more than the overlap gains: measured +12 % at 64 members through spd_parallel_step -- so it is issued serially)
    const int G = (m->split_dyn_physics || m->profile > 0 || nsteps == 1) ? 1 : m->nchunks;
    // a range check that was put off rides in the first spectral -> grid launch of this call -- when that is ONE launch over all
    // members on the stream the check was put off on; otherwise it goes out on its own first
    if (m->deferred.active && (G > 1 || s != m->deferred.stream || record))
        if (int rc = settle_deferred_check(m)) return rc;
    hipStream_t gs[4] = {s, nullptr, nullptr, nullptr};
    if (G > 1) {
        if (int rc = ensure_group_streams(m, G)) return rc;
        M_HIP(hipEventRecord(m->ev_start, s));
        for (int g = 0; g < G; ++g) {
            gs[g] = m->cstream[g];
            M_HIP(hipStreamWaitEvent(gs[g], m->ev_start, 0));
        }
    }
    // whatever way this function is left, the caller's stream continues behind everything the group streams were given
    struct Join {
        hipStream_t s, *gs;
        hipEvent_t *ev;
        int G;
        ~Join() {
            if (G > 1)
                for (int g = 0; g < G; ++g)
                    if (hipEventRecord(ev[g], gs[g]) == hipSuccess) (void)hipStreamWaitEvent(s, ev[g], 0);
        }
    } join{s, gs, m->cev, G};
    // Two groups that start together stay together: they have the same work, so both run their transform launches at the same
    // time and their column launches at the same time, and only the tails overlap.  The second group therefore starts when the
    // first has issued three quarters of its first step (behind its grid -> spectral launch): from then on one group's
    // latency-bound transforms run beside the other's streaming column / spectral kernels.  Measured at 64 members: 0.241 ms
    // per step every time, against 0.243 ... 0.250 when left to chance (profiles/r03_member_groups.txt); 96 members -2.8 %;
    // nothing at 32 / 48 members or with 3 groups.  It costs a call the time its first and last three quarters of a step run
    // alone; applied to calls of at least 36 steps (PYSPEEDY_AMD_GROUP_OFFSET=n: from n steps, 0: never -- in 20-step calls it
    // measured 0 ... +1.5 %; in the 36-step calls of a time loop with daily hooks -1.8 %: 9.58 -> 9.41 ms per simulated day,
    // round 6; 72 until then).
    static const int offset_from = getenv("PYSPEEDY_AMD_GROUP_OFFSET") ? atoi(getenv("PYSPEEDY_AMD_GROUP_OFFSET")) : 36;
    const bool offset = offset_from > 0 && G == 2 && nsteps >= offset_from;
    if (offset && !m->ev_offset) M_HIP(hipEventCreateWithFlags(&m->ev_offset, hipEventDisableTiming));
    // rounds (see block_members): the members of a round go through all steps of the call before the next round starts
    // (also with ONE group when the caller says so by setting member_groups to 1 on a large model -- the outer boundary does for
    // the two device models it keeps a large ensemble in, which are each other's groups: csrc/driver.cpp -- but never while profiling
    // or in the split-launch mode, whose single group is the whole model by definition)
    int rounds = 1;
    const bool may_round = G > 1 || (m->nchunks == 1 && m->profile == 0 && !m->split_dyn_physics);
    if (may_round && nsteps > 1 && m->block_members > 0 && m->M >= 4 * m->block_members)
        rounds = (m->M + G * m->block_members - 1) / (G * m->block_members);
    struct HostState {  // what a step changes on the host side of the model
        Calendar cal;
        int current_step, phi_cur;
        bool surf_cache_valid, phi_ahead, sppt_first;
        long long sppt_step;
        double co2;
    };
    const HostState start{m->cal, m->current_step, m->phi_cur, m->surf_cache_valid, m->phi_ahead, m->sppt_first, m->sppt_step,
                          m->air_absortivity_co2};
    if (rounds > 1 && m->sst_anomaly_flag) {  // (what can refuse a step is asked for ALL steps before the first round goes out)
        Calendar ahead = m->cal;
        for (int it = 0; it < nsteps; ++it) {
            ahead.advance();
            const TimeInterp w = time_interp(ahead);
            if (w.a0 < 0 || w.a1 < 0 || w.a0 >= m->anom_planes || w.a1 >= m->anom_planes)
                return m_fail(SPD_E_ARG, "SST anomaly planes do not cover the simulated period (speedy.py:338-372)");
        }
    }
    int rc = SPD_OK;
    auto note_accepted = [&](int row) {  // (record: what the host side of the model looks like after `row` steps of the call)
        if (!record) return;
        int32_t *a = m->steps_accepted.data() + 7 * static_cast<size_t>(row);
        a[0] = m->current_step; a[1] = m->cal.year; a[2] = m->cal.month; a[3] = m->cal.day; a[4] = m->cal.hour; a[5] = m->cal.minute;
        a[6] = m->cal.month_idx;
    };
    if (record) m->steps_accepted.assign(7 * static_cast<size_t>(nsteps + 1), 0);
    note_accepted(0);
    bool launched = false, device_failed = false;  // a launch of this call went out / a device call of it failed
    const int tl_check = 1;  // the check looks at time level 2 (do_single_step checks the state the step has just produced)
    for (int round = 0, round_first = 0; round < rounds && rc == SPD_OK; ++round) {
        const int round_count = m->M / rounds + (round < m->M % rounds ? 1 : 0);
        if (round > 0) {  // the same steps again, for the next members
            m->cal = start.cal;
            m->current_step = start.current_step;
            m->phi_cur = start.phi_cur;
            m->surf_cache_valid = start.surf_cache_valid;
            m->phi_ahead = start.phi_ahead;
            m->sppt_first = start.sppt_first;
            m->sppt_step = start.sppt_step;
            m->air_absortivity_co2 = start.co2;
        }
        const int base = round_count / G, extra = round_count % G;
        for (int it = 0; it < nsteps && rc == SPD_OK; ++it) {
            const bool new_day = m->current_step % 36 == 0;
            ZonalDevice zd{};
            if (new_day) zd = forcing_host(m, 1);
            const int sw = (m->current_step % 3 == 0) ? 1 : 0;
            const int diag = (m->diag_every_step || it == nsteps - 1) ? 1 : 0;
            // The land / sea-ice coupling that follows the step (speedy.f90:72) happens at the date AFTER the step and for the
            // incremented step counter.  The interpolation weights of the climatologies change at midnight only: the first
            // coupling of a day (or of a state the host touched) interpolates, the others re-use what it stored (surface.hip).
            Calendar next = m->cal;
            next.advance();
            const TimeInterp w = time_interp(next);
            const int fresh = (!m->surf_cache_valid || (next.hour == 0 && next.minute == 0)) ? 1 : 0;
            if (m->sst_anomaly_flag && (w.a0 < 0 || w.a1 < 0 || w.a0 >= m->anom_planes || w.a1 >= m->anom_planes)) {
                rc = m_fail(SPD_E_ARG, "SST anomaly planes do not cover the simulated period (speedy.py:338-372)");
                break;
            }
            const bool run_geo = begin_step_geopotential(m);
            const bool first_of_call = it == 0 && round == 0;
            for (int g = 0, first = round_first; g < G && rc == SPD_OK; ++g) {
                const int count = base + (g < extra ? 1 : 0);
                if (count == 0) continue;
                launched = true;
                if (new_day) {
                    rc = forcing_range(m, zd, first, count, gs[g]);
                    device_failed = device_failed || rc != SPD_OK;
                }
                CouplerArgs cpl{m->S, w, first, count, 1 + (m->current_step + 1) / 36, m->land_coupling_flag, m->sst_anomaly_flag,
                                m->anom_planes, fresh};
                const bool ride = m->coupler_in_spectral;
                if (rc == SPD_OK) {
                    hipError_t e = hipSuccess;
                    if (offset && first_of_call && g == 1) e = hipStreamWaitEvent(gs[1], m->ev_offset, 0);
                    // (the check of the step before this one, for these members: written to that step's row)
                    const CheckArgs chk{m->P.vor, m->P.div, m->P.t, tl_check, record && it > 0 ? m->h_steps_err + static_cast<size_t>(it - 1) * m->M : nullptr,
                                        nullptr, m->steps_ticket, first};
                    if (e == hipSuccess)
                        e = step_range(m, 2, 2, 2 * delt, sw, first, count, diag, run_geo, ride ? &cpl : nullptr, gs[g],
                                       (offset && first_of_call && g == 0) ? m->ev_offset : nullptr, record && it > 0 ? &chk : nullptr);
                    if (e != hipSuccess) {
                        (void)hipGetLastError();
                        rc = m_fail(SPD_E_DEVICE, std::string(who) + ": " + hipGetErrorString(e));
                        device_failed = true;
                    }
                }
                if (rc == SPD_OK && !ride) {
                    ProfScope ps(m, SPD_K_COUPLER, count, gs[g]);
                    const hipError_t e = run_coupler(m->S, first, count, w, 1 + (m->current_step + 1) / 36, m->land_coupling_flag,
                                                     m->sst_anomaly_flag, m->anom_planes, fresh, gs[g]);
                    if (e != hipSuccess) {
                        rc = m_fail(SPD_E_DEVICE, std::string("couple_sea_land: ") + hipGetErrorString(e));
                        device_failed = true;
                    }
                }
                if (rc == SPD_OK && record && it == nsteps - 1) {  // the last step's check: nothing comes behind it to carry it
                    const hipError_t e = run_diagnostics_range(m->P, m->ctx->dev, first, count, tl_check,
                                                               m->h_steps_err + static_cast<size_t>(it) * m->M, nullptr, m->steps_ticket, gs[g]);
                    if (e != hipSuccess) {
                        rc = m_fail(SPD_E_DEVICE, std::string(who) + ": " + hipGetErrorString(e));
                        device_failed = true;
                    }
                    ++m->checks_alone;
                }
                first += count;
            }
            if (rc != SPD_OK) break;
            sppt_advance(m);
            m->current_step += 1;
            m->cal = next;
            m->surf_cache_valid = true;
            if (round == 0) note_accepted(it + 1);
        }
        round_first += round_count;
    }
    // A device error after the first launch of the call went out: a member group may have taken a step (or a part of one: the
    // launches of a step write its work arrays one after the other) that the others -- and the host side of the model, whose step
    // counter and date only move with complete steps -- have not.  Whatever the plan (one group, several, rounds), the model is
    // unusable until spd_model_init has rebuilt its state; the message of the failure itself stays the call's error text.
    if (rc != SPD_OK && launched && device_failed) {
        const std::string text = spd_last_error();
        m->initialized = false;
        m->step_poison = "a device error interrupted a model step after some of its launches had gone out (" + text + ")";
        (void)m_fail(rc, text);
    }
    return rc;
}

int spd_model_step(spd_model_handle m, int nsteps, void *stream) { return step_impl(m, nsteps, stream, false, "spd_model_step"); }

// The same call with the range check of EVERY step recorded on the device (diagnostics.f90:16-76 after each do_single_step, as the
// reference's time loop sees it: pyspeedy/speedy.py:396-405), for hosts that know they will not look at the state before `nsteps`
// steps have passed (no callback due): one call instead of nsteps, the member groups and rounds of the multi-step plan, and still
// every step's code.  _begin enqueues everything and returns; _end waits and reports, per member, the first step of the call whose
// check failed (0-based; -1: none).  The device does not stop at a failed check -- the steps behind it run on a state the model
// does not accept, as the second step of spd_parallel_step_begin does; after a failure the only defined continuation is
// spd_model_init.  One such call may be in flight per model; nsteps <= 4096.
int spd_model_step_checked_begin(spd_model_handle m, int nsteps, void *stream) {
    const char *who = "spd_model_step_checked_begin";
    if (!m) return m_fail(SPD_E_ARG, std::string(who) + ": null model");
    if (nsteps < 1 || nsteps > 4096) return m_fail(SPD_E_ARG, std::string(who) + ": 1 ... 4096 steps per call");
    if (m->steps_pending) return m_fail(SPD_E_ARG, std::string(who) + ": a checked multi-step call is in flight already");
    if (int rc = ensure_steps_record(m, nsteps)) return rc;
    m->steps_ticket = next_ticket(m);
    if (int rc = step_impl(m, nsteps, stream, true, who)) return rc;
    // (step_impl has joined the group streams into the caller's stream: the event is behind every launch of the call)
    M_HIP(hipEventRecord(m->steps_event, static_cast<hipStream_t>(stream)));
    m->steps_pending = nsteps;
    return SPD_OK;
}

int spd_model_step_checked_end(spd_model_handle m, int32_t *first_failed_step, int32_t *accepted) {
    const char *who = "spd_model_step_checked_end";
    if (!m || !first_failed_step) return m_fail(SPD_E_ARG, std::string(who) + ": null argument");
    if (!m->steps_pending) return m_fail(SPD_E_ARG, std::string(who) + ": no checked multi-step call is in flight");
    const int K = m->steps_pending, M = m->M;
    m->steps_pending = 0;
    M_HIP(hipSetDevice(m->ctx->device));
    M_HIP(hipEventSynchronize(m->steps_event));
    std::atomic_thread_fence(std::memory_order_acquire);
    const volatile int *codes = m->h_steps_err;
    for (int i = 0; i < M; ++i) first_failed_step[i] = -1;
    for (int k = K - 1; k >= 0; --k)
        for (int i = 0; i < M; ++i) {
            const int c = codes[static_cast<size_t>(k) * M + i];
            if ((c >> 2) != m->steps_ticket) return m_fail(SPD_E_DEVICE, std::string(who) + ": a range check finished without publishing its code");
            if (c & 1) first_failed_step[i] = k;
        }
    if (accepted)  // a member's last accepted step: the one before its first failure, or the last of the call
        for (int i = 0; i < M; ++i)
            std::memcpy(accepted + 7 * static_cast<size_t>(i),
                        m->steps_accepted.data() + 7 * static_cast<size_t>(first_failed_step[i] < 0 ? K : first_failed_step[i]), 7 * sizeof(int32_t));
    return SPD_OK;
}



int spd_model_current_step(spd_model_handle m) { return m ? m->current_step : SPD_E_ARG; }

// Profiling with HIP events on the launch stream.  Level 1: the spec2grid launch of every step (the roofline kernel of
// bench.py); level 2: every kernel of the step.  The events are attached to the kernels' dispatch packets (launch_events.hpp),
// and while the level is not 0 the step is issued as one member group (spd_model_step).
int spd_model_profile(spd_model_handle m, int level) {
    if (!m) return m_fail(SPD_E_ARG, "spd_model_profile: null model");
    if (level < 0 || level > 2) return m_fail(SPD_E_ARG, "spd_model_profile: level is 0, 1 or 2");
    m->profile = level;
    m->prof_used = 0;
    return SPD_OK;
}

int spd_model_profile_read(spd_model_handle m, double *mean_ms, int *launches, int *fields_per_launch) {
    if (!m || !mean_ms || !launches || !fields_per_launch) return m_fail(SPD_E_ARG, "spd_model_profile_read: null argument");
    double sum = 0.0;
    int n = 0, fields = m->inv_per_member * m->M;
    for (size_t i = 0; i < m->prof_used; ++i) {
        if (m->prof_kernel[i] != SPD_K_SPEC2GRID) continue;
        float ms = 0.f;
        M_HIP(hipEventSynchronize(m->prof_events[i].second));
        M_HIP(hipEventElapsedTime(&ms, m->prof_events[i].first, m->prof_events[i].second));
        sum += ms;
        if (n == 0) fields = m->prof_fields[i];
        ++n;
    }
    *launches = n;
    *mean_ms = n ? sum / n : 0.0;
    *fields_per_launch = fields;
    return SPD_OK;
}

// per kernel id (SPD_K_*): mean and minimum bracket time in ms, number of brackets, units (fields or members) per bracket
int spd_model_profile_read_kernels(spd_model_handle m, double *mean_ms, double *min_ms, int *launches, int *units) {
    if (!m || !mean_ms || !min_ms || !launches || !units) return m_fail(SPD_E_ARG, "spd_model_profile_read_kernels: null argument");
    for (int k = 0; k < SPD_K_COUNT; ++k) {
        mean_ms[k] = min_ms[k] = 0.0;
        launches[k] = units[k] = 0;
    }
    for (size_t i = 0; i < m->prof_used; ++i) {
        const int k = m->prof_kernel[i];
        if (k < 0 || k >= SPD_K_COUNT) continue;
        float ms = 0.f;
        M_HIP(hipEventSynchronize(m->prof_events[i].second));
        M_HIP(hipEventElapsedTime(&ms, m->prof_events[i].first, m->prof_events[i].second));
        mean_ms[k] += ms;
        if (launches[k] == 0 || ms < min_ms[k]) min_ms[k] = ms;
        units[k] = m->prof_fields[i];
        ++launches[k];
    }
    for (int k = 0; k < SPD_K_COUNT; ++k)
        if (launches[k]) mean_ms[k] /= launches[k];
    return SPD_OK;
}

int spd_model_mark_initialized(spd_model_handle m, int current_step, int year, int month, int day, int hour, int minute) {
    if (!m) return m_fail(SPD_E_ARG, "spd_model_mark_initialized: null model");
    if (int rc = usable(m, "spd_model_mark_initialized")) return rc;
    m->cal.set(year, month, day, hour, minute);
    m->current_step = current_step;
    m->surf_cache_valid = false;
    m->ablco2_ref = m->air_absortivity_co2;  // set_forcing(imode = 0), forcing.f90:40
    m->initialized = true;
    return SPD_OK;
}

int spd_model_get_control(spd_model_handle m, spd_model_control *out) {
    if (!m || !out) return m_fail(SPD_E_ARG, "spd_model_get_control: null argument");
    spd_model_control c{};
    c.current_step = m->current_step;
    c.year = m->cal.year; c.month = m->cal.month; c.day = m->cal.day; c.hour = m->cal.hour; c.minute = m->cal.minute;
    c.month_idx = m->cal.month_idx;
    c.land_coupling_flag = m->land_coupling_flag;
    c.sst_anomaly_coupling_flag = m->sst_anomaly_flag;
    c.increase_co2 = m->increase_co2;
    c.sppt_on = m->sppt_on ? 1 : 0;
    c.sppt_first = m->sppt_first ? 1 : 0;
    c.physics_fp32 = m->phys_fp32;
    c.sppt_step = m->sppt_step;
    c.sppt_first_member_id = m->sppt_member_base;
    c.sppt_seed = m->sppt_seed;
    c.air_absortivity_co2 = m->air_absortivity_co2;
    c.ablco2_ref = m->ablco2_ref;
    *out = c;
    return SPD_OK;
}

int spd_model_set_control(spd_model_handle m, const spd_model_control *in) {
    if (!m || !in) return m_fail(SPD_E_ARG, "spd_model_set_control: null argument");
    if (int rc = usable(m, "spd_model_set_control")) return rc;
    if (in->month < 1 || in->month > 12 || in->day < 1 || in->day > 31 || in->month_idx < 1 || in->current_step < 0)
        return m_fail(SPD_E_ARG, "spd_model_set_control: bad date, month index or step counter");
    if (in->sppt_on && !m->sppt_spec)
        return m_fail(SPD_E_ARG, "spd_model_set_control: SPPT is on in the control block: call spd_model_set_sppt and load sppt_spec first");
    // (the precision of the column physics carries a storage format with it: spd_model_set_physics_precision converts)
    if (int rc = spd_model_set_physics_precision(m, in->physics_fp32)) return rc;
    m->cal.set(in->year, in->month, in->day, in->hour, in->minute);
    m->cal.month_idx = in->month_idx;
    m->surf_cache_valid = false;
    m->current_step = in->current_step;
    m->land_coupling_flag = in->land_coupling_flag ? 1 : 0;
    m->sst_anomaly_flag = in->sst_anomaly_coupling_flag ? 1 : 0;
    m->increase_co2 = in->increase_co2 ? 1 : 0;
    m->sppt_on = in->sppt_on != 0;
    m->sppt_first = in->sppt_first != 0;
    m->sppt_step = in->sppt_step;
    m->sppt_member_base = in->sppt_first_member_id;
    m->sppt_seed = in->sppt_seed;
    m->air_absortivity_co2 = in->air_absortivity_co2;
    m->ablco2_ref = in->ablco2_ref;
    m->initialized = true;
    return SPD_OK;
}

int spd_model_get_date(spd_model_handle m, int *ymdhm) {
    if (!m || !ymdhm) return m_fail(SPD_E_ARG, "spd_model_get_date: null argument");
    ymdhm[0] = m->cal.year; ymdhm[1] = m->cal.month; ymdhm[2] = m->cal.day; ymdhm[3] = m->cal.hour; ymdhm[4] = m->cal.minute;
    return SPD_OK;
}

int spd_model_group_streams(spd_model_handle m, int32_t *created, int32_t *apart) {
    if (!m) return m_fail(SPD_E_ARG, "spd_model_group_streams: null model");
    int n = 0;
    for (int g = 0; g < 4; ++g) n += m->cstream[g] ? 1 : 0;
    if (created) *created = n;
    if (apart) *apart = m->groups_apart ? 1 : 0;
    return SPD_OK;
}

int spd_model_get_config(spd_model_handle m, int32_t *cfg) {
    if (!m || !cfg) return m_fail(SPD_E_ARG, "spd_model_get_config: null argument");
    cfg[0] = m->inv_per_member;
    cfg[1] = m->diag_every_step ? 1 : 0;
    cfg[2] = m->nchunks;
    cfg[3] = m->split_dyn_physics ? 1 : 0;
    cfg[4] = m->fold_geo ? 1 : 0;
    cfg[5] = m->coupler_in_spectral ? 1 : 0;
    cfg[6] = m->phys_fp32;
    cfg[7] = m->stored32 ? 1 : 0;
    return SPD_OK;
}

int spd_model_get_option(spd_model_handle m, const char *name, int32_t *value) {
    if (!m || !name || !value) return m_fail(SPD_E_ARG, "spd_model_get_option: null argument");
    const std::string key(name);
    if (key == "diag_every_step") *value = m->diag_every_step ? 1 : 0;
    else if (key == "coupler_in_spectral") *value = m->coupler_in_spectral ? 1 : 0;
    else if (key == "split_dyn") *value = m->split_dyn_physics ? 1 : 0;
    else if (key == "spectral_early") *value = m->spectral_early;
    else if (key == "member_groups") *value = m->nchunks;
    else if (key == "block_members") *value = m->block_members;
    else if (key == "physics_storage32") *value = m->phys_store32 ? 1 : 0;
    else return m_fail(SPD_E_ARG, "spd_model_get_option: unknown option: " + key);
    return SPD_OK;
}

int spd_model_set_option(spd_model_handle m, const char *name, int32_t value) {
    if (!m || !name) return m_fail(SPD_E_ARG, "spd_model_set_option: null argument");
    const std::string key(name);
    const bool flag = value == 0 || value == 1;
    if (key == "diag_every_step" && flag) m->diag_every_step = value != 0;
    else if (key == "coupler_in_spectral" && flag) m->coupler_in_spectral = value != 0;
    else if (key == "split_dyn" && flag) m->split_dyn_physics = value != 0;
    else if (key == "spectral_early" && value >= -1 && value <= 1) m->spectral_early = value;
    else if (key == "member_groups" && value >= 1 && value <= 4) m->nchunks = value < m->M ? value : m->M;
    else if (key == "block_members" && value >= 0) m->block_members = value;
    else if (key == "fail_launch_after" && value >= -1) m->fail_launch_after = value;  // (fault injection: tests)
    else if (key == "prepare_multi_step" && value == 1) {  // what multi-step calls need once per model, now instead of at the first one
        if (int rc = ensure_steps_record(m, 360)) return rc;  // (a stretch of the facade's time loops is at most 360 steps long)
        if (m->nchunks > 1 && !m->split_dyn_physics) return ensure_group_streams(m, m->nchunks);
    }
    else if (key == "physics_storage32" && flag) {
        m->phys_store32 = value != 0;
        return apply_storage(m, m->phys_fp32 && m->phys_store32);
    }
    else return m_fail(SPD_E_ARG, "spd_model_set_option: unknown option or value out of range: " + key);
    return SPD_OK;
}

// bring the storage of the RegEntry::f32 arrays in line with what the model's settings ask for
static int apply_storage(spd_model *m, bool want32) {
    if (want32 == m->stored32) return SPD_OK;
    if (int rc = usable(m, "spd_model_set_physics_precision")) return rc;
    const int fp32 = want32 ? 1 : 0;
    // the arrays only the column physics reads back change their storage with its arithmetic: converted here, once (values
    // that came out of the fp32 physics are fp32 numbers already; what the fp64 physics left is rounded as that kernel's loads
    // would have rounded it)
    M_HIP(hipSetDevice(m->ctx->device));
    M_HIP(hipDeviceSynchronize());
    // everything that can fail WITHOUT having touched an array comes first: the descriptor tables of the fp32 layout, the scratch
    if (want32)
        if (int rc = ensure_tables32(m)) return rc;
    size_t largest = 0;
    for (const auto &kv : m->reg)
        if (kv.second.f32) largest = std::max(largest, kv.second.bytes_member * static_cast<size_t>(m->M));
    void *scratch = nullptr;
    M_HIP(hipMalloc(&scratch, largest));
    hipError_t e = hipSuccess;
    int converted = 0;
    for (const auto &kv : m->reg) {
        if (!kv.second.f32 || e != hipSuccess) continue;
        const long n = static_cast<long>(kv.second.bytes_member / sizeof(double)) * m->M;
        e = run_change_storage(static_cast<double *>(kv.second.ptr), n, fp32 != 0, scratch, nullptr);
        ++converted;
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    (void)hipFree(scratch);
    if (e != hipSuccess) {
        // some arrays are in the new format, some in the old, and nothing records which: the state cannot be read any more
        if (converted > 0) {
            m->initialized = false;
            m->poisoned = std::string("a device error (") + hipGetErrorString(e) + ") interrupted the change of its fp32 / fp64 storage; "
                          "create and initialise a new model";
        }
        return m_fail(SPD_E_DEVICE, std::string("spd_model_set_physics_precision: ") + hipGetErrorString(e));
    }
    m->stored32 = want32;  // (only now: every array is in the new format)
    return SPD_OK;
}

int spd_model_set_physics_precision(spd_model_handle m, int fp32) {
    if (!m) return m_fail(SPD_E_ARG, "spd_model_set_physics_precision: null model");
    m->phys_fp32 = fp32 ? 1 : 0;
    return apply_storage(m, m->phys_fp32 && m->phys_store32);
}

int spd_model_set_flags(spd_model_handle m, int land_coupling_flag, int sst_anomaly_coupling_flag, int increase_co2) {
    if (!m) return m_fail(SPD_E_ARG, "spd_model_set_flags: null model");
    m->land_coupling_flag = land_coupling_flag ? 1 : 0;
    m->sst_anomaly_flag = sst_anomaly_coupling_flag ? 1 : 0;
    m->increase_co2 = increase_co2 ? 1 : 0;
    m->surf_cache_valid = false;
    return SPD_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// grid-space views of the prognostic state (prognostics.f90:125-219) for members [first, first + count)
// ---------------------------------------------------------------------------------------------------------------
static int member_range(spd_model_handle m, int first, int count, const char *who) {
    if (!m) return m_fail(SPD_E_ARG, std::string(who) + ": null model");
    if (first < 0 || count < 0 || first + count > m->M) return m_fail(SPD_E_ARG, std::string(who) + ": member range out of bounds");
    return usable(m, who);
}

// One grid-space registry variable ((ix, il) or (ix, il, kx) per member) of the members [first, first + count) as a NetCDF-3
// file carries it: float32, BIG-endian, levels bottom-up -- into `dst_device` (count * levels * 4608 * 4 bytes), on `stream`.
// For hosts that write files: what crosses PCIe afterwards is the file's payload itself.
int spd_model_export_pack(spd_model_handle m, const char *name, int first, int count, void *dst_device, size_t dst_bytes, void *stream) {
    if (!name || !dst_device) return m_fail(SPD_E_ARG, "spd_model_export_pack: null argument");
    if (int rc = member_range(m, first, count, "spd_model_export_pack")) return rc;
    if (int rc = usable(m, "spd_model_export_pack")) return rc;
    auto it = m->reg.find(name);
    if (it == m->reg.end()) return m_fail(SPD_E_ARG, std::string("spd_model_export_pack: unknown variable '") + name + "'");
    const RegEntry &e = it->second;
    const size_t plane = static_cast<size_t>(NG) * sizeof(double);
    const int levels = static_cast<int>(e.bytes_member / plane);
    if (e.bytes_member % plane != 0 || (levels != 1 && levels != KX))
        return m_fail(SPD_E_ARG, std::string("spd_model_export_pack: '") + name + "' is not a grid-space (ix, il[, kx]) variable");
    const size_t need = static_cast<size_t>(count) * levels * NG * sizeof(float);
    if (dst_bytes < need) return m_fail(SPD_E_SIZE, "spd_model_export_pack: destination too small");
    if (count == 0) return SPD_OK;
    M_HIP(hipSetDevice(m->ctx->device));
    if (int rc = settle_deferred_check(m)) return rc;
    const bool narrow = e.f32 && m->stored32;  // (stored as fp32 in the first half of the allocation)
    const char *src = static_cast<const char *>(e.ptr) + static_cast<size_t>(first) * (narrow ? e.bytes_member / 2 : e.bytes_member);
    const hipError_t err = run_export_pack(src, narrow, dst_device, levels, count, static_cast<hipStream_t>(stream));
    if (err != hipSuccess) return m_fail(SPD_E_DEVICE, std::string("spd_model_export_pack: ") + hipGetErrorString(err));
    return SPD_OK;
}

// spectral2grid: u, v from vorticity / divergence; q in kg/kg; phi in m; ps in Pa
int spd_model_spectral2grid(spd_model_handle m, int first, int count, void *stream) {
    if (int rc = member_range(m, first, count, "spd_model_spectral2grid")) return rc;
    if (count == 0) return SPD_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const DeviceTables &T = m->ctx->dev;
    const size_t S = NSPEC * C, half = static_cast<size_t>(m->M) * 16, off = static_cast<size_t>(first) * 16;
    // vort2vel over both time levels of the members (contiguous); only level 1 is transformed
    hipError_t e = run_vort2vel(T, m->P.vor + off * S, m->P.div + off * S, m->P.sv + off * S, m->P.sv + (half + off) * S,
                                count * 16, s);
    if (e == hipSuccess) e = run_spec2grid_table(T, m->exp_inv_table[m->phi_cur] + static_cast<size_t>(first) * 41, count * 41, s);
    if (e == hipSuccess)
        e = run_export_units(m->q_grid + static_cast<size_t>(first) * 8 * NG, m->phi_grid + static_cast<size_t>(first) * 8 * NG,
                             m->ps_grid + static_cast<size_t>(first) * NG, static_cast<long>(count) * NG, s);
    if (e != hipSuccess) return m_fail(SPD_E_DEVICE, std::string("spd_model_spectral2grid: ") + hipGetErrorString(e));
    return SPD_OK;
}

// grid2spectral: the inverse mapping into time level 1 (ps_grid is not modified; its logarithm goes through scratch)
int spd_model_grid2spectral(spd_model_handle m, int first, int count, void *stream) {
    if (int rc = member_range(m, first, count, "spd_model_grid2spectral")) return rc;
    if (count == 0) return SPD_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const DeviceTables &T = m->ctx->dev;
    const size_t S = NSPEC * C, half = static_cast<size_t>(m->M) * 16;
    if (int rc = settle_deferred_check(m)) return rc;
    m->phi_ahead = false;  // the temperature changes under the look-ahead geopotential
    hipError_t e = run_grid2spec_table(T, m->exp_fwd_table[m->phi_cur] + static_cast<size_t>(first) * 40, count * 40, s);
    for (int i = first; i < first + count && e == hipSuccess; ++i) {
        const size_t s1 = static_cast<size_t>(i) * 16;
        e = run_vel2vort(T, m->P.sv + s1 * S, m->P.sv + (half + s1) * S, m->P.vor + s1 * S, m->P.div + s1 * S, 8, s);
        if (e == hipSuccess) e = run_export_spec_units(m->P.tr + s1 * S, m->P.phi + static_cast<size_t>(i) * 8 * S, 8 * NSPEC, s);
    }
    if (e == hipSuccess)
        e = run_log_ps(m->ps_grid + static_cast<size_t>(first) * NG, m->corh_t + static_cast<size_t>(first) * NG,
                       static_cast<long>(count) * NG, s);
    for (int i = first; i < first + count && e == hipSuccess; ++i)
        e = run_grid2spec(T, 0, m->corh_t + static_cast<size_t>(i) * NG, m->P.ps + static_cast<size_t>(i) * 2 * S, 0, 1, s);
    if (e != hipSuccess) return m_fail(SPD_E_DEVICE, std::string("spd_model_grid2spectral: ") + hipGetErrorString(e));
    return SPD_OK;
}

// grid_filter: spectral truncation of the six grid-space variables, in place
int spd_model_grid_filter(spd_model_handle m, int first, int count, void *stream) {
    if (int rc = member_range(m, first, count, "spd_model_grid_filter")) return rc;
    if (count == 0) return SPD_OK;
    double *v3[5] = {m->u_grid, m->v_grid, m->t_grid, m->q_grid, m->phi_grid};
    for (double *v : v3) {
        double *p = v + static_cast<size_t>(first) * 8 * NG;
        if (int rc = spd_grid_filter(m->ctx, p, p, count * 8, stream)) return rc;
    }
    double *p = m->ps_grid + static_cast<size_t>(first) * NG;
    return spd_grid_filter(m->ctx, p, p, count, stream);
}

int spd_model_set_sppt(spd_model_handle m, int on, uint64_t seed, int64_t first_member_id) {
    if (!m) return m_fail(SPD_E_ARG, "spd_model_set_sppt: null model");
    if (first_member_id < 0 || first_member_id + m->M >= (1ll << 24)) return m_fail(SPD_E_ARG, "spd_model_set_sppt: member id out of range");
    M_HIP(hipSetDevice(m->ctx->device));
    if (on && !m->sppt_spec) {
        const size_t M = m->M;
        if (int rc = dalloc(m, M * 8 * NSPEC * C, &m->sppt_spec, "sppt_spec", 8 * NSPEC * C * sizeof(double))) return rc;
        if (int rc = dalloc(m, M * 8 * NG, &m->sppt_grid, "sppt_pattern", static_cast<size_t>(8) * NG * sizeof(double))) return rc;
        TableBatch batch;
        build_inverse_tables(m, true, false, m->inv_table_sppt, batch);
        if (int rc = batch.upload(m)) return rc;
        if (m->stored32)
            if (int rc = ensure_tables32(m)) return rc;
    }
    m->sppt_on = on != 0;
    m->sppt_seed = seed;
    m->sppt_member_base = first_member_id;
    m->sppt_first = true;
    m->sppt_step = 0;
    return SPD_OK;
}

// sst_anom(ix, il, 0:n_months+1) for every member (modelstate_init_sst_anom, speedy_driver.f90.j2:225-237); zero-filled
int spd_model_init_sst_anom(spd_model_handle m, int n_months) {
    if (!m) return m_fail(SPD_E_ARG, "spd_model_init_sst_anom: null model");
    if (n_months < 1) return m_fail(SPD_E_ARG, "spd_model_init_sst_anom: n_months must be at least 1");
    M_HIP(hipSetDevice(m->ctx->device));
    const size_t planes = static_cast<size_t>(n_months) + 2;
    double *p = nullptr;
    if (int rc = dalloc(m, static_cast<size_t>(m->M) * planes * NG, &p, "sst_anom", planes * NG * sizeof(double))) return rc;
    m->S.sst_anom = p;  // the previous array stays allocated until spd_model_destroy
    m->anom_planes = static_cast<int>(planes);
    m->surf_cache_valid = false;
    return SPD_OK;
}

// copy every registered variable of member `si` of `src` into member `di` of `dst` (same device, same variable sizes)
int spd_model_copy_member(spd_model_handle dst, int di, spd_model_handle src, int si, void *stream) {
    if (!dst || !src) return m_fail(SPD_E_ARG, "spd_model_copy_member: null model");
    if (di < 0 || di >= dst->M || si < 0 || si >= src->M) return m_fail(SPD_E_ARG, "spd_model_copy_member: member index out of range");
    if (dst->ctx->device != src->ctx->device) return m_fail(SPD_E_ARG, "spd_model_copy_member: models live on different devices");
    if (int rc = usable(dst, "spd_model_copy_member")) return rc;
    if (int rc = usable(src, "spd_model_copy_member")) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    M_HIP(hipSetDevice(dst->ctx->device));
    if (int rc = settle_deferred_check(dst)) return rc;
    if (int rc = settle_deferred_check(src)) return rc;  // (the source usually dies next: its check must be out before)
    M_HIP(hipDeviceSynchronize());  // the two models may have been driven on different streams
    // a member's arrays are copied as they are stored: both models must store them the same way, and the storage belongs to the
    // precision of the column physics (fp32 storage is only ever read by the fp32 kernel): the receiving model takes both over
    dst->phys_fp32 = src->phys_fp32;
    dst->phys_store32 = src->phys_store32;
    if (int rc = apply_storage(dst, src->stored32)) return rc;
    dst->surf_cache_valid = dst->phi_ahead = false;
    CopyList list{};
    for (const auto &kv : src->reg) {
        auto it = dst->reg.find(kv.first);
        if (it == dst->reg.end() || it->second.bytes_member != kv.second.bytes_member)
            return m_fail(SPD_E_SIZE, "spd_model_copy_member: variable '" + kv.first + "' differs between the models");
        // (an array kept as fp32 is compact in fp32: member i starts half as far into the allocation and is half as long)
        const size_t b = (kv.second.f32 && src->stored32) ? kv.second.bytes_member / 2 : kv.second.bytes_member;
        char *to = static_cast<char *>(it->second.ptr) + b * di;
        const char *from = static_cast<const char *>(kv.second.ptr) + b * si;
        if (b % 16 != 0 || b > 0xffffffffu || reinterpret_cast<uintptr_t>(to) % 16 != 0 || reinterpret_cast<uintptr_t>(from) % 16 != 0) {
            M_HIP(hipMemcpyAsync(to, from, b, hipMemcpyDeviceToDevice, s));
            continue;
        }
        // all the arrays of the member in one launch (one more whenever the list is full)
        list.src[list.n] = from;
        list.dst[list.n] = to;
        list.bytes[list.n] = static_cast<unsigned>(b);
        if (++list.n == kCopyListMax) {
            M_HIP(run_multi_copy(list, s));
            list.n = 0;
        }
    }
    M_HIP(run_multi_copy(list, s));
    return SPD_OK;
}

// Copy the named registry variables of member `si` of `src` into member `di` of `dst`; the two models may live on different
// devices (then the bytes travel device to device: hipMemcpyPeerAsync, xGMI between the GPUs of a node).  Asynchronous on
// `stream` (a stream of the DESTINATION device); both devices are synchronised first, as in spd_model_copy_member.
int spd_model_copy_vars(spd_model_handle dst, int di, spd_model_handle src, int si, const char *const *names, int nnames,
                        void *stream) {
    if (!dst || !src || (nnames > 0 && !names)) return m_fail(SPD_E_ARG, "spd_model_copy_vars: null argument");
    const int ddev = dst->ctx->device, sdev = src->ctx->device;
    if (sdev != ddev) {
        M_HIP(hipSetDevice(sdev));
        M_HIP(hipDeviceSynchronize());
    }
    M_HIP(hipSetDevice(ddev));
    M_HIP(hipDeviceSynchronize());
    return spd_model_copy_vars_enqueue(dst, di, src, si, names, nnames, stream);
}

// The same copies without the two device synchronisations in front: for a caller that hands the same fields to many members and
// has synchronised the devices once itself (spd_broadcast_boundary).  Leaves the DESTINATION device current.
int spd_model_copy_vars_enqueue(spd_model_handle dst, int di, spd_model_handle src, int si, const char *const *names, int nnames,
                                void *stream) {
    if (!dst || !src || (nnames > 0 && !names)) return m_fail(SPD_E_ARG, "spd_model_copy_vars: null argument");
    if (di < 0 || di >= dst->M || si < 0 || si >= src->M) return m_fail(SPD_E_ARG, "spd_model_copy_vars: member index out of range");
    const int ddev = dst->ctx->device, sdev = src->ctx->device;
    hipStream_t s = static_cast<hipStream_t>(stream);
    M_HIP(hipSetDevice(ddev));
    if (int rc = settle_deferred_check(dst)) return rc;
    M_HIP(hipSetDevice(ddev));
    dst->surf_cache_valid = dst->phi_ahead = false;
    for (int i = 0; i < nnames; ++i) {
        auto a = src->reg.find(names[i]), b = dst->reg.find(names[i]);
        if (a == src->reg.end() || b == dst->reg.end())
            return m_fail(SPD_E_ARG, std::string("spd_model_copy_vars: unknown variable '") + names[i] + "'");
        if (a->second.bytes_member != b->second.bytes_member)
            return m_fail(SPD_E_SIZE, std::string("spd_model_copy_vars: variable '") + names[i] + "' differs between the models");
        if (a->second.f32 && src->stored32 != dst->stored32)
            return m_fail(SPD_E_ARG, std::string("spd_model_copy_vars: variable '") + names[i] + "' is stored as fp32 in one model and as fp64 in the other");
        const size_t n = (a->second.f32 && src->stored32) ? a->second.bytes_member / 2 : a->second.bytes_member;  // (as stored)
        char *to = static_cast<char *>(b->second.ptr) + n * di;
        const char *from = static_cast<const char *>(a->second.ptr) + n * si;
        if (to == from) continue;
        if (sdev == ddev) M_HIP(hipMemcpyAsync(to, from, n, hipMemcpyDeviceToDevice, s));
        else M_HIP(hipMemcpyPeerAsync(to, ddev, from, sdev, n, s));
    }
    return SPD_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// One collective broadcast of the named variables between device models that live on DIFFERENT GPUs of this process: member
// members[root] of models[root] into member members[i] of every other models[i].  RCCL (ncclBroadcast inside one group call,
// one communicator per device created with ncclCommInitAll -- the single-process form) over xGMI: the start-up hand-over of
// the shared boundary fields of a one-process ensemble, what torch.distributed does for the process-per-GPU layout.
// RCCL is loaded when this is first called (librccl.so.1: the copy PyTorch has already brought into the process, else ROCm's):
// a host that keeps its ensemble on one GPU never touches it, and the library carries no link-time dependency on it.
// Runs on each device's null stream; the caller synchronises (spd_broadcast_boundary does, once per device).
// ---------------------------------------------------------------------------------------------------------------
}  // extern "C"

namespace {
struct Rccl {
    void *lib = nullptr;
    std::string why;  // why it could not be loaded
    int (*CommInitAll)(void **comms, int ndev, const int *devlist) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Broadcast)(const void *send, void *recv, size_t count, int datatype, int root, void *comm, hipStream_t stream) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::map<std::vector<int>, std::vector<void *>> comms;  // by device list; kept for the life of the process
    std::mutex mutex;
    std::string wedged;  // a call into RCCL did not come back (spd_model_broadcast_vars): it is not called again in this process
};
Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        r.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!r.lib) r.lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!r.lib) {
            const char *e = dlerror();
            r.why = e ? e : "librccl.so.1 not found";
            return;
        }
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(r.lib, "ncclCommInitAll"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(r.lib, "ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(r.lib, "ncclGroupEnd"));
        r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(dlsym(r.lib, "ncclBroadcast"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
        if (!r.CommInitAll || !r.GroupStart || !r.GroupEnd || !r.Broadcast || !r.GetErrorString) {
            r.why = "librccl.so.1 lacks an entry point of the collective API";
            r.lib = nullptr;
        }
    });
    return r;
}
constexpr int kNcclFloat64 = 8;  // ncclDataType_t (rccl.h)
}  // namespace

extern "C" {

int spd_model_broadcast_vars(const spd_model_handle *models, const int *members, int n, int root, const char *const *names,
                             int nnames) {
    if (!models || !members || n < 1 || root < 0 || root >= n || (nnames > 0 && !names))
        return m_fail(SPD_E_ARG, "spd_model_broadcast_vars: bad argument");
    std::vector<int> devices(n);
    for (int i = 0; i < n; ++i) {
        if (!models[i] || members[i] < 0 || members[i] >= models[i]->M) return m_fail(SPD_E_ARG, "spd_model_broadcast_vars: bad model / member");
        devices[i] = models[i]->ctx->device;
        for (int j = 0; j < i; ++j)
            if (devices[j] == devices[i]) return m_fail(SPD_E_ARG, "spd_model_broadcast_vars: one model per GPU (same-device copies are spd_model_copy_vars)");
    }
    // Every (variable, model) pair is checked BEFORE the collective layer is touched: a variable refused inside the group call
    // would leave the ranks before it with one broadcast more than the rest, and mismatched collectives do not fail, they hang.
    struct Piece {
        void *ptr;
        size_t count;
    };
    std::vector<std::vector<Piece>> pieces(nnames, std::vector<Piece>(n));  // [variable][model]
    for (int v = 0; v < nnames; ++v) {
        if (!names[v]) return m_fail(SPD_E_ARG, "spd_model_broadcast_vars: null variable name");
        auto r0 = models[root]->reg.find(names[v]);
        for (int i = 0; i < n; ++i) {
            auto e = models[i]->reg.find(names[v]);
            if (e == models[i]->reg.end() || r0 == models[root]->reg.end())
                return m_fail(SPD_E_ARG, std::string("spd_model_broadcast_vars: unknown variable '") + names[v] + "'");
            if (e->second.bytes_member != r0->second.bytes_member)
                return m_fail(SPD_E_SIZE, std::string("spd_model_broadcast_vars: variable '") + names[v] + "' differs between the models");
            if (e->second.f32)
                return m_fail(SPD_E_ARG, std::string("spd_model_broadcast_vars: variable '") + names[v] + "' may be stored as fp32 (spd_model_copy_vars moves those)");
            pieces[v][i] = {static_cast<char *>(e->second.ptr) + e->second.bytes_member * members[i], e->second.bytes_member / sizeof(double)};
        }
    }
    for (int i = 0; i < n; ++i)
        if (i != root)
            if (int rc = settle_deferred_check(models[i])) return rc;
    Rccl &R = rccl();
    if (!R.lib) return m_fail(SPD_E_DEVICE, "spd_model_broadcast_vars: RCCL is not available: " + R.why);
    std::lock_guard<std::mutex> lock(R.mutex);
    if (!R.wedged.empty()) return m_fail(SPD_E_DEVICE, "spd_model_broadcast_vars: RCCL is not used again in this process: " + R.wedged);
    // RCCL's single-process initialisation and its group call talk to every GPU of the list; where a device or a link does not
    // answer they do not fail, they block.  Both run on a thread of their own with a bound on the wait (bounded_call.hpp;
    // PYSPEEDY_AMD_RCCL_TIMEOUT seconds, default 30).  An initialisation that does not come back has enqueued nothing: SPD_E_DEVICE,
    // and the caller may take the point-to-point path (spd_broadcast_boundary does).  A group call or a broadcast that does not
    // come back leaves work of unknown state on the devices' null streams: SPD_E_TIMEOUT, nothing may be queued behind it.
    const double bound = [] {
        const char *e = getenv("PYSPEEDY_AMD_RCCL_TIMEOUT");
        const double v = e ? atof(e) : 30.0;
        return v > 0.0 ? v : 30.0;
    }();
    auto nccl_text = [&R](int rc) { return std::string(R.GetErrorString(rc)); };
    auto it = R.comms.find(devices);
    if (it == R.comms.end()) {
        auto comms = std::make_shared<std::vector<void *>>(n, nullptr);
        auto init = R.CommInitAll;
        const spd::BoundedResult r = spd::run_bounded([init, comms, devices, n] { return init(comms->data(), n, devices.data()); }, bound);
        if (!r.finished) {
            R.wedged = "ncclCommInitAll over " + std::to_string(n) + " devices did not return within " + std::to_string(static_cast<int>(bound)) + " s";
            return m_fail(SPD_E_DEVICE, "spd_model_broadcast_vars: " + R.wedged);
        }
        if (r.rc) return m_fail(SPD_E_DEVICE, "spd_model_broadcast_vars: ncclCommInitAll: " + nccl_text(r.rc));
        it = R.comms.emplace(devices, *comms).first;
    }
    const std::vector<void *> comms = it->second;
    struct GroupStatus {
        int start = 0, first_error = 0, end = 0, set_device = 0;
    };
    auto status = std::make_shared<GroupStatus>();
    auto start = R.GroupStart, finish = R.GroupEnd;
    auto broadcast = R.Broadcast;
    const spd::BoundedResult g = spd::run_bounded([=] {
        status->start = start();
        if (status->start) return 1;
        for (int v = 0; v < nnames; ++v)
            for (int i = 0; i < n; ++i) {
                if (hipSetDevice(devices[i]) != hipSuccess) {
                    status->set_device = 1;
                    continue;  // (the group is still closed below; the call fails as a whole)
                }
                const Piece &p = pieces[v][i];
                const int r = broadcast(p.ptr, p.ptr, p.count, kNcclFloat64, root, comms[i], nullptr);
                if (r && !status->first_error) status->first_error = r;
            }
        status->end = finish();  // (always closed, also after an error inside the group)
        return 0;
    }, bound);
    if (!g.finished) {
        R.wedged = "the group call of " + std::to_string(nnames) + " broadcasts over " + std::to_string(n) + " devices did not return within " +
                   std::to_string(static_cast<int>(bound)) + " s";
        return m_fail(SPD_E_TIMEOUT, "spd_model_broadcast_vars: " + R.wedged);
    }
    if (status->start) return m_fail(SPD_E_DEVICE, "spd_model_broadcast_vars: ncclGroupStart: " + nccl_text(status->start));
    if (status->set_device) return m_fail(SPD_E_TIMEOUT, "spd_model_broadcast_vars: hipSetDevice failed inside the group call");
    if (status->first_error) return m_fail(SPD_E_TIMEOUT, "spd_model_broadcast_vars: ncclBroadcast: " + nccl_text(status->first_error));
    if (status->end) return m_fail(SPD_E_TIMEOUT, "spd_model_broadcast_vars: ncclGroupEnd: " + nccl_text(status->end));
    // ... and the broadcasts themselves: an event behind them on every device's null stream, asked until the bound is up
    {
        std::vector<hipEvent_t> events(n, nullptr);
        bool ok = true;
        for (int i = 0; i < n && ok; ++i)
            ok = hipSetDevice(devices[i]) == hipSuccess && hipEventCreateWithFlags(&events[i], hipEventDisableTiming) == hipSuccess &&
                 hipEventRecord(events[i], nullptr) == hipSuccess;
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(bound);
        int pending = ok ? n : 0;
        hipError_t bad = hipSuccess;
        while (pending > 0 && bad == hipSuccess && std::chrono::steady_clock::now() < deadline) {
            pending = 0;
            for (int i = 0; i < n; ++i) {
                const hipError_t q = hipEventQuery(events[i]);
                if (q == hipErrorNotReady) ++pending;
                else if (q != hipSuccess) bad = q;
            }
            if (pending) std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
        for (int i = 0; i < n; ++i)
            if (events[i] && (pending == 0 || bad != hipSuccess)) (void)hipEventDestroy(events[i]);  // (a pending event is left alone)
        if (!ok) return m_fail(SPD_E_TIMEOUT, "spd_model_broadcast_vars: could not record the completion events of the broadcast");
        if (bad != hipSuccess) return m_fail(SPD_E_TIMEOUT, std::string("spd_model_broadcast_vars: ") + hipGetErrorString(bad));
        if (pending) {
            R.wedged = "the broadcast did not complete on " + std::to_string(pending) + " of " + std::to_string(n) + " devices within " +
                       std::to_string(static_cast<int>(bound)) + " s";
            return m_fail(SPD_E_TIMEOUT, "spd_model_broadcast_vars: " + R.wedged);
        }
    }
    for (int i = 0; i < n; ++i)
        if (i != root) models[i]->surf_cache_valid = models[i]->phi_ahead = false;
    return SPD_OK;
}

}  // extern "C"
