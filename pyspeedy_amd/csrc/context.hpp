// The object behind spd_handle: host tables, their device copies and scratch memory.
#pragma once
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "device_tables.hpp"
#include "model.hpp"

// The tables of the dynamics (horizontal diffusion, semi-implicit scheme; horizontal_diffusion.f90:80-107, implicit.f90:71-218)
// depend on the geometry and on the time step, not on any model state: one immutable copy per context and time step, made when
// a model first asks for that step (a run uses three: delt / 2, delt, 2 delt).  A model that changes its time step switches
// pointers; nothing is rebuilt, uploaded or waited for, and launches in flight keep the set they were given.
struct spd_dyn_tables {
    spd::DynHostTables host;
    spd::DynDeviceTables dev{};
    explicit spd_dyn_tables(const spd::DynHostTables &base) : host(base) {}
};

struct spd_context {
    int device = 0;
    spd::HostTables host;
    spd::DeviceTables dev{};
    std::vector<void *> allocations;
    // scratch for composite operators (grid_vel2vort, grid_filter); grows on demand, never shrinks
    double *scratch = nullptr;
    size_t scratch_bytes = 0;
    std::mutex scratch_mutex;
    // buffer of spd_stream_probe (stream_probe.hip): grows on demand, lives as long as the context; one probe at a time
    void *probe_buf = nullptr;
    size_t probe_bytes = 0;
    std::mutex probe_mutex;
    // Memory blocks of models that have died, kept for the next model of the same size (model.hip: arena_alloc / spd_model_destroy):
    // a host with the reference's call sequence creates and closes one-member models by the hundred, and hipFree costs 0.22 ms.
    struct IdleBlock {
        void *base;
        size_t size;
    };
    std::vector<IdleBlock> idle_blocks;
    size_t idle_bytes = 0;
    std::mutex idle_mutex;
    std::mutex dyn_mutex;
    std::unique_ptr<spd_dyn_tables> dyn_base;                       // time step 0: the dt-independent tables only
    std::map<double, std::unique_ptr<spd_dyn_tables>> dyn_by_step;  // never shrinks while the context lives (kMaxDynSteps)
};

// records the message returned by spd_last_error() (thread-local) and returns `code`
int spd_set_error(int code, const std::string &msg);
