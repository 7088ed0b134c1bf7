// The object behind spd_handle: host tables, their device copies and scratch memory.
#pragma once
#include <mutex>
#include <string>
#include <vector>

#include "device_tables.hpp"

struct spd_context {
    int device = 0;
    spd::HostTables host;
    spd::DeviceTables dev{};
    std::vector<void *> allocations;
    // scratch for composite operators (grid_vel2vort, grid_filter); grows on demand, never shrinks
    double *scratch = nullptr;
    size_t scratch_bytes = 0;
    std::mutex scratch_mutex;
};

// records the message returned by spd_last_error() (thread-local) and returns `code`
int spd_set_error(int code, const std::string &msg);
