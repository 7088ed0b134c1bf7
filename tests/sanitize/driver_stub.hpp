// TEST INFRASTRUCTURE (see driver_stub.cpp): knobs of the pretend GPU side and the two rules by which its toy model moves,
// shared with driver_sanitize.cpp so that the test can predict what every container must hold.
#pragma once
#include <cstdint>
#include <cstring>

void stub_set_device_count(int n);
void stub_set_check_delay_us(int us);         // how long spd_model_check_end "waits for the GPU"
void stub_fail_next_check_begin(int count);   // the next `count` range checks cannot be enqueued (a device error after the step)
void stub_fail_next_steps_begin(int count);   // the next `count` spd_model_step_checked_begin calls fail before anything is enqueued
long stub_device_syncs();
long stub_peer_copies();
long stub_local_copies();
long stub_collectives();                      // collective broadcasts between device models
void stub_fail_model_create_in(int n);  // the n-th spd_model_create from now fails like a hipMalloc that is out of memory (0: none)
void stub_set_collective_available(int yes);  // 0: spd_model_broadcast_vars fails as it does when RCCL cannot be loaded; 1: it works;
                                              // 2: as when ncclCommInitAll does not come back inside its bound (SPD_E_DEVICE, nothing
                                              // enqueued); 3: as when the broadcast does not complete inside its bound (SPD_E_TIMEOUT)
int stub_current_device();
void stub_set_current_device(int d);

// one model step: 40 minutes on a 365-day calendar; month_idx counts the months begun since the start
inline void stub_advance(int32_t &year, int32_t &month, int32_t &day, int32_t &hour, int32_t &minute, int32_t &month_idx) {
    static const int days[12] = {31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31};
    minute += 40;
    if (minute >= 60) { minute -= 60; hour += 1; }
    if (hour >= 24) { hour -= 24; day += 1; }
    if (day > days[month - 1]) { day = 1; month += 1; month_idx += 1; }
    if (month > 12) { month = 1; year += 1; }
}

// the fingerprint of a member after a step taken when its model's step counter stood at `step`
inline double stub_fingerprint(double before, int32_t step) {
    uint64_t u;
    std::memcpy(&u, &before, sizeof(u));
    u = (u * 6364136223846793005ull + 1442695040888963407ull + static_cast<uint64_t>(step)) & 0x000fffffffffffffull;  // stays a finite double
    double out;
    std::memcpy(&out, &u, sizeof(out));
    return out;
}
