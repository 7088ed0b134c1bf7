// TEST INFRASTRUCTURE: drives the product's outer boundary (pyspeedy_amd/csrc/driver.cpp, through include/pyspeedy_amd_driver.h)
// over the pretend GPU side of driver_stub.cpp.  Built three times by tests/test_sanitizers.py -- AddressSanitizer +
// UndefinedBehaviorSanitizer, ThreadSanitizer, and plain -- and run as `driver_sanitize [single|threads|all]`.
//
// What is checked is the host logic: containers, placement and device switching, gathering / splitting, the kept plan, the
// pending-step tokens and their error paths, dates, and -- with several host threads stepping disjoint sets of containers while
// the library's lock is released around every wait -- that none of it races.  Every container's toy state after n steps is
// predictable (driver_stub.hpp), whatever grouping the driver chose on the way.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pyspeedy_amd.h"
#include "../../include/pyspeedy_amd_driver.h"
#include "../../pyspeedy_amd/csrc/bounded_call.hpp"
#include "driver_stub.hpp"

#define EXPECT(cond)                                                                                      \
    do {                                                                                                  \
        if (!(cond)) {                                                                                    \
            std::fprintf(stderr, "%s:%d: EXPECT(%s) failed; last error: %s\n", __FILE__, __LINE__, #cond, spd_last_error()); \
            std::abort();                                                                                 \
        }                                                                                                 \
    } while (0)

namespace {
constexpr size_t NG = 96 * 48;

struct Member {
    int64_t state = 0, control = 0, d0 = 0, d1 = 0;
    double seed = 0.0;
    int steps = 0;                 // steps this container has taken since its init
    int32_t date[5] = {1982, 1, 30, 0, 0};
    int32_t month_idx = 1;
    bool poisoned = false;

    void predict_step() {
        stub_advance(date[0], date[1], date[2], date[3], date[4], month_idx);
        steps += 1;
    }
    double fingerprint() const {
        double f = seed;
        for (int k = 0; k < steps; ++k) f = stub_fingerprint(f, k);
        return f;
    }
};

void set_seed(Member &m, double seed) {
    std::vector<double> orog(NG, 0.0);
    orog[0] = seed;
    EXPECT(spd_set(m.state, "orog", orog.data(), NG * sizeof(double)) == 0);
    m.seed = seed;
}

void make_controls(Member &m) {
    EXPECT(spd_create_datetime(1982, 1, 30, 0, 0, &m.d0) == 0);
    EXPECT(spd_create_datetime(1983, 1, 1, 0, 0, &m.d1) == 0);
    EXPECT(spd_controlparams_init(&m.control, m.d0, m.d1) == 0);
}

void init(Member &m) {
    int32_t code = 7;
    EXPECT(spd_init(m.state, m.control, &code) == 0 && code == 0);
    m.steps = 0;
    const int32_t start[5] = {1982, 1, 30, 0, 0};
    std::memcpy(m.date, start, sizeof(start));
    m.month_idx = 1;
    m.poisoned = false;
}

void close(Member &m) {
    EXPECT(spd_modelstate_close(m.state) == 0);
    EXPECT(spd_controlparams_close(m.control) == 0);
    EXPECT(spd_close_datetime(m.d0) == 0 && spd_close_datetime(m.d1) == 0);
}

void verify(const Member &m) {
    std::vector<double> olr(NG);
    EXPECT(spd_get(m.state, "olr", olr.data(), NG * sizeof(double)) == 0);
    EXPECT(olr[0] == static_cast<double>(m.steps));
    EXPECT(olr[1] == m.fingerprint());
    int32_t now[5], midx = 0;
    EXPECT(spd_controlparams_get_model_datetime(m.control, now, &midx) == 0);
    EXPECT(std::memcmp(now, m.date, sizeof(now)) == 0 && midx == m.month_idx);
    int32_t step = -1;
    EXPECT(spd_get(m.state, "current_step", &step, sizeof(step)) == 0 && step == m.steps);
}

struct Lists {
    std::vector<int64_t> s, c;
    std::vector<int32_t> codes;
    explicit Lists(const std::vector<Member *> &ms) {
        for (Member *m : ms) {
            s.push_back(m->state);
            c.push_back(m->control);
        }
        codes.assign(ms.size(), 99);
    }
};

// one synchronous parallel_step over `ms`, all expected to succeed
void step_all(const std::vector<Member *> &ms) {
    Lists l(ms);
    EXPECT(spd_parallel_step(l.s.data(), l.c.data(), l.codes.data(), static_cast<int32_t>(ms.size())) == 0);
    for (size_t i = 0; i < ms.size(); ++i) {
        EXPECT(l.codes[i] == 0);
        ms[i]->predict_step();
    }
}

int models_alive() {
    int32_t n = -1;
    EXPECT(spd_driver_stats(0, &n, nullptr) == 0);
    return n;
}

int members_in_model(const Member &m) {
    int32_t n = -1;
    EXPECT(spd_driver_stats(m.state, nullptr, &n) == 0);
    return n;
}

// ---------------------------------------------------------------------------------------------------------------------
void single_thread() {
    stub_set_device_count(2);
    stub_set_check_delay_us(0);
    stub_set_current_device(1);  // the caller's current device: must still be 1 after every call below
    EXPECT(models_alive() == 0);
    EXPECT(spd_set_device_placement(3) < 0 && spd_set_device_placement(2) == 0);
    std::vector<Member> m(6);
    for (int i = 0; i < 6; ++i) {
        EXPECT(spd_modelstate_init(&m[i].state) == 0);
        int32_t dev = -1;
        EXPECT(spd_modelstate_device(m[i].state, &dev) == 0 && dev == i % 2);  // round-robin in creation order
        make_controls(m[i]);
        EXPECT(spd_modelstate_init_sst_anom(m[i].state, i == 5 ? 4 : 2) == 0);  // (the last one with another anomaly length)
        EXPECT(stub_current_device() == 1);
    }
    EXPECT(spd_set_device_placement(0) == 0);
    // ---- boundary broadcast: the fields cross to the other device once -- point to point where there is no collective
    // library (or PYSPEEDY_AMD_BROADCAST=peer), with ONE collective broadcast otherwise
    set_seed(m[0], 1234.5);
    std::vector<int64_t> all;
    for (auto &x : m) all.push_back(x.state);
    long peer0 = stub_peer_copies(), local0 = stub_local_copies(), sync0 = stub_device_syncs(), coll0 = stub_collectives();
    stub_set_collective_available(0);
    EXPECT(spd_broadcast_boundary(all.data(), 6, 0) == 0);
    EXPECT(stub_peer_copies() - peer0 == 1 && stub_local_copies() - local0 == 4 && stub_collectives() == coll0);
    EXPECT(stub_device_syncs() - sync0 == 5);  // each of the two devices once before and once after, the root's once more
    int32_t peer = -1, local = -1, coll = -1;
    EXPECT(spd_broadcast_boundary_stats(&peer, &local, &coll) == 0 && peer == 1 && local == 4 && coll == 0);
    stub_set_collective_available(1);
    set_seed(m[0], 4321.0);
    peer0 = stub_peer_copies(), local0 = stub_local_copies();
    EXPECT(spd_broadcast_boundary(all.data(), 6, 0) == 0);
    // m[1] (device 1) received collectively; m[5] shares the device but not the anomaly length: no container can hand it 13 fields
    EXPECT(stub_collectives() - coll0 == 1 && stub_peer_copies() == peer0 && stub_local_copies() - local0 == 4);
    EXPECT(spd_broadcast_boundary_stats(&peer, &local, &coll) == 0 && peer == 0 && local == 4 && coll == 1);
    for (auto &x : m) {
        std::vector<double> orog(NG);
        EXPECT(spd_get(x.state, "orog", orog.data(), NG * sizeof(double)) == 0 && orog[0] == 4321.0);
    }
    set_seed(m[0], 1234.5);
    EXPECT(spd_broadcast_boundary(all.data(), 6, 0) == 0);
    EXPECT(stub_current_device() == 1);
    for (auto &x : m) {
        std::vector<double> orog(NG);
        EXPECT(spd_get(x.state, "orog", orog.data(), NG * sizeof(double)) == 0 && orog[0] == 1234.5);
    }
    EXPECT(spd_broadcast_boundary(all.data(), 6, 6) < 0 && spd_broadcast_boundary(all.data(), 0, 0) < 0);
    EXPECT(std::string(spd_broadcast_boundary_note()).find("one RCCL broadcast to 1 other device") == 0);
    // ---- a collective library that does not answer.  Its initialisation not coming back inside the bound has enqueued nothing:
    // the fields go point to point and the note says why.  A broadcast that was enqueued and does not complete cannot be worked
    // around (whatever is queued behind it waits with it): the call fails with SPD_E_TIMEOUT and the reason.
    stub_set_collective_available(2);
    set_seed(m[0], 777.0);
    peer0 = stub_peer_copies(), coll0 = stub_collectives();
    EXPECT(spd_broadcast_boundary(all.data(), 6, 0) == 0);
    EXPECT(stub_peer_copies() - peer0 == 1 && stub_collectives() == coll0);
    EXPECT(spd_broadcast_boundary_stats(&peer, &local, &coll) == 0 && peer == 1 && local == 4 && coll == 0);
    {
        const std::string note = spd_broadcast_boundary_note();
        EXPECT(note.find("peer copies, because:") == 0 && note.find("did not return within") != std::string::npos);
    }
    for (auto &x : m) {
        std::vector<double> orog(NG);
        EXPECT(spd_get(x.state, "orog", orog.data(), NG * sizeof(double)) == 0 && orog[0] == 777.0);
    }
    stub_set_collective_available(3);
    peer0 = stub_peer_copies(), local0 = stub_local_copies();
    EXPECT(spd_broadcast_boundary(all.data(), 6, 0) == SPD_E_TIMEOUT);
    EXPECT(stub_peer_copies() == peer0 && stub_local_copies() == local0);  // nothing was queued behind the stuck collective
    EXPECT(std::string(spd_broadcast_boundary_note()).find("failed:") == 0);
    EXPECT(std::string(spd_last_error()).find("did not complete") != std::string::npos);
    stub_set_collective_available(1);
    EXPECT(spd_broadcast_boundary(all.data(), 6, 0) == 0);
    // ---- the bound itself (csrc/bounded_call.hpp, what spd_model_broadcast_vars wraps RCCL's calls in): a call that returns is
    // reported with its value; a call that NEVER returns costs the caller the bound and nothing else -- its thread is left
    // behind with everything it uses kept alive by the state it owns (released here at the end so that the run ends clean)
    {
        auto quick = spd::run_bounded([] { return 42; }, 5.0);
        EXPECT(quick.finished && quick.rc == 42);
        auto gate = std::make_shared<std::pair<std::mutex, std::condition_variable>>();
        auto open = std::make_shared<std::atomic<bool>>(false);
        auto entered = std::make_shared<std::atomic<int>>(0);
        const auto t0 = std::chrono::steady_clock::now();
        auto stuck = spd::run_bounded([gate, open, entered] {
            entered->fetch_add(1);
            std::unique_lock<std::mutex> lk(gate->first);
            gate->second.wait(lk, [&] { return open->load(); });
            entered->fetch_add(1);
            return 7;
        }, 0.3);
        const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        EXPECT(!stuck.finished && waited >= 0.29 && waited < 5.0 && entered->load() == 1);
        {
            std::lock_guard<std::mutex> lk(gate->first);
            open->store(true);
        }
        gate->second.notify_all();
        for (int spin = 0; spin < 2000 && entered->load() < 2; ++spin) std::this_thread::sleep_for(std::chrono::milliseconds(1));
        EXPECT(entered->load() == 2);
        std::this_thread::sleep_for(std::chrono::milliseconds(20));  // (the thread's last lines run after the counter moved)
    }
    // ---- an ensemble whose LATER device model cannot be created (a GPU out of memory): all containers or none.  40 members on
    // one device are two device models, 12 over two devices one per device; the second creation fails -- the first model is
    // given back, every container id of the call is 0, the error is the allocation's, and the next ensemble comes up as usual.
    {
        const int before = models_alive();
        std::vector<int64_t> ens(40, -1);
        stub_fail_model_create_in(2);
        EXPECT(spd_modelstate_init_ensemble(ens.data(), 40) == SPD_E_DEVICE);
        EXPECT(std::string(spd_last_error()).find("out of memory") != std::string::npos);
        for (int64_t c : ens) EXPECT(c == 0);
        EXPECT(models_alive() == before);
        std::vector<int64_t> two(12, -1);
        stub_fail_model_create_in(2);
        EXPECT(spd_modelstate_init_ensemble_on(two.data(), 12, 2) == SPD_E_DEVICE);
        for (int64_t c : two) EXPECT(c == 0);
        EXPECT(models_alive() == before);
        stub_fail_model_create_in(1);
        int64_t single = -1;
        EXPECT(spd_modelstate_init(&single) == SPD_E_DEVICE && models_alive() == before);
        EXPECT(spd_modelstate_init_ensemble(ens.data(), 40) == 0 && models_alive() == before + 2);
        for (int64_t c : ens) EXPECT(c > 0 && spd_modelstate_close(c) == 0);
        EXPECT(models_alive() == before);
    }
    // ---- uninitialised containers step with code -1 and keep their date
    {
        std::vector<Member *> ms = {&m[0], &m[1]};
        Lists l(ms);
        EXPECT(spd_parallel_step(l.s.data(), l.c.data(), l.codes.data(), 2) == 0 && l.codes[0] == -1 && l.codes[1] == -1);
    }
    for (int i = 0; i < 6; ++i) {
        set_seed(m[i], 100.0 + i);
        init(m[i]);
        EXPECT(stub_current_device() == 1);
    }
    EXPECT(models_alive() == 6);
    std::vector<Member *> everyone;
    for (auto &x : m) everyone.push_back(&x);
    // ---- the first parallel_step gathers per device and per anomaly length; 80 more steps cross into February
    step_all(everyone);
    EXPECT(models_alive() == 3 && members_in_model(m[0]) == 3 && members_in_model(m[1]) == 2 && members_in_model(m[5]) == 1);
    for (int k = 0; k < 80; ++k) step_all(everyone);
    EXPECT(m[0].date[1] == 2 && m[0].month_idx == 2);
    for (auto &x : m) verify(x);
    EXPECT(stub_current_device() == 1);
    // ---- another grouping: the batches are taken apart, the containers regroup by date afterwards
    step_all({&m[0], &m[1]});
    EXPECT(models_alive() == 6);
    step_all(everyone);  // m[0], m[1] are a step ahead of the others: {0}, {2, 4}, {1}, {3}, {5}
    EXPECT(models_alive() == 5 && members_in_model(m[2]) == 2 && members_in_model(m[0]) == 1);
    step_all({&m[2], &m[3], &m[4], &m[5]});  // ... which catch up,
    step_all(everyone);                       // and everybody is together again
    for (auto &x : m) verify(x);
    // the same container twice, a dead container, a dead control container
    {
        Lists l({&m[0], &m[0]});
        EXPECT(spd_parallel_step(l.s.data(), l.c.data(), l.codes.data(), 2) < 0);
        Lists d({&m[0], &m[1]});
        d.s[1] = 987654;
        EXPECT(spd_parallel_step(d.s.data(), d.c.data(), d.codes.data(), 2) < 0);
        d.s[1] = m[1].state;
        d.c[0] = 987654;
        EXPECT(spd_parallel_step(d.s.data(), d.c.data(), d.codes.data(), 2) < 0);
    }
    for (auto &x : m) verify(x);
    // ---- begin / end: two steps in flight per model, the third is refused before anything is enqueued
    {
        Lists l(everyone);
        const int32_t n = 6;
        int64_t t1 = 0, t2 = 0, t3 = 0;
        EXPECT(spd_parallel_step_begin(l.s.data(), l.c.data(), n, &t1) == 0);
        EXPECT(spd_parallel_step_begin(l.s.data(), l.c.data(), n, &t2) == 0);
        EXPECT(spd_parallel_step_begin(l.s.data(), l.c.data(), n, &t3) == 0);  // accepted as a token, refused per device model
        EXPECT(spd_parallel_step_end(t1, l.codes.data()) == 0);
        for (int i = 0; i < n; ++i) EXPECT(l.codes[i] == 0);
        EXPECT(spd_parallel_step_end(t3, l.codes.data()) < 0);
        EXPECT(std::strstr(spd_last_error(), "in flight") != nullptr);
        for (int i = 0; i < n; ++i) EXPECT(l.codes[i] == -3);
        EXPECT(spd_parallel_step_end(t2, l.codes.data()) == 0);
        EXPECT(spd_parallel_step_end(t2, l.codes.data()) < 0);  // a token ends once
        for (auto *x : everyone) {
            x->predict_step();
            x->predict_step();
        }
        for (auto &x : m) verify(x);
        EXPECT(stub_current_device() == 1);
        // a pipelined loop, as SpeedyEns.run issues it
        int64_t pending = 0, next = 0;
        EXPECT(spd_parallel_step_begin(l.s.data(), l.c.data(), n, &pending) == 0);
        for (int k = 0; k < 25; ++k) {
            EXPECT(spd_parallel_step_begin(l.s.data(), l.c.data(), n, &next) == 0);
            EXPECT(spd_parallel_step_end(pending, l.codes.data()) == 0);
            pending = next;
        }
        EXPECT(spd_parallel_step_end(pending, l.codes.data()) == 0);
        for (auto *x : everyone)
            for (int k = 0; k < 26; ++k) x->predict_step();
        for (auto &x : m) verify(x);
    }
    // ---- a member leaves the accepted range: it keeps its date, the others go on, and it is stepped on its own from then on
    {
        std::vector<double> olr(NG);
        EXPECT(spd_get(m[2].state, "olr", olr.data(), NG * sizeof(double)) == 0);
        olr[2] = 1.0;
        EXPECT(spd_set(m[2].state, "olr", olr.data(), NG * sizeof(double)) == 0);
        Lists l(everyone);
        EXPECT(spd_parallel_step(l.s.data(), l.c.data(), l.codes.data(), 6) == 0);
        for (int i = 0; i < 6; ++i) EXPECT(l.codes[i] == (i == 2 ? -2 : 0));
        for (int i = 0; i < 6; ++i)
            if (i != 2) m[i].predict_step();
        m[2].steps += 1;  // (its state moved, its date did not: speedy.f90:57-71)
        int32_t now[5], midx;
        EXPECT(spd_controlparams_get_model_datetime(m[2].control, now, &midx) == 0 && std::memcmp(now, m[2].date, sizeof(now)) == 0);
        int32_t code = 0;
        EXPECT(spd_check(m[2].state, &code) == 0 && code == -2);
        EXPECT(spd_check(m[0].state, &code) == 0 && code == 0);
        // begin / end with the failing member in the list: its date is put back at _end, the others keep theirs
        int64_t t = 0;
        EXPECT(spd_parallel_step_begin(l.s.data(), l.c.data(), 6, &t) == 0);
        EXPECT(spd_parallel_step_end(t, l.codes.data()) == 0);
        for (int i = 0; i < 6; ++i) EXPECT(l.codes[i] == (i == 2 ? -2 : 0));
        for (int i = 0; i < 6; ++i)
            if (i != 2) m[i].predict_step();
        m[2].steps += 1;
        EXPECT(spd_controlparams_get_model_datetime(m[2].control, now, &midx) == 0 && std::memcmp(now, m[2].date, sizeof(now)) == 0);
        set_seed(m[2], 555.0);
        init(m[2]);  // a new start is the only defined continuation
        init(m[4]);  // (and one that was fine, from inside its batch)
        step_all(everyone);
        for (auto &x : m) verify(x);
    }
    // ---- a device error between the step and its check: -3, the date stays, the model refuses to go on until initialised anew
    {
        Lists l({&m[1], &m[3]});
        const int alive = models_alive();
        stub_fail_next_check_begin(1);
        EXPECT(spd_parallel_step(l.s.data(), l.c.data(), l.codes.data(), 2) < 0);
        EXPECT(l.codes[0] == -3 && l.codes[1] == -3);
        EXPECT(std::strstr(spd_last_error(), "injected") != nullptr);
        EXPECT(spd_parallel_step(l.s.data(), l.c.data(), l.codes.data(), 2) < 0 && l.codes[0] == -3);
        EXPECT(std::strstr(spd_last_error(), "initialise") != nullptr);
        int64_t t = 0;
        EXPECT(spd_parallel_step_begin(l.s.data(), l.c.data(), 2, &t) == 0);
        EXPECT(spd_parallel_step_end(t, l.codes.data()) < 0 && l.codes[1] == -3);
        int32_t now[5], midx;
        EXPECT(spd_controlparams_get_model_datetime(m[1].control, now, &midx) == 0 && std::memcmp(now, m[1].date, sizeof(now)) == 0);
        step_all({&m[0], &m[2], &m[4], &m[5]});  // the other device model is not held up
        init(m[1]);
        init(m[3]);
        step_all({&m[1], &m[3]});
        step_all(everyone);
        for (auto &x : m) verify(x);
        (void)alive;
    }
    // ---- k steps as one call (spd_parallel_steps_begin / _end): every step's check is there; a failure in the MIDDLE of a stretch
    //      gives the member the code of that step, the steps before it and the date after its last accepted step
    {
        Lists l(everyone);
        std::vector<int32_t> done(6, -7);
        int64_t t = 0;
        EXPECT(spd_parallel_steps_begin(l.s.data(), l.c.data(), 6, 0, &t) < 0);     // at least one step
        EXPECT(spd_parallel_steps_begin(l.s.data(), l.c.data(), 6, 5000, &t) < 0);  // at most 4096
        EXPECT(spd_parallel_steps_begin(l.s.data(), l.c.data(), 6, 9, &t) == 0);
        EXPECT(spd_parallel_step_end(t, l.codes.data()) < 0);                       // (not a token of the single-step form)
        {   // no single step on the same containers while the stretch is in flight
            int64_t t2 = 0;
            std::vector<int32_t> c2(6, 99);
            EXPECT(spd_parallel_steps_begin(l.s.data(), l.c.data(), 6, 2, &t2) == 0);
            EXPECT(spd_parallel_steps_end(t2, c2.data(), nullptr) < 0 && c2[0] == -3);
            EXPECT(std::strstr(spd_last_error(), "in flight") != nullptr);
        }
        EXPECT(spd_parallel_steps_end(t, l.codes.data(), done.data()) == 0);
        EXPECT(spd_parallel_steps_end(t, l.codes.data(), done.data()) < 0);  // a token is good for one _end
        for (int i = 0; i < 6; ++i) {
            EXPECT(l.codes[i] == 0 && done[i] == 9);
            for (int k = 0; k < 9; ++k) m[i].predict_step();
        }
        for (auto &x : m) verify(x);
        // member 3 leaves the accepted range once its step counter reaches steps + 4: the 4th step of a stretch of 7 fails
        std::vector<double> olr(NG);
        EXPECT(spd_get(m[3].state, "olr", olr.data(), NG * sizeof(double)) == 0);
        olr[2] = 1.0;
        olr[3] = static_cast<double>(m[3].steps + 4);
        EXPECT(spd_set(m[3].state, "olr", olr.data(), NG * sizeof(double)) == 0);
        EXPECT(spd_parallel_steps_begin(l.s.data(), l.c.data(), 6, 7, &t) == 0);
        EXPECT(spd_parallel_steps_end(t, l.codes.data(), done.data()) == 0);
        for (int i = 0; i < 6; ++i) {
            EXPECT(l.codes[i] == (i == 3 ? -2 : 0) && done[i] == (i == 3 ? 3 : 7));
            if (i != 3)
                for (int k = 0; k < 7; ++k) m[i].predict_step();
        }
        for (int k = 0; k < 3; ++k) m[3].predict_step();  // its date: after the three steps it completed
        int32_t now[5], midx;
        EXPECT(spd_controlparams_get_model_datetime(m[3].control, now, &midx) == 0 && std::memcmp(now, m[3].date, sizeof(now)) == 0 &&
               midx == m[3].month_idx);
        for (int i = 0; i < 6; ++i)
            if (i != 3) verify(m[i]);
        set_seed(m[3], 777.0);
        init(m[3]);  // a new start is the only defined continuation
        step_all(everyone);
        for (auto &x : m) verify(x);
        // a device error while the stretch is enqueued: -3 for that model's members, the date stays, initialise again
        Lists l2({&m[1], &m[3]});
        stub_fail_next_steps_begin(2);  // (the two may be two device models by now: both refuse)
        EXPECT(spd_parallel_steps_begin(l2.s.data(), l2.c.data(), 2, 4, &t) == 0);
        EXPECT(spd_parallel_steps_end(t, l2.codes.data(), done.data()) < 0 && l2.codes[0] == -3 && l2.codes[1] == -3 && done[0] == 0);
        stub_fail_next_steps_begin(0);
        EXPECT(std::strstr(spd_last_error(), "injected") != nullptr);
        EXPECT(spd_controlparams_get_model_datetime(m[1].control, now, &midx) == 0 && std::memcmp(now, m[1].date, sizeof(now)) == 0);
        step_all({&m[1], &m[3]});  // (the refusal left the model as it was: nothing had been enqueued)
        for (auto &x : m) verify(x);
    }
    // ---- scalars are per device model: a member that wants its own leaves the batch
    {
        const int before = models_alive();
        int32_t flag = 0;
        EXPECT(spd_set(m[0].state, "land_coupling_flag", &flag, sizeof(flag)) == 0);
        EXPECT(models_alive() > before || members_in_model(m[0]) == 1);
        EXPECT(spd_get(m[0].state, "land_coupling_flag", &flag, sizeof(flag)) == 0 && flag == 0);
        EXPECT(spd_get(m[2].state, "land_coupling_flag", &flag, sizeof(flag)) == 0 && flag == 1);
        step_all(everyone);
        for (auto &x : m) verify(x);
        float lat[48];
        EXPECT(spd_get(m[0].state, "lat", lat, sizeof(lat)) == 0 && spd_set(m[0].state, "lat", lat, sizeof(lat)) < 0);
        double wrong[3];
        EXPECT(spd_get(m[0].state, "olr", wrong, sizeof(wrong)) == -3);  // SPD_E_SIZE
        std::vector<double> host(NG, 2.5), back(NG);
        EXPECT(spd_set(m[0].state, "snowcv", host.data(), NG * sizeof(double)) == 0);
        EXPECT(spd_get(m[0].state, "snowcv", back.data(), NG * sizeof(double)) == 0 && back[17] == 2.5);
        EXPECT(spd_transform_spectral2grid(m[0].state) == 0 && spd_transform_grid2spectral(m[0].state) == 0 &&
               spd_apply_grid_filter(m[0].state) == 0);
        EXPECT(stub_current_device() == 1);
    }
    // ---- abandoned tokens, containers closed while a step is pending
    {
        Lists l(everyone);
        int64_t t1 = 0, t2 = 0;
        EXPECT(spd_parallel_step_begin(l.s.data(), l.c.data(), 6, &t1) == 0);
        EXPECT(spd_parallel_step_begin(l.s.data(), l.c.data(), 6, &t2) == 0);
        for (auto &x : m) close(x);
        EXPECT(spd_parallel_step_end(t1, l.codes.data()) == 0);  // the device models lived on with the pending step
        for (int i = 0; i < 6; ++i) EXPECT(l.codes[i] == 0);
        EXPECT(models_alive() > 0);                               // (t2 still holds them)
        EXPECT(spd_parallel_step_end(t2, l.codes.data()) == 0);
        EXPECT(models_alive() == 0);
        EXPECT(stub_current_device() == 1);
    }
    // ---- ensembles batched from the start: one device model per device up to 31 members of a device, two from 32 up
    {
        std::vector<int64_t> ids(70);
        EXPECT(spd_modelstate_init_ensemble_on(ids.data(), 40, 2) == 0);
        EXPECT(models_alive() == 2);
        int32_t dev = -1;
        EXPECT(spd_modelstate_device(ids[19], &dev) == 0 && dev == 0 && spd_modelstate_device(ids[20], &dev) == 0 && dev == 1);
        for (int i = 0; i < 40; ++i) EXPECT(spd_modelstate_close(ids[i]) == 0);
        EXPECT(models_alive() == 0);
        EXPECT(spd_modelstate_init_ensemble_on(ids.data(), 70, 2) == 0);
        EXPECT(models_alive() == 4);
        EXPECT(spd_modelstate_init_ensemble_on(ids.data(), 3, 5) < 0);
        // an ensemble steps as it was created; its members are initialised one by one from inside the batch
        std::vector<Member> e(70);
        std::vector<Member *> ep;
        for (int i = 0; i < 70; ++i) {
            e[i].state = ids[i];
            make_controls(e[i]);
            set_seed(e[i], 1000.0 + i);
            if (i < 35) init(e[i]);  // (the two device models of device 0 member by member, ...)
            ep.push_back(&e[i]);
        }
        {   // ... the two of device 1 in one pass each
            std::vector<Member *> rest(ep.begin() + 35, ep.end());
            Lists l(rest);
            EXPECT(spd_init_ensemble(l.s.data(), l.c.data(), l.codes.data(), 35) == 0);
            for (int i = 0; i < 35; ++i) EXPECT(l.codes[i] == 0);
            EXPECT(spd_init_ensemble(l.s.data(), l.c.data(), l.codes.data(), -1) < 0);
        }
        EXPECT(models_alive() == 4);
        for (int k = 0; k < 5; ++k) step_all(ep);
        EXPECT(models_alive() == 4);
        for (auto &x : e) verify(x);
        {   // a multi-step call merges the two device models of each device into ONE (the model's own multi-step plan then forms
            // the member groups); a host that goes back to single steps gets the two halves back
            Lists l(ep);
            int64_t t = 0;
            std::vector<int32_t> done(70, -1);
            EXPECT(spd_parallel_steps_begin(l.s.data(), l.c.data(), 70, 6, &t) == 0);
            EXPECT(models_alive() == 2);
            EXPECT(spd_parallel_steps_end(t, l.codes.data(), done.data()) == 0);
            for (int i = 0; i < 70; ++i) {
                EXPECT(l.codes[i] == 0 && done[i] == 6);
                for (int k = 0; k < 6; ++k) e[i].predict_step();
            }
            for (auto &x : e) verify(x);
            EXPECT(spd_parallel_steps_begin(l.s.data(), l.c.data(), 70, 3, &t) == 0);  // (the kept plan: nothing is re-cut)
            EXPECT(models_alive() == 2);
            EXPECT(spd_parallel_steps_end(t, l.codes.data(), done.data()) == 0);
            for (int i = 0; i < 70; ++i)
                for (int k = 0; k < 3; ++k) e[i].predict_step();
            step_all(ep);
            EXPECT(models_alive() == 4);
            step_all(ep);
            EXPECT(models_alive() == 4);
            for (auto &x : e) verify(x);
        }
        stub_set_current_device(0);
        EXPECT(spd_modelstate_init_ensemble(ids.data(), 3) == 0);  // no placement: the current device
        EXPECT(spd_modelstate_device(ids[2], &dev) == 0 && dev == 0);
        for (int i = 0; i < 3; ++i) EXPECT(spd_modelstate_close(ids[i]) == 0);
        for (auto &x : e) close(x);
        EXPECT(models_alive() == 0);
    }
    std::printf("driver sanitize: single thread ok\n");
}

// ---------------------------------------------------------------------------------------------------------------------
// Several host threads, each with its own containers (the `!f2py threadsafe` contract: calls on the SAME container do not
// overlap); a further thread keeps the container tables busy.  The range checks "take" 100 us, so that waits overlap.
std::atomic<bool> g_stop{false};

void stepping_thread(int id, int rounds) {
    const int device = id % 2;
    stub_set_current_device(device);
    std::vector<Member> m(5);
    for (int i = 0; i < 5; ++i) {
        EXPECT(spd_modelstate_init_on(&m[i].state, (id + i) % 2) == 0);
        make_controls(m[i]);
        set_seed(m[i], 10000.0 * (id + 1) + i);
        init(m[i]);
    }
    std::vector<Member *> all;
    for (auto &x : m) all.push_back(&x);
    for (int r = 0; r < rounds; ++r) {
        step_all(all);
        {  // pipelined steps
            Lists l(all);
            int64_t pending = 0, next = 0;
            EXPECT(spd_parallel_step_begin(l.s.data(), l.c.data(), 5, &pending) == 0);
            for (int k = 0; k < 3; ++k) {
                EXPECT(spd_parallel_step_begin(l.s.data(), l.c.data(), 5, &next) == 0);
                EXPECT(spd_parallel_step_end(pending, l.codes.data()) == 0);
                pending = next;
            }
            EXPECT(spd_parallel_step_end(pending, l.codes.data()) == 0);
            for (auto *x : all)
                for (int k = 0; k < 4; ++k) x->predict_step();
        }
        if (r % 3 == 1) {  // regroup: a subset on its own, then the rest catches up
            step_all({&m[0], &m[3]});
            step_all({&m[1], &m[2], &m[4]});
        }
        if (r % 4 == 2) {  // a container is replaced
            close(m[r % 5]);
            EXPECT(spd_modelstate_init_on(&m[r % 5].state, (id + r) % 2) == 0);
            make_controls(m[r % 5]);
            set_seed(m[r % 5], 77.0 + r);
            init(m[r % 5]);
            for (int k = 0; k < m[(r + 1) % 5].steps; ++k) step_all({&m[r % 5]});  // ... and brought to the others' date
        }
        if (r % 5 == 3) {
            int32_t code = 9;
            EXPECT(spd_step(m[1].state, m[1].control, &code) == 0 && code == 0);
            m[1].predict_step();
            step_all({&m[0], &m[2], &m[3], &m[4]});
        }
        verify(m[r % 5]);
        EXPECT(stub_current_device() == device);
    }
    for (auto &x : m) verify(x);
    for (auto &x : m) close(x);
}

void table_thread() {
    while (!g_stop.load()) {
        int64_t d0 = 0, d1 = 0, c = 0;
        EXPECT(spd_create_datetime(1990, 5, 5, 0, 0, &d0) == 0 && spd_create_datetime(1990, 6, 5, 0, 0, &d1) == 0);
        EXPECT(spd_controlparams_init(&c, d0, d1) == 0);
        int32_t now[5], midx;
        EXPECT(spd_controlparams_get_model_datetime(c, now, &midx) == 0 && now[0] == 1990);
        EXPECT(spd_controlparams_close(c) == 0 && spd_close_datetime(d0) == 0 && spd_close_datetime(d1) == 0);
        int32_t alive = 0;
        EXPECT(spd_driver_stats(0, &alive, nullptr) == 0 && alive >= 0);
        char name[32];
        int32_t dt, nd, shape[5], ro;
        EXPECT(spd_registry_entry(3, name, &dt, &nd, shape, &ro) > 0);
        EXPECT(spd_driver_trace(1) == 0);
        int32_t pairs[64];
        EXPECT(spd_driver_trace_read(pairs, 32) >= 0);
        EXPECT(spd_driver_trace(0) == 0);
        int64_t s = 0;
        EXPECT(spd_modelstate_init_on(&s, 1) == 0 && spd_modelstate_close(s) == 0);
    }
}

// a container is closed by another thread while its owner waits for the check of a step that includes it
void close_during_wait() {
    stub_set_check_delay_us(20000);
    std::vector<Member> m(4);
    for (int i = 0; i < 4; ++i) {
        EXPECT(spd_modelstate_init_on(&m[i].state, 0) == 0);
        make_controls(m[i]);
        set_seed(m[i], 31.0 + i);
        init(m[i]);
    }
    std::vector<Member *> all;
    for (auto &x : m) all.push_back(&x);
    step_all(all);
    Lists l(all);
    int64_t token = 0;
    EXPECT(spd_parallel_step_begin(l.s.data(), l.c.data(), 4, &token) == 0);
    std::atomic<bool> waiting{false};
    std::thread closer([&] {
        while (!waiting.load()) std::this_thread::yield();
        std::this_thread::sleep_for(std::chrono::milliseconds(5));  // (the owner is inside spd_parallel_step_end by now)
        EXPECT(spd_modelstate_close(m[3].state) == 0);
    });
    waiting = true;
    EXPECT(spd_parallel_step_end(token, l.codes.data()) == 0);
    for (int i = 0; i < 4; ++i) EXPECT(l.codes[i] == 0);
    closer.join();
    stub_set_check_delay_us(100);
    for (int i = 0; i < 3; ++i) m[i].predict_step();
    step_all({&m[0], &m[1], &m[2]});  // the three that are left regroup and go on
    for (int i = 0; i < 3; ++i) verify(m[i]);
    int32_t code = 0;
    EXPECT(spd_check(m[3].state, &code) < 0);
    for (int i = 0; i < 3; ++i) close(m[i]);
    EXPECT(spd_controlparams_close(m[3].control) == 0);
}

// The last pending step of a begin / end loop has no next step for its put-off check to ride in: spd_parallel_step_end has to
// send it out itself.  It does so under the library's lock, before it lets go of the lock for the wait -- otherwise that launch
// (work INSIDE the model: the stub aborts on two threads inside one model) would run beside whatever another host thread does to
// the same model under the lock.  Here a second thread writes a variable of the same container all through the wait.
void end_beside_set() {
    stub_set_check_delay_us(20000);
    Member m;
    EXPECT(spd_modelstate_init_on(&m.state, 0) == 0);
    make_controls(m);
    set_seed(m, 5.0);
    init(m);
    for (int round = 0; round < 6; ++round) {
        int64_t token = 0;
        int32_t code = 9;
        EXPECT(spd_parallel_step_begin(&m.state, &m.control, 1, &token) == 0);
        std::atomic<bool> waiting{false}, done{false};
        std::thread writer([&] {
            while (!waiting.load()) std::this_thread::yield();
            std::vector<double> field(NG, 1.0 + round);
            while (!done.load()) EXPECT(spd_set(m.state, "alb0", field.data(), NG * sizeof(double)) == 0);
        });
        waiting = true;
        EXPECT(spd_parallel_step_end(token, &code) == 0 && code == 0);
        done = true;
        writer.join();
        m.predict_step();
    }
    verify(m);
    stub_set_check_delay_us(100);
    close(m);
}

void threads() {
    stub_set_device_count(2);
    stub_set_check_delay_us(100);
    g_stop = false;
    std::thread tables(table_thread);
    std::vector<std::thread> workers;
    for (int t = 0; t < 4; ++t) workers.emplace_back(stepping_thread, t, 12);
    for (auto &w : workers) w.join();
    close_during_wait();
    end_beside_set();
    g_stop = true;
    tables.join();
    EXPECT(models_alive() == 0);
    std::printf("driver sanitize: threads ok\n");
}
}  // namespace

int main(int argc, char **argv) {
    const std::string what = argc > 1 ? argv[1] : "all";
    if (what == "single" || what == "all") single_thread();
    if (what == "threads" || what == "all") threads();
    std::printf("driver sanitize ok\n");
    return 0;
}
