// C-ABI plumbing of libim360_kernels.so: error reporting, version, per-kernel-class HIP-event
// profiling used by bench.py's roofline leg.  No torch types, no allocation of user data, no
// synchronisation except in im360_prof_collect (which exists to read event timings).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <mutex>
#include <vector>

#include "prof.h"

static thread_local char g_err[512] = "";

extern "C" __attribute__((visibility("hidden"))) void im360_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" __attribute__((visibility("default"))) const char* im360_last_error(void) { return g_err; }

// 2 (round 5): im360_conv_fwd / im360_linear_fwd take a trailing gn_partial pointer, im360_linear_ln_fwd's table rows include c2
// (round 4 changed both without bumping the number; callers built against version 1 must not load this library)
// 3 (round 6): im360_attn_fwd takes the block maps of its packed bias matrices (three trailing arguments)
extern "C" __attribute__((visibility("default"))) int im360_abi_version(void) { return 5; }

// bit 0: built with -DIM360_ABLATE (`make ablate`): the rejected A/B variants and the ablation kernels are in the library
extern "C" __attribute__((visibility("default"))) int im360_build_flags(void) {
#ifdef IM360_ABLATE
    return 1;
#else
    return 0;
#endif
}

// ------------------------------------------------------------------------------------------------
namespace {
struct Slot { hipEvent_t a, b; int kind; };
std::mutex g_mu;
std::vector<Slot> g_used, g_free;
unsigned g_mask = 0;
}  // namespace

extern "C" __attribute__((visibility("default"))) void im360_prof_enable(unsigned mask) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_mask = mask;
}

namespace im360 {
ProfScope::ProfScope(int kind, void* stream) : slot_(-1), stream_(stream) {
    if (!(g_mask & (1u << kind))) return;
    std::lock_guard<std::mutex> lk(g_mu);
    Slot s;
    if (!g_free.empty()) {
        s = g_free.back();
        g_free.pop_back();
    } else {
        if (hipEventCreate(&s.a) != hipSuccess || hipEventCreate(&s.b) != hipSuccess) return;
    }
    s.kind = kind;
    (void)hipEventRecord(s.a, (hipStream_t)stream);
    g_used.push_back(s);
    slot_ = (int)g_used.size() - 1;
}
ProfScope::~ProfScope() {
    if (slot_ < 0) return;
    std::lock_guard<std::mutex> lk(g_mu);
    (void)hipEventRecord(g_used[slot_].b, (hipStream_t)stream_);
}
}  // namespace im360

// Sums the elapsed time of every recorded launch of `kind` since the last collect; blocks on the events.
extern "C" __attribute__((visibility("default"))) int im360_prof_collect(int kind, double* total_ms, long* launches) {
    std::lock_guard<std::mutex> lk(g_mu);
    double tot = 0.0;
    long n = 0;
    std::vector<Slot> keep;
    for (auto& s : g_used) {
        if (s.kind != kind) {
            keep.push_back(s);
            continue;
        }
        float ms = 0.f;
        if (hipEventSynchronize(s.b) == hipSuccess && hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
            tot += ms;
            ++n;
        }
        g_free.push_back(s);
    }
    g_used.swap(keep);
    if (total_ms) *total_ms = tot;
    if (launches) *launches = n;
    return 0;
}

// ------------------------------------------------------------------------------------------------
namespace {
struct Knobs {
    std::atomic<int> v[im360::KNOB_COUNT];
    Knobs() {
        static const struct { int k; const char* env; int def; } init[] = {
            {im360::KNOB_ATTN_QB, "IM360_ATTN_QB", 0},       {im360::KNOB_CONV_BIG, "IM360_CONV_BIG", 1},
            {im360::KNOB_CONV_BK, "IM360_CONV_BK", 0},       {im360::KNOB_TATTN_SCALAR, "IM360_TATTN_SCALAR", 0},
            {im360::KNOB_CONV_RING, "IM360_CONV_RING", 1},    {im360::KNOB_CONV_HALO, "IM360_CONV_HALO", 0},
            {im360::KNOB_CONV_CM, "IM360_CONV_CM", 1},        {im360::KNOB_LN_PACKED, "IM360_LN_PACKED", 1},
            {im360::KNOB_RING_GROUPS, "IM360_RING_GROUPS", 0}, {im360::KNOB_ATTN_X, "IM360_ATTN_X", 3}, {im360::KNOB_ATTN_DS, "IM360_ATTN_DS", 0}, {im360::KNOB_ATTN_ONE, "IM360_ATTN_ONE", 1}, {im360::KNOB_ATTN_HG, "IM360_ATTN_HG", 0}, {im360::KNOB_CONV_SMALL, "IM360_CONV_SMALL", 2}, {im360::KNOB_ATTN_W3, "IM360_ATTN_W3", 1}, {im360::KNOB_ATTN_PIPE, "IM360_ATTN_PIPE", -1}, {im360::KNOB_CONV_STAG, "IM360_CONV_STAG", 0}, {im360::KNOB_CONV_PERSIST, "IM360_CONV_PERSIST", 0}, {im360::KNOB_GN_APPLY, "IM360_GN_APPLY", 2}, {im360::KNOB_TATTN_NT, "IM360_TATTN_NT", 0}, {im360::KNOB_NT, "IM360_NT", 1}, {im360::KNOB_G4, "IM360_G4", 0}, {im360::KNOB_GN_WGS, "IM360_GN_WGS", 0}, {im360::KNOB_CONV_KSPLIT, "IM360_CONV_KSPLIT", 1},
        };
        for (auto& x : v) x.store(0);
        for (auto& i : init) {
            const char* e = getenv(i.env);
            v[i.k].store(e ? atoi(e) : i.def);
        }
    }
};
Knobs g_knobs;
}  // namespace

int im360::knob(int k) { return (k >= 0 && k < im360::KNOB_COUNT) ? g_knobs.v[k].load(std::memory_order_relaxed) : 0; }

extern "C" __attribute__((visibility("default"))) int im360_tuning_set(int knob, int value) {
    if (knob < 0 || knob >= im360::KNOB_COUNT) {
        im360_set_error("tuning_set: unknown knob %d", knob);
        return -1;
    }
    g_knobs.v[knob].store(value);
    return 0;
}
