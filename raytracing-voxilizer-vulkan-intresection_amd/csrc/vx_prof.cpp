// vx_prof.cpp -- optional per-kernel timing with HIP events on the launch stream (off by default).
// bench.py turns it on to obtain the dominant kernel's average launch duration for the roofline figure; the numbers
// must agree with `rocprofv3 --kernel-trace --stats` of the same run.
#include "vx_internal.h"

#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace vx {
namespace {
struct Pending { const char* name; hipEvent_t a, b; };
struct Acc { double ms = 0.0; uint64_t n = 0; };
std::mutex g_mu;
bool g_on = false;
std::string g_only;  // when set, only launches of this kernel are timed (two events per launch are not free)
std::vector<Pending> g_pending;
std::vector<hipEvent_t> g_free;
std::map<std::string, Acc> g_acc;

hipEvent_t get_event()
{
    if (!g_free.empty()) { hipEvent_t e = g_free.back(); g_free.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

void collect_locked()
{
    for (Pending& p : g_pending) {
        float ms = 0.f;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            Acc& a = g_acc[p.name];
            a.ms += ms;
            a.n += 1;
        }
        g_free.push_back(p.a);
        g_free.push_back(p.b);
    }
    g_pending.clear();
}
}  // namespace

void prof_enable(bool on) { std::lock_guard<std::mutex> lk(g_mu); g_on = on; }
bool prof_enabled() { return g_on; }
void prof_select(const char* name) { std::lock_guard<std::mutex> lk(g_mu); g_only = name ? name : ""; }
void prof_reset() { std::lock_guard<std::mutex> lk(g_mu); collect_locked(); g_acc.clear(); }

ProfScope::ProfScope(const char* name, hipStream_t s) : name_(name), s_(s), a_(nullptr)
{
    if (!g_on) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_only.empty()) {
        // launched names look like "k_scan_sums" or "(k_trace<L, SEG>)": compare the bare kernel name
        const char* q = name;
        while (*q == '(') ++q;
        size_t n = 0;
        while (q[n] && q[n] != '<' && q[n] != ')') ++n;
        if (g_only.size() != n || g_only.compare(0, n, q, n) != 0) return;
    }
    a_ = get_event();
    if (a_) (void)hipEventRecord(a_, s_);
}
ProfScope::~ProfScope()
{
    if (!a_) return;
    std::lock_guard<std::mutex> lk(g_mu);
    hipEvent_t b = get_event();
    if (!b) { g_free.push_back(a_); return; }
    (void)hipEventRecord(b, s_);
    g_pending.push_back(Pending{name_, a_, b});
}

int prof_read(int slot, char* name, size_t cap, double* ms, uint64_t* n)
{
    std::lock_guard<std::mutex> lk(g_mu);
    collect_locked();
    if (slot < 0 || (size_t)slot >= g_acc.size()) return -1;
    auto it = g_acc.begin();
    std::advance(it, slot);
    if (name && cap) { snprintf(name, cap, "%s", it->first.c_str()); }
    if (ms) *ms = it->second.ms;
    if (n) *n = it->second.n;
    return 0;
}
}  // namespace vx
