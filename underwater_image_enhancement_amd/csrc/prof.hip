// Optional per-kernel timing with HIP events recorded on the caller's stream (used by bench.py for the roofline
// figure).  Disabled by default: a ProfScope is a no-op unless uwie_profile_enable(ctx, 1) was called.
#include <map>
#include <string>
#include <vector>

#include "common.h"

namespace uwie {

struct Profiler {
    bool on = false;
    std::string only;  // when not empty: record launches of this kernel name only (two events per launch perturb the run)
    struct Rec {
        const char *name;
        hipEvent_t a, b;
    };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    size_t used = 0;
    struct Row {
        std::string name;
        double ms;
        int calls;
    };
    std::vector<Row> rows;

    hipEvent_t take()
    {
        if (used == pool.size()) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return nullptr;
            pool.push_back(e);
        }
        return pool[used++];
    }
};

static thread_local Profiler *g_current = nullptr;

void prof_bind(Profiler *p) { g_current = (p && p->on) ? p : nullptr; }

ProfScope::ProfScope(const char *name, hipStream_t st) : rec_(-1), st_(st)
{
    Profiler *p = g_current;
    if (!p) return;
    if (!p->only.empty() && p->only != name) return;
    hipEvent_t a = p->take(), b = p->take();
    if (!a || !b) return;
    (void)hipEventRecord(a, st);
    rec_ = (int)p->recs.size();
    p->recs.push_back({name, a, b});
}

ProfScope::~ProfScope()
{
    Profiler *p = g_current;
    if (!p || rec_ < 0) return;
    (void)hipEventRecord(p->recs[rec_].b, st_);
}

Profiler *prof_create() { return new Profiler(); }

void prof_destroy(Profiler *p)
{
    if (!p) return;
    for (hipEvent_t e : p->pool) (void)hipEventDestroy(e);
    delete p;
}

void prof_filter(Profiler *p, const char *name) { p->only = name ? name : ""; }

void prof_enable(Profiler *p, bool on)
{
    p->on = on;
    p->recs.clear();
    p->rows.clear();
    p->used = 0;
}

// Synchronises the device, folds the recorded intervals by kernel name, resets the recorder.
int prof_collect(Profiler *p)
{
    UWIE_HIP_CHECK(hipDeviceSynchronize());
    std::map<std::string, std::pair<double, int>> acc;
    std::vector<std::string> order;
    for (const auto &r : p->recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
        auto it = acc.find(r.name);
        if (it == acc.end()) {
            acc[r.name] = {ms, 1};
            order.push_back(r.name);
        } else {
            it->second.first += ms;
            it->second.second += 1;
        }
    }
    p->rows.clear();
    for (const auto &n : order) p->rows.push_back({n, acc[n].first, acc[n].second});
    p->recs.clear();
    p->used = 0;
    return (int)p->rows.size();
}

int prof_row(Profiler *p, int i, const char **name, double *ms, int *calls)
{
    if (i < 0 || i >= (int)p->rows.size()) return UWIE_E_INVALID;
    *name = p->rows[i].name.c_str();
    *ms = p->rows[i].ms;
    *calls = p->rows[i].calls;
    return UWIE_OK;
}

}  // namespace uwie
