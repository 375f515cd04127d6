// Measurement hooks shared by the inference (vitseg_api.hip) and training (vitseg_train.hip) drivers: while enabled,
// a ProfScope brackets the launches issued during its lifetime with a pair of hipEvents on the launch stream and
// records the algorithmic work of those launches (see include/vitseg.h, vitseg_profile_*).
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

namespace vitseg {

struct ProfRec {
    int kind;
    double work;
    hipEvent_t e0, e1;
};

struct Profiler {
    bool on = false;
    std::vector<ProfRec> recs;
    std::vector<hipEvent_t> pool;
    hipEvent_t get() {
        if (!pool.empty()) {
            hipEvent_t e = pool.back();
            pool.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
    void clear() {
        for (auto& r : recs) {
            pool.push_back(r.e0);
            pool.push_back(r.e1);
        }
        recs.clear();
    }
};

Profiler& profiler();  // process-global instance (vitseg_api.hip)

struct ProfScope {
    hipStream_t st;
    bool active;
    ProfRec r;
    ProfScope(int kind, double work, hipStream_t s) : st(s), active(profiler().on) {
        if (!active) return;
        Profiler& g = profiler();
        r.kind = kind;
        r.work = work;
        r.e0 = g.get();
        r.e1 = g.get();
        (void)hipEventRecord(r.e0, st);
    }
    ~ProfScope() {
        if (!active) return;
        (void)hipEventRecord(r.e1, st);
        profiler().recs.push_back(r);
    }
};

}  // namespace vitseg
