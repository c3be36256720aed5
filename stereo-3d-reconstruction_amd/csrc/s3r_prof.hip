// The event profiler of libs3r_hip.so: per-launch-group HIP events on the stream the kernels are launched on (bench.py's
// roofline, Stereo2Voxel.autotune).  Off unless s3r_profile_enable(n > 0) was called; host code only.
#include "s3r_host.h"

#include <atomic>
#include <mutex>

namespace s3rh {
namespace {

struct Prof {
    std::mutex mu;
    std::atomic<bool> on{false};
    int cap = 0;
    unsigned gen = 0;             // bumped by every enable / disable: a scope opened under an older pool skips its stop
    int device = -1;              // the device the event pool was created on
    std::vector<hipEvent_t> ev;   // 2 per record
    std::vector<s3r_prof_record> rec;
} g_prof;

thread_local int g_cur_tag = 0;      // tag of the innermost live F_MFMA scope on this thread: what its nested F_AUX passes carry
std::atomic<int> g_detail{0};        // s3r_profile_detail: 1 = the aux passes get records of their own

}  // namespace

ProfScope::ProfScope(hipStream_t s, int fam, int tag, double flops, double bytes) : family(fam), stream(s) {
    if (fam == F_MFMA) { prev_tag = g_cur_tag; g_cur_tag = tag; }
    if (!g_prof.on.load(std::memory_order_relaxed)) return;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    if (!g_prof.on.load(std::memory_order_relaxed) || dev != g_prof.device || (int)g_prof.rec.size() >= g_prof.cap) return;
    slot = (int)g_prof.rec.size();
    s3r_prof_record r;
    r.family = fam; r.tag = tag; r.ms = 0.f; r.flops = flops; r.bytes = bytes; r.launches = 1;
    r.exec_flops = flops; r.algo = 0; r.reserved = 0;
    g_prof.rec.push_back(r);
    gen = g_prof.gen;
    e0 = g_prof.ev[2 * slot];
    e1 = g_prof.ev[2 * slot + 1];
    active = hipEventRecord(e0, stream) == hipSuccess;
    if (!active) g_prof.rec.pop_back();
}

ProfScope::~ProfScope() {
    if (family == F_MFMA) g_cur_tag = prev_tag;
    if (!active) return;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    if (gen != g_prof.gen) return;                       // the pool this scope belongs to is gone
    (void)hipEventRecord(e1, stream);
    if (slot < (int)g_prof.rec.size()) {
        g_prof.rec[slot].launches = launches;
        g_prof.rec[slot].algo = algo;
        if (exec >= 0.0) g_prof.rec[slot].exec_flops = exec;
    }
}

}  // namespace s3rh

namespace s3r {

// the launchers' hook (s3r_kernels.h): one pass without matrix work, labelled with the layer being run
AuxScope::AuxScope(hipStream_t s, double bytes) : impl(nullptr) {
    // (only on request: an event pair between two kernels of a layer costs a few microseconds of queue time in an eager run,
    // which a profile taken for the LAYER's duration must not carry)
    if (s3rh::g_detail.load(std::memory_order_relaxed)) impl = new s3rh::ProfScope(s, s3rh::F_AUX, s3rh::g_cur_tag, 0.0, bytes);
}
AuxScope::~AuxScope() { delete static_cast<s3rh::ProfScope*>(impl); }

}  // namespace s3r

using namespace s3rh;

extern "C" {

int s3r_profile_enable(int max_records) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.on = false;
    ++g_prof.gen;                                            // live scopes of the old pool skip their stop record
    for (hipEvent_t ev : g_prof.ev) (void)hipEventDestroy(ev);
    g_prof.ev.clear();
    g_prof.rec.clear();
    g_prof.cap = 0;
    g_prof.device = -1;
    if (max_records <= 0) return S3R_OK;
    if (hipGetDevice(&g_prof.device) != hipSuccess) g_prof.device = -1;
    g_prof.ev.assign((size_t)2 * max_records, nullptr);
    for (auto& ev : g_prof.ev) {
        hipError_t e = hipEventCreate(&ev);
        if (e != hipSuccess) {
            for (hipEvent_t x : g_prof.ev) if (x) (void)hipEventDestroy(x);
            g_prof.ev.clear();
            return hip_fail(e, "hipEventCreate");
        }
    }
    g_prof.rec.reserve(max_records);
    g_prof.cap = max_records;
    g_prof.on = true;
    return S3R_OK;
}

int s3r_profile_detail(int level) {
    g_detail.store(level > 0 ? 1 : 0, std::memory_order_relaxed);
    return S3R_OK;
}

int s3r_profile_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.rec.clear();
    return S3R_OK;
}

int s3r_profile_read(s3r_prof_record* out, int max_records) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    const int n = (int)g_prof.rec.size() < max_records ? (int)g_prof.rec.size() : max_records;
    for (int i = 0; i < n; ++i) {
        hipError_t e = hipEventSynchronize(g_prof.ev[2 * i + 1]);
        if (e != hipSuccess) return hip_fail(e, "hipEventSynchronize");
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]);
        if (e != hipSuccess) return hip_fail(e, "hipEventElapsedTime");
        g_prof.rec[i].ms = ms;
        if (out) out[i] = g_prof.rec[i];
    }
    return n;
}

}  // extern "C"
