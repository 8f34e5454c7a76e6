// Per-kernel HIP-event profiler + misc C-ABI entry points of libofx.so.
//
// bench.py needs the average duration of the dominant kernel measured *inside* the timed region on
// the stream the kernels are launched on; every launcher in this library brackets its launch with an
// OfxProfScope, which records a pair of hipEvents on that stream when profiling is enabled.
#include "ofx_internal.h"

#include <map>
#include <mutex>
#include <string>
#include <vector>
#include <cstdio>
#include <cstring>

namespace {
struct Rec {
    int name_id;
    hipEvent_t a, b;
};
std::mutex g_mu;
bool g_on = false;
std::vector<std::string> g_names;
std::map<std::string, int> g_name_ids;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;

hipEvent_t get_event() {
    if (!g_pool.empty()) {
        hipEvent_t e = g_pool.back();
        g_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
}  // namespace

OfxProfScope::OfxProfScope(const char* name, hipStream_t s) : slot(-1), stream(s) {
    if (!g_on) return;
    std::lock_guard<std::mutex> lk(g_mu);
    int id;
    auto it = g_name_ids.find(name);
    if (it == g_name_ids.end()) {
        id = (int)g_names.size();
        g_names.push_back(name);
        g_name_ids[name] = id;
    } else {
        id = it->second;
    }
    Rec r;
    r.name_id = id;
    r.a = get_event();
    r.b = get_event();
    if (!r.a || !r.b) return;
    (void)hipEventRecord(r.a, stream);
    slot = (int)g_recs.size();
    g_recs.push_back(r);
}

OfxProfScope::~OfxProfScope() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_mu);
    (void)hipEventRecord(g_recs[slot].b, stream);
}

extern "C" {

int ofx_version(void) { return OFX_VERSION; }

const char* ofx_error_string(int code) {
    switch (code) {
        case 0: return "success";
        case OFX_EINVAL: return "ofx: invalid argument";
        case OFX_EALIGN: return "ofx: pointer or leading dimension not 16-byte aligned";
        case OFX_ENOMEM: return "ofx: workspace too small";
        case OFX_EKEY: return "ofx: weight tensor missing or wrong shape";
        case OFX_ENODEV: return "ofx: no gfx950 device";
        default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "ofx: unknown error";
}

int ofx_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_on = on != 0;
    return 0;
}

int ofx_prof_collect(char* json_out, size_t cap) {
    OFX_REQUIRE(json_out && cap > 2, OFX_EINVAL);
    OFX_HIP_CHECK(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lk(g_mu);
    std::vector<double> ms(g_names.size(), 0.0);
    std::vector<long> calls(g_names.size(), 0);
    for (auto& r : g_recs) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) {
            ms[r.name_id] += t;
            calls[r.name_id] += 1;
        }
        g_pool.push_back(r.a);
        g_pool.push_back(r.b);
    }
    g_recs.clear();
    std::string s = "{";
    bool first = true;
    for (size_t i = 0; i < g_names.size(); ++i) {
        if (!calls[i]) continue;
        char buf[256];
        snprintf(buf, sizeof buf, "%s\"%s\": {\"calls\": %ld, \"ms\": %.6f}", first ? "" : ", ",
                 g_names[i].c_str(), calls[i], ms[i]);
        s += buf;
        first = false;
    }
    s += "}";
    if (s.size() + 1 > cap) return OFX_ENOMEM;
    memcpy(json_out, s.c_str(), s.size() + 1);
    return 0;
}

}  // extern "C"
