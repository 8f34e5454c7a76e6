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

thread_local hipEvent_t ofx_tl_stop_event = nullptr;

namespace {
struct Rec {
    int name_id;
    hipEvent_t a, b;
    double flops;
};
std::mutex g_mu;
int g_on = 0;                          // 0 off, 1 per kernel family, 2 per layer (family:tag)
thread_local const char* g_tag = nullptr;
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
    std::string key(name);
    if (g_on == 2 && g_tag) key.append(":").append(g_tag);
    auto it = g_name_ids.find(key);
    if (it == g_name_ids.end()) {
        id = (int)g_names.size();
        g_names.push_back(key);
        g_name_ids[key] = id;
    } else {
        id = it->second;
    }
    Rec r;
    r.name_id = id;
    r.flops = 0.0;
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

void OfxProfScope::flops(double f) {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_mu);
    g_recs[slot].flops = f;
}

void ofx_prof_set_tag(const char* tag) { g_tag = tag; }

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
    g_on = on < 0 ? 0 : (on > 2 ? 2 : on);
    return 0;
}

int ofx_prof_collect(char* json_out, size_t cap) {
    OFX_REQUIRE(json_out && cap > 2, OFX_EINVAL);
    OFX_HIP_CHECK(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lk(g_mu);
    std::vector<double> ms(g_names.size(), 0.0);
    std::vector<long> calls(g_names.size(), 0);
    std::vector<double> flops(g_names.size(), 0.0);
    for (auto& r : g_recs) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) {
            ms[r.name_id] += t;
            calls[r.name_id] += 1;
            flops[r.name_id] += r.flops;
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
        snprintf(buf, sizeof buf, "%s\"%s\": {\"calls\": %ld, \"ms\": %.6f, \"flops\": %.6e}", first ? "" : ", ",
                 g_names[i].c_str(), calls[i], ms[i], flops[i]);
        s += buf;
        first = false;
    }
    s += "}";
    if (s.size() + 1 > cap) return OFX_ENOMEM;
    memcpy(json_out, s.c_str(), s.size() + 1);
    return 0;
}

}  // extern "C"
