// Library identification, error strings and the live per-family timing used by bench.py's roofline
// line: HIP events recorded on the launch stream around each entry point (off by default).
#include <deque>
#include <mutex>

#include "common.h"

namespace endo {

struct ProfSlot {
    hipEvent_t start, stop;
    int family;
    double flops, bytes;
};

static std::mutex g_prof_mutex;
static std::deque<ProfSlot> g_prof_slots;
static unsigned g_prof_mask = 0;   // bit f set = time family f
static int g_prof_period = 1;      // time one launch in g_prof_period of a family (endo_prof_sample)
static int64_t g_prof_seen[ENDO_PROF_FAMILIES] = {};      // launches of an enabled family since endo_prof_enable, timed or not

ProfScope::ProfScope(int family_, hipStream_t stream_, double flops, double bytes) : family(family_), stream(stream_), slot(nullptr) {
    if (!((g_prof_mask >> family_) & 1u)) return;
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    if (family_ >= 0 && family_ < ENDO_PROF_FAMILIES && (g_prof_seen[family_]++ % g_prof_period) != 0) return;
    ProfSlot s;
    s.family = family_;
    s.flops = flops;
    s.bytes = bytes;
    if (hipEventCreate(&s.start) != hipSuccess) return;
    if (hipEventCreate(&s.stop) != hipSuccess) { (void)hipEventDestroy(s.start); return; }
    (void)hipEventRecord(s.start, stream_);
    g_prof_slots.push_back(s);
    slot = &g_prof_slots.back();
}

ProfScope::~ProfScope() {
    if (slot) (void)hipEventRecord(static_cast<ProfSlot*>(slot)->stop, stream);
}

static void prof_clear() {
    for (auto& s : g_prof_slots) {
        (void)hipEventDestroy(s.start);
        (void)hipEventDestroy(s.stop);
    }
    g_prof_slots.clear();
}

static const char* const kFamilyNames[ENDO_PROF_FAMILIES] = {
    "conv3x3_dense_fwd", "conv3x3_up_fwd", "conv1x1_pool_fwd", "conv_first_fwd", "conv_final",
    "dgrad_dense", "wgrad_dense", "dgrad_other", "wgrad_other", "small", "geometry", "loss",
    "optimizer", "reserved13", "reserved14", "reserved15"};

}  // namespace endo

using namespace endo;

extern "C" int endo_abi_version(void) { return ENDO_ABI_VERSION; }

extern "C" const char* endo_error_string(int code) {
    if (code == 0) return "ok";
    if (code == ENDO_E_BADARG) return "endo: bad argument (null pointer or non-positive size)";
    if (code == ENDO_E_UNSUPPORTED) return "endo: unsupported shape or format (network: H and W must be multiples of 32; JPEG: sequential Huffman, 8 bit, grey / 4:4:4 / 4:2:2 / 4:2:0)";
    if (code > 0) return hipGetErrorString(static_cast<hipError_t>(code));
    return "endo: unknown error";
}

extern "C" int endo_prof_enable(int family_mask) {
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    prof_clear();
    for (auto& n : g_prof_seen) n = 0;
    g_prof_mask = static_cast<unsigned>(family_mask);
    return 0;
}

extern "C" int endo_prof_sample(int period) {
    if (period < 1) return ENDO_E_BADARG;
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    g_prof_period = period;
    return 0;
}

extern "C" int endo_prof_seen(int family, int64_t* launches) {
    if (family < 0 || family >= ENDO_PROF_FAMILIES || !launches) return ENDO_E_BADARG;
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    *launches = g_prof_seen[family];
    return 0;
}

extern "C" int endo_prof_read(int family, double* total_ms, int64_t* launches, double* total_flops, double* total_bytes) {
    if (family < 0 || family >= ENDO_PROF_FAMILIES || !total_ms || !launches || !total_flops || !total_bytes) return ENDO_E_BADARG;
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    double ms = 0.0, fl = 0.0, by = 0.0;
    int64_t cnt = 0;
    for (auto& s : g_prof_slots) {
        if (s.family != family) continue;
        ENDO_CHECK(hipEventSynchronize(s.stop));
        float t = 0.f;
        ENDO_CHECK(hipEventElapsedTime(&t, s.start, s.stop));
        ms += t;
        fl += s.flops;
        by += s.bytes;
        ++cnt;
    }
    *total_ms = ms;
    *launches = cnt;
    *total_flops = fl;
    *total_bytes = by;
    return 0;
}

extern "C" const char* endo_prof_family_name(int family) {
    if (family < 0 || family >= ENDO_PROF_FAMILIES) return "";
    return kFamilyNames[family];
}
