// Library-wide state of the mvfusion C ABI: version and the thread-local error
// string.  There is no other global state: every buffer belongs to the caller.
#include "mvf_common.h"

namespace mvf {

static thread_local char g_error[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

// ---- optional per-stage timing with HIP events ------------------------------
struct Profile {
    bool on = false;
    int n = 0;                     // events recorded by the last profiled call
    hipEvent_t ev[MVF_PROFILE_MAX_STAGES + 1] = {};
    const char *name[MVF_PROFILE_MAX_STAGES + 1] = {};
    bool created = false;
};
static thread_local Profile g_prof;

void prof_begin(hipStream_t st)
{
    Profile &p = g_prof;
    if (!p.on) return;
    if (!p.created) {
        for (auto &e : p.ev) (void)hipEventCreate(&e);
        p.created = true;
    }
    p.n = 0;
    p.name[0] = "begin";
    (void)hipEventRecord(p.ev[0], st);
    p.n = 1;
}

void prof_mark(const char *name, hipStream_t st)
{
    Profile &p = g_prof;
    if (!p.on || p.n == 0 || p.n > MVF_PROFILE_MAX_STAGES) return;
    p.name[p.n] = name;
    (void)hipEventRecord(p.ev[p.n], st);
    ++p.n;
}

}  // namespace mvf

extern "C" int mvf_profile_enable(int enable)
{
    mvf::g_prof.on = enable != 0;
    mvf::g_prof.n = 0;
    return MVF_OK;
}

extern "C" int mvf_profile_read(float *h_ms, const char **h_names, int capacity)
{
    mvf::Profile &p = mvf::g_prof;
    if (!p.on || p.n < 2) return 0;
    int stages = p.n - 1;
    if (stages > capacity) stages = capacity;
    for (int i = 0; i < stages; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.ev[i], p.ev[i + 1]) != hipSuccess) {
            mvf::set_error("profile_read: events not complete (synchronise the stream first)");
            return MVF_E_LAUNCH;
        }
        h_ms[i] = ms;
        if (h_names) h_names[i] = p.name[i + 1];
    }
    return stages;
}

extern "C" int mvf_version(void) { return MVF_VERSION; }
extern "C" const char *mvf_last_error(void) { return mvf::g_error; }
