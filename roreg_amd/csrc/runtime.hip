// Library state: last-error text and the device copies of the icosahedral group tables.
#include "common.h"
#include <stdarg.h>
#include <mutex>

namespace roreg {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static GroupTablesDev g_tables = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, false};
static std::mutex g_tables_mu;

const GroupTablesDev &group_tables() { return g_tables; }

}  // namespace roreg

extern "C" int roreg_abi_version(void) { return 1; }

extern "C" const char *roreg_last_error(void) { return roreg::g_err; }

extern "C" int roreg_set_group_tables(const int32_t *P_host, const int32_t *Nei_host, const double *R_host) {
    using namespace roreg;
    std::lock_guard<std::mutex> lk(g_tables_mu);
    ROREG_REQUIRE(P_host && Nei_host && R_host, "roreg_set_group_tables: null table");
    for (int i = 0; i < 3600; ++i) ROREG_REQUIRE(P_host[i] >= 0 && P_host[i] < 60, "roreg_set_group_tables: P out of range");
    for (int i = 0; i < 780; ++i) ROREG_REQUIRE(Nei_host[i] >= 0 && Nei_host[i] < 60, "roreg_set_group_tables: Nei out of range");
    GroupTablesDev &t = g_tables;
#define RT_CK(x)                                                                           \
    do {                                                                                   \
        hipError_t e = (x);                                                                \
        if (e != hipSuccess) {                                                             \
            set_error("roreg_set_group_tables: %s: %s", #x, hipGetErrorString(e));         \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)
    if (!t.P) {
        RT_CK(hipMalloc(&t.P, 3600 * sizeof(int32_t)));
        RT_CK(hipMalloc(&t.Nei, 780 * sizeof(int32_t)));
        RT_CK(hipMalloc(&t.P8, 3600));
        RT_CK(hipMalloc(&t.P8t, 3600));
        RT_CK(hipMalloc(&t.R, 540 * sizeof(double)));
        RT_CK(hipMalloc(&t.Rf, 540 * sizeof(float)));
    }
    uint8_t p8[3600], p8t[3600];
    float rf[540];
    for (int i = 0; i < 3600; ++i) p8[i] = (uint8_t)P_host[i];
    for (int a = 0; a < 60; ++a)
        for (int g = 0; g < 60; ++g) p8t[g * 60 + a] = (uint8_t)P_host[a * 60 + g];
    for (int i = 0; i < 540; ++i) rf[i] = (float)R_host[i];
    RT_CK(hipMemcpy(t.P, P_host, 3600 * sizeof(int32_t), hipMemcpyHostToDevice));
    RT_CK(hipMemcpy(t.Nei, Nei_host, 780 * sizeof(int32_t), hipMemcpyHostToDevice));
    RT_CK(hipMemcpy(t.P8, p8, 3600, hipMemcpyHostToDevice));
    RT_CK(hipMemcpy(t.P8t, p8t, 3600, hipMemcpyHostToDevice));
    RT_CK(hipMemcpy(t.R, R_host, 540 * sizeof(double), hipMemcpyHostToDevice));
    RT_CK(hipMemcpy(t.Rf, rf, 540 * sizeof(float), hipMemcpyHostToDevice));
#undef RT_CK
    t.ready = true;
    return 0;
}
