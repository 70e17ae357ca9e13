// Library state: last-error text and the device copies of the icosahedral group tables.
#include "common.h"
#include <cstdio>
#include <thread>
#include <vector>
#include <utility>
#include <stdarg.h>
#include <mutex>
#include <utility>
#include <vector>

namespace roreg {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static GroupTablesDev g_tables = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, false};
static std::mutex g_tables_mu;

const GroupTablesDev &group_tables() { return g_tables; }

// ---- optional kernel timing ---------------------------------------------------------------------------------------------------------
static bool g_prof = false;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_ev[PROF_N];
static hipEvent_t g_prof_open[PROF_N];
bool prof_on() { return g_prof; }
void prof_begin(int slot, hipStream_t s) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, s);
    g_prof_open[slot] = e;
}
void prof_end(int slot, hipStream_t s) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, s);
    g_prof_ev[slot].push_back({g_prof_open[slot], e});
}

}  // namespace roreg

extern "C" int roreg_profile_enable(int on) {
    using namespace roreg;
    for (int k = 0; k < PROF_N; ++k) {
        for (auto &pr : g_prof_ev[k]) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
        g_prof_ev[k].clear();
    }
    g_prof = on != 0;
    return 0;
}

extern "C" int roreg_profile_read(int slot, double *total_ms, int *launches) {
    using namespace roreg;
    ROREG_REQUIRE(slot >= 0 && slot < PROF_N && total_ms && launches, "roreg_profile_read: bad arguments");
    double t = 0.0;
    for (auto &pr : g_prof_ev[slot]) {
        float ms = 0.f;
        if (hipEventSynchronize(pr.second) != hipSuccess || hipEventElapsedTime(&ms, pr.first, pr.second) != hipSuccess) {
            set_error("roreg_profile_read: event query failed");
            return 1;
        }
        t += ms;
    }
    *total_ms = t; *launches = (int)g_prof_ev[slot].size();
    return 0;
}

extern "C" int roreg_abi_version(void) { return ROREG_ABI_VERSION; }

extern "C" const char *roreg_last_error(void) { return roreg::g_err; }

// Bank-split form of a 60 x 60 table Pm (lane a of the gathered correlation reads element Pm[a][g] of a 60-float row, g = 0 .. 59).
// A half-wave of 32 lanes reads 30 (or 32) DIFFERENT elements per step, and with the row stored as it comes two of them always share one of
// the 32 LDS banks (elements j and j + 32): every access takes two passes (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.48 measured).
// Pm is a translation table of the rotation group: x -> x[Pm[a][.]] commutes with the multiplications from the other side.  For an
// involution t of the group (a 180-degree rotation; A5 has fifteen) let nu(a) = Pm[t][a]: then Pm[nu(a)][g] = mu(Pm[a][g]) for EVERY g with
// one fixed-point-free involution mu of the row positions.  So: the two elements of a mu-orbit get LDS slots k and 32 + k (one bank), the
// two group elements of a nu-orbit get lanes l and 32 + l (different half-waves) -- whatever g is, a half-wave reads exactly one element of
// every mu-orbit: 30 different banks, no conflict.  A table without such a symmetry (any valid table is accepted) gets the plain
// assignment; the result never depends on the assignment, only the conflict rate does.
static void bank_split_table(const uint8_t *Pm, uint8_t *out) {
    using namespace roreg;
    uint8_t lane_elem[64], slot[60];
    int nu[60], mu[60];
    bool found = false;
    for (int t = 0; t < 60 && !found; ++t) {
        bool ok = true;
        for (int a = 0; a < 60 && ok; ++a) { nu[a] = Pm[t * 60 + a]; }
        for (int a = 0; a < 60 && ok; ++a) ok = nu[a] != a && nu[nu[a]] == a;
        if (!ok) continue;
        for (int j = 0; j < 60; ++j) mu[j] = -1;
        for (int a = 0; a < 60; ++a) mu[Pm[a * 60]] = Pm[nu[a] * 60];
        for (int j = 0; j < 60 && ok; ++j) ok = mu[j] >= 0 && mu[j] != j && mu[mu[j]] == j;
        for (int g = 0; g < 60 && ok; ++g)
            for (int a = 0; a < 60 && ok; ++a) ok = mu[Pm[a * 60 + g]] == Pm[nu[a] * 60 + g];
        found = ok;
    }
    for (int l = 0; l < 64; ++l) lane_elem[l] = 0xff;
    if (found) {
        int l = 0, k = 0;
        bool seen[60] = {false};
        for (int a = 0; a < 60; ++a)
            if (!seen[a]) { seen[a] = seen[nu[a]] = true; lane_elem[l] = (uint8_t)a; lane_elem[32 + l] = (uint8_t)nu[a]; ++l; }
        bool placed[60] = {false};
        for (int j = 0; j < 60; ++j)
            if (!placed[j]) { placed[j] = placed[mu[j]] = true; slot[j] = (uint8_t)k; slot[mu[j]] = (uint8_t)(32 + k); ++k; }
    } else {
        for (int a = 0; a < 60; ++a) { lane_elem[a] = (uint8_t)a; slot[a] = (uint8_t)a; }
    }
    for (int l = 0; l < 64; ++l) {
        int src = l;
        if (lane_elem[src] == 0xff) src = l & 32;                 // (lanes 0 and 32 own an element under either assignment)
        const int a = lane_elem[src];
        for (int g = 0; g < 60; ++g) out[SPLIT_Q + l * 60 + g] = slot[Pm[a * 60 + g]];
    }
    for (int l = 0; l < 64; ++l) out[SPLIT_LANE + l] = lane_elem[l];
    for (int j = 0; j < 60; ++j) out[SPLIT_SLOT + j] = slot[j];
    for (int i = SPLIT_SLOT + 60; i < SPLIT_BYTES; ++i) out[i] = 0;
}

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
        RT_CK(hipMalloc(&t.split, SPLIT_BYTES));
        RT_CK(hipMalloc(&t.split_t, SPLIT_BYTES));
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
    uint8_t sp[SPLIT_BYTES];
    bank_split_table(p8, sp);
    RT_CK(hipMemcpy(t.split, sp, SPLIT_BYTES, hipMemcpyHostToDevice));
    bank_split_table(p8t, sp);
    RT_CK(hipMemcpy(t.split_t, sp, SPLIT_BYTES, hipMemcpyHostToDevice));
    RT_CK(hipMemcpy(t.R, R_host, 540 * sizeof(double), hipMemcpyHostToDevice));
    RT_CK(hipMemcpy(t.Rf, rf, 540 * sizeof(float), hipMemcpyHostToDevice));
#undef RT_CK
    t.ready = true;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// YOHO-C hypothesis draws (host code; test/estimator.py:220-230).  The reference's loop calls, per try,
//   np.random.choice(range(60), p=prob)      -> one legacy double (two MT19937 words: (a>>5, b>>6)), cdf.searchsorted(u, 'right')
//   np.random.choice(bin_members, 3)         -> randint(0, n, 3): three masked-rejection draws, one 32-bit word per attempt
// and its control flow depends on nothing but the generator, so the whole loop is replayed here over a block of raw generator words
// (the caller draws them from the same global generator and afterwards advances it by exactly *words_used).
extern "C" int roreg_yohoc_draw(const uint32_t *words, long long n_words, const double *cdf, const int32_t *bin_size, int max_iter,
                                int max_tries, int32_t *bin_out, int64_t *pick_out, int32_t *n_hyp_out, long long *words_used) {
    ROREG_REQUIRE(words && cdf && bin_size && bin_out && pick_out && n_hyp_out && words_used && max_iter >= 0 && n_words >= 0,
                  "roreg_yohoc_draw: bad arguments");
    long long pos = 0;
    int n_hyp = 0, tries = 0;
    while (n_hyp < max_iter) {
        if (tries > max_tries) break;                             // `if exec_time > max_time: break` before the increment
        ++tries;
        if (pos + 2 > n_words) return 3;                          // block exhausted: the caller retries with a larger one
        const uint32_t a = words[pos] >> 5, b = words[pos + 1] >> 6;
        pos += 2;
        const double u = (a * 67108864.0 + b) / 9007199254740992.0;
        int lo = 0, hi = 60;                                      // first index with cdf[i] > u   (searchsorted side='right')
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
        }
        ROREG_REQUIRE(lo < 60, "roreg_yohoc_draw: cdf does not end at 1");
        const int n = bin_size[lo];
        if (n < 2) continue;
        const uint32_t rng = (uint32_t)(n - 1);
        uint32_t mask = rng;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        for (int t = 0; t < 3; ++t) {
            uint32_t v;
            do {
                if (pos >= n_words) return 3;
                v = words[pos++] & mask;
            } while (v > rng);
            pick_out[(size_t)n_hyp * 3 + t] = v;
        }
        bin_out[n_hyp++] = lo;
    }
    *n_hyp_out = n_hyp;
    *words_used = pos;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Seeded shuffles of the matcher's keypoint sampling and the one-shot estimator's hypothesis order (HOST code; test/matcher.py:83-88,
// test/estimator.py:423-425): per job `np.random.seed(seed)` followed by, for each of its `per_job` lists, `idx = np.arange(n);
// np.random.shuffle(idx); idx[:take]`.  numpy's legacy generator is MT19937 seeded by init_genrand(seed); its shuffle of a 1-d array is
// Fisher-Yates from the top -- for i = n-1 .. 1: j = random_interval(i), swap(x[i], x[j]) -- and random_interval(max) draws 32-bit words
// masked to the smallest 2^k - 1 >= max until one is <= max.  The same calls in C, jobs spread over host threads: the Python loop cost
// ~35 us per pair with the GPU idle behind a synchronisation (7 ms per scene).
namespace {
struct Mt19937 {
    uint32_t key[624];
    int pos;
    explicit Mt19937(uint32_t seed) {
        for (int i = 0; i < 624; ++i) { key[i] = seed; seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)i + 1u; }
        pos = 624;
    }
    Mt19937(const uint32_t *state, int position) : pos(position) { for (int i = 0; i < 624; ++i) key[i] = state[i]; }
    void refill() {
        const uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MATRIX_A = 0x9908b0dfu;
        int i = 0;
        for (; i < 624 - 397; ++i) { const uint32_t y = (key[i] & UPPER) | (key[i + 1] & LOWER); key[i] = key[i + 397] ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u); }
        for (; i < 623; ++i) { const uint32_t y = (key[i] & UPPER) | (key[i + 1] & LOWER); key[i] = key[i + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u); }
        const uint32_t y = (key[623] & UPPER) | (key[0] & LOWER);
        key[623] = key[396] ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u);
        pos = 0;
    }
    uint32_t next() {
        if (pos == 624) refill();
        uint32_t y = key[pos++];
        y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
        return y;
    }
};
}  // namespace

extern "C" int roreg_mt_shuffle_prefix(const uint32_t *seeds, int n_jobs, const int32_t *sizes, int per_job, int take, int64_t *out, int n_threads) {
    if (n_jobs == 0) return 0;
    ROREG_REQUIRE(seeds && sizes && out && n_jobs > 0 && per_job > 0 && take >= 0, "roreg_mt_shuffle_prefix: bad arguments");
    for (long long i = 0; i < (long long)n_jobs * per_job; ++i)
        ROREG_REQUIRE(sizes[i] >= 0, "roreg_mt_shuffle_prefix: negative list size");
    auto work = [&](int j0, int j1) {
        std::vector<int64_t> x;
        for (int job = j0; job < j1; ++job) {
            Mt19937 g(seeds[job]);
            for (int s = 0; s < per_job; ++s) {
                const int n = sizes[(size_t)job * per_job + s];
                x.resize((size_t)n);
                for (int i = 0; i < n; ++i) x[i] = i;
                for (int i = n - 1; i >= 1; --i) {
                    uint32_t mask = (uint32_t)i;
                    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
                    uint32_t v;
                    do { v = g.next() & mask; } while (v > (uint32_t)i);
                    std::swap(x[i], x[v]);
                }
                int64_t *dst = out + ((size_t)job * per_job + s) * take;
                for (int i = 0; i < take; ++i) dst[i] = i < n ? x[i] : -1;
            }
        }
    };
    int nt = n_threads < 1 ? 1 : (n_threads > n_jobs ? n_jobs : n_threads);
    if (nt == 1) { work(0, n_jobs); return 0; }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back(work, (int)((long long)n_jobs * t / nt), (int)((long long)n_jobs * (t + 1) / nt));
    for (auto &t : th) t.join();
    return 0;
}

// v6: the same shuffles drawn from ONE running stream -- the process-global generator an unseeded Test.py consumes (test/matcher.py:83-88,
// test/estimator.py:423-425).  key[624] / *pos are np.random.get_state()'s key and position on entry and the stream's state after the last list
// on return (for np.random.set_state): `for n in sizes: idx = np.arange(n); np.random.shuffle(idx); idx[:take]`.  One thread: list i + 1 starts
// where list i stopped.  449 pairs x 2 lists of 5000 cost ~40 ms in numpy on the launching thread, 13 ms of it with the GPU idle.
extern "C" int roreg_mt_stream_shuffle_prefix(uint32_t *key, int32_t *pos, const int32_t *sizes, int n_lists, int take, int64_t *out) {
    if (n_lists == 0) return 0;
    ROREG_REQUIRE(key && pos && sizes && out && n_lists > 0 && take >= 0 && *pos >= 0 && *pos <= 624, "roreg_mt_stream_shuffle_prefix: bad arguments");
    for (int i = 0; i < n_lists; ++i)
        ROREG_REQUIRE(sizes[i] >= 0, "roreg_mt_stream_shuffle_prefix: negative list size");
    Mt19937 g(key, *pos);
    std::vector<int64_t> x;
    for (int s = 0; s < n_lists; ++s) {
        const int n = sizes[s];
        x.resize((size_t)n);
        for (int i = 0; i < n; ++i) x[i] = i;
        for (int i = n - 1; i >= 1; --i) {
            uint32_t mask = (uint32_t)i;
            mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
            uint32_t v;
            do { v = g.next() & mask; } while (v > (uint32_t)i);
            std::swap(x[i], x[v]);
        }
        int64_t *dst = out + (size_t)s * take;
        for (int i = 0; i < take; ++i) dst[i] = i < n ? x[i] : -1;
    }
    for (int i = 0; i < 624; ++i) key[i] = g.key[i];
    *pos = g.pos;
    return 0;
}

// v6: the YOHO-C draws of MANY pairs, each from a generator stream of its own -- per pair exactly what
//     rng = np.random.RandomState(seed); yohoc_draws(anchors, max_iter, rng)          (roreg_amd/test/estimator.py; test/estimator.py:119-137,214-230)
// does: the rotation-bin statistic (counts, num = counts / 100, prob = num (num - 0.01) (num - 0.02) for bins of two or more, normalised by numpy's
// pairwise float64 sum), the sampling loop of roreg_yohoc_draw over the stream's raw words, and the rows of the pair's correspondence list
// (members of a bin in increasing order).  A pair with no bin of two correspondences gives up like the reference (n_hyp = -1) and gets the 16
// doubles of `rng.rand(4, 4)` from its untouched stream.  Host code, pairs spread over threads: the per-pair Python loop cost ~0.6 ms per pair
// on the thread that feeds the GPU (270 ms per 449-pair scene).
static double np_pairwise_sum60(const double *a) {                  // numpy's float64 add.reduce over 60 contiguous values (pairwise_sum, n < 128)
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i = 8;
    for (; i < 56; i += 8)
        for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < 60; ++i) res += a[i];
    return res;
}

extern "C" int roreg_yohoc_draw_many(const uint32_t *seeds, int n_pairs, const int64_t *anchors, const int64_t *offsets, int max_iter, int max_tries,
                                     int64_t *rows_out, int32_t *n_hyp_out, double *giveup_out, int n_threads) {
    if (n_pairs == 0) return 0;
    ROREG_REQUIRE(seeds && anchors && offsets && rows_out && n_hyp_out && giveup_out && n_pairs > 0 && max_iter >= 0, "roreg_yohoc_draw_many: bad arguments");
    for (int p = 0; p < n_pairs; ++p) ROREG_REQUIRE(offsets[p + 1] >= offsets[p], "roreg_yohoc_draw_many: offsets must not decrease");
    for (long long i = offsets[0]; i < offsets[n_pairs]; ++i) ROREG_REQUIRE(anchors[i] >= 0 && anchors[i] < 60, "roreg_yohoc_draw_many: anchor out of range");
    auto work = [&](int p0, int p1) {
        std::vector<int64_t> members;
        for (int p = p0; p < p1; ++p) {
            const int64_t *an = anchors + offsets[p];
            const long long n = offsets[p + 1] - offsets[p];
            Mt19937 g(seeds[p]);
            auto next_double = [&]() { const uint32_t a = g.next() >> 5, b = g.next() >> 6; return (a * 67108864.0 + b) / 9007199254740992.0; };
            long long counts[60] = {0}, starts[60];
            for (long long i = 0; i < n; ++i) ++counts[an[i]];
            double prob[60];
            for (int r = 0; r < 60; ++r) {
                const double num = (double)counts[r] / 100.0;
                prob[r] = counts[r] < 2 ? 0.0 : num * (num - 0.01) * (num - 0.02);
            }
            const double total = np_pairwise_sum60(prob);
            if (total == 0.0) for (int r = 0; r < 60; ++r) prob[r] = 0.0;
            else for (int r = 0; r < 60; ++r) prob[r] = prob[r] / total;
            if (np_pairwise_sum60(prob) < 1e-5) {                      // no rotation bin with two correspondences: the reference's random answer
                n_hyp_out[p] = -1;
                for (int q = 0; q < 16; ++q) giveup_out[(size_t)p * 16 + q] = next_double();
                continue;
            }
            double cdf[60], run = 0.0;
            for (int r = 0; r < 60; ++r) { run += prob[r]; cdf[r] = run; }
            const double last = cdf[59];
            for (int r = 0; r < 60; ++r) cdf[r] /= last;
            long long acc = 0;
            for (int r = 0; r < 60; ++r) { starts[r] = acc; acc += counts[r]; }
            members.resize((size_t)n);
            { long long fill[60]; for (int r = 0; r < 60; ++r) fill[r] = starts[r];
              for (long long i = 0; i < n; ++i) members[(size_t)fill[an[i]]++] = i; }      // stable: increasing order inside a bin
            int n_hyp = 0, tries = 0;
            int64_t *rows = rows_out + (size_t)p * max_iter * 3;
            while (n_hyp < max_iter) {
                if (tries > max_tries) break;
                ++tries;
                const double u = next_double();
                int lo = 0, hi = 60;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (cdf[mid] <= u) lo = mid + 1; else hi = mid; }
                if (lo >= 60) lo = 59;                                 // (cdf[59] == 1 > u always; defensive)
                const long long nb = counts[lo];
                if (nb < 2) continue;
                const uint32_t rng = (uint32_t)(nb - 1);
                uint32_t mask = rng;
                mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
                for (int t = 0; t < 3; ++t) {
                    uint32_t v;
                    do { v = g.next() & mask; } while (v > rng);
                    rows[(size_t)n_hyp * 3 + t] = members[(size_t)(starts[lo] + v)];
                }
                ++n_hyp;
            }
            n_hyp_out[p] = n_hyp;
        }
    };
    int nt = n_threads < 1 ? 1 : (n_threads > n_pairs ? n_pairs : n_threads);
    if (nt == 1) { work(0, n_pairs); return 0; }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back(work, (int)((long long)n_pairs * t / nt), (int)((long long)n_pairs * (t + 1) / nt));
    for (auto &t : th) t.join();
    return 0;
}

// v6 (host function): write n files, file q = header[q] (header_len[q] bytes: the .npy header the caller formatted) followed by data[q]
// (nbytes[q] bytes), on n_threads -- the engine's StageFileWriter leaves ~1800 small .npy files per 449-pair scene, and Python's np.save
// spends ~80 us per file holding the interpreter lock the launching thread needs.  Returns 0, or 1 + the index of the first file that failed.
extern "C" int roreg_write_files(const char *const *paths, const void *const *headers, const int32_t *header_len, const void *const *data,
                                 const int64_t *nbytes, int n, int n_threads) {
    if (n == 0) return 0;
    ROREG_REQUIRE(paths && headers && header_len && data && nbytes && n > 0, "roreg_write_files: bad arguments");
    std::vector<int> failed((size_t)(n_threads < 1 ? 1 : n_threads), -1);
    auto work = [&](int t, int q0, int q1) {
        for (int q = q0; q < q1; ++q) {
            FILE *f = fopen(paths[q], "wb");
            bool ok = f != nullptr;
            if (ok && header_len[q] > 0) ok = fwrite(headers[q], 1, (size_t)header_len[q], f) == (size_t)header_len[q];
            if (ok && nbytes[q] > 0) ok = fwrite(data[q], 1, (size_t)nbytes[q], f) == (size_t)nbytes[q];
            if (f) ok = (fclose(f) == 0) && ok;
            if (!ok && failed[t] < 0) failed[t] = q;
        }
    };
    const int nt = n_threads < 1 ? 1 : (n_threads > n ? n : n_threads);
    if (nt == 1) work(0, 0, n);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) th.emplace_back(work, t, (int)((long long)n * t / nt), (int)((long long)n * (t + 1) / nt));
        for (auto &t : th) t.join();
    }
    for (int t = 0; t < nt; ++t)
        if (failed[t] >= 0) { roreg::set_error("roreg_write_files: could not write %s", paths[failed[t]]); return 1 + failed[t]; }
    return 0;
}
