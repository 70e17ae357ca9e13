// Shared helpers for libroreg_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "roreg_hip.h"

#define ROREG_G 60
#define ROREG_K 13
#define ROREG_F 32

namespace roreg {

void set_error(const char *fmt, ...);

// device-resident group tables (uploaded once by roreg_set_group_tables)
struct GroupTablesDev {
    int32_t *P;      // [60*60]  P[a*60+g]
    int32_t *Nei;    // [60*13]
    uint8_t *P8;     // [60*60]  same as P, one byte per entry (LDS friendly)
    uint8_t *P8t;    // [60*60]  transposed: P8t[h*60+g] = P[g*60+h]
    uint8_t *split;  // [4096]   bank-split form of P8 for the gathered correlation (des2r.hip, SplitTab): per-lane rows in LDS-slot numbers,
    uint8_t *split_t;//          lane -> group element, group element -> LDS slot; split_t: the same for P8t
    double *R;       // [60*9]   float64 rotations
    float *Rf;       // [60*9]   float32-rounded rotations (test/estimator.py:279 .astype(np.float32))
    bool ready;
};
const GroupTablesDev &group_tables();

// Layout of GroupTablesDev::split (bytes): Q[64 lanes][60] at 0 (lane l's row of the table, entries already LDS slot numbers; a lane without a
// group element repeats an active lane of its half), lane_elem[64] at 3840 (0xff = none), slot[60] at 3904.
constexpr int SPLIT_Q = 0, SPLIT_LANE = 3840, SPLIT_SLOT = 3904, SPLIT_BYTES = 4096;

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// one pair of the batched local-transform stage (mirrors roreg_lt_task, include/roreg_hip.h)
struct LtTask {
    const float *before0, *before1, *after0, *after1;   // the two clouds' FCGF-in / YOHO-out group features [*,32,60]
    const double *keys0, *keys1;                        // keypoints [*,3]
    const int64_t *matches;                             // [M,2] interleaved rows (cloud 0, cloud 1)
    const int64_t *sel;                                 // [n] rows of `matches` to evaluate, or null (= 0..n-1)
    int32_t n, pad_;
    int64_t off;                                        // first output row of this pair
    const float *coef0, *coef1;                         // irrep coefficients of after0 / after1 [*,32,60] (roreg_feat_coefs), or null
};
__device__ __forceinline__ void lt_rows(const LtTask &t, int i, size_t &r0, size_t &r1) {
    const int64_t m = t.sel ? t.sel[i] : (int64_t)i;
    r0 = (size_t)t.matches[2 * m]; r1 = (size_t)t.matches[2 * m + 1];
}
void launch_des2r_batch(const LtTask *tasks, int n_tasks, int max_n, int64_t *dr_all, bool irrep, bool feat_bf16, hipStream_t s);
bool des2r_tables_ready();

// Optional per-kernel timing (roreg_profile_enable): HIP events recorded on the launch stream around selected launches.
enum ProfSlot { PROF_MM_TILE = 0, PROF_RANSAC_SCORE = 1, PROF_DES2R = 2, PROF_FT_NONLIN = 3, PROF_SINKHORN = 4, PROF_TOPK = 5, PROF_N = 6 };
bool prof_on();
void prof_begin(int slot, hipStream_t s);
void prof_end(int slot, hipStream_t s);
struct ProfScope {
    int slot; hipStream_t s; bool on;
    ProfScope(int slot_, hipStream_t s_) : slot(slot_), s(s_), on(prof_on()) { if (on) prof_begin(slot, s); }
    ~ProfScope() { if (on) prof_end(slot, s); }
};

// Sinkhorn iterations without a materialised coupling matrix (csrc/ot_flash.hip): potentials (natural log) of every pair to u_out / v_out
size_t ot_flash_workspace_bytes(int n_seg, int max_m, int max_n);
int ot_flash_early_exit(int on);                       // 1 / 0 / < 0 = query -> previous setting
int ot_flash_iteration_stats(long long *iters_sum, long long *pairs, int reset, hipStream_t s);
int ot_flash_iterations(const float *src, const float *tgt, const int32_t *seg_src, const int32_t *seg_tgt, const float *consts, int n_seg,
                        int max_m, int max_n, int min_n, bool coop_wanted, float alpha, int iters, float *u_out, float *v_out, size_t uv_stride, void *ws, hipStream_t s);

// the matcher's 1x1 layers as float32 fmaf chains on the matrix cores (csrc/linear_chain.hip; bitwise the vector-pipe kernels): true = shape served
bool linear_chain(const float *x, int L, int Cin, const float *W, const float *b, int Cout, float *y, hipStream_t s);
// round 6: the same chains software-pipelined (lc2_kernel); mlp_head_chain = first convolution + residual branch of mlp_2layer in one launch + the per-pair
// InstanceNorm statistics of h from the tiles' float64 channel sums (cat3_* non-null: the value MLP's assembled [m k, 96] rows)
// one-time self-check (first use; synchronises `s` once per process): false + roreg_last_error() if the matrix cores do not evaluate the fmaf chain
bool mfma_chain_verified(hipStream_t s);
bool linear_chain2_on();
void linear_chain2_set(bool on);
bool linear_chain2(const float *x, int L, int Cin, const float *W, const float *b, int Cout, float *y, hipStream_t s);
bool linear_tail_chain2(const float *h, int L, int Cmid, const float *mean_rstd, const float *W2, const float *b2, float *y, const int *seg_off, int n_seg,
                        int mult, hipStream_t s);
bool mlp_head_chain(const float *x, const float *cat3_table, const int64_t *cat3_idx, const float *cat3_conf, int cat3_k, int L, int Cin, const float *W1,
                    const float *b1, int C1, const float *Wr, const float *br, float *h, float *y, const int *seg_off, int n_seg, int mult, float eps,
                    float *mean_rstd, double *part, hipStream_t s);
bool linear_chain_cat3(const float *x, const float *table, const int64_t *idx, const float *conf, int m, int k, const float *W, const float *b, int Cout,
                       float *y, hipStream_t s);
bool linear_tail_chain(const float *h, int L, int Cmid, const float *mean_rstd, const float *W2, const float *b2, float *y, const int *seg_off, int n_seg,
                       int mult, hipStream_t s);
// the matcher's 1x1 layers on the matrix cores (csrc/linear_mfma.hip): true = shape served, launch issued
bool linear_mfma(const float *x, int L, int Cin, const float *W, const float *b, int Cout, float *y, hipStream_t s);
bool linear_tail_mfma(const float *h, int L, int Cmid, const float *mean_rstd, const float *W2, const float *b2, float *y, const int *seg_off, int n_seg,
                      int mult, hipStream_t s);

#define ROREG_CHECK_LAUNCH(name)                                                     \
    do {                                                                             \
        hipError_t e__ = hipGetLastError();                                          \
        if (e__ != hipSuccess) {                                                     \
            roreg::set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return 1;                                                                \
        }                                                                            \
    } while (0)

#define ROREG_REQUIRE(cond, ...)         \
    do {                                 \
        if (!(cond)) {                   \
            roreg::set_error(__VA_ARGS__); \
            return 2;                    \
        }                                \
    } while (0)

}  // namespace roreg
