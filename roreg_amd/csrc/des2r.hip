// Per-correspondence 60x60 local-rotation cross-correlation and its argmax (coarse rotation index).
//
//   cor[a] = sum_f ( sum_g d1[f, P[a,g]] * d2[f, g] ),   a* = first argmax_a cor[a]
//
// One wavefront per correspondence: lane a (< 60) owns cor[a].  d1 (the side that is permuted) is staged in
// LDS because each lane walks it through its own row of the permutation table; d2 is read by every lane at
// the same address, so it stays in registers (lane g holds column g) and is broadcast with v_readlane.
// The accumulation order is the contract (s_f over g in order, then over f in order; fp32, no FMA
// contraction) so the index is bit-identical to the oracle's on the same input.
// Reference: extractor_dr_index.Batch_Des2R_torch, test/estimator.py:85-89 (gathers at :108-110).
#include "common.h"

// Bit-exactness contract: no fused multiply-add may be formed from separate * and + in this file (hipcc's
// default is -ffp-contract=fast, and the __f*_rn helpers are plain operators); sqrtf and / are correctly
// rounded under hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt.
#pragma clang fp contract(off)

namespace {

// one wavefront = one correspondence: feats1 row r1 is the permuted side, feats0 row r0 the broadcast side; result row b
__device__ __forceinline__ void des2r_body(const float *__restrict__ feats1, size_t r1, const float *__restrict__ feats0, size_t r0,
                                           bool live, const uint8_t *__restrict__ P8, size_t b, int64_t *__restrict__ idx_out,
                                           float *__restrict__ cor_out) {
    __shared__ float d1s[4][ROREG_F * ROREG_G];
    __shared__ uint8_t Pl[ROREG_G * ROREG_G];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < ROREG_G * ROREG_G; i += 256) Pl[i] = P8[i];
    {
        const float4 *src = reinterpret_cast<const float4 *>(feats1 + r1 * (ROREG_F * ROREG_G));
        float4 *dst = reinterpret_cast<float4 *>(d1s[w]);
        for (int i = lane; i < ROREG_F * ROREG_G / 4; i += 64) dst[i] = src[i];
    }
    const bool act = lane < ROREG_G;
    float d2[ROREG_F];
    {
        const float *src = feats0 + r0 * (ROREG_F * ROREG_G);
#pragma unroll
        for (int f = 0; f < ROREG_F; ++f) d2[f] = act ? src[f * ROREG_G + lane] : 0.f;
    }
    __syncthreads();
    if (!live) return;

    float s[ROREG_F];
#pragma unroll
    for (int f = 0; f < ROREG_F; ++f) s[f] = 0.f;
    const uint8_t *prow = Pl + (act ? lane : 0) * ROREG_G;
    for (int g = 0; g < ROREG_G; ++g) {
        const int pg = prow[g];
#pragma unroll
        for (int f = 0; f < ROREG_F; ++f) {
            const float bcast = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, d2[f]), g));
            s[f] = __fadd_rn(s[f], __fmul_rn(d1s[w][f * ROREG_G + pg], bcast));
        }
    }
    float cor = 0.f;
#pragma unroll
    for (int f = 0; f < ROREG_F; ++f) cor = __fadd_rn(cor, s[f]);
    if (cor_out && act) cor_out[b * ROREG_G + lane] = cor;
    // first argmax over lanes 0..59: order by (value desc, lane asc)
    float bv = act ? cor : -__builtin_inff();
    int bi = act ? lane : 0x7fffffff;
    if (bv != bv) bv = -__builtin_inff();      // NaN never wins (torch.argmax would pick it; inputs are finite)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o);
        const int oi = __shfl_xor(bi, o);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0 && idx_out) idx_out[b] = bi;
}

__global__ __launch_bounds__(256) void des2r_kernel(const float *__restrict__ feats1, const int64_t *__restrict__ rows1,
                                                    const float *__restrict__ feats0, const int64_t *__restrict__ rows0,
                                                    const uint8_t *__restrict__ P8, int M, int64_t *__restrict__ idx_out,
                                                    float *__restrict__ cor_out) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool live = b < M;
    const size_t r1 = live ? (rows1 ? (size_t)rows1[b] : (size_t)b) : 0;
    const size_t r0 = live ? (rows0 ? (size_t)rows0[b] : (size_t)b) : 0;
    des2r_body(feats1, r1, feats0, r0, live, P8, (size_t)b, idx_out, cor_out);
}

// all pairs of a scene in one launch: blockIdx.y = pair, cloud 1 is the permuted side (test/estimator.py:108-110)
__global__ __launch_bounds__(256) void des2r_batch_kernel(const roreg::LtTask *__restrict__ tasks, const uint8_t *__restrict__ P8,
                                                          int64_t *__restrict__ dr_all) {
    const roreg::LtTask t = tasks[blockIdx.y];
    if ((int)blockIdx.x * 4 >= t.n) return;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool live = i < t.n;
    size_t r0 = 0, r1 = 0;
    if (live) roreg::lt_rows(t, i, r0, r1);
    des2r_body(t.after1, r1, t.after0, r0, live, P8, (size_t)(t.off + i), dr_all, nullptr);
}

}  // namespace

void roreg::launch_des2r_batch(const LtTask *tasks, int n_tasks, int max_n, int64_t *dr_all, hipStream_t s) {
    roreg::ProfScope prof(roreg::PROF_DES2R, s);
    hipLaunchKernelGGL(des2r_batch_kernel, dim3((max_n + 3) / 4, n_tasks), dim3(256), 0, s, tasks, roreg::group_tables().P8, dr_all);
}

extern "C" int roreg_des2r(const float *feats1, const int64_t *rows1, const float *feats0, const int64_t *rows0, int M,
                           int64_t *idx_out, float *cor_out, void *stream) {
    return roreg_group_corr(feats1, rows1, feats0, rows0, M, 0, idx_out, cor_out, stream);
}

extern "C" int roreg_group_corr(const float *perm_feats, const int64_t *perm_rows, const float *bcast_feats, const int64_t *bcast_rows,
                                int M, int transpose_table, int64_t *idx_out, float *cor_out, void *stream) {
    if (M == 0) return 0;
    ROREG_REQUIRE(perm_feats && bcast_feats && (idx_out || cor_out) && M > 0, "roreg_group_corr: bad arguments");
    ROREG_REQUIRE(roreg::group_tables().ready, "roreg_group_corr: group tables not set");
    if (M == 0) return 0;
    hipLaunchKernelGGL(des2r_kernel, dim3((M + 3) / 4), dim3(256), 0, roreg::as_stream(stream), perm_feats, perm_rows, bcast_feats,
                       bcast_rows, transpose_table ? roreg::group_tables().P8t : roreg::group_tables().P8, M, idx_out, cor_out);
    ROREG_CHECK_LAUNCH("roreg_group_corr");
    return 0;
}
