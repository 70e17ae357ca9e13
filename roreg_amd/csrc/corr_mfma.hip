// The 60 x 60 group cross-correlation as ONE small matrix product per point + 60 coset sums.
//
//   cor[b, a] = sum_f sum_g A[f, T[a, g]] B[f, g]           (A, B: the two points' [32, 60] group features; T = P or P^T)
// is the matcher's R_indicator feature (network/rot_coh_match.py:154-163).  The literal kernel (des2r_kernel) gathers A through the
// permutation for every multiply-add: 115,200 four-byte LDS reads per point, LDS-bound at 1.15 ms per 80 k points (0.13 of HBM peak, 7 % of
// BASELINE configs[3]'s path).  But with C[p, g] = sum_f A[f, p] B[f, g] -- a [60 x 32] . [32 x 60] product --
//   cor[b, a] = sum_g C[T[a, g], g]:
// the permutation only addresses the 60 x 60 result.  Here one wavefront per point forms C with 64 v_mfma_f32_32x32x2_f32 (float32 operands
// straight from memory: an MFMA operand of k step j is row f = 2 j + lane / 32 of the feature matrix, 128 contiguous bytes per half-wave;
// no split, no conversion), parks it in LDS and lets lane a add its 60 entries C[T[a, g], g], g ascending (for fixed g the 60 lanes hit 60
// different rows: conflict-free with the pitch of 65).  4096 matrix-pipe cycles and 15.4 KB of input per point: HBM-bound.
// Same function, another summation order than the literal kernel (float32 throughout, ~1e-6 of |A||B| apart): used where the correlation
// is a FEATURE (the stacked matcher); the arg-max contracts (Des2R) keep the literal / bounded kernels.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int G = 60, F = 32, PITCH = 65;

__global__ __launch_bounds__(256) void group_corr_mfma_kernel(const float *__restrict__ perm_feats, const int64_t *__restrict__ perm_rows,
                                                              const float *__restrict__ bcast_feats, const int64_t *__restrict__ bcast_rows,
                                                              const uint8_t *__restrict__ T8, int M, int per_wave, float *__restrict__ cor_out) {
    __shared__ float Cs[4][64 * PITCH];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j = lane & 31, kh = lane >> 5;
    // lane a's row of the table, once: T[a][g], g = 0..59
    uint32_t trow[G / 4];                                                     // four table entries per register (rows of T are 60 bytes: 4-byte aligned)
    {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(T8 + (lane < G ? lane : G - 1) * G);
#pragma unroll
        for (int q = 0; q < G / 4; ++q) trow[q] = src[q];
    }
    float *cs = Cs[w];
    const bool hi_ok = 32 + j < G;
    for (int t = 0; t < per_wave; ++t) {
        const int b = (blockIdx.x * 4 + w) * per_wave + t;
        if (b >= M) break;
        const float *A = perm_feats + (size_t)(perm_rows ? perm_rows[b] : (int64_t)b) * (F * G);
        const float *B = bcast_feats + (size_t)(bcast_rows ? bcast_rows[b] : (int64_t)b) * (F * G);
        float a0[16], a1[16], b0[16], b1[16];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const int f = 2 * ks + kh;
            a0[ks] = A[f * G + j]; b0[ks] = B[f * G + j];
            a1[ks] = hi_ok ? A[f * G + 32 + j] : 0.f; b1[ks] = hi_ok ? B[f * G + 32 + j] : 0.f;
        }
        f32x16 c00 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, c01 = c00, c10 = c00, c11 = c00;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[ks], b0[ks], c00, 0, 0, 0);
            c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[ks], b1[ks], c01, 0, 0, 0);
            c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[ks], b0[ks], c10, 0, 0, 0);
            c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[ks], b1[ks], c11, 0, 0, 0);
        }
        // accumulator register r of lane l: row 8 (r / 4) + 4 (l / 32) + r % 4, column l % 32 of its 32 x 32 tile
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int p = 8 * (r >> 2) + 4 * kh + (r & 3);
            cs[p * PITCH + j] = c00[r]; cs[p * PITCH + 32 + j] = c01[r];
            cs[(32 + p) * PITCH + j] = c10[r]; cs[(32 + p) * PITCH + 32 + j] = c11[r];
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < G; ++g) s += cs[(int)((trow[g >> 2] >> (8 * (g & 3))) & 0xffu) * PITCH + g];
        if (lane < G) cor_out[(size_t)b * G + lane] = s;
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    // the next point overwrites this wave's C
    }
}

}  // namespace

extern "C" int roreg_group_corr_mfma(const float *perm_feats, const int64_t *perm_rows, const float *bcast_feats, const int64_t *bcast_rows, int M,
                                     int transpose_table, float *cor_out, void *stream) {
    if (M == 0) return 0;
    ROREG_REQUIRE(perm_feats && bcast_feats && cor_out && M > 0, "roreg_group_corr_mfma: bad arguments");
    ROREG_REQUIRE(roreg::group_tables().ready, "roreg_group_corr_mfma: group tables not set");
    int per_wave = M / (4 * 4096);                     // >= ~4096 workgroups before a wave takes a second point (its table row is loaded once)
    if (per_wave < 1) per_wave = 1;
    if (per_wave > 8) per_wave = 8;
    const int wgs = (M + 4 * per_wave - 1) / (4 * per_wave);
    hipLaunchKernelGGL(group_corr_mfma_kernel, dim3(wgs), dim3(256), 0, roreg::as_stream(stream), perm_feats, perm_rows, bcast_feats, bcast_rows,
                       transpose_table ? roreg::group_tables().P8t : roreg::group_tables().P8, M, per_wave, cor_out);
    ROREG_CHECK_LAUNCH("roreg_group_corr_mfma");
    return 0;
}
