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

// The same correlation, the same operation order (every s_f its own chain over g in order, products and sums rounded separately, then the
// s_f in order), arranged for the LDS and the vector pipe (round 6; the matcher's R_indicator ran at 0.48 bank conflicts per LDS cycle with
// the LDS index pipe 96 % and the vector pipe 83 % busy, profiles/r06_rd_rm_k5000_pmc.txt):
//   * the permuted row sits in LDS in the bank-split order of runtime.hip's bank_split_table (pitch 64): whatever g is, the 30 lanes of a
//     half-wave read 30 different banks;
//   * lane l owns the group element lane_elem[l] (the two elements of a nu-orbit in different half-waves);
//   * two channels per vector instruction: s_{2k}, s_{2k+1} advance together through v_pk_mul_f32 / v_pk_add_f32 (IEEE per element: the
//     same roundings as the scalar form), their broadcast factors as an SGPR pair -- which bought nothing: a packed float32 instruction takes
//     two issue slots on this part.  Kept for A/B (ROREG_DES2R_SPLIT=1); the default is the third form below.
typedef float des2r_f2 __attribute__((ext_vector_type(2)));
constexpr int DES2R_SPITCH = 64;

__device__ __forceinline__ void des2r_split_body(const float *__restrict__ feats1, size_t r1, const float *__restrict__ feats0, size_t r0,
                                                 bool live, const uint8_t *__restrict__ split, size_t b, int64_t *__restrict__ idx_out,
                                                 float *__restrict__ cor_out) {
    __shared__ float d1s[4][ROREG_F * DES2R_SPITCH];
    __shared__ uint8_t Ql[64 * ROREG_G];
    __shared__ uint8_t slot_l[64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < 64 * ROREG_G / 4; i += 256)
        reinterpret_cast<uint32_t *>(Ql)[i] = reinterpret_cast<const uint32_t *>(split + roreg::SPLIT_Q)[i];
    if (tid < ROREG_G) slot_l[tid] = split[roreg::SPLIT_SLOT + tid];
    const int elem = split[roreg::SPLIT_LANE + lane];              // the group element this lane accumulates (0xff: none)
    const bool act = elem != 0xff;
    float d2[ROREG_F];
    {
        const float *src = feats0 + r0 * (ROREG_F * ROREG_G);
#pragma unroll
        for (int f = 0; f < ROREG_F; ++f) d2[f] = lane < ROREG_G ? src[f * ROREG_G + lane] : 0.f;     // lane g holds column g of the broadcast side
    }
    __syncthreads();                                               // slot_l
    {
        const float4 *src = reinterpret_cast<const float4 *>(feats1 + r1 * (ROREG_F * ROREG_G));
        float *dst = d1s[w];
        for (int i = lane; i < ROREG_F * ROREG_G / 4; i += 64) {
            const float4 v = src[i];
            const int f = (i * 4) / ROREG_G, j = (i * 4) % ROREG_G;           // (60 is a multiple of 4: a float4 never straddles two channels)
            float *row = dst + f * DES2R_SPITCH;
            row[slot_l[j]] = v.x; row[slot_l[j + 1]] = v.y; row[slot_l[j + 2]] = v.z; row[slot_l[j + 3]] = v.w;
        }
    }
    __syncthreads();
    if (!live) return;

    des2r_f2 s2[ROREG_F / 2];
#pragma unroll
    for (int k = 0; k < ROREG_F / 2; ++k) s2[k] = des2r_f2{0.f, 0.f};
    const uint8_t *qrow = Ql + lane * ROREG_G;
    const float *base = d1s[w];
    for (int g = 0; g < ROREG_G; ++g) {
        const float *col = base + qrow[g];
#pragma unroll
        for (int k = 0; k < ROREG_F / 2; ++k) {
            des2r_f2 bc;
            bc.x = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, d2[2 * k]), g));
            bc.y = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, d2[2 * k + 1]), g));
            des2r_f2 a;
            a.x = col[(2 * k) * DES2R_SPITCH];
            a.y = col[(2 * k + 1) * DES2R_SPITCH];
            const des2r_f2 prod = a * bc;                          // (contract off: a product, then a sum)
            s2[k] = s2[k] + prod;
        }
    }
    float cor = 0.f;
#pragma unroll
    for (int k = 0; k < ROREG_F / 2; ++k) { cor = __fadd_rn(cor, s2[k].x); cor = __fadd_rn(cor, s2[k].y); }
    if (cor_out && act) cor_out[b * ROREG_G + elem] = cor;
    // first argmax over the group elements: order by (value desc, element asc)
    float bv = act ? cor : -__builtin_inff();
    int bi = act ? elem : 0x7fffffff;
    if (bv != bv) bv = -__builtin_inff();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o);
        const int oi = __shfl_xor(bi, o);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0 && idx_out) idx_out[b] = bi;
}

// Third form: the broadcast factor comes through the DPP row_newbcast modifier of the multiply itself (lane C of every row of 16 lanes feeds
// the whole row) instead of a v_readlane into an SGPR: measured, the vector pipe was the bound of the split form (SQ_ACTIVE_INST_VALU 96 % of the
// kernel's cycles) and a v_readlane costs as much as a multiply.  Lane l keeps column 16 c + (l & 15) of the broadcast row in register
// d2r[c][.] (c = 0 .. 3), so every row of 16 holds all 60 columns; DES2R_NCH (8 | 16) channels per pass: 76 / 120 registers.
// Same chains, same roundings: s_f = (...((0 + p_0) + p_1)...) over g, then the s_f in order.
template <int C>
__device__ __forceinline__ float des2r_mul_row_bcast(float d1, float d2) {      // d1 * (d2 of lane C of this lane's row of 16), one instruction
    float p;                                                                    // (written out: left to the compiler, one product in five became a
    asm("v_mul_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"   //  v_mov_b32_dpp feeding a packed multiply -- three issue slots, not two)
        : "=v"(p) : "v"(d2), "v"(d1), "n"(C));
    return p;
}


template <int G0, int DES2R_NCH>
__device__ __forceinline__ void des2r_dpp_steps(const float *__restrict__ base, const uint8_t *__restrict__ qrow, const float (&d2r)[4][DES2R_NCH],
                                                float (&s)[DES2R_NCH]) {
    if constexpr (G0 < ROREG_G) {
        const float *col = base + qrow[G0];
#pragma unroll
        for (int k = 0; k < DES2R_NCH; ++k)
            s[k] = __fadd_rn(s[k], des2r_mul_row_bcast<G0 % 16>(col[k * DES2R_SPITCH], d2r[G0 / 16][k]));
        des2r_dpp_steps<G0 + 1, DES2R_NCH>(base, qrow, d2r, s);
    }
}

template <int DES2R_NCH>
__device__ __forceinline__ void des2r_dpp_body(const float *__restrict__ feats1, size_t r1, const float *__restrict__ feats0, size_t r0,
                                               bool live, const uint8_t *__restrict__ split, size_t b, int64_t *__restrict__ idx_out,
                                               float *__restrict__ cor_out) {
    __shared__ float d1s[4][ROREG_F * DES2R_SPITCH];
    __shared__ uint8_t Ql[64 * ROREG_G];
    __shared__ uint8_t slot_l[64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < 64 * ROREG_G / 4; i += 256)
        reinterpret_cast<uint32_t *>(Ql)[i] = reinterpret_cast<const uint32_t *>(split + roreg::SPLIT_Q)[i];
    if (tid < ROREG_G) slot_l[tid] = split[roreg::SPLIT_SLOT + tid];
    const int elem = split[roreg::SPLIT_LANE + lane];
    const bool act = elem != 0xff;
    __syncthreads();
    {
        const float4 *src = reinterpret_cast<const float4 *>(feats1 + r1 * (ROREG_F * ROREG_G));
        float *dst = d1s[w];
        for (int i = lane; i < ROREG_F * ROREG_G / 4; i += 64) {
            const float4 v = src[i];
            const int f = (i * 4) / ROREG_G, j = (i * 4) % ROREG_G;
            float *row = dst + f * DES2R_SPITCH;
            row[slot_l[j]] = v.x; row[slot_l[j + 1]] = v.y; row[slot_l[j + 2]] = v.z; row[slot_l[j + 3]] = v.w;
        }
    }
    __syncthreads();
    if (!live) return;
    const uint8_t *qrow = Ql + lane * ROREG_G;
    const float *src0 = feats0 + r0 * (ROREG_F * ROREG_G);
    const int c16 = lane & 15;
    float cor = 0.f;
    for (int fh = 0; fh < ROREG_F / DES2R_NCH; ++fh) {
        float d2r[4][DES2R_NCH];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int k = 0; k < DES2R_NCH; ++k)
                d2r[c][k] = (16 * c + c16 < ROREG_G) ? src0[(fh * DES2R_NCH + k) * ROREG_G + 16 * c + c16] : 0.f;
        float s[DES2R_NCH];
#pragma unroll
        for (int k = 0; k < DES2R_NCH; ++k) s[k] = 0.f;
        des2r_dpp_steps<0, DES2R_NCH>(d1s[w] + fh * DES2R_NCH * DES2R_SPITCH, qrow, d2r, s);
#pragma unroll
        for (int k = 0; k < DES2R_NCH; ++k) cor = __fadd_rn(cor, s[k]);
    }
    if (cor_out && act) cor_out[b * ROREG_G + elem] = cor;
    float bv = act ? cor : -__builtin_inff();
    int bi = act ? elem : 0x7fffffff;
    if (bv != bv) bv = -__builtin_inff();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o);
        const int oi = __shfl_xor(bi, o);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0 && idx_out) idx_out[b] = bi;
}

template <int DES2R_NCH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void des2r_dpp_kernel(const float *__restrict__ feats1, const int64_t *__restrict__ rows1,
                                                        const float *__restrict__ feats0, const int64_t *__restrict__ rows0,
                                                        const uint8_t *__restrict__ split, int M, int64_t *__restrict__ idx_out,
                                                        float *__restrict__ cor_out) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool live = b < M;
    const size_t r1 = live ? (rows1 ? (size_t)rows1[b] : (size_t)b) : 0;
    const size_t r0 = live ? (rows0 ? (size_t)rows0[b] : (size_t)b) : 0;
    des2r_dpp_body<DES2R_NCH>(feats1, r1, feats0, r0, live, split, (size_t)b, idx_out, cor_out);
}

__global__ __launch_bounds__(256) void des2r_split_kernel(const float *__restrict__ feats1, const int64_t *__restrict__ rows1,
                                                          const float *__restrict__ feats0, const int64_t *__restrict__ rows0,
                                                          const uint8_t *__restrict__ split, int M, int64_t *__restrict__ idx_out,
                                                          float *__restrict__ cor_out) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool live = b < M;
    const size_t r1 = live ? (rows1 ? (size_t)rows1[b] : (size_t)b) : 0;
    const size_t r0 = live ? (rows0 ? (size_t)rows0[b] : (size_t)b) : 0;
    des2r_split_body(feats1, r1, feats0, r0, live, split, (size_t)b, idx_out, cor_out);
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


// ---------------------------------------------------------------------------------------------------------------------------------------
// Des2R in the irrep domain: bound on the matrix of all 60 correlations from 244 multiply-adds per channel, exact re-evaluation of the
// candidates.
//
// x -> x[P[a, .]] is a translation on the group, so with the orthonormal transform of roreg_amd/fourier.py (coefficient matrices
// X(rho) [d x d], d = 1,3,3,4,5) the correlation is
//     cor[a] = sum_rho sum_ij rho(a)[j][i] * C_rho[i][j],     C_rho = sum_f X2_f(rho) . X1_f(rho)^T          (X1: permuted side)
// i.e. per channel sum_rho d^3 = 244 multiply-adds for C instead of 3600, plus ONE 60 x 60 product per correspondence -- 10x fewer
// operations, no permuted gather.  The per-keypoint coefficients are computed once per cloud (roreg_feat_coefs).  That value is an
// approximation of the float32 number the reference computes (different arithmetic), so it only BOUNDS: every a with
// cor~[a] >= max cor~ - margin is a candidate, margin = 1e-4 |d1| |d2| >= 2 (|cor~ - exact| + |literal f32 - exact|):
//   * the literal evaluation is two-level, 60 then 32 terms: error <= (93 eps) sum |terms| <= 5.6e-6 |d1||d2|;
//   * the irrep evaluation: its terms are rho(a)[j][i] C_rho[i][j] with sum over (rho,i,j) of |terms| <= || |rho(a)| ||_2 |d1||d2| and the
//     entrywise-absolute representation matrices have spectral norm <= sqrt(5) (largest irrep, d = 5), NOT 1 -- so the 160 + 60 chained
//     fmas err by <= 220 eps sqrt(5) |d1||d2| = 2.9e-5 |d1||d2|, plus the coefficient transform's own rounding (a few eps per coefficient
//     in f32 / bf16x3 mode, ~2^-22 relative in fp16x2 mode) <= 1e-6 |d1||d2|: together <= 3.0e-5 |d1||d2|;
//   2 (3.0e-5 + 5.6e-6) = 7.1e-5 < 1e-4 (round 2 used 6e-5 from an under-estimate of the first bound; the candidate rate rises from 0.7 %
//   to ~1.2 % of noise-level correspondences), so the literal first arg-max is always among the candidates.  One candidate: done.
// Several (near ties, duplicates -- a few per cent of random correspondences, none of the well-matched ones): the candidates are
// re-evaluated with the literal formula in the reference's order (des2r_body above, bit for bit) from the group-domain rows, and the first
// maximum of those wins.  Result: the index of the literal evaluation, always.
//
// One wave per correspondence, eight per workgroup; every workgroup walks DES2R_ITER correspondences per wave with the next rows
// prefetched into registers under the current one's arithmetic.
struct Des2rTabs {
    const uint8_t *ia, *ib;     // [60][5]: lane q = (rho,i,j): X2 index off + i*d + k, X1 index off + j*d + k  (k < cnt[q])
    const uint8_t *cnt;         // [60]
    const float *NT;            // [60 q][60 a] = rho(a)[j][i]
};
Des2rTabs g_tabs = {nullptr, nullptr, nullptr, nullptr};        // x -> x[P[a,.]]  (Des2R)
Des2rTabs g_tabs_t = {nullptr, nullptr, nullptr, nullptr};      // x -> x[P[.,a]]  (the matcher's R_indicator)
int32_t *g_recheck = nullptr;                                 // device counter of correspondences that took the exact path (diagnostics)

constexpr int DES2R_ITER = 4;
constexpr int DES2R_ROW = ROREG_F * ROREG_G;                  // 1920 floats per keypoint
constexpr int DES2R_PITCH = 64;                               // LDS row pitch: columns 60..63 stay zero (the target of padded product terms)
constexpr int DES2R_LROW = ROREG_F * DES2R_PITCH;             // floats per staged keypoint

template <typename FT> __device__ __forceinline__ float feat_ld(const FT *p, size_t i) { return (float)p[i]; }

struct Des2rRows { const float *c1, *c0; size_t r1, r0; bool live; };

// COR_ONLY: write the 60 (approximate) correlations instead of the exact arg-max (no candidate stage): the matcher's R_indicator feature.
template <typename FT, bool COR_ONLY, typename RowFn>
__device__ __forceinline__ void des2r_irrep_loop(RowFn rows_of, int n_iter, const void *feats1_v, const void *feats0_v, Des2rTabs tabs,
                                                 const uint8_t *__restrict__ P8, int64_t *__restrict__ idx_out, size_t out_base, int32_t *recheck,
                                                 float *__restrict__ cor_out = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *NTs = lds;                                         // [60][60]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    float *X1 = lds + ROREG_G * ROREG_G + w * (2 * DES2R_LROW), *X2 = X1 + DES2R_LROW;
    for (int i = tid; i < ROREG_G * ROREG_G; i += 512) NTs[i] = tabs.NT[i];
    if (lane < 32) {                                          // zero pads of this wave's rows (never overwritten)
        *reinterpret_cast<float4 *>(X1 + lane * DES2R_PITCH + ROREG_G) = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4 *>(X2 + lane * DES2R_PITCH + ROREG_G) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const bool act = lane < ROREG_G;
    const int q = act ? lane : 0;
    int ia[5], ib[5];                                         // terms beyond the irrep's dimension read the zero pad: branch-free loop
    const int dl = tabs.cnt[q];
#pragma unroll
    for (int k = 0; k < 5; ++k) { ia[k] = k < dl ? tabs.ia[q * 5 + k] : ROREG_G; ib[k] = k < dl ? tabs.ib[q * 5 + k] : ROREG_G; }
    // prefetch registers: 480 float4 per row = 7.5 per lane
    float4 p1[8], p0[8];
    auto fetch = [&](const Des2rRows &r) {
        const float4 *s1 = reinterpret_cast<const float4 *>(r.c1 + r.r1 * DES2R_ROW), *s0 = reinterpret_cast<const float4 *>(r.c0 + r.r0 * DES2R_ROW);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = lane + 64 * i;
            if (e < DES2R_ROW / 4) { p1[i] = s1[e]; p0[i] = s0[e]; }
        }
    };
    Des2rRows cur = rows_of(0, w);
    fetch(cur);
    __syncthreads();                                          // NTs is filled; from here on every wave works on its own LDS rows only
    auto wave_fence = [] { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); };
    for (int it = 0; it < n_iter; ++it) {
        // (no workgroup barrier in the loop: a wave's LDS operations execute in program order, and a correspondence that takes the
        // exact path then delays only its own wave)
        wave_fence();
        float n1 = 0.f, n0 = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = lane + 64 * i;
            if (e < DES2R_ROW / 4) {
                const int le = (e / 15) * (DES2R_PITCH / 4) + e % 15;          // 15 float4 per 60-float row -> pitch of 16
                reinterpret_cast<float4 *>(X1)[le] = p1[i]; reinterpret_cast<float4 *>(X2)[le] = p0[i];
                n1 += p1[i].x * p1[i].x + p1[i].y * p1[i].y + p1[i].z * p1[i].z + p1[i].w * p1[i].w;
                n0 += p0[i].x * p0[i].x + p0[i].y * p0[i].y + p0[i].z * p0[i].z + p0[i].w * p0[i].w;
            }
        }
        const Des2rRows me = cur;
        if (it + 1 < n_iter) { cur = rows_of(it + 1, w); fetch(cur); }          // in flight under this correspondence's arithmetic
        wave_fence();
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { n1 += __shfl_xor(n1, o); n0 += __shfl_xor(n0, o); }
        // ---- C[q] = sum_f sum_k X2[f][ia_k] * X1[f][ib_k] ----------------------------------------------------------------------------
        float c = 0.f;
#pragma unroll 8
        for (int f = 0; f < ROREG_F; ++f) {
            const float *x2 = X2 + f * DES2R_PITCH, *x1 = X1 + f * DES2R_PITCH;
#pragma unroll
            for (int k = 0; k < 5; ++k) c = fmaf(x2[ia[k]], x1[ib[k]], c);
        }
        if (!act) c = 0.f;
        // ---- cor~[a] = sum_q NT[q][a] * C[q] -----------------------------------------------------------------------------------------
        float cor = 0.f;
#pragma unroll
        for (int qq = 0; qq < ROREG_G; ++qq) {
            const float cq = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c), qq));
            cor = fmaf(NTs[qq * ROREG_G + q], cq, cor);
        }
        if constexpr (COR_ONLY) {
            if (me.live && act) cor_out[(out_base + (size_t)it * 8 + w) * ROREG_G + lane] = cor;
            continue;
        }
        float bv = act ? cor : -__builtin_inff();
        if (bv != bv) bv = -__builtin_inff();
        int bi = act ? lane : 0x7fffffff;
        const float mine = bv;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o);
            const int oi = __shfl_xor(bi, o);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        const float margin = 1e-4f * sqrtf(n1 * n0);
        unsigned long long cand = __ballot(act && !(mine < bv - margin));        // (NaN-safe: a NaN correlation stays a candidate)
        if (!(margin == margin) || !(bv > -__builtin_inff())) cand = 0xfffffffffffffffull;     // non-finite input: evaluate every a literally
        if (__popcll(cand) > 1 && me.live) {
            // ---- exact path: the literal evaluation (des2r_body's arithmetic) for the candidates only ---------------------------------
            wave_fence();
            const FT *f1 = reinterpret_cast<const FT *>(feats1_v) + me.r1 * DES2R_ROW, *f0 = reinterpret_cast<const FT *>(feats0_v) + me.r0 * DES2R_ROW;
            for (int e = lane; e < DES2R_ROW; e += 64) {
                const int le = (e / ROREG_G) * DES2R_PITCH + e % ROREG_G;
                X1[le] = feat_ld(f1, e); X2[le] = feat_ld(f0, e);
            }
            wave_fence();
            float best = -__builtin_inff();
            int best_a = 0x7fffffff;
            const int fl = lane & 31;
            while (cand) {
                const int a = __builtin_ctzll(cand);
                cand &= cand - 1;
                const int pmine = P8[a * ROREG_G + q];          // lane g holds P[a][g]; broadcast per step
                float sf = 0.f;
                for (int g = 0; g < ROREG_G; ++g) {
                    const int pg = __builtin_amdgcn_readlane(pmine, g);
                    sf = __fadd_rn(sf, __fmul_rn(X1[fl * DES2R_PITCH + pg], X2[fl * DES2R_PITCH + g]));
                }
                float v = 0.f;
#pragma unroll
                for (int f = 0; f < ROREG_F; ++f)
                    v = __fadd_rn(v, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sf), f)));
                if (v != v) v = -__builtin_inff();
                if (v > best || best_a == 0x7fffffff) { best = v; best_a = a; }
            }
            bi = best_a;
            if (recheck && lane == 0) atomicAdd(recheck, 1);
        }
        if (me.live && lane == 0) idx_out[out_base + (size_t)it * 8 + w] = bi;
    }
}

template <typename FT>
__global__ __launch_bounds__(512) void des2r_irrep_kernel(const float *__restrict__ coef1, const int64_t *__restrict__ rows1,
                                                          const float *__restrict__ coef0, const int64_t *__restrict__ rows0,
                                                          const void *__restrict__ feats1, const void *__restrict__ feats0, Des2rTabs tabs,
                                                          const uint8_t *__restrict__ P8, int M, int64_t *__restrict__ idx_out, int32_t *recheck) {
    const int base = blockIdx.x * (8 * DES2R_ITER);
    const int n_iter = min(DES2R_ITER, (M - base + 7) / 8);
    auto rows_of = [&](int it, int w) {
        const int b = base + it * 8 + w;
        Des2rRows r;
        r.live = b < M;
        const int bb = r.live ? b : M - 1;
        r.c1 = coef1; r.c0 = coef0;
        r.r1 = rows1 ? (size_t)rows1[bb] : (size_t)bb; r.r0 = rows0 ? (size_t)rows0[bb] : (size_t)bb;
        return r;
    };
    des2r_irrep_loop<FT, false>(rows_of, n_iter, feats1, feats0, tabs, P8, idx_out, (size_t)base, recheck);
}

__global__ __launch_bounds__(512) void group_corr_irrep_kernel(const float *__restrict__ coef1, const int64_t *__restrict__ rows1,
                                                               const float *__restrict__ coef0, const int64_t *__restrict__ rows0, Des2rTabs tabs,
                                                               int M, float *__restrict__ cor_out) {
    const int base = blockIdx.x * (8 * DES2R_ITER);
    const int n_iter = min(DES2R_ITER, (M - base + 7) / 8);
    auto rows_of = [&](int it, int w) {
        const int b = base + it * 8 + w;
        Des2rRows r;
        r.live = b < M;
        const int bb = r.live ? b : M - 1;
        r.c1 = coef1; r.c0 = coef0;
        r.r1 = rows1 ? (size_t)rows1[bb] : (size_t)bb; r.r0 = rows0 ? (size_t)rows0[bb] : (size_t)bb;
        return r;
    };
    des2r_irrep_loop<float, true>(rows_of, n_iter, nullptr, nullptr, tabs, nullptr, nullptr, (size_t)base, nullptr, cor_out);
}

// all pairs of a scene in one launch: blockIdx.y = pair, cloud 1 is the permuted side (test/estimator.py:108-110)
template <typename FT>
__global__ __launch_bounds__(512) void des2r_irrep_batch_kernel(const roreg::LtTask *__restrict__ tasks, Des2rTabs tabs,
                                                                const uint8_t *__restrict__ P8, int64_t *__restrict__ dr_all, int32_t *recheck) {
    const roreg::LtTask t = tasks[blockIdx.y];
    const int base = blockIdx.x * (8 * DES2R_ITER);
    if (base >= t.n) return;
    const int n_iter = min(DES2R_ITER, (t.n - base + 7) / 8);
    auto rows_of = [&](int it, int w) {
        const int i = base + it * 8 + w;
        Des2rRows r;
        r.live = i < t.n;
        r.c1 = t.coef1; r.c0 = t.coef0;
        roreg::lt_rows(t, r.live ? i : t.n - 1, r.r0, r.r1);
        return r;
    };
    des2r_irrep_loop<FT, false>(rows_of, n_iter, t.after1, t.after0, tabs, P8, dr_all, (size_t)(t.off + base), recheck);
}

}  // namespace

void roreg::launch_des2r_batch(const LtTask *tasks, int n_tasks, int max_n, int64_t *dr_all, bool irrep, bool feat_bf16, hipStream_t s) {
    roreg::ProfScope prof(roreg::PROF_DES2R, s);
    if (!irrep) {
        hipLaunchKernelGGL(des2r_batch_kernel, dim3((max_n + 3) / 4, n_tasks), dim3(256), 0, s, tasks, roreg::group_tables().P8, dr_all);
        return;
    }
    const size_t lds = (ROREG_G * ROREG_G + 8 * 2 * DES2R_LROW) * sizeof(float);
    const dim3 grid((max_n + 8 * DES2R_ITER - 1) / (8 * DES2R_ITER), n_tasks);
    if (feat_bf16) {
        auto kern = des2r_irrep_batch_kernel<__bf16>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, tasks, g_tabs, roreg::group_tables().P8, dr_all, g_recheck);
    } else {
        auto kern = des2r_irrep_batch_kernel<float>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, tasks, g_tabs, roreg::group_tables().P8, dr_all, g_recheck);
    }
}

bool roreg::des2r_tables_ready() { return g_tabs.NT != nullptr; }

extern "C" int roreg_set_des2r_tables(int transpose_table, const uint8_t *ia_host, const uint8_t *ib_host, const uint8_t *cnt_host, const float *NT_host) {
    ROREG_REQUIRE(ia_host && ib_host && cnt_host && NT_host && (transpose_table == 0 || transpose_table == 1), "roreg_set_des2r_tables: bad arguments");
    for (int q = 0; q < ROREG_G; ++q) {
        ROREG_REQUIRE(cnt_host[q] >= 1 && cnt_host[q] <= 5, "roreg_set_des2r_tables: bad count");
        for (int k = 0; k < 5; ++k) ROREG_REQUIRE(ia_host[q * 5 + k] < ROREG_G && ib_host[q * 5 + k] < ROREG_G, "roreg_set_des2r_tables: index out of range");
    }
    Des2rTabs &t = transpose_table ? g_tabs_t : g_tabs;
    if (!t.NT) {
        uint8_t *b = nullptr; float *f = nullptr;
        if (hipMalloc(&b, 1024) != hipSuccess || hipMalloc(&f, 3600 * sizeof(float)) != hipSuccess) {
            roreg::set_error("roreg_set_des2r_tables: hipMalloc failed");
            return 1;
        }
        t.ia = b; t.ib = b + 320; t.cnt = b + 640; t.NT = f;
    }
    if (!g_recheck) {
        if (hipMalloc(&g_recheck, 64) != hipSuccess) { roreg::set_error("roreg_set_des2r_tables: hipMalloc failed"); return 1; }
        (void)hipMemset(g_recheck, 0, 64);
    }
    if (hipMemcpy(const_cast<uint8_t *>(t.ia), ia_host, 300, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(const_cast<uint8_t *>(t.ib), ib_host, 300, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(const_cast<uint8_t *>(t.cnt), cnt_host, 60, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(const_cast<float *>(t.NT), NT_host, 3600 * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        roreg::set_error("roreg_set_des2r_tables: hipMemcpy failed");
        return 1;
    }
    return 0;
}

extern "C" int roreg_group_corr_irrep(const float *perm_coefs, const int64_t *perm_rows, const float *bcast_coefs, const int64_t *bcast_rows, int M,
                                      int transpose_table, float *cor_out, void *stream) {
    if (M == 0) return 0;
    ROREG_REQUIRE(perm_coefs && bcast_coefs && cor_out && M > 0, "roreg_group_corr_irrep: bad arguments");
    const Des2rTabs &t = transpose_table ? g_tabs_t : g_tabs;
    ROREG_REQUIRE(t.NT, "roreg_group_corr_irrep: roreg_set_des2r_tables has not been called for this table");
    const size_t lds = (ROREG_G * ROREG_G + 8 * 2 * DES2R_LROW) * sizeof(float);
    auto kern = group_corr_irrep_kernel;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3((M + 8 * DES2R_ITER - 1) / (8 * DES2R_ITER)), dim3(512), lds, roreg::as_stream(stream), perm_coefs, perm_rows,
                       bcast_coefs, bcast_rows, t, M, cor_out);
    ROREG_CHECK_LAUNCH("roreg_group_corr_irrep");
    return 0;
}

extern "C" int roreg_des2r_recheck_count(int reset, int32_t *count_out) {
    ROREG_REQUIRE(g_recheck && count_out, "roreg_des2r_recheck_count: tables not set");
    if (hipMemcpy(count_out, g_recheck, sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) { roreg::set_error("roreg_des2r_recheck_count: copy failed"); return 1; }
    if (reset) (void)hipMemset(g_recheck, 0, sizeof(int32_t));
    return 0;
}

extern "C" int roreg_des2r_irrep(const float *coef1, const int64_t *rows1, const float *coef0, const int64_t *rows0, const void *feats1,
                                 const void *feats0, int feat_bf16, int M, int64_t *idx_out, void *stream) {
    if (M == 0) return 0;
    ROREG_REQUIRE(coef1 && coef0 && feats1 && feats0 && idx_out && M > 0, "roreg_des2r_irrep: bad arguments");
    ROREG_REQUIRE(roreg::group_tables().ready && g_tabs.NT, "roreg_des2r_irrep: group / des2r tables not set");
    hipStream_t s = roreg::as_stream(stream);
    const size_t lds = (ROREG_G * ROREG_G + 8 * 2 * DES2R_LROW) * sizeof(float);
    const dim3 grid((M + 8 * DES2R_ITER - 1) / (8 * DES2R_ITER));
    if (feat_bf16) {
        auto kern = des2r_irrep_kernel<__bf16>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, coef1, rows1, coef0, rows0, feats1, feats0, g_tabs, roreg::group_tables().P8, M, idx_out, g_recheck);
    } else {
        auto kern = des2r_irrep_kernel<float>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, coef1, rows1, coef0, rows0, feats1, feats0, g_tabs, roreg::group_tables().P8, M, idx_out, g_recheck);
    }
    ROREG_CHECK_LAUNCH("roreg_des2r_irrep");
    return 0;
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
    static const int split = [] { const char *e = getenv("ROREG_DES2R_SPLIT"); return e ? atoi(e) : 2; }();      // (0: the round-2 kernel, 1: bank-split + packed, for A/B)
    static const int nch = [] { const char *e = getenv("ROREG_DES2R_NCH"); return e ? atoi(e) : 16; }();       // channels per pass of the DPP form (8 | 16)
    if (split == 2 && nch == 8)
        hipLaunchKernelGGL(des2r_dpp_kernel<8>, dim3((M + 3) / 4), dim3(256), 0, roreg::as_stream(stream), perm_feats, perm_rows, bcast_feats,
                           bcast_rows, transpose_table ? roreg::group_tables().split_t : roreg::group_tables().split, M, idx_out, cor_out);
    else if (split == 2)
        hipLaunchKernelGGL(des2r_dpp_kernel<16>, dim3((M + 3) / 4), dim3(256), 0, roreg::as_stream(stream), perm_feats, perm_rows, bcast_feats,
                           bcast_rows, transpose_table ? roreg::group_tables().split_t : roreg::group_tables().split, M, idx_out, cor_out);
    else if (split)
        hipLaunchKernelGGL(des2r_split_kernel, dim3((M + 3) / 4), dim3(256), 0, roreg::as_stream(stream), perm_feats, perm_rows, bcast_feats,
                           bcast_rows, transpose_table ? roreg::group_tables().split_t : roreg::group_tables().split, M, idx_out, cor_out);
    else
        hipLaunchKernelGGL(des2r_kernel, dim3((M + 3) / 4), dim3(256), 0, roreg::as_stream(stream), perm_feats, perm_rows, bcast_feats,
                           bcast_rows, transpose_table ? roreg::group_tables().P8t : roreg::group_tables().P8, M, idx_out, cor_out);
    ROREG_CHECK_LAUNCH("roreg_group_corr");
    return 0;
}
