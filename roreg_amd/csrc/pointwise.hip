// Streaming (HBM-bound) kernels of the path: descriptor normalisation, invariant descriptors, the detector's
// 60x60 self-correlation, ET input assembly and the quaternion -> local-transform assembly.
#include "common.h"

// Bit-exactness contract: no fused multiply-add may be formed from separate * and + in this file (hipcc's
// default is -ffp-contract=fast, and the __f*_rn helpers are plain operators); sqrtf and / are correctly
// rounded under hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt.
#pragma clang fp contract(off)

namespace {

// numpy's pairwise float32 sum for a contiguous run of n (8 <= n <= 128) elements, reproduced exactly
// (numpy/_core/src/umath/loops_utils.h.src, @TYPE@_pairwise_sum): eight strided partial sums, a fixed
// combine tree, then the tail.  Used so that the matcher's invariant descriptor (np.mean / np.sum,
// test/matcher.py:69-72) is bit-identical to the reference's on the same input.
__device__ __forceinline__ float np_pairwise_sum(const float *a, int n) {
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = __fadd_rn(r[j], a[i + j]);
    }
    float res = __fadd_rn(__fadd_rn(__fadd_rn(r[0], r[1]), __fadd_rn(r[2], r[3])),
                          __fadd_rn(__fadd_rn(r[4], r[5]), __fadd_rn(r[6], r[7])));
    for (; i < n; ++i) res = __fadd_rn(res, a[i]);
    return res;
}

// ---- gf_finalize: one wave per keypoint ---------------------------------------------------------------
// OT = float, or __bf16: the descriptors are STORED in bfloat16 (round to nearest even), BASELINE config 5
template <typename OT>
__global__ __launch_bounds__(256) void gf_finalize_kernel(const float *__restrict__ raw, void *__restrict__ eqv_v,
                                                          float *__restrict__ inv, int B) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + w;
    if (b >= B) return;
    const float *src = raw + (size_t)b * (ROREG_F * ROREG_G);
    OT *dst = reinterpret_cast<OT *>(eqv_v) + (size_t)b * (ROREG_F * ROREG_G);
    const bool act = lane < ROREG_G;
    float v[ROREG_F];
    float n2 = 0.f;
#pragma unroll
    for (int f = 0; f < ROREG_F; ++f) {
        v[f] = act ? src[f * ROREG_G + lane] : 0.f;
        n2 += v[f] * v[f];
    }
    const float nrm = fmaxf(sqrtf(n2), 1e-4f);
    if (act) {
#pragma unroll
        for (int f = 0; f < ROREG_F; ++f) dst[f * ROREG_G + lane] = (OT)(v[f] / nrm);
    }
    if (inv) {
        // mean over g of the un-normalised features, then normalise over the 32 channels
        float mine = 0.f, m2 = 0.f;
#pragma unroll
        for (int f = 0; f < ROREG_F; ++f) {
            float s = v[f];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            s *= (1.0f / ROREG_G);
            m2 += s * s;
            if (lane == f) mine = s;
        }
        const float mn = fmaxf(sqrtf(m2), 1e-4f);
        if (lane < ROREG_F) inv[(size_t)b * ROREG_F + lane] = mine / mn;
    }
}

// ---- inv_descriptor: one wave per keypoint; lanes 0..31 own one channel row each ---------------------
// FT = float or __bf16 (descriptors stored in bfloat16; the arithmetic below is float32 on the stored values either way)
template <typename FT>
__global__ __launch_bounds__(256) void inv_descriptor_kernel(const void *__restrict__ eqv_v, float *__restrict__ inv, int N) {
    __shared__ float tile[4][ROREG_F * ROREG_G + 32];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int n = blockIdx.x * 4 + w;
    const bool live = n < N;
    if (live) {
        const FT *src = reinterpret_cast<const FT *>(eqv_v) + (size_t)n * (ROREG_F * ROREG_G);
        for (int i = lane; i < ROREG_F * ROREG_G; i += 64) tile[w][i] = (float)src[i];
    }
    __syncthreads();
    if (!live) return;
    float m = 0.f;
    if (lane < ROREG_F) {
        float row[ROREG_G];
#pragma unroll
        for (int g = 0; g < ROREG_G; ++g) row[g] = tile[w][lane * ROREG_G + g];
        // np.mean(axis=-1) on float32: float32 pairwise sum, then true_divide by the np.intp count, which
        // numpy >= 2 evaluates in float64 and casts back (numpy/_core/_methods.py:_mean)
        m = (float)((double)np_pairwise_sum(row, ROREG_G) / 60.0);
    }
    // sum of squares over the 32 channels in numpy's pairwise order (n=32: 8 partials, 4 rounds)
    float sq[ROREG_F];
#pragma unroll
    for (int f = 0; f < ROREG_F; ++f) {
        const float mf = __shfl(m, f);
        sq[f] = __fmul_rn(mf, mf);
    }
    const float nrm = __fadd_rn(sqrtf(np_pairwise_sum(sq, ROREG_F)), 1e-5f);
    if (lane < ROREG_F) inv[(size_t)n * ROREG_F + lane] = __fdiv_rn(m, nrm);
}

// ---- det_score: one wave per keypoint ---------------------------------------------------------------
__global__ __launch_bounds__(256) void det_score_kernel(const float *__restrict__ enc, const uint8_t *__restrict__ P8,
                                                        float *__restrict__ scores, int B) {
    __shared__ float fn[4][16 * ROREG_G];
    __shared__ uint8_t Pl[ROREG_G * ROREG_G];
    for (int i = threadIdx.x; i < ROREG_G * ROREG_G; i += 256) Pl[i] = P8[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + w;
    const bool live = b < B;
    const float *src = enc + (size_t)(live ? b : 0) * (16 * ROREG_G);
    const bool act = lane < ROREG_G;
    float v[16];
    float n2 = 0.f;
#pragma unroll
    for (int f = 0; f < 16; ++f) {
        v[f] = act ? src[f * ROREG_G + lane] : 0.f;
        n2 += v[f] * v[f];
    }
    const float rn = sqrtf(n2);
    if (act) {
#pragma unroll
        for (int f = 0; f < 16; ++f) fn[w][f * ROREG_G + lane] = v[f] / rn;
    }
    __syncthreads();
    if (!live) return;
    float c = 0.f;
    if (act) {
        float s[16];
#pragma unroll
        for (int f = 0; f < 16; ++f) s[f] = 0.f;
        for (int g = 0; g < ROREG_G; ++g) {
            const int pg = Pl[lane * ROREG_G + g];
#pragma unroll
            for (int f = 0; f < 16; ++f) s[f] += fn[w][f * ROREG_G + pg] * fn[w][f * ROREG_G + g];
        }
#pragma unroll
        for (int f = 0; f < 16; ++f) c += s[f];
    }
    // unbiased std over the 60 lanes
    float sum = act ? c : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum * (1.0f / ROREG_G);
    float d = act ? (c - mean) : 0.f;
    float ss = d * d;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    if (lane == 0) scores[b] = sqrtf(ss / (ROREG_G - 1));
}

// ---- et_gather: one block per correspondence ------------------------------------------------------------
// FT = float or __bf16: the clouds' group features as stored (BASELINE config 5 keeps them in bfloat16); the assembled ET input is float32
template <typename FT>
__device__ __forceinline__ void et_gather_body(const void *__restrict__ before0_v, const void *__restrict__ before1_v,
                                               const void *__restrict__ after0_v, const void *__restrict__ after1_v, size_t r0, size_t r1,
                                               int a, const int32_t *__restrict__ P, float *__restrict__ dst,
                                               const float *__restrict__ bn_scale = nullptr, const float *__restrict__ bn_shift = nullptr,
                                               float *__restrict__ bound_row = nullptr) {
    __shared__ int perm[ROREG_G];
    __shared__ float red[4];
    if (threadIdx.x < ROREG_G) perm[threadIdx.x] = P[a * ROREG_G + threadIdx.x];
    __syncthreads();
    const FT *s_b1 = reinterpret_cast<const FT *>(before1_v) + r1 * (ROREG_F * ROREG_G), *s_b0 = reinterpret_cast<const FT *>(before0_v) + r0 * (ROREG_F * ROREG_G);
    const FT *s_a1 = reinterpret_cast<const FT *>(after1_v) + r1 * (ROREG_F * ROREG_G), *s_a0 = reinterpret_cast<const FT *>(after0_v) + r0 * (ROREG_F * ROREG_G);
    float mx = 0.f;
    for (int i = threadIdx.x; i < ROREG_F * ROREG_G; i += 256) {
        const int c = i / ROREG_G, g = i - c * ROREG_G;
        const int pg = c * ROREG_G + perm[g];
        const float v0 = (float)s_b1[pg], v1 = (float)s_b0[i], v2 = (float)s_a1[pg], v3 = (float)s_a0[i];
        __builtin_nontemporal_store(v0, dst + i);                  // (streamed: ET's first transform reads the row after gigabytes of other rows)
        __builtin_nontemporal_store(v1, dst + ROREG_F * ROREG_G + i);
        __builtin_nontemporal_store(v2, dst + 2 * ROREG_F * ROREG_G + i);
        __builtin_nontemporal_store(v3, dst + 3 * ROREG_F * ROREG_G + i);
        if (bound_row) {       // the row's bound for the fp16 x 2 split of FT(ReLU(BN(x))): what roreg_row_bound computes, without re-reading x
            mx = fmaxf(mx, fmaxf(fmaf(v0, bn_scale[c], bn_shift[c]), 0.f));
            mx = fmaxf(mx, fmaxf(fmaf(v1, bn_scale[ROREG_F + c], bn_shift[ROREG_F + c]), 0.f));
            mx = fmaxf(mx, fmaxf(fmaf(v2, bn_scale[2 * ROREG_F + c], bn_shift[2 * ROREG_F + c]), 0.f));
            mx = fmaxf(mx, fmaxf(fmaf(v3, bn_scale[3 * ROREG_F + c], bn_shift[3 * ROREG_F + c]), 0.f));
        }
    }
    if (bound_row) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
        __syncthreads();
        if (threadIdx.x == 0) *bound_row = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * 7.7536f;      // sqrt(60) with the margin of row_bound_kernel
    }
}

template <typename FT>
__global__ __launch_bounds__(256) void et_gather_kernel(const void *__restrict__ before0, const void *__restrict__ before1,
                                                        const void *__restrict__ after0, const void *__restrict__ after1,
                                                        const int64_t *__restrict__ rows0, const int64_t *__restrict__ rows1,
                                                        const int64_t *__restrict__ pre_idx, const int32_t *__restrict__ P,
                                                        float *__restrict__ x, int M) {
    const int b = blockIdx.x;
    if (b >= M) return;
    const size_t r0 = rows0 ? (size_t)rows0[b] : (size_t)b, r1 = rows1 ? (size_t)rows1[b] : (size_t)b;
    et_gather_body<FT>(before0, before1, after0, after1, r0, r1, (int)pre_idx[b], P, x + (size_t)b * (4 * ROREG_F * ROREG_G));
}

template <typename FT>
__global__ __launch_bounds__(256) void et_gather_batch_kernel(const roreg::LtTask *__restrict__ tasks, const int64_t *__restrict__ dr_all,
                                                              const int32_t *__restrict__ P, float *__restrict__ x_all,
                                                              const float *__restrict__ bn_scale, const float *__restrict__ bn_shift,
                                                              float *__restrict__ bound_all) {
    const roreg::LtTask t = tasks[blockIdx.y];
    const int i = blockIdx.x;
    if (i >= t.n) return;
    size_t r0, r1;
    roreg::lt_rows(t, i, r0, r1);
    et_gather_body<FT>(t.before0, t.before1, t.after0, t.after1, r0, r1, (int)dr_all[t.off + i], P, x_all + (size_t)(t.off + i) * (4 * ROREG_F * ROREG_G),
                       bn_scale, bn_shift, bound_all ? bound_all + t.off + i : nullptr);
}

// ---- quat_to_trans: one thread per correspondence --------------------------------------------------------
__device__ __forceinline__ void quat_to_trans_body(const float *__restrict__ q4, int anchor, const double *__restrict__ keys0, size_t r0,
                                                   const double *__restrict__ keys1, size_t r1, const float *__restrict__ Rf,
                                                   double *__restrict__ o, float *__restrict__ quat_out4) {
    float w = q4[0], x = q4[1], y = q4[2], z = q4[3];
    // torch.norm(dim=1) then divide (network/eqv_trans.py:137)
    const float n = sqrtf(__fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(w, w), __fmul_rn(x, x)), __fmul_rn(y, y)), __fmul_rn(z, z)));
    w = __fdiv_rn(w, n); x = __fdiv_rn(x, n); y = __fdiv_rn(y, n); z = __fdiv_rn(z, n);
    if (quat_out4) { quat_out4[0] = w; quat_out4[1] = x; quat_out4[2] = y; quat_out4[3] = z; }
    // utils/r_eval.py:90-106 evaluated in float32 (the quaternion is a float32 array), left to right, no FMA
    auto two = [](float a, float b) { return __fmul_rn(__fmul_rn(2.0f, a), b); };
    float m[9];
    m[0] = __fsub_rn(__fsub_rn(1.0f, two(y, y)), two(z, z));
    m[1] = __fsub_rn(two(x, y), two(z, w));
    m[2] = __fadd_rn(two(x, z), two(y, w));
    m[3] = __fadd_rn(two(x, y), two(z, w));
    m[4] = __fsub_rn(__fsub_rn(1.0f, two(x, x)), two(z, z));
    m[5] = __fsub_rn(two(y, z), two(x, w));
    m[6] = __fsub_rn(two(x, z), two(y, w));
    m[7] = __fadd_rn(two(y, z), two(x, w));
    m[8] = __fsub_rn(__fsub_rn(1.0f, two(x, x)), two(y, y));
    const float *A = Rf + anchor * 9;
    double R[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            R[r * 3 + c] = __dadd_rn(__dadd_rn(__dmul_rn((double)m[r * 3], (double)A[c]), __dmul_rn((double)m[r * 3 + 1], (double)A[3 + c])),
                                     __dmul_rn((double)m[r * 3 + 2], (double)A[6 + c]));
    const double k0x = keys0[r0 * 3], k0y = keys0[r0 * 3 + 1], k0z = keys0[r0 * 3 + 2];
    const double k1x = keys1[r1 * 3], k1y = keys1[r1 * 3 + 1], k1z = keys1[r1 * 3 + 2];
    const double k0[3] = {k0x, k0y, k0z};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        o[r * 4 + 0] = R[r * 3 + 0]; o[r * 4 + 1] = R[r * 3 + 1]; o[r * 4 + 2] = R[r * 3 + 2];
        const double rot = __dadd_rn(__dadd_rn(__dmul_rn(k1x, R[r * 3]), __dmul_rn(k1y, R[r * 3 + 1])), __dmul_rn(k1z, R[r * 3 + 2]));
        o[r * 4 + 3] = __dsub_rn(k0[r], rot);
    }
}

__global__ __launch_bounds__(256) void quat_to_trans_kernel(const float *__restrict__ q, const int64_t *__restrict__ anchor,
                                                            const double *__restrict__ keys0, const int64_t *__restrict__ rows0,
                                                            const double *__restrict__ keys1, const int64_t *__restrict__ rows1,
                                                            const float *__restrict__ Rf, int M, double *__restrict__ T,
                                                            float *__restrict__ quat_out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const size_t r0 = rows0 ? (size_t)rows0[i] : (size_t)i, r1 = rows1 ? (size_t)rows1[i] : (size_t)i;
    quat_to_trans_body(q + (size_t)i * 4, (int)anchor[i], keys0, r0, keys1, r1, Rf, T + (size_t)i * 12, quat_out ? quat_out + (size_t)i * 4 : nullptr);
}

__global__ __launch_bounds__(256) void quat_to_trans_batch_kernel(const roreg::LtTask *__restrict__ tasks, const float *__restrict__ q_all,
                                                                  const int64_t *__restrict__ dr_all, const float *__restrict__ Rf,
                                                                  double *__restrict__ T_all) {
    const roreg::LtTask t = tasks[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= t.n) return;
    size_t r0, r1;
    roreg::lt_rows(t, i, r0, r1);
    const size_t row = (size_t)(t.off + i);
    quat_to_trans_body(q_all + row * 4, (int)dr_all[row], t.keys0, r0, t.keys1, r1, Rf, T_all + row * 12, nullptr);
}

__global__ __launch_bounds__(256) void gather_rows_f64_kernel(const double *__restrict__ src, const int64_t *__restrict__ rows,
                                                              int M, int width, double *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M * width) return;
    const int r = i / width, c = i - r * width;
    out[i] = src[(size_t)rows[r] * width + c];
}

// Row gathers of several (source tensor, row list) pairs in ONE launch: task t copies rows[0..n) of its source (row_bytes each, a multiple
// of 8) to consecutive rows of its destination.  blockIdx.y = task; a workgroup moves whole rows in 8-byte pieces.
struct GatherTask { const void *src; const int64_t *rows; void *dst; int32_t n, pad_; };
__global__ __launch_bounds__(256) void gather_rows_batch_kernel(const GatherTask *__restrict__ tasks, int row_words /* 8-byte words per row */,
                                                                int rows_per_block) {
    const GatherTask t = tasks[blockIdx.y];
    const int r0 = blockIdx.x * rows_per_block;
    if (r0 >= t.n) return;
    const int r1 = min(r0 + rows_per_block, t.n);
    const uint2 *src = reinterpret_cast<const uint2 *>(t.src);
    uint2 *dst = reinterpret_cast<uint2 *>(t.dst);
    for (int r = r0; r < r1; ++r) {
        const uint2 *s = src + (size_t)t.rows[r] * row_words;
        uint2 *d = dst + (size_t)r * row_words;
        for (int i = threadIdx.x; i < row_words; i += 256) d[i] = s[i];
    }
}

}  // namespace

extern "C" int roreg_gather_rows_batch(const roreg_gather_task *tasks_dev, int n_tasks, int max_n, int row_bytes, void *stream) {
    if (n_tasks == 0 || max_n == 0) return 0;
    ROREG_REQUIRE(tasks_dev && n_tasks > 0 && max_n > 0 && row_bytes > 0 && row_bytes % 8 == 0, "roreg_gather_rows_batch: bad arguments (row_bytes must be a multiple of 8)");
    static_assert(sizeof(roreg_gather_task) == sizeof(GatherTask), "roreg_gather_task layout");
    const int row_words = row_bytes / 8;
    const int rpb = row_words >= 256 ? 4 : (row_words >= 32 ? 32 : 256);      // long rows: a few per workgroup; short rows: many
    hipLaunchKernelGGL(gather_rows_batch_kernel, dim3((max_n + rpb - 1) / rpb, n_tasks), dim3(256), 0, roreg::as_stream(stream),
                       reinterpret_cast<const GatherTask *>(tasks_dev), row_words, rpb);
    ROREG_CHECK_LAUNCH("roreg_gather_rows_batch");
    return 0;
}

extern "C" int roreg_gf_finalize(const float *eqv_raw, void *eqv, int eqv_bf16, float *inv, int B, void *stream) {
    if (B == 0) return 0;
    ROREG_REQUIRE(eqv_raw && eqv && B > 0, "roreg_gf_finalize: bad arguments");
    if (eqv_bf16) hipLaunchKernelGGL(gf_finalize_kernel<__bf16>, dim3((B + 3) / 4), dim3(256), 0, roreg::as_stream(stream), eqv_raw, eqv, inv, B);
    else hipLaunchKernelGGL(gf_finalize_kernel<float>, dim3((B + 3) / 4), dim3(256), 0, roreg::as_stream(stream), eqv_raw, eqv, inv, B);
    ROREG_CHECK_LAUNCH("roreg_gf_finalize");
    return 0;
}

extern "C" int roreg_inv_descriptor(const void *eqv, int eqv_bf16, float *inv, int N, void *stream) {
    if (N == 0) return 0;
    ROREG_REQUIRE(eqv && inv && N > 0, "roreg_inv_descriptor: bad arguments");
    if (eqv_bf16) hipLaunchKernelGGL(inv_descriptor_kernel<__bf16>, dim3((N + 3) / 4), dim3(256), 0, roreg::as_stream(stream), eqv, inv, N);
    else hipLaunchKernelGGL(inv_descriptor_kernel<float>, dim3((N + 3) / 4), dim3(256), 0, roreg::as_stream(stream), eqv, inv, N);
    ROREG_CHECK_LAUNCH("roreg_inv_descriptor");
    return 0;
}

extern "C" int roreg_det_score(const float *enc, float *scores, int B, void *stream) {
    if (B == 0) return 0;
    ROREG_REQUIRE(enc && scores && B > 0, "roreg_det_score: bad arguments");
    ROREG_REQUIRE(roreg::group_tables().ready, "roreg_det_score: group tables not set");
    if (B == 0) return 0;
    hipLaunchKernelGGL(det_score_kernel, dim3((B + 3) / 4), dim3(256), 0, roreg::as_stream(stream), enc,
                       roreg::group_tables().P8, scores, B);
    ROREG_CHECK_LAUNCH("roreg_det_score");
    return 0;
}

extern "C" int roreg_et_gather(const void *before0, const void *before1, const void *after0, const void *after1, int feat_bf16,
                               const int64_t *rows0, const int64_t *rows1, const int64_t *pre_idx, int M, float *x_out,
                               void *stream) {
    if (M == 0) return 0;
    ROREG_REQUIRE(before0 && before1 && after0 && after1 && pre_idx && x_out && M > 0, "roreg_et_gather: bad arguments");
    ROREG_REQUIRE(roreg::group_tables().ready, "roreg_et_gather: group tables not set");
    if (M == 0) return 0;
    if (feat_bf16)
        hipLaunchKernelGGL(et_gather_kernel<__bf16>, dim3(M), dim3(256), 0, roreg::as_stream(stream), before0, before1, after0, after1,
                           rows0, rows1, pre_idx, roreg::group_tables().P, x_out, M);
    else
        hipLaunchKernelGGL(et_gather_kernel<float>, dim3(M), dim3(256), 0, roreg::as_stream(stream), before0, before1, after0, after1,
                           rows0, rows1, pre_idx, roreg::group_tables().P, x_out, M);
    ROREG_CHECK_LAUNCH("roreg_et_gather");
    return 0;
}

extern "C" int roreg_quat_to_trans(const float *q, const int64_t *anchor, const double *keys0, const int64_t *rows0,
                                   const double *keys1, const int64_t *rows1, int M, double *Trans_out, float *quat_out,
                                   void *stream) {
    if (M == 0) return 0;
    ROREG_REQUIRE(q && anchor && keys0 && keys1 && Trans_out && M > 0, "roreg_quat_to_trans: bad arguments");
    ROREG_REQUIRE(roreg::group_tables().ready, "roreg_quat_to_trans: group tables not set");
    if (M == 0) return 0;
    hipLaunchKernelGGL(quat_to_trans_kernel, dim3((M + 255) / 256), dim3(256), 0, roreg::as_stream(stream), q, anchor, keys0,
                       rows0, keys1, rows1, roreg::group_tables().Rf, M, Trans_out, quat_out);
    ROREG_CHECK_LAUNCH("roreg_quat_to_trans");
    return 0;
}

extern "C" int roreg_gather_rows_f64(const double *src, const int64_t *rows, int M, int width, double *out, void *stream) {
    if (M == 0) return 0;
    ROREG_REQUIRE(src && rows && out && M > 0 && width > 0, "roreg_gather_rows_f64: bad arguments");
    if (M == 0) return 0;
    hipLaunchKernelGGL(gather_rows_f64_kernel, dim3((M * width + 255) / 256), dim3(256), 0, roreg::as_stream(stream), src, rows,
                       M, width, out);
    ROREG_CHECK_LAUNCH("roreg_gather_rows_f64");
    return 0;
}

extern "C" int roreg_lt_prepare_batch(const roreg_lt_task *tasks_dev, int n_tasks, int max_n, int flags, int64_t *dr_out, float *x_out,
                                      const float *bn_scale, const float *bn_shift, float *x_bound_out, void *stream) {
    if (n_tasks == 0 || max_n == 0) return 0;
    ROREG_REQUIRE(tasks_dev && dr_out && n_tasks > 0 && max_n > 0, "roreg_lt_prepare_batch: bad arguments");
    ROREG_REQUIRE(roreg::group_tables().ready, "roreg_lt_prepare_batch: group tables not set");
    ROREG_REQUIRE(!(flags & 1) || roreg::des2r_tables_ready(), "roreg_lt_prepare_batch: roreg_set_des2r_tables has not been called");
    ROREG_REQUIRE(!(flags & 2) || (flags & 1), "roreg_lt_prepare_batch: bfloat16 features need the irrep-domain Des2R (flags bit 0)");
    ROREG_REQUIRE(!x_bound_out || (x_out && bn_scale && bn_shift), "roreg_lt_prepare_batch: x_bound_out needs x_out and the BatchNorm constants of Conv_init");
    static_assert(sizeof(roreg_lt_task) == sizeof(roreg::LtTask), "roreg_lt_task layout");
    const roreg::LtTask *tasks = reinterpret_cast<const roreg::LtTask *>(tasks_dev);
    hipStream_t s = roreg::as_stream(stream);
    roreg::launch_des2r_batch(tasks, n_tasks, max_n, dr_out, (flags & 1) != 0, (flags & 2) != 0, s);
    if (x_out) {
        if (flags & 2)
            hipLaunchKernelGGL(et_gather_batch_kernel<__bf16>, dim3(max_n, n_tasks), dim3(256), 0, s, tasks, dr_out, roreg::group_tables().P, x_out,
                               bn_scale, bn_shift, x_bound_out);
        else
            hipLaunchKernelGGL(et_gather_batch_kernel<float>, dim3(max_n, n_tasks), dim3(256), 0, s, tasks, dr_out, roreg::group_tables().P, x_out,
                               bn_scale, bn_shift, x_bound_out);
    }
    ROREG_CHECK_LAUNCH("roreg_lt_prepare_batch");
    return 0;
}

extern "C" int roreg_lt_finish_batch(const roreg_lt_task *tasks_dev, int n_tasks, int max_n, const float *q_all, const int64_t *dr_all,
                                     double *Trans_out, void *stream) {
    if (n_tasks == 0 || max_n == 0) return 0;
    ROREG_REQUIRE(tasks_dev && q_all && dr_all && Trans_out && n_tasks > 0 && max_n > 0, "roreg_lt_finish_batch: bad arguments");
    ROREG_REQUIRE(roreg::group_tables().ready, "roreg_lt_finish_batch: group tables not set");
    hipLaunchKernelGGL(quat_to_trans_batch_kernel, dim3((max_n + 255) / 256, n_tasks), dim3(256), 0, roreg::as_stream(stream),
                       reinterpret_cast<const roreg::LtTask *>(tasks_dev), q_all, dr_all, roreg::group_tables().Rf, Trans_out);
    ROREG_CHECK_LAUNCH("roreg_lt_finish_batch");
    return 0;
}
