// One-shot RANSAC: hypothesis scoring, first-best selection and the weighted Kabsch refinement, all fp64.
//
// Scoring is one wavefront per hypothesis sweeping the M correspondences of the pair in coalesced 64-wide
// strides (k0/k1/w are a few hundred KB and stay in L2 across the <=1000 hypotheses).  The point test follows
// the oracle's operation order exactly (no FMA contraction) so inlier masks are bit-identical:
//   p_r = ((k1x*R_r0 + k1y*R_r1) + k1z*R_r2) + t_r ;  d2 = ((dx*dx + dy*dy) + dz*dz) ;  inlier <=> d2 < ird*ird
// Reference: yohoo_ransac.overlap_cal / ransac (test/estimator.py:377-382,426-439), refiner (:28-72),
// transform_points (utils/utils.py:38-46).
#include "common.h"

// Bit-exactness contract: no fused multiply-add may be formed from separate * and + in this file (hipcc's
// default is -ffp-contract=fast, and the __f*_rn helpers are plain operators); sqrtf and / are correctly
// rounded under hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt.
#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ __forceinline__ bool point_inlier(const double *__restrict__ T, double k0x, double k0y, double k0z, double k1x,
                                             double k1y, double k1z, double thr2) {
    const double px = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(k1x, T[0]), __dmul_rn(k1y, T[1])), __dmul_rn(k1z, T[2])), T[3]);
    const double py = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(k1x, T[4]), __dmul_rn(k1y, T[5])), __dmul_rn(k1z, T[6])), T[7]);
    const double pz = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(k1x, T[8]), __dmul_rn(k1y, T[9])), __dmul_rn(k1z, T[10])), T[11]);
    const double dx = __dsub_rn(k0x, px), dy = __dsub_rn(k0y, py), dz = __dsub_rn(k0z, pz);
    const double d2 = __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
    return d2 < thr2;
}

// w == nullptr means "all weights are 1.0" (the mutual matcher's scores, test/matcher.py:109): adding the literal 1.0 is the same
// floating-point operation as adding a loaded 1.0.
__device__ __forceinline__ double weight_of(const double *__restrict__ w, int i) { return w ? w[i] : 1.0; }

// ---- float32 match scores (the rotation-coherence matcher's, test/matcher.py:210): numpy's float32 sum, bit for bit ------------------
// With --RM the reference's scores are a float32 array, so `np.sum(scores[overlap])` (test/estimator.py:381) and `np.sum(scores)` of the
// refinement (:50) are float32 reductions in numpy's PAIRWISE order over the compacted inlier array, and `overlap` is that float32 sum
// divided by M in float32 (numpy 2 promotion: float32 scalar / Python int); the running `overlap > best_overlap` (:433) compares those
// float32 values.  Accumulating the same weights in float64 orders two hypotheses differently whenever their overlaps tie only after
// float32 rounding, so in this mode (w_f32) the kernels rebuild numpy's reduction tree:
//   pairwise(a, n): n < 8: ((0 + a0) + a1) + ...;  n <= 128: r[j] = a[j] + a[8+j] + ... over the multiple-of-8 prefix (j < 8),
//   ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7)), then the < 8 trailing elements one by one;  n > 128: pairwise(a, n2) + pairwise(a + n2, n - n2)
//   with n2 = 8 * floor(n / 16);  and the reduction loop hands pairwise() at most 8192 elements at a time (numpy's buffer size), adding
//   the chunk sums to a running float32 total  (numpy/_core/src/umath/loops_utils.h.src; checked against np.sum for n = 0..30000 by
//   tests/test_oracle_golden.py::test_numpy_pairwise_model).
// One wavefront owns one compacted array in LDS (`buf`, filled in increasing i by ballot + prefix count); `tab` / `lsum` hold the leaf
// table (start, length) and the leaf sums (<= NP_MAX_LEAVES for 8192 elements).
constexpr int NP_CHUNK = 8192;
constexpr int NP_MAX_LEAVES = 80;                       // leaves hold 57..128 elements once n > 128: 8192 / 128 = 64 (+ slack)

struct NpSumScratch {
    int32_t start[NP_MAX_LEAVES], len[NP_MAX_LEAVES];
    float lsum[NP_MAX_LEAVES];
    int32_t stack[40];
    float vstack[24];
};

// pairwise float32 sum of buf[0..n) (n <= NP_CHUNK) by one wavefront; every lane returns the value.  buf / sc: this wave's LDS.
__device__ float np_pairwise_sum_wave(const float *buf, int n, NpSumScratch *sc) {
    const int lane = threadIdx.x & 63;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");         // the compaction's LDS writes of the other lanes
    __builtin_amdgcn_wave_barrier();
    if (n == 0) return 0.f;
    // leaf table, in the recursion's left-to-right order (uniform control flow; lane 0 writes)
    int L = 0;
    if (lane == 0) {
        int sp = 0, at = 0;
        sc->stack[sp++] = n;
        while (sp > 0) {
            const int m = sc->stack[--sp];
            if (m <= 128) { sc->start[L] = at; sc->len[L] = m; at += m; ++L; }
            else { const int n2 = 8 * (m / 16); sc->stack[sp++] = m - n2; sc->stack[sp++] = n2; }
        }
    }
    L = __builtin_amdgcn_readfirstlane(L);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // leaf sums: eight lanes per leaf (lane j owns r[j]), eight leaves per round
    const int grp = lane >> 3, j = lane & 7;
    for (int l0 = 0; l0 < L; l0 += 8) {
        const int l = l0 + grp;
        const bool live = l < L;
        const int s0 = live ? sc->start[l] : 0, len = live ? sc->len[l] : 0;
        float res;
        if (len < 8) {                                            // only the whole array can be this short
            res = 0.f;
            for (int i = 0; i < len; ++i) res = __fadd_rn(res, buf[s0 + i]);
        } else {
            const int nb = len - (len & 7);
            float r = buf[s0 + j];
            for (int i = 8; i < nb; i += 8) r = __fadd_rn(r, buf[s0 + i + j]);
            r = __fadd_rn(r, __shfl_xor(r, 1));                   // (r0+r1), (r2+r3), ...
            r = __fadd_rn(r, __shfl_xor(r, 2));                   // ((r0+r1)+(r2+r3)), ((r4+r5)+(r6+r7))
            r = __fadd_rn(r, __shfl_xor(r, 4));
            res = r;
            for (int i = nb; i < len; ++i) res = __fadd_rn(res, buf[s0 + i]);
        }
        if (live && j == 0) sc->lsum[l] = res;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // the recursion's additions, post-order (left + right), lane 0
    float total = 0.f;
    if (lane == 0) {
        int sp = 0, vp = 0, next = 0;
        sc->stack[sp++] = n;
        while (sp > 0) {
            const int m = sc->stack[--sp];
            if (m < 0) { const float b = sc->vstack[--vp], a = sc->vstack[--vp]; sc->vstack[vp++] = __fadd_rn(a, b); }
            else if (m <= 128) sc->vstack[vp++] = sc->lsum[next++];
            else { const int n2 = 8 * (m / 16); sc->stack[sp++] = -1; sc->stack[sp++] = m - n2; sc->stack[sp++] = n2; }
        }
        total = sc->vstack[0];
    }
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, total)));
}

// np.sum of the float32 weights of the inliers among i in [0, M), in increasing i: streamed through the wave's LDS buffer in numpy's
// chunks of 8192 compacted elements.  `in_at(i)` is evaluated once per i by lane i % 64.  Every lane returns the float32 sum.
template <int CAP, class InlierFn>
__device__ float np_sum_f32_of_inliers(const double *__restrict__ w, int M, float *buf, NpSumScratch *sc, InlierFn in_at, int *count_out) {
    static_assert(CAP <= NP_CHUNK && (CAP == NP_CHUNK || CAP % 64 == 0), "buffer is one numpy chunk at most");
    const int lane = threadIdx.x & 63;
    int fill = 0, total_n = 0;
    float res = 0.f;
    bool any_chunk = false;
    for (int base = 0; base < M; base += 64) {
        const int i = base + lane;
        const bool in = i < M && in_at(i);
        const unsigned long long b = __ballot(in);
        const int pos = fill + __popcll(b & ((1ull << lane) - 1ull));
        const int cnt = __popcll(b);
        const float wi = in ? (float)w[i] : 0.f;
        if (in && pos < CAP) buf[pos] = wi;
        if (fill + cnt >= CAP && CAP == NP_CHUNK) {               // a full numpy chunk: reduce it, carry the spill over
            res = __fadd_rn(res, np_pairwise_sum_wave(buf, CAP, sc));
            any_chunk = true;
            __builtin_amdgcn_wave_barrier();
            if (in && pos >= CAP) buf[pos - CAP] = wi;
            fill = fill + cnt - CAP;
        } else {
            fill += cnt;
        }
        total_n += cnt;
    }
    if (fill > 0 || !any_chunk) res = __fadd_rn(res, np_pairwise_sum_wave(buf, fill, sc));
    if (count_out) *count_out = total_n;
    return res;
}

// float32-score form of the scoring (see above): overlap[h] = float32(np.sum(float32 w[inliers])) / float32(M), stored widened.
template <int CAP>
__device__ __forceinline__ void ransac_score_body_f32(const double *__restrict__ k0, const double *__restrict__ k1,
                                                      const double *__restrict__ w, int M, const double *__restrict__ Trans,
                                                      const int64_t *__restrict__ hyp_rows, int H, double thr2, int h,
                                                      double *__restrict__ overlap, uint8_t *__restrict__ mask, float *buf, NpSumScratch *sc) {
    const int lane = threadIdx.x & 63;
    if (h >= H) return;                                            // (wave-uniform)
    const size_t row = hyp_rows ? (size_t)hyp_rows[h] : (size_t)h;
    double T[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) T[q] = Trans[row * 12 + q];
    const float sum = np_sum_f32_of_inliers<CAP>(w, M, buf, sc, [&](int i) {
        const bool in = point_inlier(T, k0[3 * i], k0[3 * i + 1], k0[3 * i + 2], k1[3 * i], k1[3 * i + 1], k1[3 * i + 2], thr2);
        if (mask) mask[(size_t)h * M + i] = in ? 1 : 0;
        return in;
    }, nullptr);
    if (lane == 0) overlap[h] = (double)__fdiv_rn(sum, (float)M);
}

__device__ __forceinline__ void ransac_score_body(const double *__restrict__ k0, const double *__restrict__ k1,
                                                  const double *__restrict__ w, int M, const double *__restrict__ Trans,
                                                  const int64_t *__restrict__ hyp_rows, int H, double thr2, int h,
                                                  double *__restrict__ overlap, uint8_t *__restrict__ mask) {
    const int lane = threadIdx.x & 63;
    if (h >= H) return;
    const size_t row = hyp_rows ? (size_t)hyp_rows[h] : (size_t)h;
    double T[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) T[q] = Trans[row * 12 + q];
    double acc = 0.0;
    for (int i = lane; i < M; i += 64) {
        const bool in = point_inlier(T, k0[3 * i], k0[3 * i + 1], k0[3 * i + 2], k1[3 * i], k1[3 * i + 1], k1[3 * i + 2], thr2);
        if (in) acc += weight_of(w, i);
        if (mask) mask[(size_t)h * M + i] = in ? 1 : 0;
    }
    acc = wave_sum(acc);
    if (lane == 0) overlap[h] = acc / (double)M;
}

__global__ __launch_bounds__(256) void ransac_score_kernel(const double *__restrict__ k0, const double *__restrict__ k1,
                                                           const double *__restrict__ w, int M, const double *__restrict__ Trans,
                                                           const int64_t *__restrict__ hyp_rows, int H, double thr2,
                                                           double *__restrict__ overlap, uint8_t *__restrict__ mask) {
    ransac_score_body(k0, k1, w, M, Trans, hyp_rows, H, thr2, blockIdx.x * 4 + (threadIdx.x >> 6), overlap, mask);
}

// float32-score variants: WAVES hypotheses per workgroup, each wave with a compaction buffer of CAP floats in LDS
// (<4, 4096>: M <= 4096, 64 KB; <2, 8192>: any M, one numpy chunk per wave, 64 KB)
template <int WAVES, int CAP>
__global__ __launch_bounds__(WAVES * 64) void ransac_score_f32_kernel(const double *__restrict__ k0, const double *__restrict__ k1,
                                                                     const double *__restrict__ w, int M, const double *__restrict__ Trans,
                                                                     const int64_t *__restrict__ hyp_rows, int H, double thr2,
                                                                     double *__restrict__ overlap, uint8_t *__restrict__ mask) {
    __shared__ float buf[WAVES][CAP];
    __shared__ NpSumScratch sc[WAVES];
    const int wv = threadIdx.x >> 6;
    ransac_score_body_f32<CAP>(k0, k1, w, M, Trans, hyp_rows, H, thr2, blockIdx.x * WAVES + wv, overlap, mask, buf[wv], &sc[wv]);
}

// first index of the strictly greatest overlap (> 0), as the reference's running '>' scan
__device__ __forceinline__ void first_best_body(const double *__restrict__ overlap, int H, int32_t *__restrict__ best) {
    __shared__ double sv[256];
    __shared__ int si[256];
    double bv = 0.0;
    int bi = 0x7fffffff;
    for (int h = threadIdx.x; h < H; h += 256) {
        const double v = overlap[h];
        if (v > bv) { bv = v; bi = h; }      // ascending h per thread: first of equal values kept
    }
    sv[threadIdx.x] = bv; si[threadIdx.x] = bi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            const double ov = sv[threadIdx.x + s];
            const int oi = si[threadIdx.x + s];
            if (ov > sv[threadIdx.x] || (ov == sv[threadIdx.x] && oi < si[threadIdx.x])) { sv[threadIdx.x] = ov; si[threadIdx.x] = oi; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *best = (sv[0] > 0.0 && si[0] != 0x7fffffff) ? si[0] : -1;
}

__global__ __launch_bounds__(256) void first_best_kernel(const double *__restrict__ overlap, int H, int32_t *__restrict__ best) {
    first_best_body(overlap, H, best);
}

// ---- 3x3 SVD polar factor R = U V^T by one-sided Jacobi (fp64) ----------------------------------------
__device__ void polar_uvt(const double *Hm, double *R) {
    // A (columns a0,a1,a2) = H ; rotate column pairs until orthogonal: A = U S, accumulated V gives H = U S V^T
    double A[9], V[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) { A[i] = Hm[i]; V[i] = (i % 4 == 0) ? 1.0 : 0.0; }
    double scale = 0.0;
    for (int i = 0; i < 9; ++i) scale = fmax(scale, fabs(A[i]));
    if (!(scale > 0.0)) {     // H == 0 (single inlier): LAPACK returns U=V=I  -> R = I
        for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
        if (scale != scale) for (int i = 0; i < 9; ++i) R[i] = scale;   // NaN propagates like the reference
        return;
    }
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int r = 0; r < 3; ++r) {
                    alpha += A[r * 3 + p] * A[r * 3 + p];
                    beta += A[r * 3 + q] * A[r * 3 + q];
                    gamma += A[r * 3 + p] * A[r * 3 + q];
                }
                if (gamma == 0.0) continue;
                off = fmax(off, fabs(gamma) / sqrt(alpha * beta + 1e-300));
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int r = 0; r < 3; ++r) {
                    const double ap = A[r * 3 + p], aq = A[r * 3 + q];
                    A[r * 3 + p] = c * ap - s * aq;
                    A[r * 3 + q] = s * ap + c * aq;
                    const double vp = V[r * 3 + p], vq = V[r * 3 + q];
                    V[r * 3 + p] = c * vp - s * vq;
                    V[r * 3 + q] = s * vp + c * vq;
                }
            }
        if (off < 1e-15) break;
    }
    double U[9], nrm[3];
    for (int c = 0; c < 3; ++c) {
        nrm[c] = sqrt(A[c] * A[c] + A[3 + c] * A[3 + c] + A[6 + c] * A[6 + c]);
    }
    const double tol = 1e-13 * fmax(nrm[0], fmax(nrm[1], nrm[2]));
    int good[3], ngood = 0;
    for (int c = 0; c < 3; ++c) {
        good[c] = nrm[c] > tol;
        ngood += good[c];
        for (int r = 0; r < 3; ++r) U[r * 3 + c] = good[c] ? A[r * 3 + c] / nrm[c] : 0.0;
    }
    if (ngood == 2) {        // complete the basis so that U V^T is a proper rotation
        int z = !good[0] ? 0 : (!good[1] ? 1 : 2);
        int a = (z + 1) % 3, b = (z + 2) % 3;
        double cx = U[3 + a] * U[6 + b] - U[6 + a] * U[3 + b];
        double cy = U[6 + a] * U[0 + b] - U[0 + a] * U[6 + b];
        double cz = U[0 + a] * U[3 + b] - U[3 + a] * U[0 + b];
        // det(V) sign decides the orientation of the completed column
        const double detV = V[0] * (V[4] * V[8] - V[5] * V[7]) - V[1] * (V[3] * V[8] - V[5] * V[6]) + V[2] * (V[3] * V[7] - V[4] * V[6]);
        const double sg = detV >= 0 ? 1.0 : -1.0;
        U[0 + z] = sg * cx; U[3 + z] = sg * cy; U[6 + z] = sg * cz;
    } else if (ngood < 2) {  // rank <= 1: no unique answer; fall back to identity like the H==0 case
        for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
        return;
    }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) R[r * 3 + c] = U[r * 3] * V[c * 3] + U[r * 3 + 1] * V[c * 3 + 1] + U[r * 3 + 2] * V[c * 3 + 2];
}

__device__ __forceinline__ double block_sum(double v, double *red, int tid) {
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// F32W: the match scores are float32 (rotation-coherence matcher): the reference normalises them in float32 -- scores / np.sum(scores),
// test/estimator.py:50, numpy's pairwise float32 sum over the compacted inliers and a float32 division per weight -- before the float64
// centroid / cross-covariance sums, and so does this body (wave 0 rebuilds the sum; see np_pairwise_sum_wave above).
template <bool F32W>
__device__ __forceinline__ void refine_body(const double *__restrict__ k0, const double *__restrict__ k1,
                                            const double *__restrict__ w, int M, const double *__restrict__ T_in,
                                            int t_stride, const double *__restrict__ Trans,
                                            const int64_t *__restrict__ hyp_rows, const int32_t *__restrict__ best,
                                            double thr2, double *__restrict__ T_out, double *__restrict__ stats,
                                            float *npbuf = nullptr, NpSumScratch *npsc = nullptr) {
    __shared__ double red[4];
    __shared__ double Ts[12];
    __shared__ float S32s;
    const int tid = threadIdx.x;
    if (tid < 12) {
        double v;
        if (best) {
            const int bh = *best;
            if (bh < 0) v = __builtin_nan("");
            else {
                const size_t row = hyp_rows ? (size_t)hyp_rows[bh] : (size_t)bh;
                v = Trans[row * 12 + tid];
            }
        } else {
            v = T_in[(tid / 4) * t_stride + (tid % 4)];
        }
        Ts[tid] = v;
    }
    __syncthreads();
    double T[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) T[q] = Ts[q];
    float S32 = 1.f;
    if (F32W) {
        if (tid < 64) {
            const float v = np_sum_f32_of_inliers<NP_CHUNK>(w, M, npbuf, npsc, [&](int i) {
                return point_inlier(T, k0[3 * i], k0[3 * i + 1], k0[3 * i + 2], k1[3 * i], k1[3 * i + 1], k1[3 * i + 2], thr2); }, nullptr);
            if (tid == 0) S32s = v;
        }
        __syncthreads();
        S32 = S32s;
    }
    // the weight a correspondence enters the sums with: w_i (normalised by the sum afterwards) / float32(w_i) / float32 sum (already normalised)
    auto weight = [&](int i) -> double { return F32W ? (double)__fdiv_rn((float)w[i], S32) : weight_of(w, i); };
    // pass 1: sum of weights and weighted sums of the inlier keypoints
    double sw = 0, a0 = 0, a1 = 0, a2 = 0, b0 = 0, b1 = 0, b2 = 0;
    for (int i = tid; i < M; i += 256) {
        if (point_inlier(T, k0[3 * i], k0[3 * i + 1], k0[3 * i + 2], k1[3 * i], k1[3 * i + 1], k1[3 * i + 2], thr2)) {
            const double wi = weight(i);
            sw += wi;
            a0 += wi * k0[3 * i]; a1 += wi * k0[3 * i + 1]; a2 += wi * k0[3 * i + 2];
            b0 += wi * k1[3 * i]; b1 += wi * k1[3 * i + 1]; b2 += wi * k1[3 * i + 2];
        }
    }
    double cnt = 0;
    for (int i = tid; i < M; i += 256)
        if (point_inlier(T, k0[3 * i], k0[3 * i + 1], k0[3 * i + 2], k1[3 * i], k1[3 * i + 1], k1[3 * i + 2], thr2)) cnt += 1.0;
    const int n_inl = (int)block_sum(cnt, red, tid);
    sw = block_sum(sw, red, tid);
    if (F32W) sw = n_inl > 0 ? 1.0 : 0.0;                          // the weights are normalised already (their float32 sum is stats[15])
    double c0x = block_sum(a0, red, tid) / sw, c0y = block_sum(a1, red, tid) / sw, c0z = block_sum(a2, red, tid) / sw;
    double c1x = block_sum(b0, red, tid) / sw, c1y = block_sum(b1, red, tid) / sw, c1z = block_sum(b2, red, tid) / sw;
    // pass 2: H = sum w' (k0-c0)(k1-c1)^T
    double h[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = tid; i < M; i += 256) {
        if (point_inlier(T, k0[3 * i], k0[3 * i + 1], k0[3 * i + 2], k1[3 * i], k1[3 * i + 1], k1[3 * i + 2], thr2)) {
            const double wi = F32W ? weight(i) : weight_of(w, i) / sw;
            const double ax = k0[3 * i] - c0x, ay = k0[3 * i + 1] - c0y, az = k0[3 * i + 2] - c0z;
            const double bx = k1[3 * i] - c1x, by = k1[3 * i + 1] - c1y, bz = k1[3 * i + 2] - c1z;
            h[0] += wi * ax * bx; h[1] += wi * ax * by; h[2] += wi * ax * bz;
            h[3] += wi * ay * bx; h[4] += wi * ay * by; h[5] += wi * ay * bz;
            h[6] += wi * az * bx; h[7] += wi * az * by; h[8] += wi * az * bz;
        }
    }
    double Hm[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) Hm[q] = block_sum(h[q], red, tid);
    if (tid == 0 && n_inl >= 1 && n_inl <= 7) {
        // Few inliers (a failed registration): the cross-covariance is (nearly) rank-deficient and the reference's
        // U V^T is decided by rounding noise, so reproduce its statistics bit for bit -- numpy's sequential sums for
        // n < 8 and the k-ordered FMA chain of its dgemm (refiner.center_cal / SVDR_w, test/estimator.py:32-43).
        int id[7];
        int n = 0;
        for (int i = 0; i < M && n < 7; ++i)
            if (point_inlier(T, k0[3 * i], k0[3 * i + 1], k0[3 * i + 2], k1[3 * i], k1[3 * i + 1], k1[3 * i + 2], thr2)) id[n++] = i;
        double ssum = 0.0;
        double sk[7], c0[3] = {0, 0, 0}, c1[3] = {0, 0, 0};
        if (F32W) {                                                // float32 scores: the sequential float32 sum and float32 quotients
            float s32 = 0.f;
            for (int q = 0; q < n; ++q) s32 = __fadd_rn(s32, (float)w[id[q]]);
            for (int q = 0; q < n; ++q) sk[q] = (double)__fdiv_rn((float)w[id[q]], s32);
            ssum = 1.0;
        } else {
            for (int q = 0; q < n; ++q) ssum = ssum + weight_of(w, id[q]);
            for (int q = 0; q < n; ++q) sk[q] = weight_of(w, id[q]) / ssum;
        }
        for (int q = 0; q < n; ++q)
            for (int d = 0; d < 3; ++d) {
                const double pa = k0[3 * id[q] + d] * sk[q], pb = k1[3 * id[q] + d] * sk[q];
                c0[d] = q == 0 ? pa : c0[d] + pa;
                c1[d] = q == 0 ? pb : c1[d] + pb;
            }
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                double acc = 0.0;
                for (int q = 0; q < n; ++q) acc = fma((k0[3 * id[q] + r] - c0[r]) * sk[q], k1[3 * id[q] + c] - c1[c], acc);
                Hm[r * 3 + c] = acc;
            }
        sw = ssum; c0x = c0[0]; c0y = c0[1]; c0z = c0[2]; c1x = c1[0]; c1y = c1[1]; c1z = c1[2];
    }
    if (tid == 0) {
        double R[9];
        polar_uvt(Hm, R);
        if (!(sw == sw) || sw == 0.0) {          // zero inliers: the reference's 0/0 -> NaN everywhere
            for (int q = 0; q < 9; ++q) R[q] = __builtin_nan("");
        }
        for (int r = 0; r < 3; ++r) {
            T_out[r * 4 + 0] = R[r * 3]; T_out[r * 4 + 1] = R[r * 3 + 1]; T_out[r * 4 + 2] = R[r * 3 + 2];
        }
        T_out[3] = c0x - (c1x * R[0] + c1y * R[1] + c1z * R[2]);
        T_out[7] = c0y - (c1x * R[3] + c1y * R[4] + c1z * R[5]);
        T_out[11] = c0z - (c1x * R[6] + c1y * R[7] + c1z * R[8]);
        T_out[12] = 0; T_out[13] = 0; T_out[14] = 0; T_out[15] = 1;
        if (stats) {      // [H (9), c0 (3), c1 (3), sum of weights]: lets the host redo the 3x3 SVD with LAPACK when H is rank-deficient
            for (int q = 0; q < 9; ++q) stats[q] = Hm[q];
            stats[9] = c0x; stats[10] = c0y; stats[11] = c0z; stats[12] = c1x; stats[13] = c1y; stats[14] = c1z;
            stats[15] = F32W ? (double)S32 : sw;
        }
    }
}

template <bool F32W>
__global__ __launch_bounds__(256) void refine_kernel(const double *__restrict__ k0, const double *__restrict__ k1,
                                                     const double *__restrict__ w, int M, const double *__restrict__ T_in,
                                                     int t_stride, const double *__restrict__ Trans,
                                                     const int64_t *__restrict__ hyp_rows, const int32_t *__restrict__ best,
                                                     double thr2, double *__restrict__ T_out, double *__restrict__ stats) {
    __shared__ float npbuf[F32W ? NP_CHUNK : 1];
    __shared__ NpSumScratch npsc;
    refine_body<F32W>(k0, k1, w, M, T_in, t_stride, Trans, hyp_rows, best, thr2, T_out, stats, npbuf, &npsc);
}

// ---------------------------------------------------------------------------------------------------------------
// Batched estimator tail: every pair of a scene in five launches (gather, score, first-best, refine x2).  A pair's kernels are
// tiny (refine is ONE workgroup, the scoring ~250), so per-pair launches leave the chip idle; here blockIdx.y is the pair.
struct RansacTask {                // mirrors roreg_ransac_task (include/roreg_hip.h)
    const double *keys0, *keys1;   // the two clouds' keypoints [*,3]
    const int64_t *matches;        // [M,2] interleaved (row in cloud 0, row in cloud 1)
    const double *w;               // [M] or null (= ones)
    const double *Trans;           // [*,3,4] local transforms
    const int64_t *hyp_rows;       // [H] rows of Trans, or null (= identity)
    int32_t M, H;
    int64_t koff;                  // first row of this pair in the gathered keypoint workspace
};

__global__ __launch_bounds__(256) void ransac_gather_batch_kernel(const RansacTask *__restrict__ tasks, double *__restrict__ k0_all,
                                                                  double *__restrict__ k1_all) {
    const RansacTask t = tasks[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= t.M) return;
    const int64_t r0 = t.matches[2 * i], r1 = t.matches[2 * i + 1];
    double *a = k0_all + (t.koff + i) * 3, *b = k1_all + (t.koff + i) * 3;
    a[0] = t.keys0[3 * r0]; a[1] = t.keys0[3 * r0 + 1]; a[2] = t.keys0[3 * r0 + 2];
    b[0] = t.keys1[3 * r1]; b[1] = t.keys1[3 * r1 + 1]; b[2] = t.keys1[3 * r1 + 2];
}

__global__ __launch_bounds__(256) void ransac_score_batch_kernel(const RansacTask *__restrict__ tasks, const double *__restrict__ k0_all,
                                                                 const double *__restrict__ k1_all, double thr2, int pitch_h,
                                                                 double *__restrict__ overlap_all) {
    const RansacTask t = tasks[blockIdx.y];
    if ((int)blockIdx.x * 4 >= t.H) return;
    ransac_score_body(k0_all + t.koff * 3, k1_all + t.koff * 3, t.w, t.M, t.Trans, t.hyp_rows, t.H, thr2,
                      blockIdx.x * 4 + (threadIdx.x >> 6), overlap_all + (size_t)blockIdx.y * pitch_h, nullptr);
}

template <int WAVES, int CAP>
__global__ __launch_bounds__(WAVES * 64) void ransac_score_batch_f32_kernel(const RansacTask *__restrict__ tasks, const double *__restrict__ k0_all,
                                                                           const double *__restrict__ k1_all, double thr2, int pitch_h,
                                                                           double *__restrict__ overlap_all) {
    __shared__ float buf[WAVES][CAP];
    __shared__ NpSumScratch sc[WAVES];
    const RansacTask t = tasks[blockIdx.y];
    if ((int)blockIdx.x * WAVES >= t.H) return;
    const int wv = threadIdx.x >> 6;
    if (t.w)
        ransac_score_body_f32<CAP>(k0_all + t.koff * 3, k1_all + t.koff * 3, t.w, t.M, t.Trans, t.hyp_rows, t.H, thr2, blockIdx.x * WAVES + wv,
                                   overlap_all + (size_t)blockIdx.y * pitch_h, nullptr, buf[wv], &sc[wv]);
    else                                                           // no scores = ones: any order gives the exact integer count
        ransac_score_body(k0_all + t.koff * 3, k1_all + t.koff * 3, t.w, t.M, t.Trans, t.hyp_rows, t.H, thr2, blockIdx.x * WAVES + wv,
                          overlap_all + (size_t)blockIdx.y * pitch_h, nullptr);
}

__global__ __launch_bounds__(256) void first_best_batch_kernel(const RansacTask *__restrict__ tasks, int pitch_h,
                                                               const double *__restrict__ overlap_all, int32_t *__restrict__ best_all) {
    first_best_body(overlap_all + (size_t)blockIdx.x * pitch_h, tasks[blockIdx.x].H, best_all + blockIdx.x);
}

template <bool F32W>
__global__ __launch_bounds__(256) void refine_batch_kernel(const RansacTask *__restrict__ tasks, const double *__restrict__ k0_all,
                                                           const double *__restrict__ k1_all, const double *__restrict__ T_in_all,
                                                           const int32_t *__restrict__ best_all, double thr2,
                                                           double *__restrict__ T_out_all, double *__restrict__ stats_all,
                                                           const int32_t *__restrict__ sel) {
    __shared__ float npbuf[F32W ? NP_CHUNK : 1];
    __shared__ NpSumScratch npsc;
    // sel: the launch refines the tasks sel[0..grid) only (T_in / outputs are indexed by the position in sel); nullptr = every task
    const int ti = sel ? sel[blockIdx.x] : (int)blockIdx.x;
    const RansacTask t = tasks[ti];
    if (F32W && t.w)
        refine_body<true>(k0_all + t.koff * 3, k1_all + t.koff * 3, t.w, t.M, T_in_all ? T_in_all + (size_t)blockIdx.x * 16 : nullptr, 4, t.Trans,
                          t.hyp_rows, T_in_all ? nullptr : best_all + ti, thr2, T_out_all + (size_t)blockIdx.x * 16,
                          stats_all + (size_t)blockIdx.x * 16, npbuf, &npsc);
    else
    refine_body<false>(k0_all + t.koff * 3, k1_all + t.koff * 3, t.w, t.M, T_in_all ? T_in_all + (size_t)blockIdx.x * 16 : nullptr, 4, t.Trans,
                t.hyp_rows, T_in_all ? nullptr : best_all + ti, thr2, T_out_all + (size_t)blockIdx.x * 16,
                stats_all + (size_t)blockIdx.x * 16);
}

}  // namespace

extern "C" int roreg_ransac_score(const double *k0, const double *k1, const double *w, int w_f32, int M, const double *Trans,
                                  const int64_t *hyp_rows, int H, double ird, double *overlap_out, int32_t *best_out,
                                  uint8_t *mask_out, void *stream) {
    ROREG_REQUIRE(k0 && k1 && w && Trans && overlap_out && M > 0 && H >= 0, "roreg_ransac_score: bad arguments");
    hipStream_t s = roreg::as_stream(stream);
    if (H > 0 && w_f32 && M <= 4096)
        hipLaunchKernelGGL((ransac_score_f32_kernel<4, 4096>), dim3((H + 3) / 4), dim3(256), 0, s, k0, k1, w, M, Trans, hyp_rows, H, ird * ird,
                           overlap_out, mask_out);
    else if (H > 0 && w_f32)
        hipLaunchKernelGGL((ransac_score_f32_kernel<2, NP_CHUNK>), dim3((H + 1) / 2), dim3(128), 0, s, k0, k1, w, M, Trans, hyp_rows, H, ird * ird,
                           overlap_out, mask_out);
    else if (H > 0)
        hipLaunchKernelGGL(ransac_score_kernel, dim3((H + 3) / 4), dim3(256), 0, s, k0, k1, w, M, Trans, hyp_rows, H, ird * ird,
                           overlap_out, mask_out);
    if (best_out) hipLaunchKernelGGL(first_best_kernel, dim3(1), dim3(256), 0, s, overlap_out, H, best_out);
    ROREG_CHECK_LAUNCH("roreg_ransac_score");
    return 0;
}

extern "C" int roreg_refine(const double *k0, const double *k1, const double *w, int w_f32, int M, const double *T_in, int t_in_stride,
                            const double *Trans, const int64_t *hyp_rows, const int32_t *best, double dist, double *T_out,
                            double *stats_out, void *stream) {
    ROREG_REQUIRE(k0 && k1 && w && T_out && M > 0, "roreg_refine: bad arguments");
    ROREG_REQUIRE((best && Trans) || T_in, "roreg_refine: need T_in or (Trans, best)");
    if (w_f32)
        hipLaunchKernelGGL(refine_kernel<true>, dim3(1), dim3(256), 0, roreg::as_stream(stream), k0, k1, w, M, T_in, t_in_stride, Trans,
                           hyp_rows, best, dist * dist, T_out, stats_out);
    else
        hipLaunchKernelGGL(refine_kernel<false>, dim3(1), dim3(256), 0, roreg::as_stream(stream), k0, k1, w, M, T_in, t_in_stride, Trans,
                           hyp_rows, best, dist * dist, T_out, stats_out);
    ROREG_CHECK_LAUNCH("roreg_refine");
    return 0;
}

extern "C" size_t roreg_ransac_batch_workspace(int n_tasks, long long total_M, int max_H) {
    return ((size_t)total_M * 6 + (size_t)n_tasks * (size_t)(max_H > 0 ? max_H : 1)) * sizeof(double);
}

extern "C" int roreg_ransac_batch(const roreg_ransac_task *tasks_dev, int n_tasks, long long total_M, int max_M, int max_H, double ird, int w_f32,
                                  int32_t *best_out, double *T1_out, double *stats1_out, double *T2_out, double *stats2_out,
                                  void *workspace, size_t workspace_bytes, void *stream) {
    if (n_tasks == 0) return 0;
    ROREG_REQUIRE(tasks_dev && best_out && T1_out && stats1_out && T2_out && stats2_out && workspace && n_tasks > 0 && max_M > 0 && max_H >= 0,
                  "roreg_ransac_batch: bad arguments");
    ROREG_REQUIRE(workspace_bytes >= roreg_ransac_batch_workspace(n_tasks, total_M, max_H), "roreg_ransac_batch: workspace too small");
    static_assert(sizeof(roreg_ransac_task) == sizeof(RansacTask), "roreg_ransac_task layout");
    hipStream_t s = roreg::as_stream(stream);
    const RansacTask *tasks = reinterpret_cast<const RansacTask *>(tasks_dev);
    double *k0 = reinterpret_cast<double *>(workspace), *k1 = k0 + (size_t)total_M * 3, *overlap = k1 + (size_t)total_M * 3;
    const int pitch_h = max_H > 0 ? max_H : 1;
    hipLaunchKernelGGL(ransac_gather_batch_kernel, dim3((max_M + 255) / 256, n_tasks), dim3(256), 0, s, tasks, k0, k1);
    if (max_H > 0)
    {
        roreg::ProfScope prof(roreg::PROF_RANSAC_SCORE, s);
        if (w_f32 && max_M <= 4096)
            hipLaunchKernelGGL((ransac_score_batch_f32_kernel<4, 4096>), dim3((max_H + 3) / 4, n_tasks), dim3(256), 0, s, tasks, k0, k1, ird * ird, pitch_h, overlap);
        else if (w_f32)
            hipLaunchKernelGGL((ransac_score_batch_f32_kernel<2, NP_CHUNK>), dim3((max_H + 1) / 2, n_tasks), dim3(128), 0, s, tasks, k0, k1, ird * ird, pitch_h, overlap);
        else
            hipLaunchKernelGGL(ransac_score_batch_kernel, dim3((max_H + 3) / 4, n_tasks), dim3(256), 0, s, tasks, k0, k1, ird * ird, pitch_h, overlap);
    }
    hipLaunchKernelGGL(first_best_batch_kernel, dim3(n_tasks), dim3(256), 0, s, tasks, pitch_h, overlap, best_out);
    if (w_f32) {
        hipLaunchKernelGGL(refine_batch_kernel<true>, dim3(n_tasks), dim3(256), 0, s, tasks, k0, k1, (const double *)nullptr, best_out,
                           (2.0 * ird) * (2.0 * ird), T1_out, stats1_out, (const int32_t *)nullptr);
        hipLaunchKernelGGL(refine_batch_kernel<true>, dim3(n_tasks), dim3(256), 0, s, tasks, k0, k1, (const double *)T1_out, (const int32_t *)nullptr,
                           ird * ird, T2_out, stats2_out, (const int32_t *)nullptr);
    } else {
        hipLaunchKernelGGL(refine_batch_kernel<false>, dim3(n_tasks), dim3(256), 0, s, tasks, k0, k1, (const double *)nullptr, best_out,
                           (2.0 * ird) * (2.0 * ird), T1_out, stats1_out, (const int32_t *)nullptr);
        hipLaunchKernelGGL(refine_batch_kernel<false>, dim3(n_tasks), dim3(256), 0, s, tasks, k0, k1, (const double *)T1_out, (const int32_t *)nullptr,
                           ird * ird, T2_out, stats2_out, (const int32_t *)nullptr);
    }
    ROREG_CHECK_LAUNCH("roreg_ransac_batch");
    return 0;
}

// One more refinement (from given transforms) of SOME tasks of an earlier roreg_ransac_batch, in one launch: the engine's second pass over
// the rank-deficient pairs, whose first refinement was closed on the host (LAPACK's choice of U V^T for a rank <= 2 covariance).
extern "C" int roreg_refine_batch(const roreg_ransac_task *tasks_dev, const int32_t *sel_dev, int n_sel, long long total_M, const double *T_in,
                                  double dist, int w_f32, double *T_out, double *stats_out, const void *workspace, void *stream) {
    if (n_sel == 0) return 0;
    ROREG_REQUIRE(tasks_dev && sel_dev && T_in && T_out && stats_out && workspace && n_sel > 0 && total_M >= 0, "roreg_refine_batch: bad arguments");
    hipStream_t s = roreg::as_stream(stream);
    const RansacTask *tasks = reinterpret_cast<const RansacTask *>(tasks_dev);
    const double *k0 = reinterpret_cast<const double *>(workspace), *k1 = k0 + (size_t)total_M * 3;
    if (w_f32)
        hipLaunchKernelGGL(refine_batch_kernel<true>, dim3(n_sel), dim3(256), 0, s, tasks, k0, k1, T_in, (const int32_t *)nullptr, dist * dist, T_out,
                           stats_out, sel_dev);
    else
        hipLaunchKernelGGL(refine_batch_kernel<false>, dim3(n_sel), dim3(256), 0, s, tasks, k0, k1, T_in, (const int32_t *)nullptr, dist * dist, T_out,
                           stats_out, sel_dev);
    ROREG_CHECK_LAUNCH("roreg_refine_batch");
    return 0;
}
