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

__device__ __forceinline__ void refine_body(const double *__restrict__ k0, const double *__restrict__ k1,
                                            const double *__restrict__ w, int M, const double *__restrict__ T_in,
                                            int t_stride, const double *__restrict__ Trans,
                                            const int64_t *__restrict__ hyp_rows, const int32_t *__restrict__ best,
                                            double thr2, double *__restrict__ T_out, double *__restrict__ stats) {
    __shared__ double red[4];
    __shared__ double Ts[12];
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
    // pass 1: sum of weights and weighted sums of the inlier keypoints
    double sw = 0, a0 = 0, a1 = 0, a2 = 0, b0 = 0, b1 = 0, b2 = 0;
    for (int i = tid; i < M; i += 256) {
        if (point_inlier(T, k0[3 * i], k0[3 * i + 1], k0[3 * i + 2], k1[3 * i], k1[3 * i + 1], k1[3 * i + 2], thr2)) {
            const double wi = weight_of(w, i);
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
    double c0x = block_sum(a0, red, tid) / sw, c0y = block_sum(a1, red, tid) / sw, c0z = block_sum(a2, red, tid) / sw;
    double c1x = block_sum(b0, red, tid) / sw, c1y = block_sum(b1, red, tid) / sw, c1z = block_sum(b2, red, tid) / sw;
    // pass 2: H = sum w' (k0-c0)(k1-c1)^T
    double h[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = tid; i < M; i += 256) {
        if (point_inlier(T, k0[3 * i], k0[3 * i + 1], k0[3 * i + 2], k1[3 * i], k1[3 * i + 1], k1[3 * i + 2], thr2)) {
            const double wi = weight_of(w, i) / sw;
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
        for (int q = 0; q < n; ++q) ssum = ssum + weight_of(w, id[q]);
        double sk[7], c0[3] = {0, 0, 0}, c1[3] = {0, 0, 0};
        for (int q = 0; q < n; ++q) sk[q] = weight_of(w, id[q]) / ssum;
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
            stats[9] = c0x; stats[10] = c0y; stats[11] = c0z; stats[12] = c1x; stats[13] = c1y; stats[14] = c1z; stats[15] = sw;
        }
    }
}

__global__ __launch_bounds__(256) void refine_kernel(const double *__restrict__ k0, const double *__restrict__ k1,
                                                     const double *__restrict__ w, int M, const double *__restrict__ T_in,
                                                     int t_stride, const double *__restrict__ Trans,
                                                     const int64_t *__restrict__ hyp_rows, const int32_t *__restrict__ best,
                                                     double thr2, double *__restrict__ T_out, double *__restrict__ stats) {
    refine_body(k0, k1, w, M, T_in, t_stride, Trans, hyp_rows, best, thr2, T_out, stats);
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

__global__ __launch_bounds__(256) void first_best_batch_kernel(const RansacTask *__restrict__ tasks, int pitch_h,
                                                               const double *__restrict__ overlap_all, int32_t *__restrict__ best_all) {
    first_best_body(overlap_all + (size_t)blockIdx.x * pitch_h, tasks[blockIdx.x].H, best_all + blockIdx.x);
}

__global__ __launch_bounds__(256) void refine_batch_kernel(const RansacTask *__restrict__ tasks, const double *__restrict__ k0_all,
                                                           const double *__restrict__ k1_all, const double *__restrict__ T_in_all,
                                                           const int32_t *__restrict__ best_all, double thr2,
                                                           double *__restrict__ T_out_all, double *__restrict__ stats_all) {
    const RansacTask t = tasks[blockIdx.x];
    refine_body(k0_all + t.koff * 3, k1_all + t.koff * 3, t.w, t.M, T_in_all ? T_in_all + (size_t)blockIdx.x * 16 : nullptr, 4, t.Trans,
                t.hyp_rows, T_in_all ? nullptr : best_all + blockIdx.x, thr2, T_out_all + (size_t)blockIdx.x * 16,
                stats_all + (size_t)blockIdx.x * 16);
}

}  // namespace

extern "C" int roreg_ransac_score(const double *k0, const double *k1, const double *w, int M, const double *Trans,
                                  const int64_t *hyp_rows, int H, double ird, double *overlap_out, int32_t *best_out,
                                  uint8_t *mask_out, void *stream) {
    ROREG_REQUIRE(k0 && k1 && w && Trans && overlap_out && M > 0 && H >= 0, "roreg_ransac_score: bad arguments");
    hipStream_t s = roreg::as_stream(stream);
    if (H > 0)
        hipLaunchKernelGGL(ransac_score_kernel, dim3((H + 3) / 4), dim3(256), 0, s, k0, k1, w, M, Trans, hyp_rows, H, ird * ird,
                           overlap_out, mask_out);
    if (best_out) hipLaunchKernelGGL(first_best_kernel, dim3(1), dim3(256), 0, s, overlap_out, H, best_out);
    ROREG_CHECK_LAUNCH("roreg_ransac_score");
    return 0;
}

extern "C" int roreg_refine(const double *k0, const double *k1, const double *w, int M, const double *T_in, int t_in_stride,
                            const double *Trans, const int64_t *hyp_rows, const int32_t *best, double dist, double *T_out,
                            double *stats_out, void *stream) {
    ROREG_REQUIRE(k0 && k1 && w && T_out && M > 0, "roreg_refine: bad arguments");
    ROREG_REQUIRE((best && Trans) || T_in, "roreg_refine: need T_in or (Trans, best)");
    hipLaunchKernelGGL(refine_kernel, dim3(1), dim3(256), 0, roreg::as_stream(stream), k0, k1, w, M, T_in, t_in_stride, Trans,
                       hyp_rows, best, dist * dist, T_out, stats_out);
    ROREG_CHECK_LAUNCH("roreg_refine");
    return 0;
}

extern "C" size_t roreg_ransac_batch_workspace(int n_tasks, long long total_M, int max_H) {
    return ((size_t)total_M * 6 + (size_t)n_tasks * (size_t)(max_H > 0 ? max_H : 1)) * sizeof(double);
}

extern "C" int roreg_ransac_batch(const roreg_ransac_task *tasks_dev, int n_tasks, long long total_M, int max_M, int max_H, double ird,
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
        hipLaunchKernelGGL(ransac_score_batch_kernel, dim3((max_H + 3) / 4, n_tasks), dim3(256), 0, s, tasks, k0, k1, ird * ird, pitch_h, overlap);
    }
    hipLaunchKernelGGL(first_best_batch_kernel, dim3(n_tasks), dim3(256), 0, s, tasks, pitch_h, overlap, best_out);
    hipLaunchKernelGGL(refine_batch_kernel, dim3(n_tasks), dim3(256), 0, s, tasks, k0, k1, (const double *)nullptr, best_out,
                       (2.0 * ird) * (2.0 * ird), T1_out, stats1_out);
    hipLaunchKernelGGL(refine_batch_kernel, dim3(n_tasks), dim3(256), 0, s, tasks, k0, k1, (const double *)T1_out, (const int32_t *)nullptr,
                       ird * ird, T2_out, stats2_out);
    ROREG_CHECK_LAUNCH("roreg_ransac_batch");
    return 0;
}
