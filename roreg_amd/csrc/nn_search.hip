// Brute-force nearest-neighbour search with the reference's exact distance formula, and the mutual check.
//
// The reference (utils/knn_search.py:17-24) evaluates d = sqrt(sum_f (s_f - t_f)^2 + 1e-7) by explicit
// differences in float32 and takes the first minimum.  Correspondence indices must be bit-exact, so the
// kernel evaluates exactly that formula (f order, no FMA contraction, correctly rounded sqrt) on the VALU:
// at N=5000, F=32 it is 2.4 GFLOP per direction, a few tens of microseconds -- it is not worth trading the
// exact formula for a matrix-core dot-product expansion whose rounding differs.
//
// Decomposition: one thread per source row (its descriptor lives in 32 VGPRs); the target set is split into
// grid.y slices so that 5000 sources still fill the chip.  Target rows are wave-uniform, so they are fetched
// through the scalar path (s_load) and feed the VALU as SGPR operands.  Slices are merged with a single
// 64-bit atomicMin on (float_bits(d) << 32 | index): d >= 0, so the packed integer order is (d, index) order,
// which is precisely "first minimum wins".
#include "common.h"

// Bit-exactness contract: no fused multiply-add may be formed from separate * and + in this file (hipcc's
// default is -ffp-contract=fast, and the __f*_rn helpers are plain operators); sqrtf and / are correctly
// rounded under hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt.
#pragma clang fp contract(off)

namespace {

__global__ __launch_bounds__(256) void fill_u64_kernel(unsigned long long *p, unsigned long long v, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

// SQ: dist_type 'SquareL2' (utils/knn_search.py:21-22): the key is the squared distance itself (no 1e-7, no root), so two targets whose
// roots round to the same float are still told apart by their radicands.
template <int F, bool SQ>
__global__ __launch_bounds__(256) void nn_search_kernel(const float *__restrict__ src, const int64_t *__restrict__ src_rows, int m,
                                                        const float *__restrict__ tgt, const int64_t *__restrict__ tgt_rows, int n,
                                                        int slice, unsigned long long *__restrict__ packed) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int ii = i < m ? i : m - 1;
    const size_t srow = src_rows ? (size_t)src_rows[ii] : (size_t)ii;
    float s[F];
#pragma unroll
    for (int f = 0; f < F; ++f) s[f] = src[srow * F + f];
    const int j0 = blockIdx.y * slice;
    const int j1 = min(j0 + slice, n);
    float best_x = __builtin_inff(), best_d = __builtin_inff();
    int best_j = 0x7fffffff;
    for (int j = j0; j < j1; ++j) {
        const size_t trow = tgt_rows ? (size_t)tgt_rows[j] : (size_t)j;     // wave-uniform
        const float *t = tgt + trow * F;
        float acc = 0.f;
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const float d = __fsub_rn(s[f], t[f]);
            acc = __fadd_rn(acc, __fmul_rn(d, d));
        }
        if (SQ) {
            if (acc < best_d) { best_d = acc; best_j = j; }
            continue;
        }
        const float x = __fadd_rn(acc, 1e-7f);
        if (x < best_x) {                      // sqrt is monotone: only a smaller radicand can give a smaller d
            const float d = sqrtf(x);
            if (d < best_d) { best_d = d; best_j = j; }
            best_x = x;
        }
    }
    if (i < m && best_j != 0x7fffffff) {
        const unsigned long long key = ((unsigned long long)__float_as_uint(best_d) << 32) | (unsigned)best_j;
        atomicMin(&packed[i], key);
    }
}

__global__ __launch_bounds__(256) void nn_unpack_kernel(const unsigned long long *__restrict__ packed, int m,
                                                        int64_t *__restrict__ idx, float *__restrict__ dist) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const unsigned long long k = packed[i];
    idx[i] = (int64_t)(k & 0xffffffffu);
    if (dist) dist[i] = __uint_as_float((unsigned)(k >> 32));
}

// k nearest with k <= 8: each thread keeps a sorted list; full scan (used for F=3 NMS neighbourhoods).
template <int F, bool SQ>
__global__ __launch_bounds__(256) void knn_search_kernel(const float *__restrict__ src, int m, const float *__restrict__ tgt, int n,
                                                         int k, int64_t *__restrict__ idx_out, float *__restrict__ dist_out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int ii = i < m ? i : m - 1;
    float s[F];
#pragma unroll
    for (int f = 0; f < F; ++f) s[f] = src[(size_t)ii * F + f];
    float bd[8];
    int bj[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { bd[q] = __builtin_inff(); bj[q] = -1; }
    for (int j = 0; j < n; ++j) {
        const float *t = tgt + (size_t)j * F;
        float acc = 0.f;
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const float d = __fsub_rn(s[f], t[f]);
            acc = __fadd_rn(acc, __fmul_rn(d, d));
        }
        float d = SQ ? acc : sqrtf(__fadd_rn(acc, 1e-7f));
        int dj = j;
        // insertion into the sorted list: the new entry goes in front of the first strictly larger one (an equal distance keeps the earlier
        // index ahead), and from there on every entry moves down one place -- unconditionally: an entry pushed down must not be
        // compared again, or it would slip behind a later entry of the SAME distance (round 4: exactly tied neighbours came out in
        // reverse index order whenever a closer point was found after them)
        bool ins = false;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (q < k && (ins || d < bd[q])) {
                const float td = bd[q]; const int tj = bj[q];
                bd[q] = d; bj[q] = dj; d = td; dj = tj;
                ins = true;
            }
        }
    }
    if (i < m)
        for (int q = 0; q < k; ++q) {
            idx_out[(size_t)i * k + q] = bj[q];
            if (dist_out) dist_out[(size_t)i * k + q] = bd[q];
        }
}


// Any feature width, k <= 32 (round 6): the reference's modified_knn_matcher accepts every width and every k (utils/knn_search.py:13-162); the
// tuned kernels above are built for the pipeline's widths 3 and 32 and k <= 8.  This one only has to be CORRECT: one thread per source row, the
// same distance formula in the same order (explicit differences, channels ascending, separate multiply and add, correctly rounded root), the
// same sorted-list insertion (first index wins a tie), row lists optional.  The source row is re-read per target (L1), the list lives in
// registers / scratch.
constexpr int KNN_GENERIC_MAX_K = 32;
template <bool SQ>
__global__ __launch_bounds__(256) void knn_generic_kernel(const float *__restrict__ src, const int64_t *__restrict__ src_rows, int m, const float *__restrict__ tgt,
                                                          const int64_t *__restrict__ tgt_rows, int n, int F, int k, int64_t *__restrict__ idx_out,
                                                          float *__restrict__ dist_out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int ii = i < m ? i : m - 1;
    const float *srow = src + (src_rows ? (size_t)src_rows[ii] : (size_t)ii) * F;
    float bd[KNN_GENERIC_MAX_K];
    int bj[KNN_GENERIC_MAX_K];
#pragma unroll
    for (int q = 0; q < KNN_GENERIC_MAX_K; ++q) { bd[q] = __builtin_inff(); bj[q] = -1; }
    for (int j = 0; j < n; ++j) {
        const float *t = tgt + (tgt_rows ? (size_t)tgt_rows[j] : (size_t)j) * F;       // wave-uniform
        float acc = 0.f;
        for (int f = 0; f < F; ++f) {
            const float d = __fsub_rn(srow[f], t[f]);
            acc = __fadd_rn(acc, __fmul_rn(d, d));
        }
        float d = SQ ? acc : sqrtf(__fadd_rn(acc, 1e-7f));
        int dj = j;
        if (!(d < bd[k - 1])) continue;                  // (not better than the list's last entry: nothing moves; NaN never enters)
        bool ins = false;
#pragma unroll
        for (int q = 0; q < KNN_GENERIC_MAX_K; ++q) {
            if (q < k && (ins || d < bd[q])) {
                const float td = bd[q]; const int tj = bj[q];
                bd[q] = d; bj[q] = dj; d = td; dj = tj;
                ins = true;
            }
        }
    }
    if (i < m)
        for (int q = 0; q < k; ++q) {
            idx_out[(size_t)i * k + q] = bj[q];
            if (dist_out) dist_out[(size_t)i * k + q] = bd[q];
        }
}

// the same scan over one slice of the targets (grid.y slices fill the chip: one thread per source alone is 20 workgroups at m = 5000);
// per-slice sorted lists go to a workspace and are merged in slice order = index order, so ties still resolve to the first index
// Segments (several clouds per launch, blockIdx.z = cloud): seg_src / seg_tgt [P+1] are row offsets into the stacked point lists, a
// source only sees the targets of its own segment and the indices written are LOCAL to the segment.  seg_src == nullptr: one search.
template <int F, bool SQ>
__global__ __launch_bounds__(256) void knn_slice_kernel(const float *__restrict__ src, int m, const float *__restrict__ tgt, int n, int k, int slice,
                                                        float *__restrict__ pd, int *__restrict__ pj, const int *__restrict__ seg_src,
                                                        const int *__restrict__ seg_tgt, int m_total) {
    if (seg_src) {
        const int a0 = seg_src[blockIdx.z], b0 = seg_tgt[blockIdx.z];
        m = seg_src[blockIdx.z + 1] - a0; n = seg_tgt[blockIdx.z + 1] - b0;
        if ((int)(blockIdx.x * 256) >= m || n <= 0) return;
        src += (size_t)a0 * F; tgt += (size_t)b0 * F;
        slice = (n + gridDim.y - 1) / gridDim.y;
        pd += (size_t)a0 * 8; pj += (size_t)a0 * 8;                // partial lists are laid out [slice][m_total][8]
    } else {
        m_total = m;
    }
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int ii = i < m ? i : m - 1;
    float s[F];
#pragma unroll
    for (int f = 0; f < F; ++f) s[f] = src[(size_t)ii * F + f];
    float bd[8];
    int bj[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { bd[q] = __builtin_inff(); bj[q] = -1; }
    const int j0 = blockIdx.y * slice, j1 = min(j0 + slice, n);
    for (int j = j0; j < j1; ++j) {
        const float *t = tgt + (size_t)j * F;
        float acc = 0.f;
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const float d = __fsub_rn(s[f], t[f]);
            acc = __fadd_rn(acc, __fmul_rn(d, d));
        }
        float d = SQ ? acc : sqrtf(__fadd_rn(acc, 1e-7f));
        int dj = j;
        bool ins = false;                                          // (see knn_search_kernel: entries pushed down are not compared again)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (q < k && (ins || d < bd[q])) {
                const float td = bd[q]; const int tj = bj[q];
                bd[q] = d; bj[q] = dj; d = td; dj = tj;
                ins = true;
            }
        }
    }
    if (i < m) {
        float *od = pd + ((size_t)blockIdx.y * m_total + i) * 8;
        int *oj = pj + ((size_t)blockIdx.y * m_total + i) * 8;
#pragma unroll
        for (int q = 0; q < 8; ++q) { od[q] = bd[q]; oj[q] = bj[q]; }
    }
}

__global__ __launch_bounds__(256) void knn_merge_kernel(const float *__restrict__ pd, const int *__restrict__ pj, int m, int slices, int k,
                                                        int64_t *__restrict__ idx_out, float *__restrict__ dist_out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    float bd[8];
    int bj[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { bd[q] = __builtin_inff(); bj[q] = -1; }
    for (int sidx = 0; sidx < slices; ++sidx) {
        const float *vd = pd + ((size_t)sidx * m + i) * 8;
        const int *vj = pj + ((size_t)sidx * m + i) * 8;
        for (int c = 0; c < 8; ++c) {
            float d = vd[c];
            int dj = vj[c];
            if (dj < 0) break;
            bool ins = false;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (q < k && (ins || d < bd[q])) {
                    const float td = bd[q]; const int tj = bj[q];
                    bd[q] = d; bj[q] = dj; d = td; dj = tj;
                    ins = true;
                }
            }
        }
    }
    for (int q = 0; q < k; ++q) {
        idx_out[(size_t)i * k + q] = bj[q];
        if (dist_out) dist_out[(size_t)i * k + q] = bd[q];
    }
}

// The full distance matrix of modified_knn_matcher.pdist (utils/knn_search.py:17-24), any feature width: one thread per entry, sources along
// threadIdx.y (their rows are re-read from L1), targets along threadIdx.x (coalesced stores).
template <bool SQ>
__global__ __launch_bounds__(256) void pdist_kernel(const float *__restrict__ A, int m, const float *__restrict__ B, int n, int F, float *__restrict__ out) {
    const int j = blockIdx.x * 64 + (threadIdx.x & 63);
    const int i = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i >= m || j >= n) return;
    const float *a = A + (size_t)i * F, *b = B + (size_t)j * F;
    float acc = 0.f;
    for (int f = 0; f < F; ++f) {
        const float d = __fsub_rn(a[f], b[f]);
        acc = __fadd_rn(acc, __fmul_rn(d, d));
    }
    out[(size_t)i * n + j] = SQ ? acc : sqrtf(__fadd_rn(acc, 1e-7f));
}

// mutual check + ordered compaction by a single workgroup (m <= a few thousand)
__global__ __launch_bounds__(1024) void mutual_kernel(const int64_t *__restrict__ nn01, const int64_t *__restrict__ nn10, int m, int n,
                                                      const int64_t *__restrict__ sample0, const int64_t *__restrict__ sample1,
                                                      int64_t *__restrict__ match_out, int32_t *__restrict__ count_out) {
    __shared__ int wave_cnt[16];
    __shared__ int base;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int start = 0; start < m; start += 1024) {
        const int i = start + tid;
        bool keep = false;
        int64_t j = 0;
        if (i < m) {
            j = nn01[i];
            keep = j >= 0 && j < (int64_t)n && nn10[j] == (int64_t)i;       // an index no comparison ever set (NaN distances) is "unmatched", never dereferenced
        }
        const unsigned long long mask = __ballot(keep);
        const int before = __popcll(mask & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[w] = __popcll(mask);
        __syncthreads();
        int woff = 0, total = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int c = wave_cnt[q];
            if (q < w) woff += c;
            total += c;
        }
        const int b0 = base;
        if (keep) {
            const int pos = b0 + woff + before;
            match_out[2 * pos] = sample0 ? sample0[i] : (int64_t)i;
            match_out[2 * pos + 1] = sample1 ? sample1[j] : j;
        }
        __syncthreads();
        if (tid == 0) base = b0 + total;
        __syncthreads();
    }
    if (tid == 0) *count_out = base;
}


}  // namespace

extern "C" int roreg_nn_search_ex(const float *src, const int64_t *src_rows, int m, const float *tgt, const int64_t *tgt_rows,
                                  int n, int F, int squared, int64_t *idx_out, float *dist_out, uint64_t *scratch, void *stream) {
    if (m == 0) return 0;
    ROREG_REQUIRE(src && tgt && idx_out && scratch && m > 0 && n > 0, "roreg_nn_search: bad arguments");
    ROREG_REQUIRE(F >= 1, "roreg_nn_search: F must be positive (got %d)", F);
    hipStream_t s = roreg::as_stream(stream);
    if (F != 32 && F != 3) {                           // v6: any other width through the generic kernel (k = 1: the first minimum)
        if (squared) hipLaunchKernelGGL(knn_generic_kernel<true>, dim3((m + 255) / 256), dim3(256), 0, s, src, src_rows, m, tgt, tgt_rows, n, F, 1, idx_out, dist_out);
        else hipLaunchKernelGGL(knn_generic_kernel<false>, dim3((m + 255) / 256), dim3(256), 0, s, src, src_rows, m, tgt, tgt_rows, n, F, 1, idx_out, dist_out);
        ROREG_CHECK_LAUNCH("roreg_nn_search");
        return 0;
    }
    unsigned long long *g_packed = reinterpret_cast<unsigned long long *>(scratch);
    hipLaunchKernelGGL(fill_u64_kernel, dim3((m + 255) / 256), dim3(256), 0, s, g_packed, ~0ull, m);
    const int gx = (m + 255) / 256;
    int slices = (2048 + gx - 1) / gx;                 // aim at ~2048 workgroups
    if (slices > n) slices = n;
    const int slice = (n + slices - 1) / slices;
    slices = (n + slice - 1) / slice;
#define ROREG_NN(F_, SQ_) hipLaunchKernelGGL((nn_search_kernel<F_, SQ_>), dim3(gx, slices), dim3(256), 0, s, src, src_rows, m, tgt, tgt_rows, n, slice, g_packed)
    if (F == 32) { if (squared) ROREG_NN(32, true); else ROREG_NN(32, false); }
    else { if (squared) ROREG_NN(3, true); else ROREG_NN(3, false); }
#undef ROREG_NN
    hipLaunchKernelGGL(nn_unpack_kernel, dim3((m + 255) / 256), dim3(256), 0, s, g_packed, m, idx_out, dist_out);
    ROREG_CHECK_LAUNCH("roreg_nn_search");
    return 0;
}

extern "C" int roreg_nn_search(const float *src, const int64_t *src_rows, int m, const float *tgt, const int64_t *tgt_rows,
                               int n, int F, int64_t *idx_out, float *dist_out, uint64_t *scratch, void *stream) {
    return roreg_nn_search_ex(src, src_rows, m, tgt, tgt_rows, n, F, 0, idx_out, dist_out, scratch, stream);
}

extern "C" int roreg_pdist(const float *A, int m, const float *B, int n, int F, int squared, float *out, void *stream) {
    if (m == 0 || n == 0) return 0;
    ROREG_REQUIRE(A && B && out && m > 0 && n > 0 && F > 0, "roreg_pdist: bad arguments");
    const dim3 grid((n + 63) / 64, (m + 3) / 4);
    if (squared) hipLaunchKernelGGL(pdist_kernel<true>, grid, dim3(256), 0, roreg::as_stream(stream), A, m, B, n, F, out);
    else hipLaunchKernelGGL(pdist_kernel<false>, grid, dim3(256), 0, roreg::as_stream(stream), A, m, B, n, F, out);
    ROREG_CHECK_LAUNCH("roreg_pdist");
    return 0;
}

static int knn_slices(int m, int n) {
    const int gx = (m + 255) / 256;
    int slices = (512 + gx - 1) / gx;                  // aim at ~512 workgroups (more slices only lengthen the merge)
    if (slices > (n + 31) / 32) slices = (n + 31) / 32;
    return slices < 1 ? 1 : slices;
}

extern "C" size_t roreg_knn_search_workspace(int m, int n) {
    const int slices = knn_slices(m, n);
    return slices > 1 ? (size_t)slices * m * 8 * (sizeof(float) + sizeof(int)) : 0;
}

extern "C" int roreg_knn_search_ex(const float *src, int m, const float *tgt, int n, int F, int k, int squared, int64_t *idx_out, float *dist_out,
                                   void *workspace, size_t workspace_bytes, void *stream) {
    if (m == 0) return 0;
    ROREG_REQUIRE(src && tgt && idx_out && m > 0 && n > 0, "roreg_knn_search: bad arguments");
    ROREG_REQUIRE(k >= 1 && k <= KNN_GENERIC_MAX_K && k <= n, "roreg_knn_search: k must be in 1..min(32,n) (got %d)", k);
    ROREG_REQUIRE(F >= 1, "roreg_knn_search: F must be positive (got %d)", F);
    hipStream_t s = roreg::as_stream(stream);
    if ((F != 3 && F != 32) || k > 8) {                // v6: other widths / longer lists through the generic kernel
        const int64_t *no_rows = nullptr;
        if (squared) hipLaunchKernelGGL(knn_generic_kernel<true>, dim3((m + 255) / 256), dim3(256), 0, s, src, no_rows, m, tgt, no_rows, n, F, k, idx_out, dist_out);
        else hipLaunchKernelGGL(knn_generic_kernel<false>, dim3((m + 255) / 256), dim3(256), 0, s, src, no_rows, m, tgt, no_rows, n, F, k, idx_out, dist_out);
        ROREG_CHECK_LAUNCH("roreg_knn_search");
        return 0;
    }
    const int slices = knn_slices(m, n);
    const size_t need = roreg_knn_search_workspace(m, n);
    const int *none = nullptr;
    if (slices > 1 && workspace && workspace_bytes >= need) {
        float *pd = reinterpret_cast<float *>(workspace);
        int *pj = reinterpret_cast<int *>(pd + (size_t)slices * m * 8);
        const int slice = (n + slices - 1) / slices;
        const dim3 grid((m + 255) / 256, (n + slice - 1) / slice);
#define ROREG_KS(F_, SQ_) hipLaunchKernelGGL((knn_slice_kernel<F_, SQ_>), grid, dim3(256), 0, s, src, m, tgt, n, k, slice, pd, pj, none, none, m)
        if (F == 3) { if (squared) ROREG_KS(3, true); else ROREG_KS(3, false); }
        else { if (squared) ROREG_KS(32, true); else ROREG_KS(32, false); }
#undef ROREG_KS
        hipLaunchKernelGGL(knn_merge_kernel, dim3((m + 255) / 256), dim3(256), 0, s, pd, pj, m, (int)grid.y, k, idx_out, dist_out);
    } else {                                           // no workspace: the single-pass scan (one thread per source)
#define ROREG_K1(F_, SQ_) hipLaunchKernelGGL((knn_search_kernel<F_, SQ_>), dim3((m + 255) / 256), dim3(256), 0, s, src, m, tgt, n, k, idx_out, dist_out)
        if (F == 3) { if (squared) ROREG_K1(3, true); else ROREG_K1(3, false); }
        else { if (squared) ROREG_K1(32, true); else ROREG_K1(32, false); }
#undef ROREG_K1
    }
    ROREG_CHECK_LAUNCH("roreg_knn_search");
    return 0;
}

extern "C" int roreg_knn_search(const float *src, int m, const float *tgt, int n, int F, int k, int64_t *idx_out, void *workspace,
                                size_t workspace_bytes, void *stream) {
    return roreg_knn_search_ex(src, m, tgt, n, F, k, 0, idx_out, nullptr, workspace, workspace_bytes, stream);
}
extern "C" size_t roreg_knn_search_seg_workspace(long long m_total, int n_seg, int max_m, int max_n) {
    const int gx = (max_m + 255) / 256;
    int slices = (2048 + gx * n_seg - 1) / (gx * n_seg);
    if (slices > (max_n + 31) / 32) slices = (max_n + 31) / 32;
    if (slices < 1) slices = 1;
    return (size_t)slices * (size_t)m_total * 8 * (sizeof(float) + sizeof(int));
}

extern "C" int roreg_knn_search_seg(const float *src, const float *tgt, const int32_t *seg_src, const int32_t *seg_tgt, int n_seg, long long m_total,
                                    int max_m, int max_n, int F, int k, int64_t *idx_out, void *workspace, size_t workspace_bytes, void *stream) {
    if (m_total == 0 || n_seg == 0) return 0;
    ROREG_REQUIRE(src && tgt && seg_src && seg_tgt && idx_out && workspace && n_seg > 0 && max_m > 0 && max_n > 0, "roreg_knn_search_seg: bad arguments");
    ROREG_REQUIRE(k >= 1 && k <= 8, "roreg_knn_search_seg: k must be in 1..8 (got %d)", k);
    ROREG_REQUIRE(F == 3 || F == 32, "roreg_knn_search_seg: F must be 3 or 32 (got %d)", F);
    const size_t need = roreg_knn_search_seg_workspace(m_total, n_seg, max_m, max_n);
    ROREG_REQUIRE(workspace_bytes >= need, "roreg_knn_search_seg: workspace of %zu bytes needed", need);
    hipStream_t s = roreg::as_stream(stream);
    const int gx = (max_m + 255) / 256;
    int slices = (2048 + gx * n_seg - 1) / (gx * n_seg);
    if (slices > (max_n + 31) / 32) slices = (max_n + 31) / 32;
    if (slices < 1) slices = 1;
    float *pd = reinterpret_cast<float *>(workspace);
    int *pj = reinterpret_cast<int *>(pd + (size_t)slices * m_total * 8);
    const dim3 grid(gx, slices, n_seg);
    if (F == 3) hipLaunchKernelGGL((knn_slice_kernel<3, false>), grid, dim3(256), 0, s, src, 0, tgt, 0, k, 0, pd, pj, seg_src, seg_tgt, (int)m_total);
    else hipLaunchKernelGGL((knn_slice_kernel<32, false>), grid, dim3(256), 0, s, src, 0, tgt, 0, k, 0, pd, pj, seg_src, seg_tgt, (int)m_total);
    hipLaunchKernelGGL(knn_merge_kernel, dim3((unsigned)((m_total + 255) / 256)), dim3(256), 0, s, pd, pj, (int)m_total, slices, k, idx_out, (float *)nullptr);
    ROREG_CHECK_LAUNCH("roreg_knn_search_seg");
    return 0;
}

extern "C" int roreg_mutual_matches(const int64_t *nn01, const int64_t *nn10, int m, int n, const int64_t *sample0,
                                    const int64_t *sample1, int64_t *match_out, int32_t *count_out, void *stream) {
    ROREG_REQUIRE(count_out && m >= 0 && (m == 0 || (nn01 && nn10 && match_out)), "roreg_mutual_matches: bad arguments");
    hipLaunchKernelGGL(mutual_kernel, dim3(1), dim3(1024), 0, roreg::as_stream(stream), nn01, nn10, m, n, sample0, sample1,
                       match_out, count_out);
    ROREG_CHECK_LAUNCH("roreg_mutual_matches");
    return 0;
}
