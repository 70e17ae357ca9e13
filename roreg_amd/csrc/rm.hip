// Kernels of the rotation-coherence matcher (Match_ot, network/rot_coh_match.py:8-390).
//
// All per-point tensors are position-major ([points, channels] / [points, k, channels]) so that the k-NN gathers the
// network is built from are whole-row reads.  None of the N x N score matrices of the reference is materialised
// except the optimal-transport coupling (which is the output): top-k neighbours are selected on the fly from the dot
// products (the reference fully argsorts a 25M-element matrix 12 times per pair, rot_coh_match.py:34-45).
#include "common.h"

namespace {

constexpr int RM_F = 32;

// =====================================================================================================
// top-k of dot products: one thread per source row, sorted (value desc, index asc) list in registers;
// targets are wave-uniform -> scalar loads.  grid.y slices the target set; partial lists are merged.
// =====================================================================================================
template <int K>
__device__ __forceinline__ void topk_insert(float (&bv)[K], int (&bi)[K], float v, int j) {
    if (!(v > bv[K - 1])) return;          // strict: an equal value never displaces an earlier (lower) index
#pragma unroll
    for (int q = K - 1; q >= 0; --q) {
        const bool up = q > 0 && v > bv[q - 1];
        if (q > 0 && up) { bv[q] = bv[q - 1]; bi[q] = bi[q - 1]; }
        else { bv[q] = v; bi[q] = j; break; }
    }
}

template <int K>
__global__ __launch_bounds__(256) void topk_dot_kernel(const float *__restrict__ A, int m, const float *__restrict__ B, int n,
                                                       int slice, float *__restrict__ pv, int *__restrict__ pi) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int ii = i < m ? i : m - 1;
    float a[RM_F];
#pragma unroll
    for (int f = 0; f < RM_F; ++f) a[f] = A[(size_t)ii * RM_F + f];
    float bv[K];
    int bi[K];
#pragma unroll
    for (int q = 0; q < K; ++q) { bv[q] = -__builtin_inff(); bi[q] = 0x7fffffff; }
    const int j0 = blockIdx.y * slice, j1 = min(j0 + slice, n);
    for (int j = j0; j < j1; ++j) {
        const float *b = B + (size_t)j * RM_F;
        float acc = 0.f;
#pragma unroll
        for (int f = 0; f < RM_F; ++f) acc = fmaf(a[f], b[f], acc);
        topk_insert<K>(bv, bi, acc, j);
    }
    if (i < m) {
        float *ov = pv + ((size_t)blockIdx.y * m + i) * K;
        int *oi = pi + ((size_t)blockIdx.y * m + i) * K;
#pragma unroll
        for (int q = 0; q < K; ++q) { ov[q] = bv[q]; oi[q] = bi[q]; }
    }
}

template <int K>
__global__ __launch_bounds__(256) void topk_merge_kernel(const float *__restrict__ pv, const int *__restrict__ pi, int m, int slices,
                                                         int64_t *__restrict__ idx, float *__restrict__ val) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    float bv[K];
    int bi[K];
#pragma unroll
    for (int q = 0; q < K; ++q) { bv[q] = -__builtin_inff(); bi[q] = 0x7fffffff; }
    for (int s = 0; s < slices; ++s) {           // slice order == index order, so strict '>' keeps the lower index
        const float *v = pv + ((size_t)s * m + i) * K;
        const int *ix = pi + ((size_t)s * m + i) * K;
        for (int q = 0; q < K; ++q)
            if (ix[q] != 0x7fffffff) topk_insert<K>(bv, bi, v[q], ix[q]);
    }
#pragma unroll
    for (int q = 0; q < K; ++q) {
        idx[(size_t)i * K + q] = bi[q];
        if (val) val[(size_t)i * K + q] = bv[q];
    }
}

// =====================================================================================================
// pointwise linear layers (1x1 convs) on [L, CIN] rows.  One thread per position, weights through the scalar path.
//   plain : y  = W x + b
//   tail  : y += W relu((x - mean) * rstd) + b     (second conv of mlp_2layer / Contextnorm on top of the residual branch)
// =====================================================================================================
template <int CIN, int COUT, bool NORM, bool ACCUM>
__global__ __launch_bounds__(256) void linear_kernel(const float *__restrict__ x, int L, const float *__restrict__ W,
                                                     const float *__restrict__ b, const float *__restrict__ mean_rstd,
                                                     float *__restrict__ y, int ochunk) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int pp = p < L ? p : L - 1;
    float xi[CIN];
#pragma unroll
    for (int c = 0; c < CIN; ++c) {
        float v = x[(size_t)pp * CIN + c];
        if (NORM) v = fmaxf((v - mean_rstd[c]) * mean_rstd[CIN + c], 0.f);
        xi[c] = v;
    }
    if (p >= L) return;
    float *yo = y + (size_t)p * COUT;
    // blockIdx.y selects a chunk of output channels: with few positions (m = 5000 points) the output channels supply the
    // parallelism that positions alone cannot (78 waves would leave 90 % of the chip idle)
    const int o0 = blockIdx.y * ochunk, o1 = min(o0 + ochunk, COUT);
#pragma unroll 2
    for (int o = o0; o < o1; ++o) {
        float acc = b[o];
#pragma unroll
        for (int c = 0; c < CIN; ++c) acc = fmaf(xi[c], W[o * CIN + c], acc);
        yo[o] = ACCUM ? yo[o] + acc : acc;
    }
}

inline int linear_ochunk(int L, int Cout) {
    int chunk = Cout;
    while (chunk > 4 && (long long)L * (Cout / chunk) < 65536) chunk /= 2;
    return chunk;
}

// InstanceNorm statistics: per channel over all L positions (biased variance), accumulated in fp64, two stages,
// fixed reduction order (deterministic).  mean_rstd = [mean (C), 1/sqrt(var+eps) (C)].
__global__ __launch_bounds__(256) void in_stats_partial_kernel(const float *__restrict__ h, int L, int C, double *__restrict__ part) {
    extern __shared__ double sh[];                  // [256][2]
    const int lanes = 256 / C;                      // position lanes per block (C <= 128 and divides 256)
    const int c = threadIdx.x % C, pl = threadIdx.x / C;
    double s = 0, s2 = 0;
    if (pl < lanes)
        for (int p = blockIdx.x * lanes + pl; p < L; p += gridDim.x * lanes) {
            const double v = h[(size_t)p * C + c];
            s += v; s2 += v * v;
        }
    sh[threadIdx.x * 2] = s; sh[threadIdx.x * 2 + 1] = s2;
    __syncthreads();
    if (threadIdx.x < C) {
        double a = 0, a2 = 0;
        for (int q = 0; q < lanes; ++q) { a += sh[(q * C + threadIdx.x) * 2]; a2 += sh[(q * C + threadIdx.x) * 2 + 1]; }
        part[((size_t)blockIdx.x * C + threadIdx.x) * 2] = a;
        part[((size_t)blockIdx.x * C + threadIdx.x) * 2 + 1] = a2;
    }
}

__global__ __launch_bounds__(256) void in_stats_final_kernel(const double *__restrict__ part, int nblk, int L, int C, float eps,
                                                             float *__restrict__ mean_rstd) {
    __shared__ double sa[256], sb[256];
    const int c = blockIdx.x;                       // one block per channel; fixed-order tree => deterministic
    double a = 0, a2 = 0;
    if ((int)threadIdx.x < nblk) { a = part[((size_t)threadIdx.x * C + c) * 2]; a2 = part[((size_t)threadIdx.x * C + c) * 2 + 1]; }
    sa[threadIdx.x] = a; sb[threadIdx.x] = a2;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { sa[threadIdx.x] += sa[threadIdx.x + s]; sb[threadIdx.x] += sb[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double mu = sa[0] / L;
        double var = sb[0] / L - mu * mu;
        if (var < 0) var = 0;
        mean_rstd[c] = (float)mu;
        mean_rstd[C + c] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// =====================================================================================================
// k-NN attention core (4 heads x 8 dims, channel c = d*4 + head).  qp [m,32]; keys/values either dense [m,k,32] or a
// per-point table [n,32] gathered through idx [m,k].  out x [m,32] (before the merge conv).
// =====================================================================================================
template <int K>
__global__ __launch_bounds__(256) void knn_attention_kernel(const float *__restrict__ qp, const float *__restrict__ kp,
                                                            const float *__restrict__ vp, const int64_t *__restrict__ idx,
                                                            int k_table, int v_table, int m, float *__restrict__ x) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= m) return;
    float q[RM_F];
#pragma unroll
    for (int c = 0; c < RM_F; ++c) q[c] = qp[(size_t)p * RM_F + c];
    float sc[4][K];
    const float scale = 0.35355339059327373f;      // 1/sqrt(8)
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const size_t row = k_table ? (size_t)idx[(size_t)p * K + j] : (size_t)p * K + j;
        const float *kr = kp + row * RM_F;
        float a[4] = {0, 0, 0, 0};
#pragma unroll
        for (int d = 0; d < 8; ++d)
#pragma unroll
            for (int h = 0; h < 4; ++h) a[h] = fmaf(q[d * 4 + h], kr[d * 4 + h], a[h]);
#pragma unroll
        for (int h = 0; h < 4; ++h) sc[h][j] = a[h] * scale;
    }
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        float mx = sc[h][0];
#pragma unroll
        for (int j = 1; j < K; ++j) mx = fmaxf(mx, sc[h][j]);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < K; ++j) { sc[h][j] = __expf(sc[h][j] - mx); s += sc[h][j]; }
        const float inv = 1.0f / s;
#pragma unroll
        for (int j = 0; j < K; ++j) sc[h][j] *= inv;
    }
    float o[RM_F];
#pragma unroll
    for (int c = 0; c < RM_F; ++c) o[c] = 0.f;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const size_t row = v_table ? (size_t)idx[(size_t)p * K + j] : (size_t)p * K + j;
        const float *vr = vp + row * RM_F;
#pragma unroll
        for (int c = 0; c < RM_F; ++c) o[c] = fmaf(sc[c & 3][j], vr[c], o[c]);
    }
#pragma unroll
    for (int c = 0; c < RM_F; ++c) x[(size_t)p * RM_F + c] = o[c];
}

// =====================================================================================================
// small elementwise helpers
// =====================================================================================================
__global__ __launch_bounds__(256) void l2norm_rows_kernel(const float *__restrict__ x, int L, int C, float *__restrict__ y) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= L) return;
    float s = 0.f;
    for (int c = 0; c < C; ++c) { const float v = x[(size_t)p * C + c]; s = fmaf(v, v, s); }
    const float r = sqrtf(s);
    for (int c = 0; c < C; ++c) y[(size_t)p * C + c] = x[(size_t)p * C + c] / r;
}

__global__ __launch_bounds__(256) void colmax_partial_kernel(const float *__restrict__ x, int L, int C, float *__restrict__ part) {
    // block b reduces rows b, b+grid, ... ; thread c < C owns a column
    const int c = threadIdx.x;
    if (c >= C) return;
    float mx = -__builtin_inff();
    for (int p = blockIdx.x; p < L; p += gridDim.x) mx = fmaxf(mx, x[(size_t)p * C + c]);
    part[(size_t)blockIdx.x * C + c] = mx;
}

__global__ __launch_bounds__(256) void colmax_final_kernel(const float *__restrict__ part, int nblk, int C, float *__restrict__ out) {
    __shared__ float sm[256];
    const int c = blockIdx.x;
    sm[threadIdx.x] = (int)threadIdx.x < nblk ? part[(size_t)threadIdx.x * C + c] : -__builtin_inff();
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sm[threadIdx.x] = fmaxf(sm[threadIdx.x], sm[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[c] = sm[0];
}

// ctx[p] = [R[p,0..59], colmax[0..59]]
__global__ __launch_bounds__(256) void context_kernel(const float *__restrict__ R, const float *__restrict__ cmax, int m, float *__restrict__ ctx) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m * 120) return;
    const int p = i / 120, c = i - p * 120;
    ctx[i] = c < 60 ? R[(size_t)p * 60 + c] : cmax[c - 60];
}

// knn_coor[p,j,:] = (coor[idx[p,j]] - coor[p]) ; coor already divided by the normalisation step
__global__ __launch_bounds__(256) void knn_coor_kernel(const float *__restrict__ coor, const int64_t *__restrict__ idx, int m, int k,
                                                       float *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m * k) return;
    const int p = i / k;
    const size_t q = (size_t)idx[i];
#pragma unroll
    for (int d = 0; d < 3; ++d) out[(size_t)i * 3 + d] = coor[q * 3 + d] - coor[(size_t)p * 3 + d];
}

// out[p,j,:] = table[idx[p,j],:]   (C floats per row)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float *__restrict__ table, const int64_t *__restrict__ idx, size_t rows, int C,
                                                          float *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * C) return;
    const size_t r = i / C;
    const int c = (int)(i - r * C);
    out[i] = table[(size_t)idx[r] * C + c];
}

// value-MLP input: [pos (32) | knn_fea_n (32) | conf (32, broadcast over k)]  -> [m*k, 96]
__global__ __launch_bounds__(256) void value_input_kernel(const float *__restrict__ pos, const float *__restrict__ fea_n_table,
                                                          const int64_t *__restrict__ idx, const float *__restrict__ conf, int m, int k,
                                                          float *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)m * k * 96) return;
    const size_t r = i / 96;
    const int c = (int)(i - r * 96);
    const size_t p = r / k;
    float v;
    if (c < 32) v = pos[r * 32 + c];
    else if (c < 64) v = fea_n_table[(size_t)idx[r] * 32 + (c - 32)];
    else v = conf[p * 32 + (c - 64)];
    out[i] = v;
}

// out[p] = [a[p] (Ca) | b[p] (Cb) | c[p] (Cc)]
__global__ __launch_bounds__(256) void concat3_kernel(const float *__restrict__ a, int Ca, const float *__restrict__ b, int Cb,
                                                      const float *__restrict__ c, int Cc, int L, float *__restrict__ out) {
    const int C = Ca + Cb + Cc;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)L * C) return;
    const size_t p = i / C;
    const int ch = (int)(i - p * C);
    out[i] = ch < Ca ? a[p * Ca + ch] : (ch < Ca + Cb ? b[p * Cb + (ch - Ca)] : c[p * Cc + (ch - Ca - Cb)]);
}

// eqv [N,32,60] -> mean over g -> [N,32]   (torch.mean(dim=-1), rot_coh_match.py:346-347)
__global__ __launch_bounds__(256) void mean_g_kernel(const float *__restrict__ eqv, const int64_t *__restrict__ rows, int m, float *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m * RM_F) return;
    const int p = i / RM_F, f = i - p * RM_F;
    const size_t r = rows ? (size_t)rows[p] : (size_t)p;
    const float *src = eqv + (r * RM_F + f) * ROREG_G;
    float s = 0.f;
    for (int g = 0; g < ROREG_G; ++g) s += src[g];
    out[i] = s / 60.0f;
}

// =====================================================================================================
// Sinkhorn (log domain).  Each pass is a row-wise log-sum-exp of (Z0[i,:] + vec) over a row-major matrix; the column
// pass runs on the transposed copy so that both passes stream coalesced rows.
// =====================================================================================================
// Coupling matrix straight from the final descriptors: M[r][c] = <rowvec[r], colvec[c]> (score, rot_coh_match.py:363) with the
// dustbin row/column = alpha.  One thread per column (its descriptor in registers), rows arrive through the scalar path, so
// every store is coalesced; called twice (Z0 and its transpose) instead of transposing through memory.
__global__ __launch_bounds__(256) void ot_build_kernel(const float *__restrict__ rowvec, int R, const float *__restrict__ colvec, int C,
                                                       float alpha, int rows_per_block, float *__restrict__ M, int ld) {
    const int c = blockIdx.x * 256 + threadIdx.x;          // column in [0, C]  (C = dustbin)
    const int cc = c < C ? c : C - 1;
    float t[RM_F];
#pragma unroll
    for (int f = 0; f < RM_F; ++f) t[f] = colvec[(size_t)cc * RM_F + f];
    const int r0 = blockIdx.y * rows_per_block, r1 = min(r0 + rows_per_block, R + 1);
    if (c > C) return;
    for (int r = r0; r < r1; ++r) {
        float v = alpha;
        if (r < R && c < C) {
            const float *a = rowvec + (size_t)r * RM_F;
            float acc = 0.f;
#pragma unroll
            for (int f = 0; f < RM_F; ++f) acc = fmaf(a[f], t[f], acc);
            v = acc;
        }
        M[(size_t)r * ld + c] = v;
    }
}

// out[i] = log_a(i) - LSE_j(Z[i,j] + vec[j]);  log_a = normc for i < R-1, last_extra + normc for the dustbin row
__global__ __launch_bounds__(256) void row_lse_kernel(const float *__restrict__ Z, int R, int C, int ld, const float *__restrict__ vec,
                                                      float normc, float last_extra, float *__restrict__ out) {
    __shared__ float smx[4], ssum[4];
    const int i = blockIdx.x;
    const float *row = Z + (size_t)i * ld;          // ld is a multiple of 4: rows are 16-byte aligned
    float mx = -__builtin_inff(), s = 0.f;
    auto upd = [&](float x) {
        if (x > mx) { s = s * __expf(mx - x) + 1.0f; mx = x; }
        else s += __expf(x - mx);
    };
    const int C4 = C & ~3;
    for (int j = threadIdx.x * 4; j < C4; j += 1024) {
        const float4 z = *reinterpret_cast<const float4 *>(row + j);
        const float4 v = *reinterpret_cast<const float4 *>(vec + j);
        upd(z.x + v.x); upd(z.y + v.y); upd(z.z + v.z); upd(z.w + v.w);
    }
    if ((int)threadIdx.x < C - C4) upd(row[C4 + threadIdx.x] + vec[C4 + threadIdx.x]);
    // combine (mx, s) pairs: wave butterfly then across the 4 waves
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float omx = __shfl_xor(mx, o), os = __shfl_xor(s, o);
        const float nm = fmaxf(mx, omx);
        s = (mx == -__builtin_inff() ? 0.f : s * __expf(mx - nm)) + (omx == -__builtin_inff() ? 0.f : os * __expf(omx - nm));
        mx = nm;
    }
    if ((threadIdx.x & 63) == 0) { smx[threadIdx.x >> 6] = mx; ssum[threadIdx.x >> 6] = s; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float M = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
        float S = 0.f;
        for (int w = 0; w < 4; ++w) S += ssum[w] * __expf(smx[w] - M);
        const float lse = M + __logf(S);
        out[i] = (i == R - 1 ? last_extra + normc : normc) - lse;
    }
}

__global__ __launch_bounds__(256) void ot_final_kernel(const float *__restrict__ Z0, int ld, int m, int n, const float *__restrict__ u,
                                                       const float *__restrict__ v, float normc, float *__restrict__ Z) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t tot = (size_t)(m + 1) * (n + 1);
    if (i >= tot) return;
    const int r = (int)(i / (n + 1)), c = (int)(i - (size_t)r * (n + 1));
    Z[i] = Z0[(size_t)r * ld + c] + u[r] + v[c] - normc;
}

// arg-max over the first C-1 columns of rows 0..R-2 of (Z0[i,j] + vec[j]) (+ rowc[i] - normc for the value)
template <bool ROWC_FIRST>
__global__ __launch_bounds__(256) void row_argmax_kernel(const float *__restrict__ Z, int R, int C, int ld, const float *__restrict__ vec,
                                                         const float *__restrict__ rowc, float normc, int64_t *__restrict__ idx,
                                                         float *__restrict__ val) {
    __shared__ float sv[256];
    __shared__ int si[256];
    const int i = blockIdx.x;            // < R-1
    const float *row = Z + (size_t)i * ld;
    float bv = -__builtin_inff();
    int bi = 0x7fffffff;
    for (int j = threadIdx.x; j < C - 1; j += 256) {
        // Z = ((Z0 + u_row) + v_col) - norm in the reference; the transposed pass has rowc = v, vec = u
        const float x = ROWC_FIRST ? ((row[j] + rowc[i]) + vec[j]) - normc : ((row[j] + vec[j]) + rowc[i]) - normc;
        if (x > bv) { bv = x; bi = j; }
    }
    sv[threadIdx.x] = bv; si[threadIdx.x] = bi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            const float ov = sv[threadIdx.x + s];
            const int oi = si[threadIdx.x + s];
            if (ov > sv[threadIdx.x] || (ov == sv[threadIdx.x] && oi < si[threadIdx.x])) { sv[threadIdx.x] = ov; si[threadIdx.x] = oi; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { idx[i] = si[0]; val[i] = sv[0]; }
}

__global__ __launch_bounds__(256) void ot_readout_kernel(const int64_t *__restrict__ i0, const float *__restrict__ v0, int m,
                                                         const int64_t *__restrict__ i1, int n, int64_t *__restrict__ m0,
                                                         int64_t *__restrict__ m1, float *__restrict__ s0, float *__restrict__ s1) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < m) {
        const bool mu = i1[i0[t]] == (int64_t)t;
        m0[t] = mu ? i0[t] : -1;
        s0[t] = mu ? expf(v0[t]) : 0.f;
    }
    if (t < n) {
        const int64_t a = i1[t];
        const bool mu1 = i0[a] == (int64_t)t;           // mutual1; then valid0[a] is the same predicate seen from a
        m1[t] = mu1 ? a : -1;
        s1[t] = mu1 ? expf(v0[a]) : 0.f;
    }
}

}  // namespace

// -----------------------------------------------------------------------------------------------------------
extern "C" size_t roreg_topk_dot_workspace_size(int m, int n, int k) {
    const int gx = (m + 255) / 256;
    int slices = (320 + gx - 1) / gx;
    if (slices > (n + 63) / 64) slices = (n + 63) / 64;
    if (slices < 1) slices = 1;
    return (size_t)slices * m * k * 2;       // floats (values) + ints (indices), 4 bytes each
}

extern "C" int roreg_topk_dot(const float *A, int m, const float *B, int n, int k, int64_t *idx_out, float *val_out, float *ws,
                              size_t ws_floats, void *stream) {
    ROREG_REQUIRE(A && B && idx_out && ws && m > 0 && n > 0, "roreg_topk_dot: bad arguments");
    ROREG_REQUIRE(k == 16 || k == 8 || k == 1, "roreg_topk_dot: k must be 16, 8 or 1 (got %d)", k);
    ROREG_REQUIRE(k <= n, "roreg_topk_dot: k > n");
    const int gx = (m + 255) / 256;
    int slices = (320 + gx - 1) / gx;
    if (slices > (n + 63) / 64) slices = (n + 63) / 64;
    if (slices < 1) slices = 1;
    const int slice = (n + slices - 1) / slices;
    slices = (n + slice - 1) / slice;
    ROREG_REQUIRE(ws_floats >= (size_t)slices * m * k * 2, "roreg_topk_dot: workspace too small");
    float *pv = ws;
    int *pi = reinterpret_cast<int *>(ws + (size_t)slices * m * k);
    hipStream_t s = roreg::as_stream(stream);
#define RM_TOPK(KK)                                                                                                      \
    hipLaunchKernelGGL(topk_dot_kernel<KK>, dim3(gx, slices), dim3(256), 0, s, A, m, B, n, slice, pv, pi);               \
    hipLaunchKernelGGL(topk_merge_kernel<KK>, dim3(gx), dim3(256), 0, s, pv, pi, m, slices, idx_out, val_out);
    if (k == 16) { RM_TOPK(16) } else if (k == 8) { RM_TOPK(8) } else { RM_TOPK(1) }
#undef RM_TOPK
    ROREG_CHECK_LAUNCH("roreg_topk_dot");
    return 0;
}

extern "C" int roreg_linear(const float *x, int L, int Cin, const float *W, const float *b, int Cout, float *y, void *stream) {
    ROREG_REQUIRE(x && W && b && y && L > 0, "roreg_linear: bad arguments");
    hipStream_t s = roreg::as_stream(stream);
    const int oc = linear_ochunk(L, Cout);
    const dim3 g((L + 255) / 256, (Cout + oc - 1) / oc), t(256);
#define RM_LIN(CI, CO)                                                                                      \
    if (Cin == CI && Cout == CO) {                                                                           \
        hipLaunchKernelGGL((linear_kernel<CI, CO, false, false>), g, t, 0, s, x, L, W, b, nullptr, y, oc);  \
        ROREG_CHECK_LAUNCH("roreg_linear");                                                              \
        return 0;                                                                                        \
    }
    RM_LIN(32, 32) RM_LIN(96, 64) RM_LIN(3, 64) RM_LIN(120, 128) RM_LIN(64, 64)
    RM_LIN(96, 32) RM_LIN(3, 32) RM_LIN(120, 32) RM_LIN(64, 32)
#undef RM_LIN
    roreg::set_error("roreg_linear: unsupported shape %d -> %d", Cin, Cout);
    return 2;
}

extern "C" int roreg_instnorm_stats(const float *h, int L, int C, float eps, float *mean_rstd, double *ws /* 2*C*256 doubles */, void *stream) {
    ROREG_REQUIRE(h && mean_rstd && ws && L > 0, "roreg_instnorm_stats: bad arguments");
    ROREG_REQUIRE(C > 0 && C <= 128 && 256 % C == 0, "roreg_instnorm_stats: C must divide 256 (got %d)", C);
    hipStream_t s = roreg::as_stream(stream);
    const int lanes = 256 / C;
    int nblk = (L + lanes - 1) / lanes;
    if (nblk > 256) nblk = 256;
    hipLaunchKernelGGL(in_stats_partial_kernel, dim3(nblk), dim3(256), 256 * 2 * sizeof(double), s, h, L, C, ws);
    hipLaunchKernelGGL(in_stats_final_kernel, dim3(C), dim3(256), 0, s, ws, nblk, L, C, eps, mean_rstd);
    ROREG_CHECK_LAUNCH("roreg_instnorm_stats");
    return 0;
}

extern "C" int roreg_mlp_tail(const float *h, int L, int Cmid, const float *mean_rstd, const float *W2, const float *b2, float *y,
                              void *stream) {
    ROREG_REQUIRE(h && mean_rstd && W2 && b2 && y && L > 0, "roreg_mlp_tail: bad arguments");
    hipStream_t s = roreg::as_stream(stream);
    const int oc = linear_ochunk(L, 32);
    const dim3 g((L + 255) / 256, (32 + oc - 1) / oc), t(256);
    if (Cmid == 64) hipLaunchKernelGGL((linear_kernel<64, 32, true, true>), g, t, 0, s, h, L, W2, b2, mean_rstd, y, oc);
    else if (Cmid == 128) hipLaunchKernelGGL((linear_kernel<128, 32, true, true>), g, t, 0, s, h, L, W2, b2, mean_rstd, y, oc);
    else { roreg::set_error("roreg_mlp_tail: unsupported width %d", Cmid); return 2; }
    ROREG_CHECK_LAUNCH("roreg_mlp_tail");
    return 0;
}

extern "C" int roreg_knn_attention(const float *qp, const float *kp, const float *vp, const int64_t *idx, int k_is_table, int v_is_table,
                                   int m, int k, float *x_out, void *stream) {
    ROREG_REQUIRE(qp && kp && vp && x_out && m > 0, "roreg_knn_attention: bad arguments");
    ROREG_REQUIRE((!k_is_table && !v_is_table) || idx, "roreg_knn_attention: idx needed for table operands");
    hipStream_t s = roreg::as_stream(stream);
    if (k == 16) hipLaunchKernelGGL(knn_attention_kernel<16>, dim3((m + 255) / 256), dim3(256), 0, s, qp, kp, vp, idx, k_is_table, v_is_table, m, x_out);
    else if (k == 8) hipLaunchKernelGGL(knn_attention_kernel<8>, dim3((m + 255) / 256), dim3(256), 0, s, qp, kp, vp, idx, k_is_table, v_is_table, m, x_out);
    else { roreg::set_error("roreg_knn_attention: k must be 16 or 8"); return 2; }
    ROREG_CHECK_LAUNCH("roreg_knn_attention");
    return 0;
}

extern "C" int roreg_rm_elementwise(int op, const float *a, const float *b, const float *c, const int64_t *idx, int L, int k, int C,
                                    float *out, float *ws, void *stream) {
    // op 0: l2-normalise rows of a [L,C]           op 1: column max of a [L,C] -> out [C] (ws: 256*C floats)
    // op 2: ctx = [a (R [L,60]) | b (colmax [60])]  op 3: knn_coor from a (coor [*,3]) and idx [L,k]
    // op 4: gather rows: out[r,:] = a[idx[r],:] for r < L*k (C floats)   op 5: value input [pos a | table b via idx | conf c]
    // op 6: concat3 of a [L,C], b [L,C], c [L,C] (C each, or k = C of c when it differs)   op 7: mean over g of a [*,32,60] rows idx -> [L,32]
    hipStream_t s = roreg::as_stream(stream);
    ROREG_REQUIRE(a && out && L > 0, "roreg_rm_elementwise: bad arguments");
    switch (op) {
    case 0: hipLaunchKernelGGL(l2norm_rows_kernel, dim3((L + 255) / 256), dim3(256), 0, s, a, L, C, out); break;
    case 1: {
        ROREG_REQUIRE(ws && C <= 256, "roreg_rm_elementwise: colmax needs a workspace");
        int nblk = L < 256 ? L : 256;
        hipLaunchKernelGGL(colmax_partial_kernel, dim3(nblk), dim3(256), 0, s, a, L, C, ws);
        hipLaunchKernelGGL(colmax_final_kernel, dim3(C), dim3(256), 0, s, ws, nblk, C, out);
        break;
    }
    case 2: hipLaunchKernelGGL(context_kernel, dim3((L * 120 + 255) / 256), dim3(256), 0, s, a, b, L, out); break;
    case 3: hipLaunchKernelGGL(knn_coor_kernel, dim3((L * k + 255) / 256), dim3(256), 0, s, a, idx, L, k, out); break;
    case 4: hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)(((size_t)L * k * C + 255) / 256)), dim3(256), 0, s, a, idx, (size_t)L * k, C, out); break;
    case 5: hipLaunchKernelGGL(value_input_kernel, dim3((unsigned)(((size_t)L * k * 96 + 255) / 256)), dim3(256), 0, s, a, b, idx, c, L, k, out); break;
    case 6: hipLaunchKernelGGL(concat3_kernel, dim3((unsigned)(((size_t)L * (2 * C + k) + 255) / 256)), dim3(256), 0, s, a, C, b, C, c, k, L, out); break;
    case 7: hipLaunchKernelGGL(mean_g_kernel, dim3((L * RM_F + 255) / 256), dim3(256), 0, s, a, idx, L, out); break;
    default: roreg::set_error("roreg_rm_elementwise: unknown op %d", op); return 2;
    }
    ROREG_CHECK_LAUNCH("roreg_rm_elementwise");
    return 0;
}

extern "C" int roreg_sinkhorn(const float *src_final, int m, const float *tgt_final, int n, float alpha, int iters, float *Z_out,
                              int64_t *matches0, int64_t *matches1, float *mscores0, float *mscores1, float *ws, size_t ws_floats,
                              void *stream) {
    ROREG_REQUIRE(src_final && tgt_final && Z_out && ws && m > 0 && n > 0 && iters >= 0, "roreg_sinkhorn: bad arguments");
    const size_t tot = (size_t)(m + 1) * (n + 1);
    const int ldz = (n + 1 + 3) & ~3, ldt = (m + 1 + 3) & ~3;             // padded row pitches: 16-byte aligned rows for float4 streaming
    const size_t sz0 = (size_t)(m + 1) * ldz, szt = (size_t)(n + 1) * ldt;
    const int up = (m + 1 + 3) & ~3, vp = (n + 1 + 3) & ~3;
    const size_t need = sz0 + szt + up + vp + (size_t)(m + n) + 2 * (size_t)(m + n) + 8;      // Z0, Z0^T, u, v, row/col max values + indices
    ROREG_REQUIRE(ws_floats >= need, "roreg_sinkhorn: workspace of %zu floats needed (got %zu)", need, ws_floats);
    hipStream_t s = roreg::as_stream(stream);
    float *Z0 = ws, *Z0T = ws + sz0, *u = Z0T + szt, *v = u + up;
    float *val0 = v + vp, *val1 = val0 + m;
    const size_t off = ((reinterpret_cast<uintptr_t>(val1 + n) + 7) & ~(uintptr_t)7) - reinterpret_cast<uintptr_t>(ws);
    int64_t *i0 = reinterpret_cast<int64_t *>(reinterpret_cast<char *>(ws) + off);
    int64_t *i1 = i0 + m;
    ROREG_REQUIRE(reinterpret_cast<char *>(i1 + n) <= reinterpret_cast<char *>(ws + ws_floats), "roreg_sinkhorn: workspace too small");
    const float normc = -logf((float)(m + n));
    const unsigned gb = (unsigned)((tot + 255) / 256);
    {
        const int rpb = 32;
        hipLaunchKernelGGL(ot_build_kernel, dim3((n + 1 + 255) / 256, (m + 1 + rpb - 1) / rpb), dim3(256), 0, s, src_final, m, tgt_final, n,
                           alpha, rpb, Z0, ldz);
        hipLaunchKernelGGL(ot_build_kernel, dim3((m + 1 + 255) / 256, (n + 1 + rpb - 1) / rpb), dim3(256), 0, s, tgt_final, n, src_final, m,
                           alpha, rpb, Z0T, ldt);
    }
    (void)hipMemsetAsync(u, 0, sizeof(float) * up, s);
    (void)hipMemsetAsync(v, 0, sizeof(float) * vp, s);
    const float ln_n = logf((float)n), ln_m = logf((float)m);
    for (int it = 0; it < iters; ++it) {
        hipLaunchKernelGGL(row_lse_kernel, dim3(m + 1), dim3(256), 0, s, Z0, m + 1, n + 1, ldz, v, normc, ln_n, u);
        hipLaunchKernelGGL(row_lse_kernel, dim3(n + 1), dim3(256), 0, s, Z0T, n + 1, m + 1, ldt, u, normc, ln_m, v);
    }
    hipLaunchKernelGGL(ot_final_kernel, dim3(gb), dim3(256), 0, s, Z0, ldz, m, n, u, v, normc, Z_out);
    if (matches0) {
        ROREG_REQUIRE(matches1 && mscores0 && mscores1, "roreg_sinkhorn: all four readout outputs are needed");
        hipLaunchKernelGGL(row_argmax_kernel<true>, dim3(m), dim3(256), 0, s, Z0, m + 1, n + 1, ldz, v, u, normc, i0, val0);
        hipLaunchKernelGGL(row_argmax_kernel<false>, dim3(n), dim3(256), 0, s, Z0T, n + 1, m + 1, ldt, u, v, normc, i1, val1);
        const int mx = m > n ? m : n;
        hipLaunchKernelGGL(ot_readout_kernel, dim3((mx + 255) / 256), dim3(256), 0, s, i0, val0, m, i1, n, matches0, matches1, mscores0, mscores1);
    }
    ROREG_CHECK_LAUNCH("roreg_sinkhorn");
    return 0;
}

extern "C" size_t roreg_sinkhorn_workspace_size(int m, int n) {
    return (size_t)(m + 1) * ((n + 4) & ~3) + (size_t)(n + 1) * ((m + 4) & ~3) + 5 * (size_t)(m + n + 8) + 32;
}
