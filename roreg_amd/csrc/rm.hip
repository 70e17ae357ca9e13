// Kernels of the rotation-coherence matcher (Match_ot, network/rot_coh_match.py:8-390).
//
// All per-point tensors are position-major ([points, channels] / [points, k, channels]) so that the k-NN gathers the
// network is built from are whole-row reads.  None of the N x N score matrices of the reference is materialised
// except the optimal-transport coupling (which is the output): top-k neighbours are selected on the fly from the dot
// products (the reference fully argsorts a 25M-element matrix 12 times per pair, rot_coh_match.py:34-45).
#include "common.h"
#include <stdlib.h>

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

// Segments (several pairs per launch): rows of A and B are concatenations over the pairs, segA/segB [P+1] are the row offsets and
// blockIdx.z is the pair; a source row only sees the targets of its own pair and the indices written are GLOBAL rows of B, so the
// gathers downstream need no per-pair base.  segA == nullptr: one pair (m x n).
template <int K>
__global__ __launch_bounds__(256) void topk_dot_kernel(const float *__restrict__ A, int m, const float *__restrict__ B, int n,
                                                       const int *__restrict__ segA, const int *__restrict__ segB, int slices,
                                                       float *__restrict__ pv, int *__restrict__ pi) {
    const int a0 = segA ? segA[blockIdx.z] : 0, mp = segA ? segA[blockIdx.z + 1] - a0 : m;
    const int b0 = segA ? segB[blockIdx.z] : 0, np_ = segA ? segB[blockIdx.z + 1] - b0 : n;
    if ((int)(blockIdx.x * 256) >= mp || np_ <= 0) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int ii = a0 + (i < mp ? i : mp - 1);
    float a[RM_F];
#pragma unroll
    for (int f = 0; f < RM_F; ++f) a[f] = A[(size_t)ii * RM_F + f];
    float bv[K];
    int bi[K];
#pragma unroll
    for (int q = 0; q < K; ++q) { bv[q] = -__builtin_inff(); bi[q] = 0x7fffffff; }
    const int slice = (np_ + slices - 1) / slices;
    const int j0 = b0 + blockIdx.y * slice, j1 = min(j0 + slice, b0 + np_);
    for (int j = j0; j < j1; ++j) {
        const float *b = B + (size_t)j * RM_F;
        float acc = 0.f;
#pragma unroll
        for (int f = 0; f < RM_F; ++f) acc = fmaf(a[f], b[f], acc);
        topk_insert<K>(bv, bi, acc, j);
    }
    if (i < mp) {
        float *ov = pv + ((size_t)blockIdx.y * m + a0 + i) * K;
        int *oi = pi + ((size_t)blockIdx.y * m + a0 + i) * K;
#pragma unroll
        for (int q = 0; q < K; ++q) { ov[q] = bv[q]; oi[q] = bi[q]; }
    }
}

// The same search with the dot products on the matrix cores: acc = fmaf(a[f], b[f], acc) over the 32 channels from 0 is exactly what a chain of
// sixteen v_mfma_f32_32x32x2_f32 computes (a float32 fma chain over k, csrc/linear_chain.hip), so the scores -- and with them every list -- are
// BITWISE those of topk_dot_kernel at half its vector work.  A wavefront owns 32 source rows as the tile's COLUMNS (lane l: source l % 32) and
// walks its slice of the targets in 32-row tiles: a lane then holds 16 scores of ITS source per tile (target rows 8 (r / 4) + 4 (l / 32) + r % 4,
// ascending in r), which go through the same sorted insertion, in index order; the two lanes of a source (l, l + 32: disjoint target subsets)
// merge their lists at the end under (value descending, index ascending).
typedef float f32x16_tk __attribute__((ext_vector_type(16)));
template <int K>
__device__ __forceinline__ void topk_insert_idx(float (&bv)[K], int (&bi)[K], float v, int j) {      // full order: an equal value with a LOWER index goes first
    if (!(v > bv[K - 1] || (v == bv[K - 1] && j < bi[K - 1]))) return;
#pragma unroll
    for (int q = K - 1; q >= 0; --q) {
        const bool up = q > 0 && (v > bv[q - 1] || (v == bv[q - 1] && j < bi[q - 1]));
        if (q > 0 && up) { bv[q] = bv[q - 1]; bi[q] = bi[q - 1]; }
        else { bv[q] = v; bi[q] = j; break; }
    }
}
// List maintenance, packed (round 6).  A tile hands every lane 16 scores; the sorted insertion is a chain of K compare-and-shift steps that the
// whole wavefront executes whenever ANY of its 64 lanes has to insert -- with the lists warm that is one or two lanes at a time, for a third
// of all candidates: ~1550 chain executions per 2500 candidates (K = 16), the kernel's vector pipe several times busier than its matrix pipe.
// PACKED: a lane first marks which of its 16 scores beat its list's last entry (one compare each), then the wavefront runs the chain
// max-over-lanes(marked) times, every lane taking ITS next marked score (lowest target index first: the order of the plain form) -- ~375
// executions for the same candidates.  A score that no longer beats the list when its turn comes is dropped by the insertion's own test, so
// the lists are those of the plain form, entry for entry.  MEASURED (tools/probe/topk_ab.py, profiles/r06_topk_packed.txt): on random unit
// descriptors the k = 16 search of 52 stacked 5000 x 5000 pairs goes 2.62 -> 1.93 ms (k = 8: 1.60 -> 1.52); inside the --RD --RM pipeline,
// on the matcher's own features, 1443 -> 1495 us per launch at k = 16 and 1198 -> 1159 at k = 8: nothing.  Neither pipe is the bound there
// (matrix pipe 27 - 36 % busy, vector pipe ~46 %): the kernel waits -- each wavefront's tile loads sit in front of its sixteen dependent
// MFMAs, three wavefronts per SIMD.  Opt-in (ROREG_TOPK_PACKED=1).
template <bool PACKED, int K>
__global__ __launch_bounds__(256) void topk_dot_mfma_kernel(const float *__restrict__ A, int m, const float *__restrict__ B, int n,
                                                            const int *__restrict__ segA, const int *__restrict__ segB, int slices,
                                                            float *__restrict__ pv, int *__restrict__ pi) {
    const int a0 = segA ? segA[blockIdx.z] : 0, mp = segA ? segA[blockIdx.z + 1] - a0 : m;
    const int b0 = segA ? segB[blockIdx.z] : 0, np_ = segA ? segB[blockIdx.z + 1] - b0 : n;
    if ((int)(blockIdx.x * 128) >= mp || np_ <= 0) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
    const int i = blockIdx.x * 128 + w * 32 + j;                           // this lane's source row (of the pair)
    float sq[16];                                                          // B operand: source row i, channels 2 k + h
    {
        const float4 *sr = reinterpret_cast<const float4 *>(A + (size_t)(a0 + min(i, mp - 1)) * RM_F);
#pragma unroll
        for (int q = 0; q < 8; ++q) { const float4 x4 = sr[q]; sq[2 * q] = h ? x4.y : x4.x; sq[2 * q + 1] = h ? x4.w : x4.z; }
    }
    float bv[K];
    int bi[K];
#pragma unroll
    for (int q = 0; q < K; ++q) { bv[q] = -__builtin_inff(); bi[q] = 0x7fffffff; }
    const int slice = (np_ + slices - 1) / slices;
    const int j0 = blockIdx.y * slice, j1 = min(j0 + slice, np_);          // (local to the pair)
    for (int t0 = j0; t0 < j1; t0 += 32) {
        float tq[16];                                                      // A operand: target row t0 + j, channels 2 k + h
        {
            const float4 *tr = reinterpret_cast<const float4 *>(B + (size_t)(b0 + min(t0 + j, np_ - 1)) * RM_F);
#pragma unroll
            for (int q = 0; q < 8; ++q) { const float4 x4 = tr[q]; tq[2 * q] = h ? x4.y : x4.x; tq[2 * q + 1] = h ? x4.w : x4.z; }
        }
        f32x16_tk acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(tq[kk], sq[kk], acc, 0, 0, 0);
        if (!PACKED) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int t = t0 + 8 * (r >> 2) + 4 * h + (r & 3);
                if (t < j1) topk_insert<K>(bv, bi, acc[r], b0 + t);
            }
        } else {
            const float last = bv[K - 1];
            unsigned marked = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int t = t0 + 8 * (r >> 2) + 4 * h + (r & 3);
                marked |= (t < j1 && acc[r] > last) ? (1u << r) : 0u;
            }
            while (__builtin_amdgcn_ballot_w64(marked != 0)) {
                if (marked) {
                    const int r = __builtin_ctz(marked);
                    marked &= marked - 1;
                    // acc[r] for a per-lane r: a select tree over the bits of r (no indexed register file access)
                    const bool r0 = r & 1, r1 = r & 2, r2 = r & 4, r3 = r & 8;
                    const float a0_ = r0 ? acc[1] : acc[0], a1_ = r0 ? acc[3] : acc[2], a2_ = r0 ? acc[5] : acc[4], a3_ = r0 ? acc[7] : acc[6];
                    const float a4_ = r0 ? acc[9] : acc[8], a5_ = r0 ? acc[11] : acc[10], a6_ = r0 ? acc[13] : acc[12], a7_ = r0 ? acc[15] : acc[14];
                    const float c0_ = r1 ? a1_ : a0_, c1_ = r1 ? a3_ : a2_, c2_ = r1 ? a5_ : a4_, c3_ = r1 ? a7_ : a6_;
                    const float e0_ = r2 ? c1_ : c0_, e1_ = r2 ? c3_ : c2_;
                    const float v = r3 ? e1_ : e0_;
                    topk_insert<K>(bv, bi, v, b0 + t0 + 8 * (r >> 2) + 4 * h + (r & 3));
                }
            }
        }
    }
    // the other half-wave's list of the same source
#pragma unroll
    for (int q = 0; q < K; ++q) {
        const float ov = __shfl_xor(bv[q], 32);
        const int oi = __shfl_xor(bi[q], 32);
        if (h == 0 && oi != 0x7fffffff) topk_insert_idx<K>(bv, bi, ov, oi);
    }
    if (h == 0 && i < mp) {
        float *ov = pv + ((size_t)blockIdx.y * m + a0 + i) * K;
        int *oi = pi + ((size_t)blockIdx.y * m + a0 + i) * K;
#pragma unroll
        for (int q = 0; q < K; ++q) { ov[q] = bv[q]; oi[q] = bi[q]; }
    }
}

// The same search with the target tile staged ONCE per workgroup (round 6).  The four wavefronts of a workgroup walk the same target tiles;
// in the kernel above each of them fetches the whole 4 KB tile itself (a lane loads its target row's 128 bytes and uses half) and then waits
// for it in front of its sixteen dependent MFMAs -- the counters have the matrix pipe 33-40 % and the vector pipe ~42 % busy, the rest is
// waiting.  Here a lane loads 16 bytes of the NEXT tile while the current one is multiplied and inserted, the tile goes through LDS in the
// operand order (element (row j, channel c) at ((c & 1) * 16 + (c >> 1)) * 32 + j: lane (j, h) reads channel 2 k + h at (h * 16 + k) * 32 + j,
// consecutive lanes consecutive words), two buffers, one barrier per tile.  Scores, insertion order and lists are those of the kernel above.
// In the --RD --RM pipeline: 1433 -> 1298 us per launch at k = 16, 1194 -> 1015 at k = 8 (profiles/r06_topk_packed.txt).  Also measured on top
// of it: tile t + 1's MFMAs interleaved in program order with tile t's insertions (one candidate behind each MFMA) -- 1486 / 1305 us, slower
// (156 registers, three wavefronts per SIMD, and the insertion's branches sit between the MFMAs): not kept.
template <int K>
__global__ __launch_bounds__(256) void topk_dot_mfma_lds_kernel(const float *__restrict__ A, int m, const float *__restrict__ B, int n,
                                                                const int *__restrict__ segA, const int *__restrict__ segB, int slices,
                                                                float *__restrict__ pv, int *__restrict__ pi) {
    __shared__ float tile[2][RM_F * 32];
    const int a0 = segA ? segA[blockIdx.z] : 0, mp = segA ? segA[blockIdx.z + 1] - a0 : m;
    const int b0 = segA ? segB[blockIdx.z] : 0, np_ = segA ? segB[blockIdx.z + 1] - b0 : n;
    if ((int)(blockIdx.x * 128) >= mp || np_ <= 0) return;                 // (uniform over the workgroup)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
    const int i = blockIdx.x * 128 + w * 32 + j;
    float sq[16];
    {
        const float4 *sr = reinterpret_cast<const float4 *>(A + (size_t)(a0 + min(i, mp - 1)) * RM_F);
#pragma unroll
        for (int q = 0; q < 8; ++q) { const float4 x4 = sr[q]; sq[2 * q] = h ? x4.y : x4.x; sq[2 * q + 1] = h ? x4.w : x4.z; }
    }
    float bv[K];
    int bi[K];
#pragma unroll
    for (int q = 0; q < K; ++q) { bv[q] = -__builtin_inff(); bi[q] = 0x7fffffff; }
    const int slice = (np_ + slices - 1) / slices;
    const int j0 = blockIdx.y * slice, j1 = min(j0 + slice, np_);
    // this thread's share of a tile: row lr = tid / 8 (0 .. 31), channels 4 lc .. 4 lc + 3 (lc = tid % 8)
    const int lr = threadIdx.x >> 3, lc = threadIdx.x & 7;
    auto fetch = [&](int t0) -> float4 {
        return *reinterpret_cast<const float4 *>(B + (size_t)(b0 + min(t0 + lr, np_ - 1)) * RM_F + 4 * lc);
    };
    auto stage = [&](float *buf, const float4 &x) {
        // channels c = 4 lc + e: (c & 1) * 16 + (c >> 1) = (e & 1) * 16 + 2 lc + (e >> 1)
        buf[(2 * lc) * 32 + lr] = x.x;
        buf[(16 + 2 * lc) * 32 + lr] = x.y;
        buf[(2 * lc + 1) * 32 + lr] = x.z;
        buf[(16 + 2 * lc + 1) * 32 + lr] = x.w;
    };
    if (j0 < j1) stage(tile[0], fetch(j0));
    __syncthreads();
    int cur = 0;
    for (int t0 = j0; t0 < j1; t0 += 32, cur ^= 1) {
        const bool more = t0 + 32 < j1;
        float4 nx = {0.f, 0.f, 0.f, 0.f};
        if (more) nx = fetch(t0 + 32);                                     // in flight under this tile's MFMAs and list work
        float tq[16];
        const float *src = tile[cur] + h * 16 * 32 + j;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) tq[kk] = src[kk * 32];
        f32x16_tk acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(tq[kk], sq[kk], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int t = t0 + 8 * (r >> 2) + 4 * h + (r & 3);
            if (t < j1) topk_insert<K>(bv, bi, acc[r], b0 + t);
        }
        if (more) stage(tile[cur ^ 1], nx);
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < K; ++q) {
        const float ov = __shfl_xor(bv[q], 32);
        const int oi = __shfl_xor(bi[q], 32);
        if (h == 0 && oi != 0x7fffffff) topk_insert_idx<K>(bv, bi, ov, oi);
    }
    if (h == 0 && i < mp) {
        float *ov = pv + ((size_t)blockIdx.y * m + a0 + i) * K;
        int *oi = pi + ((size_t)blockIdx.y * m + a0 + i) * K;
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
        // fewer than K comparable targets (NaN dot products compare false): the empty slots point at row 0 -- a valid address for the
        // gathers downstream; the values are meaningless either way
        idx[(size_t)i * K + q] = bi[q] == 0x7fffffff ? 0 : bi[q];
        if (val) val[(size_t)i * K + q] = bv[q];
    }
}

// segment of row r in offsets off[0..n_seg] (off[n_seg] = total): the last s with off[s] <= r
__device__ __forceinline__ int seg_of(const int *__restrict__ off, int n_seg, int r) {
    int lo = 0, hi = n_seg - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (off[mid] <= r) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// =====================================================================================================
// pointwise linear layers (1x1 convs) on [L, CIN] rows.  One thread per position, weights through the scalar path.
//   plain : y  = W x + b
//   tail  : y += W relu((x - mean) * rstd) + b     (second conv of mlp_2layer / Contextnorm on top of the residual branch)
// =====================================================================================================
template <int CIN, int COUT, bool NORM, bool ACCUM>
__global__ __launch_bounds__(256) void linear_kernel(const float *__restrict__ x, int L, const float *__restrict__ W,
                                                     const float *__restrict__ b, const float *__restrict__ mean_rstd,
                                                     float *__restrict__ y, int ochunk, const int *__restrict__ seg_off = nullptr,
                                                     int n_seg = 1, int mult = 1) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int pp = p < L ? p : L - 1;
    float xi[CIN];
    if (NORM && seg_off) mean_rstd += (size_t)seg_of(seg_off, n_seg, pp / mult) * 2 * CIN;     // statistics of this row's pair
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

// The same layer for many rows (several pairs stacked: L >= 65536 and every workgroup owns all output channels): the one-thread-per-row
// form reads and writes rows at a stride of CIN / COUT floats per lane, which the memory pipeline serves line by line.  Here the
// workgroup's 256 x CIN input tile and 256 x 32 output tiles go through LDS (pitch 33: conflict-free for the row-per-lane accesses) so
// that every global access is a run of 128 contiguous bytes per row; the arithmetic (one fmaf chain per output, c ascending, starting
// from the bias) is the per-row kernel's, bit for bit.
template <int CIN, int COUT, bool NORM, bool ACCUM>
__global__ __launch_bounds__(256) void linear_tiled_kernel(const float *__restrict__ x, int L, const float *__restrict__ W,
                                                           const float *__restrict__ b, const float *__restrict__ mean_rstd,
                                                           float *__restrict__ y, const int *__restrict__ seg_off, int n_seg, int mult) {
    static_assert(CIN % 4 == 0 && COUT % 32 == 0, "linear_tiled_kernel: shapes");
    __shared__ float st[256 * 33];
    const int tid = threadIdx.x;
    const int p0 = blockIdx.x * 256;
    const int p = p0 + tid, pp = p < L ? p : L - 1;
    if (NORM && seg_off) mean_rstd += (size_t)seg_of(seg_off, n_seg, pp / mult) * 2 * CIN;
    float xi[CIN];
#pragma unroll
    for (int ch = 0; ch < CIN; ch += 32) {
        const int w = CIN - ch < 32 ? CIN - ch : 32;         // floats of this chunk per row (multiple of 4)
        const int w4 = w / 4;
        for (int f = tid; f < 256 * w4; f += 256) {
            const int row = f / w4, c4 = f - row * w4;
            const int pr = p0 + row < L ? p0 + row : L - 1;
            const float4 v = *reinterpret_cast<const float4 *>(x + (size_t)pr * CIN + ch + c4 * 4);
            float *d = st + row * 33 + c4 * 4;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 32; ++c)
            if (ch + c < CIN) {
                float v = st[tid * 33 + c];
                if (NORM) v = fmaxf((v - mean_rstd[ch + c]) * mean_rstd[CIN + ch + c], 0.f);
                xi[ch + c] = v;
            }
        __syncthreads();
    }
#pragma unroll 1
    for (int o0 = 0; o0 < COUT; o0 += 32) {
        float acc[32];
#pragma unroll
        for (int oo = 0; oo < 32; ++oo) {
            float a = b[o0 + oo];
#pragma unroll
            for (int c = 0; c < CIN; ++c) a = fmaf(xi[c], W[(o0 + oo) * CIN + c], a);
            acc[oo] = a;
        }
#pragma unroll
        for (int oo = 0; oo < 32; ++oo) st[tid * 33 + oo] = acc[oo];
        __syncthreads();
        for (int f = tid; f < 256 * 32; f += 256) {
            const int row = f >> 5, o = f & 31;
            if (p0 + row < L) {
                float *yo = y + (size_t)(p0 + row) * COUT + o0 + o;
                *yo = ACCUM ? *yo + st[row * 33 + o] : st[row * 33 + o];
            }
        }
        __syncthreads();
    }
}

inline int linear_ochunk(int L, int Cout) {
    int chunk = Cout;
    while (chunk > 4 && (long long)L * (Cout / chunk) < 65536) chunk /= 2;
    return chunk;
}

// InstanceNorm statistics: per channel over all L positions (biased variance), accumulated in fp64, two stages,
// fixed reduction order (deterministic).  mean_rstd = [mean (C), 1/sqrt(var+eps) (C)].
// With segments (seg_off [P+1] in points, `mult` rows per point) blockIdx.y is the pair and every pair keeps the block count and the
// strided assignment it would have alone, so the statistics are bitwise those of the one-pair launch.
__global__ __launch_bounds__(256) void in_stats_partial_kernel(const float *__restrict__ h, int L, int C, double *__restrict__ part,
                                                               const int *__restrict__ seg_off, int mult) {
    extern __shared__ double sh[];                  // [256][2]
    const int lanes = 256 / C;                      // position lanes per block (C <= 128 and divides 256)
    const int r0 = seg_off ? seg_off[blockIdx.y] * mult : 0, Lp = seg_off ? seg_off[blockIdx.y + 1] * mult - r0 : L;
    int nblk = (Lp + lanes - 1) / lanes;
    if (nblk > 256) nblk = 256;
    if ((int)blockIdx.x >= nblk) return;
    const int c = threadIdx.x % C, pl = threadIdx.x / C;
    double s = 0, s2 = 0;
    if (pl < lanes)
        for (int p = blockIdx.x * lanes + pl; p < Lp; p += nblk * lanes) {
            const double v = h[(size_t)(r0 + p) * C + c];
            s += v; s2 += v * v;
        }
    sh[threadIdx.x * 2] = s; sh[threadIdx.x * 2 + 1] = s2;
    __syncthreads();
    if (threadIdx.x < C) {
        double a = 0, a2 = 0;
        for (int q = 0; q < lanes; ++q) { a += sh[(q * C + threadIdx.x) * 2]; a2 += sh[(q * C + threadIdx.x) * 2 + 1]; }
        double *po = part + (size_t)blockIdx.y * 256 * C * 2;
        po[((size_t)blockIdx.x * C + threadIdx.x) * 2] = a;
        po[((size_t)blockIdx.x * C + threadIdx.x) * 2 + 1] = a2;
    }
}

__global__ __launch_bounds__(256) void in_stats_final_kernel(const double *__restrict__ part, int L, int C, float eps,
                                                             float *__restrict__ mean_rstd, const int *__restrict__ seg_off, int mult) {
    __shared__ double sa[256], sb[256];
    const int c = blockIdx.x;                       // one block per (channel, pair); fixed-order tree => deterministic
    const int Lp = seg_off ? (seg_off[blockIdx.y + 1] - seg_off[blockIdx.y]) * mult : L;
    const int lanes = 256 / C;
    int nblk = (Lp + lanes - 1) / lanes;
    if (nblk > 256) nblk = 256;
    part += (size_t)blockIdx.y * 256 * C * 2;
    mean_rstd += (size_t)blockIdx.y * 2 * C;
    double a = 0, a2 = 0;
    if ((int)threadIdx.x < nblk) { a = part[((size_t)threadIdx.x * C + c) * 2]; a2 = part[((size_t)threadIdx.x * C + c) * 2 + 1]; }
    sa[threadIdx.x] = a; sb[threadIdx.x] = a2;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { sa[threadIdx.x] += sa[threadIdx.x + s]; sb[threadIdx.x] += sb[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double mu = sa[0] / Lp;
        double var = sb[0] / Lp - mu * mu;
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

// l2norm_rows_kernel for 32-channel rows through an LDS tile (coalesced 128-byte rows in and out, the same sequential sum per row)
__global__ __launch_bounds__(256) void l2norm_rows32_kernel(const float *__restrict__ x, int L, float *__restrict__ y) {
    __shared__ float st[256 * 33];
    const int tid = threadIdx.x, p0 = blockIdx.x * 256;
    for (int f = tid; f < 256 * 8; f += 256) {
        const int row = f >> 3, c4 = f & 7;
        const int pr = p0 + row < L ? p0 + row : L - 1;
        const float4 v = *reinterpret_cast<const float4 *>(x + (size_t)pr * 32 + c4 * 4);
        float *d = st + row * 33 + c4 * 4;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    float v[32], s = 0.f;
#pragma unroll
    for (int c = 0; c < 32; ++c) { v[c] = st[tid * 33 + c]; s = fmaf(v[c], v[c], s); }
    const float r = sqrtf(s);
#pragma unroll
    for (int c = 0; c < 32; ++c) st[tid * 33 + c] = v[c] / r;
    __syncthreads();
    for (int f = tid; f < 256 * 32; f += 256) {
        const int row = f >> 5, c = f & 31;
        if (p0 + row < L) y[(size_t)(p0 + row) * 32 + c] = st[row * 33 + c];
    }
}

__global__ __launch_bounds__(256) void colmax_partial_kernel(const float *__restrict__ x, int L, int C, float *__restrict__ part,
                                                             const int *__restrict__ seg_off) {
    // block (b, pair) reduces the pair's rows b, b+grid, ... ; thread c < C owns a column
    const int c = threadIdx.x;
    if (c >= C) return;
    const int r0 = seg_off ? seg_off[blockIdx.y] : 0, Lp = seg_off ? seg_off[blockIdx.y + 1] - r0 : L;
    float mx = -__builtin_inff();
    for (int p = blockIdx.x; p < Lp; p += gridDim.x) mx = fmaxf(mx, x[(size_t)(r0 + p) * C + c]);
    part[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * C + c] = mx;
}

__global__ __launch_bounds__(256) void colmax_final_kernel(const float *__restrict__ part, int nblk, int C, float *__restrict__ out) {
    __shared__ float sm[256];
    const int c = blockIdx.x;
    part += (size_t)blockIdx.y * nblk * C;
    sm[threadIdx.x] = (int)threadIdx.x < nblk ? part[(size_t)threadIdx.x * C + c] : -__builtin_inff();
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sm[threadIdx.x] = fmaxf(sm[threadIdx.x], sm[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[(size_t)blockIdx.y * C + c] = sm[0];
}

// ctx[p] = [R[p,0..59], colmax[pair of p][0..59]]
__global__ __launch_bounds__(256) void context_kernel(const float *__restrict__ R, const float *__restrict__ cmax, int m, float *__restrict__ ctx,
                                                      const int *__restrict__ seg_off, int n_seg) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m * 120) return;
    const int p = i / 120, c = i - p * 120;
    if (c >= 60 && seg_off) cmax += (size_t)seg_of(seg_off, n_seg, p) * 60;
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
// Batched form (several pairs per launch): blockIdx.z = pair; segR/segC [P+1] are the row offsets of the two descriptor lists, every
// pair owns a slab of `slab` floats of the workspace and R, C are re-derived per pair.  segR == nullptr: the one-pair call.
__global__ __launch_bounds__(256) void ot_build_kernel(const float *__restrict__ rowvec, int R, const float *__restrict__ colvec, int C,
                                                       float alpha, int rows_per_block, float *__restrict__ M, int ld,
                                                       const int *__restrict__ segR, const int *__restrict__ segC, size_t slab) {
    if (segR) {
        const int r0 = segR[blockIdx.z], c0 = segC[blockIdx.z];
        R = segR[blockIdx.z + 1] - r0; C = segC[blockIdx.z + 1] - c0;
        rowvec += (size_t)r0 * RM_F; colvec += (size_t)c0 * RM_F; M += blockIdx.z * slab;
        if (R <= 0 || C <= 0) return;
    }
    const int c = blockIdx.x * 256 + threadIdx.x;          // column in [0, C]  (C = dustbin); the pitch padding (C, ld) is zeroed so that
    if ((int)(blockIdx.x * 256) >= ld) return;             // kernels that stream whole float4 pieces of a row only ever see finite values
    const int cc = c < C ? c : C - 1;
    float t[RM_F];
#pragma unroll
    for (int f = 0; f < RM_F; ++f) t[f] = colvec[(size_t)cc * RM_F + f];
    const int r0 = blockIdx.y * rows_per_block, r1 = min(r0 + rows_per_block, R + 1);
    if (c >= ld) return;
    for (int r = r0; r < r1; ++r) {
        float v = c > C ? 0.f : alpha;
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

// Per-pair geometry of the batched Sinkhorn passes: pair = blockIdx.y, matrix rows = seg_rows (+1 dustbin), columns = seg_cols (+1).
struct OtBatch {
    const int *seg_rows, *seg_cols;      // nullptr = one pair
    size_t slab;                         // floats between the pairs' workspaces
    const float *consts;                 // per pair: -log(m+n), log(#columns of this pass) -- host logf values, as in the one-pair call
};

// out[i] = log_a(i) - LSE_j(Z[i,j] + vec[j]);  log_a = normc for i < R-1, last_extra + normc for the dustbin row
// One WAVE per row (four rows per workgroup, no LDS, no barrier): a lane streams float4 pieces of its row with an online
// (max, sum-of-exp) pair updated once per four elements, then the 64 pairs are merged by a butterfly.
__global__ __launch_bounds__(256) void row_lse_kernel(const float *__restrict__ Z, int R, int C, int ld, const float *__restrict__ vec,
                                                      float normc, float last_extra, float *__restrict__ out, OtBatch ob) {
    if (ob.seg_rows) {
        R = ob.seg_rows[blockIdx.y + 1] - ob.seg_rows[blockIdx.y] + 1;
        C = ob.seg_cols[blockIdx.y + 1] - ob.seg_cols[blockIdx.y] + 1;
        if ((int)(blockIdx.x * 4) >= R || R <= 1 || C <= 1) return;
        Z += blockIdx.y * ob.slab; vec += blockIdx.y * ob.slab; out += blockIdx.y * ob.slab;
        normc = ob.consts[blockIdx.y * 2]; last_extra = ob.consts[blockIdx.y * 2 + 1];
    }
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= R) return;
    const float *row = Z + (size_t)i * ld;          // ld is a multiple of 4: rows are 16-byte aligned
    float mx = -__builtin_inff(), s = 0.f;
    const int C4 = C & ~3;
#pragma unroll 4
    for (int j = lane * 4; j < C4; j += 256) {
        const float4 z = *reinterpret_cast<const float4 *>(row + j);
        const float4 v = *reinterpret_cast<const float4 *>(vec + j);
        const float x0 = z.x + v.x, x1 = z.y + v.y, x2 = z.z + v.z, x3 = z.w + v.w;
        const float m4 = fmaxf(fmaxf(x0, x1), fmaxf(x2, x3));
        if (m4 > mx) { s *= __expf(mx - m4); mx = m4; }                      // exp(-inf) = 0 on the first piece
        s += (__expf(x0 - mx) + __expf(x1 - mx)) + (__expf(x2 - mx) + __expf(x3 - mx));
    }
    if (lane < C - C4) {
        const float x = row[C4 + lane] + vec[C4 + lane];
        if (x > mx) { s = s * __expf(mx - x) + 1.0f; mx = x; }
        else s += __expf(x - mx);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float omx = __shfl_xor(mx, o), os = __shfl_xor(s, o);
        const float nm = fmaxf(mx, omx);
        s = (mx == -__builtin_inff() ? 0.f : s * __expf(mx - nm)) + (omx == -__builtin_inff() ? 0.f : os * __expf(omx - nm));
        mx = nm;
    }
    if (lane == 0) out[i] = (i == R - 1 ? last_extra + normc : normc) - (mx + __logf(s));
}

// One Sinkhorn iteration in ONE pass over the coupling matrix (batched path): a workgroup owns 32 consecutive rows and every
// thread a fixed set of columns (4 consecutive per 1024).  Per row i the workgroup reduces the row in the log domain like the
// two-matrix pass: M_i = max_j (Z[i,j] + v[j]), e_ij = exp(Z[i,j] + v[j] - M_i), S_i = sum_j e_ij, u[i] = log_mu(i) - M_i - log S_i.
// The column update needs LSE_i(Z[i,j] + u[i]); since Z[i,j] + u[i] = log e_ij + (M_i + u[i]) - v[j], it is
//     LSE_i(Z[i,j] + u[i]) = log( sum_i e_ij * a_i ) - v[j]        with a_i = exp(M_i + u[i]) = mu_i / S_i   (<= 1),
// so the e_ij already in registers are folded into the column sums with ONE fma per element and no further exponential: half the
// transcendental work of two log-domain passes and the matrix read once per iteration instead of twice.  After its rows the workgroup
// writes one partial sum per column; ot_col_merge_kernel adds the row blocks' partials in a fixed order and updates v.  Every term is
// in [0, 1]; a column whose terms all underflowed (it would have to sit e^-87 below the row maxima in EVERY row, the dustbin row included,
// i.e. v[j] < max v - 87) is clamped to the smallest normal number instead of producing log 0.  The transposed copy is only read by
// the final column arg-max.
constexpr int OT_RB = 32;

// Wave-wide reductions on the DPP data path (no LDS crossbar): quad swaps, half-row / row mirrors, then the row broadcasts of gfx9;
// the full result is in lane 63 and is handed back through a scalar register.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, x), __builtin_bit_cast(int, x), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_max(float x) {
    x = fmaxf(x, dpp_mov<0xB1, 0xf>(x));       // quad_perm [1,0,3,2]
    x = fmaxf(x, dpp_mov<0x4E, 0xf>(x));       // quad_perm [2,3,0,1]
    x = fmaxf(x, dpp_mov<0x141, 0xf>(x));      // row_half_mirror
    x = fmaxf(x, dpp_mov<0x140, 0xf>(x));      // row_mirror: every lane of a 16-lane row holds the row's maximum
    x = fmaxf(x, dpp_mov<0x142, 0xa>(x));      // row_bcast:15 into rows 1 and 3
    x = fmaxf(x, dpp_mov<0x143, 0xc>(x));      // row_bcast:31 into rows 2 and 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}
__device__ __forceinline__ float wave_sum(float x) {
    x += dpp_mov<0xB1, 0xf>(x);
    x += dpp_mov<0x4E, 0xf>(x);
    x += dpp_mov<0x141, 0xf>(x);
    x += dpp_mov<0x140, 0xf>(x);
    {   // rows that are not written keep their own value in the moved operand; only the written rows matter downstream
        const float t = dpp_mov<0x142, 0xa>(x);
        x = ((threadIdx.x >> 4) & 1) ? x + t : x;
    }
    {
        const float t = dpp_mov<0x143, 0xc>(x);
        x = ((threadIdx.x >> 5) & 1) ? x + t : x;
    }
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}

template <int NV>
__global__ __launch_bounds__(256) void ot_fused_pass_kernel(const float *__restrict__ Z, int ld, const float *__restrict__ vvec,
                                                            float *__restrict__ uout, float *__restrict__ part, size_t part_stride, OtBatch ob) {
    __shared__ float smx[2][4], ssum[2][4];
    const int pair = blockIdx.y;
    const int R = ob.seg_rows[pair + 1] - ob.seg_rows[pair] + 1, C = ob.seg_cols[pair + 1] - ob.seg_cols[pair] + 1;
    const int r0 = blockIdx.x * OT_RB;
    if (r0 >= R) return;
    const int r1 = min(r0 + OT_RB, R);
    Z += pair * ob.slab; vvec += pair * ob.slab; uout += pair * ob.slab;
    part += pair * part_stride + (size_t)blockIdx.x * ld;
    const float normc = ob.consts[pair * 2], last_extra = ob.consts[pair * 2 + 1];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float NEG = -__builtin_inff();
    float vv[NV][4], acc[NV][4];
#pragma unroll
    for (int t = 0; t < NV; ++t) {
        const int j = t * 1024 + tid * 4;
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j < ld) q = *reinterpret_cast<const float4 *>(vvec + j);
        vv[t][0] = q.x; vv[t][1] = q.y; vv[t][2] = q.z; vv[t][3] = q.w;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (j + e >= C) vv[t][e] = NEG;                  // padding columns: Z + v = -inf, their exponentials are exactly 0
            acc[t][e] = 0.f;
        }
    }
    // rows stream through a ring of four register buffers: three rows are in flight while one is reduced (one row per workgroup in
    // flight is ~10 KB; memory latency under load is several microseconds, so depth is what buys bandwidth here)
    float4 buf[4][NV];
    // unconditional loads (pieces beyond the pitch re-read the row's last piece and are masked by `< C` below): exec-masked loads make
    // the compiler's wait-count bookkeeping fall back to vmcnt(0), which would drain the whole prefetch ring at every row
    int jc[NV];
#pragma unroll
    for (int t = 0; t < NV; ++t) jc[t] = min(t * 1024 + tid * 4, ld - 4);
    auto load_row = [&](int r, float4 (&dst)[NV]) {
        const float *row = Z + (size_t)(r < r1 ? r : r1 - 1) * ld;
#pragma unroll
        for (int t = 0; t < NV; ++t) dst[t] = *reinterpret_cast<const float4 *>(row + jc[t]);
    };
    float my_u = 0.f;
    auto process = [&](int r, const float4 (&cur)[NV]) {
        float x[NV][4];
        float mx = NEG;
#pragma unroll
        for (int t = 0; t < NV; ++t) {
            const float z[4] = {cur[t].x, cur[t].y, cur[t].z, cur[t].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                x[t][e] = z[e] + vv[t][e];
                mx = fmaxf(mx, x[t][e]);
            }
        }
        // exponentials relative to the WAVE's maximum first (no cross-wave dependency), one barrier to combine the four (max, sum)
        // pairs, then the wave's factor exp(M_w - M) rides on a_i
        mx = wave_max(mx);
        const float LOG2E = 1.44269504088896340736f;
        const float mneg = mx > NEG ? -mx * LOG2E : 0.f;     // exp(x - M_w) = 2^(x log2 e - M_w log2 e): one fma + v_exp_f32 per element
        float sm = 0.f;
#pragma unroll
        for (int t = 0; t < NV; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) { x[t][e] = __builtin_amdgcn_exp2f(fmaf(x[t][e], LOG2E, mneg)); sm += x[t][e]; }   // e_ij * exp(M - M_w); 0 for the padding
        sm = wave_sum(sm);
        const int par = r & 1;
        if (lane == 0) { smx[par][w] = mx; ssum[par][w] = sm; }
        __syncthreads();
        const float M = fmaxf(fmaxf(smx[par][0], smx[par][1]), fmaxf(smx[par][2], smx[par][3]));      // finite: the dustbin column is
        float S = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) S += smx[par][q] > NEG ? ssum[par][q] * __expf(smx[par][q] - M) : 0.f;
        const float lmu = r == R - 1 ? last_extra + normc : normc;
        const float logS = __logf(S);
        if (tid == r - r0) my_u = lmu - (M + logS);                // written after the loop: a store inside it would be waited on (vmcnt) and drain the prefetch
        const float ai = (mx > NEG && r < r1) ? __expf((lmu - logS) + (mx - M)) : 0.f;                            // exp(M_w + u[i]) = (mu_i / S_i) exp(M_w - M)
#pragma unroll
        for (int t = 0; t < NV; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[t][e] = fmaf(x[t][e], ai, acc[t][e]);
    };
    // program order of the row loads is pinned (VMEM operations complete in order, so "row r has landed" is a count of younger loads)
    load_row(r0, buf[0]); __builtin_amdgcn_sched_barrier(0);
    load_row(r0 + 1, buf[1]); __builtin_amdgcn_sched_barrier(0);
    load_row(r0 + 2, buf[2]); __builtin_amdgcn_sched_barrier(0);
    for (int r = r0; r < r1; r += 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {                           // branch-free body (rows past the block re-read its last row and add
            load_row(r + k + 3, buf[(k + 3) & 3]);              // nothing): one basic block keeps the wait counts exact
            __builtin_amdgcn_sched_barrier(0);
            process(r + k, buf[k]);
        }
    }
    if (tid < r1 - r0) uout[r0 + tid] = my_u;
#pragma unroll
    for (int t = 0; t < NV; ++t) {
        const int j = t * 1024 + tid * 4;
        if (j < ld) *reinterpret_cast<float4 *>(part + j) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
    }
}

// v[j] <- log_nu(j) - log( sum over the row blocks of the partial column sums ) + v[j]   (see ot_fused_pass_kernel).  A workgroup owns 64
// columns; its four waves add every fourth row block, then the four sums are added in wave order.
// If every coupling of a column underflowed in the row pass (its entries lie more than ~87 below the row maxima in EVERY row -- a score
// spread no trained matcher produces, but a legal input), the accumulated sum is 0 and its logarithm would be a clamp, not the value: that
// column is then evaluated exactly in the log domain from the transposed matrix, like the one-pair path (roreg_sinkhorn) does for all columns.
__global__ __launch_bounds__(256) void ot_col_merge_kernel(const float *__restrict__ part, size_t part_stride, int ld, float *__restrict__ vio,
                                                           OtBatch ob, const float *__restrict__ Z0T, int ldt, const float *__restrict__ u) {
    __shared__ float sh[4][64];
    const int pair = blockIdx.y;            // ob is the COLUMN pass geometry: its "rows" are the matrix columns
    const int C = ob.seg_rows[pair + 1] - ob.seg_rows[pair] + 1, R = ob.seg_cols[pair + 1] - ob.seg_cols[pair] + 1;
    if ((int)(blockIdx.x * 64) >= C) return;
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + lane, jj = j < C ? j : C - 1;
    part += pair * part_stride + jj;
    const int nblk = (R + OT_RB - 1) / OT_RB;
    float T = 0.f;
#pragma unroll 4
    for (int b = g; b < nblk; b += 4) T += part[(size_t)b * ld];
    sh[g][lane] = T;
    __syncthreads();
    if (g == 0 && j < C) {
        T = (sh[0][lane] + sh[1][lane]) + (sh[2][lane] + sh[3][lane]);
        const float normc = ob.consts[pair * 2], last_extra = ob.consts[pair * 2 + 1];       // column constants: (normc, log m)
        float *v = vio + pair * ob.slab + j;
        const float lognu = j == C - 1 ? last_extra + normc : normc;
        if (T > 1.17549435e-38f) {
            *v = lognu - __logf(T) + *v;
        } else {                                             // all-underflowed column: exact log-sum-exp over the column (rare, one lane)
            const float *zc = Z0T + pair * ob.slab + (size_t)j * ldt, *uu = u + pair * ob.slab;
            float mx = -__builtin_inff();
            for (int i = 0; i < R; ++i) mx = fmaxf(mx, zc[i] + uu[i]);
            float sum = 0.f;
            for (int i = 0; i < R; ++i) sum += __expf(zc[i] + uu[i] - mx);
            *v = lognu - (mx + __logf(sum));
        }
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

// the same with the normalisation constant read from the device (the stacked call's constants array; one pair)
__global__ __launch_bounds__(256) void ot_final_batch1_kernel(const float *__restrict__ Z0, int ld, int m, int n, const float *__restrict__ u,
                                                              const float *__restrict__ v, const float *__restrict__ consts, float *__restrict__ Z) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t tot = (size_t)(m + 1) * (n + 1);
    if (i >= tot) return;
    const int r = (int)(i / (n + 1)), c = (int)(i - (size_t)r * (n + 1));
    Z[i] = Z0[(size_t)r * ld + c] + u[r] + v[c] - consts[0];
}

// arg-max over the first C-1 columns of rows 0..R-2 of (Z0[i,j] + vec[j]) (+ rowc[i] - normc for the value)
template <bool ROWC_FIRST>
__global__ __launch_bounds__(256) void row_argmax_kernel(const float *__restrict__ Z, int R, int C, int ld, const float *__restrict__ vec,
                                                         const float *__restrict__ rowc, float normc, int64_t *__restrict__ idx,
                                                         float *__restrict__ val, OtBatch ob) {
    __shared__ float sv[256];
    __shared__ int si[256];
    if (ob.seg_rows) {
        const int r0 = ob.seg_rows[blockIdx.y];
        R = ob.seg_rows[blockIdx.y + 1] - r0 + 1;
        C = ob.seg_cols[blockIdx.y + 1] - ob.seg_cols[blockIdx.y] + 1;
        if ((int)blockIdx.x >= R - 1 || C <= 1) return;
        Z += blockIdx.y * ob.slab; vec += blockIdx.y * ob.slab; rowc += blockIdx.y * ob.slab;
        idx += r0; val += r0;                          // read-outs are concatenated over the pairs
        normc = ob.consts[blockIdx.y * 2];
    }
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

// ---- the read-out without a matrix ---------------------------------------------------------------------------------------------------
// Row and column arg-max of x_ij = ((Z0_ij + u_i) + v_j) - norm over the points (network/rot_coh_match.py:369-371) straight from the descriptors:
// Z0_ij = the float32 fmaf chain over the 32 channels that ot_build_kernel evaluates -- and that v_mfma_f32_32x32x2_f32 evaluates bit for bit
// (csrc/linear_chain.hip) -- so a workgroup takes a 32-row strip of sources, its four wavefronts walk the 32-column tiles of targets (16 chained
// MFMAs each), form x in the reference's association and keep, per accumulator element, packed keys (orderable float bits << 32 | ~index): a
// 64-bit maximum is "larger value, then lower index" -- the first maximum, as torch.max returns it.  Rows: reduced over the lanes and the four
// wavefronts at the end of the strip.  Columns: the strip's best row per column goes to a global 64-bit atomic maximum (order-independent).
// The two (m+1) x (n+1) matrices per pair that served only this read-out (2 x 100 MB at 5000 points) are neither built nor read.
typedef float f32x16_rd __attribute__((ext_vector_type(16)));
__device__ __forceinline__ unsigned long long rd_key(float x, unsigned idx) {
    const unsigned b = __float_as_uint(x);
    const unsigned o = (b & 0x80000000u) ? ~b : (b | 0x80000000u);         // order-preserving map of finite floats and infinities
    return ((unsigned long long)o << 32) | (unsigned long long)(0xffffffffu - idx);
}
__device__ __forceinline__ float rd_val(unsigned long long k) {
    const unsigned o = (unsigned)(k >> 32);
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}
__global__ __launch_bounds__(256) void ot_argmax_mfma_kernel(const float *__restrict__ S, const float *__restrict__ T, const int *__restrict__ segS,
                                                             const int *__restrict__ segT, const float *__restrict__ u, const float *__restrict__ v,
                                                             size_t uv_stride, const float *__restrict__ consts, unsigned long long *__restrict__ colbest,
                                                             int64_t *__restrict__ i0, float *__restrict__ val0) {
    const int pair = blockIdx.y, strip = blockIdx.x;
    const int s0 = segS[pair], m = segS[pair + 1] - s0, t0 = segT[pair], n = segT[pair + 1] - t0;
    if (strip * 32 >= m) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 31, h = lane >> 5;
    const float normc = consts[pair * 2];
    const float *up = u + pair * uv_stride, *vp = v + pair * uv_stride;
    float a[16];                                                          // A fragments of the strip: row strip * 32 + j, channels 2 k + h
    {
        const float4 *sr = reinterpret_cast<const float4 *>(S + (size_t)(s0 + min(strip * 32 + j, m - 1)) * RM_F);
#pragma unroll
        for (int q = 0; q < 8; ++q) { const float4 x4 = sr[q]; a[2 * q] = h ? x4.y : x4.x; a[2 * q + 1] = h ? x4.w : x4.z; }
    }
    float ur[16];
    bool rok[16];
    unsigned long long bk[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = strip * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
        rok[r] = row < m;
        ur[r] = up[min(row, m - 1)];
        bk[r] = 0ull;
    }
    const int ntile = (n + 31) / 32;
    for (int jt = w; jt < ntile; jt += 4) {
        const int col = jt * 32 + j;
        const bool cok = col < n;
        float bq[16];
        {
            const float4 *tr = reinterpret_cast<const float4 *>(T + (size_t)(t0 + min(col, n - 1)) * RM_F);
#pragma unroll
            for (int q = 0; q < 8; ++q) { const float4 x4 = tr[q]; bq[2 * q] = h ? x4.y : x4.x; bq[2 * q + 1] = h ? x4.w : x4.z; }
        }
        const float vj = vp[min(col, n - 1)];
        f32x16_rd acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], bq[kk], acc, 0, 0, 0);      // = fmaf chain over channels 0..31 from 0
        unsigned long long ck = 0ull;                                     // this lane's best row for column `col`
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float x = (((acc[r] + ur[r]) + vj) - normc) + 0.f;      // (+ 0: -0 becomes +0, so that equal floats have equal keys)
            if (cok && rok[r] && x == x) {                                // (NaN never wins: the sentinel stays)
                const unsigned long long kr = rd_key(x, (unsigned)col);
                if (kr > bk[r]) bk[r] = kr;
                const unsigned long long kc = rd_key(x, (unsigned)(strip * 32 + 8 * (r >> 2) + 4 * h + (r & 3)));
                if (kc > ck) ck = kc;
            }
        }
        const unsigned long long other = __shfl_xor(ck, 32);
        if (other > ck) ck = other;
        if (h == 0 && cok && ck) atomicMax(colbest + t0 + col, ck);
    }
    // rows: over the 32 lanes of each half-wave, then over the four wavefronts
    __shared__ unsigned long long srow[4][32];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        unsigned long long k = bk[r];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) { const unsigned long long x = __shfl_xor(k, o); if (x > k) k = x; }
        if (j == 0) srow[w][8 * (r >> 2) + 4 * h + (r & 3)] = k;
    }
    __syncthreads();
    if (tid < 32) {
        const int row = strip * 32 + tid;
        if (row < m) {
            unsigned long long k = srow[0][tid];
#pragma unroll
            for (int q = 1; q < 4; ++q) if (srow[q][tid] > k) k = srow[q][tid];
            i0[s0 + row] = k ? (int64_t)(0xffffffffu - (unsigned)k) : (int64_t)0x7fffffff;
            val0[s0 + row] = k ? rd_val(k) : -__builtin_inff();
        }
    }
}

__global__ __launch_bounds__(256) void ot_colbest_decode_kernel(const unsigned long long *__restrict__ colbest, long long total, int64_t *__restrict__ i1) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t < total) { const unsigned long long k = colbest[t]; i1[t] = k ? (int64_t)(0xffffffffu - (unsigned)k) : (int64_t)0x7fffffff; }
}

// batched: blockIdx.y = pair; i0/v0/m0/s0 are concatenated by seg0, i1/m1/s1 by seg1; indices stay LOCAL to the pair
__global__ __launch_bounds__(256) void ot_readout_kernel(const int64_t *__restrict__ i0, const float *__restrict__ v0, int m,
                                                         const int64_t *__restrict__ i1, int n, int64_t *__restrict__ m0,
                                                         int64_t *__restrict__ m1, float *__restrict__ s0, float *__restrict__ s1,
                                                         const int *__restrict__ seg0, const int *__restrict__ seg1) {
    if (seg0) {
        const int a = seg0[blockIdx.y], b = seg1[blockIdx.y];
        m = seg0[blockIdx.y + 1] - a; n = seg1[blockIdx.y + 1] - b;
        i0 += a; v0 += a; m0 += a; s0 += a; i1 += b; m1 += b; s1 += b;
    }
    const int t = blockIdx.x * 256 + threadIdx.x;
    // an arg-max over a row of NaNs (non-finite descriptors) leaves the sentinel index: such a point is unmatched, never dereferenced
    if (t < m) {
        const int64_t a = i0[t];
        const bool mu = a >= 0 && a < n && i1[a] == (int64_t)t;
        m0[t] = mu ? a : -1;
        s0[t] = mu ? expf(v0[t]) : 0.f;
    }
    if (t < n) {
        const int64_t a = i1[t];
        const bool mu1 = a >= 0 && a < m && i0[a] == (int64_t)t;           // mutual1; then valid0[a] is the same predicate seen from a
        m1[t] = mu1 ? a : -1;
        s1[t] = mu1 ? expf(v0[a]) : 0.f;
    }
}

}  // namespace

// -----------------------------------------------------------------------------------------------------------
// target slices x row blocks: ~1 workgroup per CU is enough for one pair (more slices only lengthen the merge); stacked pairs are
// bandwidth hungrier per launch and want ~4 per CU
static int topk_slices(int gx_total, int max_n, bool stacked) {
    static const int target_env = getenv("ROREG_TOPK_TARGET_WGS") ? atoi(getenv("ROREG_TOPK_TARGET_WGS")) : 0;      // (measurements)
    const int target = target_env > 0 ? target_env : (stacked ? 1024 : 320);
    int slices = (target + gx_total - 1) / gx_total;
    if (slices > (max_n + 63) / 64) slices = (max_n + 63) / 64;
    return slices < 1 ? 1 : slices;
}

extern "C" size_t roreg_topk_dot_workspace_size(int m, int n, int k) {
    // enough for any segmentation of m x n: the slice count only shrinks when the rows are spread over several pairs
    const int slices = topk_slices((m + 255) / 256, n, true);
    return (size_t)slices * m * k * 2;       // floats (values) + ints (indices), 4 bytes each
}

extern "C" int roreg_topk_dot(const float *A, int m, const float *B, int n, int k, int64_t *idx_out, float *val_out, float *ws,
                              size_t ws_floats, const int32_t *segA, const int32_t *segB, int n_seg, int max_m, int max_n, void *stream) {
    ROREG_REQUIRE(A && B && idx_out && ws && m > 0 && n > 0, "roreg_topk_dot: bad arguments");
    ROREG_REQUIRE(k == 16 || k == 8 || k == 1, "roreg_topk_dot: k must be 16, 8 or 1 (got %d)", k);
    ROREG_REQUIRE((segA == nullptr) == (segB == nullptr), "roreg_topk_dot: both segment tables or none");
    if (!segA) { n_seg = 1; max_m = m; max_n = n; }
    ROREG_REQUIRE(n_seg > 0 && max_m > 0 && max_n > 0 && max_m <= m && max_n <= n, "roreg_topk_dot: bad segment description");
    ROREG_REQUIRE(segA || k <= n, "roreg_topk_dot: k > n");      // with segments the caller guarantees k <= every pair's target count
    // ROREG_TOPK_VALU=1: the vector-pipe kernel (one thread per source row); default: the dot products as float32 MFMA chains, bitwise the same lists
    static const bool valu = getenv("ROREG_TOPK_VALU") && atoi(getenv("ROREG_TOPK_VALU")) == 1;
    // ROREG_TOPK_PACKED=1: the packed list maintenance (measured: 27 % faster at k = 16 on random descriptors, no gain inside the pipeline -- off by default)
    static const bool packed = getenv("ROREG_TOPK_PACKED") && atoi(getenv("ROREG_TOPK_PACKED")) == 1;
    static const bool lds_tiles = !(getenv("ROREG_TOPK_LDS") && atoi(getenv("ROREG_TOPK_LDS")) == 0);       // (0: every wavefront fetches its own tiles, for A/B)
    const int gx = valu ? (max_m + 255) / 256 : (max_m + 127) / 128;
    int slices = topk_slices((max_m + 255) / 256 * n_seg, max_n, segA != nullptr);
    if (!segA) {                                 // one pair: the slice width the kernel derives must cover n with this many slices
        const int slice = (n + slices - 1) / slices;
        slices = (n + slice - 1) / slice;
    }
    ROREG_REQUIRE(ws_floats >= (size_t)slices * m * k * 2, "roreg_topk_dot: workspace too small");
    float *pv = ws;
    int *pi = reinterpret_cast<int *>(ws + (size_t)slices * m * k);
    hipStream_t s = roreg::as_stream(stream);
    const int gm = (m + 255) / 256;
    if (!valu && !roreg::mfma_chain_verified(s)) return 4;          // (one-time self-check of the fma-chain property; roreg_last_error() says what to switch)
    roreg::ProfScope prof(roreg::PROF_TOPK, s);
#define RM_TOPK(KK)                                                                                                                  \
    if (valu) hipLaunchKernelGGL(topk_dot_kernel<KK>, dim3(gx, slices, n_seg), dim3(256), 0, s, A, m, B, n, segA, segB, slices, pv, pi);       \
    else if (lds_tiles) hipLaunchKernelGGL(topk_dot_mfma_lds_kernel<KK>, dim3(gx, slices, n_seg), dim3(256), 0, s, A, m, B, n, segA, segB, slices, pv, pi);  \
    else if (packed) hipLaunchKernelGGL((topk_dot_mfma_kernel<true, KK>), dim3(gx, slices, n_seg), dim3(256), 0, s, A, m, B, n, segA, segB, slices, pv, pi);  \
    else hipLaunchKernelGGL((topk_dot_mfma_kernel<false, KK>), dim3(gx, slices, n_seg), dim3(256), 0, s, A, m, B, n, segA, segB, slices, pv, pi);       \
    hipLaunchKernelGGL(topk_merge_kernel<KK>, dim3(gm), dim3(256), 0, s, pv, pi, m, slices, idx_out, val_out);
    if (k == 16) { RM_TOPK(16) } else if (k == 8) { RM_TOPK(8) } else { RM_TOPK(1) }
#undef RM_TOPK
    ROREG_CHECK_LAUNCH("roreg_topk_dot");
    return 0;
}

// The same layers on the matrix cores (csrc/linear_mfma.hip): layers with >= 32 inputs; the 3 -> 64 / 32 position-MLP inputs fall through to
// the vector-pipe kernel.
extern "C" int roreg_linear(const float *x, int L, int Cin, const float *W, const float *b, int Cout, float *y, void *stream);
extern "C" int roreg_linear_mfma(const float *x, int L, int Cin, const float *W, const float *b, int Cout, float *y, void *stream) {
    ROREG_REQUIRE(x && W && b && y && L > 0, "roreg_linear_mfma: bad arguments");
    if (roreg::linear_mfma(x, L, Cin, W, b, Cout, y, roreg::as_stream(stream))) {
        ROREG_CHECK_LAUNCH("roreg_linear_mfma");
        return 0;
    }
    return roreg_linear(x, L, Cin, W, b, Cout, y, stream);
}

extern "C" int roreg_mlp_tail(const float *h, int L, int Cmid, const float *mean_rstd, const float *W2, const float *b2, float *y,
                              const int32_t *seg_off, int n_seg, int mult, void *stream);
extern "C" int roreg_mlp_tail_mfma(const float *h, int L, int Cmid, const float *mean_rstd, const float *W2, const float *b2, float *y,
                                   const int32_t *seg_off, int n_seg, int mult, void *stream) {
    ROREG_REQUIRE(h && mean_rstd && W2 && b2 && y && L > 0, "roreg_mlp_tail_mfma: bad arguments");
    if (!seg_off) { n_seg = 1; mult = 1; }
    if (roreg::linear_tail_mfma(h, L, Cmid, mean_rstd, W2, b2, y, seg_off, n_seg, mult, roreg::as_stream(stream))) {
        ROREG_CHECK_LAUNCH("roreg_mlp_tail_mfma");
        return 0;
    }
    return roreg_mlp_tail(h, L, Cmid, mean_rstd, W2, b2, y, seg_off, n_seg, mult, stream);
}

// 0 (default): the fmaf chains of roreg_linear / roreg_mlp_tail run on the matrix cores (csrc/linear_chain.hip: v_mfma_f32_32x32x2_f32 is a
// float32 fmaf chain over k, bit for bit; since round 6 software-pipelined, lc2_kernel); 1: on the vector pipe (the kernels above); 2: the matrix cores
// through round 5's kernels.  Returns the previous setting.  Same results every way.
static int g_linear_path = 0;
extern "C" int roreg_linear_path(int path) {
    const int prev = g_linear_path != 0 ? g_linear_path : (roreg::linear_chain2_on() ? 0 : 2);
    if (path == 0 || path == 1) g_linear_path = path;
    if (path == 2) g_linear_path = 0;                    // v6: 2 = the matrix cores through round 5's un-pipelined kernels (one launch per convolution; A/B, tests)
    if (path == 0 || path == 2) roreg::linear_chain2_set(path == 0);
    return prev;
}

// The value MLP's first layers without their materialised input: row r of [m * k, 96] = [pos[r] | table[idx[r]] | conf[r / k]] is assembled while
// the kernel stages its rows (value_input_kernel + two reads of its 96-float rows leave the matcher's trace).  Matrix-core path only.
extern "C" int roreg_linear_cat3(const float *pos, const float *table, const int64_t *idx, const float *conf, int m, int k, const float *W, const float *b,
                                 int Cout, float *y, void *stream) {
    ROREG_REQUIRE(pos && table && idx && conf && W && b && y && m > 0 && k > 0, "roreg_linear_cat3: bad arguments");
    ROREG_REQUIRE((long long)m * k < (1ll << 31), "roreg_linear_cat3: %d x %d rows", m, k);
    ROREG_REQUIRE(g_linear_path == 0, "roreg_linear_cat3: the vector-pipe path has no fused form (materialise the rows and call roreg_linear)");
    if (!roreg::mfma_chain_verified(roreg::as_stream(stream))) return 4;
    if (!roreg::linear_chain_cat3(pos, table, idx, conf, m, k, W, b, Cout, y, roreg::as_stream(stream))) {
        roreg::set_error("roreg_linear_cat3: Cout = %d not served (64 | 32)", Cout);
        return 2;
    }
    ROREG_CHECK_LAUNCH("roreg_linear_cat3");
    return 0;
}

// v6: mlp_2layer's first convolution (Cin -> C1, output h) and residual branch (Cin -> 32, output y) in ONE launch -- the same fmaf chains on the
// matrix cores, the input staged once -- plus the per-pair InstanceNorm statistics of h (mean_rstd [n_seg][2 C1]) from per-tile float64 channel sums
// (no kernel reads h again).  pos / table / idx / conf (m, k): the value MLP's assembled rows as in roreg_linear_cat3 (x == NULL then, Cin = 96,
// L = m k).  ws: roreg_mlp_head_workspace(L, n_seg, C1) doubles.  Returns 3 (and launches nothing) for a shape or path it does not serve.
extern "C" size_t roreg_mlp_head_workspace(int L, int n_seg, int C1) { return ((size_t)L / 128 + (size_t)(n_seg > 0 ? n_seg : 1) + 1) * (size_t)C1 * 2; }
extern "C" int roreg_mlp_head(const float *x, const float *pos, const float *table, const int64_t *idx, const float *conf, int m, int k, int L, int Cin,
                              const float *W1, const float *b1, int C1, const float *Wr, const float *br, float *h, float *y, const int32_t *seg_off,
                              int n_seg, int mult, float eps, float *mean_rstd, double *ws, void *stream) {
    ROREG_REQUIRE((x || (pos && table && idx && conf && k > 0 && (long long)m * k == L)) && W1 && b1 && Wr && br && h && y && mean_rstd && ws && L > 0,
                  "roreg_mlp_head: bad arguments");
    if (!seg_off) { n_seg = 1; mult = 1; }
    ROREG_REQUIRE(n_seg > 0 && mult > 0, "roreg_mlp_head: bad segment description");
    if (g_linear_path != 0 || !roreg::linear_chain2_on()) return 3;
    if (!roreg::mfma_chain_verified(roreg::as_stream(stream))) return 4;
    if (!roreg::mlp_head_chain(x ? x : pos, x ? nullptr : table, idx, conf, k, L, Cin, W1, b1, C1, Wr, br, h, y, seg_off, n_seg, mult, eps, mean_rstd, ws,
                               roreg::as_stream(stream)))
        return 3;
    ROREG_CHECK_LAUNCH("roreg_mlp_head");
    return 0;
}

extern "C" int roreg_linear(const float *x, int L, int Cin, const float *W, const float *b, int Cout, float *y, void *stream) {
    ROREG_REQUIRE(x && W && b && y && L > 0, "roreg_linear: bad arguments");
    hipStream_t s = roreg::as_stream(stream);
    if (g_linear_path == 0 && !roreg::mfma_chain_verified(s)) return 4;
    if (g_linear_path == 0 && ((roreg::linear_chain2_on() && roreg::linear_chain2(x, L, Cin, W, b, Cout, y, s)) || roreg::linear_chain(x, L, Cin, W, b, Cout, y, s))) {
        ROREG_CHECK_LAUNCH("roreg_linear");
        return 0;
    }
    const int oc = linear_ochunk(L, Cout);
    const dim3 g((L + 255) / 256, (Cout + oc - 1) / oc), t(256);
#define RM_LIN_T(CI, CO)                                                                                                         \
    if (Cin == CI && Cout == CO && oc == Cout) {                                                                                 \
        hipLaunchKernelGGL((linear_tiled_kernel<CI, CO, false, false>), dim3((L + 255) / 256), t, 0, s, x, L, W, b, nullptr, y,  \
                           nullptr, 1, 1);                                                                                       \
        ROREG_CHECK_LAUNCH("roreg_linear");                                                                                      \
        return 0;                                                                                                                \
    }
    RM_LIN_T(32, 32) RM_LIN_T(96, 64) RM_LIN_T(120, 128) RM_LIN_T(64, 64) RM_LIN_T(96, 32) RM_LIN_T(120, 32) RM_LIN_T(64, 32)
#undef RM_LIN_T
#define RM_LIN(CI, CO)                                                                                      \
    if (Cin == CI && Cout == CO) {                                                                           \
        hipLaunchKernelGGL((linear_kernel<CI, CO, false, false>), g, t, 0, s, x, L, W, b, nullptr, y, oc, nullptr, 1, 1);  \
        ROREG_CHECK_LAUNCH("roreg_linear");                                                              \
        return 0;                                                                                        \
    }
    RM_LIN(32, 32) RM_LIN(96, 64) RM_LIN(3, 64) RM_LIN(120, 128) RM_LIN(64, 64)
    RM_LIN(96, 32) RM_LIN(3, 32) RM_LIN(120, 32) RM_LIN(64, 32)
#undef RM_LIN
    roreg::set_error("roreg_linear: unsupported shape %d -> %d", Cin, Cout);
    return 2;
}

extern "C" int roreg_instnorm_stats(const float *h, int L, int C, float eps, float *mean_rstd, double *ws /* n_seg*2*C*256 doubles */,
                                    const int32_t *seg_off, int n_seg, int mult, void *stream) {
    ROREG_REQUIRE(h && mean_rstd && ws && L > 0, "roreg_instnorm_stats: bad arguments");
    ROREG_REQUIRE(C > 0 && C <= 128 && 256 % C == 0, "roreg_instnorm_stats: C must divide 256 (got %d)", C);
    if (!seg_off) { n_seg = 1; mult = 1; }
    ROREG_REQUIRE(n_seg > 0 && mult > 0, "roreg_instnorm_stats: bad segment description");
    hipStream_t s = roreg::as_stream(stream);
    const int lanes = 256 / C;
    int nblk = 256;
    if (!seg_off) { nblk = (L + lanes - 1) / lanes; if (nblk > 256) nblk = 256; }
    hipLaunchKernelGGL(in_stats_partial_kernel, dim3(nblk, n_seg), dim3(256), 256 * 2 * sizeof(double), s, h, L, C, ws, seg_off, mult);
    hipLaunchKernelGGL(in_stats_final_kernel, dim3(C, n_seg), dim3(256), 0, s, ws, L, C, eps, mean_rstd, seg_off, mult);
    ROREG_CHECK_LAUNCH("roreg_instnorm_stats");
    return 0;
}

extern "C" int roreg_mlp_tail(const float *h, int L, int Cmid, const float *mean_rstd, const float *W2, const float *b2, float *y,
                              const int32_t *seg_off, int n_seg, int mult, void *stream) {
    ROREG_REQUIRE(h && mean_rstd && W2 && b2 && y && L > 0, "roreg_mlp_tail: bad arguments");
    if (!seg_off) { n_seg = 1; mult = 1; }
    hipStream_t s = roreg::as_stream(stream);
    if (g_linear_path == 0 && !roreg::mfma_chain_verified(s)) return 4;
    if (g_linear_path == 0 && ((roreg::linear_chain2_on() && roreg::linear_tail_chain2(h, L, Cmid, mean_rstd, W2, b2, y, seg_off, n_seg, mult, s)) ||
                               roreg::linear_tail_chain(h, L, Cmid, mean_rstd, W2, b2, y, seg_off, n_seg, mult, s))) {
        ROREG_CHECK_LAUNCH("roreg_mlp_tail");
        return 0;
    }
    const int oc = linear_ochunk(L, 32);
    const dim3 g((L + 255) / 256, (32 + oc - 1) / oc), t(256);
    if (oc == 32 && Cmid == 64) hipLaunchKernelGGL((linear_tiled_kernel<64, 32, true, true>), dim3((L + 255) / 256), t, 0, s, h, L, W2, b2, mean_rstd, y, seg_off, n_seg, mult);
    else if (oc == 32 && Cmid == 128) hipLaunchKernelGGL((linear_tiled_kernel<128, 32, true, true>), dim3((L + 255) / 256), t, 0, s, h, L, W2, b2, mean_rstd, y, seg_off, n_seg, mult);
    else if (Cmid == 64) hipLaunchKernelGGL((linear_kernel<64, 32, true, true>), g, t, 0, s, h, L, W2, b2, mean_rstd, y, oc, seg_off, n_seg, mult);
    else if (Cmid == 128) hipLaunchKernelGGL((linear_kernel<128, 32, true, true>), g, t, 0, s, h, L, W2, b2, mean_rstd, y, oc, seg_off, n_seg, mult);
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

extern "C" int roreg_context_colmax(const float *R, int L, const int32_t *seg_off, int n_seg, int max_len, float *ctx_out, float *ws,
                                    void *stream) {
    ROREG_REQUIRE(R && ctx_out && ws && L > 0, "roreg_context_colmax: bad arguments");
    if (!seg_off) { n_seg = 1; max_len = L; }
    ROREG_REQUIRE(n_seg > 0 && max_len > 0, "roreg_context_colmax: bad segment description");
    hipStream_t s = roreg::as_stream(stream);
    const int nblk = max_len < 256 ? max_len : 256;
    float *cmax = ws + (size_t)n_seg * 256 * 60;
    hipLaunchKernelGGL(colmax_partial_kernel, dim3(nblk, n_seg), dim3(256), 0, s, R, L, 60, ws, seg_off);
    hipLaunchKernelGGL(colmax_final_kernel, dim3(60, n_seg), dim3(256), 0, s, ws, nblk, 60, cmax);
    hipLaunchKernelGGL(context_kernel, dim3((L * 120 + 255) / 256), dim3(256), 0, s, R, cmax, L, ctx_out, seg_off, n_seg);
    ROREG_CHECK_LAUNCH("roreg_context_colmax");
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
    case 0:
        if (C == 32) hipLaunchKernelGGL(l2norm_rows32_kernel, dim3((L + 255) / 256), dim3(256), 0, s, a, L, out);
        else hipLaunchKernelGGL(l2norm_rows_kernel, dim3((L + 255) / 256), dim3(256), 0, s, a, L, C, out);
        break;
    case 1: {
        ROREG_REQUIRE(ws && C <= 256, "roreg_rm_elementwise: colmax needs a workspace");
        int nblk = L < 256 ? L : 256;
        hipLaunchKernelGGL(colmax_partial_kernel, dim3(nblk), dim3(256), 0, s, a, L, C, ws, (const int *)nullptr);
        hipLaunchKernelGGL(colmax_final_kernel, dim3(C), dim3(256), 0, s, ws, nblk, C, out);
        break;
    }
    case 2: hipLaunchKernelGGL(context_kernel, dim3((L * 120 + 255) / 256), dim3(256), 0, s, a, b, L, out, (const int *)nullptr, 1); break;
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
    const OtBatch one = {nullptr, nullptr, 0, nullptr};
    {
        const int rpb = 32;
        hipLaunchKernelGGL(ot_build_kernel, dim3((n + 1 + 255) / 256, (m + 1 + rpb - 1) / rpb), dim3(256), 0, s, src_final, m, tgt_final, n,
                           alpha, rpb, Z0, ldz, (const int *)nullptr, (const int *)nullptr, (size_t)0);
        hipLaunchKernelGGL(ot_build_kernel, dim3((m + 1 + 255) / 256, (n + 1 + rpb - 1) / rpb), dim3(256), 0, s, tgt_final, n, src_final, m,
                           alpha, rpb, Z0T, ldt, (const int *)nullptr, (const int *)nullptr, (size_t)0);
    }
    (void)hipMemsetAsync(u, 0, sizeof(float) * up, s);
    (void)hipMemsetAsync(v, 0, sizeof(float) * vp, s);
    const float ln_n = logf((float)n), ln_m = logf((float)m);
    for (int it = 0; it < iters; ++it) {
        hipLaunchKernelGGL(row_lse_kernel, dim3((m + 4) / 4), dim3(256), 0, s, Z0, m + 1, n + 1, ldz, v, normc, ln_n, u, one);
        hipLaunchKernelGGL(row_lse_kernel, dim3((n + 4) / 4), dim3(256), 0, s, Z0T, n + 1, m + 1, ldt, u, normc, ln_m, v, one);
    }
    hipLaunchKernelGGL(ot_final_kernel, dim3(gb), dim3(256), 0, s, Z0, ldz, m, n, u, v, normc, Z_out);
    if (matches0) {
        ROREG_REQUIRE(matches1 && mscores0 && mscores1, "roreg_sinkhorn: all four readout outputs are needed");
        hipLaunchKernelGGL(row_argmax_kernel<true>, dim3(m), dim3(256), 0, s, Z0, m + 1, n + 1, ldz, v, u, normc, i0, val0, one);
        hipLaunchKernelGGL(row_argmax_kernel<false>, dim3(n), dim3(256), 0, s, Z0T, n + 1, m + 1, ldt, u, v, normc, i1, val1, one);
        const int mx = m > n ? m : n;
        hipLaunchKernelGGL(ot_readout_kernel, dim3((mx + 255) / 256), dim3(256), 0, s, i0, val0, m, i1, n, matches0, matches1, mscores0, mscores1,
                           (const int *)nullptr, (const int *)nullptr);
    }
    ROREG_CHECK_LAUNCH("roreg_sinkhorn");
    return 0;
}

extern "C" size_t roreg_sinkhorn_workspace_size(int m, int n) {
    return (size_t)(m + 1) * ((n + 4) & ~3) + (size_t)(n + 1) * ((m + 4) & ~3) + 5 * (size_t)(m + n + 8) + 32;
}

// ---- several pairs per launch --------------------------------------------------------------------------------
// Workspace: per pair a slab [Z0 | Z0^T | u | v] sized for (max_m, max_n); then, concatenated over the pairs, the row/column arg-max
// values (floats) and indices (int64), then the per-pair constants.
static size_t ot_slab(int max_m, int max_n) {
    const size_t ldz = (max_n + 4) & ~3, ldt = (max_m + 4) & ~3;
    return (size_t)(max_m + 1) * ldz + (size_t)(max_n + 1) * ldt + ldt + ldz;
}

static size_t ot_part_stride(int max_m, int max_n) {          // floats of column partials per pair: [row blocks][ldz]
    const size_t ldz = (max_n + 4) & ~3;
    return (size_t)((max_m + 1 + OT_RB - 1) / OT_RB) * ldz;
}

extern "C" size_t roreg_sinkhorn_batch_workspace_size(int n_seg, int max_m, int max_n, long long total_m, long long total_n) {
    return (size_t)n_seg * (ot_slab(max_m, max_n) + ot_part_stride(max_m, max_n)) + 3 * (size_t)(total_m + total_n) + 64;
}

// Host helper: the per-pair constants of the two passes, [n_seg][2] (-log(m+n), log n) then [n_seg][2] (-log(m+n), log m), computed
// with the same host logf calls as roreg_sinkhorn so that the batched passes are bitwise the one-pair ones.
extern "C" int roreg_sinkhorn_batch_consts(const int32_t *seg_src_host, const int32_t *seg_tgt_host, int n_seg, float *consts_host) {
    ROREG_REQUIRE(seg_src_host && seg_tgt_host && consts_host && n_seg > 0, "roreg_sinkhorn_batch_consts: bad arguments");
    for (int p = 0; p < n_seg; ++p) {
        const int m = seg_src_host[p + 1] - seg_src_host[p], n = seg_tgt_host[p + 1] - seg_tgt_host[p];
        ROREG_REQUIRE(m > 0 && n > 0, "roreg_sinkhorn_batch_consts: pair %d is empty", p);
        const float normc = -logf((float)(m + n));
        consts_host[2 * p] = normc; consts_host[2 * p + 1] = logf((float)n);
        consts_host[2 * n_seg + 2 * p] = normc; consts_host[2 * n_seg + 2 * p + 1] = logf((float)m);
    }
    return 0;
}

static size_t ot_flash_floats(int n_seg, int max_m, int max_n) { return (roreg::ot_flash_workspace_bytes(n_seg, max_m, max_n) + 3) / 4 + 8; }

extern "C" size_t roreg_sinkhorn_batch2_workspace_size(int n_seg, int max_m, int max_n, long long total_m, long long total_n) {
    return roreg_sinkhorn_batch_workspace_size(n_seg, max_m, max_n, total_m, total_n) + ot_flash_floats(n_seg, max_m, max_n);
}

// recomputed iterations + read-out from the descriptors (no log-couplings asked for): per pair only the two potential vectors
static bool ot_no_matrix(int recompute, bool want_Z) {
    static const bool rd_mfma = !(getenv("ROREG_OT_READOUT_MFMA") && atoi(getenv("ROREG_OT_READOUT_MFMA")) == 0);
    return recompute && !want_Z && rd_mfma;
}
static size_t ot_base_floats(int n_seg, int max_m, int max_n, long long tm, long long tn, bool no_matrix) {
    if (!no_matrix) return roreg_sinkhorn_batch_workspace_size(n_seg, max_m, max_n, tm, tn);
    const size_t ldz = (max_n + 4) & ~3, ldt = (max_m + 4) & ~3;
    return (size_t)n_seg * (ldt + ldz) + 2 * (size_t)tn + 3 * (size_t)(tm + tn) + 64;       // potentials, column keys, read-out values + indices
}
// v5: the workspace of roreg_sinkhorn_batch3 for THIS mode -- with recompute != 0 and no log-couplings the two (m+1) x (n+1) matrices per pair
// are not needed (0.25 GB per pair at 5000 points): ~4 (m + n) floats per pair + the fragment workspace of the iterations
extern "C" size_t roreg_sinkhorn_batch3_workspace_size(int n_seg, int max_m, int max_n, long long total_m, long long total_n, int recompute, int want_Z) {
    return ot_base_floats(n_seg, max_m, max_n, total_m, total_n, ot_no_matrix(recompute, want_Z != 0)) + (recompute ? ot_flash_floats(n_seg, max_m, max_n) : 0);
}

static int sinkhorn_batch_impl(const float *src_final, const float *tgt_final, const int32_t *seg_src, const int32_t *seg_tgt,
                               const int32_t *seg_src_host, const int32_t *seg_tgt_host, const float *consts, int n_seg, float alpha, int iters,
                               int64_t *matches0, int64_t *matches1, float *mscores0, float *mscores1, float *ws, size_t ws_floats,
                               int recompute, float *Z_out, void *stream) {
    ROREG_REQUIRE(src_final && tgt_final && seg_src && seg_tgt && seg_src_host && seg_tgt_host && consts && n_seg > 0 && iters >= 0 && ws &&
                      matches0 && matches1 && mscores0 && mscores1, "roreg_sinkhorn_batch: bad arguments");
    ROREG_REQUIRE(!Z_out || n_seg == 1, "roreg_sinkhorn_batch3: the log-couplings are returned for a call of ONE pair (got %d)", n_seg);
    int max_m = 0, max_n = 0, min_n = 0x7fffffff;
    for (int p = 0; p < n_seg; ++p) {
        const int m = seg_src_host[p + 1] - seg_src_host[p], n = seg_tgt_host[p + 1] - seg_tgt_host[p];
        ROREG_REQUIRE(m > 0 && n > 0, "roreg_sinkhorn_batch: pair %d is empty", p);
        if (m > max_m) max_m = m;
        if (n > max_n) max_n = n;
        if (n < min_n) min_n = n;
    }
    const long long tm = seg_src_host[n_seg], tn = seg_tgt_host[n_seg];
    const bool no_matrix = ot_no_matrix(recompute, Z_out != nullptr);
    const size_t base_floats = ot_base_floats(n_seg, max_m, max_n, tm, tn, no_matrix);
    ROREG_REQUIRE(ws_floats >= base_floats + (recompute ? ot_flash_floats(n_seg, max_m, max_n) : 0), "roreg_sinkhorn_batch: workspace too small");
    hipStream_t s = roreg::as_stream(stream);
    const int ldz = (max_n + 4) & ~3, ldt = (max_m + 4) & ~3;
    // recomputed iterations and no log-couplings asked for: nothing needs the two matrices -- the read-out comes from the descriptors too, and a
    // pair's slab shrinks to its two potential vectors
    const size_t slab = no_matrix ? (size_t)(ldt + ldz) : ot_slab(max_m, max_n);
    float *Z0 = ws, *Z0T = Z0 + (size_t)(max_m + 1) * ldz, *u = no_matrix ? ws : Z0T + (size_t)(max_n + 1) * ldt, *v = u + ldt;
    const size_t pstride = no_matrix ? (size_t)0 : ot_part_stride(max_m, max_n);
    float *part = ws + (size_t)n_seg * slab;
    unsigned long long *colbest = nullptr;                               // (no_matrix) packed (value, ~row) keys of the columns' best rows, [sum n]
    float *tail = part + (size_t)n_seg * pstride;
    if (no_matrix) {
        colbest = reinterpret_cast<unsigned long long *>((reinterpret_cast<uintptr_t>(part) + 15) & ~(uintptr_t)15);
        tail = reinterpret_cast<float *>(colbest + tn);
    }
    float *val0 = tail, *val1 = val0 + tm;
    const size_t off = ((reinterpret_cast<uintptr_t>(val1 + tn) + 7) & ~(uintptr_t)7) - reinterpret_cast<uintptr_t>(ws);
    int64_t *i0 = reinterpret_cast<int64_t *>(reinterpret_cast<char *>(ws) + off);
    int64_t *i1 = i0 + tm;
    ROREG_REQUIRE(reinterpret_cast<char *>(i1 + tn) <= reinterpret_cast<char *>(ws + base_floats), "roreg_sinkhorn_batch: workspace too small");
    const OtBatch rows = {seg_src, seg_tgt, slab, consts}, cols = {seg_tgt, seg_src, slab, consts + 2 * n_seg};
    const int rpb = 32;
    if (!no_matrix) {
        hipLaunchKernelGGL(ot_build_kernel, dim3((max_n + 1 + 255) / 256, (max_m + 1 + rpb - 1) / rpb, n_seg), dim3(256), 0, s, src_final, 0,
                           tgt_final, 0, alpha, rpb, Z0, ldz, seg_src, seg_tgt, slab);
        hipLaunchKernelGGL(ot_build_kernel, dim3((max_m + 1 + 255) / 256, (max_n + 1 + rpb - 1) / rpb, n_seg), dim3(256), 0, s, tgt_final, 0,
                           src_final, 0, alpha, rpb, Z0T, ldt, seg_tgt, seg_src, slab);
    }
    if (recompute) {
        // the iterations never touch Z0 / Z0T: every pass recomputes the scores on the matrix cores (csrc/ot_flash.hip); the two matrices
        // above only serve the read-out below
        roreg::ProfScope prof(roreg::PROF_SINKHORN, s);
        if (roreg::ot_flash_iterations(src_final, tgt_final, seg_src, seg_tgt, consts, n_seg, max_m, max_n, min_n, recompute == 2, alpha, iters, u, v, slab,
                                       ws + base_floats, s) != 0) return 1;
    } else {
        for (int p = 0; p < n_seg; ++p) (void)hipMemsetAsync(u + p * slab, 0, sizeof(float) * (ldt + ldz), s);       // u and v are adjacent
        const int nv = (ldz + 1023) / 1024;                  // float4 pieces of a row per thread
        const dim3 gp((max_m + 1 + OT_RB - 1) / OT_RB, n_seg), gm((max_n + 1 + 63) / 64, n_seg);
        {
        roreg::ProfScope prof(roreg::PROF_SINKHORN, s);      // (the iterations only: `iters` passes over every pair's coupling matrix)
        for (int it = 0; it < iters; ++it) {
            if (nv > 8) {                                    // rows beyond 8192 columns: the two-matrix passes
                hipLaunchKernelGGL(row_lse_kernel, dim3((max_m + 4) / 4, n_seg), dim3(256), 0, s, Z0, 0, 0, ldz, v, 0.f, 0.f, u, rows);
                hipLaunchKernelGGL(row_lse_kernel, dim3((max_n + 4) / 4, n_seg), dim3(256), 0, s, Z0T, 0, 0, ldt, u, 0.f, 0.f, v, cols);
                continue;
            }
#define OT_PASS(NV) hipLaunchKernelGGL(ot_fused_pass_kernel<NV>, gp, dim3(256), 0, s, Z0, ldz, v, u, part, pstride, rows)
            switch (nv) {
            case 1: OT_PASS(1); break; case 2: OT_PASS(2); break; case 3: OT_PASS(3); break; case 4: OT_PASS(4); break;
            case 5: OT_PASS(5); break; case 6: OT_PASS(6); break; case 7: OT_PASS(7); break; default: OT_PASS(8); break;
            }
#undef OT_PASS
            hipLaunchKernelGGL(ot_col_merge_kernel, gm, dim3(256), 0, s, part, pstride, ldz, v, cols, Z0T, ldt, u);
        }
        }
    }
    if (Z_out) {                                             // Z = ((Z0 + u) + v) - norm (network/rot_coh_match.py:313): the one pair's (m+1) x (n+1) log-couplings
        const size_t tot = (size_t)(max_m + 1) * (max_n + 1);
        hipLaunchKernelGGL(ot_final_batch1_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, Z0, ldz, max_m, max_n, u, v, consts, Z_out);
    }
    if (no_matrix) {
        if (!roreg::mfma_chain_verified(s)) return 4;
        (void)hipMemsetAsync(colbest, 0, (size_t)tn * sizeof(unsigned long long), s);
        hipLaunchKernelGGL(ot_argmax_mfma_kernel, dim3((max_m + 31) / 32, n_seg), dim3(256), 0, s, src_final, tgt_final, seg_src, seg_tgt, u, v, slab, consts,
                           colbest, i0, val0);
        hipLaunchKernelGGL(ot_colbest_decode_kernel, dim3((unsigned)((tn + 255) / 256)), dim3(256), 0, s, colbest, tn, i1);
    } else {
        hipLaunchKernelGGL(row_argmax_kernel<true>, dim3(max_m, n_seg), dim3(256), 0, s, Z0, 0, 0, ldz, v, u, 0.f, i0, val0, rows);
        hipLaunchKernelGGL(row_argmax_kernel<false>, dim3(max_n, n_seg), dim3(256), 0, s, Z0T, 0, 0, ldt, u, v, 0.f, i1, val1, cols);
    }
    const int mx = max_m > max_n ? max_m : max_n;
    hipLaunchKernelGGL(ot_readout_kernel, dim3((mx + 255) / 256, n_seg), dim3(256), 0, s, i0, val0, 0, i1, 0, matches0, matches1, mscores0,
                       mscores1, seg_src, seg_tgt);
    ROREG_CHECK_LAUNCH("roreg_sinkhorn_batch");
    return 0;
}

extern "C" int roreg_sinkhorn_batch(const float *src_final, const float *tgt_final, const int32_t *seg_src, const int32_t *seg_tgt,
                                    const int32_t *seg_src_host, const int32_t *seg_tgt_host, const float *consts, int n_seg, float alpha, int iters,
                                    int64_t *matches0, int64_t *matches1, float *mscores0, float *mscores1, float *ws, size_t ws_floats,
                                    void *stream) {
    return sinkhorn_batch_impl(src_final, tgt_final, seg_src, seg_tgt, seg_src_host, seg_tgt_host, consts, n_seg, alpha, iters, matches0, matches1,
                               mscores0, mscores1, ws, ws_floats, 0, nullptr, stream);
}

extern "C" int roreg_sinkhorn_batch2(const float *src_final, const float *tgt_final, const int32_t *seg_src, const int32_t *seg_tgt,
                                     const int32_t *seg_src_host, const int32_t *seg_tgt_host, const float *consts, int n_seg, float alpha, int iters,
                                     int64_t *matches0, int64_t *matches1, float *mscores0, float *mscores1, float *ws, size_t ws_floats,
                                     int recompute, void *stream) {
    return sinkhorn_batch_impl(src_final, tgt_final, seg_src, seg_tgt, seg_src_host, seg_tgt_host, consts, n_seg, alpha, iters, matches0, matches1,
                               mscores0, mscores1, ws, ws_floats, recompute, nullptr, stream);
}

extern "C" int roreg_sinkhorn_early_exit(int on) { return roreg::ot_flash_early_exit(on); }

extern "C" int roreg_sinkhorn_iteration_stats(long long *iterations, long long *pairs, int reset, void *stream) {
    ROREG_REQUIRE(iterations && pairs, "roreg_sinkhorn_iteration_stats: null output pointer");
    ROREG_REQUIRE(roreg::ot_flash_iteration_stats(iterations, pairs, reset, (hipStream_t)stream) == 0, "roreg_sinkhorn_iteration_stats: device copy failed");
    return 0;
}

extern "C" int roreg_sinkhorn_batch3(const float *src_final, const float *tgt_final, const int32_t *seg_src, const int32_t *seg_tgt,
                                     const int32_t *seg_src_host, const int32_t *seg_tgt_host, const float *consts, int n_seg, float alpha, int iters,
                                     int64_t *matches0, int64_t *matches1, float *mscores0, float *mscores1, float *ws, size_t ws_floats,
                                     int recompute, float *Z_out, void *stream) {
    return sinkhorn_batch_impl(src_final, tgt_final, seg_src, seg_tgt, seg_src_host, seg_tgt_host, consts, n_seg, alpha, iters, matches0, matches1,
                               mscores0, mscores1, ws, ws_floats, recompute, Z_out, stream);
}
