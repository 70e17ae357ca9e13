// The matcher's pointwise 1x1 layers as float32 fmaf chains ON THE MATRIX CORES -- bit for bit the vector-pipe kernels of rm.hip.
//
//   plain : y  = W x + b                                  (mlp_2layer's first conv, attention projections; rot_coh_match.py:14-32,95-119)
//   tail  : y += W relu((x - mean) * rstd) + b             (second conv of mlp_2layer / Contextnorm on top of the residual branch, :21-31,63-81)
//
// Match_ot.forward() is pinned to the reference by evaluating every output as ONE float32 fmaf chain over the input channels in ascending
// order, starting from the bias (linear_kernel / linear_tiled_kernel, rm.hip): the layers feed top-k neighbour selections, where another
// rounding of the same sum can flip a neighbour (the fp16 hi/lo kernels of linear_mfma.hip do, on the reference's 5000-point golden).
// Measured on gfx950 (tools/probe/mfma_f32_order.hip): v_mfma_f32_32x32x2_f32 IS that chain -- D = fma(a_k1, b_k1, fma(a_k0, b_k0, C)),
// each step rounded to float32, k ascending; 1024 of 1024 random outputs bitwise equal to the fmaf loop (and 16x16x4 likewise).  So the
// same arithmetic runs here at the matrix cores' operand reuse: a wavefront owns 32 rows x all outputs, D starts as the bias, CIN / 2
// chained MFMAs per 32-output tile; rows go through LDS (coalesced 16-byte global accesses, conflict-free fragment reads at an odd pitch),
// the weights sit in LDS in fragment order.  A row's result depends on that row alone.  Odd CIN (the 3-channel position inputs) is padded
// with a zero column: fma(0, 0, acc) = acc.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

__device__ __forceinline__ int lc_seg_of(const int *__restrict__ off, int n_seg, int r) {
    int lo = 0, hi = n_seg - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (off[mid] <= r) lo = mid; else hi = mid - 1;
    }
    return lo;
}

constexpr int LC_ROWS = 128;               // rows per workgroup tile: 4 wavefronts x 32

// CAT3: the input row is never materialised -- row r = [x[r] (32) | table[idx[r]] (32) | conf[r / k] (32)], the value MLP's input of the attention
// blocks (rot_coh_match.py:95-119: positional encoding, normalised neighbour feature, the point's own confidence): the staging reads the three
// sources; the chains see the same 96 values in the same order
struct Cat3 { const float *table; const int64_t *idx; const float *conf; int k; };

// LDS: weights [COUT / 32][KP / 2][64 lanes] floats (lane l of step j: W[32 nt + l % 32][2 j + l / 32]) + rows [128][KP + 1] floats
template <int CIN, int COUT, bool NORM, bool ACCUM, bool CAT3 = false>
__global__ __launch_bounds__(256) void linear_chain_kernel(const float *__restrict__ x, int L, const float *__restrict__ W, const float *__restrict__ b,
                                                           const float *__restrict__ mean_rstd, float *__restrict__ y, const int *__restrict__ seg_off,
                                                           int n_seg, int mult, int tiles, Cat3 cat) {
    static_assert(!CAT3 || (CIN == 96 && !NORM), "CAT3: three 32-wide sources");
    static_assert(COUT % 32 == 0, "linear_chain_kernel: outputs in tiles of 32");
    constexpr int KP = (CIN + 1) & ~1, KS = KP / 2, NT = COUT / 32, PITCH = KP + 1;
    extern __shared__ __attribute__((aligned(16))) char lc_smem[];
    float *wf = reinterpret_cast<float *>(lc_smem);                      // [NT][KS][64]
    float *xs = wf + NT * KS * 64;                                        // [LC_ROWS][PITCH]
    __shared__ int s_seg[LC_ROWS];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int f = tid; f < NT * KS * 64; f += 256) {
        const int l = f & 63, j = (f >> 6) % KS, nt = f / (64 * KS);
        const int o = nt * 32 + (l & 31), c = 2 * j + (l >> 5);
        wf[f] = c < CIN ? W[o * CIN + c] : 0.f;
    }
    float bias[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bias[nt] = b[nt * 32 + (lane & 31)];
    for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int p0 = t * LC_ROWS;
        __syncthreads();                                                  // (the previous tile's fragments are read; the first trip: the weights are written)
        if (NORM && seg_off && tid < LC_ROWS) s_seg[tid] = lc_seg_of(seg_off, n_seg, min(p0 + tid, L - 1) / mult);
        if (NORM && seg_off) __syncthreads();
        // ---- the tile's rows, coalesced, normalised if asked, into LDS ----
        if (CIN % 4 == 0) {
            constexpr int C4 = CIN / 4;
            for (int f = tid; f < LC_ROWS * C4; f += 256) {
                const int row = f / C4, c4 = f - row * C4;
                const int pr = min(p0 + row, L - 1);
                float4 v;
                if constexpr (CAT3) {
                    const int c = c4 * 4;
                    const float *src = c < 32 ? x + (size_t)pr * 32 + c : c < 64 ? cat.table + (size_t)cat.idx[pr] * 32 + (c - 32) : cat.conf + (size_t)(pr / cat.k) * 32 + (c - 64);
                    v = *reinterpret_cast<const float4 *>(src);
                } else {
                    v = *reinterpret_cast<const float4 *>(x + (size_t)pr * CIN + c4 * 4);
                }
                if (NORM) {
                    const float *mr = mean_rstd + (seg_off ? (size_t)s_seg[row] * 2 * CIN : 0);
                    const int c = c4 * 4;
                    v.x = fmaxf((v.x - mr[c]) * mr[CIN + c], 0.f); v.y = fmaxf((v.y - mr[c + 1]) * mr[CIN + c + 1], 0.f);
                    v.z = fmaxf((v.z - mr[c + 2]) * mr[CIN + c + 2], 0.f); v.w = fmaxf((v.w - mr[c + 3]) * mr[CIN + c + 3], 0.f);
                }
                float *d = xs + row * PITCH + c4 * 4;
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
        } else {
            for (int f = tid; f < LC_ROWS * KP; f += 256) {
                const int row = f / KP, c = f - row * KP;
                const int pr = min(p0 + row, L - 1);
                float v = c < CIN ? x[(size_t)pr * CIN + c] : 0.f;
                if (NORM && c < CIN) {
                    const float *mr = mean_rstd + (seg_off ? (size_t)s_seg[row] * 2 * CIN : 0);
                    v = fmaxf((v - mr[c]) * mr[CIN + c], 0.f);
                }
                xs[row * PITCH + c] = v;
            }
        }
        __syncthreads();
        // ---- the wavefront's 32 rows x all outputs: D = bias, then CIN / 2 chained MFMAs per output tile (k ascending) ----
        const float *xr = xs + (w * 32 + (lane & 31)) * PITCH + (lane >> 5);             // A fragment of step j: xr[2 j]
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = bias[nt];
#pragma unroll 4
        for (int j = 0; j < KS; ++j) {
            const float a = xr[2 * j];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wf[(nt * KS + j) * 64 + lane], acc[nt], 0, 0, 0);
        }
        // accumulator register r of lane l: row 8 (r / 4) + 4 (l / 32) + r % 4, output l % 32: 128-byte runs per row
        // (ACCUM: the sixteen old values are fetched together -- read inside the store loop each was load, s_waitcnt vmcnt(0), add, store)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float prev[16];
            if (ACCUM) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int p = min(p0 + w * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3), L - 1);
                    prev[r] = y[(size_t)p * COUT + nt * 32 + (lane & 31)];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = p0 + w * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
                if (p < L) y[(size_t)p * COUT + nt * 32 + (lane & 31)] = ACCUM ? prev[r] + acc[nt][r] : acc[nt][r];
            }
        }
    }
}

template <int CIN, int COUT, bool NORM, bool ACCUM, bool CAT3 = false>
void launch_chain(const float *x, int L, const float *W, const float *b, const float *mean_rstd, float *y, const int *seg_off, int n_seg, int mult,
                  hipStream_t s, Cat3 cat = Cat3{nullptr, nullptr, nullptr, 1}) {
    constexpr int KP = (CIN + 1) & ~1;
    constexpr size_t smem = ((size_t)(COUT / 32) * (KP / 2) * 64 + (size_t)LC_ROWS * (KP + 1)) * sizeof(float);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(linear_chain_kernel<CIN, COUT, NORM, ACCUM, CAT3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr = true;
    }
    const int tiles = (L + LC_ROWS - 1) / LC_ROWS;
    const int grid = tiles < 1024 ? tiles : 1024;                       // (a workgroup stages the weights once and walks its tiles)
    hipLaunchKernelGGL((linear_chain_kernel<CIN, COUT, NORM, ACCUM, CAT3>), dim3(grid), dim3(256), smem, s, x, L, W, b, mean_rstd, y, seg_off, n_seg, mult, tiles, cat);
}


// One-time self-check behind every kernel that relies on it: v_mfma_f32_32x32x2_f32 must BE the float32 fmaf chain over k, each
// step rounded (measured on gfx950, tools/probe/mfma_f32_order.hip).  One 32 x 32 tile with K = 32: operands spanning 1e-3 .. 1e3, exact zeros,
// subnormal products, and columns built to cancel (a_k b_k + a_{k+1} b_{k+1} ~ 0: a fused pair or another association shows at once).
__global__ __launch_bounds__(64) void mfma_chain_check_kernel(int *__restrict__ bad) {
    const int lane = threadIdx.x, row = lane & 31, kh = lane >> 5;
    auto gen = [](int i, int k, int side) {                             // deterministic operands
        unsigned h = (unsigned)(i * 73856093u) ^ (unsigned)(k * 19349663u) ^ (unsigned)(side * 83492791u);
        h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
        float v = ((int)(h & 0xffff) - 32768) * (1.0f / 32768.0f);
        const int e = (int)((h >> 16) % 21) - 10;                       // 2^-10 .. 2^10
        v = ldexpf(v, e);
        if ((h >> 24) % 11 == 0) v = 0.f;
        if ((h >> 24) % 13 == 1) v *= 1e-30f;                           // subnormal products
        return v;
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.25f;
    for (int j = 0; j < 16; ++j) {                                      // A[row][k], B[k][col]: lane l carries k = 2 j + l / 32
        const int k = 2 * j + kh;
        float a = gen(row, k, 0), b = gen(row, k, 1);
        if ((k & 1) && (row & 3) == 2) { a = gen(row, k - 1, 0); }      // A[row][k] = A[row][k-1]: with the B below the pair cancels for column == row
        if ((k & 1) && (row & 3) == 2) { b = -gen(row, k - 1, 1); }
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    int wrong = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3), c = lane & 31;     // D[i][c]
        float want = 0.25f;
        for (int k = 0; k < 32; ++k) {
            float a = gen(i, k, 0), b = gen(c, k, 1);
            if ((k & 1) && (i & 3) == 2) a = gen(i, k - 1, 0);
            if ((k & 1) && (c & 3) == 2) b = -gen(c, k - 1, 1);
            want = __fmaf_rn(a, b, want);
        }
        if (__float_as_uint(want) != __float_as_uint(acc[r])) ++wrong;
    }
    if (wrong) atomicAdd(bad, wrong);
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// Round 6: the same chains, software-pipelined, with the two convolutions that read one input in ONE launch.
//
// mlp_2layer (rot_coh_match.py:14-32) applies two 1x1 convolutions to the same input -- the first layer (CIN -> C1, its output h goes through
// InstanceNorm) and the residual branch (CIN -> C2 = 32) -- and then needs h's per-pair channel statistics.  lc2_kernel stages a 128-row tile
// once and runs the chains of both weight sets on it (y1 = W1 x + b1, y2 = W2 x + b2: the same fmaf chains as before, bit for bit); with
// STATS it also leaves the tile's float64 channel sums of y1, so no kernel reads h again for the statistics (in_stats_partial_kernel read
// 61 MB per 5000-point pair and k = 16).  Tiles never straddle two pairs (a pair's statistics then associate the same way whatever is stacked
// beside it): tile t of segment s covers rows [off[s] mult + 128 t, ...).  Per tile: the NEXT tile's rows are requested from global memory
// into registers before the chains start and land under them (the round-5 kernel loaded, waited, staged, synchronised and only then
// multiplied: 16-25 % of the float32 MFMA rate on the 96-wide layers); with ACCUM the sixteen old values per output tile are requested there too.
struct LcSegs { const int *off; int n_seg, mult, L, rows; };      // rows: rows per tile (32 per wavefront of the workgroup)

__device__ __forceinline__ int lc_seg_rows(const LcSegs &sg, int s) { return sg.off ? (sg.off[s + 1] - sg.off[s]) * sg.mult : sg.L; }
__device__ __forceinline__ int lc_total_tiles(const LcSegs &sg) {
    int n = 0;
    for (int s = 0; s < sg.n_seg; ++s) n += (lc_seg_rows(sg, s) + sg.rows - 1) / sg.rows;
    return n;
}
// tile t (>= the tile the cursor stands on) -> (first row, rows, segment, index of its first 128-row statistics unit); the cursor (segment, its first
// tile, its first unit) only moves forward.  Statistics units are ALWAYS 128 rows (LC_ROWS) whatever the tile height, so the channel sums
// associate the same way under 128- and 256-row tiles.
__device__ __forceinline__ void lc_locate(const LcSegs &sg, int t, int &cs, int &cbase, int &ubase, int &row0, int &nrows, int &unit0) {
    for (;;) {
        const int len = lc_seg_rows(sg, cs), nt = (len + sg.rows - 1) / sg.rows;
        if (t < cbase + nt || cs + 1 >= sg.n_seg) {
            const int r = (t - cbase) * sg.rows;
            row0 = (sg.off ? sg.off[cs] * sg.mult : 0) + r;
            nrows = min(sg.rows, len - r);
            unit0 = ubase + r / LC_ROWS;
            return;
        }
        cbase += nt; ubase += (len + LC_ROWS - 1) / LC_ROWS; ++cs;
    }
}

template <int CIN, int C1, int C2, bool NORM, bool ACCUM, bool CAT3, bool STATS, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void lc2_kernel(const float *__restrict__ x, const float *__restrict__ W1, const float *__restrict__ b1, const float *__restrict__ W2,
                                                  const float *__restrict__ b2, const float *__restrict__ mean_rstd, float *__restrict__ y1, float *__restrict__ y2,
                                                  double *__restrict__ part, LcSegs sg, Cat3 cat) {
    static_assert(!CAT3 || (CIN == 96 && !NORM), "CAT3: three 32-wide sources");
    static_assert(C1 % 32 == 0 && C2 % 32 == 0 && (!ACCUM || C2 == 0), "lc2_kernel: outputs in tiles of 32");
    constexpr int KP = (CIN + 1) & ~1, KS = KP / 2, NT1 = C1 / 32, NT = (C1 + C2) / 32, PITCH = KP + 1;
    constexpr bool VEC = CIN % 4 == 0;
    constexpr int ROWS = 32 * WAVES, NTH = 64 * WAVES;               // WAVES = 8: two wavefronts per SIMD share the staged weights and cover each other's LDS latency
    constexpr int C4 = VEC ? CIN / 4 : 1, NV = VEC ? (ROWS * C4 + NTH - 1) / NTH : (ROWS * KP + NTH - 1) / NTH;
    extern __shared__ __attribute__((aligned(16))) char lc_smem[];
    float *wf = reinterpret_cast<float *>(lc_smem);                      // [NT][KS][64]
    float *xs = wf + NT * KS * 64;                                        // [ROWS][PITCH]
    __shared__ double s_st[STATS ? WAVES : 1][STATS ? C1 : 1][2];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int f = tid; f < NT * KS * 64; f += NTH) {
        const int l = f & 63, j = (f >> 6) % KS, nt = f / (64 * KS);
        const int o = nt * 32 + (l & 31), c = 2 * j + (l >> 5);
        wf[f] = c < CIN ? (o < C1 ? W1[o * CIN + c] : W2[(o - C1) * CIN + c]) : 0.f;
    }
    float bias[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bias[nt] = nt < NT1 ? b1[nt * 32 + (lane & 31)] : b2[(nt - NT1) * 32 + (lane & 31)];
    const int total = lc_total_tiles(sg);
    int cs = 0, cbase = 0, ubase = 0;                                     // the tile cursor
    // ---- the rows of a tile, 16 bytes per thread and trip, into registers ----
    float4 pre[VEC ? NV : 1];
    float pre1[VEC ? 1 : NV];
    auto fetch = [&](int row0, int nrows) {
        if (VEC) {
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                const int f = tid + NTH * q;
                const int row = f / C4, c4 = f - row * C4;
                const int pr = row0 + min(row, nrows - 1);               // (rows past the tile's end repeat its last row: never stored)
                if (ROWS * C4 % NTH != 0 && f >= ROWS * C4) { pre[q] = float4{0.f, 0.f, 0.f, 0.f}; continue; }
                if constexpr (CAT3) {
                    const int c = c4 * 4;
                    const float *src = c < 32 ? x + (size_t)pr * 32 + c : c < 64 ? cat.table + (size_t)cat.idx[pr] * 32 + (c - 32) : cat.conf + (size_t)(pr / cat.k) * 32 + (c - 64);
                    pre[q] = *reinterpret_cast<const float4 *>(src);
                } else {
                    pre[q] = *reinterpret_cast<const float4 *>(x + (size_t)pr * CIN + c4 * 4);
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                const int f = tid + NTH * q;
                const int row = f / KP, c = f - row * KP;
                const int pr = row0 + min(row, nrows - 1);
                pre1[q] = (f < ROWS * KP && c < CIN) ? x[(size_t)pr * CIN + c] : 0.f;
            }
        }
    };
    auto stage = [&](int seg) {                                          // registers -> LDS (normalised if asked)
        const float *mr = NORM ? mean_rstd + (size_t)seg * 2 * CIN : nullptr;
        if (VEC) {
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                const int f = tid + NTH * q;
                if (ROWS * C4 % NTH != 0 && f >= ROWS * C4) continue;
                const int row = f / C4, c4 = f - row * C4;
                float4 v = pre[q];
                if (NORM) {
                    const int c = c4 * 4;
                    v.x = fmaxf((v.x - mr[c]) * mr[CIN + c], 0.f); v.y = fmaxf((v.y - mr[c + 1]) * mr[CIN + c + 1], 0.f);
                    v.z = fmaxf((v.z - mr[c + 2]) * mr[CIN + c + 2], 0.f); v.w = fmaxf((v.w - mr[c + 3]) * mr[CIN + c + 3], 0.f);
                }
                float *d = xs + row * PITCH + c4 * 4;
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
        } else {
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                const int f = tid + NTH * q;
                if (f >= ROWS * KP) continue;
                const int row = f / KP, c = f - row * KP;
                float v = pre1[q];
                if (NORM && c < CIN) v = fmaxf((v - mr[c]) * mr[CIN + c], 0.f);
                xs[row * PITCH + c] = v;
            }
        }
    };
    int t = blockIdx.x, row0 = 0, nrows = 0, unit0 = 0;
    if (t < total) { lc_locate(sg, t, cs, cbase, ubase, row0, nrows, unit0); fetch(row0, nrows); }
    for (; t < total; t += gridDim.x) {
        const int seg = cs, p0 = row0, nr = nrows, u0 = unit0;
        __syncthreads();                                                  // (the previous tile's fragments are read; the first trip: the weights are written)
        stage(seg);
        __syncthreads();
        if (t + (int)gridDim.x < total) { lc_locate(sg, t + gridDim.x, cs, cbase, ubase, row0, nrows, unit0); fetch(row0, nrows); }   // in flight under the chains
        float prev[ACCUM ? NT1 : 1][16];
        if (ACCUM) {
#pragma unroll
            for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int p = p0 + min(w * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3), nr - 1);
                    prev[nt][r] = y1[(size_t)p * C1 + nt * 32 + (lane & 31)];
                }
        }
        // ---- the wavefront's 32 rows x all outputs: D = bias, then CIN / 2 chained MFMAs per output tile (k ascending) ----
        const float *xr = xs + (w * 32 + (lane & 31)) * PITCH + (lane >> 5);             // A fragment of step j: xr[2 j]
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = bias[nt];
#pragma unroll 4
        for (int j = 0; j < KS; ++j) {
            const float a = xr[2 * j];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wf[(nt * KS + j) * 64 + lane], acc[nt], 0, 0, 0);
        }
        // accumulator register r of lane l: row 8 (r / 4) + 4 (l / 32) + r % 4, output l % 32: 128-byte runs per row
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float *y = nt < NT1 ? y1 : y2;
            const int CO = nt < NT1 ? C1 : C2, o0 = (nt < NT1 ? nt : nt - NT1) * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = w * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
                if (rr < nr) y[(size_t)(p0 + rr) * CO + o0 + (lane & 31)] = (ACCUM && nt < NT1) ? prev[nt < NT1 ? nt : 0][r] + acc[nt][r] : acc[nt][r];
            }
        }
        if (STATS) {                                                      // the tile's channel sums of y1 (float64): rows of this wave, the lane pair (l, l + 32), the four waves
#pragma unroll
            for (int nt = 0; nt < NT1; ++nt) {
                double s1 = 0.0, s2 = 0.0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = w * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
                    if (rr < nr) { const double v = (double)acc[nt][r]; s1 += v; s2 += v * v; }
                }
                s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
                if (lane < 32) { s_st[w][nt * 32 + lane][0] = s1; s_st[w][nt * 32 + lane][1] = s2; }
            }
            __syncthreads();
            // one partial per 128-row unit: the four wavefronts of the unit in a fixed tree (a 256-row tile leaves two units; its second one only if it has rows)
            for (int e = tid; e < (WAVES / 4) * C1; e += NTH) {
                const int half = e / C1, c = e - half * C1;
                if (half * LC_ROWS >= nr) continue;
                double *po = part + ((size_t)(u0 + half) * C1 + c) * 2;
                po[0] = (s_st[4 * half][c][0] + s_st[4 * half + 1][c][0]) + (s_st[4 * half + 2][c][0] + s_st[4 * half + 3][c][0]);
                po[1] = (s_st[4 * half][c][1] + s_st[4 * half + 1][c][1]) + (s_st[4 * half + 2][c][1] + s_st[4 * half + 3][c][1]);
            }
        }
    }
}

// mean / rstd of a pair from its 128-row units' channel sums: thread q adds units q, q + 256, ... in order, then a fixed tree; biased variance in float64
__global__ __launch_bounds__(256) void lc_stats_final_kernel(const double *__restrict__ part, LcSegs sg, int C, float eps, float *__restrict__ mean_rstd) {
    __shared__ double sa[256], sb[256];
    const int c = blockIdx.x, seg = blockIdx.y;
    int base = 0;
    for (int s = 0; s < seg; ++s) base += (lc_seg_rows(sg, s) + sg.rows - 1) / sg.rows;
    const int Lp = lc_seg_rows(sg, seg), nt = (Lp + LC_ROWS - 1) / LC_ROWS;
    double a = 0, a2 = 0;
    for (int q = threadIdx.x; q < nt; q += 256) { a += part[((size_t)(base + q) * C + c) * 2]; a2 += part[((size_t)(base + q) * C + c) * 2 + 1]; }
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
        mean_rstd[(size_t)seg * 2 * C + c] = (float)mu;
        mean_rstd[(size_t)seg * 2 * C + C + c] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

template <int CIN, int C1, int C2, bool NORM, bool ACCUM, bool CAT3, bool STATS, int WAVES>
void launch_lc2w(const float *x, const float *W1, const float *b1, const float *W2, const float *b2, const float *mean_rstd, float *y1, float *y2, double *part,
                 LcSegs sg, hipStream_t s, Cat3 cat) {
    constexpr int KP = (CIN + 1) & ~1, ROWS = 32 * WAVES;
    constexpr size_t smem = ((size_t)((C1 + C2) / 32) * (KP / 2) * 64 + (size_t)ROWS * (KP + 1)) * sizeof(float);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(lc2_kernel<CIN, C1, C2, NORM, ACCUM, CAT3, STATS, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr = true;
    }
    sg.rows = ROWS;
    const int upper = sg.L / ROWS + sg.n_seg;                             // >= the tile count (the kernel counts them itself)
    const int per_cu = smem > 80 * 1024 ? 1 : smem > 52 * 1024 ? 2 : 3;  // resident workgroups per compute unit by their LDS
    const int grid = upper < 256 * per_cu ? upper : 256 * per_cu;       // (a workgroup stages the weights once and walks its tiles)
    hipLaunchKernelGGL((lc2_kernel<CIN, C1, C2, NORM, ACCUM, CAT3, STATS, WAVES>), dim3(grid), dim3(64 * WAVES), smem, s, x, W1, b1, W2, b2, mean_rstd, y1, y2, part, sg, cat);
}

// rows per tile: 256 (eight wavefronts: two per SIMD) where the weights + a 256-row tile fit the LDS and there are tiles enough to fill the chip, else 128
template <int CIN, int C1, int C2, bool NORM, bool ACCUM, bool CAT3, bool STATS>
int launch_lc2(const float *x, const float *W1, const float *b1, const float *W2, const float *b2, const float *mean_rstd, float *y1, float *y2, double *part,
               const LcSegs &sg, hipStream_t s, Cat3 cat = Cat3{nullptr, nullptr, nullptr, 1}) {
    constexpr int KP = (CIN + 1) & ~1;
    constexpr size_t smem8 = ((size_t)((C1 + C2) / 32) * (KP / 2) * 64 + (size_t)256 * (KP + 1)) * sizeof(float);
    static const int force = getenv("ROREG_LC2_WAVES") ? atoi(getenv("ROREG_LC2_WAVES")) : 0;          // (measurements: 4 or 8)
    if constexpr (smem8 <= 160 * 1024) {
        if (force != 4 && (force == 8 || sg.L >= 256 * 512)) {
            launch_lc2w<CIN, C1, C2, NORM, ACCUM, CAT3, STATS, 8>(x, W1, b1, W2, b2, mean_rstd, y1, y2, part, sg, s, cat);
            return 256;
        }
    }
    launch_lc2w<CIN, C1, C2, NORM, ACCUM, CAT3, STATS, 4>(x, W1, b1, W2, b2, mean_rstd, y1, y2, part, sg, s, cat);
    return 128;
}


}  // namespace

namespace roreg {

// true = shape served, launch issued
bool linear_chain(const float *x, int L, int Cin, const float *W, const float *b, int Cout, float *y, hipStream_t s) {
#define LC(CI, CO) if (Cin == CI && Cout == CO) { launch_chain<CI, CO, false, false>(x, L, W, b, nullptr, y, nullptr, 1, 1, s); return true; }
    LC(32, 32) LC(96, 64) LC(120, 128) LC(64, 64) LC(96, 32) LC(120, 32) LC(64, 32) LC(3, 64) LC(3, 32)
#undef LC
    return false;
}

// y [m * k, Cout] = W [x[r] | table[idx[r]] | conf[r / k]] + b, Cout = 64 | 32
bool linear_chain_cat3(const float *x, const float *table, const int64_t *idx, const float *conf, int m, int k, const float *W, const float *b, int Cout,
                       float *y, hipStream_t s) {
    const Cat3 cat = {table, idx, conf, k};
    if (Cout == 64) { launch_chain<96, 64, false, false, true>(x, m * k, W, b, nullptr, y, nullptr, 1, 1, s, cat); return true; }
    if (Cout == 32) { launch_chain<96, 32, false, false, true>(x, m * k, W, b, nullptr, y, nullptr, 1, 1, s, cat); return true; }
    return false;
}

bool linear_tail_chain(const float *h, int L, int Cmid, const float *mean_rstd, const float *W2, const float *b2, float *y, const int *seg_off, int n_seg,
                       int mult, hipStream_t s) {
    if (Cmid == 64) { launch_chain<64, 32, true, true>(h, L, W2, b2, mean_rstd, y, seg_off, n_seg, mult, s); return true; }
    if (Cmid == 128) { launch_chain<128, 32, true, true>(h, L, W2, b2, mean_rstd, y, seg_off, n_seg, mult, s); return true; }
    return false;
}


// 0 = not checked yet, 1 = the matrix cores evaluate the fmaf chain (checked once per process), -1 = they do not
static int g_chain_ok = 0;
bool mfma_chain_verified(hipStream_t s) {
    if (g_chain_ok == 0) {
        int *bad = nullptr, host = -1;
        if (hipMalloc(&bad, sizeof(int)) != hipSuccess) return false;
        (void)hipMemsetAsync(bad, 0, sizeof(int), s);
        hipLaunchKernelGGL(mfma_chain_check_kernel, dim3(1), dim3(64), 0, s, bad);
        const bool copied = hipMemcpyAsync(&host, bad, sizeof(int), hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
        (void)hipFree(bad);
        g_chain_ok = copied && host == 0 ? 1 : -1;
        if (g_chain_ok < 0)
            set_error("v_mfma_f32_32x32x2_f32 does not evaluate the float32 fmaf chain on this device (%d of 1024 outputs differ): the matcher's pinned "
                      "arithmetic would drift -- run with roreg_linear_path(1) (vector-pipe chains) and ROREG_TOPK_MFMA=0 / ROREG_OT_READOUT_MFMA=0", host);
    }
    return g_chain_ok > 0;
}

// Round 6 entry points: the pipelined kernel (lc2_kernel) behind every shape the matcher uses.
static bool g_lc2 = !(getenv("ROREG_LC2") && atoi(getenv("ROREG_LC2")) == 0);    // ROREG_LC2=0: round 5's kernels (A/B, tests)

bool linear_chain2_on() { return g_lc2; }
void linear_chain2_set(bool on) { g_lc2 = on; }

// mlp_2layer's first convolution and residual branch in one launch (+ the per-pair statistics of h): h [L, C1], y [L, 32], mean_rstd [n_seg][2 C1];
// part: >= (L / 128 + n_seg) * C1 * 2 doubles.  cat != nullptr: the value MLP's assembled rows (Cin = 96).  true = shape served.
bool mlp_head_chain(const float *x, const float *cat3_table, const int64_t *cat3_idx, const float *cat3_conf, int cat3_k, int L, int Cin, const float *W1,
                    const float *b1, int C1, const float *Wr, const float *br, float *h, float *y, const int *seg_off, int n_seg, int mult, float eps,
                    float *mean_rstd, double *part, hipStream_t s) {
    LcSegs sg = {seg_off, seg_off ? n_seg : 1, seg_off ? mult : 1, L, LC_ROWS};
    bool ok = true;
    if (cat3_table) {
        const Cat3 cat = {cat3_table, cat3_idx, cat3_conf, cat3_k};
        if (Cin == 96 && C1 == 64) launch_lc2<96, 64, 32, false, false, true, true>(x, W1, b1, Wr, br, nullptr, h, y, part, sg, s, cat);
        else ok = false;
    }
    else if (Cin == 96 && C1 == 64) launch_lc2<96, 64, 32, false, false, false, true>(x, W1, b1, Wr, br, nullptr, h, y, part, sg, s);
    else if (Cin == 64 && C1 == 64) launch_lc2<64, 64, 32, false, false, false, true>(x, W1, b1, Wr, br, nullptr, h, y, part, sg, s);
    else if (Cin == 120 && C1 == 128) launch_lc2<120, 128, 32, false, false, false, true>(x, W1, b1, Wr, br, nullptr, h, y, part, sg, s);
    else if (Cin == 3 && C1 == 64) launch_lc2<3, 64, 32, false, false, false, true>(x, W1, b1, Wr, br, nullptr, h, y, part, sg, s);
    else ok = false;
    if (!ok) return false;
    hipLaunchKernelGGL(lc_stats_final_kernel, dim3(C1, sg.n_seg), dim3(256), 0, s, part, sg, C1, eps, mean_rstd);
    return true;
}

bool linear_tail_chain2(const float *h, int L, int Cmid, const float *mean_rstd, const float *W2, const float *b2, float *y, const int *seg_off, int n_seg,
                        int mult, hipStream_t s) {
    LcSegs sg = {seg_off, seg_off ? n_seg : 1, seg_off ? mult : 1, L, LC_ROWS};
    if (Cmid == 64) { launch_lc2<64, 32, 0, true, true, false, false>(h, W2, b2, nullptr, nullptr, mean_rstd, y, nullptr, nullptr, sg, s); return true; }
    if (Cmid == 128) { launch_lc2<128, 32, 0, true, true, false, false>(h, W2, b2, nullptr, nullptr, mean_rstd, y, nullptr, nullptr, sg, s); return true; }
    return false;
}

bool linear_chain2(const float *x, int L, int Cin, const float *W, const float *b, int Cout, float *y, hipStream_t s) {
    const LcSegs sg = {nullptr, 1, 1, L, LC_ROWS};
#define LC(CI, CO) if (Cin == CI && Cout == CO) { launch_lc2<CI, CO, 0, false, false, false, false>(x, W, b, nullptr, nullptr, nullptr, y, nullptr, nullptr, sg, s); return true; }
    LC(32, 32) LC(96, 64) LC(120, 128) LC(64, 64) LC(96, 32) LC(120, 32) LC(64, 32) LC(3, 64) LC(3, 32)
#undef LC
    return false;
}

}  // namespace roreg
