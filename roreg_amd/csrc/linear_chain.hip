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

}  // namespace roreg
