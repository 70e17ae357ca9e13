// Sinkhorn iterations that never read a coupling matrix: every pass recomputes the scores on the matrix cores.
//
// The log-domain Sinkhorn of the rotation-coherence matcher (network/rot_coh_match.py:285-314) alternates
//     u = log_mu - LSE_j(Z + v),      v = log_nu - LSE_i(Z + u)
// 100 times over the (m+1) x (n+1) coupling matrix Z = [[<s_i, t_j>, alpha], [alpha, alpha]] (alpha = the dustbin score).  Materialised,
// that is 25 MB per pair at keynum 2500, read once per iteration by the fused pass of rm.hip: 2.5 GB per pair, HBM-bound at ~5.3 TB/s
// (0.48 ms per pair; the 256 MB Infinity Cache reads no faster than HBM: profiles/r04_mall_rate.txt).  But Z has rank 32 + 2: here a
// wavefront owns 32 rows (their descriptors sit in registers as MFMA fragments), streams the other cloud's descriptors -- 395 KB per
// pair, L2-resident -- in 32-column tiles, and gets
//     acc[i][j] = log2(e) (Z[i][j] + u[i] + v[j])
// straight out of the matrix cores: the descriptors as fp16 hi + lo pieces (three cross products per 16-wide k step, f32 accumulate:
// the f32-accurate product of the GEMM kernels), and ONE more MFMA whose k slots carry the potentials (three fp16 pieces each, against
// constant ones), the dustbin row and column (alpha against indicator slots) and the padding (-60000 in a potential slot: its exponential
// is exactly 0).  The vector pipe then only exponentiates and adds: S_i = sum_j 2^acc_ij, and
//     u_new[i] = u[i] + log_mu[i] - log S_i
// -- the same update written with the CURRENT potentials as the stabiliser instead of the row maximum: after a column update every
// column of exp(Z + u + v) sums to nu_j <= 1, so no term overflows, and a row's terms cannot all underflow unless its potential moves by
// e^87 in one iteration (then the update kernel evaluates that row exactly in the log domain from the float32 descriptors).  The very
// first row pass has no previous normalisation to lean on: its stabiliser is minus the row's maximum score, from one extra max-only pass.
// The column update is the same kernel with the two sides exchanged.  Partial sums over column chunks simply add (a common stabiliser),
// so the grid is (row tiles / 4, chunks, pairs) and fills the chip whatever the pair count.
//
// Per element and iteration: 2 x (7/1024 MFMA + v_exp_f32 + v_add_f32) against one 4-byte HBM read + ~10 vector instructions before.
// The read-out (arg-max of Z + u + v over rows and columns) still runs on a matrix built once per pair (rm.hip).
//
// Two forms of the iteration.  of_pass_kernel + of_update_kernel, twice per iteration (rows, then columns): any size.  of_iter_kernel /
// of_iter_coop_kernel: ONE recomputation per iteration, a 32-row strip's exponentials stay in registers between its row sums and its
// share of the column sums -- one workgroup per strip up to 2559 target points (yoho_mat's default keynum 2500), two cooperating
// workgroups up to 5119 (`Test.py --keynum 5000`, test/evaluator.py:20,46); details at the kernels.  Which of the two a pair takes depends
// on its own target length, and every sum associates independently of what is stacked beside a pair; only a group with a target cloud
// beyond 5119 points takes the two-pass form as a whole.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int OF_F = 32;                  // descriptor width
constexpr int OF_PLANES = 5;              // fragment planes of a 32-row tile: hi k0-15, hi k16-31, extras, lo k0-15, lo k16-31
constexpr int OF_TILE_HALFS = OF_PLANES * 64 * 8;
constexpr float OF_PAD = -60000.0f;       // potential of a padding row / column (fp16-representable; 2^-60000 = 0)
constexpr float OF_LO_SCALE = 4096.0f;    // third pieces travel as x 2^12 against a 2^-12 slot: never an fp16 subnormal
constexpr int OF_CHT = 16;                // column tiles per chunk (512 columns): the unit of a row's partial sums
constexpr float LOG2E = 1.44269504088896340736f, LN2 = 0.69314718055994530942f;

struct Side {                             // one side (source rows or target columns) of one pass
    const float *desc;                    // [total, 32] float32 descriptors, all pairs
    const int *seg;                       // [n_seg + 1] offsets
    _Float16 *frag;                       // fragments, per pair `frag_stride` halfs
    float *pot;                           // potentials in log2 units, per pair `pot_stride`
    float *pot2;                          // the fused iteration's second buffer of ROW potentials (it reads one, writes the other: see of_iter_body)
    float *part;                          // partial sums [NCH][pot_stride] per pair
    size_t frag_stride, pot_stride;
    const float *consts;                  // per pair (normc, log of the OTHER side's length): natural logs (roreg_sinkhorn_batch_consts)
};

// Per-pair convergence record of the iterations (ot_flash_iterations): move[pair][t] = the largest step any of the pair's m + n + 2 potentials
// took in iteration t, in units of OF_TOL (max(2^-22 |u|, 2^-20) in log2 units: 2 .. 4 units in the last place of the float32 potential), as
// float bits; 0 for an iteration that did not run (the array is cleared up front).  Iteration t >= 1 is SKIPPED for a pair when
//   (a) move[t-1] <= 1: nothing moved beyond float32 resolution, or
//   (b) move[t-1] <= OF_PLATEAU_MAX and move[t-1] >= move[t-2]: the steps are small and this one is NOT SMALLER than the last -- the
//       iteration has reached the noise floor of its own arithmetic (the scores come out of fp16 hi/lo MFMAs accumulated in float32 at
//       magnitudes |Z| + |u| + |v| ~ 40: a step of one ulp in u re-rounds them, and the potentials then jitter by 3 .. 4 ulps for ever).
// Either way the pair sits at the fixed point of the float32 iteration; the reference's loop (network/rot_coh_match.py:289-292 always runs
// `iters` = 100 of them) only moves last bits from there on.  A sequence that still converges -- however slowly: a slow mode shrinks its steps
// monotonically -- never satisfies (b); at the noise floor the steps fluctuate (equal values are common: they are a few ulps), so (b) fires
// within an iteration or two of reaching it.  Skipping is sticky (a skipped iteration records 0, which is (a) for the next one) and a pair's record depends on its own data only.
// hist == nullptr: everything runs and nothing is recorded (early exit off).
struct Conv {
    unsigned *hist;                            // move[pair 0][0]; pair p's record at hist + p * stride
    int t, stride;
    float tol_rel, tol_abs;
};
constexpr float OF_TOL_REL = 0x1p-22f, OF_TOL_ABS = 0x1p-20f, OF_PLATEAU_MAX = 8.0f, OF_PLATEAU_RATIO = 1.0f;
__device__ __forceinline__ bool conv_stop(const unsigned *rec, int t) {
    if (t < 1) return false;
    const float r1 = __uint_as_float(rec[t - 1]);
    if (r1 <= 1.0f) return true;
    return t >= 2 && r1 <= OF_PLATEAU_MAX && r1 >= OF_PLATEAU_RATIO * __uint_as_float(rec[t - 2]);
}
__device__ __forceinline__ bool conv_done(const Conv &c, int pair) { return c.hist != nullptr && conv_stop(c.hist + (size_t)pair * c.stride, c.t); }
// every lane of the wave calls this (np = old for lanes without a potential); one atomic per wave
__device__ __forceinline__ void conv_note(const Conv &c, int pair, float np, float old) {
    if (c.hist == nullptr) return;
    float r = fabsf(np - old) / fmaxf(fabsf(np) * c.tol_rel, c.tol_abs);
    if (!(r >= 0.f)) r = 0.f;                                         // (non-finite potentials do not steer the record)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) r = fmaxf(r, __shfl_xor(r, o));
    if ((threadIdx.x & 63) == 0 && r > 0.f) atomicMax(c.hist + (size_t)pair * c.stride + c.t, __float_as_uint(r));
}
// iterations a pair has run: the first t >= 1 that was skipped, else all of them
__device__ __forceinline__ int conv_executed(const unsigned *hist, int stride, int pair, int iters) {
    if (hist == nullptr) return iters;
    int t = iters > 0 ? 1 : 0;
    while (t < iters && !conv_stop(hist + (size_t)pair * stride, t)) ++t;
    return t;
}

__device__ __forceinline__ void split3(float x, _Float16 &h, _Float16 &m, _Float16 &l) {
    h = (_Float16)x;
    const float r = x - (float)h;
    m = (_Float16)r;
    l = (_Float16)((r - (float)m) * OF_LO_SCALE);
}

__global__ __launch_bounds__(256) void of_absmax_kernel(Side a, Side b, unsigned *__restrict__ amax) {
    const int pair = blockIdx.y >> 1, side = blockIdx.y & 1;
    const Side &s = side ? b : a;
    const int r0 = s.seg[pair], len = s.seg[pair + 1] - r0;
    const float4 *src = reinterpret_cast<const float4 *>(s.desc + (size_t)r0 * OF_F);       // rows are 128 bytes: 16-byte aligned
    float v = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < len * (OF_F / 4); i += gridDim.x * 256) {
        const float4 x = src[i];
        const float m4 = fmaxf(fmaxf(fabsf(x.x), fabsf(x.y)), fmaxf(fabsf(x.z), fabsf(x.w)));
        if (m4 < __builtin_inff()) v = fmaxf(v, m4);             // non-finite descriptors do not steer the scale (their scores are NaN anyway)
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    if ((threadIdx.x & 63) == 0 && v > 0.f) atomicMax(&amax[pair * 2 + side], __float_as_uint(v));
}

// Power-of-two balance of the two sides' magnitudes (exact, the product is unchanged): both maxima end up near the geometric mean, far
// inside fp16's range unless the scores themselves are beyond 2^28.
__device__ __forceinline__ int balance_exp(const unsigned *amax, int pair) {
    const float as = __uint_as_float(amax[pair * 2]) * LOG2E, at = __uint_as_float(amax[pair * 2 + 1]);
    if (!(as > 0.f) || !(at > 0.f)) return 0;
    int es, et;
    (void)frexpf(as, &es); (void)frexpf(at, &et);
    return (et - es) / 2;
}

// Fragment order of v_mfma_f32_32x32x16_f16 (A and B alike): lane l carries row (l % 32), k = 8 (l / 32) + e.  One workgroup per
// (tile, side, pair); thread (plane, lane) writes its 8 halfs.
__global__ __launch_bounds__(320) void of_prep_kernel(Side a, Side b, const unsigned *__restrict__ amax, float alpha) {
    const int pair = blockIdx.z, side = blockIdx.y, t = blockIdx.x;
    const Side &s = side ? b : a;
    const int r0 = s.seg[pair], len = s.seg[pair + 1] - r0;            // rows 0..len-1 are points, row len is the dustbin
    if (t > len / 32 + 1) return;                                     // (tile len / 32 + 1 is the PAD tile: all rows invalid; of_iter_kernel reads it for tiles a pair does not have)
    const int p = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int i = t * 32 + (l & 31), kg = l >> 5;
    const int e0 = balance_exp(amax, pair);
    const float c = side ? ldexpf(1.0f, -e0) : ldexpf(LOG2E, e0);
    f16x8 out;
    if (p != 2) {
        const int ks = p < 2 ? p : p - 3;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float x = 0.f;
            if (i < len) x = s.desc[(size_t)(r0 + i) * OF_F + ks * 16 + kg * 8 + e] * c;
            const _Float16 h = (_Float16)x;
            out[e] = p < 2 ? h : (_Float16)(x - (float)h);
        }
    } else {
        _Float16 ah, am, al;
        split3(alpha * LOG2E, ah, am, al);
        const _Float16 one = (_Float16)1.0f, tiny = (_Float16)(1.0f / OF_LO_SCALE), zero = (_Float16)0.0f;
        const bool valid = i <= len, dust = i == len;
        _Float16 sl[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) sl[q] = zero;
        if (side == 0) {          // rows: own potential pieces | constants against v's pieces | "any valid row" x dustbin column | "dustbin row" x point columns
            sl[0] = valid ? zero : (_Float16)OF_PAD;
            sl[3] = one; sl[4] = one; sl[5] = tiny;
            if (valid) { sl[6] = one; sl[7] = one; sl[8] = tiny; }
            if (dust) { sl[9] = one; sl[10] = one; sl[11] = tiny; }
        } else {                  // columns: constants against u's pieces | own potential pieces | alpha on the dustbin column | alpha on the point columns
            sl[0] = one; sl[1] = one; sl[2] = tiny;
            sl[3] = valid ? zero : (_Float16)OF_PAD;
            if (dust) { sl[6] = ah; sl[7] = am; sl[8] = al; }
            if (i < len) { sl[9] = ah; sl[10] = am; sl[11] = al; }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) out[e] = sl[kg * 8 + e];
    }
    *reinterpret_cast<f16x8 *>(s.frag + pair * s.frag_stride + ((size_t)(t * OF_PLANES + p) * 64 + l) * 8) = out;
    if (p == 0 && l < 32 && t <= len / 32) s.pot[pair * s.pot_stride + i] = 0.f;
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, x), __builtin_bit_cast(int, x), CTRL, ROW_MASK, 0xf, false));
}
// reductions over the 32 lanes of each half of the wave (the 32 columns of an accumulator row); the result is valid in lanes 16..31 (first
// half) and 48..63 (second half)
template <int CTRL>
__device__ __forceinline__ float dpp_all(float x) {     // every lane has a source under these controls: old = 0 + bound_ctrl lets the compiler fold the move into the add (v_add_f32_dpp)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float half_sum(float x) {
    x += dpp_all<0xB1>(x);                      // quad_perm [1,0,3,2]
    x += dpp_all<0x4E>(x);                      // quad_perm [2,3,0,1]
    x += dpp_all<0x141>(x);                     // row_half_mirror
    x += dpp_all<0x140>(x);                     // row_mirror: every lane of a 16-lane row holds the row's sum
    const float t = dpp_mov<0x142, 0xa>(x);     // row_bcast:15 -> rows 1 and 3 receive the previous row's sum
    return ((threadIdx.x >> 4) & 1) ? x + t : x;
}
__device__ __forceinline__ float half_max(float x) {
    x = fmaxf(x, dpp_mov<0xB1, 0xf>(x));
    x = fmaxf(x, dpp_mov<0x4E, 0xf>(x));
    x = fmaxf(x, dpp_mov<0x141, 0xf>(x));
    x = fmaxf(x, dpp_mov<0x140, 0xf>(x));
    return fmaxf(x, dpp_mov<0x142, 0xa>(x));    // (rows 0 and 2 see their own value again: harmless for a maximum)
}

// One pass: for the R x 32 rows of side A that this wave owns, part[chunk][row] = sum (MAXP: max) over the chunk's columns of 2^acc (acc).
// Wave w of a workgroup owns row tiles R (4 blockIdx.x + w) .. + R - 1; the four waves stream the same column tiles (their fragment loads
// meet in L1).  R = 2: every column fragment fetched serves two accumulator tiles, whose MFMA chains alternate.
// Software pipeline (round 4; measured by ablation on 30 stacked 2500 x 2500 pairs: of 58 us per pass the 14 MFMAs of a trip accounted for
// 34, the exponentials 11, the fragment loads 10 -- all in series, because a wavefront issues in order and the loop was "14 MFMAs, then 32
// exponentials"): the MFMAs of column tile t + 1 are issued INTERLEAVED with the exponentials of tile t (two accumulator sets, ping-pong
// over two tiles per trip so that nothing is copied; sched_group_barrier prescribes one MFMA, three transcendentals, two vector adds, ...),
// so the vector pipe works in the matrix pipe's shadow; the fragments of tile t + 2 are in flight meanwhile.  The file is compiled with
// -amdgpu-mfma-vgpr-form: accumulators in VGPRs, no v_accvgpr_read per element.
template <bool MAXP, int R, int VAR = 0>     // VAR: ablations for measurements only (ROREG_OT_VARIANT): 1 = no exponentials, 2 = no MFMAs, 3 = no column loads in the loop
__global__ __launch_bounds__(256) void of_pass_kernel(Side a, Side b, int nch, const int *__restrict__ tseg, int skip_le, Conv cv) {
    const int pair = blockIdx.z, chunk = blockIdx.y;
    if ((tseg[pair + 1] - tseg[pair]) / 32 + 1 <= skip_le) return;          // (a pair whose TARGET cloud fits the whole-iteration kernel is left to it)
    if (conv_done(cv, pair)) return;                                        // (the pair's potentials no longer move)
    const int lane = threadIdx.x & 63;
    const int lenA = a.seg[pair + 1] - a.seg[pair], lenB = b.seg[pair + 1] - b.seg[pair];
    const int tilesA = lenA / 32 + 1, tilesB = lenB / 32 + 1;              // (len + 1 rows: the dustbin)
    const int tA0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
    if (tA0 >= tilesA) return;
    const int tb0 = chunk * OF_CHT, tb1 = min(tb0 + OF_CHT, tilesB);          // chunks of a FIXED number of column tiles: a pair's sums associate the same way whatever is stacked beside it
    const f16x8 *fb = reinterpret_cast<const f16x8 *>(b.frag + pair * b.frag_stride) + lane;
    f16x8 A[R][OF_PLANES];
#pragma unroll
    for (int q = 0; q < R; ++q) {
        const int tA = min(tA0 + q, tilesA - 1);                           // (an odd tile count: the last wave's second tile repeats its first, not stored)
        const f16x8 *fa = reinterpret_cast<const f16x8 *>(a.frag + pair * a.frag_stride) + (size_t)tA * OF_PLANES * 64 + lane;
#pragma unroll
        for (int p = 0; p < OF_PLANES; ++p) A[q][p] = fa[p * 64];
    }
    float red[R][16];
#pragma unroll
    for (int q = 0; q < R; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[q][r] = MAXP ? -__builtin_inff() : 0.f;
    auto load_b = [&](int tb, f16x8 (&dst)[OF_PLANES]) {
        if (VAR == 3) return;
        const f16x8 *src = fb + (size_t)min(tb, tb1 - 1) * OF_PLANES * 64;   // (past the chunk: a valid tile, never used)
#pragma unroll
        for (int p = 0; p < OF_PLANES; ++p) dst[p] = src[p * 64];
    };
    // the 7 R MFMAs of one column tile: small terms first; the R accumulators' chains alternate (consecutive MFMAs are independent)
    auto mm = [&](const f16x8 (&B)[OF_PLANES], f32x16 (&acc)[R]) {
#pragma unroll
        for (int q = 0; q < R; ++q) acc[q] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#define OF_STEP(PA, PB) _Pragma("unroll") for (int q = 0; q < R; ++q) { if (VAR != 2) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[q][PA], B[PB], acc[q], 0, 0, 0); else acc[q][PA] += (float)B[PB][q]; }
        OF_STEP(3, 0); OF_STEP(4, 1);          // lo . hi
        OF_STEP(0, 3); OF_STEP(1, 4);          // hi . lo
        OF_STEP(0, 0); OF_STEP(1, 1);          // hi . hi
        OF_STEP(2, 2);                         // potentials, dustbins, padding
#undef OF_STEP
    };
    auto fold = [&](const f32x16 (&acc)[R]) {
#pragma unroll
        for (int q = 0; q < R; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (MAXP) red[q][r] = fmaxf(red[q][r], acc[q][r]);
                else if (VAR == 1) red[q][r] += acc[q][r];
                else red[q][r] += __builtin_amdgcn_exp2f(acc[q][r]);
            }
    };
    auto interleave = [&]() {                  // one MFMA, then a few transcendentals and vector adds in its shadow, 7 R times
#pragma unroll
        for (int g = 0; g < 7 * R; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x400, MAXP ? 0 : (16 * R + 7 * R - 1) / (7 * R), 0);
            __builtin_amdgcn_sched_group_barrier(0x002, (16 * R + 7 * R - 1) / (7 * R), 0);
        }
    };
    f16x8 B0[OF_PLANES], B1[OF_PLANES];
    f32x16 accA[R], accB[R];
    if (tb0 < tb1) {
        if (VAR == 3) {
#pragma unroll
            for (int p = 0; p < OF_PLANES; ++p) { B0[p] = A[0][p]; B1[p] = A[R - 1][(p + 1) % OF_PLANES]; }
        }
        load_b(tb0, B0);
        load_b(tb0 + 1, B1);
        mm(B0, accA);
        int tb = tb0;
        for (; tb + 2 < tb1; tb += 2) {        // entering a trip: accA = tile tb (issued), B1 = fragments of tile tb + 1
            load_b(tb + 2, B0);
            mm(B1, accB); fold(accA);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
            load_b(tb + 3, B1);
            mm(B0, accA); fold(accB);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
        }
        if (tb + 1 < tb1) {
            mm(B1, accB); fold(accA);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
            fold(accB);
        } else {
            fold(accA);
        }
    }
    // accumulator register r of lane l is row 8 (r / 4) + 4 (l / 32) + r % 4, column l % 32
#pragma unroll
    for (int q = 0; q < R; ++q) {
        if (tA0 + q >= tilesA) break;
        float *out = a.part + pair * (a.pot_stride * nch) + (size_t)chunk * a.pot_stride + (tA0 + q) * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = MAXP ? half_max(red[q][r]) : half_sum(red[q][r]);
            if ((lane & 31) == 31) out[8 * (r >> 2) + 4 * (lane >> 5) + (r & 3)] = v;
        }
    }
}

// Exact log2-sum-exp2 of (Z' + potB) over the other side for row i of side A, from the float32 descriptors: lmu minus it is the row's new
// potential.  Only rows whose sum of exponentials left (1e-35, 1e35) come here (every term underflowed, or non-finite input).
__device__ float exact_potential(const Side &a, const Side &b, int pair, int i, float alpha, float lmu) {
    const int ra = a.seg[pair], lenA = a.seg[pair + 1] - ra;
    const int rb = b.seg[pair], lenB = b.seg[pair + 1] - rb;
    const float *potB = b.pot + pair * b.pot_stride;
    const float al = alpha * LOG2E;
    float d[OF_F];
#pragma unroll
    for (int f = 0; f < OF_F; ++f) d[f] = i < lenA ? a.desc[(size_t)(ra + i) * OF_F + f] * LOG2E : 0.f;
    float mx = -__builtin_inff(), np = 0.f;
    for (int rep = 0; rep < 2; ++rep) {
        float sum = 0.f;
        for (int j = 0; j <= lenB; ++j) {
            float x = al;
            if (i < lenA && j < lenB) {
                x = 0.f;
                const float *t = b.desc + (size_t)(rb + j) * OF_F;
#pragma unroll
                for (int f = 0; f < OF_F; ++f) x = fmaf(d[f], t[f], x);
            }
            x += potB[j];
            if (rep == 0) mx = fmaxf(mx, x);
            else sum += __builtin_amdgcn_exp2f(x - mx);
        }
        if (rep == 1) np = lmu - (mx + __log2f(sum));
    }
    return np;
}

__device__ __forceinline__ void store_pieces(const Side &a, int pair, int i, float np, int slot0) {
    _Float16 h, m, l;
    split3(np, h, m, l);
    _Float16 *dst = a.frag + pair * a.frag_stride + ((size_t)((i >> 5) * OF_PLANES + 2) * 64 + (i & 31)) * 8 + slot0;
    dst[0] = h; dst[1] = m; dst[2] = l;
}

// pot[i] <- -max (MAXP: the first row pass's stabiliser)  or  pot[i] + log2(mu_i) - log2(sum of the partial sums); the three fp16
// pieces of the new potential go into the extras plane of side A's fragments.  A sum outside (1e-35, 1e35) is replaced by the exact
// log-domain evaluation of that row.  `cht` = tiles of the OTHER side per partial sum (OF_CHT for of_pass_kernel's chunks, 1 for the row
// strips of of_iter_kernel), `nch` = partial sums allocated per row.
template <bool MAXP>
__global__ __launch_bounds__(256) void of_update_kernel(Side a, Side b, int nch, int cht, int slot0, float alpha, const int *__restrict__ tseg, int skip_le, Conv cv) {
    const int pair = blockIdx.y;
    if ((tseg[pair + 1] - tseg[pair]) / 32 + 1 <= skip_le) return;
    if (conv_done(cv, pair)) return;
    const int lenA = a.seg[pair + 1] - a.seg[pair];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x * 256 > lenA) return;
    const bool live = i <= lenA;                                      // (lanes beyond the dustbin stay for conv_note's wave-wide reduction)
    const int il = live ? i : lenA;
    const float *part = a.part + pair * (a.pot_stride * nch) + il;
    const int nch_pair = ((b.seg[pair + 1] - b.seg[pair]) / 32 + 1 + cht - 1) / cht;      // the partial sums this pair's other side really has
    float *pot = a.pot + pair * a.pot_stride;
    float np, old = 0.f;
    if (MAXP) {
        float mx = -__builtin_inff();
        for (int c = 0; c < nch_pair; ++c) mx = fmaxf(mx, part[(size_t)c * a.pot_stride]);
        np = mx > -__builtin_inff() && mx < __builtin_inff() ? -mx : 0.f;
    } else {
        float S = 0.f;                                            // partial sums added in order; 40 independent loads in flight at a time (the launch is
        for (int c0 = 0; c0 < nch_pair; c0 += 40) {                //  one small workgroup per CU: latency-bound; of_iter_kernel leaves up to 80 per column)
            float v[40];
#pragma unroll
            for (int q = 0; q < 40; ++q) v[q] = c0 + q < nch_pair ? __builtin_nontemporal_load(part + (size_t)(c0 + q) * a.pot_stride) : 0.f;
#pragma unroll
            for (int q = 0; q < 40; ++q) if (c0 + q < nch_pair) S += v[q];
        }
        const float lmu = (il == lenA ? a.consts[pair * 2] + a.consts[pair * 2 + 1] : a.consts[pair * 2]) * LOG2E;
        old = pot[il];
        if (S > 1e-35f && S < 1e35f) np = old + (lmu - __log2f(S));
        else np = exact_potential(a, b, pair, il, alpha, lmu);
        conv_note(cv, pair, live ? np : 0.f, live ? old : 0.f);
    }
    if (!live) return;
    pot[i] = np;
    store_pieces(a, pair, i, np, slot0);
}

// The column update behind of_iter_kernel: one strip sum per 32 source rows and column (79 at keynum 2500, 157 at 5000).  of_update_kernel
// (one thread per column) spends its time waiting for them -- 24 MB per 30 pairs just written by other XCDs, ~300 small workgroups on the
// chip -- so here FOUR waves share a block of 64 columns, wave q adds the q-th quarter of the strips (in order, 20 loads in flight at a
// time), and wave 0 adds the four quarter sums ((q0 + q1) + (q2 + q3)): four times the loads in flight, the association a function of
// the pair's own strip count alone.
__global__ __launch_bounds__(256) void of_update_cols_kernel(Side a, Side b, int nparts, float alpha, int fused_le, Conv cv) {
    const int pair = blockIdx.y;
    const int lenA = a.seg[pair + 1] - a.seg[pair];
    if (lenA / 32 + 1 > fused_le) return;                              // (a = the TARGET side here: pairs beyond the whole-iteration kernels take the two-pass form)
    if (conv_done(cv, pair)) return;
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    if (blockIdx.x * 64 > lenA) return;
    __shared__ float s_q[4][64];
    const int n = (b.seg[pair + 1] - b.seg[pair]) / 32 + 1;            // the strips this pair really has
    const int per = (n + 3) / 4, c0 = q * per, c1 = min(c0 + per, n);
    float S = 0.f;
    if (i <= lenA) {
        const float *part = a.part + pair * (a.pot_stride * nparts) + i;
        for (int cb = c0; cb < c1; cb += 20) {                        // (wave-uniform trip count; one trip up to 80 strips)
            float v[20];
#pragma unroll
            for (int k = 0; k < 20; ++k) v[k] = cb + k < c1 ? __builtin_nontemporal_load(part + (size_t)(cb + k) * a.pot_stride) : 0.f;
#pragma unroll
            for (int k = 0; k < 20; ++k) if (cb + k < c1) S += v[k];
        }
    }
    s_q[q][lane] = S;
    __syncthreads();
    if (q != 0) return;
    const bool live = i <= lenA;                                       // (wave 0 stays whole for conv_note's reduction)
    const int il = live ? i : lenA;
    S = (s_q[0][lane] + s_q[1][lane]) + (s_q[2][lane] + s_q[3][lane]);
    float *pot = a.pot + pair * a.pot_stride;
    const float lmu = (il == lenA ? a.consts[pair * 2] + a.consts[pair * 2 + 1] : a.consts[pair * 2]) * LOG2E;
    const float old = pot[il];
    float np;
    if (!live) np = old;
    else if (S > 1e-35f && S < 1e35f) np = old + (lmu - __log2f(S));
    else np = exact_potential(a, b, pair, il, alpha, lmu);
    conv_note(cv, pair, np, old);
    if (!live) return;
    pot[i] = np;
    store_pieces(a, pair, i, np, 3);
}

// ---------------------------------------------------------------------------------------------------------------
// One WHOLE iteration per launch (+ the column update): the exponentials of a 32-row strip stay in registers between the row sums and
// the column sums, so the scores are recomputed once per iteration instead of twice.
//     E_ij = 2^(Z'_ij + u_i + v_j)         (7 MFMAs per 32 x 32 tile + one v_exp_f32 per element, as in of_pass_kernel)
//     S_i = sum_j E_ij,   u_i += log2 mu_i - log2 S_i,   f_i = 2^(u_new - u_old)
//     C_j(strip) = sum_{i in strip} E_ij f_i            (= the strip's part of sum_i 2^(Z' + u_new + v): one fma per element)
// A workgroup of eight wavefronts owns the strip (one row tile of side A) and OF_HALF = 80 column tiles; wave w owns the column tiles
// w, w + 8, ..., NT <= 10 of them (NT from the group's longest target cloud; tiles beyond a pair's own are the PAD tile of_prep_kernel
// writes behind the pair's last one: padding potentials, exponentials exactly 0, so the code has no branches and every sum the same
// association whatever NT is): 160 accumulator registers per lane hold the wave's part of E (2 waves per SIMD, <= 256 VGPRs) -- a compute
// unit's register file holds 32 rows x 2560 columns and no more.  The strip's column sums go to part[strip][column] of side B and
// of_update_cols_kernel adds the strips in a fixed order: an association independent of what is stacked beside the pair.  Rows beyond the
// dustbin have E = 0 and are not updated.  A row whose sum left (1e-35, 1e35) -- every term underflowed, or non-finite input -- makes the
// workgroup redo its strip with the row maxima as stabilisers (mode 1 = max_j acc, mode 2 = E = 2^(acc - M_i), f_i = mu_i / S_i).
//
// Target clouds of 2560 .. 5119 points (keynum 5000: 157 tiles): TWO workgroups per strip (of_iter_kernel<NT1, true>), half 0 owns column
// tiles 0 .. 79 (exactly the single-workgroup layout), half 1 the tiles 80 + w + 8 k (NT1 per wave).  A row's sum is
// S_i = S_i(half 0) + S_i(half 1): each half publishes its 32 partial sums as 8-byte (value, token) words -- one relaxed device-scope
// atomic store per row, the token names (iteration, stage), so a value and its validity arrive together and nothing is ever reset -- and
// polls the partner's words.  Both halves then evaluate the same row update (half 0 stores it).  The two workgroups sit next to each other
// in one XCD's dispatch order (linear workgroup ids L and L + 8), so the partner of a resident workgroup is resident or next in line.  The
// wait is BOUNDED all the same: after OF_SPIN_LIMIT polls a workgroup computes the partner's sums itself (a plain loop that repeats the
// partner's operations in the partner's order: bitwise the same sums) -- slower, never wrong, never stuck.
// Which form a pair takes depends on its OWN target length only (both kernels run over a mixed group, each skipping the other's pairs).
constexpr int OF_FW = 8, OF_FT = 10, OF_HALF = OF_FW * OF_FT;
constexpr int OF_SPIN_LIMIT = 1 << 15;

struct Xchg {                              // (value, token) words of the cooperating halves: [unit][half][32]
    unsigned long long *words;
    int token0;                            // 4 * iteration + 1: stage s of this launch publishes token0 + s
};

// VAR: measurements / tests only.  ROREG_OT_FVAR: 1 no exponentials, 2 no MFMAs, 4 every fragment fetched twice, 5 no row update;
// 6 (tests) every strip through the stabilised redo; 7 (tests) the cooperating halves never see each other (always the bounded wait's fallback)
template <int NT, int NT_OTHER, bool COOP, int VAR>
__device__ __forceinline__ void of_iter_body(const Side &a, const Side &b, int nparts, int parity, int pair, int tA, int hb, unsigned long long *xw, int token0, const Conv &cv) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);       // (w in a scalar register: tile addresses are scalar)
    const int lenA = a.seg[pair + 1] - a.seg[pair], lenB = b.seg[pair + 1] - b.seg[pair];
    const int tilesB = lenB / 32 + 1;
    __shared__ float s_part[OF_FW][32];
    __shared__ float s_np[32];
    __shared__ float s_x[32];
    __shared__ int s_timeout;
    // the row this lane updates (both halves of every wave: row tA * 32 + lane % 32); its potential and log-marginal are fetched now
    // The row potentials are DOUBLE-BUFFERED (read `parity`, write the other) and their fp16 pieces are rebuilt here from the float
    // (split3 of the same number: the bits store_pieces would have left in the extras plane) instead of being read back from the
    // fragments: nothing a workgroup reads is written during the launch, so the two halves of a strip may run at any distance in time.
    const int i_row = tA * 32 + (lane & 31);
    const bool row_valid = i_row <= lenA;
    const float *pot_in = (parity ? a.pot2 : a.pot) + pair * a.pot_stride;
    float *pot = (parity ? a.pot : a.pot2) + pair * a.pot_stride;
    float old = 0.f, lmu = 0.f;
    if (row_valid) {
        old = pot_in[i_row];
        lmu = (i_row == lenA ? a.consts[pair * 2] + a.consts[pair * 2 + 1] : a.consts[pair * 2]) * LOG2E;
    }
    const f16x8 *fb = reinterpret_cast<const f16x8 *>(b.frag + pair * b.frag_stride);
    const int t_own = hb * OF_HALF, t_other = (1 - hb) * OF_HALF;
    f16x8 A[OF_PLANES];
    {
        const f16x8 *fa = reinterpret_cast<const f16x8 *>(a.frag + pair * a.frag_stride) + (size_t)tA * OF_PLANES * 64 + lane;
#pragma unroll
        for (int p = 0; p < OF_PLANES; ++p) A[p] = fa[p * 64];
        if (lane < 32 && row_valid) {                              // (lanes 0..31 carry k slots 0..7 of row `lane`; slots 0..2 = the row's potential; padding rows keep OF_PAD)
            _Float16 h, m, l;
            split3(old, h, m, l);
            A[2][0] = h; A[2][1] = m; A[2][2] = l;
        }
    }
    f32x16 E[NT];
    float red[16];
    f16x8 B[OF_PLANES];
    auto load_plane = [&](int k, int p) {                          // plane p of the wave's k-th tile (scalar base + the lane's offset)
        if (VAR == 8 && (k & 1)) return;                           // (measurement only: every other tile re-uses the previous tile's fragments -- half the traffic, wrong sums)
        const f16x8 *src = fb + (size_t)min(t_own + w + OF_FW * k, tilesB) * OF_PLANES * 64;
        B[p] = src[p * 64 + lane];
        if (VAR == 4) {                                            // (measurement only: the same traffic twice -- a second fetch of another tile's plane, result unused)
            const f16x8 *src2 = fb + (size_t)min(t_own + w + OF_FW * ((k + 3) % NT), tilesB) * OF_PLANES * 64;
            f16x8 dummy = src2[p * 64 + lane];
            asm volatile("" :: "v"(dummy));
        }
    };
    // A tile's chain of 7 MFMAs.  Order: each fragment plane of B in consecutive links, so that its registers can be reloaded for the NEXT tile
    // right behind its last use -- one fragment buffer, and every plane is requested 5-7 links before the chain that needs it.
    constexpr int PA[7] = {3, 0, 4, 1, 0, 1, 2}, PB[7] = {0, 0, 1, 1, 3, 4, 2};     // lo.hi, hi.hi (k 0-15); lo.hi, hi.hi (k 16-31); hi.lo; hi.lo; potentials
    constexpr int FREE[7] = {-1, 0, -1, 1, 3, 4, 2};                               // the plane whose last use link g is
    auto link = [&](int g, f32x16 &acc) {
        if (VAR == 2) { acc[g] = (float)B[PB[g]][0] + (float)A[PA[g]][1]; return; }
        if (g == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[PA[0]], B[PB[0]], f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[PA[g]], B[PB[g]], acc, 0, 0, 0);
    };
    auto reduce_to_lds = [&](bool is_max) {                        // red[] -> s_part[w][row]; ends behind a barrier
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = is_max ? half_max(red[r]) : half_sum(red[r]);
            if ((lane & 31) == 31) s_part[w][8 * (r >> 2) + 4 * (lane >> 5) + (r & 3)] = v;
        }
        __syncthreads();
    };
    auto phase1 = [&](auto mode_c) {
        constexpr int mode = decltype(mode_c)::value;
#pragma unroll
        for (int r = 0; r < 16; ++r) red[r] = mode == 1 ? -__builtin_inff() : 0.f;
        auto post = [&](int k, int r0, int r1) {                   // the vector-pipe part of tile k, accumulator registers r0 .. r1 - 1
            // (tile k's chain ended one link ago: 8 passes + the next link's issue slot = 9 wait states, the hazard table wants 11 between an
            //  8-pass MFMA and a VALU reader of its result, and the compiler does not count an inline-assembly reader)
            if (mode == 0 && r0 == 0) asm volatile("s_nop 3");
#pragma unroll
            for (int r = r0; r < r1; ++r) {
                if (mode == 0) {                                   // (inline assembly: the exponential replaces its operand IN PLACE -- no second copy of a tile's 16 registers)
                    float x = E[k][r];
                    if (VAR != 1) asm volatile("v_exp_f32 %0, %0\n\ts_nop 0" : "+v"(x));      // (+ the wait state a VALU reader of a transcendental's result needs on gfx940+: the hazard recogniser does not look inside)
                    else x = fabsf(x) * 1e-4f + 1e-4f;
                    E[k][r] = x;
                    red[r] += x;
                } else if (mode == 1) {
                    red[r] = fmaxf(red[r], E[k][r]);
                } else {                                           // (the maxima from LDS per use: this path is rare, registers are not)
                    E[k][r] = __builtin_amdgcn_exp2f(E[k][r] - s_np[8 * (r >> 2) + 4 * (lane >> 5) + (r & 3)]);
                    red[r] += E[k][r];
                }
            }
        };
        // Software pipeline: the links of tile k + 1's chain are issued INTERLEAVED with the exponentials and adds of tile k -- a wavefront
        // issues in order, so the vector work has to sit between the links to run in their shadow.
#pragma unroll
        for (int p = 0; p < OF_PLANES; ++p) load_plane(0, p);
#pragma unroll
        for (int g = 0; g < 7; ++g) {
            link(g, E[0]);
            __builtin_amdgcn_sched_barrier(0);
            if (NT > 1 && FREE[g] >= 0) load_plane(1, FREE[g]);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            if (k + 1 < NT) {
#pragma unroll
                for (int g = 0; g < 7; ++g) {
                    link(g, E[k + 1 < NT ? k + 1 : k]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (k + 2 < NT && FREE[g] >= 0) load_plane(k + 2, FREE[g]);
                    post(k, (16 * g) / 7, (16 * (g + 1)) / 7);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                // the last tile's chain has just been issued, and the in-place exponentials are inline assembly: the compiler's hazard
                // recogniser does not see them as readers of the MFMA result (no hardware interlock): 18 wait states by hand
                if (mode == 0) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 15\n\ts_nop 3"); __builtin_amdgcn_sched_barrier(0); }
                post(k, 0, 16);
            }
        }
        reduce_to_lds(mode == 1);
    };
    // The PARTNER's phase 1 (its sums only, nothing kept) as a plain rolled loop: the same operations on the same operands in the same
    // order as the partner's pipelined code, so s_part receives bitwise the partner's values.  One accumulator tile and one fragment plane
    // at a time beside the strip's own E (which stays where it is): the registers phase 1's fragment buffer and sums have just vacated.
    auto partner_sums = [&](int mode) {
        float pr[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) pr[r] = mode == 1 ? -__builtin_inff() : 0.f;
#pragma unroll 1
        for (int k = 0; k < NT_OTHER; ++k) {
            const f16x8 *src = fb + (size_t)min(t_other + w + OF_FW * k, tilesB) * OF_PLANES * 64 + lane;
            f32x16 acc = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < 7; ++g) {
                const f16x8 bo = src[PB[g] * 64];
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[PA[g]], bo, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (mode == 0) pr[r] += __builtin_amdgcn_exp2f(acc[r]);
                else if (mode == 1) pr[r] = fmaxf(pr[r], acc[r]);
                else pr[r] += __builtin_amdgcn_exp2f(acc[r] - s_np[8 * (r >> 2) + 4 * (lane >> 5) + (r & 3)]);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = mode == 1 ? half_max(pr[r]) : half_sum(pr[r]);
            if ((lane & 31) == 31) s_part[w][8 * (r >> 2) + 4 * (lane >> 5) + (r & 3)] = v;
        }
        __syncthreads();
    };
    using std::integral_constant;
    // this half's value of row lane % 32 (sum or maximum over its eight waves; s_part is complete behind phase1's barrier), and -- two
    // cooperating halves -- the partner's: published / polled by wave 0, handed to the other waves through LDS.  Returns the strip's value.
    auto combine = [&](auto mode_c, int stage) -> float {
        constexpr int mode = decltype(mode_c)::value;
        float own = mode == 1 ? -__builtin_inff() : 0.f;
#pragma unroll
        for (int q = 0; q < OF_FW; ++q) own = mode == 1 ? fmaxf(own, s_part[q][lane & 31]) : own + s_part[q][lane & 31];
        if (!COOP) return own;
        const unsigned token = (unsigned)(token0 + stage);
        if (w == 0) {
            unsigned long long *mine = xw + hb * 32 + (lane & 31), *theirs = xw + (1 - hb) * 32 + (lane & 31);
            if (lane < 32) __hip_atomic_store(mine, ((unsigned long long)token << 32) | __float_as_uint(own), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned long long v = 0;
            bool got = false;
            if (VAR != 7) {
                for (int spin = 0; spin < OF_SPIN_LIMIT; ++spin) {
                    if (!got) v = __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    got = (unsigned)(v >> 32) == token;
                    if (__all(got)) break;
                    __builtin_amdgcn_s_sleep(4);
                }
            }
            const bool timeout = !__all(got);
            if (lane < 32) s_x[lane] = __uint_as_float((unsigned)v);
            if (lane == 0) s_timeout = timeout ? 1 : 0;
        }
        __syncthreads();
        const bool timeout = s_timeout != 0;                       // (uniform over the workgroup)
        float other;
        if (!timeout) {
            other = s_x[lane & 31];
            __syncthreads();                                       // (s_x / s_timeout are rewritten by the next stage)
        } else {                                                   // the partner did not answer in time: its sums are computed here
            partner_sums(mode);
            other = mode == 1 ? -__builtin_inff() : 0.f;
#pragma unroll
            for (int q = 0; q < OF_FW; ++q) other = mode == 1 ? fmaxf(other, s_part[q][lane & 31]) : other + s_part[q][lane & 31];
            __syncthreads();
        }
        const float h0 = hb == 0 ? own : other, h1 = hb == 0 ? other : own;      // the same association in both halves
        return mode == 1 ? fmaxf(h0, h1) : h0 + h1;
    };
    phase1(integral_constant<int, 0>{});
    // ---- the strip's row update: every wave evaluates it for itself (same LDS data, same result: no second barrier); wave 0 stores it ----
    float frow = 0.f;                                              // f of row lane % 32
    float cn_np = 0.f, cn_old = 0.f;                               // (the step of this lane's row, for the convergence record)
    bool redo = false;
    if (VAR == 5) {
        frow = row_valid ? 1.f : 0.f;
    } else {
        const float S = combine(integral_constant<int, 0>{}, 0);
        const bool ok = S > 1e-35f && S < 1e35f;
        redo = VAR == 6 || __any(row_valid && !ok);                // (identical in all waves of both halves; VAR 6: the stabilised redo for EVERY strip -- tests only)
        if (!redo && row_valid) {
            const float np = old + (lmu - __log2f(S));             // u + log2 mu - log2 sum_j 2^(Z' + u + v)
            frow = __builtin_amdgcn_exp2f(np - old);               // E f = 2^(Z' + u_new + v)
            if (threadIdx.x < 32 && hb == 0) { pot[i_row] = np; cn_np = np; cn_old = old; }
        }
    }
    if (redo) {                                                    // rare: stabilised evaluation of the whole strip
        __syncthreads();                                           // (s_part is rewritten)
        phase1(integral_constant<int, 1>{});
        const float mxs = combine(integral_constant<int, 1>{}, 1);
        __syncthreads();
        if (threadIdx.x < 32) s_np[threadIdx.x] = mxs > -__builtin_inff() && mxs < __builtin_inff() ? mxs : 0.f;
        __syncthreads();
        phase1(integral_constant<int, 2>{});
        const float S = combine(integral_constant<int, 2>{}, 2);
        if (row_valid) {
            const float m = s_np[lane & 31];
            const float np = old + ((lmu - m) - __log2f(S));
            frow = __builtin_amdgcn_exp2f((np - old) + m);         // = mu_i / S_i: E f = 2^(Z' + u_new + v)
            if (threadIdx.x < 32 && hb == 0) { pot[i_row] = np; cn_np = np; cn_old = old; }
        }
    }
    if (w == 0) conv_note(cv, pair, cn_np, cn_old);                // (wave 0 whole: lanes 0 .. 31 of half 0 carry the strip's 32 steps)
    // ---- phase 2: the strip's part of the column sums ----
    float f[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) f[r] = __shfl(frow, 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3));
    float *out = b.part + pair * (b.pot_stride * nparts) + (size_t)tA * b.pot_stride;
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        const int t = t_own + w + OF_FW * k;
        float c = E[k][0] * f[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) c = fmaf(E[k][r], f[r], c);
        c += __shfl_xor(c, 32);
        if (t < tilesB && lane < 32) out[t * 32 + lane] = c;
    }
}

// one workgroup per strip: pairs whose target cloud has at most OF_HALF column tiles (others are left to the cooperating kernel)
template <int NT, int VAR = 0>
__global__ __launch_bounds__(64 * OF_FW) void of_iter_kernel(Side a, Side b, int nparts, int t, Conv cv) {
    const int pair = blockIdx.y, tA = blockIdx.x;
    const int tilesA = (a.seg[pair + 1] - a.seg[pair]) / 32 + 1, tilesB = (b.seg[pair + 1] - b.seg[pair]) / 32 + 1;
    if (tA >= tilesA || tilesB > OF_HALF || conv_done(cv, pair)) return;
    of_iter_body<NT, 0, false, VAR>(a, b, nparts, t & 1, pair, tA, 0, nullptr, 0, cv);
}

// two workgroups per strip: pairs with OF_HALF < column tiles <= 2 OF_HALF.  Linear workgroup id L -> XCD L % 8, position L / 8 in that
// XCD's dispatch order; positions 2 s and 2 s + 1 are the two halves of unit 8 s + XCD (unit = pair * ta + strip).
template <int NT1, int VAR = 0>
__global__ __launch_bounds__(64 * OF_FW) void of_iter_coop_kernel(Side a, Side b, int nparts, int parity, int ta, int units, Xchg x, Conv cv) {
    const int L = blockIdx.x, pos = L >> 3;
    const int unit = (pos >> 1) * 8 + (L & 7), hb = pos & 1;
    if (unit >= units) return;
    const int pair = unit / ta, tA = unit - pair * ta;
    const int tilesA = (a.seg[pair + 1] - a.seg[pair]) / 32 + 1, tilesB = (b.seg[pair + 1] - b.seg[pair]) / 32 + 1;
    if (tA >= tilesA || tilesB <= OF_HALF || tilesB > 2 * OF_HALF || conv_done(cv, pair)) return;
    unsigned long long *xw = x.words + (size_t)unit * 64;
    if (NT1 == OF_FT) of_iter_body<OF_FT, OF_FT, true, VAR>(a, b, nparts, parity, pair, tA, hb, xw, x.token0, cv);      // (one body for both halves)
    else if (hb == 0) of_iter_body<OF_FT, NT1, true, VAR>(a, b, nparts, parity, pair, tA, 0, xw, x.token0, cv);
    else of_iter_body<NT1, OF_FT, true, VAR>(a, b, nparts, parity, pair, tA, 1, xw, x.token0, cv);
}

// (tseg / fused_le / odd: the row potentials of a pair that took a whole-iteration kernel sit in the second buffer after an odd number of iterations)
// (with early exit the number of iterations a pair has run is its own: conv_executed)
__global__ __launch_bounds__(256) void of_export_kernel(Side a, float *__restrict__ out, size_t out_stride, const int *__restrict__ tseg, int fused_le, int iters,
                                                        const unsigned *__restrict__ flags, int flag_stride, unsigned long long *__restrict__ ran) {
    const int pair = blockIdx.y;
    const int len = a.seg[pair + 1] - a.seg[pair];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int done = conv_executed(flags, flag_stride, pair, iters);
    const bool second = (done & 1) && a.pot2 && (tseg[pair + 1] - tseg[pair]) / 32 + 1 <= fused_le;
    if (i <= len) out[pair * out_stride + i] = (second ? a.pot2 : a.pot)[pair * a.pot_stride + i] * LN2;
    if (ran != nullptr && i == 0) {                                    // statistics: (iterations run, pairs) summed over the calls since the last reset
        atomicAdd(reinterpret_cast<unsigned long long *>(ran), (unsigned long long)done);
        atomicAdd(reinterpret_cast<unsigned long long *>(ran) + 1, 1ull);
    }
}

__host__ __device__ inline size_t round_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace

namespace roreg {

constexpr int OF_R = 2;                                // row tiles per wave

// Column chunks per row block: ceil(tiles / OF_CHT) of the larger side -- fixed by the pair sizes alone (a wave that owns only a few column
// tiles spends its life starting up; 16 tiles per chunk and two row tiles per wave gave the best time on 30 stacked 2500-point pairs).
static int ot_flash_chunks(int max_m, int max_n) {
    const int tiles = (max_m > max_n ? max_m : max_n) / 32 + 1;
    return (tiles + OF_CHT - 1) / OF_CHT;
}

// Which form a PAIR takes depends on its own target cloud alone (so its result does not depend on what is stacked beside it): up to
// OF_HALF column tiles (2559 points) the whole-iteration kernel, one workgroup per strip; beyond, the two-pass form -- or, when the
// caller asks for it (`coop`), two cooperating workgroups per strip up to 2 OF_HALF tiles (5119 points).  Measured at 5000 x 5000 (26 stacked pairs,
// profiles/r05_sinkhorn_5000.txt): two passes 1.39 ms per pair, cooperating workgroups 1.47 -- the exchange of the row sums between the
// two halves costs what the second recomputation costs, so the simpler form is the default.  ROREG_OT_FUSED=0: two passes for everything.
static bool ot_flash_on() {
    static const bool on = !(getenv("ROREG_OT_FUSED") && atoi(getenv("ROREG_OT_FUSED")) == 0);
    return on;
}

// Early exit of converged pairs (struct Conv): on by default, ROREG_OT_EARLY_EXIT=0 or ot_flash_early_exit(0) runs every pair through all
// `iters` iterations like the reference's loop.
constexpr int OF_MAX_FLAG_ITERS = 256;                 // flags are kept for calls of up to this many iterations (longer ones run them all)
static int g_early_exit = -1;
static unsigned long long *g_iter_stats = nullptr;     // device: (iterations run, pairs) of every export since the last reset
int ot_flash_early_exit(int on) {
    if (g_early_exit < 0) g_early_exit = (getenv("ROREG_OT_EARLY_EXIT") && atoi(getenv("ROREG_OT_EARLY_EXIT")) == 0) ? 0 : 1;
    const int before = g_early_exit;
    if (on >= 0) g_early_exit = on ? 1 : 0;
    return before;
}
int ot_flash_iteration_stats(long long *iters_sum, long long *pairs, int reset, hipStream_t s) {
    unsigned long long h[2] = {0, 0};
    if (g_iter_stats != nullptr) {
        if (hipMemcpyAsync(h, g_iter_stats, sizeof(h), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return -1;
        if (reset && hipMemsetAsync(g_iter_stats, 0, sizeof(h), s) != hipSuccess) return -1;
    }
    *iters_sum = (long long)h[0]; *pairs = (long long)h[1];
    return 0;
}

// bytes of workspace of ot_flash_iterations (16-byte aligned pieces): per pair the two sides' fragments, potentials and chunk partials
size_t ot_flash_workspace_bytes(int n_seg, int max_m, int max_n) {
    const size_t ta = max_m / 32 + 1, tb = max_n / 32 + 1;
    const size_t nch = ot_flash_chunks(max_m, max_n);
    const bool fused = ot_flash_on();
    const size_t nparts = fused && ta > nch ? ta : nch;                          // side B's partial sums: one per row strip in the fused iteration
    const size_t per_pair = (ta + tb + 2) * OF_TILE_HALFS * sizeof(_Float16) + (2 * ta + tb) * 32 * sizeof(float) + (ta * nch + tb * nparts) * 32 * sizeof(float)
                            + (fused && tb > (size_t)OF_HALF ? ta * 64 * sizeof(unsigned long long) : 0);     // + the cooperating halves' words
    return (size_t)n_seg * per_pair + round_up((size_t)n_seg * 2 * sizeof(unsigned), 16) + (size_t)n_seg * OF_MAX_FLAG_ITERS * sizeof(unsigned) + 256;
}

// `iters` Sinkhorn iterations for every pair; the potentials (natural log, u[0..m], v[0..n]) are written to u_out + pair * uv_stride and
// v_out + pair * uv_stride.  seg_* are DEVICE offset arrays, consts the device array of roreg_sinkhorn_batch_consts.  min_n = the shortest
// target cloud of the group (which kernels have work); coop_wanted: cooperating workgroups instead of two passes for 2560 ... 5119 target points.
int ot_flash_iterations(const float *src, const float *tgt, const int32_t *seg_src, const int32_t *seg_tgt, const float *consts, int n_seg,
                        int max_m, int max_n, int min_n, bool coop_wanted, float alpha, int iters, float *u_out, float *v_out, size_t uv_stride, void *ws, hipStream_t s) {
    const size_t ta = max_m / 32 + 1, tb = max_n / 32 + 1;
    const int nch = ot_flash_chunks(max_m, max_n);
    char *p = reinterpret_cast<char *>(ws);
    p = reinterpret_cast<char *>(round_up(reinterpret_cast<uintptr_t>(p), 16));
    Side A, B;
    A.desc = src; A.seg = seg_src; A.consts = consts;
    B.desc = tgt; B.seg = seg_tgt; B.consts = consts + 2 * n_seg;
    A.frag_stride = (ta + 1) * OF_TILE_HALFS; B.frag_stride = (tb + 1) * OF_TILE_HALFS;      // (+ the pad tile)
    A.pot_stride = ta * 32; B.pot_stride = tb * 32;
    A.frag = reinterpret_cast<_Float16 *>(p); p += (size_t)n_seg * A.frag_stride * sizeof(_Float16);
    B.frag = reinterpret_cast<_Float16 *>(p); p += (size_t)n_seg * B.frag_stride * sizeof(_Float16);
    A.pot = reinterpret_cast<float *>(p); p += (size_t)n_seg * A.pot_stride * sizeof(float);
    A.pot2 = reinterpret_cast<float *>(p); p += (size_t)n_seg * A.pot_stride * sizeof(float);
    B.pot = reinterpret_cast<float *>(p); p += (size_t)n_seg * B.pot_stride * sizeof(float);
    B.pot2 = nullptr;
    static const int variant = getenv("ROREG_OT_VARIANT") ? atoi(getenv("ROREG_OT_VARIANT")) : 0;
    // a pair with at most fused_le target tiles takes a whole-iteration kernel, the others the two-pass form; which launches the group needs
    const int fused_le = !ot_flash_on() || variant != 0 ? 0 : (coop_wanted ? 2 : 1) * OF_HALF;
    const bool any_fused = min_n / 32 + 1 <= fused_le, any_two = (int)tb > fused_le;
    const bool coop = any_fused && (int)tb > OF_HALF && fused_le > OF_HALF, single = any_fused && min_n / 32 + 1 <= OF_HALF;
    const int nparts = any_fused && (int)ta > nch ? (int)ta : nch;
    A.part = reinterpret_cast<float *>(p); p += (size_t)n_seg * A.pot_stride * nch * sizeof(float);
    B.part = reinterpret_cast<float *>(p); p += (size_t)n_seg * B.pot_stride * nparts * sizeof(float);
    unsigned *amax = reinterpret_cast<unsigned *>(p); p += round_up((size_t)n_seg * 2 * sizeof(unsigned), 16);
    unsigned *flags = reinterpret_cast<unsigned *>(p); p += (size_t)n_seg * OF_MAX_FLAG_ITERS * sizeof(unsigned);
    Xchg X = {reinterpret_cast<unsigned long long *>(p), 0};
    const int units = n_seg * (int)ta;
    const bool early = ot_flash_early_exit(-1) != 0 && iters <= OF_MAX_FLAG_ITERS;
    if (g_iter_stats == nullptr) {
        if (hipMalloc(&g_iter_stats, 2 * sizeof(unsigned long long)) != hipSuccess) return -1;
        (void)hipMemsetAsync(g_iter_stats, 0, 2 * sizeof(unsigned long long), s);
    }
    if (early) (void)hipMemsetAsync(flags, 0, (size_t)n_seg * OF_MAX_FLAG_ITERS * sizeof(unsigned), s);
    static const float tol_scale = getenv("ROREG_OT_EXIT_TOL") ? (float)atof(getenv("ROREG_OT_EXIT_TOL")) : 1.0f;      // (measurements: multiples of the default tolerance)
    auto conv_of = [&](int it) { return early ? Conv{flags, it, OF_MAX_FLAG_ITERS, OF_TOL_REL * tol_scale, OF_TOL_ABS * tol_scale} : Conv{nullptr, 0, 0, 0.f, 0.f}; };
    (void)hipMemsetAsync(amax, 0, sizeof(unsigned) * 2 * n_seg, s);
    if (coop) (void)hipMemsetAsync(X.words, 0, (size_t)units * 64 * sizeof(unsigned long long), s);       // token 0 = nothing published
    hipLaunchKernelGGL(of_absmax_kernel, dim3(8, 2 * n_seg), dim3(256), 0, s, A, B, amax);      // 8 workgroups per (pair, side): 32 atomics each
    hipLaunchKernelGGL(of_prep_kernel, dim3((unsigned)(ta > tb ? ta : tb) + 1, 2, n_seg), dim3(320), 0, s, A, B, amax, alpha);
    const dim3 gA((unsigned)((ta + 4 * OF_R - 1) / (4 * OF_R)), nch, n_seg), gB((unsigned)((tb + 4 * OF_R - 1) / (4 * OF_R)), nch, n_seg);
    const dim3 uA((max_m + 256) / 256, n_seg), uB((max_n + 256) / 256, n_seg);
    if (iters > 0) {                                   // every pair's first stabiliser: minus the row maxima
        hipLaunchKernelGGL((of_pass_kernel<true, OF_R>), gA, dim3(256), 0, s, A, B, nch, seg_tgt, 0, Conv{nullptr, 0, 0, 0.f, 0.f});
        hipLaunchKernelGGL(of_update_kernel<true>, uA, dim3(256), 0, s, A, B, nch, OF_CHT, 0, alpha, seg_tgt, 0, Conv{nullptr, 0, 0, 0.f, 0.f});
    }
    for (int it = 0; it < iters; ++it) {
        const Conv cv = conv_of(it);
        if (variant >= 1 && variant <= 4) {            // measurements: the passes without exponentials / MFMAs / fragment loads / update launches
            using Pass = void (*)(Side, Side, int, const int *, int, Conv);
            const Pass pk = variant == 1 ? of_pass_kernel<false, OF_R, 1> : variant == 2 ? of_pass_kernel<false, OF_R, 2> : variant == 3 ? of_pass_kernel<false, OF_R, 3> : of_pass_kernel<false, OF_R>;
            hipLaunchKernelGGL(pk, gA, dim3(256), 0, s, A, B, nch, seg_tgt, 0, Conv{nullptr, 0, 0, 0.f, 0.f});
            hipLaunchKernelGGL(pk, gB, dim3(256), 0, s, B, A, nparts, seg_tgt, 0, Conv{nullptr, 0, 0, 0.f, 0.f});
            continue;
        }
        if (any_fused) {                               // the row update and the strips' column sums in one launch; then the column update
            static const int fvar = getenv("ROREG_OT_FVAR") ? atoi(getenv("ROREG_OT_FVAR")) : 0;
            if (single) {                              // pairs with <= OF_HALF column tiles
                using Kern = void (*)(Side, Side, int, int, Conv);
                static const Kern by_nt[OF_FT] = {of_iter_kernel<1>, of_iter_kernel<2>, of_iter_kernel<3>, of_iter_kernel<4>, of_iter_kernel<5>,
                                                  of_iter_kernel<6>, of_iter_kernel<7>, of_iter_kernel<8>, of_iter_kernel<9>, of_iter_kernel<10>};
                const int nt = (int)tb > OF_HALF ? OF_FT : (int)(tb + OF_FW - 1) / OF_FW;        // column tiles per wave
                Kern kern = by_nt[nt - 1];
                if (fvar && nt == OF_FT) kern = fvar == 1 ? of_iter_kernel<OF_FT, 1> : fvar == 2 ? of_iter_kernel<OF_FT, 2> : fvar == 4 ? of_iter_kernel<OF_FT, 4> : fvar == 6 ? of_iter_kernel<OF_FT, 6> : fvar == 5 ? of_iter_kernel<OF_FT, 5> : fvar == 8 ? of_iter_kernel<OF_FT, 8> : kern;
                hipLaunchKernelGGL(kern, dim3((unsigned)ta, n_seg), dim3(64 * OF_FW), 0, s, A, B, nparts, it, cv);
            }
            if (coop) {                                // pairs with more (ROREG_OT_COOP=1): two workgroups per strip
                using Kern = void (*)(Side, Side, int, int, int, int, Xchg, Conv);
                static const Kern by_nt[OF_FT / 2] = {of_iter_coop_kernel<2>, of_iter_coop_kernel<4>, of_iter_coop_kernel<6>, of_iter_coop_kernel<8>, of_iter_coop_kernel<10>};
                const int tbc = (int)tb < 2 * OF_HALF ? (int)tb : 2 * OF_HALF;
                const int nt1 = (tbc - OF_HALF + OF_FW - 1) / OF_FW;                 // the second half's column tiles per wave (instantiated for even counts)
                Kern kern = by_nt[(nt1 - 1) / 2];
                if (fvar == 6) kern = of_iter_coop_kernel<OF_FT, 6>;              // (tests: a full-length second half)
                if (fvar == 7) kern = of_iter_coop_kernel<OF_FT, 7>;
                X.token0 = 4 * it + 1;
                hipLaunchKernelGGL(kern, dim3(2 * (unsigned)round_up(units, 8)), dim3(64 * OF_FW), 0, s, A, B, nparts, it & 1, (int)ta, units, X, cv);
            }
            Side Anew = A;                             // (the column update's exact fall-back reads the row potentials just written)
            Anew.pot = (it & 1) ? A.pot : A.pot2;
            hipLaunchKernelGGL(of_update_cols_kernel, dim3((max_n + 64) / 64, n_seg), dim3(256), 0, s, B, Anew, nparts, alpha, fused_le, cv);
        }
        if (any_two) {                                 // the other pairs: rows, then columns (potentials updated in place)
            hipLaunchKernelGGL((of_pass_kernel<false, OF_R>), gA, dim3(256), 0, s, A, B, nch, seg_tgt, fused_le, cv);
            hipLaunchKernelGGL(of_update_kernel<false>, uA, dim3(256), 0, s, A, B, nch, OF_CHT, 0, alpha, seg_tgt, fused_le, cv);
            hipLaunchKernelGGL((of_pass_kernel<false, OF_R>), gB, dim3(256), 0, s, B, A, nparts, seg_tgt, fused_le, cv);
            hipLaunchKernelGGL(of_update_kernel<false>, uB, dim3(256), 0, s, B, A, nparts, OF_CHT, 3, alpha, seg_tgt, fused_le, cv);
        }
    }
    const unsigned *fl = early ? flags : nullptr;
    hipLaunchKernelGGL(of_export_kernel, uA, dim3(256), 0, s, A, u_out, uv_stride, seg_tgt, fused_le, iters, fl, OF_MAX_FLAG_ITERS, g_iter_stats);
    hipLaunchKernelGGL(of_export_kernel, uB, dim3(256), 0, s, B, v_out, uv_stride, seg_tgt, fused_le, iters, fl, OF_MAX_FLAG_ITERS, (unsigned long long *)nullptr);
    return 0;
}

}  // namespace roreg
