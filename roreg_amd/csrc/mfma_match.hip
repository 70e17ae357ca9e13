// Mutual nearest-neighbour matching of a batch of pairs on the matrix cores, with bit-exact results.
//
// The reference's distance d = sqrt(sum_f (s_f - t_f)^2 + 1e-7) (utils/knn_search.py:17-24) must give bit-identical
// indices, and its dot-product expansion |s|^2 + |t|^2 - 2 s.t rounds differently, so the expansion cannot DECIDE the nearest
// neighbour -- but it can bound it.  Per pair:
//   pass A (MFMA): a(i,j) = |s_i|^2 + |t_j|^2 - 2 s_i.t_j with the dot products as 3 x bf16 split MFMAs (f32-accurate, see
//           fourier.hip); row minima amin0[i] = min_j a and column minima amin1[j] = min_i a (one distance matrix serves both
//           search directions: the literal formula is bitwise symmetric in (s, t));
//   pass B (MFMA + exact check): every (i,j) with a(i,j) <= amin0[i] + margin (resp. amin1[j] + margin) is a CANDIDATE of row i
//           (column j); candidates -- 1-2 per row unless descriptors are near-duplicates, any number if they are -- are evaluated with
//           the literal formula (f order, no FMA contraction, correctly rounded sqrt) and merged with a 64-bit atomicMin on
//           (float_bits(d) << 32 | index) = "first minimum wins", exactly like nn_search_kernel.
// Why it is exact: |a - x| <= delta for the formula's radicand x (both are within a few 1e-6 relative of the real squared distance),
// so the true first minimum j* satisfies a(i,j*) <= x(i,j*) + delta <= x(i,j) (1 + 2^-22) + delta <= a(i,j) + 2 delta + tiny for the
// minimiser j of a: with margin = 1e-4 (|s_i|^2 + max|t|^2) + 1e-6 >= 5 (2 delta + tiny) it is always among the candidates, and the
// exact evaluation of the candidates then returns it.  The literal VALU scan (nn_search_kernel) is 56 us per 5000 x 5000 direction;
// this is the bf16 MFMA GEMM with fused mutual-NN argmin that the hot path calls for, made exact.
#include "common.h"

#pragma clang fp contract(off)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

struct MatchTask {                 // mirrors roreg_match_task (include/roreg_hip.h)
    const float *desc0, *desc1;
    const int64_t *rows0, *rows1;
    int32_t m0, m1;
};

constexpr int F = 32, TILE = 128, LP = 36;          // descriptor width, tile edge, LDS row pitch in floats (conflict-free b128 reads)
constexpr float PAD_NORM = 1e30f;                   // pad rows: never a minimum, never a candidate

// workspace of one task, in units of P = pitch (max_m rounded up to 128)
struct WS {
    float *G;                      // [2][P][32] gathered descriptors (side 0, side 1)
    float *N;                      // [2][P]     squared norms (PAD_NORM on pad rows)
    unsigned *AM;                  // [2][P]     ordered-uint keys of the approximate row / column minima
    unsigned long long *PK;        // [2][P]     packed exact (distance, index) keys
    unsigned *MX;                  // [2]        max squared norm bits of each side
};
__host__ __device__ inline size_t ws_task_bytes(int P) { return (size_t)2 * P * (F * 4 + 4 + 4 + 8) + 64; }
__device__ __forceinline__ WS ws_of(char *base, int task, int P) {
    char *p = base + (size_t)task * ws_task_bytes(P);
    WS w;
    w.PK = reinterpret_cast<unsigned long long *>(p); p += (size_t)2 * P * 8;
    w.G = reinterpret_cast<float *>(p); p += (size_t)2 * P * F * 4;
    w.N = reinterpret_cast<float *>(p); p += (size_t)2 * P * 4;
    w.AM = reinterpret_cast<unsigned *>(p); p += (size_t)2 * P * 4;
    w.MX = reinterpret_cast<unsigned *>(p);
    return w;
}

// float -> unsigned with the same total order (a can be slightly negative)
__device__ __forceinline__ unsigned f2o(float x) { const unsigned b = __float_as_uint(x); return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u); }
__device__ __forceinline__ float o2f(unsigned k) { return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu)); }

__global__ __launch_bounds__(256) void mm_prepare_kernel(const MatchTask *__restrict__ tasks, int P, char *__restrict__ wsb) {
    const int task = blockIdx.z >> 1, side = blockIdx.z & 1;
    const MatchTask t = tasks[task];
    const WS w = ws_of(wsb, task, P);
    const float *desc = side ? t.desc1 : t.desc0;
    const int64_t *rows = side ? t.rows1 : t.rows0;
    const int m = side ? t.m1 : t.m0;
    const int j = blockIdx.x * 8 + (threadIdx.x >> 5), f = threadIdx.x & 31;
    if (blockIdx.x == 0 && threadIdx.x == 0) w.MX[side] = 0u;       // (set again by every z-slice's first block: same value)
    if (j >= P) return;
    float v = 0.f;
    if (j < m) v = desc[(size_t)(rows ? rows[j] : (int64_t)j) * F + f];
    w.G[((size_t)side * P + j) * F + f] = v;
    float s = v * v;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (f == 0) {
        w.N[(size_t)side * P + j] = j < m ? s : PAD_NORM;
        w.AM[(size_t)side * P + j] = 0xffffffffu;
        w.PK[(size_t)side * P + j] = ~0ull;
    }
}

__global__ __launch_bounds__(256) void mm_maxnorm_kernel(const MatchTask *__restrict__ tasks, int P, char *__restrict__ wsb) {
    const int task = blockIdx.z >> 1, side = blockIdx.z & 1;
    const MatchTask t = tasks[task];
    const WS w = ws_of(wsb, task, P);
    const int m = side ? t.m1 : t.m0;
    const int j = blockIdx.x * 256 + threadIdx.x;
    float v = (j < m) ? w.N[(size_t)side * P + j] : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    if ((threadIdx.x & 63) == 0 && v > 0.f) atomicMax(&w.MX[side], __float_as_uint(v));
}

__device__ __forceinline__ void split3(const float (&v)[8], bf16x8 &b1, bf16x8 &b2, bf16x8 &b3) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 h1 = (__bf16)v[e];
        const float r1 = v[e] - (float)h1;
        const __bf16 h2 = (__bf16)r1;
        const float r2 = r1 - (float)h2;
        b1[e] = h1; b2[e] = h2; b3[e] = (__bf16)r2;
    }
}

__device__ __forceinline__ f32x16 mfma6(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);
    return c;
}

// fp16 x 2 operands (round 2): every descriptor ROW carries its own power-of-two scale 2^e with 2^e |x_f| <= 2^e |x| < 2^14, taken from the
// row's squared norm; hi = fp16(2^e x), lo = fp16(2^e x - hi); three cross products (lo.hi, hi.lo, hi.hi), f32 accumulate, the tile
// element rescaled by the exact 2^-(e_i + e_j).  Error of the dot product <= ~3 * 2^-22 sum_f |s_f t_f| + 32 * 2^-39 |s||t| (elements
// more than 11 binades below the row's norm keep an absolute error of 2^-39 of it) < 1e-6 |s||t|: fifty times inside the candidate
// margin of 1e-4 (|s_i|^2 + max |t|^2), which is relative to the ROW's own norm -- hence the per-row scale.  Half the MFMAs of 3 x bf16.
__device__ __forceinline__ int norm_exp(float n2) {
    if (!(n2 > 0.f) || !(n2 < __builtin_inff())) return 0;
    int ex;
    (void)frexpf(n2, &ex);                          // n2 = m 2^ex, m in [0.5, 1): |x| = sqrt(n2) < 2^ceil(ex / 2)
    return 14 - ((ex + 1) >> 1);
}
__device__ __forceinline__ void split2(const float (&v)[8], float scale, f16x8 &hi, f16x8 &lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = v[e] * scale;
        const _Float16 h1 = (_Float16)x;
        hi[e] = h1; lo[e] = (_Float16)(x - (float)h1);
    }
}
__device__ __forceinline__ f32x16 mfma3(const f16x8 (&a)[2], const f16x8 (&b)[2], f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], c, 0, 0, 0);
    return c;
}

// the reference's literal distance (nn_search_kernel's arithmetic)
__device__ __forceinline__ float exact_dist(const float *__restrict__ s, const float *__restrict__ t) {
    float acc = 0.f;
#pragma unroll
    for (int f = 0; f < F; ++f) {
        const float d = __fsub_rn(s[f], t[f]);
        acc = __fadd_rn(acc, __fmul_rn(d, d));
    }
    return sqrtf(__fadd_rn(acc, 1e-7f));
}

// One 128 x 128 tile of the distance matrix of one pair.  VERIFY = false: approximate row / column minima.  VERIFY = true:
// candidates -> exact distances -> packed keys.  4 waves as a 2 x 2 grid of 64 x 64 blocks (2 x 2 MFMA tiles each).
template <bool VERIFY, int NP /* 3: bf16 x 3, 2: fp16 x 2 with per-row scales */>
__global__ __launch_bounds__(256, 2) void mm_tile_kernel(const MatchTask *__restrict__ tasks, int P, char *__restrict__ wsb) {
    __shared__ __attribute__((aligned(16))) float S[TILE * LP], T[TILE * LP];
    __shared__ float n0s[TILE], n1s[TILE], thr0[TILE], thr1[TILE];
    __shared__ float xs0[TILE], xs1[TILE], is0[TILE], is1[TILE];     // NP = 2: the rows' scales 2^e and 2^-e (1 for NP = 3)
    const int task = blockIdx.z;
    const MatchTask t = tasks[task];
    const int i0 = blockIdx.y * TILE, j0 = blockIdx.x * TILE;
    if (i0 >= t.m0 || j0 >= t.m1) return;
    const WS w = ws_of(wsb, task, P);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv & 1, wn = wv >> 1, jl = lane & 31, h = lane >> 5;

    // ---- stage the two 128 x 32 descriptor tiles (pitch 36) and the per-row constants ------------------------------
    {
        const float4 *g0 = reinterpret_cast<const float4 *>(w.G + (size_t)i0 * F);
        const float4 *g1 = reinterpret_cast<const float4 *>(w.G + ((size_t)P + j0) * F);
        for (int q = tid; q < TILE * F / 4; q += 256) {
            const int r = q >> 3, c4 = q & 7;
            *reinterpret_cast<float4 *>(S + r * LP + c4 * 4) = g0[q];
            *reinterpret_cast<float4 *>(T + r * LP + c4 * 4) = g1[q];
        }
        if (tid < TILE) {
            const float a = w.N[i0 + tid], b = w.N[(size_t)P + j0 + tid];
            n0s[tid] = a; n1s[tid] = b;
            const int ea = NP == 2 ? norm_exp(a) : 0, eb = NP == 2 ? norm_exp(b) : 0;
            xs0[tid] = ldexpf(1.f, ea); is0[tid] = ldexpf(1.f, -ea); xs1[tid] = ldexpf(1.f, eb); is1[tid] = ldexpf(1.f, -eb);
            if (VERIFY) {
                const float mx0 = __uint_as_float(w.MX[0]), mx1 = __uint_as_float(w.MX[1]);
                thr0[tid] = o2f(w.AM[i0 + tid]) + (1e-4f * (a + mx1) + 1e-6f);
                thr1[tid] = o2f(w.AM[(size_t)P + j0 + tid]) + (1e-4f * (b + mx0) + 1e-6f);
            }
        }
    }
    __syncthreads();

    // ---- a(i,j) on the matrix cores: C = S.T^T (lanes own columns j), and for the minima also C' = T.S^T (lanes own columns i) ----
    f32x16 c[2][2], ct[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) { c[a][b][r] = 0.f; ct[a][b][r] = 0.f; }
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        using frag = typename std::conditional<NP == 3, bf16x8, f16x8>::type;
        frag sf[2][NP], tf[2][NP];                  // fragments of the wave's two 32-row groups of S (rows wm*64..) and T (rows wn*64..)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            float x8[8];
            const int rs = wm * 64 + g * 32 + jl, rt = wn * 64 + g * 32 + jl;
            const float4 *ps = reinterpret_cast<const float4 *>(S + rs * LP + st * 16 + h * 8);
            const float4 u0 = ps[0], u1 = ps[1];
            x8[0] = u0.x; x8[1] = u0.y; x8[2] = u0.z; x8[3] = u0.w; x8[4] = u1.x; x8[5] = u1.y; x8[6] = u1.z; x8[7] = u1.w;
            if constexpr (NP == 3) split3(x8, sf[g][0], sf[g][1], sf[g][2]); else split2(x8, xs0[rs], sf[g][0], sf[g][1]);
            const float4 *pt = reinterpret_cast<const float4 *>(T + rt * LP + st * 16 + h * 8);
            const float4 v0 = pt[0], v1 = pt[1];
            x8[0] = v0.x; x8[1] = v0.y; x8[2] = v0.z; x8[3] = v0.w; x8[4] = v1.x; x8[5] = v1.y; x8[6] = v1.z; x8[7] = v1.w;
            if constexpr (NP == 3) split3(x8, tf[g][0], tf[g][1], tf[g][2]); else split2(x8, xs1[rt], tf[g][0], tf[g][1]);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                if constexpr (NP == 3) {
                    c[a][b] = mfma6(sf[a], tf[b], c[a][b]);                 // rows i = wm*64 + a*32 + .., column j = wn*64 + b*32 + jl
                    if (!VERIFY) ct[a][b] = mfma6(tf[a], sf[b], ct[a][b]);  // rows j = wn*64 + a*32 + .., column i = wm*64 + b*32 + jl
                } else {
                    c[a][b] = mfma3(sf[a], tf[b], c[a][b]);
                    if (!VERIFY) ct[a][b] = mfma3(tf[a], sf[b], ct[a][b]);
                }
            }
    }
    // the dot product of row i and column j in real units: accumulator * 2^-e_i * 2^-e_j (exact; both factors are 1 for NP = 3)
#define DOT_C(a, b, r, ir, jc) (NP == 2 ? (c[a][b][r] * is0[ir]) * is1[jc] : c[a][b][r])
#define DOT_CT(a, b, r, jr, ic) (NP == 2 ? (ct[a][b][r] * is1[jr]) * is0[ic] : ct[a][b][r])

    if (!VERIFY) {
        // column minima of C: over the wave's 64 rows i, for column j  -> amin1[j]
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int jc = wn * 64 + b * 32 + jl;
            const float nj = n1s[jc];
            float mn = __builtin_inff();
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ir = wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    mn = fminf(mn, (n0s[ir] + nj) - 2.f * DOT_C(a, b, r, ir, jc));
                }
            mn = fminf(mn, __shfl_xor(mn, 32));
            if (h == 0 && j0 + jc < t.m1) atomicMin(&w.AM[(size_t)P + j0 + jc], f2o(mn));
        }
        // column minima of C': over the wave's 64 rows j, for column i -> amin0[i]
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int ic = wm * 64 + b * 32 + jl;
            const float ni = n0s[ic];
            float mn = __builtin_inff();
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int jr = wn * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    mn = fminf(mn, (ni + n1s[jr]) - 2.f * DOT_CT(a, b, r, jr, ic));
                }
            mn = fminf(mn, __shfl_xor(mn, 32));
            if (h == 0 && i0 + ic < t.m0) atomicMin(&w.AM[i0 + ic], f2o(mn));
        }
    } else {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int jc = wn * 64 + b * 32 + jl;
            const float nj = n1s[jc], tj = thr1[jc];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ir = wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const float av = (n0s[ir] + nj) - 2.f * DOT_C(a, b, r, ir, jc);
                    const bool k0 = av <= thr0[ir], k1 = av <= tj;          // candidate of row i / of column j
                    if (k0 || k1) {
                        const float d = exact_dist(S + ir * LP, T + jc * LP);
                        const unsigned long long db = (unsigned long long)__float_as_uint(d) << 32;
                        if (k0 && j0 + jc < t.m1) atomicMin(&w.PK[i0 + ir], db | (unsigned)(j0 + jc));
                        if (k1 && i0 + ir < t.m0) atomicMin(&w.PK[(size_t)P + j0 + jc], db | (unsigned)(i0 + ir));
                    }
                }
        }
    }
#undef DOT_C
#undef DOT_CT
}

// ---------------------------------------------------------------------------------------------------------------
// Row-strip form of the two passes (round 2): one workgroup owns a 128-row strip of S and walks ALL column tiles of T.
//  * S is staged and split into fragment registers once per strip; the T tiles stream through a double-buffered LDS area by LDS-DMA,
//    the NEXT tile in flight under the current tile's MFMAs and epilogue -- the per-tile form above stages 32 KB
//    and waits for it before every 48-96 MFMAs (latency-bound: 2 workgroups per CU cannot cover a 1-2 us round trip per tile);
//  * row minima are an ELEMENTWISE running minimum in registers across the strip (one cross-lane reduction at the end), so the second
//    product C' = T.S^T of pass A is not needed at all: two products per pair instead of three;
//  * column minima go to an LDS array over all columns (LDS atomics) and are flushed once per strip.
// Tiles are stored unpadded with their 16-byte pieces swizzled (piece c of row r at slot c ^ ((r >> 1) & 7)): what LDS-DMA can write
// (lane-linear destination, the permutation applied to the SOURCE address) and conflict-free for the fragment reads' lane groups.
constexpr int TP = TILE * F;                         // floats per tile
__device__ __forceinline__ int swz_piece(int row, int c) { return row * 8 + (c ^ ((row >> 1) & 7)); }
__device__ __forceinline__ float exact_dist_swz(const float *__restrict__ s_tile, int ir, const float *__restrict__ t_tile, int jc) {
    float acc = 0.f;
#pragma unroll
    for (int f = 0; f < F; ++f) {
        const float sv = s_tile[swz_piece(ir, f >> 2) * 4 + (f & 3)], tv = t_tile[swz_piece(jc, f >> 2) * 4 + (f & 3)];
        const float d = __fsub_rn(sv, tv);
        acc = __fadd_rn(acc, __fmul_rn(d, d));
    }
    return sqrtf(__fadd_rn(acc, 1e-7f));
}

template <bool VERIFY>
__global__ __launch_bounds__(256, 2) void mm_strip_kernel(const MatchTask *__restrict__ tasks, int P, char *__restrict__ wsb) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *S = reinterpret_cast<float *>(smem);                  // [128][32] swizzled
    float *Tb = S + TP;                                           // [2][128][32] swizzled
    float *n0s = Tb + 2 * TP, *xs0 = n0s + TILE, *is0 = xs0 + TILE, *thr0 = is0 + TILE;
    float *n1s = thr0 + TILE, *xs1 = n1s + 2 * TILE, *is1 = xs1 + 2 * TILE, *thr1 = is1 + 2 * TILE;      // [2][128] each
    unsigned *cm = reinterpret_cast<unsigned *>(thr1 + 2 * TILE);                                          // [P] (pass A)
    const int task = blockIdx.y;
    const MatchTask t = tasks[task];
    const int i0 = blockIdx.x * TILE;
    if (i0 >= t.m0 || t.m1 <= 0) return;
    const WS w = ws_of(wsb, task, P);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv & 1, wn = wv >> 1, jl = lane & 31, h = lane >> 5;
    const int ntj = (t.m1 + TILE - 1) / TILE;
    const float mx0 = VERIFY ? __uint_as_float(w.MX[0]) : 0.f, mx1 = VERIFY ? __uint_as_float(w.MX[1]) : 0.f;

    auto dma_tile = [&](const float *src, float *dst) {          // 128 rows x 128 B, contiguous in global memory
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int q = k * 256 + tid, row = q >> 3, cs = q & 7;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + swz_piece(row, cs) * 4),
                                             (__attribute__((address_space(3))) void *)(dst + (k * 256 + wv * 64) * 4), 16, 0, 0);
        }
    };
    // per-column constants of a tile: waves 0-1 fetch the squared norms, waves 2-3 (pass B) the approximate column minima
    auto load_consts = [&](int jt) -> float {
        const int r = tid & (TILE - 1);
        if (tid < TILE) return w.N[(size_t)P + jt * TILE + r];
        return VERIFY ? o2f(w.AM[(size_t)P + jt * TILE + r]) : 0.f;
    };
    auto store_consts = [&](int buf, float v) {                  // (the norm's thread also derives the scales; the threshold needs the norm: second half reads it)
        const int r = tid & (TILE - 1);
        if (tid < TILE) {
            const int e = norm_exp(v);
            n1s[buf * TILE + r] = v; xs1[buf * TILE + r] = ldexpf(1.f, e); is1[buf * TILE + r] = ldexpf(1.f, -e);
        } else if (VERIFY) thr1[buf * TILE + r] = v;             // approximate minimum; the margin is added at use (it needs the norm)
    };

    dma_tile(w.G + (size_t)i0 * F, S);
    dma_tile(w.G + (size_t)P * F, Tb);
    {
        const float c0 = load_consts(0);
        if (tid < TILE) {
            const float a = w.N[i0 + tid];
            const int e = norm_exp(a);
            n0s[tid] = a; xs0[tid] = ldexpf(1.f, e); is0[tid] = ldexpf(1.f, -e);
            if (VERIFY) thr0[tid] = o2f(w.AM[i0 + tid]) + (1e-4f * (a + mx1) + 1e-6f);
        }
        store_consts(0, c0);
        if (!VERIFY)
            for (int j = tid; j < P; j += 256) cm[j] = 0xffffffffu;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();

    // the strip's S fragments: [st][g][hi, lo]
    f16x8 sf[2][2][2];
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int rs = wm * 64 + g * 32 + jl;
            const float4 u0 = *reinterpret_cast<const float4 *>(S + swz_piece(rs, st * 4 + h * 2) * 4);
            const float4 u1 = *reinterpret_cast<const float4 *>(S + swz_piece(rs, st * 4 + h * 2 + 1) * 4);
            const float x8[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
            split2(x8, xs0[rs], sf[st][g][0], sf[st][g][1]);
        }
    f32x16 rmin[2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) rmin[a][r] = __builtin_inff();

    for (int jt = 0; jt < ntj; ++jt) {
        const int buf = jt & 1, j0 = jt * TILE;
        const float *T = Tb + buf * TP;
        float cnext = 0.f;
        if (jt + 1 < ntj) {                                      // (uniform) the next tile travels under this tile's work
            dma_tile(w.G + ((size_t)P + j0 + TILE) * F, Tb + (buf ^ 1) * TP);
            cnext = load_consts(jt + 1);
        }
        f32x16 c[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) c[a][b][r] = 0.f;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            f16x8 tf[2][2];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int rt = wn * 64 + g * 32 + jl;
                const float4 v0 = *reinterpret_cast<const float4 *>(T + swz_piece(rt, st * 4 + h * 2) * 4);
                const float4 v1 = *reinterpret_cast<const float4 *>(T + swz_piece(rt, st * 4 + h * 2 + 1) * 4);
                const float x8[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                split2(x8, xs1[buf * TILE + rt], tf[g][0], tf[g][1]);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) c[a][b] = mfma3(sf[st][a], tf[b], c[a][b]);
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int jc = wn * 64 + b * 32 + jl;
            const float nj = n1s[buf * TILE + jc], isj = is1[buf * TILE + jc];
            if constexpr (!VERIFY) {
                float mn = __builtin_inff();
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ir = wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        const float av = (n0s[ir] + nj) - 2.f * ((c[a][b][r] * is0[ir]) * isj);
                        mn = fminf(mn, av);
                        rmin[a][r] = fminf(rmin[a][r], av);
                    }
                mn = fminf(mn, __shfl_xor(mn, 32));
                if (h == 0) atomicMin(cm + j0 + jc, f2o(mn));                      // LDS
            } else {
                const float tj = thr1[buf * TILE + jc] + (1e-4f * (nj + mx0) + 1e-6f);
                // candidate masks first (bit a*16 + r), the rare exact evaluations afterwards in ONE loop body: 64 inlined copies of the
                // literal distance made the kernel 100 KB of code and spilled registers
                unsigned m0 = 0u, m1 = 0u;
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ir = wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        const float av = (n0s[ir] + nj) - 2.f * ((c[a][b][r] * is0[ir]) * isj);
                        m0 |= (av <= thr0[ir] ? 1u : 0u) << (a * 16 + r);          // candidate of row i
                        m1 |= (av <= tj ? 1u : 0u) << (a * 16 + r);                // candidate of column j
                    }
                unsigned m = m0 | m1;
                while (m) {
                    const int q = __ffs(m) - 1;
                    m &= m - 1;
                    const int ir = wm * 64 + (q >> 4) * 32 + (q & 3) + 8 * ((q & 15) >> 2) + 4 * h;
                    const float d = exact_dist_swz(S, ir, T, jc);
                    const unsigned long long db = (unsigned long long)__float_as_uint(d) << 32;
                    if (((m0 >> q) & 1u) && j0 + jc < t.m1) atomicMin(&w.PK[i0 + ir], db | (unsigned)(j0 + jc));
                    if (((m1 >> q) & 1u) && i0 + ir < t.m0) atomicMin(&w.PK[(size_t)P + j0 + jc], db | (unsigned)(i0 + ir));
                }
            }
        }
        if (jt + 1 < ntj) store_consts(buf ^ 1, cnext);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");              // the next tile has landed, this wave is done with the current one
        __syncthreads();
    }
    if constexpr (!VERIFY) {
        // row minima: reduce the elementwise minima over the 32 column lanes; the two column-half waves of a row group meet in the atomic
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = rmin[a][r];
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
                const int ir = wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (jl == 0 && i0 + ir < t.m0) atomicMin(&w.AM[i0 + ir], f2o(v));
            }
        for (int j = tid; j < t.m1; j += 256) atomicMin(&w.AM[(size_t)P + j], cm[j]);     // (complete: the loop ended with a barrier)
    }
}

__global__ __launch_bounds__(1024) void mm_mutual_kernel(const MatchTask *__restrict__ tasks, int P, char *__restrict__ wsb, int out_pitch,
                                                         int64_t *__restrict__ match_all, int32_t *__restrict__ counts) {
    __shared__ int wave_cnt[16];
    __shared__ int base;
    const int task = blockIdx.x;
    const MatchTask t = tasks[task];
    const WS w = ws_of(wsb, task, P);
    const unsigned long long *p01 = w.PK, *p10 = w.PK + P;
    int64_t *match_out = match_all + (size_t)task * out_pitch * 2;
    const int m = t.m1 > 0 ? t.m0 : 0;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int start = 0; start < m; start += 1024) {
        const int i = start + tid;
        bool keep = false;
        int64_t j = 0;
        if (i < m) {
            j = (int64_t)(p01[i] & 0xffffffffu);
            // a key that no candidate ever updated (non-finite descriptors: every comparison with NaN is false) keeps its initial
            // index bits; such a point is unmatched, and the index is never dereferenced
            keep = j < (int64_t)t.m1 && (int64_t)(p10[j] & 0xffffffffu) == (int64_t)i;
        }
        const unsigned long long mask = __ballot(keep);
        const int before = __popcll(mask & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wv] = __popcll(mask);
        __syncthreads();
        int woff = 0, total = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int cq = wave_cnt[q];
            if (q < wv) woff += cq;
            total += cq;
        }
        const int b0 = base;
        if (keep) {
            const int pos = b0 + woff + before;
            match_out[2 * pos] = t.rows0 ? t.rows0[i] : (int64_t)i;
            match_out[2 * pos + 1] = t.rows1 ? t.rows1[j] : j;
        }
        __syncthreads();
        if (tid == 0) base = b0 + total;
        __syncthreads();
    }
    if (tid == 0) counts[task] = base;
}

}  // namespace

extern "C" size_t roreg_mutual_match_batch_workspace(int n_tasks, int max_m) {
    const int P = (max_m + TILE - 1) / TILE * TILE;
    return (size_t)n_tasks * ws_task_bytes(P > 0 ? P : TILE);
}

extern "C" int roreg_mutual_match_batch(const roreg_match_task *tasks_dev, int n_tasks, int max_m, int64_t *match_out, int32_t *counts_out,
                                        void *workspace, size_t workspace_bytes, void *stream) {
    if (n_tasks == 0) return 0;
    ROREG_REQUIRE(tasks_dev && match_out && counts_out && workspace && n_tasks > 0 && max_m >= 0, "roreg_mutual_match_batch: bad arguments");
    ROREG_REQUIRE(workspace_bytes >= roreg_mutual_match_batch_workspace(n_tasks, max_m), "roreg_mutual_match_batch: workspace too small");
    ROREG_REQUIRE(n_tasks <= 65535, "roreg_mutual_match_batch: too many tasks in one call (%d)", n_tasks);
    static_assert(sizeof(roreg_match_task) == sizeof(MatchTask), "roreg_match_task layout");
    hipStream_t s = roreg::as_stream(stream);
    const MatchTask *tasks = reinterpret_cast<const MatchTask *>(tasks_dev);
    const int P = max_m > 0 ? (max_m + TILE - 1) / TILE * TILE : TILE;
    char *ws = reinterpret_cast<char *>(workspace);
    const int out_pitch = (max_m + 1) & ~1;
    if (max_m > 0) {
        hipLaunchKernelGGL(mm_prepare_kernel, dim3(P / 8, 1, 2 * n_tasks), dim3(256), 0, s, tasks, P, ws);
        hipLaunchKernelGGL(mm_maxnorm_kernel, dim3(P / 256 + 1, 1, 2 * n_tasks), dim3(256), 0, s, tasks, P, ws);
        const dim3 grid(P / TILE, P / TILE, n_tasks);
        roreg::ProfScope prof(roreg::PROF_MM_TILE, s);        // (one scope = the two passes of the distance matrix)
        // ROREG_MATCH_BF16X3=1 selects the 3 x bf16 operand split of round 1 (twice the MFMAs; same results: the exact check decides);
        // ROREG_MATCH_TILES=1 the per-tile form of the fp16 x 2 kernel instead of the row-strip form
        static const bool bf16x3 = [] { const char *e = getenv("ROREG_MATCH_BF16X3"); return e && e[0] == '1'; }();
        static const bool tiles = [] { const char *e = getenv("ROREG_MATCH_TILES"); return e && e[0] == '1'; }();
        const size_t lds_strip = (size_t)(3 * TP + 12 * TILE) * 4 + (size_t)P * 4;
        if (!bf16x3 && !tiles && lds_strip <= 80 * 1024) {
            const dim3 sgrid(P / TILE, n_tasks);
            auto ka = mm_strip_kernel<false>; auto kb = mm_strip_kernel<true>;
            hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void *>(ka), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_strip);
            hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void *>(kb), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_strip);
            if (e1 != hipSuccess || e2 != hipSuccess) { roreg::set_error("roreg_mutual_match_batch: hipFuncSetAttribute(%zu) failed", lds_strip); return 1; }
            hipLaunchKernelGGL(ka, sgrid, dim3(256), lds_strip, s, tasks, P, ws);
            hipLaunchKernelGGL(kb, sgrid, dim3(256), lds_strip, s, tasks, P, ws);
        } else if (bf16x3) {
            hipLaunchKernelGGL((mm_tile_kernel<false, 3>), grid, dim3(256), 0, s, tasks, P, ws);
            hipLaunchKernelGGL((mm_tile_kernel<true, 3>), grid, dim3(256), 0, s, tasks, P, ws);
        } else {
            hipLaunchKernelGGL((mm_tile_kernel<false, 2>), grid, dim3(256), 0, s, tasks, P, ws);
            hipLaunchKernelGGL((mm_tile_kernel<true, 2>), grid, dim3(256), 0, s, tasks, P, ws);
        }
    }
    hipLaunchKernelGGL(mm_mutual_kernel, dim3(n_tasks), dim3(1024), 0, s, tasks, P, ws, out_pitch, match_out, counts_out);
    ROREG_CHECK_LAUNCH("roreg_mutual_match_batch");
    return 0;
}
