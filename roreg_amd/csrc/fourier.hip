// Group-Fourier (irrep-domain) evaluation of the icosahedral group convolution.
//
// The 13-stencil group conv is a correlation on the 60-element group, so in the basis of the real irreps
// (d = 1,3,3,4,5) it is, per irrep, ONE dense row-major GEMM
//     Out_rho [M = d*O][N = d*B]  =  W_rho [M][K = d*C]  .  X_rho [K][N]
// with sum_d d^3 = 244 multiply-adds per (o,c) pair instead of 60*13 = 780 (roreg_amd/fourier.py has the algebra).
// Coefficient tensors are channel-major: X_rho[(l,c)][(i,b)] holds x~_{b,c}(rho)[i][l]; keypoints b are the fastest
// axis, so both kernels below stream fully coalesced rows.
//
//   irrep_gemm_kernel : the five GEMMs of a layer in one launch (tile table), exact-f32 MFMA, double-buffered LDS
//                       tiles of X, weights packed in fragment order exactly like group_conv.hip.
//   ft_nonlin_kernel  : per (keypoint, channel): inverse transform -> +bias (+residual) -> BatchNorm -> ReLU ->
//                       forward transform, as two chained 60x60 MFMA products whose intermediate never leaves the
//                       accumulator registers (the K order of the second product is chosen to be the C/D register
//                       layout of the first, so no transpose is needed).
#include "common.h"
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <utility>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int NIRR = 5;

struct GemmDescs {
    const float *X[NIRR];
    float *Out[NIRR];
    const float *Add[NIRR];       // optional: Out = W.X + Add (residual in the irrep domain), or all null
    const float4 *W[NIRR];
    int K[NIRR], M[NIRR], Mpad[NIRR], N[NIRR];
};

inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

// ---------------------------------------------------------------------------------------------------------------
// X tiles go global -> LDS with the LDS-DMA path (global_load_lds_dwordx4: one wave instruction moves one 1-KiB tile row, no
// VGPR staging), which frees the registers to double-buffer the weight fragments of the next K chunk as well: during the
// 128 MFMAs of a chunk both operands of the next chunk are in flight.
template <int CT>
__global__ __launch_bounds__(256, 2) void irrep_gemm_kernel(GemmDescs p, const int *__restrict__ tiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *xs = reinterpret_cast<float *>(smem);                 // [2][CT][256]
    constexpr int NCOL = 256, OT = 128, NQ = CT / 8;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int irr = tiles[blockIdx.x * 3], mt = tiles[blockIdx.x * 3 + 1], nt = tiles[blockIdx.x * 3 + 2];
    if (irr < 0) return;
    const float *__restrict__ X = p.X[irr];
    const float4 *__restrict__ W = p.W[irr];
    const int K = p.K[irr], M = p.M[irr], Mpad = p.Mpad[irr], N = p.N[irr];
    const int wo = w & 1, wb = w >> 1;
    const int m_wave = mt * OT + wo * 64;
    const int n0 = nt * NCOL;
    const int ncol_wave = wb * 128;

    f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][t][r] = 0.f;

    // this lane's 16-byte column group of a tile row (clamped inside the matrix: out-of-range columns are never stored)
    int ncol = n0 + lane * 4;
    if (ncol > N - 4) ncol = N - 4;
    const float *xrow = X + ncol;
    auto issue_tile = [&](int k0, int buf) {
#pragma unroll
        for (int i = 0; i < CT / 4; ++i) {
            const int c = w * (CT / 4) + i;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(xrow + (size_t)(k0 + c) * N),
                                             (__attribute__((address_space(3))) void *)(xs + (buf * CT + c) * NCOL), 16, 0, 0);
        }
    };
    float4 a_cur[NQ][2], a_nxt[NQ][2];
    auto load_a = [&](int k0, float4 (&a)[NQ][2]) {
        const float4 *wk = W + ((size_t)(k0 / 8) * Mpad + m_wave + j) * 2 + h;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int ot = 0; ot < 2; ++ot) a[q][ot] = wk[((size_t)q * Mpad + ot * 32) * 2];
    };

    issue_tile(0, 0);
    load_a(0, a_cur);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < K; k0 += CT) {
        const bool more = k0 + CT < K;
        if (more) {
            issue_tile(k0 + CT, buf ^ 1);
            load_a(k0 + CT, a_nxt);
        }
        const float *xt = xs + buf * (CT * NCOL) + ncol_wave + j;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float bv[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) bv[t] = xt[(q * 8 + r * 2 + h) * NCOL + t * 32];
#pragma unroll
                for (int ot = 0; ot < 2; ++ot) {
                    const float av = r == 0 ? a_cur[q][ot].x : r == 1 ? a_cur[q][ot].y : r == 2 ? a_cur[q][ot].z : a_cur[q][ot].w;
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[ot][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[t], acc[ot][t], 0, 0, 0);
                }
            }
        }
        if (more) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's tile rows have landed in LDS
#pragma unroll
            for (int q = 0; q < NQ; ++q) { a_cur[q][0] = a_nxt[q][0]; a_cur[q][1] = a_nxt[q][1]; }
        }
        __syncthreads();
        buf ^= 1;
    }
    float *__restrict__ Out = p.Out[irr];
    const float *__restrict__ Add = p.Add[irr];
#pragma unroll
    for (int ot = 0; ot < 2; ++ot)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int n = n0 + ncol_wave + t * 32 + j;
            if (n >= N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m_wave + ot * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < M) {
                    float o = acc[ot][t][r];
                    if (Add) o += Add[(size_t)m * N + n];
                    Out[(size_t)m * N + n] = o;
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------
// f32-accurate GEMM on the bf16 matrix cores ("3 x bf16 split"): every f32 operand is the exact sum of three bf16 pieces
// (8 + 8 + 8 significant bits); the six cross products of order <= 4, (1,1) (1,2) (2,1) (1,3) (2,2) (3,1), are accumulated in f32.
// The dropped products are below 2^-24 relative, i.e. at the f32 rounding level (measured error equals a plain f32 GEMM's, see
// tests), while six bf16 MFMAs (32 cycles, K=16 each) replace eight f32 MFMAs (64 cycles, K=2 each): 2.67x fewer matrix-core cycles.
// Weights are split and packed once on the host ([split][k/16][m][k-half][8] bf16 = one 16-byte load per fragment); activations are
// split on the fly from the f32 LDS tile (v_cvt_pk_bf16_f32, round-to-nearest-even), which the VALU does under the MFMAs' shadow.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct GemmSplitDescs {
    const float *X[NIRR];         // NP = 3: float32 coefficients.  NP = 2: fp16 hi | lo << 16 pairs written by ft_nonlin (same 4-byte pitch)
    float *Out[NIRR];
    const float *Add[NIRR];
    const float *xbound;          // NP = 2: per-keypoint bound [Bp] the packed activations were scaled with (column scale 2^bound_exp)
    int w_exp;                    // NP = 2: the weights were scaled by 2^w_exp before the fp16 split
    const float *nb_u, *nb_v;     // NP = 2, optional [O]: the next nonlinearity's bound  |FT(act(IFT(T)))| <= max_{o,q} (u_o |T_oq| + v_o)
    float *out_bound;             // NP = 2, optional [Bp] (zeroed by the caller): receives that bound per keypoint (atomic max)
    int O;                        // output channels (row m of irrep rho = j * O + o)
    const void *W[NIRR];          // [NP][K/16][2][Mpad] 16-byte fragments
    int K[NIRR], M[NIRR], Mpad[NIRR], N[NIRR];
};

// Block scale of the fp16 x 2 operand split: e with bound * 2^e < 2^14 (bound = f * 2^ex, f in [0.5, 1)).  A pure function of the
// keypoint's own bound, so the producer (ft_nonlin) and the consumer (the GEMM's epilogue) derive the same exponent independently.
__device__ __forceinline__ int bound_exp(float mx) {
    int e = 0;
    if (mx > 0.f && mx < __builtin_inff()) { int ex; (void)frexpf(mx, &ex); e = 14 - ex; }
    return e > 100 ? 100 : (e < -100 ? -100 : e);
}
__constant__ int kIrrDim[NIRR] = {1, 3, 3, 4, 5};
// keypoint of GEMM column n of an irrep of dimension d (columns are blocked by 32 keypoints: n = (b/32)*32d + i*32 + b%32)
__device__ __forceinline__ int column_keypoint(int n, int d) { return ((n >> 5) / d) * 32 + (n & 31); }

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

// fp16 x 2: hi = fp16(v * scale), lo = fp16(v * scale - hi)   (round-to-nearest-even conversions, the remainder is exact in f32)
__device__ __forceinline__ void split2(const float (&v)[8], float scale, f16x8 &hi, f16x8 &lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = v[e] * scale;
        const _Float16 h1 = (_Float16)x;
        hi[e] = h1; lo[e] = (_Float16)(x - (float)h1);
    }
}

// Which column of the wave's 128 the accumulator block t holds in lane j (the MFMA B operand's lane):
//   INTERLEAVED (round 3): 4 j + t -- a lane's four blocks are four ADJACENT output columns, so the epilogue stores (and reads the residual)
//   16 bytes per instruction: 32 global_store_dwordx4 per wave instead of 128 global_store_dword (the epilogue is issue-bound: it took 8-12 k
//   cycles per 256 x 256 tile, 5.4 % of the kernel).  The B-fragment LDS slot of column n makes both sides conflict-free: block t's 32
//   columns sit in 32 consecutive slots rotated by 4 t (16-lane ds_read_b128 groups read 16 consecutive slots; the 8-lane ds_write_b128
//   groups of the staging -- 8 consecutive columns = 4 blocks x 2 lanes -- hit bank quads q, q+4, q+8, q+12, q+1, ...: all distinct).
//   BLOCKED (rounds 1-2): t * 32 + j, slot n ^ ((n >> 3) & 1).
//   PAIRED (the LDS-DMA kernel): t * 32 + j / 2 + 16 (j % 2) -- the order in which ft_nonlin's half-block layout holds a 32-column block
//   (16-bit position e = column e / 2 + 16 (e % 2): its 32-bit words pair columns w and w + 16, see ft_nonlin_kernel's store).
// The arithmetic per output element is the same either way (same K order, same MFMA sequence): results are bitwise identical.
enum : int { COLS_BLOCKED = 0, COLS_INTERLEAVED = 1, COLS_PAIRED = 2 };
template <int MODE>
__device__ __forceinline__ int wave_col(int t, int j) { return MODE == COLS_INTERLEAVED ? 4 * j + t : MODE == COLS_PAIRED ? t * 32 + (j >> 1) + 16 * (j & 1) : t * 32 + j; }
template <int MODE>
__device__ __forceinline__ int col_slot(int n /* column inside the 256-column tile */) {
    static_assert(MODE != COLS_PAIRED, "the LDS-DMA kernel does not stage through fragment slots");
    if constexpr (MODE == COLS_BLOCKED) return n ^ ((n >> 3) & 1);
    const int t = n & 3, q = (n & 127) >> 2;
    return (n & ~127) | (t * 32 + ((q + 4 * t) & 31));
}

// Epilogue shared by the fp16 x 2 / bf16 x 3 GEMM kernels: rescale, optional residual, store, optional bound propagation.
template <int NP, int WO, int COLS = COLS_INTERLEAVED>
__device__ __forceinline__ void gemm_split_epilogue(const GemmSplitDescs &p, int irr, int mt, int n0, int wo, int ncol_wave, f32x16 (&acc)[2][4],
                                                    char *smem) {
    constexpr int NCOL = 256, OT = WO * 64, NT = WO * 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int j = lane & 31, h = lane >> 5;
    const unsigned jc = (unsigned)wave_col<COLS == COLS_PAIRED ? COLS_PAIRED : COLS_BLOCKED>(0, j);      // this lane's column inside a 32-column block (scalar-store modes)
    const int M = p.M[irr], N = p.N[irr];
    const int dirr = kIrrDim[irr];
    float *__restrict__ Out = p.Out[irr];
    const float *__restrict__ Add = p.Add[irr];
    // NP = 2: every column carries its keypoint's own power-of-two scale (undone here together with the weights' 2^w_exp), and the
    // epilogue can emit the bound the NEXT transform needs to split its output: max over this tile's rows of u_o |T| + v_o per column,
    // merged per keypoint with an atomic max (order-independent, hence deterministic).
    float oscale[4] = {1.f, 1.f, 1.f, 1.f};
    float bmax[4] = {0.f, 0.f, 0.f, 0.f};
    const bool want_bound = NP == 2 && p.out_bound != nullptr;
    float *su = reinterpret_cast<float *>(smem), *sv = su + OT;
    unsigned *cm = reinterpret_cast<unsigned *>(sv + OT);
    if constexpr (NP == 2) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int n = n0 + ncol_wave + wave_col<COLS>(t, j);
            if (n < N) oscale[t] = ldexpf(1.f, -(bound_exp(p.xbound[column_keypoint(n, dirr)]) + p.w_exp));
        }
        if (want_bound) {                                        // (the LDS tiles are dead: the loop ended with a barrier)
            for (int i = tid; i < OT; i += NT) {
                const int m = mt * OT + i;
                su[i] = m < M ? p.nb_u[m % p.O] : 0.f; sv[i] = m < M ? p.nb_v[m % p.O] : 0.f;
            }
            for (int i = tid; i < NCOL; i += NT) cm[i] = 0u;
            __syncthreads();
        }
    }
    // Interior tiles (every tile of GF's and ET's layers when B % 256 == 0): straight-line stores from one uniform base pointer with
    // 32-bit lane offsets, the four (residual, bound) combinations as separate branch-free bodies -- the generic body below spends ~200
    // cycles per element on its per-element range / option branches (measured: 27.6 k cycles per workgroup, 12 % of the d = 5 tile).
    const bool interior = (mt + 1) * OT <= M && n0 + NCOL <= N;
    if (interior) {
        const size_t base = (size_t)(mt * OT + wo * 64) * N + n0 + ncol_wave;
        float *__restrict__ ob = Out + base;
        const float *__restrict__ ab = Add ? Add + base : nullptr;
        const unsigned un = (unsigned)N;
        auto body = [&](auto has_add, auto has_bound) {
#pragma unroll
            for (int ot = 0; ot < 2; ++ot)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {                 // four accumulator rows at a time: their residual loads go out together
                    float res[4][4];
                    if constexpr (decltype(has_add)::value) {
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) {
                            const unsigned roff = (unsigned)(ot * 32 + rr + 8 * rq + 4 * h) * un;
                            if constexpr (COLS == COLS_INTERLEAVED) {
                                const float4 q = *reinterpret_cast<const float4 *>(ab + roff + 4u * (unsigned)j);
                                res[rr][0] = q.x; res[rr][1] = q.y; res[rr][2] = q.z; res[rr][3] = q.w;
                            } else {
#pragma unroll
                                for (int t = 0; t < 4; ++t) res[rr][t] = ab[roff + jc + t * 32];
                            }
                        }
                    }
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int r = rq * 4 + rr;
                        const int rl = ot * 32 + rr + 8 * rq + 4 * h;          // row inside the wave's 64
                        float ur = 0.f, vr = 0.f;
                        if constexpr (decltype(has_bound)::value) { ur = su[wo * 64 + rl]; vr = sv[wo * 64 + rl]; }
                        float o4[4];
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            float o = acc[ot][t][r];
                            if constexpr (NP == 2) o *= oscale[t];
                            if constexpr (decltype(has_add)::value) o += res[rr][t];
                            o4[t] = o;
                            if constexpr (decltype(has_bound)::value) bmax[t] = fmaxf(bmax[t], fmaf(ur, fabsf(o), vr));
                        }
                        if constexpr (COLS == COLS_INTERLEAVED) {
                            *reinterpret_cast<float4 *>(ob + (unsigned)rl * un + 4u * (unsigned)j) = make_float4(o4[0], o4[1], o4[2], o4[3]);
                        } else {
#pragma unroll
                            for (int t = 0; t < 4; ++t) ob[(unsigned)rl * un + jc + t * 32] = o4[t];
                        }
                    }
                }
        };
        if (Add) { if (want_bound) body(std::true_type{}, std::true_type{}); else body(std::true_type{}, std::false_type{}); }
        else     { if (want_bound) body(std::false_type{}, std::true_type{}); else body(std::false_type{}, std::false_type{}); }
    } else {
#pragma unroll
        for (int ot = 0; ot < 2; ++ot)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wo * 64 + ot * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int m = mt * OT + row;
                if (m >= M) continue;
                float ur = 0.f, vr = 0.f;
                if constexpr (NP == 2) { if (want_bound) { ur = su[row]; vr = sv[row]; } }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int n = n0 + ncol_wave + wave_col<COLS>(t, j);
                    if (n >= N) continue;
                    float o = acc[ot][t][r];
                    if constexpr (NP == 2) o *= oscale[t];
                    if (Add) o += Add[(size_t)m * N + n];
                    Out[(size_t)m * N + n] = o;
                    if constexpr (NP == 2) { if (want_bound) bmax[t] = fmaxf(bmax[t], fmaf(ur, fabsf(o), vr)); }
                }
            }
    }
    if constexpr (NP == 2) {
        if (want_bound) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float bm = fmaxf(bmax[t], __shfl_xor(bmax[t], 32));
                if (h == 0) atomicMax(cm + ncol_wave + wave_col<COLS>(t, j), __float_as_uint(bm));      // non-negative floats order like their bit patterns
            }
            __syncthreads();
            for (int i = tid; i < NCOL; i += NT) {
                const int n = n0 + i;
                if (n < N) atomicMax(reinterpret_cast<unsigned *>(p.out_bound) + column_keypoint(n, dirr), cm[i]);
            }
        }
    }
}

// Data flow of one K16 step: the packed weight fragments of the NEXT step (3 planes x 2 k-octets x 128 rows = 12 KiB) go
// global -> LDS with LDS-DMA (no VGPR staging); every thread owns an 8(k) x 2(n) patch of the f32 activations, loaded two steps
// ahead into registers, split ONCE (v_cvt_pk_bf16_f32) and written as three 16-byte k-octets into LDS in B-fragment order, so the
// MFMA phase reads every bf16 fragment with one conflict-free ds_read_b128.  The ~110 VALU instructions of the conversion are
// interleaved with the 48 MFMAs of the step (sched_group_barrier), i.e. they issue in the matrix pipe's shadow.  Both LDS areas
// are double-buffered: one barrier per step.
//
// NP = 3: bf16 x 3 (24 significant bits per operand, six cross products).  NP = 2: fp16 x 2 with power-of-two block scaling -- the
// activations are scaled by 2^e with e chosen from the tensor's absolute maximum (tracked by the producer kernel) so that |x| <= 2^14,
// the weights by their own 2^w_exp; hi = fp16(x), lo = fp16(x - hi) keep 22 significant bits of every operand in the top 18 binades
// below the maximum (smaller values keep an ABSOLUTE error below 2^-39 of the maximum), three cross products hi.hi, hi.lo, lo.hi, the
// accumulator is rescaled by the exact 2^-(e + w_exp) in the epilogue.  Half the matrix-core work of NP = 3.
// WO = number of 64-row wave groups: WO = 2 -> 128 x 256 workgroup tile, 4 waves (two workgroups per CU); WO = 4 -> 256 x 256 tile, 8 waves
// (one workgroup per CU, the same 2 waves per SIMD): a third less L2 -> CU traffic per MFMA and half the conversion work per MFMA,
// because one converted activation tile now feeds 256 output rows.
// BIG: a name tag only (C * O = 256 * 512, GF's two dominant layers), so that profiler output can be filtered to exactly the launch
// population bench.py prices in `roofline` -- the instantiations are otherwise identical.
template <int CT, int NP, int WO, int BIG, int PIPE = 0>
__global__ __launch_bounds__(WO * 128, 2) void irrep_gemm_split_kernel(GemmSplitDescs p, const int *__restrict__ tiles) {
    using frag = typename std::conditional<NP == 3, bf16x8, f16x8>::type;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NCOL = 256, OT = WO * 64, NT = WO * 128, CPT = 512 / NT;      // CPT: activation columns converted per thread
    constexpr int XBUF = NP * 2 * NCOL, ABUF = NP * 2 * OT;       // fragments (16 B) per buffer
    frag *xs = reinterpret_cast<frag *>(smem);                   // [2 buf][NP split][2 k-octet][256 n (swizzled)]
    frag *as = xs + 2 * XBUF;                                    // [2 buf][NP split][2 k-octet][128 m]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int irr = tiles[blockIdx.x * 3], mt = tiles[blockIdx.x * 3 + 1], nt = tiles[blockIdx.x * 3 + 2];
    if (irr < 0) return;
    const float *__restrict__ X = p.X[irr];
    const frag *__restrict__ W = reinterpret_cast<const frag *>(p.W[irr]);
    const int K = p.K[irr], Mpad = p.Mpad[irr], N = p.N[irr];
    const size_t split_stride = (size_t)(K / 16) * 2 * Mpad;     // in 16-byte fragments
    const int wo = w % WO, wb = w / WO;
    const int n0 = nt * NCOL;
    const int ncol_wave = wb * 128;

    f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][t][r] = 0.f;

    // staging patch of this thread: columns n0 + 2*pp, +1 (clamped inside the matrix: out-of-range columns are never stored),
    // k-octet po of the step.  Column n lives in fragment slot col_slot(n) (see wave_col / col_slot above): conflict-free for the 8-lane
    // groups of ds_write_b128 (8 neighbouring columns, or the even / odd columns of 8 neighbouring threads) and for the 16-lane groups of
    // ds_read_b128.
    const int pp = tid % (NCOL / CPT), po = tid / (NCOL / CPT);
    int ncol = n0 + CPT * pp;
    if (ncol > N - CPT) ncol = N - CPT;
    const float *xcol = X + ncol + (size_t)(8 * po) * N;
    const int nsteps = K / 16;                                   // even (C % 32 == 0)
    typedef float xpatch __attribute__((ext_vector_type(CPT)));
    auto load_x = [&](int kstep, xpatch (&xr)[8]) {
        const float *q = xcol + (size_t)(kstep < nsteps ? kstep : nsteps - 1) * 16 * N;
#pragma unroll
        for (int e = 0; e < 8; ++e) xr[e] = *reinterpret_cast<const xpatch *>(q + (size_t)e * N);
    };
    int slot[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) slot[c] = po * NCOL + col_slot<COLS_INTERLEAVED>(CPT * pp + c);
    auto convert_store = [&](int buf, const xpatch (&xr)[8]) {
        frag *dst = xs + buf * XBUF;
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = xr[e][c];
            if constexpr (NP == 3) {
                bf16x8 b1, b2, b3;
                split3(v, b1, b2, b3);
                dst[slot[c]] = b1; dst[2 * NCOL + slot[c]] = b2; dst[4 * NCOL + slot[c]] = b3;
            } else {
                // the activations arrive split: word e = fp16 hi | fp16 lo << 16 of k = 8 po + e; a k-octet is four byte permutes per plane
                u32x4 H, L;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned a = __float_as_uint(v[2 * i]), b = __float_as_uint(v[2 * i + 1]);
                    H[i] = __builtin_amdgcn_perm(b, a, 0x05040100u);
                    L[i] = __builtin_amdgcn_perm(b, a, 0x07060302u);
                }
                dst[slot[c]] = __builtin_bit_cast(f16x8, H); dst[2 * NCOL + slot[c]] = __builtin_bit_cast(f16x8, L);
            }
        }
    };
    // weight fragments of a step: per split plane [2 k-octets][OT rows] = NT fragments, one per thread, 64 consecutive per wave
    const frag *wsrc = W + (size_t)(tid / OT) * Mpad + mt * OT + (tid % OT);
    auto issue_a = [&](int kstep, int buf) {
        const frag *q = wsrc + (size_t)(kstep < nsteps ? kstep : nsteps - 1) * 2 * Mpad;
#pragma unroll
        for (int sp = 0; sp < NP; ++sp)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(q + sp * split_stride),
                                             (__attribute__((address_space(3))) void *)(as + buf * ABUF + sp * (2 * OT) + w * 64), 16, 0, 0);
    };
    // fragment slots this lane reads in the MFMA phase
    int xslot[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) xslot[t] = h * NCOL + col_slot<COLS_INTERLEAVED>(ncol_wave + wave_col<COLS_INTERLEAVED>(t, j));
    const int aslot = h * OT + wo * 64 + j;

    auto step = [&](int ks, int buf, xpatch (&xr_load)[8], const xpatch (&xr_use)[8]) {
        issue_a(ks + 1, buf ^ 1);                                 // land under the MFMAs of this step
        load_x(ks + 2, xr_load);                                  // consumed during the next step
        __builtin_amdgcn_sched_barrier(0);
        const frag *xt = xs + buf * XBUF;
        const frag *at = as + buf * ABUF + aslot;
        frag a[2][NP];
#pragma unroll
        for (int ot = 0; ot < 2; ++ot)
#pragma unroll
            for (int sp = 0; sp < NP; ++sp) a[ot][sp] = at[sp * (2 * OT) + ot * 32];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x16 c0 = acc[0][t], c1 = acc[1][t];
            if constexpr (NP == 3) {
                const bf16x8 b1 = xt[xslot[t]], b2 = xt[2 * NCOL + xslot[t]], b3 = xt[4 * NCOL + xslot[t]];
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][2], b1, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][2], b1, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][1], b2, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][1], b2, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][0], b3, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][0], b3, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][1], b1, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][1], b1, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][0], b2, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][0], b2, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][0], b1, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][0], b1, c1, 0, 0, 0);
            } else {
                const f16x8 bh = xt[xslot[t]], bl = xt[2 * NCOL + xslot[t]];
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][1], bh, c0, 0, 0, 0);       // lo.hi
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][1], bh, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][0], bl, c0, 0, 0, 0);       // hi.lo
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][0], bl, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][0], bh, c0, 0, 0, 0);       // hi.hi
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][0], bh, c1, 0, 0, 0);
            }
            acc[0][t] = c0; acc[1][t] = c1;
        }
        convert_store(buf ^ 1, xr_use);                           // the next step's activations, split under the MFMAs' shadow
#pragma unroll
        for (int i = 0; i < (NP == 3 ? 48 : 24); ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // one MFMA ...
            __builtin_amdgcn_sched_group_barrier(0x002, NP == 3 ? 3 : 4, 0);    // ... then a few VALU instructions in its shadow
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // next weights (LDS-DMA) and the staged patch have landed
        __syncthreads();
    };

    xpatch xr0[8], xr1[8];
    if constexpr (PIPE == 1 && NP == 2) {
        // Fragment-pipelined loop: every MFMA of step k reads fragment REGISTERS that were filled during step k - 1, so no MFMA waits on
        // an LDS read issued after a barrier.  Per step: LDS-DMA of the weights of step k + 2 and the staging store of the activations of
        // step k + 2 go into stage k % 2 (all of whose fragments were read during step k - 1); the fragments of step k + 1 are read from
        // stage (k + 1) % 2 -- the four weight fragments into the alternate register set, the activation fragments of column block t
        // into the registers the MFMAs of block t have just released.  Two LDS stages, one barrier per step, +16 VGPRs.
        frag aA[2][2], aB[2][2], b[4][2];
        auto read_a = [&](int buf, frag (&a)[2][2]) {
            const frag *at = as + buf * ABUF + aslot;
#pragma unroll
            for (int ot = 0; ot < 2; ++ot)
#pragma unroll
                for (int sp = 0; sp < 2; ++sp) a[ot][sp] = at[sp * (2 * OT) + ot * 32];
        };
        auto read_b = [&](int buf, int t) {
            const frag *xt = xs + buf * XBUF;
            b[t][0] = xt[xslot[t]]; b[t][1] = xt[2 * NCOL + xslot[t]];
        };
        // The activation loads are issued from inline assembly: the compiler's wait-count pass then neither sees them nor widens the
        // waits of this loop to vmcnt(0) (it loses count across the raw barrier); wait_x is the explicit, counted wait -- the patch
        // registers are its in/out operands, so no use can be scheduled above it.
        // (a patch word = the thread's CPT columns of one k row: 4 bytes for the 8-wave tile, 8 bytes for the 4-wave tile)
        typedef typename std::conditional<CPT == 1, float, unsigned long long>::type xword;
        auto load_one = [&](xword &dst, const float *src) {
            if constexpr (CPT == 1) asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(src) : "memory");
            else asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(dst) : "v"(src) : "memory");
        };
        auto load_xa = [&](int kstep, xword (&xr)[8]) {
            const float *q = xcol + (size_t)(kstep < nsteps ? kstep : nsteps - 1) * 16 * N;
#pragma unroll
            for (int e = 0; e < 8; ++e) load_one(xr[e], q + (size_t)e * N);
        };
        auto word_of = [&](const xword &v, int c) -> unsigned {
            if constexpr (CPT == 1) return __float_as_uint(v); else return (unsigned)(v >> (32 * c));
        };
#define ROREG_WAIT_X(cnt, xr) asm volatile("s_waitcnt vmcnt(" #cnt ")" : "+v"(xr[0]), "+v"(xr[1]), "+v"(xr[2]), "+v"(xr[3]), "+v"(xr[4]), "+v"(xr[5]), "+v"(xr[6]), "+v"(xr[7]) :: "memory")
        // word e = fp16 hi | fp16 lo << 16 of k = 8 po + e -> the k-octet of plane `plane` (0 = hi, 1 = lo) of the thread's column c
        auto store_plane = [&](frag *dst, const xword (&v)[8], int c, int plane) {
            u32x4 Q;
#pragma unroll
            for (int i = 0; i < 4; ++i) Q[i] = __builtin_amdgcn_perm(word_of(v[2 * i + 1], c), word_of(v[2 * i], c), plane ? 0x07060302u : 0x05040100u);
            dst[plane * 2 * NCOL + slot[c]] = __builtin_bit_cast(f16x8, Q);
        };
        auto store_x = [&](int buf, const xword (&v)[8]) {
#pragma unroll
            for (int c = 0; c < CPT; ++c) { store_plane(xs + buf * XBUF, v, c, 0); store_plane(xs + buf * XBUF, v, c, 1); }
        };
        // One step = 24 MFMAs per wave in 12 pairs; every memory operation of the step sits BETWEEN two pairs (sched_barrier pins the order),
        // never in a burst at the step's head: a wave that is queueing its ten VMEM instructions cannot issue MFMAs (in-order issue), and
        // with all eight waves doing so right after the barrier the matrix pipes idled ~1000 cycles per step (measured: 2450 cycles per
        // step with the burst, 1820 with no memory operations at all, 1536 = the MFMAs alone).
        auto step2 = [&](int ks, int buf, xword (&xr_load)[8], xword (&xr_use)[8], const frag (&a)[2][2], frag (&an)[2][2]) {
            const int kw = ks + 2 < nsteps ? ks + 2 : nsteps - 1, kx = ks + 3 < nsteps ? ks + 3 : nsteps - 1;
            const frag *wq = wsrc + (size_t)kw * 2 * Mpad;
            const float *xq = xcol + (size_t)kx * 16 * N;
            const frag *xn = xs + (buf ^ 1) * XBUF, *aq = as + (buf ^ 1) * ABUF + aslot;
            frag *xd = xs + buf * XBUF;
            auto mm = [&](int t, int i) {                        // pair i of column block t: 0 = lo.hi, 1 = hi.lo, 2 = hi.hi (both row blocks)
                const f16x8 bb = b[t][i == 1 ? 1 : 0];
                const int ai = i == 0 ? 1 : 0;
                acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][ai], bb, acc[0][t], 0, 0, 0);
                acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][ai], bb, acc[1][t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            };
            auto dma = [&](int sp) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wq + sp * split_stride),
                                                 (__attribute__((address_space(3))) void *)(as + buf * ABUF + sp * (2 * OT) + w * 64), 16, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            };
            auto ldx = [&](int e0, int e1) {
#pragma unroll
                for (int e = e0; e < e1; ++e) load_one(xr_load[e], xq + (size_t)e * N);
                __builtin_amdgcn_sched_barrier(0);
            };
            auto rb = [&](int t) { b[t][0] = xn[xslot[t]]; b[t][1] = xn[2 * NCOL + xslot[t]]; __builtin_amdgcn_sched_barrier(0); };
            auto ra = [&](int ot) { an[ot][0] = aq[ot * 32]; an[ot][1] = aq[2 * OT + ot * 32]; __builtin_amdgcn_sched_barrier(0); };
            __builtin_amdgcn_sched_barrier(0);
            mm(0, 0); dma(0); mm(0, 1); dma(1); mm(0, 2); rb(0);
            mm(1, 0); ra(0);  mm(1, 1); ra(1);  mm(1, 2); rb(1);
            mm(2, 0);
            ROREG_WAIT_X(2, xr_use);                             // the patch loaded during the previous step: only this step's two LDS-DMA pieces are newer
            store_plane(xd, xr_use, 0, 0);
            if constexpr (CPT == 2) store_plane(xd, xr_use, 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            mm(2, 1);
            store_plane(xd, xr_use, 0, 1);
            if constexpr (CPT == 2) store_plane(xd, xr_use, 1, 1);
            __builtin_amdgcn_sched_barrier(0);
            mm(2, 2); rb(2);
            mm(3, 0); ldx(0, 3); mm(3, 1); ldx(3, 6); mm(3, 2); ldx(6, 8); rb(3);
            // the two LDS-DMA pieces are the oldest of this step's ten VMEM operations: vmcnt(8) = they have landed (the activation
            // loads stay in flight across the barrier); lgkmcnt(0) = this wave's staging stores and fragment reads are done
            asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        };
        xword xa0[8], xa1[8];
        load_xa(0, xa0); issue_a(0, 0);
        load_xa(1, xa1); issue_a(1, 1);
        ROREG_WAIT_X(0, xa0);
        ROREG_WAIT_X(0, xa1);
        store_x(0, xa0);
        load_xa(2, xa0);
        store_x(1, xa1);
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        read_a(0, aA);
#pragma unroll
        for (int t = 0; t < 4; ++t) read_b(0, t);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");        // stage 0 is overwritten from step 0 on
        for (int ks = 0; ks < nsteps; ks += 2) {
            step2(ks, 0, xa1, xa0, aA, aB);
            step2(ks + 1, 1, xa0, xa1, aB, aA);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the clamped look-ahead loads still target the patch registers
#undef ROREG_WAIT_X
    } else {
        load_x(0, xr0);
        issue_a(0, 0);
        load_x(1, xr1);
        convert_store(0, xr0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int ks = 0; ks < nsteps; ks += 2) {
            step(ks, 0, xr0, xr1);
            step(ks + 1, 1, xr1, xr0);
        }
    }
    gemm_split_epilogue<NP, WO>(p, irr, mt, n0, wo, ncol_wave, acc, smem);
}


// ---------------------------------------------------------------------------------------------------------------
// fp16 x 2 GEMM, 256 x 256 tile, 8 waves, with the ACTIVATIONS delivered by LDS-DMA (round 3; the review's "X by LDS-DMA in fragment order").
// ft_nonlin writes the operand in HALF-BLOCK layout (NonlinParams::out_planes): the row pitch and the 32-column blocks of the word layout, but
// inside a block the 32 fp16 hi values first, then the 32 lo values, each half in the order 0, 16, 1, 17, ... (the transform then stores
// consecutive words from consecutive lanes, as in the word layout, and a GEMM tile still reads 1 KB per k row; the column an MFMA lane holds
// is wave_col<COLS_PAIRED>).  A 16-byte LDS-DMA piece is then 8 consecutive columns of one k row of one plane; the pieces of a K16 step are laid out
// in LDS as [k / 4][column panel of 16][4 rows][16 columns] (a 128-byte block per (k quad, panel)), which is exactly what gfx950's transposing
// ds_read_b64_tr_b16 turns into MFMA B fragments: a 16-lane group reads one block and every lane receives the 4 k values of its column; two
// such reads (k quads 2h and 2h + 1) are the lane's 8-value fragment.  Both sides are conflict-free (the DMA writes lane-linear; the four
// groups of a read cover 2 x 256 contiguous bytes).  No activation passes through a VGPR: per K16 step and wave 2 + 2 DMA instructions
// instead of 2 + 8 loads, no byte permutes, no ds_write.  The weights, the MFMA order per accumulator and the epilogue are those of
// irrep_gemm_split_kernel<.., 2, 4, .., 1> with the BLOCKED column mapping, so the results are bitwise the same.
// Stages: weights double-buffered as before (DMA one step ahead); activations in THREE stages, DMA two steps ahead (they stream from
// HBM, the weights from L2): at the end of step k `s_waitcnt vmcnt(2)` leaves only the two activation pieces of step k + 3 in flight.
template <int BIG>
__global__ __launch_bounds__(512, 2) void irrep_gemm_xdma_kernel(GemmSplitDescs p, const int *__restrict__ tiles) {
    using frag = f16x8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NCOL = 256, OT = 256;
    constexpr int XIMG = 16 * NCOL * 2, XSTAGE = 2 * XIMG;       // bytes: one plane's K16 x 256 image; both planes
    constexpr int ABUF = 2 * 2 * OT;                             // weight fragments per stage
    char *xs = smem;                                             // [3 stages][2 planes][XIMG]
    frag *as = reinterpret_cast<frag *>(smem + 3 * XSTAGE);      // [2 stages][2 planes][2 k-octets][256 m]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int irr = tiles[blockIdx.x * 3], mt = tiles[blockIdx.x * 3 + 1], nt = tiles[blockIdx.x * 3 + 2];
    if (irr < 0) return;
    const frag *__restrict__ W = reinterpret_cast<const frag *>(p.W[irr]);
    const int K = p.K[irr], Mpad = p.Mpad[irr], N = p.N[irr];
    const size_t split_stride = (size_t)(K / 16) * 2 * Mpad;     // in 16-byte fragments
    const int wo = w % 4, wb = w / 4;
    const int n0 = nt * NCOL;
    const int ncol_wave = wb * 128;
    const int nsteps = K / 16;                                   // even (C % 32 == 0)

    f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][t][r] = 0.f;

    // ---- activation DMA: wave w issues pieces m = 2w, 2w + 1 of the 16 KB step (8 KB per plane = 8 instructions of 64 pieces) ----
    const int x_plane = w >> 2;
    const char *xsrc[2];
    int xdst[2];
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2) {
        const int mm = (2 * w + i2) & 7;
        const int pi = mm * 64 + lane;                           // 16-byte piece inside the plane's image: block beta = pi / 8
        const int beta = pi >> 3, r = (pi & 7) >> 1, half = pi & 1;
        const int k = 4 * (beta >> 4) + r;
        int col = n0 + 16 * (beta & 15) + 8 * half;
        if (col > N - 8) col = N - 8;                            // (columns beyond N are never stored)
        xsrc[i2] = reinterpret_cast<const char *>(p.X[irr]) + (size_t)k * N * 4 + (col >> 5) * 128 + x_plane * 64 + (col & 31) * 2;
        xdst[i2] = x_plane * XIMG + mm * 1024;
    }
    const size_t xstep = (size_t)16 * N * 4;                     // bytes between K16 steps
    auto dma_x = [&](int i2, int kstep, int stage) {
        const char *q = xsrc[i2] + (size_t)(kstep < nsteps ? kstep : nsteps - 1) * xstep;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)q,
                                         (__attribute__((address_space(3))) void *)(xs + stage * XSTAGE + xdst[i2]), 16, 0, 0);
    };
    // ---- weight DMA, as in irrep_gemm_split_kernel ----
    const frag *wsrc = W + (size_t)(tid / OT) * Mpad + mt * OT + (tid % OT);
    auto dma_w = [&](int sp, int kstep, int buf) {
        const frag *q = wsrc + (size_t)(kstep < nsteps ? kstep : nsteps - 1) * 2 * Mpad;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(q + sp * split_stride),
                                         (__attribute__((address_space(3))) void *)(as + buf * ABUF + sp * (2 * OT) + w * 64), 16, 0, 0);
    };
    const int aslot = h * OT + wo * 64 + j;
    // transposing fragment read: this lane's address inside a stage's plane-0 image for column block t = 0, k quad 2h
    const unsigned xlane = (unsigned)(uintptr_t)xs + (unsigned)((2 * h * 16 + wb * 8 + ((lane >> 4) & 1)) * 128 + (lane & 15) * 8);
    frag aA[2][2], aB[2][2], b[4][2];
    auto read_a = [&](int buf, frag (&a)[2][2]) {
        const frag *at = as + buf * ABUF + aslot;
#pragma unroll
        for (int ot = 0; ot < 2; ++ot)
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) a[ot][sp] = at[sp * (2 * OT) + ot * 32];
    };
    // (inline assembly: through the builtin the compiler guards every transposing read with s_waitcnt vmcnt(0) -- it cannot tell the read from
    //  the LDS-DMA pieces in flight -- which would expose the DMA latency four times per step; the explicit waits below cover the real
    //  dependences: a fragment is read after the barrier that follows its pieces' landing, and used after the next end-of-step wait, whose
    //  asm statement takes the fragment registers as operands so that no use can be scheduled above it)
    auto read_b = [&](unsigned stage_off, auto tc) {
        constexpr int t = decltype(tc)::value;
        const unsigned a = xlane + stage_off;
        unsigned long long q[4];
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(q[0]) : "v"(a), "n"(t * 256));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(q[1]) : "v"(a), "n"(t * 256 + 2048));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(q[2]) : "v"(a), "n"(XIMG + t * 256));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(q[3]) : "v"(a), "n"(XIMG + t * 256 + 2048));
        struct Pair { unsigned long long lo, hi; };
        b[t][0] = __builtin_bit_cast(frag, Pair{q[0], q[1]});
        b[t][1] = __builtin_bit_cast(frag, Pair{q[2], q[3]});
    };
#define ROREG_WAIT_FRAGS(waits)                                                                                                          \
    asm volatile(waits : "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[2][0]), "+v"(b[2][1]), "+v"(b[3][0]), "+v"(b[3][1]) \
                 :: "memory")
    // One step = 24 MFMAs per wave in 12 pairs with the step's memory operations BETWEEN the pairs (see irrep_gemm_split_kernel's step2).
    auto step = [&](int ks, int wbuf, unsigned xs_next /* stage of step ks + 1 */, int xs_fill /* stage of step ks (free), receives step ks + 3 */,
                    const frag (&a)[2][2], frag (&an)[2][2]) {
        const frag *aq = as + (wbuf ^ 1) * ABUF + aslot;
        auto mm = [&](int t, int i) {                            // pair i of column block t: 0 = lo.hi, 1 = hi.lo, 2 = hi.hi (both row blocks)
            const f16x8 bb = b[t][i == 1 ? 1 : 0];
            const int ai = i == 0 ? 1 : 0;
            acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][ai], bb, acc[0][t], 0, 0, 0);
            acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][ai], bb, acc[1][t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        auto rb = [&](auto tc) { read_b(xs_next, tc); __builtin_amdgcn_sched_barrier(0); };
        auto ra = [&](int ot) { an[ot][0] = aq[ot * 32]; an[ot][1] = aq[2 * OT + ot * 32]; __builtin_amdgcn_sched_barrier(0); };
        using std::integral_constant;
        __builtin_amdgcn_sched_barrier(0);
        mm(0, 0); dma_w(0, ks + 2, wbuf); __builtin_amdgcn_sched_barrier(0);
        mm(0, 1); dma_w(1, ks + 2, wbuf); __builtin_amdgcn_sched_barrier(0);
        mm(0, 2); rb(integral_constant<int, 0>{});
        mm(1, 0); ra(0); mm(1, 1); ra(1); mm(1, 2); rb(integral_constant<int, 1>{});
        mm(2, 0); dma_x(0, ks + 3, xs_fill); __builtin_amdgcn_sched_barrier(0);
        mm(2, 1); dma_x(1, ks + 3, xs_fill); __builtin_amdgcn_sched_barrier(0);
        mm(2, 2); rb(integral_constant<int, 2>{});
        mm(3, 0); mm(3, 1); mm(3, 2); rb(integral_constant<int, 3>{});
        // in issue order the step's two activation pieces are the newest: vmcnt(2) = this step's weights and the previous step's
        // activations (needed by the next step's fragment reads) have landed; lgkmcnt(0) = this wave's fragment reads are done
        ROREG_WAIT_FRAGS("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier");
    };
    dma_x(0, 0, 0); dma_x(1, 0, 0); dma_x(0, 1, 1); dma_x(1, 1, 1); dma_x(0, 2, 2); dma_x(1, 2, 2);
    dma_w(0, 0, 0); dma_w(1, 0, 0); dma_w(0, 1, 1); dma_w(1, 1, 1);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    read_a(0, aA);
    read_b(0u, std::integral_constant<int, 0>{}); read_b(0u, std::integral_constant<int, 1>{});
    read_b(0u, std::integral_constant<int, 2>{}); read_b(0u, std::integral_constant<int, 3>{});
    ROREG_WAIT_FRAGS("s_waitcnt lgkmcnt(0)\n\ts_barrier");                      // stage 0 (weights and activations) is overwritten from step 0 on
    int cur = 0;                                                 // activation stage of step ks
    for (int ks = 0; ks < nsteps; ks += 2) {
        const int n1 = cur == 2 ? 0 : cur + 1, n2 = n1 == 2 ? 0 : n1 + 1;
        step(ks, 0, (unsigned)(n1 * XSTAGE), cur, aA, aB);
        step(ks + 1, 1, (unsigned)(n2 * XSTAGE), n1, aB, aA);
        cur = n2;
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");              // the clamped look-ahead pieces still target LDS the epilogue reuses
#undef ROREG_WAIT_FRAGS
    gemm_split_epilogue<2, 4, COLS_PAIRED>(p, irr, mt, n0, wo, ncol_wave, acc, smem);
}

template <int N, class F> __device__ __forceinline__ void static_for(F &&f);      // (defined with the transforms below)

// Epilogue of irrep_gemm_xdma16_kernel: accumulator block (rb, cb), register r of lane (j = l % 16, q = l / 16) = tile row
// wo * 64 + 16 rb + 4 q + r, position 16 cb + j of the wave's 128 columns (position p of a 32-column block = column p / 2 + 16 (p % 2)).
// A lane's 16 columns per row block are scattered (32-byte runs): stored from the accumulators directly the epilogue cost 2.5 x the
// other kernel's (2.6 ms of a 12.0 ms launch).  So every wave passes its tile through LDS, 16 rows x 128 columns at a time (the rings are
// dead): scalar writes in accumulator order, then 16-byte reads along the rows -- lane l % 32 owns FOUR ADJACENT columns, l / 32 the row
// parity -- and everything else happens on that side: rescale by the columns' (keypoints') and the weights' power of two, optional
// residual (16-byte loads), 16-byte stores (1 KB per instruction: two rows x 512 contiguous bytes), optional bound of the next
// transform (max over the rows of u_o |T| + v_o per column, merged per keypoint with atomic max: order-independent).
typedef float f32x4_t __attribute__((ext_vector_type(4)));
// What the epilogue needs from memory besides the residual, fetched at the START of the workgroup's life (round 5) -- the first DMA pieces'
// trip from HBM covers these loads, the values ride through the loop in six registers, and the epilogue begins without a memory round trip
// (it used to open with one: the column scales' bounds, then the rows' (u, v) of the next bound behind a barrier):
struct Epi16Pre {
    float osc[4];                 // 2^-(e(keypoint) + w_exp) of this lane's four adjacent columns (1 beyond N); until gemm_split_epilogue16_scales: the keypoints' bounds
    float u, v;                   // thread t < 256: (nb_u, nb_v) of tile row t (0 beyond M, or without bound propagation)
};
__device__ __forceinline__ Epi16Pre gemm_split_epilogue16_request(const GemmSplitDescs &p, int irr, int mt, int n0, int wb) {
    constexpr int OT = 256;
    const int tid = threadIdx.x, rl = tid & 31;
    const int N = p.N[irr], M = p.M[irr], dirr = kIrrDim[irr];
    const int ncol = n0 + wb * 128 + 4 * rl;                     // (N % 32 == 0: the four columns are inside or outside together; four adjacent
    Epi16Pre e;                                                  //  columns of a 32-column block = four adjacent keypoints: one 16-byte load)
    const f32x4_t xb = ncol < N ? *reinterpret_cast<const f32x4_t *>(p.xbound + column_keypoint(ncol, dirr)) : f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) e.osc[i] = xb[i];
    e.u = 0.f; e.v = 0.f;
    if (p.out_bound != nullptr && tid < OT) {
        const int m = mt * OT + tid;
        if (m < M) { e.u = p.nb_u[m % p.O]; e.v = p.nb_v[m % p.O]; }
    }
    return e;
}
__device__ __forceinline__ void gemm_split_epilogue16_scales(const GemmSplitDescs &p, Epi16Pre &e, bool col_ok) {
#pragma unroll
    for (int i = 0; i < 4; ++i) e.osc[i] = col_ok ? ldexpf(1.f, -(bound_exp(e.osc[i]) + p.w_exp)) : 1.f;
}
template <int NT = 512, int NCOL = 256>      // threads per workgroup, columns of its tile (256 x 256 with eight waves; 256 x 128 with four: wb = 0)
__device__ __forceinline__ void gemm_split_epilogue16(const GemmSplitDescs &p, int irr, int mt, int n0, int wo, int wb, f32x4_t (&acc)[4][8], char *smem,
                                                      const Epi16Pre &pre) {
    constexpr int OT = 256, P = 192;                            // P: row pitch of a wave's LDS tile in floats (a multiple of the 64 banks: the row's bank shift is rowpad alone)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int M = p.M[irr], N = p.N[irr];
    const int dirr = kIrrDim[irr];
    float *__restrict__ Out = p.Out[irr];
    const float *__restrict__ Add = p.Add[irr];
    const bool want_bound = p.out_bound != nullptr;
    float *su = reinterpret_cast<float *>(smem), *sv = su + OT;
    unsigned *cm = reinterpret_cast<unsigned *>(sv + OT);
    float *buf = reinterpret_cast<float *>(smem + 4096) + w * (16 * P);
    int cw[8];                                                   // write side: the column (inside the wave's 128) of block cb in this lane
#pragma unroll
    for (int cb = 0; cb < 8; ++cb) {
        const int pp = (cb & 1) * 16 + j;
        cw[cb] = (cb >> 1) * 32 + (pp >> 1) + 16 * (pp & 1);
    }
    // a write instruction's 64 lanes: 16 columns {c .. c + 7, c + 16 .. c + 23} x the 4 row groups q -- shifted by 0, 8, 32, 40 floats per
    // group they cover the 64 banks exactly once
    const int rowpad_w = 8 * (q & 1) + 32 * (q >> 1);
    const int rl = lane & 31, rr = lane >> 5;                    // read side: columns 4 rl .. 4 rl + 3, row parity rr
    const int ncol = n0 + wb * 128 + 4 * rl;
    const bool col_ok = ncol < N;
    const int ncl = col_ok ? ncol : N - 4;                       // (clamped: the residual is always read from a valid address)
    float bm[4] = {0.f, 0.f, 0.f, 0.f};
    if (want_bound) {                                            // (the LDS rings are dead: the loop ended with a barrier)
        if (tid < OT) { su[tid] = pre.u; sv[tid] = pre.v; }
        for (int i = tid; i < NCOL; i += NT) cm[i] = 0u;
        __syncthreads();
    }
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
        // the row block's eight residual pieces are requested first and arrive under the LDS round trip (round 5; they used to be
        // 8 x (load, wait, add, store) per row block: the compiler drains vmcnt in front of a load's first use while stores are pending)
        size_t off[8];
        bool ok[8];
        f32x4_t ad[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int m = mt * OT + wo * 64 + rb * 16 + 2 * it + rr;
            ok[it] = m < M && col_ok;
            off[it] = (size_t)(m < M ? m : M - 1) * N + ncl;
        }
        if (Add) {
#pragma unroll
            for (int it = 0; it < 8; ++it) ad[it] = *reinterpret_cast<const f32x4_t *>(Add + off[it]);
        }
#pragma unroll
        for (int cb = 0; cb < 8; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) buf[(4 * q + r) * P + rowpad_w + cw[cb]] = acc[rb][cb][r];
        f32x4_t o[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row_l = 2 * it + rr;
            const f32x4_t v = *reinterpret_cast<const f32x4_t *>(buf + row_l * P + 8 * ((row_l >> 2) & 1) + 32 * ((row_l >> 3) & 1) + 4 * rl);
            o[it] = f32x4_t{v[0] * pre.osc[0], v[1] * pre.osc[1], v[2] * pre.osc[2], v[3] * pre.osc[3]};
        }
        if (Add) {
#pragma unroll
            for (int it = 0; it < 8; ++it) o[it] = o[it] + ad[it];
        }
#pragma unroll
        for (int it = 0; it < 8; ++it)
            if (ok[it]) *reinterpret_cast<f32x4_t *>(Out + off[it]) = o[it];
        if (want_bound) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = wo * 64 + rb * 16 + 2 * it + rr;
                const float ur = su[row], vr = sv[row];
                if (ok[it]) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) bm[e] = fmaxf(bm[e], fmaf(ur, fabsf(o[it][e]), vr));
                }
            }
        }
    }
    if (want_bound) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x = fmaxf(bm[e], __shfl_xor(bm[e], 32));
            if (rr == 0) atomicMax(cm + wb * 128 + 4 * rl + e, __float_as_uint(x));      // non-negative floats order like their bit patterns
        }
        __syncthreads();
        for (int i = tid; i < NCOL; i += NT) {
            const int n = n0 + i;
            if (n < N) atomicMax(reinterpret_cast<unsigned *>(p.out_bound) + column_keypoint(n, dirr), cm[i]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same GEMM on v_mfma_f32_16x16x32_f16 (round 4).  Measured with a speed probe first (NOTES.md): two 16x16x32 MFMAs in place of every
// 32x32x16 make the big layers 4.6 % faster on real operands -- an accumulator register is read and written once per 32 k instead of once per
// 16, and on this power-limited part that is clock.  What changes against irrep_gemm_xdma_kernel:
//   * a reduction step is K = 32 = two K16 stages of the SAME LDS images (same DMA pieces, same layouts): lane (j = l % 16, q = l / 16)
//     carries row / column j and the k octet q of the 32, i.e. octet q % 2 of stage q / 2 -- for the weights a ds_read_b128 at a per-lane
//     stage offset, for the activations the same two transposing reads per plane (k quads 2 (q % 2), + 1) in stage q / 2;
//   * the wave's 64 x 128 tile is 4 x 8 blocks of 16 x 16: 96 MFMAs per step, column block by column block (12 each: the three products of
//     the four row blocks, consecutive MFMAs independent); the NEXT column block's fragments are read under the current one's MFMAs, the
//     next step's weight fragments one per column block, the DMA pieces of the step after next one per column block;
//   * rings: activations 6 K16 stages (the step in use, the next one landed -- so that its first column block can be read before the
//     step's barrier --, the one after in flight), weights 4 (the step in use sits in registers): 96 + 64 KB = the whole LDS;
//   * accumulator block (rb, cb), register r of lane (j, q) = row 16 rb + 4 q + r, column (position) 16 cb + j of the wave's 128; a
//     position p of a 32-column block is column p / 2 + 16 (p % 2) (ft_nonlin's half-block layout), as in the other kernel;
//   * a k octet's 8 products are summed inside one MFMA in both kernels, but 32 k now meet in one instruction: results differ from the
//     32x32x16 kernel's in the last bits (same error bound), so "bitwise the register-staged kernel" no longer holds for this one.
template <int BIG>
__global__ __launch_bounds__(512, 2) void irrep_gemm_xdma16_kernel(GemmSplitDescs p, const int *__restrict__ tiles) {
    using frag = f16x8;
    using f32x4 = f32x4_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NCOL = 256, OT = 256;
    constexpr int XIMG = 16 * NCOL * 2, XSTAGE = 2 * XIMG;       // bytes: one plane's K16 x 256 image; both planes
    constexpr int ABUF = 2 * 2 * OT;                             // weight fragments per K16 stage
    constexpr int NXS = 6, NWS = 4;
    char *xs = smem;                                             // [6 stages][2 planes][XIMG]
    frag *as = reinterpret_cast<frag *>(smem + NXS * XSTAGE);    // [4 stages][2 planes][2 k-octets][256 m]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int irr = tiles[blockIdx.x * 3], mt = tiles[blockIdx.x * 3 + 1], nt = tiles[blockIdx.x * 3 + 2];
    if (irr < 0) return;
    const frag *__restrict__ W = reinterpret_cast<const frag *>(p.W[irr]);
    const int K = p.K[irr], Mpad = p.Mpad[irr], N = p.N[irr];
    const size_t split_stride = (size_t)(K / 16) * 2 * Mpad;     // in 16-byte fragments
    const int wo = w % 4, wb = w / 4;
    const int n0 = nt * NCOL;
    const int nsteps = K / 16, nss = K / 32;                     // K % 32 == 0 (C % 32 == 0)

    f32x4 acc[4][8];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) acc[rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- DMA, exactly the other kernel's pieces: per K16 step wave w issues activation pieces 2w, 2w + 1 and two weight pieces ----
    const int x_plane = w >> 2;
    const char *xsrc[2];
    int xdst[2];
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2) {
        const int mm = (2 * w + i2) & 7;
        const int pi = mm * 64 + lane;
        const int beta = pi >> 3, r = (pi & 7) >> 1, half = pi & 1;
        const int k = 4 * (beta >> 4) + r;
        // (k quads 2 and 3 hold panel P at block position P ^ 1: the four 16-lane groups of a fragment read then touch both 128-byte bank
        //  halves -- groups q and q + 1 read k quads 0 and 2 of the same panel, 4 KB apart = the same banks: measured 0.36 conflict cycles
        //  per active LDS cycle without the swap, none in the 32x32x16 kernel, whose paired groups read neighbouring panels)
        int col = n0 + 16 * ((beta & 15) ^ ((beta >> 5) & 1)) + 8 * half;
        if (col > N - 8) col = N - 8;
        xsrc[i2] = reinterpret_cast<const char *>(p.X[irr]) + (size_t)k * N * 4 + (col >> 5) * 128 + x_plane * 64 + (col & 31) * 2;
        xdst[i2] = x_plane * XIMG + mm * 1024;
    }
    const size_t xstep = (size_t)16 * N * 4;
    auto dma_x = [&](int i2, int kstep) {
        const char *src = xsrc[i2] + (size_t)(kstep < nsteps ? kstep : nsteps - 1) * xstep;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(xs + (kstep % NXS) * XSTAGE + xdst[i2]), 16, 0, 0);
    };
    const frag *wsrc = W + (size_t)(tid / OT) * Mpad + mt * OT + ((tid % OT) ^ (8 * (tid / OT)));      // (octet 1: row m at slot m ^ 8, for the same reason)
    auto dma_w = [&](int sp, int kstep) {
        const frag *src = wsrc + (size_t)(kstep < nsteps ? kstep : nsteps - 1) * 2 * Mpad;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + sp * split_stride),
                                         (__attribute__((address_space(3))) void *)(as + (kstep % NWS) * ABUF + sp * (2 * OT) + w * 64), 16, 0, 0);
    };
    // ---- fragment addresses ----
    // weights of step S: stage (2 S + q / 2) % 4, octet q % 2, row wo * 64 + rb * 16 + j
    const int a_lane = (q >> 1) * ABUF + (q & 1) * OT + wo * 64 + (j ^ (8 * (q & 1)));      // + (S % 2) * 2 * ABUF + plane * 2 * OT + rb * 16
    // activations of step S: stage (2 S) % 6 + q / 2, k quads 2 (q % 2) and + 1, panel wb * 8 + cb, position j
    const unsigned x_lane = (unsigned)(uintptr_t)xs + (unsigned)((q >> 1) * XSTAGE + ((2 * (q & 1)) * 16 + wb * 8) * 128 + j * 8);     // + (S % 3) * 2 * XSTAGE + plane * XIMG + cb * 128 (+ 2048)
    const unsigned x_swap = (q & 1) ? 128u : 0u;                 // k quads 2, 3: panel P sits at position P ^ 1 (+ 128 bytes for even column blocks, - 128 for odd ones)
    frag aA[4][2], aB[4][2];
    frag b0[2], b1[2];
    auto read_a = [&](int S, auto rb_c, auto pl_c, frag (&a)[4][2]) {
        constexpr int rb = decltype(rb_c)::value, pl = decltype(pl_c)::value;
        a[rb][pl] = as[(S & 1) * 2 * ABUF + a_lane + pl * (2 * OT) + rb * 16];
    };
    // (inline assembly, as in the other kernel: through the builtin the compiler would guard every transposing read with vmcnt(0) against
    //  the LDS-DMA in flight; the explicit waits carry the real dependences)
    auto read_b = [&](unsigned xoff, auto cb_c, frag (&b)[2]) {
        constexpr int cb = decltype(cb_c)::value;
        const unsigned a = (cb & 1) ? x_lane + xoff - x_swap : x_lane + xoff + x_swap;
        unsigned long long u[4];
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(u[0]) : "v"(a), "n"(cb * 128));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(u[1]) : "v"(a), "n"(cb * 128 + 2048));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(u[2]) : "v"(a), "n"(XIMG + cb * 128));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(u[3]) : "v"(a), "n"(XIMG + cb * 128 + 2048));
        struct Pair { unsigned long long lo, hi; };
        b[0] = __builtin_bit_cast(frag, Pair{u[0], u[1]});
        b[1] = __builtin_bit_cast(frag, Pair{u[2], u[3]});
    };
#define ROREG_PIN_B(waits, B) asm volatile(waits : "+v"(B[0]), "+v"(B[1]) :: "memory")
    using std::integral_constant;
    // One K32 step S.  a = this step's weight fragments (registers); an receives step S + 1's.  Who may touch what, and when:
    //   * column blocks 0-3 request the WEIGHTS of step S + 2 into the weight stages of step S (whose fragments were read during step S - 1),
    //     column blocks 4-7 the ACTIVATIONS of step S + 2 into the activation stages of step S - 1;
    //   * the one barrier of the step sits in front of column block 7, behind s_waitcnt vmcnt(3): in issue order only the three activation
    //     pieces of blocks 4-6 may be in flight, so this step's weight pieces and everything older (the activations of step S + 1) have landed;
    //     the same wait covers lgkmcnt(0): block 7's own fragments and `an`'s last one (requested in block 6) have been READ out of the
    //     stages the next DMA pieces overwrite before any wave passes the barrier (round 4 relied on DMA latency >> LDS latency there;
    //     the wait is free: block 7 waits for the same fragments immediately behind the barrier);
    //   * behind it column block 7 reads step S + 1's first column block, and step S + 1 reads its weight fragments; in front of it lie all
    //     reads of this step's activation stages (the last: block 7's fragments, requested in block 6) and of step S + 1's weight stages
    //     (the eight fragments of `an`, column blocks 0-6) -- the stages the next step's DMA overwrites.
    auto step = [&](int S, const frag (&a)[4][2], frag (&an)[4][2]) {
        const unsigned xoff = (unsigned)((S % 3) * 2 * XSTAGE), xoff_next = (unsigned)(((S + 1) % 3) * 2 * XSTAGE);
        static_for<8>([&](auto cb_c) {
            constexpr int cb = decltype(cb_c)::value;
            frag (&bc)[2] = (cb & 1) ? b1 : b0;
            frag (&bn)[2] = (cb & 1) ? b0 : b1;
            if constexpr (cb == 7) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            ROREG_PIN_B("s_waitcnt lgkmcnt(0)", bc);                                  // this column block's fragments (requested one block ago)
            auto mm = [&](int i) {                                                     // product i of the four row blocks: 0 = lo.hi, 1 = hi.lo, 2 = hi.hi
                const frag bb = bc[i == 1 ? 1 : 0];
                const int ai = i == 0 ? 1 : 0;
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[rb][ai], bb, acc[rb][cb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            };
            __builtin_amdgcn_sched_barrier(0);
            // the NEXT column block's fragments are requested first: this block's twelve MFMAs cover the LDS round trip
            if constexpr (cb < 7) read_b(xoff, integral_constant<int, cb + 1>{}, bn);
            else read_b(xoff_next, integral_constant<int, 0>{}, bn);                                      // ... of the next step (landed: behind the barrier)
            // ... and so is the next step's weight fragment of this block (the wait at the top of the next block covers both)
            if constexpr (cb == 0) { read_a(S + 1, integral_constant<int, 0>{}, integral_constant<int, 0>{}, an); read_a(S + 1, integral_constant<int, 0>{}, integral_constant<int, 1>{}, an); }
            else if constexpr (cb < 7) read_a(S + 1, integral_constant<int, (cb + 1) / 2>{}, integral_constant<int, (cb + 1) % 2>{}, an);
            __builtin_amdgcn_sched_barrier(0);
            mm(0);
            if constexpr (cb < 4) dma_w(cb & 1, 2 * (S + 2) + (cb >> 1));
            else dma_x(cb & 1, 2 * (S + 2) + ((cb - 4) >> 1));
            __builtin_amdgcn_sched_barrier(0);
            mm(1);
            mm(2);
        });
    };
    // ---- prologue: steps 0 and 1 (K16 stages 0-3) requested, landed; the weight fragments of step 0 and the first column block into registers ----
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { dma_w(0, ks); dma_w(1, ks); }
    for (int ks = 0; ks < 2; ++ks) { dma_x(0, ks); dma_x(1, ks); }
    Epi16Pre pre = gemm_split_epilogue16_request(p, irr, mt, n0, wb);      // (what the epilogue would otherwise open with: in flight together with the first stages)
    for (int ks = 2; ks < 4; ++ks) { dma_x(0, ks); dma_x(1, ks); }
    // in issue order the four pieces of activation stages 2, 3 are the newest: step 0's operands and step 1's weights have landed (those stages
    // are first read behind step 0's barrier, whose vmcnt(3) covers them)
    asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
    static_for<8>([&](auto c) { read_a(0, integral_constant<int, decltype(c)::value / 2>{}, integral_constant<int, decltype(c)::value % 2>{}, aA); });
    read_b(0u, integral_constant<int, 0>{}, b0);
    // (the first step's DMA pieces target stages 4, 5 of the activations -- never used -- and stages 0, 1 of the weights, whose fragments
    //  every wave must have read first)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int S = 0; S < nss; S += 2) {
        step(S, aA, aB);
        if (S + 1 < nss) step(S + 1, aB, aA);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // the clamped look-ahead pieces still target LDS the epilogue reuses
#undef ROREG_PIN_B
    gemm_split_epilogue16_scales(p, pre, n0 + wb * 128 + 4 * (lane & 31) < N);
    gemm_split_epilogue16(p, irr, mt, n0, wo, wb, acc, smem, pre);
}

// ---------------------------------------------------------------------------------------------------------------
// irrep_gemm_xdma16_kernel on HALF tiles, two workgroups per CU (round 5).  The timeline of that kernel (profiles/r05_gemm_tile_timeline.txt)
// shows 12 us of a ~100 us tile -- dispatch, the first stages' trip from HBM, the epilogue's stores -- with the matrix pipes idle, because its one
// workgroup owns the CU.  Here a workgroup is FOUR waves (one per SIMD) on a 256 x 128 tile in 80 KB of LDS, so two of them share a CU and one's
// ends run under the other's loop.  Per wave nothing changes: 64 rows x 128 columns, the same fragments, the same 96 MFMAs per K32 step in the
// same order -- results bitwise those of the 256 x 256 kernel.  What changes:
//   * activations: a K16 stage is 128 columns (8 KB), six stages as before (48 KB);
//   * weights: ONE K32 buffer (two K16 stages, 32 KB) instead of a ring of two: step S + 1's weights are requested at the start of step S into the
//     buffer whose fragments (step S) were read at the end of step S - 1, land under step S (they come from L2), and are read into registers
//     under the last column block; a thread fetches both k octets of its row, so a wave reads only what it fetched itself: no barrier for them;
//   * the 256-row weight slice is fetched by both halves of a 256 x 256 tile (L2 -> LDS traffic of the weights doubles); in exchange an XCD holds 64
//     tiles = one whole 8 x 8 block of the work list at a time (16 operand streams for 64 tiles instead of 12 for 32);
//   * workgroup b -> entry (b / 16) * 8 + b % 8 of the list, column half (b / 8) % 2: both halves on the XCD of the entry's stream.
template <int BIG>
__global__ __launch_bounds__(256, 2) void irrep_gemm_xdma16h_kernel(GemmSplitDescs p, const int *__restrict__ tiles) {
    using frag = f16x8;
    using f32x4 = f32x4_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NCOL = 128, OT = 256, NPAN = NCOL / 16;
    constexpr int XIMG = 16 * NCOL * 2, XSTAGE = 2 * XIMG;       // bytes: one plane's K16 x 128 image (4 KB); both planes
    constexpr int ABUF = 2 * 2 * OT;                             // weight fragments per K16 stage
    constexpr int NXS = 6;
    char *xs = smem;                                             // [6 stages][2 planes][XIMG]
    frag *as = reinterpret_cast<frag *>(smem + NXS * XSTAGE);    // [2 K16 stages][2 planes][2 k-octets][256 m]: one K32 step
    const int tid = threadIdx.x, lane = tid & 63, wo = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int entry = (blockIdx.x >> 4) * 8 + (blockIdx.x & 7), half_n = (blockIdx.x >> 3) & 1;
    const int irr = tiles[entry * 3], mt = tiles[entry * 3 + 1], nt = tiles[entry * 3 + 2];
    if (irr < 0) return;
    const frag *__restrict__ W = reinterpret_cast<const frag *>(p.W[irr]);
    const int K = p.K[irr], Mpad = p.Mpad[irr], N = p.N[irr];
    const int n0 = nt * 256 + half_n * NCOL;
    if (n0 >= N) return;                                         // (the right half of a ragged last column tile may be empty)
    const size_t split_stride = (size_t)(K / 16) * 2 * Mpad;     // in 16-byte fragments
    const int nsteps = K / 16, nss = K / 32;

    f32x4 acc[4][8];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) acc[rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- activation DMA: per K16 stage and plane 256 pieces of 16 bytes; wave wo issues pieces (2 wo + i2) % 4 * 64 + lane of plane wo / 2 ----
    const int x_plane = wo >> 1;
    const char *xsrc[2];
    int xdst[2];
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2) {
        const int mm = (2 * wo + i2) & 3;
        const int pi = mm * 64 + lane;                           // piece inside the plane's image: 128-byte block beta = pi / 8 = k quad * 8 + panel
        const int beta = pi >> 3, r = (pi & 7) >> 1, half = pi & 1;
        const int k = 4 * (beta >> 3) + r;
        // (k quads 2 and 3 hold panel P at block position P ^ 1, as in the 256-column image: the 16-lane groups q and q + 1 of a fragment read
        //  touch k quads 0 and 2 of the same panel, 2 KB apart = the same banks)
        int col = n0 + 16 * ((beta & 7) ^ ((beta >> 4) & 1)) + 8 * half;
        if (col > N - 8) col = N - 8;
        xsrc[i2] = reinterpret_cast<const char *>(p.X[irr]) + (size_t)k * N * 4 + (col >> 5) * 128 + x_plane * 64 + (col & 31) * 2;
        xdst[i2] = x_plane * XIMG + mm * 1024;
    }
    const size_t xstep = (size_t)16 * N * 4;
    auto dma_x = [&](int i2, int kstep) {
        const char *src = xsrc[i2] + (size_t)(kstep < nsteps ? kstep : nsteps - 1) * xstep;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(xs + (kstep % NXS) * XSTAGE + xdst[i2]), 16, 0, 0);
    };
    // ---- weight DMA: per K16 stage [2 planes][2 k-octets][256 m]; a thread fetches row m = tid of both octets of both planes ----
    const frag *wsrc0 = W + mt * OT + tid, *wsrc1 = W + (size_t)Mpad + mt * OT + (tid ^ 8);      // (octet 1: row m at slot m ^ 8, as in the other kernel)
    auto dma_w = [&](int sp, int oct, int kstep) {
        const frag *src = (oct ? wsrc1 : wsrc0) + (size_t)(kstep < nsteps ? kstep : nsteps - 1) * 2 * Mpad;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + sp * split_stride),
                                         (__attribute__((address_space(3))) void *)(as + (kstep & 1) * ABUF + sp * (2 * OT) + oct * OT + wo * 64), 16, 0, 0);
    };
    // ---- fragment addresses ----
    const int a_lane = (q >> 1) * ABUF + (q & 1) * OT + wo * 64 + (j ^ (8 * (q & 1)));      // + plane * 2 * OT + rb * 16
    const unsigned x_lane = (unsigned)(uintptr_t)xs + (unsigned)((q >> 1) * XSTAGE + ((2 * (q & 1)) * NPAN) * 128 + j * 8);      // + (S % 3) * 2 * XSTAGE + plane * XIMG + cb * 128 (+ NPAN * 128)
    const unsigned x_swap = (q & 1) ? 128u : 0u;
    frag aA[4][2], aB[4][2];
    frag b0[2], b1[2];
    auto read_a = [&](auto rb_c, auto pl_c, frag (&a)[4][2]) {
        constexpr int rb = decltype(rb_c)::value, pl = decltype(pl_c)::value;
        a[rb][pl] = as[a_lane + pl * (2 * OT) + rb * 16];
    };
    auto read_b = [&](unsigned xoff, auto cb_c, frag (&b)[2]) {
        constexpr int cb = decltype(cb_c)::value;
        const unsigned a = (cb & 1) ? x_lane + xoff - x_swap : x_lane + xoff + x_swap;
        unsigned long long u[4];
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(u[0]) : "v"(a), "n"(cb * 128));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(u[1]) : "v"(a), "n"(cb * 128 + NPAN * 128));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(u[2]) : "v"(a), "n"(XIMG + cb * 128));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(u[3]) : "v"(a), "n"(XIMG + cb * 128 + NPAN * 128));
        struct Pair { unsigned long long lo, hi; };
        b[0] = __builtin_bit_cast(frag, Pair{u[0], u[1]});
        b[1] = __builtin_bit_cast(frag, Pair{u[2], u[3]});
    };
#define ROREG_PIN_B(waits, B) asm volatile(waits : "+v"(B[0]), "+v"(B[1]) :: "memory")
    using std::integral_constant;
    // One K32 step S.  a = this step's weight fragments (registers); an receives step S + 1's.
    //   * the weight buffer needs NO barrier: a wave fetches exactly the 64 rows (both octets, both planes) whose fragments it reads itself, so its
    //     lgkmcnt(0) at the top of column block 0 (the reads of step S's fragments, requested under the previous step's last block, are done) orders
    //     them against its own requests of step S + 1's weights in column blocks 0-3 (two pieces each), and its vmcnt wait in front of block 7
    //     orders those against its reads; column blocks 4-7 request the ACTIVATIONS of step S + 2 into the stages of step S - 1;
    //   * the one barrier sits in front of column block 7, behind s_waitcnt vmcnt(3) lgkmcnt(0): in issue order only the activation pieces of
    //     blocks 4-6 may be in flight, so step S + 1's activations (requested a step ago, by all four waves) have landed for everyone and every
    //     read of this step's activation stages is done -- and this wave's weights of step S + 1 have landed;
    //   * behind it block 7 reads step S + 1's eight weight fragments and its first column block's activation fragments.
    auto step = [&](int S, const frag (&a)[4][2], frag (&an)[4][2]) {
        const unsigned xoff = (unsigned)((S % 3) * 2 * XSTAGE), xoff_next = (unsigned)(((S + 1) % 3) * 2 * XSTAGE);
        static_for<8>([&](auto cb_c) {
            constexpr int cb = decltype(cb_c)::value;
            frag (&bc)[2] = (cb & 1) ? b1 : b0;
            frag (&bn)[2] = (cb & 1) ? b0 : b1;
            if constexpr (cb == 7) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            ROREG_PIN_B("s_waitcnt lgkmcnt(0)", bc);          // (cb == 0: also this wave's reads of step S's weight fragments -- see below)
            auto mm = [&](int i) {                                                     // product i of the four row blocks: 0 = lo.hi, 1 = hi.lo, 2 = hi.hi
                const frag bb = bc[i == 1 ? 1 : 0];
                const int ai = i == 0 ? 1 : 0;
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[rb][ai], bb, acc[rb][cb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            };
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (cb < 7) read_b(xoff, integral_constant<int, cb + 1>{}, bn);
            else {
                read_b(xoff_next, integral_constant<int, 0>{}, bn);
                static_for<8>([&](auto c) { read_a(integral_constant<int, decltype(c)::value / 2>{}, integral_constant<int, decltype(c)::value % 2>{}, an); });
            }
            __builtin_amdgcn_sched_barrier(0);
            mm(0);
            if constexpr (cb < 4) { dma_w(cb >> 1, cb & 1, 2 * (S + 1)); dma_w(cb >> 1, cb & 1, 2 * (S + 1) + 1); }
            else dma_x(cb & 1, 2 * (S + 2) + ((cb - 4) >> 1));
            __builtin_amdgcn_sched_barrier(0);
            mm(1);
            mm(2);
        });
    };
    // ---- prologue: activations of steps 0 and 1 (K16 stages 0-3), weights of step 0; the scale / bound operands of the epilogue beside them ----
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) { dma_w(0, 0, ks); dma_w(0, 1, ks); dma_w(1, 0, ks); dma_w(1, 1, ks); }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) { dma_x(0, ks); dma_x(1, ks); }
    Epi16Pre pre = gemm_split_epilogue16_request(p, irr, mt, n0, 0);
#pragma unroll
    for (int ks = 2; ks < 4; ++ks) { dma_x(0, ks); dma_x(1, ks); }
    asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");              // (stages 2, 3 of the activations land under step 0: its second barrier covers them)
    static_for<8>([&](auto c) { read_a(integral_constant<int, decltype(c)::value / 2>{}, integral_constant<int, decltype(c)::value % 2>{}, aA); });
    read_b(0u, integral_constant<int, 0>{}, b0);
    for (int S = 0; S < nss; S += 2) {
        step(S, aA, aB);
        if (S + 1 < nss) step(S + 1, aB, aA);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // the clamped look-ahead pieces still target LDS the epilogue reuses
#undef ROREG_PIN_B
    gemm_split_epilogue16_scales(p, pre, n0 + 4 * (lane & 31) < N);
    gemm_split_epilogue16<256, NCOL>(p, irr, mt, n0, wo, 0, acc, smem, pre);
}

// ---------------------------------------------------------------------------------------------------------------
// irrep_gemm_xdma16_kernel as PERSISTENT workgroups (round 5).  That kernel's workgroup owns a CU (all 160 KB of LDS), so nothing runs under
// the ends of its life: the dispatch of its successor, the tile decode, the first DMA pieces' trip from HBM (the loop cannot start before
// two K32 steps of both operands have landed) and the epilogue are exposed once per 256 x 256 tile.  Here gridDim.x = 8 c workgroups (c per
// XCD, one per CU) walk the work list: workgroup b takes positions b / 8, b / 8 + c, ... of stream b % 8 (the list is eight interleaved
// per-XCD streams, roreg_irrep_gemm_tiles_m; which CU runs a workgroup decides nothing but speed), and requests the next tile's first operand
// stages BEFORE it has stored the finished one:
//     loop end (rings dead) | next entry (scalar loads, under the epilogue's own first loads) | epilogue passes 0 .. RA-1 | DMA: weights K16
//     stages 0-3 + activations 0-2 -> ring slots the epilogue does not touch | passes RA .. 7 | barrier | DMA: activation stage 3 | next loop
// The main loop (steps, waits, barrier) is the other kernel's, instruction for instruction, and so is every accumulator's MFMA sequence:
// results are bitwise the same.  The epilogue works in 48 KB .. 88 KB of the LDS (activation stages 3-5): a wave passes 16 rows x 64 columns
// at a time through a 16 x 72 float buffer (8 passes per tile instead of 4 of 128 columns: 4.5 KB per wave instead of 12); the write side's
// four 16-lane groups are shifted by 0, 32, 8, 40 banks (row pitch 72 = 8 mod 64, four rows = 32, + 8 floats for the rows of q >= 2), the
// read side's 16-lane groups read 256 contiguous bytes: no bank conflicts on either side.
template <int BIG, int RA>
__global__ __launch_bounds__(512, 2) void irrep_gemm_xdma16p_kernel(GemmSplitDescs p, const int *__restrict__ tiles, int longest) {
    using frag = f16x8;
    using f32x4 = f32x4_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NCOL = 256, OT = 256, NT = 512;
    constexpr int XIMG = 16 * NCOL * 2, XSTAGE = 2 * XIMG;
    constexpr int ABUF = 2 * 2 * OT;
    constexpr int NXS = 6, NWS = 4;
    constexpr int EP = 72;                                       // row pitch (floats) of a wave's epilogue buffer
    constexpr int EPI0 = 3 * XSTAGE, EPI_BYTES = 4096 + 8 * 16 * EP * 4;      // 48 KB .. 88 KB
    static_assert(EPI0 + EPI_BYTES <= NXS * XSTAGE, "the epilogue stays inside activation stages 3-5");
    char *xs = smem;
    frag *as = reinterpret_cast<frag *>(smem + NXS * XSTAGE);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int wo = w % 4, wb = w / 4;
    const int stream_k = blockIdx.x & 7, pos_step = gridDim.x >> 3;
    int pos = blockIdx.x >> 3;
    if (pos >= longest) return;
    int irr = tiles[(size_t)(pos * 8 + stream_k) * 3], mt = tiles[(size_t)(pos * 8 + stream_k) * 3 + 1], nt = tiles[(size_t)(pos * 8 + stream_k) * 3 + 2];
    if (irr < 0) return;                                         // (padding: the stream is shorter than the longest one)

    // ---- DMA of the tile (irr, mt, nt): exactly the other kernel's pieces ----
    const int x_plane = w >> 2;
    const char *xsrc[2];
    int xdst[2];
    const frag *wsrc = nullptr;
    size_t xstep = 0, split_stride = 0;
    int nsteps = 0, nss = 0, Mpad = 0;
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2) { xsrc[i2] = nullptr; xdst[i2] = x_plane * XIMG + ((2 * w + i2) & 7) * 1024; }
    auto setup = [&]() {
        const int K = p.K[irr], N = p.N[irr];
        Mpad = p.Mpad[irr];
        nsteps = K / 16; nss = K / 32;
        split_stride = (size_t)(K / 16) * 2 * Mpad;
        xstep = (size_t)16 * N * 4;
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {
            const int mm = (2 * w + i2) & 7;
            const int pi = mm * 64 + lane;
            const int beta = pi >> 3, r = (pi & 7) >> 1, half = pi & 1;
            const int k = 4 * (beta >> 4) + r;
            int col = nt * NCOL + 16 * ((beta & 15) ^ ((beta >> 5) & 1)) + 8 * half;
            if (col > N - 8) col = N - 8;
            xsrc[i2] = reinterpret_cast<const char *>(p.X[irr]) + (size_t)k * N * 4 + (col >> 5) * 128 + x_plane * 64 + (col & 31) * 2;
        }
        wsrc = reinterpret_cast<const frag *>(p.W[irr]) + (size_t)(tid / OT) * Mpad + mt * OT + ((tid % OT) ^ (8 * (tid / OT)));
    };
    auto dma_x = [&](int i2, int kstep) {
        const char *src = xsrc[i2] + (size_t)(kstep < nsteps ? kstep : nsteps - 1) * xstep;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(xs + (kstep % NXS) * XSTAGE + xdst[i2]), 16, 0, 0);
    };
    auto dma_w = [&](int sp, int kstep) {
        const frag *src = wsrc + (size_t)(kstep < nsteps ? kstep : nsteps - 1) * 2 * Mpad;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + sp * split_stride),
                                         (__attribute__((address_space(3))) void *)(as + (kstep % NWS) * ABUF + sp * (2 * OT) + w * 64), 16, 0, 0);
    };
    auto request_a = [&]() {                                     // everything of the first two K32 steps that lands outside 48 KB .. 96 KB
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { dma_w(0, ks); dma_w(1, ks); }
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) { dma_x(0, ks); dma_x(1, ks); }
    };
    auto request_b = [&]() { dma_x(0, 3); dma_x(1, 3); };

    // ---- fragment addresses (tile independent) ----
    const int a_lane = (q >> 1) * ABUF + (q & 1) * OT + wo * 64 + (j ^ (8 * (q & 1)));
    const unsigned x_lane = (unsigned)(uintptr_t)xs + (unsigned)((q >> 1) * XSTAGE + ((2 * (q & 1)) * 16 + wb * 8) * 128 + j * 8);
    const unsigned x_swap = (q & 1) ? 128u : 0u;
    frag aA[4][2], aB[4][2];
    frag b0[2], b1[2];
    f32x4 acc[4][8];
    auto read_a = [&](int S, auto rb_c, auto pl_c, frag (&a)[4][2]) {
        constexpr int rb = decltype(rb_c)::value, pl = decltype(pl_c)::value;
        a[rb][pl] = as[(S & 1) * 2 * ABUF + a_lane + pl * (2 * OT) + rb * 16];
    };
    auto read_b = [&](unsigned xoff, auto cb_c, frag (&b)[2]) {
        constexpr int cb = decltype(cb_c)::value;
        const unsigned a = (cb & 1) ? x_lane + xoff - x_swap : x_lane + xoff + x_swap;
        unsigned long long u[4];
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(u[0]) : "v"(a), "n"(cb * 128));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(u[1]) : "v"(a), "n"(cb * 128 + 2048));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(u[2]) : "v"(a), "n"(XIMG + cb * 128));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(u[3]) : "v"(a), "n"(XIMG + cb * 128 + 2048));
        struct Pair { unsigned long long lo, hi; };
        b[0] = __builtin_bit_cast(frag, Pair{u[0], u[1]});
        b[1] = __builtin_bit_cast(frag, Pair{u[2], u[3]});
    };
#define ROREG_PIN_B(waits, B) asm volatile(waits : "+v"(B[0]), "+v"(B[1]) :: "memory")
    using std::integral_constant;
    auto step = [&](int S, const frag (&a)[4][2], frag (&an)[4][2]) {      // (irrep_gemm_xdma16_kernel's step: see there for who may touch what, and when)
        const unsigned xoff = (unsigned)((S % 3) * 2 * XSTAGE), xoff_next = (unsigned)(((S + 1) % 3) * 2 * XSTAGE);
        static_for<8>([&](auto cb_c) {
            constexpr int cb = decltype(cb_c)::value;
            frag (&bc)[2] = (cb & 1) ? b1 : b0;
            frag (&bn)[2] = (cb & 1) ? b0 : b1;
            if constexpr (cb == 7) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            ROREG_PIN_B("s_waitcnt lgkmcnt(0)", bc);
            auto mm = [&](int i) {
                const frag bb = bc[i == 1 ? 1 : 0];
                const int ai = i == 0 ? 1 : 0;
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[rb][ai], bb, acc[rb][cb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            };
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (cb < 7) read_b(xoff, integral_constant<int, cb + 1>{}, bn);
            else read_b(xoff_next, integral_constant<int, 0>{}, bn);
            if constexpr (cb == 0) { read_a(S + 1, integral_constant<int, 0>{}, integral_constant<int, 0>{}, an); read_a(S + 1, integral_constant<int, 0>{}, integral_constant<int, 1>{}, an); }
            else if constexpr (cb < 7) read_a(S + 1, integral_constant<int, (cb + 1) / 2>{}, integral_constant<int, (cb + 1) % 2>{}, an);
            __builtin_amdgcn_sched_barrier(0);
            mm(0);
            if constexpr (cb < 4) dma_w(cb & 1, 2 * (S + 2) + (cb >> 1));
            else dma_x(cb & 1, 2 * (S + 2) + ((cb - 4) >> 1));
            __builtin_amdgcn_sched_barrier(0);
            mm(1);
            mm(2);
        });
    };

    // ---- epilogue state (of the tile that has just been multiplied) ----
    float *su = reinterpret_cast<float *>(smem + EPI0), *sv = su + OT;
    unsigned *cm = reinterpret_cast<unsigned *>(sv + OT);
    float *ebuf = reinterpret_cast<float *>(smem + EPI0 + 4096) + w * (16 * EP);
    int cwl[4];                                                  // write side: the column (inside the pass's 64) of block cbl in this lane
#pragma unroll
    for (int cbl = 0; cbl < 4; ++cbl) {
        const int pp = (cbl & 1) * 16 + j;
        cwl[cbl] = (cbl >> 1) * 32 + (pp >> 1) + 16 * (pp & 1);
    }
    const int wr_off = 4 * q * EP + 8 * (q >> 1);                // + r * EP + cwl
    const int rl = lane & 15, rrow = lane >> 4;                  // read side: columns 4 rl .. 4 rl + 3 of the 64, row rrow of every four
    int e_mt = 0, e_M = 0, e_N = 0, e_dirr = 1;
    float *__restrict__ e_Out = nullptr;
    const float *__restrict__ e_Add = nullptr;
    const bool want_bound = p.out_bound != nullptr;
    int e_ncol[2];
    float osc[2][4], bm[2][4];
    auto epi_begin = [&]() {                                     // (the rings are dead: the loop ended with a barrier)
        e_mt = mt; e_M = p.M[irr]; e_N = p.N[irr]; e_dirr = kIrrDim[irr];
        e_Out = p.Out[irr]; e_Add = p.Add[irr];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            e_ncol[hf] = nt * NCOL + wb * 128 + hf * 64 + 4 * rl;      // (N % 32 == 0: the four columns are inside or outside together)
            // (four adjacent columns of a 32-column block = four adjacent keypoints: one 16-byte load)
            const f32x4 xb = e_ncol[hf] < e_N ? *reinterpret_cast<const f32x4 *>(p.xbound + column_keypoint(e_ncol[hf], e_dirr)) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                osc[hf][e] = e_ncol[hf] < e_N ? ldexpf(1.f, -(bound_exp(xb[e]) + p.w_exp)) : 1.f;
                bm[hf][e] = 0.f;
            }
        }
        if (want_bound) {
            for (int i = tid; i < OT; i += NT) {
                const int m = e_mt * OT + i;
                su[i] = m < e_M ? p.nb_u[m % p.O] : 0.f; sv[i] = m < e_M ? p.nb_v[m % p.O] : 0.f;
            }
            for (int i = tid; i < NCOL; i += NT) cm[i] = 0u;
            __syncthreads();
        }
    };
    auto epi_pass = [&](auto rb_c, auto hf_c) {
        constexpr int rb = decltype(rb_c)::value, hf = decltype(hf_c)::value;
        // the pass's four residual rows are requested first (from clamped, always valid addresses) and arrive under the LDS round trip: the
        // compiler drains vmcnt completely in front of the first use of a load while stores are pending, so loads and stores are kept in
        // two groups per pass instead of 4 x (load, wait, store)
        const bool col_ok = e_ncol[hf] < e_N;
        const int ncl = col_ok ? e_ncol[hf] : e_N - 4;
        size_t off[4];
        bool ok[4];
        f32x4 ad[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int m = e_mt * OT + wo * 64 + rb * 16 + 4 * it + rrow;
            ok[it] = m < e_M && col_ok;
            off[it] = (size_t)(m < e_M ? m : e_M - 1) * e_N + ncl;
        }
        if (e_Add) {
#pragma unroll
            for (int it = 0; it < 4; ++it) ad[it] = *reinterpret_cast<const f32x4 *>(e_Add + off[it]);
        }
#pragma unroll
        for (int cbl = 0; cbl < 4; ++cbl)
#pragma unroll
            for (int r = 0; r < 4; ++r) ebuf[wr_off + r * EP + cwl[cbl]] = acc[rb][4 * hf + cbl][r];
        f32x4 o[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(ebuf + (4 * it + rrow) * EP + 8 * (it >> 1) + 4 * rl);
            o[it] = f32x4{v[0] * osc[hf][0], v[1] * osc[hf][1], v[2] * osc[hf][2], v[3] * osc[hf][3]};
        }
        if (e_Add) {
#pragma unroll
            for (int it = 0; it < 4; ++it) o[it] = o[it] + ad[it];
        }
#pragma unroll
        for (int it = 0; it < 4; ++it)
            if (ok[it]) *reinterpret_cast<f32x4 *>(e_Out + off[it]) = o[it];
        if (want_bound) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = wo * 64 + rb * 16 + 4 * it + rrow;
                const float ur = su[row], vr = sv[row];
                if (ok[it]) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) bm[hf][e] = fmaxf(bm[hf][e], fmaf(ur, fabsf(o[it][e]), vr));
                }
            }
        }
    };
    auto epi_end = [&]() {
        if (want_bound) {
            const int e_n0 = e_ncol[0] - (wb * 128 + 4 * rl);
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = fmaxf(bm[hf][e], __shfl_xor(bm[hf][e], 16));
                    x = fmaxf(x, __shfl_xor(x, 32));
                    if (rrow == 0) atomicMax(cm + wb * 128 + hf * 64 + 4 * rl + e, __float_as_uint(x));      // non-negative floats order like their bit patterns
                }
            __syncthreads();
            for (int i = tid; i < NCOL; i += NT) {
                const int n = e_n0 + i;
                if (n < e_N) atomicMax(reinterpret_cast<unsigned *>(p.out_bound) + column_keypoint(n, e_dirr), cm[i]);
            }
        }
    };

    // ---- first tile ----
    setup(); request_a(); request_b();
    for (;;) {
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) acc[rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
        // in issue order the two pieces of activation stage 3 are the newest: everything of steps 0 and 1 but that stage has landed (the
        // stage is first read behind step 0's barrier, whose vmcnt(3) covers it)
        asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
        static_for<8>([&](auto c) { read_a(0, integral_constant<int, decltype(c)::value / 2>{}, integral_constant<int, decltype(c)::value % 2>{}, aA); });
        read_b(0u, integral_constant<int, 0>{}, b0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        for (int S = 0; S < nss; S += 2) {
            step(S, aA, aB);
            if (S + 1 < nss) step(S + 1, aB, aA);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // the clamped look-ahead pieces have landed: the rings are dead
        // the next entry of this workgroup's walk: scalar loads, in flight together with the epilogue's first loads
        pos += pos_step;
        int n_irr = -1, n_mt = 0, n_nt = 0;
        if (pos < longest) {
            const int *t = tiles + (size_t)(pos * 8 + stream_k) * 3;
            n_irr = t[0]; n_mt = t[1]; n_nt = t[2];
        }
        epi_begin();
        const bool more = n_irr >= 0;
        static_for<8>([&](auto ps) {
            constexpr int P = decltype(ps)::value;
            if constexpr (P == RA) {
                if (more) { irr = n_irr; mt = n_mt; nt = n_nt; setup(); request_a(); }
            }
            epi_pass(integral_constant<int, P / 2>{}, integral_constant<int, P % 2>{});
        });
        epi_end();
        __syncthreads();                                         // every wave is done with 48 KB .. 88 KB: activation stage 3 may be overwritten
        if (!more) break;
        request_b();
    }
#undef ROREG_PIN_B
}

// ---------------------------------------------------------------------------------------------------------------
struct NonlinParams {
    const float *Xin;            // flat coefficient buffer [60*C*B] (nullptr when the input is spatial)
    float *Xout;                 // flat coefficient buffer out (nullptr when the output is spatial)
    const float *x_spatial;      // [B,C,60] input in the group domain
    const float *r_spatial;      // [B,C,60] residual added in the group domain (before BN/ReLU; only with spatial output)
    float *out_spatial;          // [B,C,Lout]
    const int *g_map;            // optional [60] -> compact output column (or -1 = not written); Lout columns are written
    const float *bias, *bias2, *bn_scale, *bn_shift;      // per channel; any may be null
    const float *A1, *A2;        // fragment-ordered transform tables (roreg_set_fourier_tables)
    const bf16x8 *A1s, *A2s;     // the same tables as 3 x bf16 split fragments of the K=16 bf16 MFMA (SPLIT = 3 kernels)
    const f16x8 *A1h, *A2h;      // ... and as fp16 hi/lo fragments scaled by 2^f_exp (SPLIT = 2 kernels)
    int f_exp;
    const float *out_bound;      // SPLIT = 2, coefficient output: per-keypoint bound [Bp] on |coefficient| (from the producing GEMM's epilogue or
                                 // roreg_row_bound); the output is then written SPLIT for the GEMM: fp16 hi | fp16 lo << 16 of coef * 2^bound_exp(bound)
    float *out_rowmax;           // optional, group-domain output: per-keypoint max |value written| [B] (zeroed by the caller; atomic max), the
                                 // block scale of the fp16 x 2 convolution that consumes the tensor
    int x_bf16;                  // the group-domain input / residual tensors are bfloat16 instead of float32
    int out_planes;              // SPLIT = 2 coefficient output in HALF-BLOCK layout (see irrep_gemm_xdma_kernel): every 32-column block of a row holds
                                 // its 32 fp16 hi values (order 0, 16, 1, 17, ...), then its 32 lo values, instead of 32 words hi | lo << 16
    int sp_pack;                 // SPLIT = 2, group-domain output: the values leave as WORDS fp16 hi | fp16 lo << 16 of value * 2^bound_exp(out_bound[b]) (the operand of
                                 // roreg_group_conv_f16x2_packed: its staging then only unpacks -- no BatchNorm, no conversion); out_bound [Bp] is required
    float *raw_col;              // with sp_pack, optional [B][C]: the value BEFORE BatchNorm / ReLU of group element raw_g (ET's identity short cut reads column g = 0)
    int raw_g;
    int B, Bp, C, tiles_per_c, Lout, Lvalid;     // B valid keypoints; Bp = B rounded up to 32 = the column pitch unit of the coefficient buffers
};

// Coefficient layout: irrep rho occupies [off_rho*C*Bp, off_{rho+1}*C*Bp) as the row-major GEMM operand [d*C][d*Bp]; row (l, c), and
// inside a row the columns are blocked by 32 keypoints: column = (b/32)*(32*d) + i*32 + (b%32).  The GEMMs never look inside a row,
// and a 32-keypoint tile of this kernel reads, per (rho, l, c), ONE contiguous segment of d*128 bytes (16 segments of avg 480 B per
// tile instead of 60 scattered 128-byte pieces).  coefficient q = (rho, i, l) of (b, c):
//   offset = (alpha_q*C + c*d_q) * Bp + (b/32)*32*d_q + i_q*32 + b%32          with alpha_q = offset_rho + l*d.
// The table is a compile-time constant: with the transform loops unrolled every row index is (constant*C + constant + c*constant),
// i.e. scalar-ALU work on the wave-uniform channel c, and a lane only selects between the two candidates of its half-wave.
struct QRows { int alpha[64], d[64], i[64]; };
constexpr QRows make_qrows() {
    QRows t{};
    const int dims[NIRR] = {1, 3, 3, 4, 5};
    int off = 0, q = 0;
    for (int r = 0; r < NIRR; ++r) {
        for (int i = 0; i < dims[r]; ++i)
            for (int l = 0; l < dims[r]; ++l, ++q) { t.alpha[q] = off + l * dims[r]; t.d[q] = dims[r]; t.i[q] = i; }
        off += dims[r] * dims[r];
    }
    for (; q < 64; ++q) { t.alpha[q] = 0; t.d[q] = 1; t.i[q] = 0; }
    return t;
}
constexpr QRows kQ = make_qrows();

// Which coefficient the forward transform produces in which MFMA output row.  The two half-waves of one store instruction hold output rows
// m and m + 4; the assignment of coefficients to rows is free (it only permutes the rows of the forward table), so the 32 row pairs are filled
// such that a pair's two addresses differ by a WAVE-UNIFORM amount:
//   24 pairs (rho, i, l), (rho, i + 1, l)   : neighbouring 128-byte blocks of one row segment -> the instruction covers 256 contiguous bytes,
//    4 pairs (rho, 0, l), (rho, 0, l + 1)   : the same block of two consecutive rows (d odd: rows of 3 / 5 blocks leave one over),
//    4 pairs (rho, 0, l_last), none         : the last left-over of the d = 1, 3, 3, 5 irreps with one of the four unused rows 60..63.
// Every store is then (scalar row base) + (one of three per-lane byte offsets fixed for the whole kernel) + immediate: no per-access vector
// address arithmetic (with rows in (rho, i, l) order every access needed a 64-bit select and add: the kernel is issue-bound at its occupancy
// and address arithmetic was a third of its instructions).  The INVERSE transform keeps its K slots in (rho, i, l) order: permuting them
// would change the order of its float32 sums, i.e. the low bits of every feature.
struct PairSlots {
    int qa[32], qb[32];          // canonical coefficient index off_rho + i*d + l of the two members; qb = 64: none
    int row[32];                 // row id of member a: 0 | 1..3 | 4..6 | 7..10 | 11..15  (rho, l)
    int i[32];                   // block of member a inside its row segment
    int kind[32];                // 0: member b is the next block; 1: member b is the same block of the next row; 2: no member b
    int d[32];
    int row_alpha[16], row_d[16];         // a row's first coefficient-row index off_rho + l*d, and its irrep dimension
};
constexpr PairSlots make_pair_slots() {
    PairSlots t{};
    const int dims[NIRR] = {1, 3, 3, 4, 5};
    int off = 0, s = 0, row0 = 0;
    for (int r = 0; r < NIRR; ++r) {
        const int d = dims[r];
        for (int l = 0; l < d; ++l) {
            t.row_alpha[row0 + l] = off + l * d; t.row_d[row0 + l] = d;
            for (int i = d % 2; i < d; i += 2, ++s) {
                t.qa[s] = off + i * d + l; t.qb[s] = off + (i + 1) * d + l; t.row[s] = row0 + l; t.i[s] = i; t.kind[s] = 0; t.d[s] = d;
            }
        }
        if (d % 2)
            for (int l = 0; l < d; l += 2, ++s) {
                t.qa[s] = off + l; t.row[s] = row0 + l; t.i[s] = 0; t.d[s] = d;
                if (l + 1 < d) { t.qb[s] = off + l + 1; t.kind[s] = 1; } else { t.qb[s] = 64; t.kind[s] = 2; }
            }
        off += d * d; row0 += d;
    }
    return t;
}
constexpr PairSlots kPS = make_pair_slots();
// coefficient produced in row m = 32 t + (r & 3) + 8 (r >> 2) + 4 h of the forward transform (pair slot t*16 + r); >= 60: none
constexpr int slot_out_q(int m) { const int mm = m & 31, s = (m >> 5) * 16 + (mm & 3) + 4 * (mm >> 3); return (mm & 4) ? kPS.qb[s] : kPS.qa[s]; }

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N-1>{})
template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F &&f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

// SPLIT: both 60x60 transforms run on the bf16 matrix cores with f32 accuracy (operands split 3 x bf16, six cross products, f32
// accumulate -- see irrep_gemm_split_kernel): 96 K=16 MFMAs (3072 cycles) per 32-keypoint tile instead of 124 f32 MFMAs (7936), which
// turns the kernel from matrix-core-bound into HBM-bound.  K orders: inverse step st feeds coefficients q = 16 st + 8 h + e; the forward
// product's step st consumes this lane's accumulator registers v[st>>1][8 (st&1) + e], i.e. again no transpose between the products.
// OUT_ROWS (with IN_SPATIAL): the coefficients leave in per-keypoint layout [b][c][60] (float32) instead of the GEMM operand layout -- the
// operand of the irrep-domain Des2R (roreg_feat_coefs).
template <bool IN_SPATIAL, bool OUT_SPATIAL, int SPLIT /* 0: f32 MFMA, 3: bf16 x 3, 2: fp16 x 2 with per-column scales */, int NW /* waves per workgroup */, int MINW /* waves per SIMD to fit */, bool OUT_ROWS = false>
__global__ __launch_bounds__(NW * 64, MINW) void ft_nonlin_kernel(NonlinParams p) {
    constexpr int NT = NW * 64;
    const int lane = threadIdx.x & 63;
    const int jn = lane & 31, h = lane >> 5;
    const int wave_global = __builtin_amdgcn_readfirstlane((blockIdx.x * NT + threadIdx.x) >> 6);   // wave-uniform: tile, c are scalars
    const int n_waves = gridDim.x * NW;
    const int B = p.B, C = p.C;
    const int n_tiles = C * p.tiles_per_c;

    // transform fragments A1[s][tile][lane] (inverse) and A2[step][tile][lane] (forward, K order = C/D register order) live in LDS
    // (31 KB per workgroup, conflict-free lane-contiguous reads): keeping them out of the register file leaves room for 4+ waves
    // per SIMD, which this streaming kernel needs to cover its 30 scattered 128-byte row reads per tile.
    constexpr int NPL = SPLIT == 2 ? 2 : 3;                      // planes of the split tables
    constexpr int NA1 = SPLIT ? 4 * 2 * NPL * 64 * 4 : 30 * 2 * 64, NA2 = SPLIT ? 4 * 2 * NPL * 64 * 4 : 32 * 2 * 64;   // floats (a fragment = 4 floats)
    __shared__ __attribute__((aligned(16))) float sA1[IN_SPATIAL ? 64 : NA1];
    __shared__ __attribute__((aligned(16))) float sA2[OUT_SPATIAL ? 64 : NA2];
    __shared__ float sT[(IN_SPATIAL || OUT_SPATIAL || OUT_ROWS) ? NW * 32 * 65 : 64];     // per wave: [32 keypoints][65] transpose buffer
    {
        const float *g1 = SPLIT == 3 ? reinterpret_cast<const float *>(p.A1s) : SPLIT == 2 ? reinterpret_cast<const float *>(p.A1h) : p.A1;
        const float *g2 = SPLIT == 3 ? reinterpret_cast<const float *>(p.A2s) : SPLIT == 2 ? reinterpret_cast<const float *>(p.A2h) : p.A2;
        if (!IN_SPATIAL)
            for (int i = threadIdx.x; i < NA1; i += NT) sA1[i] = g1[i];
        if (!OUT_SPATIAL)
            for (int i = threadIdx.x; i < NA2; i += NT) sA2[i] = g2[i];
    }
    // per-channel epilogue constants also live in LDS: as vector loads inside the tile loop they would force s_waitcnt vmcnt(0), i.e.
    // drain the coefficient prefetch of the next tile (VMEM operations of a wave complete in order)
    __shared__ float sBias[512], sScale[512], sShift[512];
    __shared__ int sGmap[64];                                   // output column of group element g (-1 = not written); same reason
    if (threadIdx.x < 64) sGmap[threadIdx.x] = (OUT_SPATIAL && p.g_map && threadIdx.x < ROREG_G) ? p.g_map[threadIdx.x] : (int)threadIdx.x;
    for (int i = threadIdx.x; i < p.C; i += NT) {
        sBias[i] = (p.bias ? p.bias[i] : 0.f) + (p.bias2 ? p.bias2[i] : 0.f);
        sScale[i] = p.bn_scale ? p.bn_scale[i] : 1.f;
        sShift[i] = p.bn_shift ? p.bn_shift[i] : 0.f;
    }
    __syncthreads();
    const bf16x8 *sA1s = reinterpret_cast<const bf16x8 *>(sA1), *sA2s = reinterpret_cast<const bf16x8 *>(sA2);   // [st][tile][plane][lane]
    const f16x8 *sA1h = reinterpret_cast<const f16x8 *>(sA1), *sA2h = reinterpret_cast<const f16x8 *>(sA2);
    // flat offset (without the lane's keypoint jn) of coefficient q of channel c in keypoint tile tb; the table entries are compile-time
    // constants, so both half-wave candidates are scalar-ALU values and a lane only selects
    const size_t Bp = (size_t)p.Bp;
    const size_t hmask = (size_t)0 - (size_t)h;                  // all ones for the second half-wave
#define OFF_Q(Q, c, tb) ((size_t)(kQ.alpha[Q] * C + (c) * kQ.d[Q]) * Bp + (size_t)(tb) * (32 * kQ.d[Q]) + kQ.i[Q] * 32)
#define OFF_OF(Q0, Q1, c, tb) (OFF_Q(Q0, c, tb) + ((OFF_Q(Q1, c, tb) - OFF_Q(Q0, c, tb)) & hmask))   // arithmetic select (mask, not a quarter-rate multiply): one load, no exec-masked pair
    // Pair-slot addressing of the stores (kPS): byte address = buffer + 4 * row_off(row, c, tb) [scalar] + 128 * i [immediate] + lane offset [one of three VGPRs]
    const unsigned lane_next = (unsigned)lane * 4u;                                                     // second half-wave: the next 128-byte block
    const unsigned lane_row3 = (unsigned)jn * 4u + (h ? 12u * (unsigned)C * (unsigned)p.Bp : 0u);       // ... the same block one row down, d = 3
    const unsigned lane_row5 = (unsigned)jn * 4u + (h ? 20u * (unsigned)C * (unsigned)p.Bp : 0u);       // ... d = 5   (< 2^32: checked by the launcher)
    auto row_off = [&](auto rc, int c, int tb) -> size_t {        // float offset of row `rc` (see PairSlots::row) of channel c, keypoint tile tb: wave-uniform
        constexpr int R = decltype(rc)::value;
        return ((size_t)(kPS.row_alpha[R] * C) + (size_t)(c * kPS.row_d[R])) * Bp + (size_t)tb * (32 * kPS.row_d[R]);
    };
    auto slot_ptr = [&](auto sc, const void *buf, int c, int tb) -> const char * {      // address of pair slot `sc`'s two members for this lane
        constexpr int S = decltype(sc)::value;
        const char *rb = reinterpret_cast<const char *>(buf) + 4 * row_off(std::integral_constant<int, kPS.row[S]>{}, c, tb) + 128 * kPS.i[S];
        return rb + (kPS.kind[S] == 1 ? (kPS.d[S] == 3 ? lane_row3 : lane_row5) : lane_next);
    };

    // software pipeline: the 30 coefficient rows of the NEXT column tile are requested before the 124 MFMAs of the current one
    constexpr int NCV = IN_SPATIAL ? 32 : (SPLIT != 0 ? 32 : 30);
    constexpr bool PACK_OUT = SPLIT == 2 && !OUT_SPATIAL && !OUT_ROWS;      // coefficients leave as fp16 hi/lo pairs under the keypoint's block scale
    // group-domain tensors are float32 or (x_bf16) bfloat16 as stored (BASELINE config 5)
    auto ld_sp = [&](const float *base, size_t i) -> float {
        return p.x_bf16 ? __uint_as_float((unsigned)reinterpret_cast<const unsigned short *>(base)[i] << 16) : base[i];
    };
    float cn[NCV + 1];                                           // slot NCV: the keypoint's bound (PACK_OUT), prefetched with the coefficients
    auto load_coefs = [&](int tile, float (&dst)[NCV + 1]) {
        const int c = (IN_SPATIAL || OUT_SPATIAL) ? tile % C : tile / p.tiles_per_c;      // group-domain tensors are [b][c][.]: channel-fastest tiles
        const int tb = (IN_SPATIAL || OUT_SPATIAL) ? tile / C : tile - c * p.tiles_per_c;         // make the waves of a workgroup touch adjacent rows
        if constexpr (PACK_OUT) dst[NCV] = p.out_bound[tb * 32 + jn];
        if constexpr (OUT_SPATIAL && SPLIT == 2) { if (p.sp_pack) dst[NCV] = p.out_bound[tb * 32 + jn]; }
        if constexpr (IN_SPATIAL) {
            // group-domain input [b][c][60]: ONE keypoint's contiguous 240-byte row per load instruction (lane = group element), slot i = the
            // tile's keypoint i; process() transposes through the wave's LDS buffer.  (Lane = keypoint reads -- 64 addresses 30 KB apart per
            // instruction -- ran at 2.7 TB/s.)
            const int b0 = tb * 32;
            // (bfloat16 storage: the raw 16 bits are loaded here and widened where the tile is consumed -- with the conversion at the load the
            //  compiler put s_waitcnt vmcnt(0) behind every one of the 32 row loads, i.e. the prefetch was 32 serialised round trips)
            if (p.x_bf16) {
                const unsigned short *xs16 = reinterpret_cast<const unsigned short *>(p.x_spatial);
                static_for<32>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    const size_t src = ((size_t)(b0 + i < B ? b0 + i : B - 1) * C + c) * ROREG_G;
                    dst[i] = __uint_as_float(lane < ROREG_G ? (unsigned)xs16[src + lane] : 0u);
                });
            } else {
                static_for<32>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    const size_t src = ((size_t)(b0 + i < B ? b0 + i : B - 1) * C + c) * ROREG_G;
                    dst[i] = lane < ROREG_G ? p.x_spatial[src + lane] : 0.f;
                });
            }
        } else if constexpr (SPLIT != 0) {
            static_for<32>([&](auto ic) {                     // slot st*8+e holds coefficient q = 16 st + 8 h + e
                constexpr int st = decltype(ic)::value / 8, e = decltype(ic)::value % 8;
                constexpr int q0 = 16 * st + e, q1 = 16 * st + 8 + e;
                // (streamed once: non-temporal loads and stores -- the transforms' time 310.1 -> 307.2 ms per step on one box; the same hint on the
                //  GEMM epilogue's stores and residual loads changed nothing)
                if constexpr (q1 < ROREG_G) dst[st * 8 + e] = __builtin_nontemporal_load(p.Xin + OFF_OF(q0, q1, c, tb) + jn);
                else dst[st * 8 + e] = __builtin_nontemporal_load(p.Xin + OFF_Q(q0, c, tb) + jn);      // q1 does not exist: the second half-wave's copy is zeroed at use
            });
        } else {
            static_for<30>([&](auto ic) {
                constexpr int s = decltype(ic)::value;
                constexpr int q0 = 2 * s, q1 = 2 * s + 1;
                dst[s] = p.Xin[OFF_OF(q0, q1, c, tb) + jn];
            });
        }
    };
    // six bf16 MFMAs = one f32-accurate product of the split fragments (a1+a2+a3)(b1+b2+b3), terms of order <= 4
    auto mfma6 = [&](const bf16x8 *a, const bf16x8 &b1, const bf16x8 &b2, const bf16x8 &b3, f32x16 c) {
        const bf16x8 a1 = a[0], a2 = a[64], a3 = a[128];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, c, 0, 0, 0);
        return c;
    };
    // fp16 x 2: three MFMAs per product, and because every column (keypoint) of these transforms is independent, the block scale is
    // PER COLUMN: 2^e from the column's own maximum (this lane's 32 values and its partner half-wave's), undone on this lane's accumulators
    auto mfma3 = [&](const f16x8 *a, const f16x8 &bh, const f16x8 &bl, f32x16 c) {
        const f16x8 ah = a[0], al = a[64];
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
        return c;
    };
    auto column_scale = [&](float mx, float &xscale, float &oscale) {
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        int e = 0;
        if (mx > 0.f && mx < __builtin_inff()) { int ex; (void)frexpf(mx, &ex); e = 14 - ex; }
        xscale = ldexpf(1.f, e); oscale = ldexpf(1.f, -(e + p.f_exp));
    };
    // Pipeline with two register sets (no copies): while a tile is computed from one set, the coefficients of the wave's next tile
    // are in flight into the other one.  VMEM operations of a wave complete in order, so a tile only waits for loads that are a whole
    // tile old (the stores and loads issued after them stay in flight).  The explicit wait after the first loads gives the loop ONE
    // wait state on both entry paths -- otherwise the compiler merges "first loads just issued" with the steady state and drains the
    // prefetch at every tile.  A wave with an odd number of tiles processes its last tile twice (idempotent stores): branch-free.
    auto process = [&](int tile, int next_tile, float (&cv)[NCV + 1], float (&cnext)[NCV + 1]) {
        asm volatile("" ::: "memory");      // keep the transform fragments in LDS: without this the compiler hoists all 62 of them into VGPRs
        const int c = (IN_SPATIAL || OUT_SPATIAL) ? tile % C : tile / p.tiles_per_c;
        const int tbi = (IN_SPATIAL || OUT_SPATIAL) ? tile / C : tile - c * p.tiles_per_c;
        const int b = tbi * 32 + jn;
        const bool valid = b < B;
        const int bb = valid ? b : B - 1;
        f32x16 v[2];
        float wmax = 0.f;                                        // max |value written| for this lane's keypoint (group-domain output)
        if (IN_SPATIAL) {
            load_coefs(next_tile, cnext);                 // the next tile's rows are in flight while this one is transformed
            __builtin_amdgcn_sched_barrier(0);
            float *ti = sT + (threadIdx.x >> 6) * (32 * 65);
#pragma unroll
            for (int i = 0; i < 32; ++i) ti[i * 65 + lane] = p.x_bf16 ? __uint_as_float(__float_as_uint(cv[i]) << 16) : cv[i];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the buffer is this wave's own: no barrier)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int g = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    v[t][r] = g < ROREG_G ? ti[jn * 65 + g] : 0.f;
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // read out before OUT_ROWS reuses the buffer
        } else {
            load_coefs(next_tile, cnext);
            __builtin_amdgcn_sched_barrier(0);          // the prefetch stays ahead of this tile's MFMAs (the scheduler would sink it to save registers)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) v[t][r] = 0.f;
            if constexpr (SPLIT == 3) {
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    float x8[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) x8[e] = (16 * st + 8 + e < ROREG_G || h == 0) ? cv[st * 8 + e] : 0.f;
                    bf16x8 b1, b2, b3;
                    split3(x8, b1, b2, b3);
                    v[0] = mfma6(sA1s + ((st * 2 + 0) * 3) * 64 + lane, b1, b2, b3, v[0]);
                    v[1] = mfma6(sA1s + ((st * 2 + 1) * 3) * 64 + lane, b1, b2, b3, v[1]);
                }
            } else if constexpr (SPLIT == 2) {
                float mx = 0.f;
#pragma unroll
                for (int s = 0; s < 32; ++s) mx = fmaxf(mx, ((16 * (s >> 3) + 8 + (s & 7) < ROREG_G || h == 0) ? fabsf(cv[s]) : 0.f));
                float xsc, osc;
                column_scale(mx, xsc, osc);
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    float x8[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) x8[e] = (16 * st + 8 + e < ROREG_G || h == 0) ? cv[st * 8 + e] : 0.f;
                    f16x8 bh, bl;
                    split2(x8, xsc, bh, bl);
                    v[0] = mfma3(sA1h + ((st * 2 + 0) * 2) * 64 + lane, bh, bl, v[0]);
                    v[1] = mfma3(sA1h + ((st * 2 + 1) * 2) * 64 + lane, bh, bl, v[1]);
                }
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[t][r] *= osc;
            } else {
#pragma unroll
                for (int s = 0; s < 30; ++s) {
                    v[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(sA1[(s * 2 + 0) * 64 + lane], cv[s], v[0], 0, 0, 0);
                    v[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(sA1[(s * 2 + 1) * 64 + lane], cv[s], v[1], 0, 0, 0);
                }
            }
        }
        // ---- group-domain epilogue --------------------------------------------------------------------------------
        const float bsum = sBias[c];
        const bool bn = p.bn_scale != nullptr;
        const float sc = sScale[c], sh = sShift[c];
        const bool has_rs = OUT_SPATIAL && p.r_spatial;
        // Without BatchNorm, column map and tracked maximum the residual is simply added to what is written: it is then read in the store
        // loop below, one keypoint's contiguous row per load (lane = group element), eight rows at a time -- not here with lane = keypoint,
        // 64 scattered dwords per load and an s_waitcnt vmcnt(0) per element (32 serialised round trips per tile, which also drained the
        // next tile's prefetch).  (v + bias) + residual either way.
        const bool rs_late = has_rs && !bn && !p.g_map && !p.out_rowmax;
        const size_t rs = ((size_t)bb * C + c) * ROREG_G;
        float *tb = (OUT_SPATIAL || OUT_ROWS) ? sT + (threadIdx.x >> 6) * (32 * 65) : nullptr;      // this wave's [32 keypoints][65] transpose buffer
        const bool sp_pack = OUT_SPATIAL && SPLIT == 2 && p.sp_pack;
        const float pk_scale = sp_pack ? ldexpf(1.f, bound_exp(cv[NCV])) : 1.f;      // the keypoint's block scale (the consumer derives the same exponent from the same bound)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int g = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                float x = v[t][r] + bsum;
                if (OUT_SPATIAL && has_rs && !rs_late && g < ROREG_G) x += ld_sp(p.r_spatial, rs + g);
                if (sp_pack && p.raw_col && g == p.raw_g && valid) p.raw_col[(size_t)bb * C + c] = x;
                if (bn) x = fmaxf(fmaf(x, sc, sh), 0.f);
                if (g >= ROREG_G || !valid) x = 0.f;          // pad keypoints carry zeros through the forward transform: their coefficients are exact 0
                v[t][r] = x;
                if (OUT_SPATIAL && g < ROREG_G) {
                    const int go = sGmap[g];
                    if (go >= 0) {
                        if (sp_pack) {                            // word = fp16(x 2^e) | fp16(x 2^e - hi) << 16
                            const float xs = x * pk_scale;
                            const _Float16 h1 = (_Float16)xs;
                            const _Float16 l1 = (_Float16)(xs - (float)h1);
                            tb[jn * 65 + go] = __uint_as_float((unsigned)__builtin_bit_cast(unsigned short, h1) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16));
                        } else {
                            tb[jn * 65 + go] = x; wmax = fmaxf(wmax, fabsf(x));      // (pad keypoints were zeroed above)
                        }
                    }
                }
            }
        if (OUT_SPATIAL) {
            // lanes own keypoints, but the output is [b][c][column]: go through LDS so that every store instruction writes one
            // keypoint's contiguous row of Lout floats instead of 64 scattered dwords
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int b0 = tbi * 32;
            if (b0 + 32 <= B) {
                // interior tile: eight rows per group -- their LDS reads go out together instead of one exposed LDS round trip per row
                // (the rolled loop below waits lgkmcnt(0) 32 times per tile)
                float *ob = p.out_spatial + ((size_t)b0 * C + c) * p.Lout + lane;
                const size_t ostep = (size_t)C * p.Lout;
                const size_t rb = ((size_t)b0 * C + c) * ROREG_G + (lane < ROREG_G ? lane : 0), rstep = (size_t)C * ROREG_G;
                const unsigned short *r16 = reinterpret_cast<const unsigned short *>(p.r_spatial);
#pragma unroll
                for (int g8 = 0; g8 < 4; ++g8) {
                    float tv[8], rv[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) tv[q] = tb[(g8 * 8 + q) * 65 + lane];
                    if (rs_late) {                       // (the storage type is tested once per group, not per load: see load_coefs)
                        if (p.x_bf16) {
#pragma unroll
                            for (int q = 0; q < 8; ++q) rv[q] = __uint_as_float((unsigned)r16[rb + (size_t)(g8 * 8 + q) * rstep]);
#pragma unroll
                            for (int q = 0; q < 8; ++q) tv[q] += __uint_as_float(__float_as_uint(rv[q]) << 16);
                        } else {
#pragma unroll
                            for (int q = 0; q < 8; ++q) rv[q] = p.r_spatial[rb + (size_t)(g8 * 8 + q) * rstep];
#pragma unroll
                            for (int q = 0; q < 8; ++q) tv[q] += rv[q];
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        if (lane < p.Lout) ob[(size_t)(g8 * 8 + q) * ostep] = lane < p.Lvalid ? tv[q] : 0.f;
                }
            } else {
                for (int bl = 0; bl < 32 && b0 + bl < B; ++bl)
                    if (lane < p.Lout) {
                        float tv = tb[bl * 65 + lane];
                        if (rs_late && lane < ROREG_G) tv += ld_sp(p.r_spatial, ((size_t)(b0 + bl) * C + c) * ROREG_G + lane);
                        p.out_spatial[((size_t)(b0 + bl) * C + c) * p.Lout + lane] = lane < p.Lvalid ? tv : 0.f;
                    }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (p.out_rowmax) {
                // every lane issues the atomic (no branch around a VMEM operation in the pipelined loop): both half-waves hold their
                // keypoint's maximum, pad keypoints contribute 0 to the last valid one -- never a shared dump word, which would serialise
                // every wave of the chip on one L2 atomic unit
                wmax = fmaxf(wmax, __shfl_xor(wmax, 32));
                atomicMax(reinterpret_cast<unsigned *>(p.out_rowmax) + bb, __float_as_uint(valid ? wmax : 0.f));
            }
        }
        if (!OUT_SPATIAL) {
            f32x16 o[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
            if constexpr (SPLIT == 3) {
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    float x8[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) x8[e] = v[st >> 1][8 * (st & 1) + e];
                    bf16x8 b1, b2, b3;
                    split3(x8, b1, b2, b3);
                    o[0] = mfma6(sA2s + ((st * 2 + 0) * 3) * 64 + lane, b1, b2, b3, o[0]);
                    o[1] = mfma6(sA2s + ((st * 2 + 1) * 3) * 64 + lane, b1, b2, b3, o[1]);
                }
            } else if constexpr (SPLIT == 2) {
                float mx = 0.f;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, fabsf(v[t][r]));
                float xsc, osc;
                column_scale(mx, xsc, osc);
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    float x8[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) x8[e] = v[st >> 1][8 * (st & 1) + e];
                    f16x8 bh, bl;
                    split2(x8, xsc, bh, bl);
                    o[0] = mfma3(sA2h + ((st * 2 + 0) * 2) * 64 + lane, bh, bl, o[0]);
                    o[1] = mfma3(sA2h + ((st * 2 + 1) * 2) * 64 + lane, bh, bl, o[1]);
                }
                if constexpr (PACK_OUT) {
                    // undo the transform's scale and apply the keypoint's block scale in one exact multiplication, then split:
                    // word = fp16(x) | fp16(x - hi) << 16  (what the GEMM's staging reads)
                    const float psc = osc * ldexpf(1.f, bound_exp(cv[NCV]));
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float x = o[t][r] * psc;
                            const _Float16 h1 = (_Float16)x;
                            const _Float16 l1 = (_Float16)(x - (float)h1);
                            o[t][r] = __uint_as_float((unsigned)__builtin_bit_cast(unsigned short, h1) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16));
                        }
                } else {
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[t][r] *= osc;
                }
            } else {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int s = t * 16 + r;
                        o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(sA2[(s * 2 + 0) * 64 + lane], v[t][r], o[0], 0, 0, 0);
                        o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(sA2[(s * 2 + 1) * 64 + lane], v[t][r], o[1], 0, 0, 0);
                    }
            }
            if constexpr (OUT_ROWS) {
                // per-keypoint layout [b][c][q]: through the wave's transpose buffer, one keypoint's 60 coefficients per store instruction
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int qq = h ? kPS.qb[t * 16 + r] : kPS.qa[t * 16 + r];      // canonical coefficient index of this lane's row (64: none)
                        if (qq < ROREG_G) tb[jn * 65 + qq] = o[t][r];
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const int b0 = tbi * 32;
                if (b0 + 32 <= B) {
                    float *ob = p.out_spatial + ((size_t)b0 * C + c) * ROREG_G + lane;
                    const size_t ostep = (size_t)C * ROREG_G;
#pragma unroll
                    for (int g8 = 0; g8 < 4; ++g8) {
                        float tv[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) tv[q] = tb[(g8 * 8 + q) * 65 + lane];
#pragma unroll
                        for (int q = 0; q < 8; ++q)
                            if (lane < ROREG_G) ob[(size_t)(g8 * 8 + q) * ostep] = tv[q];
                    }
                } else {
                    for (int bl = 0; bl < 32 && b0 + bl < B; ++bl)
                        if (lane < ROREG_G) p.out_spatial[((size_t)(b0 + bl) * C + c) * ROREG_G + lane] = tb[bl * 65 + lane];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else if (PACK_OUT && p.out_planes) {
                // Half-block layout for the LDS-DMA GEMM: every 32-column block of a row (128 bytes in either layout) holds 16 words of fp16
                // hi values, then 16 words of lo values; word w pairs the block's columns w and w + 16.  The two lanes of a pair (jn, jn + 16:
                // the same coefficient of two keypoints) exchange their hi | lo << 16 words with one v_permlane16_swap; the lower lane
                // stores hi(w) | hi(w + 16) << 16 at word w, the upper lane lo(w) | lo(w + 16) << 16 at word 16 + w: ONE 4-byte store per lane
                // and coefficient to consecutive addresses, exactly the word layout's access pattern.
                const int tb = tbi;
                const unsigned sel = (jn & 16) ? 0x07060302u : 0x05040100u;    // v_perm(S0 = column w + 16, S1 = column w): the lo or the hi halves
                static_for<32>([&](auto ic) {
                    constexpr int S = decltype(ic)::value, t = S / 16, r = S % 16;          // pair slot S = registers o[t][r] of the two half-waves
                    const unsigned own = __float_as_uint(o[t][r]);
                    // swaps the odd 16-lane rows of the first operand with the even rows of the second: [0] = the pair's lower lane's
                    // word, [1] = the upper lane's word, in both lanes of the pair
                    const auto pr = __builtin_amdgcn_permlane16_swap(own, own, false, false);
                    const unsigned word = __builtin_amdgcn_perm(pr[1], pr[0], sel);
                    unsigned *dst = reinterpret_cast<unsigned *>(const_cast<char *>(slot_ptr(ic, p.Xout, c, tb)));
                    if (kPS.kind[S] != 2 || h == 0) __builtin_nontemporal_store(word, dst);          // (no member b: the second half-wave is masked off, the store is still issued)
                });
            } else {
                const int tb = tbi;                           // pad keypoints (b >= B) get zeros: the buffers stay fully defined
                static_for<32>([&](auto ic) {
                    constexpr int S = decltype(ic)::value, t = S / 16, r = S % 16;
                    float *dst = reinterpret_cast<float *>(const_cast<char *>(slot_ptr(ic, p.Xout, c, tb)));
                    if (kPS.kind[S] != 2 || h == 0) __builtin_nontemporal_store(o[t][r], dst);
                });
            }
        }
    };
    if (wave_global < n_tiles) {
        const int last = wave_global + ((n_tiles - 1 - wave_global) / n_waves) * n_waves;        // this wave's last tile
        float cb[NCV + 1];
        load_coefs(wave_global, cn);
        __builtin_amdgcn_s_waitcnt(0x0F70);                       // vmcnt(0)
        for (int tile = wave_global; tile <= last; tile += 2 * n_waves) {
            const int t1 = min(tile + n_waves, last), t2 = min(tile + 2 * n_waves, last);
            process(tile, t1, cn, cb);
            process(t1, t2, cb, cn);
        }
    }
}

#undef OFF_OF
#undef OFF_Q

float *g_A1 = nullptr, *g_A2 = nullptr;
uint16_t *g_A1s = nullptr, *g_A2s = nullptr, *g_A1h = nullptr, *g_A2h = nullptr;
int g_f_exp = 0;

}  // namespace

extern "C" int roreg_set_fourier_tables(const float *F_host /* [60 (q)][60 (g)], orthonormal */) {
    ROREG_REQUIRE(F_host, "roreg_set_fourier_tables: null table");
    // A1[s][tile][lane] = F[q = 2s + (lane>>5)][g = tile*32 + (lane&31)]   (inverse transform: x(g) = sum_q F[q][g] coef[q])
    // A2[(t,r)][tile][lane] = F[q' = slot_out_q(tile*32 + (lane&31))][g = t*32 + (r&3) + 8(r>>2) + 4(lane>>5)]
    static float A1[30 * 2 * 64], A2[32 * 2 * 64];
    for (int s = 0; s < 30; ++s)
        for (int tile = 0; tile < 2; ++tile)
            for (int lane = 0; lane < 64; ++lane) {
                const int qq = 2 * s + (lane >> 5), g = tile * 32 + (lane & 31);
                A1[(s * 2 + tile) * 64 + lane] = g < 60 ? F_host[qq * 60 + g] : 0.f;
            }
    for (int t = 0; t < 2; ++t)
        for (int r = 0; r < 16; ++r)
            for (int tile = 0; tile < 2; ++tile)
                for (int lane = 0; lane < 64; ++lane) {
                    const int qq = slot_out_q(tile * 32 + (lane & 31)), g = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    A2[((t * 16 + r) * 2 + tile) * 64 + lane] = (qq < 60 && g < 60) ? F_host[qq * 60 + g] : 0.f;
                }
    if (!g_A1) {
        if (hipMalloc(&g_A1, sizeof(A1)) != hipSuccess || hipMalloc(&g_A2, sizeof(A2)) != hipSuccess) {
            roreg::set_error("roreg_set_fourier_tables: hipMalloc failed");
            return 1;
        }
    }
    if (hipMemcpy(g_A1, A1, sizeof(A1), hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(g_A2, A2, sizeof(A2), hipMemcpyHostToDevice) != hipSuccess) {
        roreg::set_error("roreg_set_fourier_tables: hipMemcpy failed");
        return 1;
    }
    // the same tables as 3 x bf16 split fragments of the K=16 MFMA: [step st][tile][plane][lane][e]
    //   A1s: F[q = 16 st + 8 h + e][g = tile*32 + j]                                  (inverse)
    //   A2s: F[q' = slot_out_q(tile*32 + j)][g = 32 (st>>1) + (r&3) + 8 (r>>2) + 4 h], r = 8 (st&1) + e   (forward; K order = accumulator registers)
    static uint16_t A1s[4 * 2 * 3 * 64 * 8], A2s[4 * 2 * 3 * 64 * 8];
    auto split3_host = [](float x, uint16_t out[3]) {                  // round-to-nearest-even pieces of the exact remainders
        float rem = x;
        for (int sp = 0; sp < 3; ++sp) {
            uint32_t u; memcpy(&u, &rem, 4);
            const uint32_t r = ((u >> 16) & 1u) + 0x7fffu;
            const uint16_t hi = (uint16_t)((u + r) >> 16);
            out[sp] = hi;
            const uint32_t back = (uint32_t)hi << 16;
            float v; memcpy(&v, &back, 4);
            rem = rem - v;
        }
    };
    for (int st = 0; st < 4; ++st)
        for (int tile = 0; tile < 2; ++tile)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int j = lane & 31, h = lane >> 5;
                    const int q1 = 16 * st + 8 * h + e, g1 = tile * 32 + j;
                    const float f1 = (q1 < 60 && g1 < 60) ? F_host[q1 * 60 + g1] : 0.f;
                    const int r = 8 * (st & 1) + e, q2 = slot_out_q(tile * 32 + j), g2 = 32 * (st >> 1) + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const float f2 = (q2 < 60 && g2 < 60) ? F_host[q2 * 60 + g2] : 0.f;
                    uint16_t p1[3], p2[3];
                    split3_host(f1, p1); split3_host(f2, p2);
                    for (int sp = 0; sp < 3; ++sp) {
                        const size_t at = ((((size_t)st * 2 + tile) * 3 + sp) * 64 + lane) * 8 + e;
                        A1s[at] = p1[sp]; A2s[at] = p2[sp];
                    }
                }
    if (!g_A1s) {
        if (hipMalloc(&g_A1s, sizeof(A1s)) != hipSuccess || hipMalloc(&g_A2s, sizeof(A2s)) != hipSuccess) {
            roreg::set_error("roreg_set_fourier_tables: hipMalloc failed");
            return 1;
        }
    }
    if (hipMemcpy(g_A1s, A1s, sizeof(A1s), hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(g_A2s, A2s, sizeof(A2s), hipMemcpyHostToDevice) != hipSuccess) {
        roreg::set_error("roreg_set_fourier_tables: hipMemcpy failed");
        return 1;
    }
    // ... and as fp16 hi/lo fragments under the power-of-two scale 2^f_exp (|F| * 2^f_exp <= 2^14): [st][tile][2 planes][lane][e]
    static uint16_t A1h[4 * 2 * 2 * 64 * 8], A2h[4 * 2 * 2 * 64 * 8];
    float fmx = 0.f;
    for (int i = 0; i < 3600; ++i) fmx = std::max(fmx, std::fabs(F_host[i]));
    int fex = 0;
    (void)std::frexp(fmx, &fex);
    g_f_exp = fmx > 0.f ? 14 - fex : 0;
    auto f2h = [](float x) -> uint16_t {                                  // float -> fp16 bits, round-to-nearest-even (values are normal or zero here)
        _Float16 hv = (_Float16)x;
        uint16_t b; memcpy(&b, &hv, 2);
        return b;
    };
    auto h2f = [](uint16_t b) -> float { _Float16 hv; memcpy(&hv, &b, 2); return (float)hv; };
    for (int st = 0; st < 4; ++st)
        for (int tile = 0; tile < 2; ++tile)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int j = lane & 31, h = lane >> 5;
                    const int q1 = 16 * st + 8 * h + e, g1 = tile * 32 + j;
                    const float f1 = (q1 < 60 && g1 < 60) ? std::ldexp(F_host[q1 * 60 + g1], g_f_exp) : 0.f;
                    const int r = 8 * (st & 1) + e, q2 = slot_out_q(tile * 32 + j), g2 = 32 * (st >> 1) + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const float f2 = (q2 < 60 && g2 < 60) ? std::ldexp(F_host[q2 * 60 + g2], g_f_exp) : 0.f;
                    const uint16_t h1 = f2h(f1), h2 = f2h(f2);
                    const size_t hi_at = ((((size_t)st * 2 + tile) * 2 + 0) * 64 + lane) * 8 + e, lo_at = hi_at + 64 * 8;
                    A1h[hi_at] = h1; A1h[lo_at] = f2h(f1 - h2f(h1));
                    A2h[hi_at] = h2; A2h[lo_at] = f2h(f2 - h2f(h2));
                }
    if (!g_A1h) {
        if (hipMalloc(&g_A1h, sizeof(A1h)) != hipSuccess || hipMalloc(&g_A2h, sizeof(A2h)) != hipSuccess) {
            roreg::set_error("roreg_set_fourier_tables: hipMalloc failed");
            return 1;
        }
    }
    if (hipMemcpy(g_A1h, A1h, sizeof(A1h), hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(g_A2h, A2h, sizeof(A2h), hipMemcpyHostToDevice) != hipSuccess) {
        roreg::set_error("roreg_set_fourier_tables: hipMemcpy failed");
        return 1;
    }
    return 0;
}

// Work list of one layer.  Workgroup b is observed to run on XCD b % 8 (speed only, never correctness), and each XCD has a private
// 4 MB L2: the list is built as eight per-XCD streams, each a sequence of 8 x 8 (m-tile x n-tile) blocks, interleaved so that entry
// 8*i + k belongs to stream k.  The ~64 workgroups resident on an XCD then share 8 weight slices and 8 X tiles (minimum of
// |X|*mts/a + |W|*nts/b under a*b = 64), instead of every XCD streaming every X tile.  Streams are balanced by (tiles x irrep dim).
extern "C" size_t roreg_irrep_gemm_tiles_m(int O, int B, int tile_m, int32_t *tiles_host);
extern "C" size_t roreg_irrep_gemm_tiles(int O, int B, int32_t *tiles_host /* nullable; [n][3] */) { return roreg_irrep_gemm_tiles_m(O, B, 128, tiles_host); }

extern "C" size_t roreg_irrep_gemm_tiles_m(int O, int B, int tile_m /* 128 | 256 */, int32_t *tiles_host) {
    static const int dims[5] = {1, 3, 3, 4, 5};
    struct Block { int r, mg, ng, cost; };
    std::vector<Block> blocks;
    for (int r = 4; r >= 0; --r) {
        const int d = dims[r];
        const int mts = round_up(d * O, tile_m) / tile_m, nts = (d * B + 255) / 256;
        for (int ng = 0; ng * 8 < nts; ++ng)
            for (int mg = 0; mg * 8 < mts; ++mg) {
                const int a = std::min(8, mts - mg * 8), b = std::min(8, nts - ng * 8);
                blocks.push_back({r, mg, ng, a * b * d});
            }
    }
    std::stable_sort(blocks.begin(), blocks.end(), [](const Block &x, const Block &y) { return x.cost > y.cost; });
    std::vector<std::vector<int>> stream(8);
    long long load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (const Block &bk : blocks) {
        int k = 0;
        for (int q = 1; q < 8; ++q)
            if (load[q] < load[k]) k = q;
        load[k] += bk.cost;
        const int d = dims[bk.r];
        const int mts = round_up(d * O, tile_m) / tile_m, nts = (d * B + 255) / 256;
        for (int nt = bk.ng * 8; nt < std::min(nts, bk.ng * 8 + 8); ++nt)
            for (int mt = bk.mg * 8; mt < std::min(mts, bk.mg * 8 + 8); ++mt) {
                stream[k].push_back(bk.r); stream[k].push_back(mt); stream[k].push_back(nt);
            }
    }
    size_t longest = 0;
    for (int k = 0; k < 8; ++k) longest = std::max(longest, stream[k].size() / 3);
    const size_t n = longest * 8;
    if (tiles_host)
        for (size_t i = 0; i < longest; ++i)
            for (int k = 0; k < 8; ++k) {
                int32_t *t = tiles_host + (i * 8 + k) * 3;
                if (i * 3 < stream[k].size()) { t[0] = stream[k][i * 3]; t[1] = stream[k][i * 3 + 1]; t[2] = stream[k][i * 3 + 2]; }
                else { t[0] = -1; t[1] = 0; t[2] = 0; }          // padding entry: the workgroup exits immediately
            }
    return n;
}

extern "C" int roreg_irrep_gemm(const float *const *X, float *const *Out, const float *const *Add, const float *const *Wpack, int C, int O, int B,
                                const int32_t *tiles_dev, int n_tiles, void *stream) {
    ROREG_REQUIRE(X && Out && Wpack && tiles_dev && C > 0 && O > 0 && B > 0 && n_tiles > 0, "roreg_irrep_gemm: bad arguments");
    ROREG_REQUIRE(C % 32 == 0, "roreg_irrep_gemm: C must be a multiple of 32 (got %d)", C);
    ROREG_REQUIRE(B % 4 == 0, "roreg_irrep_gemm: B must be a multiple of 4 (got %d); pad the keypoint batch", B);
    static const int dims[5] = {1, 3, 3, 4, 5};
    GemmDescs p;
    for (int r = 0; r < 5; ++r) {
        p.X[r] = X[r]; p.Out[r] = Out[r]; p.Add[r] = Add ? Add[r] : nullptr; p.W[r] = reinterpret_cast<const float4 *>(Wpack[r]);
        p.K[r] = dims[r] * C; p.M[r] = dims[r] * O; p.Mpad[r] = round_up(dims[r] * O, 128); p.N[r] = dims[r] * B;
    }
    constexpr int CT = 32;
    const size_t lds = 2 * CT * 256 * sizeof(float);
    auto kern = irrep_gemm_kernel<CT>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { roreg::set_error("roreg_irrep_gemm: hipFuncSetAttribute: %s", hipGetErrorString(e)); return 1; }
    hipLaunchKernelGGL(kern, dim3(n_tiles), dim3(256), lds, roreg::as_stream(stream), p, tiles_dev);
    ROREG_CHECK_LAUNCH("roreg_irrep_gemm");
    return 0;
}

// ---- the persistent GEMM's host side: switch, CU count ----
#define ROREG_GEMM_PERSIST_DEFAULT 0
static std::atomic<int> g_gemm_persist{-1};                  // -1: not decided yet (environment at first use); 0 per tile, 1 persistent, 2 half tiles
static int gemm_launch_form() {
    int v = g_gemm_persist.load(std::memory_order_relaxed);
    if (v < 0) {
        const char *e = getenv("ROREG_GEMM_PERSIST");
        v = e ? (e[0] == '1' ? 1 : e[0] == '2' ? 2 : 0) : ROREG_GEMM_PERSIST_DEFAULT;
        g_gemm_persist.store(v, std::memory_order_relaxed);
    }
    return v;
}
extern "C" int roreg_gemm_persistent(int on) {
    const int prev = gemm_launch_form();
    if (on >= 0 && on <= 2) g_gemm_persist.store(on, std::memory_order_relaxed);
    return prev;
}
static int gemm_cu_count() {
    static const int n = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        return cus;
    }();
    return n;
}
template <int NP, int WO>
static int launch_gemm_split(const char *what, const float *const *X, float *const *Out, const float *const *Add, const void *const *Wsplit,
                             const float *xbound, int w_exp, const float *nb_u, const float *nb_v, float *out_bound, int C, int O, int B,
                             const int32_t *tiles_dev, int n_tiles, void *stream, int x_planes = 0) {
    static const int dims[5] = {1, 3, 3, 4, 5};
    GemmSplitDescs p;
    for (int r = 0; r < 5; ++r) {
        p.X[r] = X[r]; p.Out[r] = Out[r]; p.Add[r] = Add ? Add[r] : nullptr; p.W[r] = Wsplit[r];
        p.K[r] = dims[r] * C; p.M[r] = dims[r] * O; p.Mpad[r] = round_up(dims[r] * O, 128); p.N[r] = dims[r] * B;
        if (WO == 4 && p.Mpad[r] % 256 != 0) { roreg::set_error("%s: tile_m = 256 needs O %% 256 == 0 (got %d)", what, O); return 2; }
    }
    p.xbound = xbound; p.w_exp = w_exp; p.nb_u = nb_u; p.nb_v = nb_v; p.out_bound = out_bound; p.O = O;
    constexpr int CT = 32;
    const size_t lds = 2 * (NP * 2 * 256 + NP * (WO * 64) * 2) * 16;     // two buffers of (activation planes + weight fragments) of a K16 step
    if (x_planes) {
        if constexpr (NP == 2 && WO == 4) {
            // activations in half-block layout (ft_nonlin out_planes), delivered by LDS-DMA: irrep_gemm_xdma_kernel
            const size_t lds_x = 3 * (2 * 16 * 256 * 2) + 2 * (2 * 2 * 256) * 16;      // three activation stages + two weight stages = 80 KB
            // x_planes == 2: the 16x16x32 kernel (K = 32 steps, the whole LDS as rings); 1: the 32x32x16 kernel (bitwise the register-staged one)
            const bool mfma16 = x_planes == 2;
            const size_t lds_use = mfma16 ? (size_t)(6 + 4) * 16384 : lds_x;
            auto kx = mfma16 ? ((long long)C * O == 256ll * 512 ? irrep_gemm_xdma16_kernel<1> : irrep_gemm_xdma16_kernel<0>)
                             : ((long long)C * O == 256ll * 512 ? irrep_gemm_xdma_kernel<1> : irrep_gemm_xdma_kernel<0>);
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kx), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_use);
            if (e != hipSuccess) { roreg::set_error("%s: hipFuncSetAttribute: %s", what, hipGetErrorString(e)); return 1; }
            if (mfma16 && gemm_launch_form() == 2) {
                auto kh = (long long)C * O == 256ll * 512 ? irrep_gemm_xdma16h_kernel<1> : irrep_gemm_xdma16h_kernel<0>;
                const size_t lds_h = (size_t)6 * 8192 + 2 * 16384;      // six 128-column activation stages + one K32 step of weights = 80 KB: two workgroups per CU
                e = hipFuncSetAttribute(reinterpret_cast<const void *>(kh), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_h);
                if (e != hipSuccess) { roreg::set_error("%s: hipFuncSetAttribute: %s", what, hipGetErrorString(e)); return 1; }
                hipLaunchKernelGGL(kh, dim3(2 * n_tiles), dim3(256), lds_h, roreg::as_stream(stream), p, tiles_dev);
            } else if (mfma16 && gemm_launch_form() == 1) {
                const bool big = (long long)C * O == 256ll * 512;
                static const int ra = [] { const char *v = getenv("ROREG_GEMM_PERSIST_RA"); return v ? atoi(v) : 2; }();
                auto kp = ra == 0 ? (big ? irrep_gemm_xdma16p_kernel<1, 0> : irrep_gemm_xdma16p_kernel<0, 0>)
                        : ra == 4 ? (big ? irrep_gemm_xdma16p_kernel<1, 4> : irrep_gemm_xdma16p_kernel<0, 4>)
                                  : (big ? irrep_gemm_xdma16p_kernel<1, 2> : irrep_gemm_xdma16p_kernel<0, 2>);
                e = hipFuncSetAttribute(reinterpret_cast<const void *>(kp), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_use);
                if (e != hipSuccess) { roreg::set_error("%s: hipFuncSetAttribute: %s", what, hipGetErrorString(e)); return 1; }
                const int wgs = std::min(n_tiles, std::max(8, gemm_cu_count() / 8 * 8));      // one workgroup per CU (each needs the whole LDS), a multiple of 8 like n_tiles
                hipLaunchKernelGGL(kp, dim3(wgs), dim3(512), lds_use, roreg::as_stream(stream), p, tiles_dev, n_tiles / 8);
            } else {
                hipLaunchKernelGGL(kx, dim3(n_tiles), dim3(512), lds_use, roreg::as_stream(stream), p, tiles_dev);
            }
            hipError_t e2 = hipGetLastError();
            if (e2 != hipSuccess) { roreg::set_error("%s: launch failed: %s", what, hipGetErrorString(e2)); return 1; }
            return 0;
        } else {
            roreg::set_error("%s: the half-block layout needs the fp16 x 2 kernel with tile_m = 256", what);
            return 2;
        }
    }
    // ROREG_GEMM_PIPE=0 selects the loop without fragment pipelining (kept for A/B runs: results are bitwise the same)
    static const int pipe = [] { const char *e = getenv("ROREG_GEMM_PIPE"); return e ? atoi(e) : 1; }();
    auto kern = (long long)C * O == 256ll * 512 ? irrep_gemm_split_kernel<CT, NP, WO, 1> : irrep_gemm_split_kernel<CT, NP, WO, 0>;
    if constexpr (NP == 2) { if (pipe == 1) kern = (long long)C * O == 256ll * 512 ? irrep_gemm_split_kernel<CT, NP, WO, 1, 1> : irrep_gemm_split_kernel<CT, NP, WO, 0, 1>; }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { roreg::set_error("%s: hipFuncSetAttribute: %s", what, hipGetErrorString(e)); return 1; }
    hipLaunchKernelGGL(kern, dim3(n_tiles), dim3(WO * 128), lds, roreg::as_stream(stream), p, tiles_dev);
    hipError_t e2 = hipGetLastError();
    if (e2 != hipSuccess) { roreg::set_error("%s: launch failed: %s", what, hipGetErrorString(e2)); return 1; }
    return 0;
}

extern "C" int roreg_irrep_gemm_split(const float *const *X, float *const *Out, const float *const *Add, const void *const *Wsplit, int C, int O, int B,
                                      const int32_t *tiles_dev, int n_tiles, void *stream) {
    ROREG_REQUIRE(X && Out && Wsplit && tiles_dev && C > 0 && O > 0 && B > 0 && n_tiles > 0, "roreg_irrep_gemm_split: bad arguments");
    ROREG_REQUIRE(C % 32 == 0 && B % 4 == 0, "roreg_irrep_gemm_split: C %% 32 and B %% 4 must be 0 (got %d, %d)", C, B);
    return launch_gemm_split<3, 2>("roreg_irrep_gemm_split", X, Out, Add, Wsplit, nullptr, 0, nullptr, nullptr, nullptr, C, O, B, tiles_dev, n_tiles, stream);
}

extern "C" int roreg_irrep_gemm_f16x2(const float *const *X, float *const *Out, const float *const *Add, const void *const *Wsplit2,
                                      const float *x_bound_dev, int w_exp, const float *next_u_dev, const float *next_v_dev, float *out_bound_dev,
                                      int C, int O, int B, const int32_t *tiles_dev, int n_tiles, int tile_m, int x_planes, void *stream) {
    ROREG_REQUIRE(X && Out && Wsplit2 && tiles_dev && C > 0 && O > 0 && B > 0 && n_tiles > 0, "roreg_irrep_gemm_f16x2: bad arguments");
    ROREG_REQUIRE(!x_planes || tile_m == 256, "roreg_irrep_gemm_f16x2: x_planes needs tile_m = 256");
    ROREG_REQUIRE(C % 32 == 0 && B % 32 == 0, "roreg_irrep_gemm_f16x2: C %% 32 and B %% 32 must be 0 (got %d, %d)", C, B);
    ROREG_REQUIRE(x_bound_dev, "roreg_irrep_gemm_f16x2: x_bound_dev (the per-keypoint bound the activations were split under) is required");
    ROREG_REQUIRE(!out_bound_dev || (next_u_dev && next_v_dev), "roreg_irrep_gemm_f16x2: out_bound_dev needs next_u_dev / next_v_dev");
    ROREG_REQUIRE(tile_m == 128 || tile_m == 256, "roreg_irrep_gemm_f16x2: tile_m must be 128 or 256 (the value the tile list was built with)");
    if (tile_m == 256)
        return launch_gemm_split<2, 4>("roreg_irrep_gemm_f16x2", X, Out, Add, Wsplit2, x_bound_dev, w_exp, next_u_dev, next_v_dev, out_bound_dev, C, O, B,
                                       tiles_dev, n_tiles, stream, x_planes);
    return launch_gemm_split<2, 2>("roreg_irrep_gemm_f16x2", X, Out, Add, Wsplit2, x_bound_dev, w_exp, next_u_dev, next_v_dev, out_bound_dev, C, O, B,
                                   tiles_dev, n_tiles, stream);
}

// Per-keypoint bound on the coefficients of FT(act(x)) for a group-domain tensor x [B][C][60]: the transform is orthonormal, so
// |coefficient| <= |act(x_c)|_2 <= sqrt(60) max_g |act(x_c(g))|, act = ReLU(scale_c x + shift_c) or the identity.  One wave per keypoint;
// entries [B, Bp) (pad keypoints: their coefficients are exact zeros) get 0.
__global__ __launch_bounds__(256) void row_bound_kernel(const float *__restrict__ x, int x_bf16, const float *__restrict__ bn_scale,
                                                        const float *__restrict__ bn_shift, float *__restrict__ bound, int B, int Bp, int C) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= Bp) return;
    float mx = 0.f;
    if (b < B) {
        const size_t row = (size_t)b * C * ROREG_G;
        for (int i = lane; i < C * ROREG_G; i += 64) {
            float v = x_bf16 ? __uint_as_float((unsigned)reinterpret_cast<const unsigned short *>(x)[row + i] << 16) : x[row + i];
            if (bn_scale) { const int c = i / ROREG_G; v = fmaxf(fmaf(v, bn_scale[c], bn_shift[c]), 0.f); }
            mx = fmaxf(mx, fabsf(v));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    }
    if (lane == 0) bound[b] = mx * 7.7536f;                      // sqrt(60) = 7.7460, plus slack for the f32 rounding of the transform itself
}

extern "C" int roreg_row_bound(const void *x_spatial, int x_bf16, const float *bn_scale, const float *bn_shift, float *bound_out, int B, int C,
                               void *stream) {
    ROREG_REQUIRE(x_spatial && bound_out && B > 0 && C > 0, "roreg_row_bound: bad arguments");
    ROREG_REQUIRE((bn_scale == nullptr) == (bn_shift == nullptr), "roreg_row_bound: bn_scale and bn_shift go together");
    const int Bp = (B + 31) / 32 * 32;
    hipLaunchKernelGGL(row_bound_kernel, dim3((Bp + 3) / 4), dim3(256), 0, roreg::as_stream(stream), reinterpret_cast<const float *>(x_spatial), x_bf16, bn_scale, bn_shift,
                       bound_out, B, Bp, C);
    ROREG_CHECK_LAUNCH("roreg_row_bound");
    return 0;
}

// The transform kernels are persistent: every wave walks tiles wave, wave + n_waves, ...  The grid is a whole number of "rounds" of what the
// chip holds at once (occupancy differs per variant: 2-3 workgroups per SIMD row), so that no round runs part-filled.
template <void (*KERN)(NonlinParams)>
static void launch_ft(const NonlinParams &p, long long n_tiles, hipStream_t s) {
    static const int resident = [] {
        int per_cu = 0, dev = 0, cus = 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, KERN, 256, 0) != hipSuccess || per_cu < 1) per_cu = 2;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        return per_cu * cus;
    }();
    static const int rounds = [] { const char *e = getenv("ROREG_FT_ROUNDS"); const int r = e ? atoi(e) : 0; return r > 0 ? r : 4; }();
    const long long want = (n_tiles + 3) / 4, cap = (long long)resident * rounds;
    hipLaunchKernelGGL(KERN, dim3((unsigned)(want > cap ? cap : want)), dim3(256), 0, s, p);
}

extern "C" int roreg_ft_nonlin(const float *Xin, const float *x_spatial, const float *bias, const float *bias2,
                               const float *bn_scale, const float *bn_shift, const float *resid_spatial, float *Xout, float *out_spatial,
                               const int32_t *g_map, int Lout, int Lvalid, int B, int C, int split, const float *out_bound, float *out_rowmax,
                               int spatial_bf16, int out_planes, void *stream) {
    ROREG_REQUIRE(g_A1 && g_A2, "roreg_ft_nonlin: roreg_set_fourier_tables has not been called");
    ROREG_REQUIRE((Xin != nullptr) != (x_spatial != nullptr), "roreg_ft_nonlin: exactly one of Xin / x_spatial");
    ROREG_REQUIRE((Xout != nullptr) != (out_spatial != nullptr), "roreg_ft_nonlin: exactly one of Xout / out_spatial");
    ROREG_REQUIRE((bn_scale == nullptr) == (bn_shift == nullptr) && B > 0 && C > 0 && C <= 512, "roreg_ft_nonlin: bad arguments (C <= 512)");
    ROREG_REQUIRE((long long)60 * C * B < (1ll << 40), "roreg_ft_nonlin: tensor too large");
    ROREG_REQUIRE(!g_map || (Lout >= Lvalid && Lvalid > 0 && Lout <= 64), "roreg_ft_nonlin: bad Lout/Lvalid");
    NonlinParams p;
    memset(&p, 0, sizeof(p));
    p.Xin = Xin; p.Xout = Xout;
    p.x_spatial = x_spatial; p.r_spatial = resid_spatial; p.out_spatial = out_spatial;
    p.g_map = g_map; p.Lout = g_map ? Lout : ROREG_G; p.Lvalid = g_map ? Lvalid : ROREG_G;
    p.bias = bias; p.bias2 = bias2; p.bn_scale = bn_scale; p.bn_shift = bn_shift; p.A1 = g_A1; p.A2 = g_A2;
    p.A1s = reinterpret_cast<const bf16x8 *>(g_A1s); p.A2s = reinterpret_cast<const bf16x8 *>(g_A2s); p.out_bound = out_bound; p.out_rowmax = out_rowmax; p.x_bf16 = spatial_bf16;
    p.A1h = reinterpret_cast<const f16x8 *>(g_A1h); p.A2h = reinterpret_cast<const f16x8 *>(g_A2h); p.f_exp = g_f_exp;
    p.B = B; p.Bp = (B + 31) / 32 * 32; p.C = C; p.tiles_per_c = (B + 31) / 32;
    p.out_planes = out_planes;
    ROREG_REQUIRE((unsigned long long)C * p.Bp * 20ull < (1ull << 32), "roreg_ft_nonlin: C * round_up(B, 32) must stay below 2^32 / 20 (32-bit lane offsets)");
    ROREG_REQUIRE(!out_planes || (split == 2 && Xout), "roreg_ft_nonlin: out_planes is a layout of the fp16 x 2 coefficient output");
    const long long n_tiles = (long long)C * p.tiles_per_c;
    hipStream_t s = roreg::as_stream(stream);
    roreg::ProfScope prof(roreg::PROF_FT_NONLIN, s);
    const bool in_sp = x_spatial != nullptr, out_sp = out_spatial != nullptr;
    if (in_sp && out_sp) { roreg::set_error("roreg_ft_nonlin: spatial -> spatial is not a transform"); return 2; }
    ROREG_REQUIRE(split != 2 || out_sp || out_bound, "roreg_ft_nonlin: split = 2 writes the coefficients as fp16 hi/lo pairs and needs out_bound");
    ROREG_REQUIRE(!spatial_bf16 || in_sp || resid_spatial, "roreg_ft_nonlin: spatial_bf16 refers to x_spatial / resid_spatial");
    ROREG_REQUIRE(!out_rowmax || out_sp, "roreg_ft_nonlin: out_rowmax goes with a group-domain output");
    ROREG_REQUIRE(split >= 0 && split <= 2, "roreg_ft_nonlin: split must be 0 (f32 MFMA), 1 (bf16 x 3) or 2 (fp16 x 2)");
    if (split == 2) {
        if (in_sp) launch_ft<ft_nonlin_kernel<true, false, 2, 4, 1>>(p, n_tiles, s);
        else if (!out_sp) launch_ft<ft_nonlin_kernel<false, false, 2, 4, 1>>(p, n_tiles, s);
        else launch_ft<ft_nonlin_kernel<false, true, 2, 4, 1>>(p, n_tiles, s);
    } else if (split == 1) {
        // (6-wave workgroups at 3 waves per SIMD -- <.., 6, 3>, 384 threads, 139 VGPRs -- measured SLOWER: 4.6 vs 4.0 ms at C=512, B=65000)
        if (in_sp) launch_ft<ft_nonlin_kernel<true, false, 3, 4, 1>>(p, n_tiles, s);
        else if (!out_sp) launch_ft<ft_nonlin_kernel<false, false, 3, 4, 1>>(p, n_tiles, s);
        else launch_ft<ft_nonlin_kernel<false, true, 3, 4, 1>>(p, n_tiles, s);
    } else {
        if (in_sp) launch_ft<ft_nonlin_kernel<true, false, 0, 4, 1>>(p, n_tiles, s);
        else if (!out_sp) launch_ft<ft_nonlin_kernel<false, false, 0, 4, 1>>(p, n_tiles, s);
        else launch_ft<ft_nonlin_kernel<false, true, 0, 4, 1>>(p, n_tiles, s);
    }
    ROREG_CHECK_LAUNCH("roreg_ft_nonlin");
    return 0;
}

// Inverse transform + bias + BatchNorm + ReLU with the group-domain result written as fp16 hi / lo WORDS under a per-row block scale taken from
// a bound that exists before the tensor does (the producing GEMM's propagated bound): the operand of roreg_group_conv_f16x2_packed.
extern "C" int roreg_ft_nonlin_packed(const float *Xin, const float *bias, const float *bn_scale, const float *bn_shift, uint32_t *out_words,
                                      const int32_t *g_map, int Lout, int Lvalid, int B, int C, const float *out_bound, float *raw_col, int raw_g,
                                      void *stream) {
    ROREG_REQUIRE(g_A1 && g_A2, "roreg_ft_nonlin_packed: roreg_set_fourier_tables has not been called");
    ROREG_REQUIRE(Xin && out_words && out_bound && B > 0 && C > 0 && C <= 512 && (bn_scale == nullptr) == (bn_shift == nullptr), "roreg_ft_nonlin_packed: bad arguments");
    ROREG_REQUIRE(!g_map || (Lout >= Lvalid && Lvalid > 0 && Lout <= 64), "roreg_ft_nonlin_packed: bad Lout/Lvalid");
    ROREG_REQUIRE(!raw_col || (raw_g >= 0 && raw_g < ROREG_G), "roreg_ft_nonlin_packed: raw_g out of range");
    NonlinParams p;
    memset(&p, 0, sizeof(p));
    p.Xin = Xin; p.out_spatial = reinterpret_cast<float *>(out_words);
    p.g_map = g_map; p.Lout = g_map ? Lout : ROREG_G; p.Lvalid = g_map ? Lvalid : ROREG_G;
    p.bias = bias; p.bn_scale = bn_scale; p.bn_shift = bn_shift; p.A1 = g_A1; p.A2 = g_A2;
    p.A1s = reinterpret_cast<const bf16x8 *>(g_A1s); p.A2s = reinterpret_cast<const bf16x8 *>(g_A2s);
    p.A1h = reinterpret_cast<const f16x8 *>(g_A1h); p.A2h = reinterpret_cast<const f16x8 *>(g_A2h); p.f_exp = g_f_exp;
    p.out_bound = out_bound; p.sp_pack = 1; p.raw_col = raw_col; p.raw_g = raw_g;
    p.B = B; p.Bp = (B + 31) / 32 * 32; p.C = C; p.tiles_per_c = (B + 31) / 32;
    ROREG_REQUIRE((unsigned long long)C * p.Bp * 20ull < (1ull << 32), "roreg_ft_nonlin_packed: C * round_up(B, 32) must stay below 2^32 / 20 (32-bit lane offsets)");
    hipStream_t s = roreg::as_stream(stream);
    roreg::ProfScope prof(roreg::PROF_FT_NONLIN, s);
    launch_ft<ft_nonlin_kernel<false, true, 2, 4, 1>>(p, (long long)C * p.tiles_per_c, s);
    ROREG_CHECK_LAUNCH("roreg_ft_nonlin_packed");
    return 0;
}

extern "C" int roreg_feat_coefs(const void *x, int x_bf16, float *out, int B, int C, int split, void *stream) {
    if (B == 0) return 0;
    ROREG_REQUIRE(g_A1 && g_A2, "roreg_feat_coefs: roreg_set_fourier_tables has not been called");
    ROREG_REQUIRE(x && out && B > 0 && C > 0 && C <= 512 && split >= 0 && split <= 2, "roreg_feat_coefs: bad arguments");
    NonlinParams p;
    memset(&p, 0, sizeof(p));
    p.x_spatial = reinterpret_cast<const float *>(x); p.x_bf16 = x_bf16; p.out_spatial = out;
    p.Lout = ROREG_G; p.Lvalid = ROREG_G; p.A1 = g_A1; p.A2 = g_A2;
    p.A1s = reinterpret_cast<const bf16x8 *>(g_A1s); p.A2s = reinterpret_cast<const bf16x8 *>(g_A2s);
    p.A1h = reinterpret_cast<const f16x8 *>(g_A1h); p.A2h = reinterpret_cast<const f16x8 *>(g_A2h); p.f_exp = g_f_exp;
    p.B = B; p.Bp = (B + 31) / 32 * 32; p.C = C; p.tiles_per_c = (B + 31) / 32;
    const long long n_tiles = (long long)C * p.tiles_per_c;
    hipStream_t s = roreg::as_stream(stream);
    if (split == 2) launch_ft<ft_nonlin_kernel<true, false, 2, 4, 1, true>>(p, n_tiles, s);
    else if (split == 1) launch_ft<ft_nonlin_kernel<true, false, 3, 4, 1, true>>(p, n_tiles, s);
    else launch_ft<ft_nonlin_kernel<true, false, 0, 4, 1, true>>(p, n_tiles, s);
    ROREG_CHECK_LAUNCH("roreg_feat_coefs");
    return 0;
}
