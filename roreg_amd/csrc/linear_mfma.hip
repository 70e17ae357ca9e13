// The pointwise 1x1 layers of the rotation-coherence matcher on the matrix cores.
//
//   plain : y  = W x + b                                  (mlp_2layer's first conv, attention projections; rot_coh_match.py:14-32,95-119)
//   tail  : y += W relu((x - mean) * rstd) + b             (second conv of mlp_2layer / Contextnorm on top of the residual branch, :21-31,63-81)
// on [L, CIN] position-major rows, L = 2.5 k (one pair) ... 1.3 M (30 stacked pairs x 16 neighbours).  Rounds 1-3 evaluated them as one fmaf
// chain per (row, output) on the vector pipe: 16 TFLOP/s and 0.8 TB/s at 1.28 M x 96 -> 64, bound by neither (15 % of BASELINE configs[3]'s
// path, profiles/r04_rd_rm_config_flash_kernel_trace.txt).  Here a wavefront owns 32 rows: their inputs are read once (32 B per lane and k
// step: lane l holds row l % 32, k = 8 (l / 32) + e -- v_mfma_f32_32x32x16_f16's A fragment), normalised if asked, split into fp16 hi + lo
// and multiplied with the weight fragments the workgroup split once into LDS: ALL FOUR cross products (hi.hi, hi.lo, lo.hi, lo.lo), f32
// accumulate.  Every row is scaled by an exact power of two that puts its largest |input| in [2^13, 2^14) (the weights likewise, one
// scale per tensor), so each operand keeps 22 bits relative to its row's / the tensor's maximum and the result is at the level of a
// float32 fmaf chain (measured <= 5e-7 of sum |w||x| per element).  That level matters: the layers feed top-k neighbour selections, and a
// coarser product (three cross products on unscaled rows: 1.2e-6) flipped one neighbour at keynum 2500 and moved the log-couplings by 2e-3.
// A row's result depends on that row alone (fixed k order, no cross-row arithmetic, its own scale): the same kernel serves every L, so a
// pair's result does not depend on how many pairs are stacked.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

__device__ __forceinline__ int lm_seg_of(const int *__restrict__ off, int n_seg, int r) {
    int lo = 0, hi = n_seg - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (off[mid] <= r) lo = mid; else hi = mid - 1;
    }
    return lo;
}

template <int CIN, int COUT, bool NORM, bool ACCUM>
__global__ __launch_bounds__(256) void linear_mfma_kernel(const float *__restrict__ x, int L, const float *__restrict__ W, const float *__restrict__ b,
                                                          const float *__restrict__ mean_rstd, float *__restrict__ y, const int *__restrict__ seg_off,
                                                          int n_seg, int mult, int tiles_per_wave) {
    static_assert(CIN % 8 == 0 && COUT % 32 == 0, "linear_mfma_kernel: shapes");
    constexpr int KS = (CIN + 15) / 16, NT = COUT / 32;
    extern __shared__ __attribute__((aligned(16))) char lm_smem[];
    f16x8 *wf = reinterpret_cast<f16x8 *>(lm_smem);                         // [2 planes][KS][NT][64 lanes]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // ---- the weights' power-of-two scale (one per tensor): max |W| -> [2^13, 2^14) ----
    __shared__ float wmax_s[4];
    float wm = 0.f;
    for (int f = tid; f < CIN * COUT; f += 256) {
        const float a = fabsf(W[f]);
        if (a < __builtin_inff()) wm = fmaxf(wm, a);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wm = fmaxf(wm, __shfl_xor(wm, o));
    if (lane == 0) wmax_s[w] = wm;
    __syncthreads();
    wm = fmaxf(fmaxf(wmax_s[0], wmax_s[1]), fmaxf(wmax_s[2], wmax_s[3]));
    int ew = 0;
    if (wm > 0.f) { (void)frexpf(wm, &ew); ew = 14 - ew; }                  // wm 2^ew in [2^13, 2^14)
    const float wscale = ldexpf(1.f, ew), wback = ldexpf(1.f, -ew);
    // ---- the workgroup's weight fragments, once: B fragment of (k step, output tile): lane l holds output l % 32, k = 16 ks + 8 (l / 32) + e ----
    for (int f = tid; f < KS * NT * 64; f += 256) {
        const int l = f & 63, nt = (f >> 6) % NT, ks = f / (64 * NT);
        const int o = nt * 32 + (l & 31), k0 = ks * 16 + 8 * (l >> 5);
        f16x8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = k0 + e < CIN ? W[(size_t)o * CIN + k0 + e] * wscale : 0.f;
            const _Float16 h = (_Float16)v;
            hi[e] = h; lo[e] = (_Float16)(v - (float)h);
        }
        wf[(size_t)(ks * NT + nt) * 64 + l] = hi;
        wf[(size_t)((KS + ks) * NT + nt) * 64 + l] = lo;
    }
    __syncthreads();
    const int kg = lane >> 5;
    for (int t = 0; t < tiles_per_wave; ++t) {
        const int row0 = ((blockIdx.x * 4 + w) * tiles_per_wave + t) * 32;
        if (row0 >= L) break;
        const int row = min(row0 + (lane & 31), L - 1);
        const float *xr = x + (size_t)row * CIN;
        const float *ms = mean_rstd;
        if (NORM && seg_off) ms += (size_t)lm_seg_of(seg_off, n_seg, row / mult) * 2 * CIN;     // statistics of this row's pair
        // ---- the row's inputs: 8 floats per k step and lane ----
        float v[KS][8];
        float mx = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int k0 = ks * 16 + kg * 8;
            if (k0 < CIN) {                                                   // (CIN % 8 == 0: a k octet is whole or absent)
                const float4 a = *reinterpret_cast<const float4 *>(xr + k0), c = *reinterpret_cast<const float4 *>(xr + k0 + 4);
                v[ks][0] = a.x; v[ks][1] = a.y; v[ks][2] = a.z; v[ks][3] = a.w; v[ks][4] = c.x; v[ks][5] = c.y; v[ks][6] = c.z; v[ks][7] = c.w;
                if (NORM) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[ks][e] = fmaxf((v[ks][e] - ms[k0 + e]) * ms[CIN + k0 + e], 0.f);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[ks][e] = 0.f;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) mx = fmaxf(mx, fabsf(v[ks][e]));
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));                                  // the row's maximum (its two k halves sit 32 lanes apart)
        float down = 1.f, up = wback;
        if (mx > 0.f && mx < __builtin_inff()) {                              // (all-zero rows need no scale; non-finite rows: NaN / inf results either way)
            int ex;
            (void)frexpf(mx, &ex);                                            // mx = f 2^ex, f in [0.5, 1)
            ex = ex < -100 ? -100 : ex;                                       // (denormal-sized rows: keep the factors finite)
            down = ldexpf(1.f, 14 - ex); up = ldexpf(wback, ex - 14);
        }
        float upr[16];                                                        // the factor of each accumulator register's row (lane rr holds row rr's)
#pragma unroll
        for (int r = 0; r < 16; ++r) upr[r] = __shfl(up, 8 * (r >> 2) + 4 * kg + (r & 3));
        f16x8 ah[KS], al[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float s = v[ks][e] * down;
                const _Float16 h = (_Float16)s;
                ah[ks][e] = h; al[ks][e] = (_Float16)(s - (float)h);
            }
        // ---- output tiles ----
#pragma unroll 1
        for (int nt = 0; nt < NT; ++nt) {
            f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const f16x8 bh = wf[(size_t)(ks * NT + nt) * 64 + lane], bl = wf[(size_t)((KS + ks) * NT + nt) * 64 + lane];
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks], bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks], bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bh, acc, 0, 0, 0);
            }
            // accumulator register r of lane l: row 8 (r / 4) + 4 (l / 32) + r % 4 of the tile, output nt * 32 + l % 32
            const int o = nt * 32 + (lane & 31);
            const float bias = b[o];
            float prev[16];                                                 // (ACCUM: fetched together, not load / wait / add / store per row)
            if (ACCUM) {
#pragma unroll
                for (int r = 0; r < 16; ++r) prev[r] = y[(size_t)min(row0 + 8 * (r >> 2) + 4 * kg + (r & 3), L - 1) * COUT + o];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = 8 * (r >> 2) + 4 * kg + (r & 3);
                const float val = acc[r] * upr[r] + bias;                   // exact power-of-two rescale, then the bias
                if (row0 + rr < L) y[(size_t)(row0 + rr) * COUT + o] = ACCUM ? prev[r] + val : val;
            }
        }
    }
}

template <int CIN, int COUT, bool NORM, bool ACCUM>
int launch_linear_mfma(const float *x, int L, const float *W, const float *b, const float *mean_rstd, float *y, const int *seg_off, int n_seg, int mult,
                       hipStream_t s) {
    constexpr int KS = (CIN + 15) / 16, NT = COUT / 32;
    const size_t lds = (size_t)2 * KS * NT * 64 * sizeof(f16x8);
    const int tiles = (L + 31) / 32;
    int tpw = tiles / (4 * 2048);                                             // >= ~2048 workgroups before a wave takes a second tile
    if (tpw < 1) tpw = 1;
    if (tpw > 16) tpw = 16;
    const int wgs = (tiles + 4 * tpw - 1) / (4 * tpw);
    auto kern = linear_mfma_kernel<CIN, COUT, NORM, ACCUM>;
    static bool attr_set = false;
    if (!attr_set && lds > 48 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), lds, s, x, L, W, b, mean_rstd, y, seg_off, n_seg, mult, tpw);
    return 0;
}

}  // namespace

namespace roreg {

// -> true if the shape is served by the matrix-core kernel (and the launch was issued)
bool linear_mfma(const float *x, int L, int Cin, const float *W, const float *b, int Cout, float *y, hipStream_t s) {
#define LM(CI, CO) if (Cin == CI && Cout == CO) { launch_linear_mfma<CI, CO, false, false>(x, L, W, b, nullptr, y, nullptr, 1, 1, s); return true; }
    LM(32, 32) LM(96, 64) LM(120, 128) LM(64, 64) LM(96, 32) LM(120, 32) LM(64, 32)
#undef LM
    return false;
}

bool linear_tail_mfma(const float *h, int L, int Cmid, const float *mean_rstd, const float *W2, const float *b2, float *y, const int *seg_off, int n_seg,
                      int mult, hipStream_t s) {
    if (Cmid == 64) { launch_linear_mfma<64, 32, true, true>(h, L, W2, b2, mean_rstd, y, seg_off, n_seg, mult, s); return true; }
    if (Cmid == 128) { launch_linear_mfma<128, 32, true, true>(h, L, W2, b2, mean_rstd, y, seg_off, n_seg, mult, s); return true; }
    return false;
}

}  // namespace roreg
