// Icosahedral group convolution on the gfx950 matrix cores (exact-f32 MFMA, v_mfma_f32_32x32x2_f32).
//
//   out[b,o,j] = bias[o] + sum_{k<KS} sum_{c<Cin} W[o,c,k] * act(x[b,c,gather[j,k]])   (+ residual[b,o,j])
//
// GEMM view: rows = output channels o (MFMA A operand = weights), columns = flattened (keypoint b, live group
// column j) pairs (MFMA B operand = activations), reduction = (k, c) with c innermost.  The 13-stencil gather
// is never materialised: each workgroup stages the [keypoints][CT channels][Lin] activation slab of its
// columns in LDS once per channel chunk (BatchNorm(eval)+ReLU applied while staging, because ReLU sits
// between BN and the conv so BN cannot be folded into W), and every lane reads its B value at
// slab[row(b)][c][gather[j,k]] with one ds_read_b32 -- the stencil is an LDS address, not a tensor.
//
// Weights are pre-packed (roreg_group_conv_pack_weights) so that the A fragment of four consecutive MFMA
// k-steps is ONE 16-byte load per lane, fully coalesced (1 KiB per wave instruction):
//   wpack float4 index ((k*(Cin/8) + c/8)*CoutPad + o)*2 + h  holds  W[o, 8*(c/8) + 2r + h, k], r = 0..3
// (h = lane>>5 is the MFMA k-half, so step r of the block multiplies channel pair (8*(c/8)+2r, +1)).
//
// Workgroup = 4 waves; wave tile = (OTW*32 output channels) x (4 column tiles of 32); the WO x WB wave grid
// gives a workgroup tile of (WO*OTW*32) x (WB*128).  Accumulators: OTW*4 tiles x 16 VGPRs.
//
// Reference semantics: network/group_feat.py:16-33, network/ops.py:11-64, network/eqv_trans.py:88-117,130-136.
#include "common.h"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

struct GCParams {
    const float *x;
    const float4 *wp;
    const float *bias;
    const float *bn_scale;
    const float *bn_shift;
    const float *residual;
    float *out;
    const int32_t *gather;
    float *partial;        // split-K workspace [ksplit][B*Cout*Lout] (nullptr when ksplit == 1)
    int B, Cin, Cout, CoutPad, Lout, ncols, gt_bytes;
    int ksplit, cin_per_slice;
};

template <int KS, int LIN, int CT, int WO, int WB, int OTW>
__global__ __launch_bounds__(256, 2) void group_conv_kernel(GCParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int *gt = reinterpret_cast<int *>(smem);
    float *xs = reinterpret_cast<float *>(smem + p.gt_bytes);

    constexpr int OT = WO * OTW * 32;
    constexpr int NCOL = WB * 128;
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int n_ot = p.CoutPad / OT;
    const int ot_idx = blockIdx.x % n_ot, ct_idx = blockIdx.x / n_ot;
    const int wo = w % WO, wb = w / WO;
    const int o_wave = ot_idx * OT + wo * (OTW * 32);
    const int col0 = ct_idx * NCOL;
    const int b_first = col0 / p.Lout;
    const int col_last = min(col0 + NCOL, p.ncols) - 1;
    const int nkp = col_last / p.Lout - b_first + 1;

    for (int i = tid; i < p.Lout * KS; i += 256) gt[i] = p.gather[i];

    int rowbase[4], gi[4], bcol[4];
    bool valid[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int n = col0 + wb * 128 + t * 32 + j;
        valid[t] = n < p.ncols;
        const int nn = valid[t] ? n : col0;
        const int b = nn / p.Lout;
        gi[t] = nn - b * p.Lout;
        bcol[t] = b;
        rowbase[t] = (b - b_first) * (CT * LIN) + h * LIN;
    }

    f32x16 acc[OTW][4];
#pragma unroll
    for (int a = 0; a < OTW; ++a)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][t][r] = 0.f;

    const bool has_bn = p.bn_scale != nullptr;

    const int c_begin = blockIdx.y * p.cin_per_slice, c_end = c_begin + p.cin_per_slice;
    for (int c0 = c_begin; c0 < c_end; c0 += CT) {
        __syncthreads();   // previous chunk fully consumed (also orders the gather-table fill)
        // ---- stage act(x[b_first .. b_first+nkp) [c0 .. c0+CT) [0 .. LIN)) into LDS -----------------
        if constexpr (LIN % 4 == 0) {
            constexpr int PER_KP4 = CT * LIN / 4;
            const int total4 = nkp * PER_KP4;
            const float4 *x4 = reinterpret_cast<const float4 *>(p.x);
            float4 *xs4 = reinterpret_cast<float4 *>(xs);
            for (int i = tid; i < total4; i += 256) {
                const int kp = i / PER_KP4, e4 = i - kp * PER_KP4;
                const int c = (e4 * 4) / LIN;
                float4 v = x4[((size_t)(b_first + kp) * p.Cin + c0) * (LIN / 4) + e4];
                if (has_bn) {
                    const float s = p.bn_scale[c0 + c], sh = p.bn_shift[c0 + c];
                    v.x = fmaxf(fmaf(v.x, s, sh), 0.f);
                    v.y = fmaxf(fmaf(v.y, s, sh), 0.f);
                    v.z = fmaxf(fmaf(v.z, s, sh), 0.f);
                    v.w = fmaxf(fmaf(v.w, s, sh), 0.f);
                }
                xs4[i] = v;
            }
        } else {
            constexpr int PER_KP = CT * LIN;
            const int total = nkp * PER_KP;
            for (int i = tid; i < total; i += 256) {
                const int kp = i / PER_KP, e = i - kp * PER_KP;
                const int c = e / LIN;
                float v = p.x[((size_t)(b_first + kp) * p.Cin + c0) * LIN + e];
                if (has_bn) v = fmaxf(fmaf(v, p.bn_scale[c0 + c], p.bn_shift[c0 + c]), 0.f);
                xs[i] = v;
            }
        }
        __syncthreads();

        // ---- MFMA over (k, c in chunk); the weight fragments of stencil position k+1 are in flight during position k -----
        constexpr int NQ = CT / 8;
        float4 a_cur[NQ][OTW], a_nxt[NQ][OTW];
        auto load_a = [&](int k, float4 (&a)[NQ][OTW]) {
            const float4 *wk = p.wp + ((size_t)(k * (p.Cin / 8) + c0 / 8) * p.CoutPad + o_wave + j) * 2 + h;
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int ot = 0; ot < OTW; ++ot) a[q][ot] = wk[((size_t)q * p.CoutPad + ot * 32) * 2];
        };
        load_a(0, a_cur);
#pragma unroll 1
        for (int k = 0; k < KS; ++k) {
            if (k + 1 < KS) load_a(k + 1, a_nxt);
            int off[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) off[t] = rowbase[t] + gt[gi[t] * KS + k];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float bv[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) bv[t] = xs[off[t] + (q * 8 + r * 2) * LIN];
#pragma unroll
                    for (int ot = 0; ot < OTW; ++ot) {
                        const float av = r == 0 ? a_cur[q][ot].x : r == 1 ? a_cur[q][ot].y : r == 2 ? a_cur[q][ot].z : a_cur[q][ot].w;
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            acc[ot][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[t], acc[ot][t], 0, 0, 0);
                    }
                }
            }
            if (k + 1 < KS) {
#pragma unroll
                for (int q = 0; q < NQ; ++q)
#pragma unroll
                    for (int ot = 0; ot < OTW; ++ot) a_cur[q][ot] = a_nxt[q][ot];
            }
        }
    }

    // ---- epilogue: bias (+ residual), masked store.  C/D map: col = lane&31, row = (r&3)+8*(r>>2)+4*(lane>>5)
    if (p.ksplit > 1) {     // split-K: raw partial sums; bias/residual are applied by the reduce kernel
        float *part = p.partial + (size_t)blockIdx.y * ((size_t)p.B * p.Cout * p.Lout);
#pragma unroll
        for (int ot = 0; ot < OTW; ++ot) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (!valid[t]) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = o_wave + ot * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (o < p.Cout) part[((size_t)bcol[t] * p.Cout + o) * p.Lout + gi[t]] = acc[ot][t][r];
                }
            }
        }
        return;
    }
#pragma unroll
    for (int ot = 0; ot < OTW; ++ot) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (!valid[t]) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = o_wave + ot * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (o < p.Cout) {
                    const size_t idx = ((size_t)bcol[t] * p.Cout + o) * p.Lout + gi[t];
                    float v = acc[ot][t][r] + p.bias[o];
                    if (p.residual) v += p.residual[idx];
                    p.out[idx] = v;
                }
            }
        }
    }
}

// split-K reduction in a fixed slice order (deterministic): out = bias + sum_s partial[s] (+ residual)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float *__restrict__ partial, const float *__restrict__ bias,
                                                            const float *__restrict__ residual, float *__restrict__ out,
                                                            size_t n, int Cout, int Lout, int ksplit) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int o = (int)((i / Lout) % Cout);
    float v = 0.f;
    for (int s = 0; s < ksplit; ++s) v += partial[(size_t)s * n + i];
    v += bias[o];
    if (residual) v += residual[i];
    out[i] = v;
}

// ---------------------------------------------------------------------------------------------------------------
// The same convolution on the bf16 matrix cores with f32 accuracy ("3 x bf16 split", see fourier.hip): activations and weights are
// the exact sums of three bf16 pieces, the six cross products of order <= 4 are accumulated in f32.  Reduction step = one stencil
// position k x 16 channels (the K of v_mfma_f32_32x32x16_bf16); the B operand of a lane is the 16-byte k-octet
//   slab[plane][h][keypoint][gather[j,k]]   (8 channels of one input column),
// i.e. the stencil is still an LDS address.  BatchNorm + ReLU + the three-way split are applied ONCE while staging a 16-channel
// chunk.  Workgroup = 4 waves x (64 output channels x 128 columns) = 256 x 128; weights stream from L2 in fragment order
//   wsplit[plane][k][c/16][h][CoutPad][8]  (one 16-byte load per fragment, next stencil position in flight).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// NP = 3: bf16 x 3 pieces.  NP = 2: fp16 hi/lo with power-of-two block scaling; the block is ONE ROW b of x (a keypoint / correspondence):
// its scale comes from the bound |act(x[b])| <= act_smax * in_rowmax[b] + act_tmax (act_smax = max |BN scale| or 1, act_tmax = max |BN
// shift| or 0; in_rowmax[b] = max |x[b]| tracked by the producing kernel), the weights carry 2^w_exp; the kernel tracks max |out[b]| per row
// for the next layer (atomic max: order-independent).  A row's result therefore never depends on the other rows of the launch.
struct SplitScale {
    const float *in_rowmax;        // [B] (NP = 2)
    float act_smax, act_tmax;
    int w_exp;
    float *out_rowmax;             // [B] or null
};
__device__ __forceinline__ int row_scale_exp(const SplitScale &q, int b) {
    const float mx = q.act_smax * q.in_rowmax[b] + q.act_tmax;
    int e = 0;
    if (mx > 0.f && mx < __builtin_inff()) { int ex; (void)frexpf(mx, &ex); e = 14 - ex; }
    return e > 100 ? 100 : (e < -100 ? -100 : e);
}
__device__ __forceinline__ void gc_split2(const float (&v)[8], float scale, f16x8 &hi, f16x8 &lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = v[e] * scale;
        const _Float16 h1 = (_Float16)x;
        hi[e] = h1; lo[e] = (_Float16)(x - (float)h1);
    }
}

struct GCSplitParams {
    const float *x;
    const void *ws;
    const float *bias, *bn_scale, *bn_shift;
    float *out;
    const int32_t *gather;
    int B, Cin, Cout, CoutPad, Lin, Lout, ncols, gt_bytes, nkp_max;
    SplitScale sc;
    // LDS slot order of the activation slab: input column c of keypoint kp lives at fragment slot kp * S + order[c] (order[c] < 0: column c
    // is never gathered and is not staged).  order == null: the natural order, S = Lin.  The gathered B-operand reads of a 16-lane
    // ds_read_b128 group hit bank quad (kp * S + order[gather[j][k]]) % 16; with the natural order of the ET trunk's 13-of-45 stencil 3.2
    // lanes of a group collide on average (65 % of the kernel's LDS cycles were bank conflicts); tools/lds_perm_search.py finds an order
    // with 1.97.  Pure data movement: results are bitwise unchanged.
    const int32_t *order;
    int S;
    // NP = 2 only: x holds WORDS fp16 hi | fp16 lo << 16 of act(x[b]) * 2^row_scale_exp(b) (roreg_ft_nonlin_packed wrote them: BatchNorm, ReLU and the
    // split happened in the producer); the staging then only regroups halves -- bn_scale / bn_shift must be null, sc.act_smax = 1, act_tmax = 0
    // and sc.in_rowmax = the bound the producer scaled with.
    int packed;
};

__device__ __forceinline__ void gc_split3(const float (&v)[8], bf16x8 &b1, bf16x8 &b2, bf16x8 &b3) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 h1 = (__bf16)v[e];
        const float r1 = v[e] - (float)h1;
        const __bf16 h2 = (__bf16)r1;
        const float r2 = r1 - (float)h2;
        b1[e] = h1; b2[e] = h2; b3[e] = (__bf16)r2;
    }
}

// PK: 0 = float32 input (BatchNorm, ReLU and the split happen while staging); 1 = packed words (fp16 hi | lo << 16 under the row's block scale),
// staged between the same two barriers: the staging only regroups halves.  (A double-buffered form -- chunk c + 1 regrouped into a second slab
// under chunk c's MFMAs, one barrier per chunk, 134 KB of LDS = one workgroup per CU -- measured slower, 671 vs 693 pairs/s contract-complete:
// the second resident workgroup hides more than the staging phase costs.  Removed; NOTES.md round 5.)
template <int KS, int NP, int PK = 0>
__global__ __launch_bounds__(256, 2) void group_conv_split_kernel(GCSplitParams p) {
    constexpr bool PACKED = PK != 0;
    static_assert(!PACKED || NP == 2, "packed input words are fp16 hi / lo pairs");
    using frag = typename std::conditional<NP == 3, bf16x8, f16x8>::type;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int *gt = reinterpret_cast<int *>(smem);                             // [Lout * KS] slot of every gathered column, then [Lin] slot of every input column
    frag *slab = reinterpret_cast<frag *>(smem + p.gt_bytes);            // [NP planes][2 k-octets][nkp_max][S slots]
    constexpr int OT = 256, NCOL = 128;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int Lin = p.Lin, Lout = p.Lout;
    const int n_ot = p.CoutPad / OT;
    const int ot_idx = blockIdx.x % n_ot, ct_idx = blockIdx.x / n_ot;
    const int o_wave = ot_idx * OT + w * 64;
    const int col0 = ct_idx * NCOL;
    const int b_first = col0 / Lout;
    const int col_last = min(col0 + NCOL, p.ncols) - 1;
    const int nkp = col_last / Lout - b_first + 1;
    const int S = p.S;
    const int plane_stride = 2 * p.nkp_max * S, h_stride = p.nkp_max * S;         // in 16-byte fragments
    int *slot_of = gt + Lout * KS;

    const bool has_bn = p.bn_scale != nullptr;
    for (int i = tid; i < Lout * KS; i += 256) gt[i] = p.order ? p.order[p.gather[i]] : p.gather[i];
    for (int i = tid; i < Lin; i += 256) slot_of[i] = p.order ? p.order[i] : i;

    int rowbase[4], gi[4], bcol[4];
    bool valid[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int n = col0 + t * 32 + j;
        valid[t] = n < p.ncols;
        const int nn = valid[t] ? n : col0;
        const int b = nn / Lout;
        gi[t] = nn - b * Lout;
        bcol[t] = b;
        rowbase[t] = h * h_stride + (b - b_first) * S;
    }

    f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][t][r] = 0.f;

    float oscale[4] = {1.f, 1.f, 1.f, 1.f};
    if constexpr (NP == 2) {
#pragma unroll
        for (int t = 0; t < 4; ++t) oscale[t] = ldexpf(1.f, -(row_scale_exp(p.sc, bcol[t]) + p.sc.w_exp));
    }
    const frag *wsb = reinterpret_cast<const frag *>(p.ws);
    const size_t ws_plane = (size_t)KS * (p.Cin / 16) * 2 * p.CoutPad;    // fragments per split plane

    // Activation staging: the raw float32 chunk x[b_first .. +nkp)[c0 .. c0+16)[0 .. Lin) -- 16 * Lin contiguous floats per keypoint -- of the
    // NEXT chunk travels global -> LDS by LDS-DMA while this chunk's MFMAs run (no registers, no wave waits on HBM latency); after the
    // stencil loop every thread converts its k-octets from the raw LDS copy (BatchNorm, ReLU, split) into the fragment slab.  (The earlier
    // form loaded each k-octet with eight dependent global loads, one item after the other: 12-16 us of exposed latency per chunk against
    // 4 us of MFMAs -- the kernel ran at 42 % matrix-pipe duty, independent of the operands, i.e. stall-bound, not power-bound.)
    float *raw = reinterpret_cast<float *>(slab + (size_t)NP * plane_stride);       // [nkp_max][16][Lin]
    const int raw_floats = p.nkp_max * 16 * Lin;
    // BatchNorm parameters of all input channels and the block scale of this tile's keypoints, staged ONCE: convert() runs between two
    // barriers of every chunk, and with these as global loads every one of its 4-5 iterations per thread exposed two L2 round trips
    // (the ISA showed s_waitcnt vmcnt(0) twice per iteration) -- more time per chunk than the chunk's MFMAs.
    float *bn_s = raw + raw_floats, *bn_h = bn_s + p.Cin, *kp_scale = bn_h + p.Cin;       // [Cin], [Cin], [nkp_max]
    for (int i = tid; i < p.Cin; i += 256) { bn_s[i] = has_bn ? p.bn_scale[i] : 1.f; bn_h[i] = has_bn ? p.bn_shift[i] : 0.f; }
    if constexpr (NP == 2)
        for (int i = tid; i < nkp; i += 256) kp_scale[i] = ldexpf(1.f, row_scale_exp(p.sc, b_first + i));
    const int pieces = nkp * 4 * Lin;                                                // 16-byte pieces of a chunk
    // A lane's DMA pieces are the same for every chunk (only the channel offset moves): their source offsets are worked out once -- the per-piece
    // integer division by 4 Lin was ~400 vector instructions per wavefront and chunk, more than the chunk's 312 MFMAs have issue slots to spare.
    constexpr int RAW_ITERS = 9;                                                     // ceil(nkp_max * 4 * Lin / 256) for Lout = 13, Lin <= 48
    int raw_off[RAW_ITERS];                                                          // float offset of the piece in x at c0 = 0 (-1: none)
#pragma unroll
    for (int q = 0; q < RAW_ITERS; ++q) {
        const int pc = w * 64 + q * 256 + lane;
        raw_off[q] = -1;
        if (pc < pieces) {
            const int kp = pc / (4 * Lin), r = pc - kp * (4 * Lin);
            raw_off[q] = ((b_first + kp) * p.Cin) * Lin + 4 * r;                     // (B * Cin * Lin < 2^31: checked by the launcher)
        }
    }
    auto issue_raw = [&](int c0) {
        const float *xc = p.x + (size_t)c0 * Lin;
        float *dst0 = raw + (size_t)(w * 64) * 4;
#pragma unroll
        for (int q = 0; q < RAW_ITERS; ++q) {
            if (w * 64 + q * 256 < pieces) {                                         // (wave-uniform)
                if (raw_off[q] >= 0)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(xc + raw_off[q]),
                                                     (__attribute__((address_space(3))) void *)(dst0 + (size_t)q * 1024), 16, 0, 0);
            }
        }
    };
    // The staging items of a thread are the same for every chunk: (source offset in the raw chunk, destination fragment) are worked out once
    // (two integer divisions per item and chunk before: the conversion was 4.4 vector instructions per MFMA of the whole kernel,
    // profiles/r05_et_conv_pmc.txt).
    constexpr int CV_ITEMS = 5;                                                      // ceil(nkp_max * 2 * Lin / 256) for Lout = 13, Lin <= 48
    int cv_src[CV_ITEMS], cv_dst[CV_ITEMS];
    {
        const int items = nkp * 2 * Lin;
#pragma unroll
        for (int q = 0; q < CV_ITEMS; ++q) {
            const int i = tid + q * 256;
            cv_src[q] = -1; cv_dst[q] = 0;
            if (i < items) {
                const int col = i % Lin, r = i / Lin;
                const int ho = r & 1, kp = r >> 1;
                const int sl = p.order ? p.order[col] : col;
                if (sl >= 0) { cv_src[q] = (kp * 16 + 8 * ho) * Lin + col; cv_dst[q] = ho * h_stride + kp * S + sl; }      // (kp_scale index = cv_src / (16 Lin))
            }
        }
    }
    // one item's eight words out of the raw chunk ...
    auto cv_load = [&](int q, unsigned (&wv)[8]) {
        if (cv_src[q] < 0) return;
        const float *src = raw + cv_src[q];
#pragma unroll
        for (int e = 0; e < 8; ++e) wv[e] = __float_as_uint(src[e * Lin]);
    };
    // ... and, regrouped into the hi octet and the lo octet, into the slab
    auto cv_store = [&](int q, const unsigned (&wv)[8]) {
        if (cv_src[q] < 0) return;
        unsigned hw[4], lw[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            hw[e] = __builtin_amdgcn_perm(wv[2 * e + 1], wv[2 * e], 0x05040100u);
            lw[e] = __builtin_amdgcn_perm(wv[2 * e + 1], wv[2 * e], 0x07060302u);
        }
        struct Q4 { unsigned a, b, c, d; };
        frag *dst = slab + cv_dst[q];
        dst[0] = __builtin_bit_cast(frag, Q4{hw[0], hw[1], hw[2], hw[3]});
        dst[plane_stride] = __builtin_bit_cast(frag, Q4{lw[0], lw[1], lw[2], lw[3]});
    };
    auto convert_packed = [&]() {                                                 // all items between the chunk's two barriers
#pragma unroll
        for (int q = 0; q < CV_ITEMS; ++q) {
            unsigned wv[8];
            cv_load(q, wv);
            cv_store(q, wv);
        }
    };
    auto convert_float = [&](int c0) {
        const int items = nkp * 2 * Lin;
        for (int i = tid; i < items; i += 256) {
            const int col = i % Lin, r = i / Lin;
            const int ho = r & 1, kp = r >> 1;
            const int sl = slot_of[col];
            if (sl < 0) continue;                                                    // a column no output gathers (padding)
            const float *src = raw + (kp * 16 + 8 * ho) * Lin + col;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = src[e * Lin];
            if (has_bn) {
                const float4 *sc4 = reinterpret_cast<const float4 *>(bn_s + c0 + 8 * ho), *sh4 = reinterpret_cast<const float4 *>(bn_h + c0 + 8 * ho);
                const float4 s0 = sc4[0], s1 = sc4[1], h0 = sh4[0], h1 = sh4[1];
                const float scv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, shv[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(fmaf(v[e], scv[e], shv[e]), 0.f);
            }
            frag *dst = slab + ho * h_stride + kp * S + sl;
            if constexpr (NP == 3) {
                bf16x8 b1, b2, b3;
                gc_split3(v, b1, b2, b3);
                dst[0] = b1; dst[plane_stride] = b2; dst[2 * plane_stride] = b3;
            } else {
                f16x8 hi, lo;
                gc_split2(v, kp_scale[kp], hi, lo);
                dst[0] = hi; dst[plane_stride] = lo;
            }
        }
    };
    auto convert = [&](int c0) {
        if constexpr (PACKED) convert_packed(); else convert_float(c0);
    };
    issue_raw(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                                                 // (also orders the gather-table fill)
    convert(0);
    int gix0[4];                                                                     // slot of every column block's first gathered column: the same for all chunks
#pragma unroll
    for (int t = 0; t < 4; ++t) gix0[t] = gt[gi[t] * KS];

    for (int c0 = 0; c0 < p.Cin; c0 += 16) {
        const bool more = c0 + 16 < p.Cin;
        __syncthreads();                                                             // the slab of this chunk is complete, the raw buffer is free
        const frag *slab_cur = slab;

        // ---- MFMA over the stencil; the weight fragments of position k+1 are in flight during position k ----------------
        // (the stencil loop is fully unrolled with two named fragment sets: a rolled loop with a register copy made the compiler wait for
        //  the weight loads of position k + 1 BEFORE the MFMAs of position k -- one exposed L2 round trip per stencil position)
        frag aw[2][2][NP];
        auto load_a = [&](int k, frag (&a)[2][NP]) {
            const frag *wk = wsb + (((size_t)k * (p.Cin / 16) + c0 / 16) * 2 + h) * p.CoutPad + o_wave + j;
#pragma unroll
            for (int ot = 0; ot < 2; ++ot)
#pragma unroll
                for (int sp = 0; sp < NP; ++sp) a[ot][sp] = wk[sp * ws_plane + ot * 32];
        };
        load_a(0, aw[0]);
        // The gathered activation fragments are software-pipelined one (position, column block) ahead of their MFMAs, the gather-table
        // entries one stencil position ahead: the dependent LDS chain index -> fragment (with 65 % of the kernel's LDS cycles being bank
        // conflicts of the gathered reads) is then off the MFMAs' critical path.
        int gix[2][4];
        frag fb[2][NP];
#pragma unroll
        for (int t = 0; t < 4; ++t) gix[0][t] = gix0[t];
        auto read_frags = [&](int slot, int t, int set) {
            const frag *bp = slab_cur + rowbase[t] + slot;
#pragma unroll
            for (int sp = 0; sp < NP; ++sp) fb[set][sp] = bp[sp * plane_stride];
        };
        read_frags(gix[0][0], 0, 0);
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const frag (&a_cur)[2][NP] = aw[k & 1];
            if (k + 1 < KS) load_a(k + 1, aw[(k + 1) & 1]);
            if (k == 0 && more) issue_raw(c0 + 16);                 // (after the weight loads of k = 1: their wait does not cover the DMA)
            if (k + 1 < KS) {
#pragma unroll
                for (int t = 0; t < 4; ++t) gix[(k + 1) & 1][t] = gt[gi[t] * KS + k + 1];
            }
            __builtin_amdgcn_sched_barrier(0);                      // the loads stay ahead of this position's MFMAs (the scheduler would sink them to their use)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int s = k * 4 + t;
                if (t < 3) read_frags(gix[k & 1][t + 1], t + 1, (s + 1) & 1);
                else if (k + 1 < KS) read_frags(gix[(k + 1) & 1][0], 0, (s + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
                const frag (&f)[NP] = fb[s & 1];
                f32x16 c0v = acc[0][t], c1v = acc[1][t];
                if constexpr (NP == 3) {
                    const bf16x8 b1 = f[0], b2 = f[1], b3 = f[2];
                    c0v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[0][2], b1, c0v, 0, 0, 0);
                    c1v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[1][2], b1, c1v, 0, 0, 0);
                    c0v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[0][1], b2, c0v, 0, 0, 0);
                    c1v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[1][1], b2, c1v, 0, 0, 0);
                    c0v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[0][0], b3, c0v, 0, 0, 0);
                    c1v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[1][0], b3, c1v, 0, 0, 0);
                    c0v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[0][1], b1, c0v, 0, 0, 0);
                    c1v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[1][1], b1, c1v, 0, 0, 0);
                    c0v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[0][0], b2, c0v, 0, 0, 0);
                    c1v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[1][0], b2, c1v, 0, 0, 0);
                    c0v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[0][0], b1, c0v, 0, 0, 0);
                    c1v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[1][0], b1, c1v, 0, 0, 0);
                } else {
                    const f16x8 bh = f[0], bl = f[1];
                    c0v = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur[0][1], bh, c0v, 0, 0, 0);
                    c1v = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur[1][1], bh, c1v, 0, 0, 0);
                    c0v = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur[0][0], bl, c0v, 0, 0, 0);
                    c1v = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur[1][0], bl, c1v, 0, 0, 0);
                    c0v = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur[0][0], bh, c0v, 0, 0, 0);
                    c1v = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur[1][0], bh, c1v, 0, 0, 0);
                }
                acc[0][t] = c0v; acc[1][t] = c1v;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (more) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                         // this wave's DMA pieces have landed ...
            __syncthreads();                                                         // ... everyone's have, and nobody reads the slab any more
            convert(c0 + 16);
        }
    }

    // ---- epilogue: bias, masked store.  C/D map: col = lane&31, row = (r&3)+8*(r>>2)+4*(lane>>5)
    // The lane's 32 biases are fetched first, all in flight together (round 5: read inside the store loop, every one of the 128 stores was
    // load, s_waitcnt vmcnt(0), add, store -- the wait also drains the store before it: 128 serialised round trips per wavefront).
    float wmax[4] = {0.f, 0.f, 0.f, 0.f};
    float bv[2][16];
#pragma unroll
    for (int ot = 0; ot < 2; ++ot)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = o_wave + ot * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            bv[ot][r] = o < p.Cout ? p.bias[o] : 0.f;
        }
#pragma unroll
    for (int ot = 0; ot < 2; ++ot) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (!valid[t]) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = o_wave + ot * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (o < p.Cout) {
                    const float v = (NP == 2 ? acc[ot][t][r] * oscale[t] : acc[ot][t][r]) + bv[ot][r];
                    p.out[((size_t)bcol[t] * p.Cout + o) * Lout + gi[t]] = v;
                    wmax[t] = fmaxf(wmax[t], fabsf(v));
                }
            }
        }
    }
    if constexpr (NP == 2) {
        if (p.sc.out_rowmax) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float m = fmaxf(wmax[t], __shfl_xor(wmax[t], 32));
                if (h == 0 && valid[t]) atomicMax(reinterpret_cast<unsigned *>(p.sc.out_rowmax) + bcol[t], __float_as_uint(m));
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Dense layer on row-major activations, 3 x bf16 split (f32-accurate):   out[b][o] = bias[o] + sum_k W[o][k] * act_k(x[b][k]) (+ res[b][o])
// with act_k(v) = max(v * scale[k] + shift[k], 0) (BatchNorm(eval)+ReLU, per k) or the identity.  This is the ET trunk's last layer
// (13 input columns -> the single column g = 0: K = 512*13 with the stencil's gather folded into the weight order) and the 1x1 head.
// Activations are the MFMA A operand (rows b: a lane's fragment is 8 CONSECUTIVE floats of one row -- no transposition anywhere),
// weights the B operand (columns o), so a lane's 16 accumulator rows are keypoints and its column is an output channel: stores are
// coalesced along o.  Workgroup = 128 rows x 256 output channels, 4 waves as 2 x 2 (64 x 128 each, 8 accumulator tiles); per K16 step
// the weight fragments of the next step (3 planes x 2 k-octets x 256 channels = 24 KiB) arrive by LDS-DMA, the activation k-octets of
// the next step are converted (BN, ReLU, split) by one thread each and written in fragment order; both areas double-buffered.
struct DenseParams {
    const float *x;                // [B][K]
    const void *ws;                // [NP][K/16][2][Opad][8], Opad = round_up(O, 256)
    const float *bias;             // [O]
    const float *scale, *shift;    // [K] or null
    const float *res;              // [B][O] or null; element (b, o) at res[(b * O + o) * res_stride]
    int res_stride;
    float *out;                    // [B][O]
    int B, K, O, Opad;
    SplitScale sc;
};

template <int NP>
__global__ __launch_bounds__(256, 2) void dense_split_kernel(DenseParams p) {
    using frag = typename std::conditional<NP == 3, bf16x8, f16x8>::type;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TB = 128, TO = 256;
    constexpr int XBUF = NP * 2 * TB, WBUF = NP * 2 * TO;        // fragments per buffer
    frag *xs = reinterpret_cast<frag *>(smem);                   // [2 buf][NP planes][2 k-octets][128 rows]
    frag *wsm = xs + 2 * XBUF;                                   // [2 buf][NP planes][2 k-octets][256 channels]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int wr = w & 1, wc = w >> 1;
    const int n_ot = p.Opad / TO;
    const int ot = blockIdx.x % n_ot, bt = blockIdx.x / n_ot;
    const int b0 = bt * TB, o0 = ot * TO;
    const int nsteps = p.K / 16;
    const size_t wplane = (size_t)nsteps * 2 * p.Opad;            // fragments per split plane
    const frag *wsb = reinterpret_cast<const frag *>(p.ws);

    f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][t][r] = 0.f;

    // activation staging: thread -> (row, k-octet)
    const int srow = tid >> 1, sho = tid & 1;
    int brow = b0 + srow;
    if (brow >= p.B) brow = p.B - 1;                              // clamped rows are never stored
    const float *xrow = p.x + (size_t)brow * p.K + 8 * sho;
    float xscale = 1.f;
    if constexpr (NP == 2) xscale = ldexpf(1.f, row_scale_exp(p.sc, brow));     // this thread stages one row: the row's own block scale
    const bool has_act = p.scale != nullptr;
    float4 xa, xb, s0, s1, t0, t1;
    auto load_x = [&](int ks) {
        const int kk = (ks < nsteps ? ks : nsteps - 1) * 16;
        xa = *reinterpret_cast<const float4 *>(xrow + kk);
        xb = *reinterpret_cast<const float4 *>(xrow + kk + 4);
        if (has_act) {      // the k-octet's BatchNorm constants travel with it (loaded inside convert_store they cost an exposed L2 round trip per step)
            const int kq = kk + 8 * sho;
            s0 = *reinterpret_cast<const float4 *>(p.scale + kq); s1 = *reinterpret_cast<const float4 *>(p.scale + kq + 4);
            t0 = *reinterpret_cast<const float4 *>(p.shift + kq); t1 = *reinterpret_cast<const float4 *>(p.shift + kq + 4);
        }
    };
    auto convert_store = [&](int ks, int buf) {
        float v[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
        if (has_act) {
            const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, sh[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(fmaf(v[e], sc[e], sh[e]), 0.f);
        }
        frag *dst = xs + buf * XBUF + sho * TB + srow;
        if constexpr (NP == 3) {
            bf16x8 b1, b2, b3;
            gc_split3(v, b1, b2, b3);
            dst[0] = b1; dst[2 * TB] = b2; dst[4 * TB] = b3;
        } else {
            f16x8 hi, lo;
            gc_split2(v, xscale, hi, lo);
            dst[0] = hi; dst[2 * TB] = lo;
        }
    };
    // weight fragments of a step: per plane [2 k-octets][256 channels] = 512 fragments = two per thread
    auto issue_w = [&](int ks, int buf) {
        const frag *q = wsb + (size_t)(ks < nsteps ? ks : nsteps - 1) * 2 * p.Opad + o0;
#pragma unroll
        for (int sp = 0; sp < NP; ++sp)
#pragma unroll
            for (int ho = 0; ho < 2; ++ho)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(q + sp * wplane + (size_t)ho * p.Opad + tid),
                                                 (__attribute__((address_space(3))) void *)(wsm + buf * WBUF + sp * (2 * TO) + ho * TO + w * 64), 16, 0, 0);
    };

    load_x(0);
    issue_w(0, 0);
    convert_store(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int buf = 0;
    for (int ks = 0; ks < nsteps; ++ks) {
        issue_w(ks + 1, buf ^ 1);                                 // both land under the MFMAs of this step
        load_x(ks + 1);
        __builtin_amdgcn_sched_barrier(0);
        const frag *xt = xs + buf * XBUF + h * TB + wr * 64 + j;
        const frag *wt = wsm + buf * WBUF + h * TO + wc * 128 + j;
        frag a[2][NP];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int sp = 0; sp < NP; ++sp) a[rt][sp] = xt[sp * (2 * TB) + rt * 32];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x16 c0 = acc[0][t], c1 = acc[1][t];
            if constexpr (NP == 3) {
                const bf16x8 b1 = wt[t * 32], b2 = wt[2 * TO + t * 32], b3 = wt[4 * TO + t * 32];
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
                const f16x8 wh = wt[t * 32], wl = wt[2 * TO + t * 32];
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][1], wh, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][1], wh, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][0], wl, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][0], wl, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][0], wh, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][0], wh, c1, 0, 0, 0);
            }
            acc[0][t] = c0; acc[1][t] = c1;
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // next weights (LDS-DMA) and the staged activations have landed
        convert_store(ks + 1, buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    // ---- epilogue: C/D map: column (lane&31) = output channel, rows (r&3)+8*(r>>2)+4*h = keypoints --------------------------
    // NP = 2: per-row rescale factors and the per-row output maxima go through LDS (the tiles are dead: the loop ended with a barrier)
    float *so = reinterpret_cast<float *>(smem);                 // [128] 2^-(e(row) + w_exp)
    unsigned *rm = reinterpret_cast<unsigned *>(so + TB);        // [128] max |out[row]| of this tile
    if constexpr (NP == 2) {
        if (tid < TB) {
            const int b = min(b0 + tid, p.B - 1);
            so[tid] = ldexpf(1.f, -(row_scale_exp(p.sc, b) + p.sc.w_exp));
            rm[tid] = 0u;
        }
        __syncthreads();
    }
    float wmax[2][16];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) wmax[rt][r] = 0.f;
    if (b0 + TB <= p.B && o0 + TO <= p.O) {
        // interior tile: no per-element range branches, and the 16 residual loads (and row scales) of a 32-row block go out together -- in the
        // generic body below every element is a branch with its own load and s_waitcnt vmcnt(0): 128 exposed L2 round trips per thread,
        // several times the K loop of the 1x1 layers.  Same arithmetic: (acc * scale + bias) + residual.
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            float sc16[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) sc16[r] = NP == 2 ? so[wr * 64 + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h] : 1.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int o = o0 + wc * 128 + t * 32 + j;
                const float bo = p.bias[o];
                const size_t base = (size_t)(b0 + wr * 64 + rt * 32 + 4 * h) * p.O + o;
                float rv[16];
                if (p.res) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) rv[r] = p.res[(base + (size_t)((r & 3) + 8 * (r >> 2)) * p.O) * p.res_stride];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = (NP == 2 ? acc[rt][t][r] * sc16[r] : acc[rt][t][r]) + bo;
                    if (p.res) v += rv[r];
                    p.out[base + (size_t)((r & 3) + 8 * (r >> 2)) * p.O] = v;
                    wmax[rt][r] = fmaxf(wmax[rt][r], fabsf(v));
                }
            }
        }
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int o = o0 + wc * 128 + t * 32 + j;
            if (o >= p.O) continue;
            const float bo = p.bias[o];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wr * 64 + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const int b = b0 + row;
                    if (b < p.B) {
                        float v = (NP == 2 ? acc[rt][t][r] * so[row] : acc[rt][t][r]) + bo;
                        if (p.res) v += p.res[((size_t)b * p.O + o) * p.res_stride];
                        p.out[(size_t)b * p.O + o] = v;
                        wmax[rt][r] = fmaxf(wmax[rt][r], fabsf(v));
                    }
                }
        }
    }
    if constexpr (NP == 2) {
        if (p.sc.out_rowmax) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r) atomicMax(rm + wr * 64 + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, __float_as_uint(wmax[rt][r]));
            __syncthreads();
            if (tid < TB && b0 + tid < p.B) atomicMax(reinterpret_cast<unsigned *>(p.sc.out_rowmax) + b0 + tid, rm[tid]);
        }
    }
}

inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

// Number of K slices.  Split-K changes the association of the channel sum, so the slice count must be a function of the LAYER alone --
// never of the batch size -- or a row's result would depend on how many other rows share the launch (the reference's batched forward
// does not).  The one layer shape whose grid is chronically under-filled is the ET trunk's last layer (13-tap stencil, a single live
// output column: columns = rows of the batch): it always runs as 8 slices of whole channel chunks (fixed-order reduction); every other
// shape runs unsplit.
inline int choose_ksplit(int KS, int Lout, int n_chunks) {
    if (KS != 13 || Lout != 1) return 1;
    int s = 1;
    for (int d = 1; d <= n_chunks && d <= 8; ++d)
        if (n_chunks % d == 0) s = d;
    return s;
}

template <int KS, int LIN, int CT, int WO, int WB, int OTW>
int launch(GCParams p, hipStream_t stream, float *ws, size_t ws_floats, size_t *query_ws) {
    constexpr int OT = WO * OTW * 32;
    constexpr int NCOL = WB * 128;
    if (p.CoutPad % OT != 0) {
        roreg::set_error("roreg_group_conv: Cout pad %d not a multiple of the %d-row tile", p.CoutPad, OT);
        return 2;
    }
    if (p.Cin % CT != 0) {
        roreg::set_error("roreg_group_conv: Cin %d not a multiple of the %d-channel chunk", p.Cin, CT);
        return 2;
    }
    p.gt_bytes = round_up(p.Lout * KS * 4, 16);
    int nkp_max = (NCOL - 1) / p.Lout + 2;
    if (nkp_max > p.B) nkp_max = p.B;
    const size_t lds = (size_t)p.gt_bytes + (size_t)nkp_max * CT * LIN * 4;
    if (lds > 160 * 1024) {
        roreg::set_error("roreg_group_conv: tile needs %zu B of LDS", lds);
        return 2;
    }
    const int n_ct = (p.ncols + NCOL - 1) / NCOL;
    const int grid = n_ct * (p.CoutPad / OT);
    const int ksplit = choose_ksplit(KS, p.Lout, p.Cin / CT);
    const size_t n_out = (size_t)p.B * p.Cout * p.Lout;
    if (query_ws) { *query_ws = ksplit > 1 ? (size_t)ksplit * n_out : 0; return 0; }
    auto kern = group_conv_kernel<KS, LIN, CT, WO, WB, OTW>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            roreg::set_error("roreg_group_conv: hipFuncSetAttribute(%zu): %s", lds, hipGetErrorString(e));
            return 1;
        }
    }
    p.ksplit = ksplit; p.cin_per_slice = p.Cin / ksplit;
    if (ksplit > 1) {
        if (!ws || ws_floats < (size_t)ksplit * n_out) {
            roreg::set_error("roreg_group_conv: split-K needs a workspace of %zu floats (got %zu)", (size_t)ksplit * n_out, ws_floats);
            return 2;
        }
        p.partial = ws;
    }
    hipLaunchKernelGGL(kern, dim3(grid, ksplit), dim3(256), lds, stream, p);
    if (ksplit > 1)
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, stream, ws, p.bias, p.residual,
                           p.out, n_out, p.Cout, p.Lout, ksplit);
    ROREG_CHECK_LAUNCH("roreg_group_conv");
    return 0;
}

}  // namespace

extern "C" size_t roreg_group_conv_packed_size(int Cin, int Cout, int KS) {
    return (size_t)KS * Cin * round_up(Cout, 32);
}

extern "C" int roreg_group_conv_pack_weights(const float *W, int Cin, int Cout, int KS, float *wp) {
    ROREG_REQUIRE(W && wp, "roreg_group_conv_pack_weights: null pointer");
    ROREG_REQUIRE(Cin % 8 == 0 && Cin > 0 && Cout > 0 && (KS == 13 || KS == 1),
                  "roreg_group_conv_pack_weights: unsupported shape Cin=%d Cout=%d KS=%d", Cin, Cout, KS);
    const int CoutPad = round_up(Cout, 32);
    memset(wp, 0, sizeof(float) * roreg_group_conv_packed_size(Cin, Cout, KS));
    for (int k = 0; k < KS; ++k)
        for (int cb = 0; cb < Cin / 8; ++cb)
            for (int o = 0; o < Cout; ++o)
                for (int h = 0; h < 2; ++h)
                    for (int r = 0; r < 4; ++r) {
                        const int c = cb * 8 + 2 * r + h;
                        wp[((((size_t)k * (Cin / 8) + cb) * CoutPad + o) * 2 + h) * 4 + r] =
                            W[((size_t)o * Cin + c) * KS + k];
                    }
    return 0;
}

static int dispatch(GCParams p, int Lin, int KS, hipStream_t s, float *ws, size_t wsf, size_t *q);

extern "C" size_t roreg_group_conv_workspace_size(int B, int Cin, int Cout, int Lin, int Lout, int KS) {
    GCParams p;
    memset(&p, 0, sizeof(p));
    p.B = B; p.Cin = Cin; p.Cout = Cout; p.CoutPad = round_up(Cout, 32); p.Lout = Lout; p.ncols = B * Lout;
    size_t q = 0;
    if (B <= 0 || dispatch(p, Lin, KS, nullptr, nullptr, 0, &q) != 0) return 0;
    return q;
}

extern "C" int roreg_group_conv(const float *x, const float *wpack, const float *bias, const float *bn_scale,
                                const float *bn_shift, const float *residual, float *out, const int32_t *gather,
                                int B, int Cin, int Cout, int Lin, int Lout, int KS, float *workspace, size_t workspace_floats,
                                void *stream) {
    if (B == 0) return 0;
    ROREG_REQUIRE(x && wpack && bias && out && gather, "roreg_group_conv: null pointer");
    ROREG_REQUIRE((bn_scale == nullptr) == (bn_shift == nullptr), "roreg_group_conv: bn_scale/bn_shift must come together");
    ROREG_REQUIRE(B >= 0 && Cin > 0 && Cout > 0 && Lout > 0 && Lin > 0, "roreg_group_conv: bad sizes");
    if (B == 0) return 0;
    ROREG_REQUIRE((long long)B * Lout < (1ll << 31), "roreg_group_conv: too many columns");
    GCParams p;
    p.x = x; p.wp = reinterpret_cast<const float4 *>(wpack); p.bias = bias; p.bn_scale = bn_scale; p.bn_shift = bn_shift;
    p.residual = residual; p.out = out; p.gather = gather;
    p.B = B; p.Cin = Cin; p.Cout = Cout; p.CoutPad = round_up(Cout, 32); p.Lout = Lout; p.ncols = B * Lout; p.gt_bytes = 0;
    p.partial = nullptr; p.ksplit = 1; p.cin_per_slice = Cin;
    return dispatch(p, Lin, KS, roreg::as_stream(stream), workspace, workspace_floats, nullptr);
}

static int dispatch(GCParams p, int Lin, int KS, hipStream_t s, float *ws, size_t wsf, size_t *q) {
    const int cp = p.CoutPad;
    if (KS == 13 && Lin == 60) {
        if (cp % 128 == 0) return launch<13, 60, 32, 2, 2, 2>(p, s, ws, wsf, q);
        if (cp % 64 == 0) return launch<13, 60, 16, 1, 4, 2>(p, s, ws, wsf, q);
        return launch<13, 60, 16, 1, 4, 1>(p, s, ws, wsf, q);
    }
    if (KS == 13 && Lin == 48) {
        // 256 output channels x 128 columns per workgroup: the 48-column slabs of the ~11 keypoints a tile touches are small enough
        // for 32-channel chunks (half the barriers, twice the MFMA work per staged byte of the 2x2 layout)
        if (cp % 256 == 0) return launch<13, 48, 32, 4, 1, 2>(p, s, ws, wsf, q);
        if (cp % 128 == 0) return launch<13, 48, 16, 2, 2, 2>(p, s, ws, wsf, q);
        return launch<13, 48, 16, 1, 4, 1>(p, s, ws, wsf, q);
    }
    if (KS == 13 && Lin == 45) {
        if (cp % 128 == 0) return launch<13, 45, 16, 2, 2, 2>(p, s, ws, wsf, q);
        return launch<13, 45, 16, 1, 4, 1>(p, s, ws, wsf, q);
    }
    if (KS == 13 && Lin == 13) {
        if (cp % 256 == 0) return launch<13, 13, 8, 4, 1, 2>(p, s, ws, wsf, q);
        return launch<13, 13, 8, 1, 4, 1>(p, s, ws, wsf, q);
    }
    if (KS == 1 && Lin == 1) {
        if (cp % 128 == 0) return launch<1, 1, 32, 2, 2, 2>(p, s, ws, wsf, q);
        return launch<1, 1, 32, 1, 4, 1>(p, s, ws, wsf, q);
    }
    roreg::set_error("roreg_group_conv: unsupported (KS=%d, Lin=%d)", KS, Lin);
    return 2;
}

template <int NP>
static int launch_conv_split(GCSplitParams p, hipStream_t s) {
    const size_t lds = (size_t)p.gt_bytes + (size_t)NP * 2 * p.nkp_max * p.S * 16 + (size_t)p.nkp_max * 16 * p.Lin * 4        // slot tables, fragment slab, raw chunk,
                       + (size_t)(2 * p.Cin + p.nkp_max) * 4;                                                                // BatchNorm parameters, keypoint scales
    ROREG_REQUIRE(lds <= 160 * 1024, "roreg_group_conv_split: tile needs %zu B of LDS", lds);
    ROREG_REQUIRE((size_t)p.nkp_max * 2 * p.Lin <= 5 * 256 && (size_t)p.nkp_max * 4 * p.Lin <= 9 * 256, "roreg_group_conv_split: %d keypoints x %d columns per tile exceed the staging plan", p.nkp_max, p.Lin);
    ROREG_REQUIRE((long long)p.B * p.Cin * p.Lin < (1ll << 31), "roreg_group_conv_split: input tensor beyond 2^31 elements");
    void (*kern)(GCSplitParams) = group_conv_split_kernel<13, NP>;
    if constexpr (NP == 2) { if (p.packed) kern = group_conv_split_kernel<13, 2, 1>; }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { roreg::set_error("roreg_group_conv_split: hipFuncSetAttribute(%zu): %s", lds, hipGetErrorString(e)); return 1; }
    const int grid = ((p.ncols + 127) / 128) * (p.Cout / 256);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, p);
    ROREG_CHECK_LAUNCH("roreg_group_conv_split");
    return 0;
}

static int conv_split_common(int np, const float *x, const void *wsplit, const float *bias, const float *bn_scale, const float *bn_shift,
                             float *out, const int32_t *gather, const int32_t *lds_order, int lds_stride, int B, int Cin, int Cout, int Lin, int Lout,
                             int KS, SplitScale sc, void *stream, int packed = 0) {
    if (B == 0) return 0;
    ROREG_REQUIRE(x && wsplit && bias && out && gather && B > 0, "roreg_group_conv_split: bad arguments");
    ROREG_REQUIRE(!packed || (np == 2 && !bn_scale), "roreg_group_conv_split: packed input words go with the fp16 x 2 kernel and carry their activation");
    ROREG_REQUIRE((bn_scale == nullptr) == (bn_shift == nullptr), "roreg_group_conv_split: bn_scale/bn_shift must come together");
    ROREG_REQUIRE(KS == 13 && Cin % 16 == 0 && Cout % 256 == 0 && Lin > 0 && Lin <= 64 && Lout > 0 && Lout <= 64,
                  "roreg_group_conv_split: unsupported shape (KS=%d Cin=%d Cout=%d Lin=%d Lout=%d)", KS, Cin, Cout, Lin, Lout);
    ROREG_REQUIRE((long long)B * Lout < (1ll << 31), "roreg_group_conv_split: too many columns");
    GCSplitParams p;
    p.x = x; p.ws = wsplit; p.bias = bias; p.bn_scale = bn_scale; p.bn_shift = bn_shift; p.out = out;
    p.gather = gather; p.B = B; p.Cin = Cin; p.Cout = Cout; p.CoutPad = Cout; p.Lin = Lin; p.Lout = Lout; p.ncols = B * Lout;
    ROREG_REQUIRE(!lds_order || (lds_stride >= 1 && lds_stride <= 64), "roreg_group_conv_split: lds_stride %d out of range", lds_stride);
    p.order = lds_order; p.S = lds_order ? lds_stride : Lin;
    p.gt_bytes = round_up((Lout * KS + Lin) * 4, 16);
    p.nkp_max = (128 - 1) / Lout + 2;
    if (p.nkp_max > B) p.nkp_max = B;
    p.sc = sc; p.packed = packed;
    return np == 3 ? launch_conv_split<3>(p, roreg::as_stream(stream)) : launch_conv_split<2>(p, roreg::as_stream(stream));
}

extern "C" int roreg_group_conv_split(const float *x, const void *wsplit, const float *bias, const float *bn_scale, const float *bn_shift,
                                      float *out, const int32_t *gather, const int32_t *lds_order, int lds_stride, int B, int Cin, int Cout, int Lin,
                                      int Lout, int KS, void *stream) {
    SplitScale sc = {nullptr, 1.f, 0.f, 0, nullptr};
    return conv_split_common(3, x, wsplit, bias, bn_scale, bn_shift, out, gather, lds_order, lds_stride, B, Cin, Cout, Lin, Lout, KS, sc, stream);
}

extern "C" int roreg_group_conv_f16x2(const float *x, const void *wsplit2, int w_exp, const float *bias, const float *bn_scale, const float *bn_shift,
                                      float act_smax, float act_tmax, const float *in_rowmax_dev, float *out, float *out_rowmax_dev,
                                      const int32_t *gather, const int32_t *lds_order, int lds_stride, int B, int Cin, int Cout, int Lin, int Lout,
                                      int KS, void *stream) {
    ROREG_REQUIRE(in_rowmax_dev, "roreg_group_conv_f16x2: in_rowmax_dev is required");
    SplitScale sc = {in_rowmax_dev, act_smax, act_tmax, w_exp, out_rowmax_dev};
    return conv_split_common(2, x, wsplit2, bias, bn_scale, bn_shift, out, gather, lds_order, lds_stride, B, Cin, Cout, Lin, Lout, KS, sc, stream);
}

// x = the WORDS roreg_ft_nonlin_packed wrote (fp16 hi | lo << 16 of act(x[b]) 2^e_b, e_b from in_bound_dev[b]): the same convolution with a staging
// that only regroups halves (no BatchNorm, no conversion, no multiplication: 40 instead of ~250 vector instructions per 16-byte fragment pair)
extern "C" int roreg_group_conv_f16x2_packed(const uint32_t *x_words, const void *wsplit2, int w_exp, const float *bias, const float *in_bound_dev,
                                             float *out, float *out_rowmax_dev, const int32_t *gather, const int32_t *lds_order, int lds_stride, int B,
                                             int Cin, int Cout, int Lin, int Lout, int KS, void *stream) {
    ROREG_REQUIRE(in_bound_dev, "roreg_group_conv_f16x2_packed: in_bound_dev is required");
    SplitScale sc = {in_bound_dev, 1.f, 0.f, w_exp, out_rowmax_dev};
    return conv_split_common(2, reinterpret_cast<const float *>(x_words), wsplit2, bias, nullptr, nullptr, out, gather, lds_order, lds_stride, B, Cin, Cout, Lin,
                             Lout, KS, sc, stream, 1);
}

template <int NP>
static int launch_dense(DenseParams p, hipStream_t s) {
    const size_t lds = (size_t)2 * (NP * 2 * 128 + NP * 2 * 256) * 16;
    auto kern = dense_split_kernel<NP>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { roreg::set_error("roreg_dense_split: hipFuncSetAttribute(%zu): %s", lds, hipGetErrorString(e)); return 1; }
    const int grid = ((p.B + 127) / 128) * (p.Opad / 256);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, p);
    ROREG_CHECK_LAUNCH("roreg_dense_split");
    return 0;
}

static int dense_common(int np, const float *x, const void *wsplit, const float *bias, const float *scale, const float *shift,
                        const float *residual, int residual_stride, float *out, int B, int K, int O, SplitScale sc, void *stream) {
    if (B == 0) return 0;
    ROREG_REQUIRE(x && wsplit && bias && out && B > 0 && K > 0 && O > 0, "roreg_dense_split: bad arguments");
    ROREG_REQUIRE((scale == nullptr) == (shift == nullptr), "roreg_dense_split: scale/shift must come together");
    ROREG_REQUIRE(K % 16 == 0, "roreg_dense_split: K must be a multiple of 16 (got %d)", K);
    ROREG_REQUIRE(!residual || residual_stride >= 1, "roreg_dense_split: residual_stride must be >= 1 (got %d)", residual_stride);
    DenseParams p;
    p.x = x; p.ws = wsplit; p.bias = bias; p.scale = scale; p.shift = shift; p.res = residual; p.res_stride = residual_stride; p.out = out;
    p.B = B; p.K = K; p.O = O; p.Opad = round_up(O, 256); p.sc = sc;
    return np == 3 ? launch_dense<3>(p, roreg::as_stream(stream)) : launch_dense<2>(p, roreg::as_stream(stream));
}

extern "C" int roreg_dense_split(const float *x, const void *wsplit, const float *bias, const float *scale, const float *shift,
                                 const float *residual, int residual_stride, float *out, int B, int K, int O, void *stream) {
    SplitScale sc = {nullptr, 1.f, 0.f, 0, nullptr};
    return dense_common(3, x, wsplit, bias, scale, shift, residual, residual_stride, out, B, K, O, sc, stream);
}

extern "C" int roreg_dense_f16x2(const float *x, const void *wsplit2, int w_exp, const float *bias, const float *scale, const float *shift,
                                 float act_smax, float act_tmax, const float *in_rowmax_dev, const float *residual, int residual_stride,
                                 float *out, float *out_rowmax_dev, int B, int K, int O, void *stream) {
    ROREG_REQUIRE(in_rowmax_dev, "roreg_dense_f16x2: in_rowmax_dev is required");
    SplitScale sc = {in_rowmax_dev, act_smax, act_tmax, w_exp, out_rowmax_dev};
    return dense_common(2, x, wsplit2, bias, scale, shift, residual, residual_stride, out, B, K, O, sc, stream);
}
