/* roreg_hip.h -- C-ABI of libroreg_hip.so: the MI355X (gfx950) kernels behind RoReg's per-pair
 * registration hot path.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host; buffers are owned by the caller;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); nothing synchronises;
 *   - every entry point returns 0 on success, non-zero on failure (roreg_last_error() has the text);
 *   - tensors are C-contiguous; "f32 [B,C,60]" means float32 with 60 fastest;
 *   - group tables (P 60x60, Nei 60x13) are uploaded once with roreg_set_group_tables().
 *
 * Each entry cites the reference code it replaces (paths relative to the RoReg checkout).
 */
#ifndef ROREG_HIP_H
#define ROREG_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- library ------------------------------------------------------------------------------------ */
/* Version of THIS header.  It is bumped whenever an existing entry point changes its argument list or a shared struct its size
 * (2: round 2's per-keypoint block scales -- roreg_gf_finalize, roreg_inv_descriptor, roreg_et_gather, roreg_dense_split/_f16x2,
 * roreg_lt_prepare_batch, roreg_group_conv_f16x2, roreg_lt_task 80 -> 96 bytes; 3: round 3 -- roreg_ransac_score / roreg_refine /
 * roreg_ransac_batch take `w_f32`, the scores' storage type; roreg_group_conv_split / _f16x2 take an LDS slot order; roreg_ft_nonlin /
 * roreg_irrep_gemm_f16x2 take the plane-layout flags; 4: round 4 -- additions only (roreg_nn_search_ex / roreg_knn_search_ex / roreg_pdist and the entries marked "v4"),
 * bumped so that a binding can rely on them; 5: round 5 -- additions only, the entries marked "v5": roreg_sinkhorn_batch3 (+ its workspace size),
 * roreg_linear_path, roreg_linear_cat3, roreg_gemm_persistent, roreg_ft_nonlin_packed, roreg_group_conv_f16x2_packed; roreg_sinkhorn_batch2's `recompute` also takes 2); 6: round 6 -- additions only, the entries marked "v6".  A binding must compare roreg_abi_version() with the ROREG_ABI_VERSION it was written against and
 * refuse to call a library that answers differently (roreg_amd/hip.py:lib() does). */
#define ROREG_ABI_VERSION 6
int roreg_abi_version(void);
const char *roreg_last_error(void);

/* Upload the icosahedral tables: P[a*60+g] = index(R_g R_a) ("60_60.npy"), Nei[g*13+k]
 * ("Nei_Index_in_SO3_ordered_13.npy"), R[60*9] float64 ("Rotation.npy").  Host pointers.
 * Replaces the per-module np.load(...).cuda() at network/group_feat.py:12-14, rot_detect.py:39-40,
 * eqv_trans.py:83-86, rot_coh_match.py:127, test/estimator.py:78,374-375. */
int roreg_set_group_tables(const int32_t *P_host, const int32_t *Nei_host, const double *R_host);

/* ---- icosahedral group convolution (the MFMA kernel) --------------------------------------------
 * out[b,o,j] = bias[o] + sum_c sum_k W[o,c,k] * act(x)[b,c,gather[j*KS+k]]  (+ residual[b,o,j])
 *   act(x) = relu(x*bn_scale[c] + bn_shift[c]) when bn_scale != NULL, else x.
 * x is [B,Cin,Lin], out/residual are [B,Cout,Lout]; gather is int32 [Lout*KS] with values in [0,Lin).
 * For the full conv Lin=Lout=60, KS=13, gather=Nei.  Pruned layers (ET: only the group columns that can
 * reach g=0 are live) use Lout<60 and a composed gather; KS=1 gives the 1x1 convs of the ET head.
 * wpack is the weight tensor re-laid out by roreg_group_conv_pack_weights().
 * Replaces: data_process gather + BatchNorm2d(eval)+ReLU+Conv2d(C,O,(1,13)) at network/group_feat.py:16-33,
 * network/ops.py:11-64, network/eqv_trans.py:88-117,130-136, network/rot_detect.py:41,46. */
size_t roreg_group_conv_packed_size(int Cin, int Cout, int KS);            /* in floats */
int roreg_group_conv_pack_weights(const float *W_host /* [Cout,Cin,KS] */, int Cin, int Cout, int KS,
                                  float *wpack_host /* roreg_group_conv_packed_size floats */);
/* Small problems (few output tiles, long reduction) are split along the channel axis; the partial sums
 * go through a caller-owned workspace and are reduced in a fixed order (deterministic).  Size in floats,
 * 0 when the launch needs none. */
size_t roreg_group_conv_workspace_size(int B, int Cin, int Cout, int Lin, int Lout, int KS);
int roreg_group_conv(const float *x, const float *wpack, const float *bias,
                     const float *bn_scale, const float *bn_shift, const float *residual,
                     float *out, const int32_t *gather,
                     int B, int Cin, int Cout, int Lin, int Lout, int KS,
                     float *workspace, size_t workspace_floats, void *stream);

/* The same convolution with f32 accuracy on the bf16 matrix cores: operands as three bf16 pieces, six cross products, f32 accumulate
 * (2.67x fewer matrix-core cycles than the f32-input MFMA; measured error = the f32 kernel's).  KS = 13, Cin % 16 == 0,
 * Cout % 256 == 0, no residual.  wsplit: bf16 bits, layout [3 planes][KS][Cin/16][2 k-octets][Cout][8 channels]
 * (plane p of W[o, 16*(c/16) + 8*h + e, k]; the pieces are round-to-nearest-even of the exact remainders). */
int roreg_group_conv_split(const float *x, const void *wsplit, const float *bias, const float *bn_scale, const float *bn_shift,
                           float *out, const int32_t *gather,
                           const int32_t *lds_order /* nullable device int32 [Lin]: LDS slot of every input column (distinct values in [0, lds_stride); < 0 for a
                           column no output gathers) -- an execution hint that spreads the gathered operand reads over the LDS banks
                           (tools/lds_perm_search.py); results do not depend on it */, int lds_stride,
                           int B, int Cin, int Cout, int Lin, int Lout, int KS, void *stream);

/* Dense layer on row-major activations with the same f32-accurate 3 x bf16 split:
 *   out[b][o] = bias[o] + sum_k W[o][k] * act_k(x[b][k]) (+ residual[b][o]),  act_k(v) = max(v*scale[k] + shift[k], 0) or identity (scale NULL).
 * x [B][K] f32 (K % 16 == 0), out [B][O]; residual element (b, o) is read at residual[(b*O + o) * residual_stride] (1 = a dense [B][O]
 * tensor; 48 = column 0 of a [B][O][48] group-domain tensor: the ET trunk's identity short cut without a gathering copy).  wsplit: bf16 bits [3 planes][K/16][2 k-octets][round_up(O,256)][8] (zero rows beyond O).
 * Used for the ET trunk's last layer (K = 512*13: the 13-column stencil of the single live output column, gather folded into the weight
 * order; network/eqv_trans.py:101-117, network/ops.py:58-62) and the 1x1 head (network/eqv_trans.py:91-99,130-136). */
int roreg_dense_split(const float *x, const void *wsplit, const float *bias, const float *scale, const float *shift,
                      const float *residual, int residual_stride, float *out, int B, int K, int O, void *stream);

/* fp16 x 2 variants of the two entries above (half the matrix-core work; operands as hi + lo fp16 under a power-of-two block scale, see
 * roreg_irrep_gemm_f16x2).  The block is ONE ROW (keypoint / correspondence) b of x: its scale is derived on the device from the bound
 * |act(x[b])| <= act_smax * in_rowmax_dev[b] + act_tmax (act_smax = max |scale| or 1, act_tmax = max |shift| or 0; in_rowmax_dev [B] =
 * max |x[b]| maintained by the producing kernel), so a row's result never depends on which other rows share the launch (the reference's
 * batch-size independence, test/extractor.py:51-58, test/estimator.py:338-352).  The weights were scaled by 2^w_exp when split (layouts as
 * above with two planes hi, lo of fp16 bits).  out_rowmax_dev (nullable, [B], zeroed by the caller) receives max |out[b]| for the next layer. */
int roreg_group_conv_f16x2(const float *x, const void *wsplit2, int w_exp, const float *bias, const float *bn_scale, const float *bn_shift,
                           float act_smax, float act_tmax, const float *in_rowmax_dev, float *out, float *out_rowmax_dev,
                           const int32_t *gather, const int32_t *lds_order /* as roreg_group_conv_split */, int lds_stride,
                           int B, int Cin, int Cout, int Lin, int Lout, int KS, void *stream);
int roreg_dense_f16x2(const float *x, const void *wsplit2, int w_exp, const float *bias, const float *scale, const float *shift,
                      float act_smax, float act_tmax, const float *in_rowmax_dev, const float *residual, int residual_stride, float *out,
                      float *out_rowmax_dev, int B, int K, int O, void *stream);
/* v5: the ET trunk's 256 -> 512 convolution (network/eqv_trans.py:101-117, network/ops.py:46-57) fed by roreg_ft_nonlin_packed: x_words [B,Cin,Lin] holds
 * fp16 hi | fp16 lo << 16 of ReLU(BN(x[b])) * 2^e_b with e_b derived from in_bound_dev[b] (the bound the producer scaled with) -- BatchNorm, ReLU and
 * the operand split happened in the producer, the kernel's staging only regroups the halves.  Same outputs as roreg_group_conv_f16x2 up to the
 * block scale (a propagated bound instead of the tracked row maximum: measured 4.6-5.3 of the 22 bits). */
int roreg_group_conv_f16x2_packed(const uint32_t *x_words, const void *wsplit2, int w_exp, const float *bias, const float *in_bound_dev,
                                  float *out, float *out_rowmax_dev, const int32_t *gather, const int32_t *lds_order, int lds_stride,
                                  int B, int Cin, int Cout, int Lin, int Lout, int KS, void *stream);

/* eqv_raw [B,32,60] -> eqv = eqv_raw / max(||.||_2 over 32 ch, 1e-4) per (b,g);
 * inv = mean_g(eqv_raw) / max(||.||, 1e-4)  (inv may be NULL).  network/group_feat.py:38-43. */
int roreg_gf_finalize(const float *eqv_raw, void *eqv, int eqv_bf16 /* store eqv as bfloat16 (round to nearest even) instead of float32 */,
                      float *inv, int B, void *stream);

/* ---- detector ------------------------------------------------------------------------------------
 * enc [B,16,60] -> scores[b] = std_a( sum_f sum_g fn[f,P[a,g]] fn[f,g] ), fn = enc/||enc||_2 over 16 ch,
 * unbiased std over the 60 a's.  network/rot_detect.py:47-52. */
int roreg_det_score(const float *enc, float *scores, int B, void *stream);

/* ---- descriptors / nearest neighbours ------------------------------------------------------------
 * eqv [N,32,60] -> inv [N,32] = mean_g / (||.||_2 + 1e-5).  test/matcher.py:69-72. */
int roreg_inv_descriptor(const void *eqv, int eqv_bf16 /* eqv is stored as bfloat16; float32 arithmetic on the stored values */, float *inv, int N,
                         void *stream);

/* For every source row the nearest target row: d = sqrt(sum_f (s_f-t_f)^2 + 1e-7) accumulated in f order
 * (fp32, no FMA), first minimum wins.  src [m,F], tgt [n,F] row-major; optional row index lists
 * (src_rows/tgt_rows, int64, NULL = identity) select sampled keypoints without a gather copy.
 * idx_out int64 [m] (position within the target list), dist_out f32 [m] (may be NULL); scratch is m
 * uint64 of caller-owned workspace (packed (distance,index) keys merged across target slices by atomicMin).
 * Replaces knn_module.KNN(1).__call__ -> find_nn_gpu  (utils/knn_search.py:17-66,138-156). */
int roreg_nn_search(const float *src, const int64_t *src_rows, int m,
                    const float *tgt, const int64_t *tgt_rows, int n, int F,
                    int64_t *idx_out, float *dist_out, uint64_t *scratch, void *stream);

/* v4: roreg_nn_search with the distance type of modified_knn_matcher.find_nn_gpu (utils/knn_search.py:26-66): squared = 0 is 'L2' (the
 * function above), squared = 1 is 'SquareL2' = sum_f (s_f-t_f)^2 itself (no 1e-7, no root; find_nn_gpu's own default): the first minimum of
 * THAT value wins, which may differ from the 'L2' winner where two roots round to the same float. */
int roreg_nn_search_ex(const float *src, const int64_t *src_rows, int m,
                       const float *tgt, const int64_t *tgt_rows, int n, int F, int squared,
                       int64_t *idx_out, float *dist_out, uint64_t *scratch, void *stream);

/* v4: the whole distance matrix out[i*n+j] of modified_knn_matcher.pdist (utils/knn_search.py:17-24), A [m,F], B [n,F], any F >= 1;
 * squared as above. */
int roreg_pdist(const float *A, int m, const float *B, int n, int F, int squared, float *out, void *stream);

/* The mutual matcher for a batch of pairs (test/matcher.py:90-107 for every pair of a scene), on the matrix cores and bit-exact:
 * approximate squared distances |s|^2+|t|^2-2s.t as 3 x bf16 split MFMAs give row / column minima; every entry within a proven margin
 * of its row (column) minimum is re-evaluated with the literal formula of roreg_nn_search and merged "first minimum wins"
 * (csrc/mfma_match.hip).  Task p: sampled descriptors desc0[rows0[i]] (m0 rows) and desc1[rows1[j]] (m1 rows), F = 32, rows NULL =
 * identity; the mutual pairs (rows0 value, rows1 value) are written in increasing i to match_out + p*pitch*2 (pitch = max_m rounded up
 * to even), their number to counts_out[p].  tasks_dev is a DEVICE array; max_m >= every m0, m1.
 * workspace: roreg_mutual_match_batch_workspace(n_tasks, max_m) bytes. */
typedef struct {
    const float *desc0, *desc1;
    const int64_t *rows0, *rows1;
    int32_t m0, m1;
} roreg_match_task;
size_t roreg_mutual_match_batch_workspace(int n_tasks, int max_m);
int roreg_mutual_match_batch(const roreg_match_task *tasks_dev, int n_tasks, int max_m, int64_t *match_out, int32_t *counts_out,
                             void *workspace, size_t workspace_bytes, void *stream);

/* k nearest (k<=8) targets per source in increasing distance, first index on ties; idx_out int64 [m,k].
 * Replaces KNN(k>=2) -> find_knn_gpu (utils/knn_search.py:68-103), used by NMS_sample with F=3, k=5
 * (test/matcher.py:21-23). */
size_t roreg_knn_search_workspace(int m, int n);
int roreg_knn_search(const float *src, int m, const float *tgt, int n, int F, int k,
                     int64_t *idx_out, void *workspace /* roreg_knn_search_workspace bytes: targets are scanned in slices that fill the chip;
                     NULL = one thread per source scans everything */, size_t workspace_bytes, void *stream);

/* v4: roreg_knn_search with the distance type (squared as in roreg_nn_search_ex) and, optionally, the k distances (dist_out f32 [m,k] or
 * NULL): modified_knn_matcher.find_knn_gpu (utils/knn_search.py:68-103). */
int roreg_knn_search_ex(const float *src, int m, const float *tgt, int n, int F, int k, int squared, int64_t *idx_out, float *dist_out,
                        void *workspace, size_t workspace_bytes, void *stream);

/* roreg_knn_search for several clouds per launch: the point lists are stacked, seg_src / seg_tgt are DEVICE int32 [n_seg+1] row
 * offsets, a source only sees the targets of its own segment; idx_out int64 [m_total,k] holds indices LOCAL to the segment (the NMS
 * sampling of every cloud of a scene in two launches, test/matcher.py:21-23). */
size_t roreg_knn_search_seg_workspace(long long m_total, int n_seg, int max_m, int max_n);
int roreg_knn_search_seg(const float *src, const float *tgt, const int32_t *seg_src, const int32_t *seg_tgt, int n_seg, long long m_total,
                         int max_m, int max_n, int F, int k, int64_t *idx_out, void *workspace, size_t workspace_bytes, void *stream);

/* Mutual check + ordered compaction: for i in 0..m-1 (increasing) keep (i, nn01[i]) iff nn10[nn01[i]]==i;
 * pairs are mapped through sample0/sample1 (int64, NULL = identity) and written to match_out int64 [*,2];
 * *count_out (device int32) receives the number kept.  test/matcher.py:98-107. */
int roreg_mutual_matches(const int64_t *nn01, const int64_t *nn10, int m, int n /* entries of nn10 */,
                         const int64_t *sample0, const int64_t *sample1,
                         int64_t *match_out, int32_t *count_out, void *stream);

/* ---- estimator -----------------------------------------------------------------------------------
 * Per correspondence b: cor[a] = sum_f ( sum_g d1[f,P[a,g]] d2[f,g] ), argmax_a (first max).
 * d1 = feats1[rows1[b]], d2 = feats0[rows0[b]] with feats* f32 [N,32,60]; rows* int64 (NULL = b itself).
 * idx_out int64 [M]; cor_out f32 [M,60] optional.
 * Replaces extractor_dr_index.Batch_Des2R_torch (test/estimator.py:85-89) and the gathers at :108-110. */
int roreg_des2r(const float *feats1, const int64_t *rows1, const float *feats0, const int64_t *rows0,
                int M, int64_t *idx_out, float *cor_out, void *stream);

/* Generalised 60x60 group cross-correlation: cor[b,a] = sum_f sum_g A[f, T[a,g]] * B[f,g] with A = perm_feats[perm_rows[b]],
 * B = bcast_feats[bcast_rows[b]] and T = P (transpose_table=0) or P^T (transpose_table=1: T[a,g] = P[g,a]).
 * idx_out (first argmax) and cor_out (all 60 values) are each optional.  roreg_des2r is the transpose_table=0 case; the
 * matcher's R_indicator (network/rot_coh_match.py:154-163) is the transpose_table=1 case in both orientations. */
int roreg_group_corr(const float *perm_feats, const int64_t *perm_rows, const float *bcast_feats, const int64_t *bcast_rows,
                     int M, int transpose_table, int64_t *idx_out, float *cor_out, void *stream);

/* v4: all 60 correlations of roreg_group_corr as ONE [60 x 32] . [32 x 60] float32 matrix product per point on the matrix cores (C[p,g] = sum_f
 * A[f,p] B[f,g]) + the 60 coset sums cor[a] = sum_g C[T[a,g], g]: the permutation addresses the result instead of every multiply-add
 * (csrc/corr_mfma.hip; HBM-bound instead of LDS-bound).  The literal kernel's function in another summation order (float32 throughout,
 * ~1e-6 |A||B| apart): for the correlation as a FEATURE (the stacked matcher's R_indicator); no arg-max. */
int roreg_group_corr_mfma(const float *perm_feats, const int64_t *perm_rows, const float *bcast_feats, const int64_t *bcast_rows, int M,
                          int transpose_table, float *cor_out, void *stream);

/* The same arg-max (test/estimator.py:85-89, bit for bit) with 10x fewer operations: the 60 correlations are first BOUNDED in the irrep
 * domain of the icosahedral group -- cor[a] = sum_rho sum_ij rho(a)[j][i] (sum_f X2_f(rho) X1_f(rho)^T)[i][j], sum_d d^3 = 244
 * multiply-adds per channel instead of 3600, from per-keypoint coefficients computed once per cloud (roreg_feat_coefs) -- and only the
 * candidates within a rigorous margin of the bound's maximum (near ties, duplicates) are re-evaluated with the literal formula in the
 * reference's evaluation order from the group-domain rows; the first maximum of those is returned.
 * coef1 / coef0 = roreg_feat_coefs(feats1 / feats0) [*,32,60] f32; feats* are float32 or (feat_bf16) bfloat16 [*,32,60].
 * roreg_set_des2r_tables uploads the index / representation tables (roreg_amd/fourier.py derives them from the multiplication table and
 * checks the identity to 1e-12): ia, ib uint8 [60][5] (coefficient indices of the k-th product term of entry q = (rho,i,j)), cnt uint8 [60]
 * (terms = d), NT float32 [60 (q)][60 (a)] = rho(a)[j][i].  roreg_des2r_recheck_count: how many correspondences took the exact path. */
int roreg_set_des2r_tables(int transpose_table /* 0: x[P[a,.]] (Des2R); 1: x[P[.,a]] (R_indicator) */, const uint8_t *ia_host,
                           const uint8_t *ib_host, const uint8_t *cnt_host, const float *NT_host);
/* All 60 correlations of roreg_group_corr from the coefficient rows alone (no arg-max, no re-check): cor_out [M,60] f32, equal to the
 * literal float32 evaluation to its rounding level (~1e-6 |d1||d2|).  For the matcher's R_indicator feature (network/rot_coh_match.py:154-163). */
int roreg_group_corr_irrep(const float *perm_coefs, const int64_t *perm_rows, const float *bcast_coefs, const int64_t *bcast_rows, int M,
                           int transpose_table, float *cor_out, void *stream);
int roreg_des2r_irrep(const float *coef1, const int64_t *rows1, const float *coef0, const int64_t *rows0, const void *feats1,
                      const void *feats0, int feat_bf16, int M, int64_t *idx_out, void *stream);
int roreg_des2r_recheck_count(int reset, int32_t *count_out);
/* out[b,c,q] (f32 [B,C,60]) = sum_g F[q][g] x[b,c,g]: the orthonormal group-Fourier coefficients of a group-domain tensor x [B,C,60]
 * (float32, or bfloat16 with x_bf16) in per-keypoint layout -- the operand of roreg_des2r_irrep.  split as roreg_ft_nonlin. */
int roreg_feat_coefs(const void *x, int x_bf16, float *out, int B, int C, int split, void *stream);

/* Build the ET network input x [M,128,60] = cat(before1[r1][:, :, P[a]], before0[r0], after1[r1][:, :, P[a]],
 * after0[r0]) for correspondence rows (r0,r1) and anchor a=pre_idx[b].
 * Replaces batch_create + the per-row permutation loop (test/estimator.py:293-306; network/eqv_trans.py:126-129). */
int roreg_et_gather(const void *before0, const void *before1, const void *after0, const void *after1,
                    int feat_bf16 /* the four feature tensors are bfloat16 instead of float32 (BASELINE config 5) */,
                    const int64_t *rows0, const int64_t *rows1, const int64_t *pre_idx, int M,
                    float *x_out, void *stream);

/* q [M,4] f32 (un-normalised head output) -> q/||q||; R = R(q) (fp32 arithmetic as utils/r_eval.py:90-106
 * on float32 inputs) promoted to f64, times Rgroup[a] (the float32-rounded table, as test/estimator.py:279);
 * t = key0 - key1 R^T (f64).  Trans_out f64 [M,3,4]; quat_out f32 [M,4] optional (normalised).
 * Replaces eqv_trans.py:137 and test/estimator.py:350-366. */
int roreg_quat_to_trans(const float *q, const int64_t *anchor, const double *keys0, const int64_t *rows0,
                        const double *keys1, const int64_t *rows1, int M,
                        double *Trans_out, float *quat_out, void *stream);

/* One-shot RANSAC scoring (fp64, no FMA).  For hypothesis h (3x4 row-major in Trans[hyp_rows[h]], hyp_rows
 * int64, NULL = identity): overlap[h] = sum_{i: ||k0_i - (R k1_i + t)||^2 < ird^2} w_i / M, accumulated in
 * increasing i.  k0,k1 f64 [M,3] (already gathered by match), w f64 [M].  best_out (device int32[1]) gets
 * the first h with the strictly greatest overlap (estimator.py:430-436; -1 if every overlap is 0);
 * mask_out (uint8 [H,M], optional) the inlier masks.
 * w_f32 != 0: the scores are the rotation-coherence matcher's float32 array (test/matcher.py:210; every w_i is a float32 value): the
 * reference's np.sum(scores[inliers]) is then numpy's PAIRWISE float32 reduction over the compacted inlier array and its quotient by M a
 * float32 division (test/estimator.py:381), which the kernel rebuilds bit for bit, so that hypotheses whose overlaps tie only after
 * float32 rounding keep the reference's order under the strict `>` of :433; overlap_out holds those float32 values widened.
 * Replaces yohoo_ransac.overlap_cal and the hypothesis loop (test/estimator.py:377-382,426-436). */
int roreg_ransac_score(const double *k0, const double *k1, const double *w, int w_f32, int M,
                       const double *Trans, const int64_t *hyp_rows, int H, double ird,
                       double *overlap_out, int32_t *best_out, uint8_t *mask_out, void *stream);

/* Refinement step: inliers of T_in (3x4 taken from T_in, or from Trans[hyp_rows[*best]] when best!=NULL)
 * at threshold `dist`; weights w/sum(w); weighted centroids; H = (k0-c0)^T diag(w) (k1-c1); R = U V^T of
 * the 3x3 SVD (no reflection guard); t = c0 - c1 R^T.  T_out f64 [4,4].  The SVD runs on the device
 * (one-sided Jacobi); when H is rank-deficient (<= 2 inliers, collinear inliers) U V^T is not unique and the
 * reference's value is whatever LAPACK's null-space basis gives, so stats_out exposes H, the centroids and
 * the weight sum for a host LAPACK evaluation of exactly that case.  w_f32 != 0 (float32 scores, as in roreg_ransac_score): the
 * normalisation scores / np.sum(scores) runs in float32 like the reference's (:50), pairwise sum and per-weight float32 quotient.
 * Replaces refiner.Refine_trans (test/estimator.py:28-72). */
int roreg_refine(const double *k0, const double *k1, const double *w, int w_f32, int M,
                 const double *T_in, int t_in_stride /* 4 for 3x4/4x4 rows */,
                 const double *Trans, const int64_t *hyp_rows, const int32_t *best,
                 double dist, double *T_out, double *stats_out /* optional f64[16]: H(9), c0(3), c1(3), sum w */,
                 void *stream);

/* The local-transform stage (Des2R + ET input assembly, then quaternion -> 3x4 transform) of every pair of a scene in 2 + 1 launches
 * around the one batched ET trunk pass (test/estimator.py:85-111,293-366 per pair; same arithmetic as roreg_des2r / roreg_et_gather /
 * roreg_quat_to_trans, bit-identical).  Task p evaluates its n correspondences matches[sel[i]] (sel NULL = the first n rows); output
 * rows [off, off+n): dr_out int64, x_out [*,128,60] f32 (ET input; NULL = Des2R only, the YOHO-C estimator's DR_index),
 * Trans_out [*,3,4] f64.  tasks_dev is a DEVICE array. */
typedef struct {
    const void *before0, *before1, *after0, *after1;   /* the clouds' group features [*,32,60]: float32, or bfloat16 with flags bit 1 */
    const double *keys0, *keys1;
    const int64_t *matches;
    const int64_t *sel;
    int32_t n, pad_;
    int64_t off;
    const float *coef0, *coef1;                        /* roreg_feat_coefs(after0 / after1) [*,32,60] f32, needed with flags bit 0 */
} roreg_lt_task;
/* flags: bit 0 = Des2R through the irrep-domain bound + exact re-check (roreg_des2r_irrep; tasks carry coef0 / coef1), else the literal
 * kernel; bit 1 = the features are bfloat16 (needs bit 0). */
int roreg_lt_prepare_batch(const roreg_lt_task *tasks_dev, int n_tasks, int max_n, int flags, int64_t *dr_out, float *x_out,
                           const float *bn_scale /* [128] */, const float *bn_shift, float *x_bound_out /* nullable, with x_out: per output row
                           sqrt(60) max |ReLU(bn_scale_c x + bn_shift_c)| = roreg_row_bound(x_out, bn) computed while assembling (the block scale of the
                           fp16 x 2 Conv_init GEMM, network/eqv_trans.py:88) */, void *stream);
int roreg_lt_finish_batch(const roreg_lt_task *tasks_dev, int n_tasks, int max_n, const float *q_all, const int64_t *dr_all,
                          double *Trans_out, void *stream);

/* The estimator tail of every pair of a scene in five launches: gather the matched keypoints, score the <= max_iter hypotheses
 * (one wavefront each), first strictly-greatest overlap, refine at 2*ird from the winning local transform, refine at ird from that
 * (test/estimator.py:405-443 per pair; same arithmetic order as roreg_ransac_score / roreg_refine, so results are bit-identical).
 * tasks_dev: DEVICE array; koff = prefix sum of M over the tasks; total_M = sum of M; max_M / max_H >= every M / H.
 * Outputs per task: best_out[p] (hypothesis index or -1), T1/T2 [p][16] row-major 4x4, stats1/stats2 [p][16] as roreg_refine. */
typedef struct {
    const double *keys0, *keys1;   /* keypoints of the two clouds [*,3] f64 */
    const int64_t *matches;        /* [M,2] interleaved rows (cloud 0, cloud 1) */
    const double *w;               /* [M] match scores, NULL = ones (test/matcher.py:109) */
    const double *Trans;           /* [*,3,4] f64 local transforms */
    const int64_t *hyp_rows;       /* [H] rows of Trans in hypothesis order, NULL = identity */
    int32_t M, H;
    int64_t koff;
} roreg_ransac_task;
size_t roreg_ransac_batch_workspace(int n_tasks, long long total_M, int max_H);
int roreg_ransac_batch(const roreg_ransac_task *tasks_dev, int n_tasks, long long total_M, int max_M, int max_H, double ird,
                       int w_f32 /* the tasks' scores are float32 values: numpy's float32 reductions, see roreg_ransac_score */,
                       int32_t *best_out, double *T1_out, double *stats1_out, double *T2_out, double *stats2_out,
                       void *workspace, size_t workspace_bytes, void *stream);

/* v4: one more refinement at `dist`, from the given transforms T_in [n_sel][16], of the tasks sel_dev[0..n_sel) (device int32) of an EARLIER
 * roreg_ransac_batch call: tasks_dev, total_M and workspace are that call's (its workspace still holds the gathered keypoints).  T_out /
 * stats_out [n_sel][16] as roreg_refine.  One launch for all of them: the engine's second refinement of the pairs whose first one had a
 * rank-deficient covariance and was closed by host LAPACK (test/estimator.py:53-72 per pair). */
int roreg_refine_batch(const roreg_ransac_task *tasks_dev, const int32_t *sel_dev, int n_sel, long long total_M, const double *T_in,
                       double dist, int w_f32, double *T_out, double *stats_out, const void *workspace, void *stream);

/* Seeded shuffles, HOST function (no device work): for every job j, `np.random.seed(seeds[j])` followed by, for each of its per_job lists
 * (sizes int32 [n_jobs][per_job]), `idx = np.arange(n); np.random.shuffle(idx); idx[:take]` -- numpy's legacy MT19937 stream and its
 * Fisher-Yates shuffle, replayed bit for bit.  out int64 [n_jobs][per_job][take] (-1 beyond a list's length).  Replaces the per-pair
 * Python loops of the matcher's keypoint sampling (test/matcher.py:83-88: two lists per pair, take = keynum) and of the one-shot
 * estimator's hypothesis order (test/estimator.py:423-425: one list per pair, take = max_iter) when every pair has a generator stream of
 * its own; jobs are spread over n_threads host threads. */
int roreg_mt_shuffle_prefix(const uint32_t *seeds, int n_jobs, const int32_t *sizes, int per_job, int take, int64_t *out, int n_threads);

/* v6.  The same shuffles from ONE running stream, HOST function: the process-global generator an unseeded Test.py consumes (test/matcher.py:83-88,
 * test/estimator.py:423-425).  key[624] / *pos: np.random.get_state()'s MT19937 key and position on entry, the stream's state after the last
 * list on return (for np.random.set_state); `for n in sizes: idx = np.arange(n); np.random.shuffle(idx); idx[:take]`, list after list on one
 * thread.  out int64 [n_lists][take] (-1 beyond a list's length). */
int roreg_mt_stream_shuffle_prefix(uint32_t *key, int32_t *pos, const int32_t *sizes, int n_lists, int take, int64_t *out);

/* YOHO-C hypothesis draws, HOST function (no device work): replays the generator calls of the reference's sampling loop
 * (test/estimator.py:220-230: np.random.choice(range(60), p=prob), then np.random.choice(bin_members, 3)) over a block of raw MT19937
 * words drawn by the caller from the global generator.  cdf f64 [60] = prob.cumsum()/prob.sum(); bin_size int32 [60] = members per
 * rotation bin; at most max_iter hypotheses and max_tries + 1 tries.  bin_out int32 [max_iter], pick_out int64 [max_iter,3] (positions
 * inside the bin's member list).  Returns 3 when the block is too short (retry with more words); *words_used = words consumed. */
int roreg_yohoc_draw(const uint32_t *words, long long n_words, const double *cdf, const int32_t *bin_size, int max_iter, int max_tries,
                     int32_t *bin_out, int64_t *pick_out, int32_t *n_hyp_out, long long *words_used);

/* v4: row gathers of several (source, row list) pairs in ONE launch -- task t copies rows[0..n) of src (row_bytes each, a multiple of 8) to
 * consecutive rows of dst; tasks_dev is a DEVICE array, max_n >= every n.  The stacked matcher's per-pair sample gathers
 * (feats[sample], keys[sample]: test/matcher.py:187-197 for every pair of a group). */
typedef struct {
    const void *src;
    const int64_t *rows;
    void *dst;
    int32_t n, pad_;
} roreg_gather_task;
int roreg_gather_rows_batch(const roreg_gather_task *tasks_dev, int n_tasks, int max_n, int row_bytes, void *stream);

/* Gather rows: out[i] = src[rows[i]] for f64 [.,3] keypoints (estimator.py:407-408). */
int roreg_gather_rows_f64(const double *src, const int64_t *rows, int M, int width, double *out, void *stream);

/* ---- rotation-coherence matcher (Match_ot, network/rot_coh_match.py) ---------------------------------
 * Per-point tensors are position-major: [points, channels] or [points, k, channels] float32.
 *
 * roreg_topk_dot: for every row of A [m,32] the k (16, 8 or 1) rows of B [n,32] with the largest dot product, descending,
 * lower index first on ties; replaces score_mat + the full descending argsort of Knn_index_extract
 * (rot_coh_match.py:8-12,34-45) without materialising the m x n matrix.  ws: roreg_topk_dot_workspace_size floats.
 *
 * Several pairs per launch ("segments"): the per-point tensors of the pairs are concatenated, seg* are DEVICE int32 [n_seg+1] row
 * offsets (in points), and every per-pair quantity (neighbour search, InstanceNorm statistics, column maxima, Sinkhorn) stays inside
 * its pair with the arithmetic of the one-pair call, bit for bit.  In roreg_topk_dot a row of A's segment p searches B's segment p and
 * idx_out holds GLOBAL rows of B; max_m / max_n = the largest segment.  seg pointers NULL = one pair (the remaining segment arguments
 * are ignored). */
size_t roreg_topk_dot_workspace_size(int m, int n, int k);
int roreg_topk_dot(const float *A, int m, const float *B, int n, int k, int64_t *idx_out, float *val_out /* optional [m,k] */,
                   float *ws, size_t ws_floats, const int32_t *segA, const int32_t *segB, int n_seg, int max_m, int max_n, void *stream);

/* y [L,Cout] = x [L,Cin] W^T + b  (the 1x1 Conv2d layers: attention projections / merge, first and residual convs of
 * mlp_2layer and Contextnorm; rot_coh_match.py:14-32,63-81,95-119): one float32 fmaf chain per (row, output), inputs ascending, starting from the bias.
 * v5: that chain runs on the matrix cores (v_mfma_f32_32x32x2_f32 is a float32 fmaf chain over k on gfx950, bit for bit: csrc/linear_chain.hip);
 * roreg_linear_path(1) selects the vector-pipe kernels instead (same bits; returns the previous setting, 0 = matrix cores).
 * v6: the matrix-core kernels are software-pipelined (lc2_kernel); roreg_linear_path(2) = the matrix cores through round 5's kernels (A/B, tests).
 * Every entry point that rests on the fma-chain property (roreg_linear / _cat3, roreg_mlp_tail, roreg_mlp_head, roreg_topk_dot's MFMA kernel, the
 * matrix-free read-out of roreg_sinkhorn_batch3) verifies it ONCE per process on the device (1024 outputs: cancelling pairs, zeros, subnormal
 * products) in front of its first launch and returns 4 with roreg_last_error() naming the vector-pipe switches if the hardware disagrees. */
int roreg_linear(const float *x, int L, int Cin, const float *W /* [Cout,Cin] */, const float *b, int Cout, float *y, void *stream);
int roreg_linear_path(int path);
/* v5: y [m * k, Cout] (Cout = 64 | 32) = W [pos[r] (32) | table[idx[r]] (32) | conf[r / k] (32)] + b -- roreg_linear on the value MLP's input
 * (rot_coh_match.py:95-119) without materialising its [m * k, 96] rows: the kernel's row staging reads the three sources.  The same chains on the same
 * values: bitwise roreg_linear on the concatenated rows.  Matrix-core path only (with roreg_linear_path(1) build the rows and call roreg_linear). */
int roreg_linear_cat3(const float *pos /* [m*k,32] */, const float *table /* [n,32] */, const int64_t *idx /* [m*k] rows of table */,
                      const float *conf /* [m,32] */, int m, int k, const float *W /* [Cout,96] */, const float *b, int Cout, float *y, void *stream);
/* v4: the same layer on the matrix cores for Cin >= 32 (other shapes: roreg_linear): fp16 hi + lo operands under exact per-row / per-tensor
 * power-of-two scales, all four cross products, f32 accumulate (<= 6e-7 of sum |w||x| per element: the level of the fmaf chain, other
 * rounding); one kernel for every L, a row's result depends on that row alone.  csrc/linear_mfma.hip; used by the stacked matcher. */
int roreg_linear_mfma(const float *x, int L, int Cin, const float *W /* [Cout,Cin] */, const float *b, int Cout, float *y, void *stream);

/* InstanceNorm2d(affine=False) statistics of h [L,C] over all L positions -> mean_rstd [2C] = mean, 1/sqrt(var_biased+eps).
 * ws: 2*C*256 doubles.  (rot_coh_match.py:19,68)   With segments (seg_off in points, `mult` rows of h per point): one statistic
 * per pair, mean_rstd [n_seg][2C], ws n_seg*2*C*256 doubles. */
int roreg_instnorm_stats(const float *h, int L, int C, float eps, float *mean_rstd, double *ws, const int32_t *seg_off, int n_seg,
                         int mult, void *stream);

/* y [L,32] += W2 relu((h - mean) * rstd) + b2  (closing conv of mlp_2layer / Contextnorm; y already holds the residual conv);
 * with segments every row uses the statistics of its pair. */
int roreg_mlp_tail(const float *h, int L, int Cmid, const float *mean_rstd, const float *W2 /* [32,Cmid] */, const float *b2,
                   float *y, const int32_t *seg_off, int n_seg, int mult, void *stream);
/* v4: roreg_mlp_tail on the matrix cores (as roreg_linear_mfma; the normalisation + ReLU are applied while the operand is split). */
int roreg_mlp_tail_mfma(const float *h, int L, int Cmid, const float *mean_rstd, const float *W2 /* [32,Cmid] */, const float *b2,
                   float *y, const int32_t *seg_off, int n_seg, int mult, void *stream);

/* ctx [L,120] = [R [L,60] | max over the points of the row's pair of R]  (Self_attention_block's ambiguity context,
 * rot_coh_match.py:201-202).  ws: n_seg*(256+1)*60 floats. */
int roreg_context_colmax(const float *R, int L, const int32_t *seg_off, int n_seg, int max_len, float *ctx_out, float *ws, void *stream);

/* Core of MultiHeadedAttention(4 heads, d_model 32) on k-NN neighbourhoods (rot_coh_match.py:84-119): qp [m,32] projected
 * queries; kp / vp projected keys / values, either dense [m,k,32] or a per-point table [n,32] addressed through idx [m,k].
 * x_out [m,32] (input of the merge conv).  Channel c belongs to head c%4, dim c/4 (the reference's .view(b,8,4,-1)). */
int roreg_knn_attention(const float *qp, const float *kp, const float *vp, const int64_t *idx, int k_is_table, int v_is_table,
                        int m, int k, float *x_out, void *stream);

/* Small data-movement / normalisation steps of the matcher graph (see csrc/rm.hip for the op list). */
int roreg_rm_elementwise(int op, const float *a, const float *b, const float *c, const int64_t *idx, int L, int k, int C,
                         float *out, float *ws, void *stream);

/* Log-domain Sinkhorn with dustbins + mutual-argmax read-out (rot_coh_match.py:285-314,363,369-379).  The coupling scores
 * <src_final[i], tgt_final[j]> ([m,32] x [n,32]) are formed on the fly; Z_out [(m+1),(n+1)]; matches0 [m], matches1 [n] (-1 = unmatched), mscores0/1 (exp of the row maximum for mutual matches).
 * ws: roreg_sinkhorn_workspace_size floats (the coupling matrix and its transpose stay resident across the 2*iters passes). */
size_t roreg_sinkhorn_workspace_size(int m, int n);
int roreg_sinkhorn(const float *src_final, int m, const float *tgt_final, int n, float alpha, int iters, float *Z_out,
                   int64_t *matches0, int64_t *matches1, float *mscores0, float *mscores1, float *ws, size_t ws_floats, void *stream);

/* The same for several pairs per launch (2*iters + 5 launches for all of them): descriptors concatenated by seg_src / seg_tgt (device
 * and host copies of the int32 offsets), read-outs concatenated the same way with indices LOCAL to the pair; the coupling matrix
 * itself is not returned.  Each iteration reads the matrix once (rows in the log domain, column sums by fma of the same exponentials);
 * read-outs equal the one-pair call's (indices identical on the tests, scores to rounding).  consts: DEVICE copy of the 4*n_seg floats of roreg_sinkhorn_batch_consts (a host function: the host
 * logf values the one-pair call uses). */
size_t roreg_sinkhorn_batch_workspace_size(int n_seg, int max_m, int max_n, long long total_m, long long total_n);
int roreg_sinkhorn_batch_consts(const int32_t *seg_src_host, const int32_t *seg_tgt_host, int n_seg, float *consts_host);
int roreg_sinkhorn_batch(const float *src_final, const float *tgt_final, const int32_t *seg_src, const int32_t *seg_tgt,
                         const int32_t *seg_src_host, const int32_t *seg_tgt_host, const float *consts, int n_seg, float alpha, int iters,
                         int64_t *matches0, int64_t *matches1, float *mscores0, float *mscores1, float *ws, size_t ws_floats,
                         void *stream);

/* v4: roreg_sinkhorn_batch with the choice of how the iterations see the coupling matrix.  recompute = 0: the function above (the matrix is
 * materialised and read once per iteration: 4 (m+1)(n+1) bytes per pair and iteration from HBM).  recompute = 1: the iterations never read
 * a matrix -- every pass recomputes the scores <s_i, t_j> on the matrix cores from the L2-resident descriptors (fp16 hi + lo operands, f32
 * accumulate; potentials, dustbins and padding ride in one more MFMA) and only exponentiates and adds (csrc/ot_flash.hip); same read-outs
 * (indices identical on the tests, scores to 1e-6).  The matrix is still built once per pair for the read-out.  A pair whose TARGET cloud
 * has at most 2559 points (yoho_mat's default keynum 2500, test/matcher.py:152) recomputes the scores ONCE per iteration (a 32-row strip's
 * exponentials stay in registers between the row and the column sums); a longer one (`Test.py --keynum 5000`, test/evaluator.py:20,46)
 * twice (rows, then columns) -- or, recompute = 2, once, with two cooperating workgroups per strip up to 5119 points.  The form is chosen
 * per pair, so a pair's result does not depend on what is stacked beside it.
 * ws: roreg_sinkhorn_batch2_workspace_size floats. */
size_t roreg_sinkhorn_batch2_workspace_size(int n_seg, int max_m, int max_n, long long total_m, long long total_n);
int roreg_sinkhorn_batch2(const float *src_final, const float *tgt_final, const int32_t *seg_src, const int32_t *seg_tgt,
                          const int32_t *seg_src_host, const int32_t *seg_tgt_host, const float *consts, int n_seg, float alpha, int iters,
                          int64_t *matches0, int64_t *matches1, float *mscores0, float *mscores1, float *ws, size_t ws_floats,
                          int recompute, void *stream);
/* v5: the same, and (Z_out non-null, n_seg == 1) the pair's log-coupling matrix Z [(m+1) x (n+1)] row-major = ((Z0 + u) + v) - norm, the
 * `scores` of Match_ot.forward (network/rot_coh_match.py:313,366) -- so that forward() and the stacked path run ONE set of Sinkhorn kernels.
 * With recompute != 0 and Z_out == NULL nothing materialises a matrix: the mutual arg-max read-out (rot_coh_match.py:369-379) is taken from float32
 * MFMA chains over the descriptors (bitwise the fmaf chains the matrix held) with the potentials added in the reference's association, and the
 * workspace (roreg_sinkhorn_batch3_workspace_size) shrinks from 2 (m+1)(n+1) floats per pair to ~4 (m+n). */
size_t roreg_sinkhorn_batch3_workspace_size(int n_seg, int max_m, int max_n, long long total_m, long long total_n, int recompute, int want_Z);
int roreg_sinkhorn_batch3(const float *src_final, const float *tgt_final, const int32_t *seg_src, const int32_t *seg_tgt,
                          const int32_t *seg_src_host, const int32_t *seg_tgt_host, const float *consts, int n_seg, float alpha, int iters,
                          int64_t *matches0, int64_t *matches1, float *mscores0, float *mscores1, float *ws, size_t ws_floats,
                          int recompute, float *Z_out, void *stream);
/* v6 (host function): write n files -- file q = headers[q] (header_len[q] bytes, e.g. a .npy header) followed by data[q] (nbytes[q] bytes) -- on
 * n_threads host threads: the device-resident engine's StageFileWriter leaves the reference's ~1800 small per-pair .npy files of a 449-pair scene
 * (test/matcher.py:108-109, test/estimator.py:111,367) through it instead of one np.save each.  Returns 0, or 1 + the index of the first failure. */
int roreg_write_files(const char *const *paths, const void *const *headers, const int32_t *header_len, const void *const *data,
                      const int64_t *nbytes, int n, int n_threads);
/* v6 (host function): the YOHO-C hypothesis draws (test/estimator.py:119-137, 214-230) of n_pairs pairs, pair p from its own generator stream
 * np.random.RandomState(seeds[p]): anchors = the pairs' coarse rotations (Des2R index of the correspondences hypotheses are drawn from),
 * concatenated, pair p at [offsets[p], offsets[p+1]).  rows_out [n_pairs][max_iter][3]: the three correspondences (positions in the pair's list)
 * of each hypothesis; n_hyp_out [n_pairs]: how many were drawn, or -1 when no rotation bin holds two correspondences -- the reference then
 * answers np.random.rand(4, 4) (giveup_out [n_pairs][16], recalltime 50000).  Same words of the same streams as the Python loop, on n_threads. */
int roreg_yohoc_draw_many(const uint32_t *seeds, int n_pairs, const int64_t *anchors, const int64_t *offsets, int max_iter, int max_tries,
                          int64_t *rows_out, int32_t *n_hyp_out, double *giveup_out, int n_threads);
/* v6: mlp_2layer / Contextnorm (network/rot_coh_match.py:14-32, 63-81): the first convolution (Cin -> C1, output h [L, C1]) and the residual
 * branch (Cin -> 32, output y [L, 32]) in ONE launch -- the float32 fmaf chains of roreg_linear on the matrix cores, bit for bit, the input tile
 * staged once, the next tile's rows in flight under the chains -- plus the per-pair InstanceNorm statistics of h (mean_rstd [n_seg][2 C1], the
 * operand of roreg_mlp_tail) from per-tile float64 channel sums added in a fixed per-pair order: roreg_instnorm_stats' numbers in another
 * association (float64 sums rounded to float32: the same floats but for an ulp once in ~1e8 values).  x [L, Cin], or x == NULL and
 * pos / table / idx / conf (m, k; L = m k; Cin = 96): the value MLP's rows as in roreg_linear_cat3.  Served: (Cin, C1) = (3, 64), (64, 64),
 * (96, 64), (120, 128) on the matrix-core path; returns 3 and launches nothing otherwise (call roreg_linear x 2 + roreg_instnorm_stats).
 * ws: roreg_mlp_head_workspace(L, n_seg, C1) doubles. */
size_t roreg_mlp_head_workspace(int L, int n_seg, int C1);
int roreg_mlp_head(const float *x, const float *pos, const float *table, const int64_t *idx, const float *conf, int m, int k, int L, int Cin,
                   const float *W1, const float *b1, int C1, const float *Wr, const float *br, float *h, float *y, const int32_t *seg_off,
                   int n_seg, int mult, float eps, float *mean_rstd, double *ws, void *stream);
/* v6: the recomputed iterations stop PER PAIR at the fixed point of the float32 iteration u <- log_mu - LSE(Z + v), v <- log_nu - LSE(Z + u)
 * (network/rot_coh_match.py:285-292).  Every iteration records the largest step any of the pair's m + n + 2 potentials took, in units of
 * max(2^-22 |u|, 2^-20) (log2 units: 2 .. 4 units in the last place of the float32 potential); the pair's remaining iterations are skipped once
 * (a) an iteration's largest step is <= 1 unit, or (b) it is <= 8 units and not smaller than the previous iteration's: the
 * potentials then only jitter in their last bits (the recomputed scores are re-rounded whenever a potential moves by an ulp).  The reference
 * always runs `iters` (= 100) iterations; past that point they move nothing but those bits, so matches are unchanged and scores agree to
 * float32 noise (tests/test_hip_rm.py and tests/test_hip_fullsize.py hold both settings to the reference's goldens).  A pair whose potentials
 * keep moving runs all `iters`; a sequence that still converges, however slowly, shrinks its steps monotonically and never satisfies (b).  roreg_sinkhorn_early_exit(on): 1 = stop settled pairs (default; ROREG_OT_EARLY_EXIT=0 in the
 * environment starts with 0), 0 = always `iters` iterations, < 0 = query; returns the previous setting.  roreg_sinkhorn_iteration_stats: sum
 * of the iterations run and number of pairs over the recomputed-iteration calls since the last reset (synchronises `stream`; host pointers). */
int roreg_sinkhorn_early_exit(int on);
int roreg_sinkhorn_iteration_stats(long long *iterations, long long *pairs, int reset, void *stream);

/* ---- group-Fourier evaluation of the group convolution (csrc/fourier.hip, roreg_amd/fourier.py) ---------------
 * In the basis of the five real irreps (d = 1,3,3,4,5) the 13-stencil group conv is one dense GEMM per irrep,
 *   Out_rho [d*O][d*B] = W_rho [d*O][d*C] . X_rho [d*C][d*B]      (row-major, keypoints fastest),
 * 244 instead of 780 multiply-adds per (o,c) pair; same network function as roreg_group_conv (network/group_feat.py:16-33).
 * roreg_set_fourier_tables: F [60 (q = (rho,i,l))][60 (g)], the orthonormal transform (host pointer).
 * roreg_irrep_gemm_tiles:   fills / counts the (irrep, m-tile, n-tile) work list of one layer.
 * roreg_irrep_gemm:         the five GEMMs in one launch; X/Out/Wpack are host arrays of 5 device pointers; Wpack[rho] is
 *                           roreg_group_conv_pack_weights(KS=1) of the [round_up(d*O,128)][d*C] matrix.  Add (nullable): 5 device
 *                           pointers shaped like Out; Out = W.X + Add (the residual short cut of network/ops.py:46-64 taken in the
 *                           irrep domain).
 * roreg_ft_nonlin:          per (keypoint, channel): inverse transform (or group-domain input) -> + bias (+ bias2)
 *                           (+ group-domain residual) -> BatchNorm(eval)+ReLU (optional) -> forward transform (or group-domain
 *                           output [B,C,60], or only the Lout columns selected by g_map -- the ET trunk keeps 45 live columns).
 *                           Coefficients are one flat buffer of 60*C*Bp floats, Bp = B rounded up to 32: irrep rho occupies
 *                           [off_rho*C*Bp, off_{rho+1}*C*Bp) as the row-major GEMM operand [d*C][d*Bp] (row (l,c); columns blocked by
 *                           32 keypoints: column = (b/32)*(32*d) + i*32 + b%32; pad keypoints hold zeros).  roreg_irrep_gemm* is
 *                           called with B := Bp. */
int roreg_set_fourier_tables(const float *F_host);
size_t roreg_irrep_gemm_tiles(int O, int B, int32_t *tiles_host);            /* 128-row m-tiles */
size_t roreg_irrep_gemm_tiles_m(int O, int B, int tile_m /* 128 | 256 */, int32_t *tiles_host);
int roreg_irrep_gemm(const float *const *X, float *const *Out, const float *const *Add, const float *const *Wpack, int C, int O, int B,
                     const int32_t *tiles_dev, int n_tiles, void *stream);
/* Same GEMMs on the bf16 matrix cores with f32 accuracy: every operand is split into three bf16 pieces and the six cross products
 * of order <= 4 are accumulated in f32 (error at the f32 rounding level; 2.67x fewer matrix-core cycles than the f32-input MFMA).
 * Wsplit[rho]: uint16 bf16 bits, layout [3 splits][d*C/16][2 k-octets][round_up(d*O,128)][8]. */
int roreg_irrep_gemm_split(const float *const *X, float *const *Out, const float *const *Add, const void *const *Wsplit, int C, int O, int B,
                           const int32_t *tiles_dev, int n_tiles, void *stream);
/* The same GEMMs with fp16 x 2 operands and power-of-two block scaling (half the matrix-core work of the bf16 x 3 split).  The block is
 * ONE KEYPOINT: all coefficients of keypoint b (every channel, every irrep) share the scale 2^e(b), e(b) = 14 - exponent(x_bound_dev[b]),
 * where x_bound_dev [B] is a bound on |coefficient| known BEFORE the coefficients exist (below), so the producer roreg_ft_nonlin(split = 2)
 * writes them already split -- X[rho] holds 32-bit words fp16(x 2^e) | fp16(x 2^e - hi) << 16 at the float pitch -- and the GEMM only
 * permutes bytes while staging.  hi + lo keep 22 significant bits of every operand within ~11 binades of the bound and an absolute error
 * below 2^-39 of the bound beneath that; products hi.hi + hi.lo + lo.hi, f32 accumulate, exact rescale per column by 2^-(e(b) + w_exp).
 * A keypoint's output depends on its own column only: results are independent of batch composition (test/extractor.py:51-58).
 * Bound propagation: with next_u_dev / next_v_dev [O] and out_bound_dev [B] (zeroed by the caller) the epilogue reduces
 * max over (o, q) of u_o |T_oq(b)| + v_o per keypoint (atomic max), which bounds the coefficients of the NEXT transform's output when
 * u_o = 60 |bn_scale_o| and v_o = sqrt(60) (|bn_scale_o| |bias_o| + |bn_shift_o|) (orthonormal transform: |IFT(T)(g)| <= sqrt(60) max_q |T_q|,
 * |FT(x)_q| <= sqrt(60) max_g |x(g)|).  Wsplit2[rho]: fp16 bits, layout [2 (hi, lo)][d*C/16][2 k-octets][round_up(d*O,128)][8]. */
int roreg_irrep_gemm_f16x2(const float *const *X, float *const *Out, const float *const *Add, const void *const *Wsplit2,
                           const float *x_bound_dev, int w_exp, const float *next_u_dev, const float *next_v_dev, float *out_bound_dev,
                           int C, int O, int B, const int32_t *tiles_dev, int n_tiles,
                           int tile_m /* 128 | 256 (O % 256 == 0): the m-tile the list was built with; 256 = 8-wave workgroups */,
                           int x_planes /* X[rho] is in HALF-BLOCK layout (roreg_ft_nonlin out_planes): row pitch and 32-column blocks of the word layout, but every
                           128-byte block holds its 32 fp16 hi values followed by its 32 lo values (each half in the column order 0, 16, 1, 17, ...) instead of 32
                           words hi | lo << 16; the
                           activations then reach LDS by LDS-DMA and the matrix cores through transposing LDS reads, no register staging
                           (needs tile_m = 256); x_planes = 1: 32x32x16 MFMAs, results bitwise those of the word layout; x_planes = 2 (round 4): the same
                           operands and LDS images under v_mfma_f32_16x16x32_f16 (K = 32 per step: ~4 % faster, the same error bound, last bits differ) */, void *stream);
/* v5: how the x_planes = 2 kernel is launched.  0: one 8-wave workgroup per 256 x 256 tile (round 4).  1: PERSISTENT workgroups, one per CU, that walk
 * the list's per-XCD streams and request the next tile's first operand stages before they store the finished one (irrep_gemm_xdma16p_kernel,
 * csrc/fourier.hip).  2: 4-wave workgroups on 256 x 128 HALF tiles in 80 KB of LDS, two per CU, so that one's prologue and epilogue run under the
 * other's loop (irrep_gemm_xdma16h_kernel).  The same MFMA sequence per output element in all three: bitwise the same results.  Other values:
 * query.  Returns the previous setting; the initial one is the environment's ROREG_GEMM_PERSIST (unset = the default). */
int roreg_gemm_persistent(int on);
/* bound_out[b] (b < round_up(B,32); 0 for pad keypoints) = sqrt(60) max_{c,g} |act(x[b,c,g])| >= every coefficient of FT(act(x[b])), act =
 * ReLU(bn_scale_c x + bn_shift_c) or the identity (bn NULL): the x_bound of a layer whose input is a group-domain tensor [B,C,60]. */
int roreg_row_bound(const void *x_spatial, int x_bf16 /* x is bfloat16 instead of float32 */, const float *bn_scale, const float *bn_shift,
                    float *bound_out, int B, int C, void *stream);
int roreg_ft_nonlin(const float *Xin /* flat [60*C*B] */, const float *x_spatial, const float *bias,
                    const float *bias2, const float *bn_scale, const float *bn_shift, const float *resid_spatial,
                    float *Xout /* flat [60*C*B] */, float *out_spatial /* [B,C,Lout] */,
                    const int32_t *g_map /* optional [60]: group column -> compact output column (< Lvalid) or -1 */,
                    int Lout /* row pitch of the compact output, >= Lvalid; pad columns are zero */, int Lvalid,
                    int B, int C, int split /* 0: f32-input MFMA; 1: 3 x bf16 split MFMAs; 2: fp16 x 2 with per-column (per-keypoint) power-of-two scales */,
                    const float *out_bound /* split = 2 with Xout: per-keypoint bound [round_up(B,32)] on |coefficient|; Xout then holds the fp16 hi/lo
                                              words roreg_irrep_gemm_f16x2 consumes (NOT floats) */,
                    float *out_rowmax /* optional, with out_spatial: [B], zeroed by the caller; receives max |out_spatial[b]| per keypoint (the block
                                         scale of roreg_group_conv_f16x2) */,
                    int spatial_bf16 /* x_spatial / resid_spatial point to bfloat16 tensors (BASELINE config 5: group features stored as bf16) */,
                    int out_planes /* split = 2 with Xout: write Xout in the HALF-BLOCK layout roreg_irrep_gemm_f16x2(x_planes = 1) consumes */,
                    void *stream);
/* v5: the inverse transform + bias + BatchNorm + ReLU of roreg_ft_nonlin with the group-domain result [B,C,Lout] written as WORDS fp16 hi | fp16 lo << 16
 * of value * 2^e_b, e_b = the block exponent of out_bound[b] (a bound on |ReLU(BN(.))| of row b that exists before the tensor does: the producing
 * GEMM's propagated bound with u_o = sqrt(60) |scale_o|, v_o = |scale_o| |bias_o| + |shift_o|) -- the operand of roreg_group_conv_f16x2_packed.
 * raw_col (nullable, [B,C] float32): the value BEFORE BatchNorm / ReLU at group element raw_g (ET's identity short cut, eqv_trans.py:112-117). */
int roreg_ft_nonlin_packed(const float *Xin, const float *bias, const float *bn_scale, const float *bn_shift, uint32_t *out_words,
                           const int32_t *g_map, int Lout, int Lvalid, int B, int C, const float *out_bound, float *raw_col, int raw_g,
                           void *stream);

/* Optional kernel timing for bench.py's measured rooflines (no reference counterpart: the reference has no profiler hooks).
 * While enabled, the library brackets selected launches with HIP events recorded ON THE LAUNCH STREAM; roreg_profile_read synchronises
 * on them and returns the summed duration and the number of brackets of a slot:
 *   0 = the two mm_tile_kernel passes of roreg_mutual_match_batch (the descriptor distance matrix on the matrix cores),
 *   1 = ransac_score_batch_kernel of roreg_ransac_batch, 2 = des2r_batch_kernel of roreg_lt_prepare_batch, 3 = roreg_ft_nonlin,
 *   4 = the `iters` Sinkhorn iterations of roreg_sinkhorn_batch (one fused pass over every pair's coupling matrix + column merge each),
 *   5 = roreg_topk_dot (slice search + merge).
 * roreg_profile_enable(1) clears earlier records; (0) stops recording. */
int roreg_profile_enable(int on);
int roreg_profile_read(int slot, double *total_ms, int *launches);

#ifdef __cplusplus
}
#endif
#endif /* ROREG_HIP_H */
