/* ST-voxel aggregation entry points (HR_Block / CR_Block of latte/models/hr_module.py); included by latte_hip.h.
 *
 * The reference loops in Python over the B anchors of a batch and, per anchor, goes host<->device six times
 * (hr_module.py:188-299).  Here all anchors of a batch are one batched pipeline: every (anchor, window frame)
 * pair is a "segment" of one flat element array, the anchor index sits in the top bits of the 64-bit voxel key,
 * so ONE radix sort orders every voxel of every anchor and the per-voxel reductions are segmented reductions
 * over sorted keys (no atomics; sums run in ascending point order like the reference's CPU index_add). */
#ifndef LATTE_HIP_ST_H
#define LATTE_HIP_ST_H

/* F.softmax(logits, 1) and sum_c prob_2_entropy (hr_module.py:236-237,251; latte/models/losses.py:10-19):
 * probs[p,:] = softmax(logits[p,:]); ety[p] = sum_c -p*log2(p+1e-30)/log2(C).  C <= 64. */
int lt_softmax_entropy(const float* logits, int64_t P, int32_t C, float* probs, float* ety, lt_stream_t stream);
/* backward of probs = softmax(logits): glogits = p * (g - sum_c g*p) */
int lt_softmax_bwd(const float* probs, const float* gprobs, int64_t P, int32_t C, float* glogits, lt_stream_t stream);

/* Pose transform (fp64) + range mask + fixed-origin voxel key for a flat element array (hr_module.py:210-233,
 * sparse_quantize :73-97, ravel_hash :58-71 incl. the in-place shift Q1).
 * Segment s covers elements [seg_out_start[s], seg_out_start[s+1]) = points[seg_src_start[s] ...] of one frame,
 * belongs to anchor seg_anchor[s] and is transformed by the row-major 3x4 fp64 matrix seg_T + 12*s
 * (seg_T == NULL: no transform, the fp32 range test of the anchor branch :227 is used).
 * key = anchor<<3b | x<<2b | y<<b | z (dense: b = coord_bits in [1,16], 2*max_range/voxel_size + 4 < 2^b) with
 * (x,y,z) = floor(p/voxel_size) - min_offset, or INT64_MAX when the point is outside max_range.  Dense keys are what
 * makes the sort cheap: anchors x 3b bits instead of 63 (lt_sort_segments' key_bits).  src[e] = index of the source
 * point.  All seg_* arrays are DEVICE arrays. */
int lt_st_keys(const float* points, const int64_t* seg_out_start, const int64_t* seg_src_start,
               const int32_t* seg_anchor, const double* seg_T, int32_t S, int64_t E, double max_range,
               double voxel_size, int32_t min_offset, int32_t coord_bits, int64_t* keys, int32_t* src,
               lt_stream_t stream);

/* np.unique(keys, return_index, return_inverse) as a sort: one stable radix sort of (key, element) pairs, then
 * segment heads.  Keys equal to INT64_MAX are "masked" and belong to no voxel.  key_bits: every unmasked key is
 * < 2^key_bits (the sort then runs over key_bits + 1 bits); <= 0 or >= 62: all 63 bits.
 * Outputs (device): sorted_elem[E]; inverse[E] (voxel id or -1); uniq_keys[cap E]; seg_start[cap E+1]
 * (positions in sorted order, seg_start[V] = number of unmasked elements).  *V (HOST).  [sync] */
int64_t lt_sort_segments_workspace_bytes(int64_t E);
int lt_sort_segments(const int64_t* keys, int64_t E, int32_t key_bits, void* ws, int64_t ws_bytes,
                     int32_t* sorted_elem, int32_t* inverse, int64_t* uniq_keys, int32_t* seg_start, int64_t* V,
                     lt_stream_t stream);

/* torch_scatter.scatter(src, index, dim=0, reduce) over sorted segments (hr_module.py:236,237,252):
 * out[v,:] = reduce_{p in [seg_start[v], seg_start[v+1])} src[row(sorted_elem[p]), :], row(e) = src_index ? src_index[e] : e.
 * mode 0 sum, 1 mean, 2 max (argmax[v,c] = element index, optional). */
int lt_segment_reduce(const float* src, int32_t C, const int32_t* src_index, const int32_t* sorted_elem,
                      const int32_t* seg_start, int64_t V, int32_t mode, float* out, int32_t* argmax,
                      lt_stream_t stream);
/* backward of mode 0/1 w.r.t. the ELEMENTS: gelem[e,:] = gout[inverse[e],:] / (mean ? count : 1), 0 when masked */
int lt_segment_reduce_bwd(const float* gout, int32_t C, const int32_t* inverse, const int32_t* seg_start, int64_t E,
                          int32_t mean, float* gelem, lt_stream_t stream);

/* ST-voxel entropy written back to the points (hr_module.py:256-259,305-306):
 * out[j] = mean over the anchors whose window holds point j's frame of voxel_ety[inverse[element of (anchor, j)]],
 * 0 when no anchor sees the point in range.  frame_start[F+1]: point offsets of the hist frames; frame_seg[F*Wmax]:
 * segments that contain frame f in ascending anchor order (-1 padded); seg_out_start as in lt_st_keys. */
int lt_st_gather_back(const float* voxel_ety, const int32_t* inverse, const int64_t* frame_start, int32_t F,
                      const int32_t* frame_seg, int32_t Wmax, const int64_t* seg_out_start, int64_t P, float* out,
                      lt_stream_t stream);

/* The same when the frames of a window are NOT contiguous in the batch (a shuffled loader).  The reference then takes the
 * hist rows sequentially from the window's first frame on (hr_module.py:203-220: init_hr_left = rows before
 * min(corr_indices), then one slice per frame of that frame's size) and writes the entropies back to that row range
 * (:256-259) -- SURVEY quirk Q5, reproduced as it is: segment s covers rows [seg_row_start[s], + its length).
 * out[j] = mean over the segments that cover row j and see the element in range, in segment (= anchor) order. */
int lt_st_gather_back_rows(const float* voxel_ety, const int32_t* inverse, const int64_t* seg_row_start /*[S]*/,
                           const int64_t* seg_out_start /*[S+1]*/, int32_t S, int64_t P, float* out, lt_stream_t stream);

/* Voxel-set intersection of anchor and hist voxels (replaces the dense [Va,Vt] compare of hr_module.py:271-283):
 * both key arrays are sorted ascending.  a_sel / t_sel (capacity min(Va,Vt)) receive the matched voxel ids in
 * ascending order (k-th entries correspond); *M (HOST) = number of matches.  [sync] */
int64_t lt_st_join_workspace_bytes(int64_t Va, int64_t Vt);
int lt_st_join(const int64_t* a_keys, int64_t Va, const int64_t* t_keys, int64_t Vt, void* ws, int64_t ws_bytes,
               int32_t* a_sel, int32_t* t_sel, int64_t* M, lt_stream_t stream);

/* refine_ety (latte/data/utils/refine_pseudo_labels.py:33-53): per class c of label[0..n), thresh_c =
 * torch.quantile(ety[label==c], q) (linear interpolation, fp32); mask[j] = !(ety[j] > thresh_label[j]).
 * ety must be >= 0.  mask has n bytes.  Up to 16 classes the thresholds come from a radix SELECTION (four 8-bit histogram
 * passes over order-preserving keys find the two order statistics exactly; no sort), above that from a radix sort of
 * (class, value) keys: the same thresholds either way.  No sync. */
int64_t lt_class_quantile_workspace_bytes(int64_t n);
int lt_class_quantile_mask(const float* ety, const int32_t* label, int64_t n, int32_t num_classes, float q, void* ws,
                           int64_t ws_bytes, uint8_t* mask, lt_stream_t stream);

/* refine_pseudo_labels (latte/data/utils/refine_pseudo_labels.py:10-30, called at latte/tta/tools/latte.py:238): per class c
 * of label[0..n), thresh_c = min(torch.median(vals[label==c]), cap) (lower middle element); keep[j] = !(vals[j] <
 * thresh_label[j]) -- the caller sets label to the ignore value where keep is 0.  vals must be >= 0.  Same workspace as
 * lt_class_quantile_mask. */
int lt_class_median_mask(const float* vals, const int32_t* label, int64_t n, int32_t num_classes, float cap, void* ws,
                         int64_t ws_bytes, uint8_t* keep, lt_stream_t stream);

/* Entropy-weighted pseudo-label fusion of the two teachers (latte/tta/tools/latte.py:207-236; losses.py:10-19) as two
 * kernels: logits_2d [Nf,C] of the in-FOV points, logits_3d [N,C] of all points, fov_idx [Nf] (unique rows of N),
 * st_ety_* the ST-voxel entropies of HR_Block (0 = no ST voxel), thr_* device scalars = the conf_perc quantile of st_ety_*
 * (NULL: no quantile filter).  Out: probs_3d [N,C] = softmax(logits_3d), pc_2d [Nf] / pc_3d [N] = point entropies with the
 * ST entropy substituted where it is valid, ps_conf [N] / ps_label [N] = max / argmax of the fused probabilities (2-D and
 * 3-D blended by exp(-entropy) weights inside the FOV, 3-D alone outside).  C <= 64. */
int lt_pseudo_label_fuse(const float* logits_2d, const float* logits_3d, const int64_t* fov_idx, const float* st_ety_2d,
                         const float* st_ety_3d, const float* thr_2d, const float* thr_3d, int64_t N, int64_t Nf, int32_t C,
                         float* probs_3d, float* pc_2d, float* pc_3d, float* ps_conf, int32_t* ps_label,
                         lt_stream_t stream);

/* torch.quantile(x, q) (linear interpolation) of a 1-D float array, result left on the device: out[0].  The thresholds
 * of latte/tta/tools/latte.py:214-215.  The two order statistics by radix selection (four 8-bit histogram passes, no
 * sort; torch sorts (key, index) pairs); rank and interpolation arithmetic as in ATen (fp32), so the result is torch's bit
 * for bit on finite inputs (-0 counts as +0).  No sync. */
int64_t lt_quantile_workspace_bytes(int64_t n);
int lt_quantile(const float* x, int64_t n, float q, void* ws, int64_t ws_bytes, float* out, lt_stream_t stream);

/* CR_Block.forward (latte/models/hr_module.py:344-420): entropy-weighted KL between matched anchor / hist ST-voxel
 * distributions.  ac, tr [M,C]: matched voxel PROBABILITIES of all anchors, rows sorted by anchor (anchor_start[B+1], device;
 * the matched lists are already aligned: the reference's two argsorts :359-360 are the identity, Q3).
 *   w[m] = exp(-t_e)/(exp(-t_e)+exp(-a_e)), a_e = ac_pc_ety[ac_p2v[m]], t_e = tr_pc_ety[tr_p2v[m]] (:373-377; the four
 *          pointers NULL together: w = 1, the ety_weight=False / self_corr branch :379-380)
 *   row_loss[m] = w[m] * sum_c tr~ (log tr~ - log ac~), x~ = x == 0 ? 1e-8 : x   (the clamp on probabilities, :405-410)
 *   loss[0] = 1/B sum_b mean_{m in anchor b, keep[m]} row_loss[m]                (keep = tr_conf_mask :367-371; NULL: all)
 * Outputs kept for the backward: row_w [M], anchor_stats [B][2] = (sum, count) of the kept rows.  Three launches, fixed
 * summation order, no sync.  C <= 64. */
int lt_cr_loss_fwd(const float* ac, const float* tr, int64_t M, int32_t C, const float* ac_pc_ety, const int64_t* ac_p2v,
                   const float* tr_pc_ety, const int64_t* tr_p2v, const uint8_t* keep, const int32_t* anchor_start,
                   int32_t B, float* row_w, float* row_loss, float* anchor_stats, float* loss, lt_stream_t stream);
/* gradient of loss[0] w.r.t. ac (tr is detached in the reference, :409): g_ac[m,c] = -gloss[0]/B * keep[m] * w[m] /
 * count_b * tr~[m,c] / ac[m,c], 0 where ac == 0 (the clamp replaced the element by a constant).  row_anchor [M] int64:
 * anchor of every row; gloss: device scalar. */
int lt_cr_loss_bwd(const float* ac, const float* tr, int64_t M, int32_t C, const float* row_w, const uint8_t* keep,
                   const int64_t* row_anchor, const float* anchor_stats, int32_t B, const float* gloss, float* g_ac,
                   lt_stream_t stream);

#endif
