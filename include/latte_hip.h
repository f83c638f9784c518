/* liblatte_hip.so -- C-ABI of the MI355X (gfx950) sparse-voxel hot path of Latte.
 *
 * The reference (AronCao49/Latte) has no FFI layer: the boundary it consumes is the Python import
 * surface of two third-party wheels (torchsparse v1.4.0, torch_scatter) plus its own
 * latte/models/hr_module.py.  Every entry point below names the reference interface it replaces
 * (file:line under /root/reference).  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *  - plain device pointers + element counts + hipStream_t; no torch types
 *  - return 0 on success, <0 on error (LT_ERR_*); text via lt_last_error() (thread-local)
 *  - the library never allocates outputs: the caller owns every buffer.  Data-dependent output sizes:
 *    the caller passes buffers of the documented maximum capacity and the call returns the count
 *    through a HOST pointer (those calls synchronise `stream`; they are marked [sync])
 *  - workspaces are opaque byte buffers sized by the matching *_workspace_bytes() query
 *  - feature matrices are row-major fp32 [rows, channels]; coordinates are int32 [rows,4] = (x,y,z,batch)
 *  - threads and devices: every entry point is re-entrant -- it keeps no state between calls beyond what the caller passes
 *    (buffers, workspaces, plans) and launches only on the given stream, on that stream's device -- so calls from
 *    several threads and on several devices of one process are independent as long as they do not share output or
 *    workspace buffers (latte_amd's prefetch thread builds index structures on a side stream while the main thread
 *    launches the convs).  The exceptions are the PROCESS-WIDE switches and diagnostics, to be called while no other
 *    thread is inside the library: lt_conv_set_precision, lt_conv_timer_begin / _arm / _end (conv launches from any
 *    thread are recorded while armed; slot hand-out is locked), lt_conv_debug_stamps; and lt_bn_stats_static, whose
 *    calls share one scratch array per GPU (see there).  lt_last_error() is thread-local.
 */
#ifndef LATTE_HIP_H
#define LATTE_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* lt_stream_t; /* == hipStream_t */

const char* lt_last_error(void);
int lt_version(void);

/* ------------------------------------------------------------------ hashing / tables
 * torchsparse.nn.functional.sphash(coords[, offsets])  (latte/models/torchsparse_utils.py:15,40,45,68,73)
 * out[k*n+i] = fold60(FNV1a64(x+ox_k, y+oy_k, z+oz_k, b)); K==0 -> out[i] without offsets. */
int lt_sphash(const int32_t* coords, int64_t n, const int32_t* offsets, int32_t K, int64_t* out, lt_stream_t stream);

/* torchsparse.nn.functional.sphashquery(queries, references)  (torchsparse_utils.py:17,46,74)
 * build: table over `keys` (value = row index; duplicate keys keep the smallest index)
 * query: out[i] = index of queries[i] in keys, -1 on miss. `n_build` must be the n given to build.
 * RESERVED KEY: int64 -1 (all bits set) marks an empty slot of the table; a key with that value is never found and, in
 * lt_unique_sorted, never reported (sphash folds to 60 bits and the voxel keys are non-negative: neither produces it). */
int64_t lt_table_bytes(int64_t n);
int lt_hash_build(const int64_t* keys, int64_t n, void* table, int64_t table_bytes, lt_stream_t stream);
int lt_hash_query(const int64_t* queries, int64_t q, const void* table, int64_t n_build, int64_t* out,
                  lt_stream_t stream);

/* torchsparse.nn.functional.spcount(idx, num)  (torchsparse_utils.py:18,47): histogram ignoring idx<0 */
int lt_spcount(const int32_t* idx, int64_t n, int32_t* out, int64_t num, lt_stream_t stream);

/* np.unique(keys, return_index, return_inverse) / torch.unique(sorted) on non-negative int64 keys:
 * (hr_module.py:87-89; torchsparse.utils.quantize.sparse_quantize at semantic_kitti_dataloader.py:562;
 *  torch.unique(pc_hash) at torchsparse_utils.py:16).
 * out_keys/out_first have capacity n, out_inverse n (any may be NULL). key_bits = number of significant
 * low bits (<=63) -> fewer radix passes.  *count (HOST) receives V.  [sync] */
int64_t lt_unique_workspace_bytes(int64_t n);
int lt_unique_sorted(const int64_t* keys, int64_t n, int32_t key_bits, void* ws, int64_t ws_bytes, int64_t* out_keys,
                     int64_t* out_first, int64_t* out_inverse, int64_t* count, lt_stream_t stream);

/* ------------------------------------------------------------------ voxel keys
 * ravel keys of the dataloader quantize contract a1: coords = round(p*scale) - min is done by the caller
 * (torch), this packs non-negative int coords (<2^20 each) into a lexicographic (x,y,z) key.
 * (torchsparse ravel_hash as copied in hr_module.py:58-71: ORDER is lexicographic in (x,y,z)). */
int lt_pack_xyz_keys(const int32_t* coords3, int64_t n, int64_t* keys, lt_stream_t stream);

/* ------------------------------------------------------------------ kernel maps (spnn.Conv3d internals)
 * F.spdownsample(coords, stride=2, kernel=2, tensor_stride): coords//s*s, unique rows sorted by (b,x,y,z)
 * (used by spnn.Conv3d(stride=2): latte/models/basic_blocks.py:17 via spvcnn.py:36,42,48,54).
 * out_coords capacity = V rows. [sync] */
int64_t lt_downsample_workspace_bytes(int64_t V);
int lt_downsample_coords(const int32_t* coords, int64_t V, int32_t sample_stride, void* ws, int64_t ws_bytes,
                         int32_t* out_coords, int64_t* count, lt_stream_t stream);

/* Neighbour table of one conv: results[k*Vout+o] = row of the input voxel at coords_out[o]+offsets[k], or -1.
 * (= torchsparse `results = sphashquery(sphash(coords_out, offsets), sphash(coords_in))` inside conv3d;
 * reference call sites basic_blocks.py:17,34,51,58 and spvcnn.py:30,32). No sync. */
int64_t lt_kmap_workspace_bytes(int64_t Vin);
int lt_kmap_build(const int32_t* coords_in, int64_t Vin, const int32_t* coords_out, int64_t Vout,
                  const int32_t* offsets, int32_t K, void* ws, int64_t ws_bytes, int32_t* results,
                  lt_stream_t stream);
/* The same table for a submanifold conv (stride 1, odd kernel: basic_blocks.py:17,51,58 with stride 1, spvcnn.py:30,32):
 * coords_in == coords_out == `coords` with UNIQUE rows, K odd and offsets[K-1-k] == -offsets[k] (torchsparse's
 * get_kernel_offsets order has that property for odd kernels).  Then results[K-1-k, i] == o <=> results[k, o] == i, and
 * K/2 + 1 probes per row fill the whole table (the other K/2 rows receive the mirrored hits).  Same bits as
 * lt_kmap_build under those preconditions, which the caller guarantees (they are not checked on the device). No sync. */
int lt_kmap_build_sub(const int32_t* coords, int64_t V, const int32_t* offsets, int32_t K, void* ws, int64_t ws_bytes,
                      int32_t* results, lt_stream_t stream);
/* torchsparse (nbmaps [M,2]=(in,out) k-major/out-ascending, nbsizes[K]) derived from `results`
 * (nonzero(results != -1) order).  nbmaps capacity K*Vout rows.  [sync] */
int64_t lt_kmap_pairs_workspace_bytes(int64_t K, int64_t Vout);
int lt_kmap_pairs(const int32_t* results, int32_t K, int64_t Vout, void* ws, int64_t ws_bytes, int32_t* nbmaps,
                  int32_t* nbsizes, int64_t* M, lt_stream_t stream);
/* reverse table for transposed conv / backward: rev[k*Vin+i] = o where results[k*Vout+o] == i, else -1 */
int lt_kmap_reverse(const int32_t* results, int32_t K, int64_t Vout, int64_t Vin, int32_t* rev, lt_stream_t stream);

/* Processing order for lt_spconv_fwd over a neighbour table [K,V] (results or rev): the rows sorted (stable) by their
 * K-bit "which offsets have a neighbour" mask, so that the 16-row MFMA strips of the conv kernel hold rows with the
 * same offsets.  No counterpart in torchsparse (it gathers per offset); results do not depend on the order. No sync. */
int64_t lt_kmap_order_workspace_bytes(int64_t V);
int lt_kmap_order(const int32_t* table, int32_t K, int64_t V, void* ws, int64_t ws_bytes, int32_t* order,
                  lt_stream_t stream);

/* Conv plan of a neighbour table [K,V]: the mask-sorted row order plus the table re-laid out per 64-row tile and one
 * 64-bit row mask per (tile, offset) -- what lt_spconv_plan_fwd consumes.  Built once per kernel map, shared by every
 * conv (forward and backward) that uses the map.  `plan` is a caller-owned buffer of lt_kmap_plan_bytes(K,V) bytes,
 * 16-byte aligned.  No counterpart in torchsparse.  No sync. */
int64_t lt_kmap_plan_bytes(int32_t K, int64_t V);
int64_t lt_kmap_plan_workspace_bytes(int64_t V);
int lt_kmap_plan(const int32_t* table, int32_t K, int64_t V, void* ws, int64_t ws_bytes, void* plan, int64_t plan_bytes,
                 lt_stream_t stream);

/* ------------------------------------------------------------------ sparse convolution
 * spnn.Conv3d forward / backward-input (basic_blocks.py:17,34,51,58,68; spvcnn.py:30,32):
 *   out[o,:] = sum_k in[nbr[k,o],:] @ Wk      (rows with nbr<0 contribute nothing)
 * fp32 in / fp32 accumulate on v_mfma_f32_16x16x4_f32, output-stationary (no scatter, no atomics).
 * Weights must first be packed into MFMA-fragment order:
 *   mode 0: Wk            (forward, weight [K,Cin,Cout])
 *   mode 1: Wk^T          (transposed conv fwd's backward / strided conv backward: use with lt_kmap_reverse)
 *   mode 2: W(K-1-k)^T    (backward-input of a submanifold stride-1 conv on the SAME table)
 *   mode 4 / 5 / 6: the bf16 hi|lo split images of modes 0 / 1 / 2 for the bf16x3 kernel (lt_conv_set_precision)
 * For modes 1,2 the packed matrix maps Cout -> Cin channels (swap Cin/Cout in lt_spconv_fwd).
 * Cin % 4 == 0 and Cout % 16 == 0 required; `order` (optional, [Vout]) is a processing permutation of the
 * output rows (locality / skip rate), results do not depend on it. */
int64_t lt_spconv_packed_floats(int32_t K, int32_t Cin, int32_t Cout);
int lt_spconv_pack_weights(const float* W, int32_t K, int32_t Cin, int32_t Cout, int32_t mode, float* packed,
                           lt_stream_t stream);
int lt_spconv_fwd(const float* in, const float* packed, const int32_t* nbr, const int32_t* order, float* out,
                  int64_t Vin, int64_t Vout, int32_t Cin, int32_t Cout, int32_t K, lt_stream_t stream);
/* Same product as lt_spconv_fwd through a plan (lt_kmap_plan) of the [K,Vout] table: only the offsets active in a
 * 64-row tile are visited, weights stream global->LDS by LDS-DMA one stage ahead of the MFMAs.  Cin % 16 == 0 and
 * Cout % 16 == 0.  Bitwise equal to lt_spconv_fwd. */
int lt_spconv_plan_fwd(const float* in, const float* packed, const void* plan, float* out, int64_t Vin, int64_t Vout,
                       int32_t Cin, int32_t Cout, int32_t K, lt_stream_t stream);
/* the same with an explicit kernel variant (tuning and tests): 0 = default (split-k shape chosen from Vout, Cout),
 * 1 = LDS-DMA staged kernel, 100*SR + NTW = register-fed kernel, one k-ascending chain per output element (bitwise equal
 * to lt_spconv_fwd), SR (1|2|4|8) 16-row strips per workgroup, NTW (1|2|4) column tiles per wave;
 * 1000 + 100*SR + 10*NTW + W = split-k: W (2|4|8) waves share the slab list of SR (1|2|4) strips x NTW (1|2|3|4|6|8)
 * column tiles and add their partial sums in wave order (deterministic; differs from the single chain by fp32 rounding);
 * 10000 + a split-k code = the same without the XCD-aware workgroup mapping (A/B measurements) */
int lt_spconv_plan_fwd_v(const float* in, const float* packed, const void* plan, float* out, int64_t Vin, int64_t Vout,
                         int32_t Cin, int32_t Cout, int32_t K, int32_t variant, lt_stream_t stream);
/* lt_spconv_plan_fwd that also leaves, in stats_ws (lt_spconv_stats_bytes), the column sums of the rows each workgroup b
 * produced: partial[(b*2+0)*Cout + c] = sum out[:,c], partial[(b*2+1)*Cout + c] = sum out[:,c]^2 at the start of
 * stats_ws, b < lt_spconv_stat_blocks(Vout,Cout) -- the BatchNorm statistics of the following spnn.BatchNorm
 * (basic_blocks.py:22,40,56) without re-reading the output; lt_bn_finalize_sums turns them into mean / variance. */
int64_t lt_spconv_stat_blocks(int64_t Vout, int32_t Cout);
int64_t lt_spconv_stats_bytes(int64_t Vout, int32_t Cout);
int lt_spconv_plan_fwd_stats(const float* in, const float* packed, const void* plan, float* out, int64_t Vin, int64_t Vout,
                             int32_t Cin, int32_t Cout, int32_t K, void* stats_ws, lt_stream_t stream);

/* The network's first conv (latte/models/spvcnn.py:29-31: spnn.Conv3d(4, cs[0], 3)): 4 input channels -> 16 | 32, a
 * streaming kernel (one lane per output row) over a plan of the table; weight is the UNPACKED kernel [K, 4, Cout]. */
int lt_spconv_stem_fwd(const float* in, const float* weight, const void* plan, float* out, int64_t Vin, int64_t Vout,
                       int32_t Cin, int32_t Cout, int32_t K, lt_stream_t stream);
/* spnn.Conv3d backward-weight (torchsparse conv3d backward: grad_weight[k] = in[nbmaps_k[:,0]]^T @ gout[nbmaps_k[:,1]]; reached
 * with TRAIN.UPDATE_PARAMS == "all": latte/common/utils/ema_util.py:44-71, latte/tta/tools/latte.py:324-326).
 * nbmaps / nbsizes: the pair list of the kernel map (lt_kmap_pairs), M = its length; transposed != 0: the conv ran the map
 * backwards (in rows = nbmaps[:,1], gout rows = nbmaps[:,0]).  dW [K,Cin,Cout] fp32.  Cin, Cout multiples of 16.  Pair
 * slices are reduced in a fixed order: bitwise reproducible. */
int64_t lt_spconv_bwd_weight_workspace_bytes(int64_t M, int32_t K, int32_t Cin, int32_t Cout);
int lt_spconv_bwd_weight(const float* in, const float* gout, const int32_t* nbmaps, const int32_t* nbsizes, int64_t M,
                         int64_t Vin, int64_t Vout, int32_t Cin, int32_t Cout, int32_t K, int32_t transposed, void* ws,
                         int64_t ws_bytes, float* dW, lt_stream_t stream);

/* Arithmetic of the planned conv kernels, process-wide (north_star: "bf16/fp32 MFMA tiles"; no counterpart in the reference,
 * whose convs are fp32 GEMMs).  0 (default): fp32 MFMA, one reproducible fmaf chain per output.  1: bf16x3 split precision
 * on v_mfma_f32_16x16x32_bf16 -- every fp32 operand is hi + lo in bf16, a product is hi*hi + hi*lo + lo*hi accumulated in
 * fp32: ~1e-5 relative to the fp32 kernel (inside the 1e-3 bar, not bit-equal), 1.1-1.5x faster per conv.  It applies to
 * the convs for which lt_conv_uses_bf16(Cin, Cout) is 1 (effective Cin >= 64 and a multiple of 32); THOSE convs' weights
 * must be packed with lt_spconv_pack_weights mode 4 / 5 / 6 (the bf16 hi|lo images of modes 0 / 1 / 2).  All others keep
 * fp32.  Explicit variants: lt_spconv_plan_fwd_v 3000 + 100*SR + 10*NTW + W (W in 1..4 divides Cin/32). */
int lt_conv_set_precision(int32_t mode);
int lt_conv_uses_bf16(int32_t Cin, int32_t Cout);

/* Conv launch timer -- measurement aid, no counterpart in the reference (bench.py's roofline leg uses it).  Between
 * begin and end every sparse-conv kernel launch of this library (lt_spconv_fwd, lt_spconv_plan_fwd*, and the ones inside
 * lt_conv_bn_fwd/bwd) is bracketed by two hipEvents on the launch stream.  lt_conv_timer_end [sync] waits for them and
 * returns the number of launches n (<= capacity), ms[i] = duration, meta[4*i..] = Cin, Cout, K, Vout of launch i. */
int lt_conv_timer_begin(int32_t capacity);
int lt_conv_timer_arm(int32_t on); /* pause (0) / resume (1) between begin and end: time a sample of the launches */
int64_t lt_conv_timer_end(float* ms, int32_t* meta, int64_t capacity);
/* Diagnostics, no counterpart in the reference (tools/conv_stamps.py): device buffer of cap_words int64 that the
 * variants 70000 + split-k code of lt_spconv_plan_fwd_v fill with 8 words per wave (s_memtime at entry / after the
 * prologue / after the main loop / at the end, s_memrealtime at entry / end, slabs | active offsets << 32,
 * HW_ID | XCC_ID << 32).  Variants 40000/50000/60000 + code read L1-resident weights / rows / both (wrong results on
 * purpose: they price the memory path).  NULL switches the stamps off. */
int lt_conv_debug_stamps(void* buf, int64_t cap_words);

/* ------------------------------------------------------------------ BatchNorm(batch stats)+ReLU+residual
 * spnn.BatchNorm + spnn.ReLU + skip add (basic_blocks.py:22-23,56-57,64,69,72,75); TTA forces batch
 * statistics (latte/common/utils/ema_util.py:26-29).  var is the biased variance. */
int64_t lt_bn_workspace_bytes(int64_t V, int32_t C);
int lt_bn_stats(const float* x, int64_t V, int32_t C, void* ws, int64_t ws_bytes, float* mean, float* var,
                lt_stream_t stream);
/* lt_bn_stats for callers that cannot allocate (HIP graph capture, serving loops, harnesses that forbid hipMalloc): NO
 * counterpart in the reference.  Same kernels and results as lt_bn_stats, but the per-block partial sums live in a
 * file-scope device array of the library (2 MiB per GPU, held until the process ends) instead of a caller workspace;
 * the call allocates nothing.  Limit: C <= 256 (LT_ERR_RANGE beyond it).  The array used is the one of the GPU that
 * `stream` lives on (hipStreamGetDevice; the null stream: the current device), whatever device is current in the calling
 * thread: the first call for a device looks its address up with that device current, later calls reuse it.  All calls on
 * one GPU share the array: calls that overlap on one GPU (different streams or threads) race -- serialise them. */
int lt_bn_stats_static(const float* x, int64_t V, int32_t C, float* mean, float* var, lt_stream_t stream);
/* mean / biased variance from the column sums lt_spconv_plan_fwd_stats left in stats_ws (V = Vout, C = Cout of that call) */
int lt_bn_finalize_sums(void* stats_ws, int64_t V, int32_t C, float* mean, float* var, lt_stream_t stream);
/* relu: 0 = y = bn(x) + residual;  1 = y = relu(bn(x) + residual)  (ResidualBlock, basic_blocks.py:72-75);
 * 3 = y = relu(bn(x)) + residual: the sum after the activation -- the point branch of SPVCNN,
 * z.F + point_transforms(z.F) (latte/models/spvcnn.py:95-111 and its forward).  residual may be NULL (relu 0 / 1). */
int lt_bn_apply(const float* x, const float* mean, const float* var, const float* gamma, const float* beta,
                float eps, const float* residual, int32_t relu, float* y, int64_t V, int32_t C,
                lt_stream_t stream);
/* The point branch with the devoxelized addend formed on the fly (latte/models/spvcnn.py:141-142,152-153,163-164:
 * z.F = voxel_to_point(x, z).F + point_transforms(z.F)):
 *   y[p,:] = relu(bn(x[p,:])) + sum_{k<8} w[p,k] * vox[idx[p,k],:]       (idx < 0: corner skipped)
 * = lt_devoxelize_fwd followed by lt_bn_apply(relu = 3), bit for bit, without writing and re-reading the [N,C]
 * devoxelized tensor.  idx [N,8] int32, w [N,8] (lt_calc_ti_weights, transposed), vox [V,C]. */
int lt_bn_apply_devox(const float* x, const float* mean, const float* var, const float* gamma, const float* beta, float eps,
                      const float* vox, const int32_t* idx, const float* w, float* y, int64_t N, int32_t C, int64_t V,
                      lt_stream_t stream);
/* g = dy * mask;  sum_g[c] = sum g;  sum_gx[c] = sum g*xhat.  relu: 0 = no ReLU (mask = 1); 1 = mask = (y > 0), y read;
 * 2 = the forward added no residual: mask = (gamma*xhat + beta > 0) rebuilt from x with the forward's own expression
 * (bit-identical to y > 0), y is not read (may be NULL), gamma / beta are needed (NULL allowed for relu 0 / 1). */
int lt_bn_bwd_reduce(const float* dy, const float* x, const float* y, const float* mean, const float* var,
                     const float* gamma, const float* beta, float eps, int32_t relu, int64_t V, int32_t C, void* ws,
                     int64_t ws_bytes, float* sum_g, float* sum_gx, lt_stream_t stream);
/* dx = gamma*rstd*(g - sum_g/V - xhat*sum_gx/V); dres = g (optional); relu and beta as in lt_bn_bwd_reduce */
int lt_bn_bwd_apply(const float* dy, const float* x, const float* y, const float* mean, const float* var,
                    const float* gamma, const float* beta, float eps, int32_t relu, const float* sum_g,
                    const float* sum_gx, float* dx, float* dres, int64_t V, int32_t C, lt_stream_t stream);

/* ------------------------------------------------------------------ conv -> BN [+ residual] [+ ReLU] in one call
 * The block pattern of latte/models/basic_blocks.py:13-76 (Conv3d, BatchNorm, ReLU / residual add) as ONE host call:
 * conv with statistics epilogue -> finalize -> apply.  conv_out [Vout,Cout] is kept (the backward needs it), y is the
 * block output.  Backward: BN backward (dgamma, dbeta, optional dres = gradient of the residual input), then the
 * gradient w.r.t. the conv input through packed_bwd (mode 1 or 2) and the plan of the backward table; din may be NULL
 * (then packed_bwd / plan_bwd are not needed), dconv is a [Vout,Cout] scratch/output buffer. */
int64_t lt_conv_bn_workspace_bytes(int64_t Vout, int32_t Cout);
int lt_conv_bn_fwd(const float* in, const float* packed, const void* plan, int64_t Vin, int64_t Vout, int32_t Cin,
                   int32_t Cout, int32_t K, const float* gamma, const float* beta, float eps, const float* residual,
                   int32_t relu, void* ws, int64_t ws_bytes, float* conv_out, float* y, float* mean, float* var,
                   lt_stream_t stream);
int lt_conv_bn_bwd(const float* dy, const float* conv_out, const float* y, const float* mean, const float* var,
                   const float* gamma, const float* beta, float eps, int32_t relu, int64_t Vout, int32_t Cout, void* ws,
                   int64_t ws_bytes,
                   float* dgamma, float* dbeta, float* dres, float* dconv, const float* packed_bwd, const void* plan_bwd,
                   int64_t Vin, int32_t Cin, int32_t K, float* din, lt_stream_t stream);

/* ------------------------------------------------------------------ heads + inverse_to_all
 * Net3DSeg's linear heads (latte/models/xmuda_arch.py:326-335) fused with inverse_to_all (latte/data/collate.py:34-55):
 * out1[p,:] = feats[idx[p],:] @ E[:, 0:C1] + bias[0:C1], out2[p,:] = feats[idx[p],:] @ E[:, C1:C1+C2] + bias[C1:C1+C2]
 * for the N raw points (idx int64 [N] into the V feature rows, NULL = identity; rows outside [0,V) read as zeros).
 * packed = lt_spconv_pack_weights(E, K=1, Cin, 16*ceil((C1+C2)/16), mode 0) of E = [W1^T | W2^T | 0]; bias padded to the
 * same width.  Cin % 16 == 0, Cin <= 128, C1 + C2 <= 64; C2 = 0: one head. */
int lt_head_gather_fwd(const float* feats, const int64_t* idx, const float* packed, const float* bias, int64_t N, int64_t V,
                       int32_t Cin, int32_t C1, int32_t C2, float* out1, float* out2, lt_stream_t stream);

/* ------------------------------------------------------------------ parameter update
 * torch.optim.Adam.step() over the norm parameters + update_ema_norm (latte/tta/tools/latte.py:317-323,
 * latte/common/solver/build.py:7-52, latte/common/utils/ema_util.py:99-118) on FLAT fp32 buffers of n scalars, one
 * launch: grad is scaled by grad_scale first (1/world after a sum all-reduce), Adam with L2 weight_decay and torch's
 * bias corrections for `step` (counted from 1), then ema = ema_alpha*ema + (1-ema_alpha)*param (ema may be NULL). */
int lt_adam_ema_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* ema, int64_t n, float lr,
                     float beta1, float beta2, float eps, float weight_decay, int64_t step, float ema_alpha,
                     float grad_scale, lt_stream_t stream);

/* ------------------------------------------------------------------ point <-> voxel
 * F.spvoxelize(feats, idx, counts)  (torchsparse_utils.py:20,23,54): out[idx[i]] += f[i]/counts[idx[i]] */
int lt_voxelize_fwd(const float* feats, const int32_t* idx, const int32_t* counts, float* out, int64_t N,
                    int32_t C, int64_t V, lt_stream_t stream);
int lt_voxelize_bwd(const float* gout, const int32_t* idx, const int32_t* counts, float* gfeats, int64_t N,
                    int32_t C, int64_t V, lt_stream_t stream);
/* F.calc_ti_weights(coords, idx_query, scale)  (torchsparse_utils.py:75): idx_query int64 [8,N] -> w [8,N] */
int lt_calc_ti_weights(const float* pcoords, const int64_t* idx_query, int64_t N, int32_t scale, float* w,
                       lt_stream_t stream);
/* F.spdevoxelize(feats, idx[N,8], w[N,8])  (torchsparse_utils.py:81,93) */
int lt_devoxelize_fwd(const float* feats, const int32_t* idx, const float* w, float* out, int64_t N, int32_t C,
                      int64_t V, lt_stream_t stream);
int lt_devoxelize_bwd(const float* gout, const int32_t* idx, const float* w, float* gfeats, int64_t N, int32_t C,
                      int64_t V, lt_stream_t stream);

/* The two scatter-adds above (spvoxelize forward, spdevoxelize backward) as atomic-free gathers over a CSR of the index
 * map, built once per map: perm[off[v] .. off[v+1]) = the entries e (ascending) with key[e] == v; keys outside [0,V) are
 * dropped.  Summation order = a sequential index_add's (the CPU order of torchsparse's fallbacks). */
int64_t lt_csr_workspace_bytes(int64_t n);
int lt_csr_build(const int32_t* key, int64_t n, int64_t V, void* ws, int64_t ws_bytes, int32_t* perm /*[n]*/,
                 int32_t* off /*[V+1]*/, lt_stream_t stream);
/* = lt_voxelize_fwd with key = idx[N] (counts = row lengths) */
int lt_voxelize_csr_fwd(const float* feats, const int32_t* perm, const int32_t* off, float* out, int64_t V, int32_t C,
                        lt_stream_t stream);
/* = lt_devoxelize_bwd with key = idx[N*8] flattened (entry e = 8*point + corner), w[N*8] */
int lt_devoxelize_csr_bwd(const float* gout, const int32_t* perm, const int32_t* off, const float* w, float* gfeats,
                          int64_t V, int32_t C, lt_stream_t stream);
/* The same backward WITHOUT a CSR of the devoxelize map (F.spdevoxelize's backward, torchsparse_utils.py:81,93).  All points
 * of one cell (floor(z.C / s), the point -> voxel map of the level: torchsparse_utils.py:40-47 and :69-76 hash the same
 * floor coordinates) share their 8 corner voxels, so the points that reach voxel v through corner c are the points of ONE
 * cell: cells[V,8,2] = that cell's row range [j0, j1) of the point -> voxel CSR ((0,0): none), filled by
 * lt_devoxelize_cells from the first point of every CSR row (p2v_perm / p2v_off over Vp rows, lt_csr_build of
 * p2v_idx[N]).  lt_devoxelize_cell_bwd then sums, per voxel, corner by corner and point-ascending inside a cell
 * (deterministic; a different fp32 order than lt_devoxelize_csr_bwd's point-major one).  Points with p2v_idx < 0 (their
 * cell is no voxel of the level) are added atomically afterwards. */
int lt_devoxelize_cells(const int32_t* v2p_idx /*[N,8]*/, const int32_t* p2v_perm, const int32_t* p2v_off /*[Vp+1]*/,
                        int32_t* cells /*[V,8,2]*/, int64_t N, int64_t Vp, int64_t V, lt_stream_t stream);
int lt_devoxelize_cell_bwd(const float* gout, const int32_t* cells, const int32_t* p2v_perm, const int32_t* p2v_off,
                           const int32_t* p2v_idx /*[N]*/, const int32_t* v2p_idx /*[N,8]*/, const float* w /*[N,8]*/,
                           float* gfeats /*[V,C]*/, int64_t N, int64_t V, int32_t C, lt_stream_t stream);

/* ------------------------------------------------------------------ ST-voxel aggregation (HR_Block / CR_Block)
 * see latte/models/hr_module.py:135-331 and :344-420; declared in the second half of this header */
#include "latte_hip_st.h"

/* ------------------------------------------------------------------ the backbone pass as one host call
 * see latte/models/spvcnn.py:113-165; declared in the third part of this header */
#include "latte_hip_net.h"

#ifdef __cplusplus
}
#endif
#endif
