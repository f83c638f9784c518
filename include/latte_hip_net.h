/* The SPVCNN backbone pass as ONE host call; included by latte_hip.h.
 *
 * Reference: latte/models/spvcnn.py:113-165 (SPVCNN.forward: stem, 4 down stages, 4 up stages, 3 point branches),
 * latte/models/basic_blocks.py:13-76 (conv -> BatchNorm -> ReLU blocks, ResidualBlock), latte/models/torchsparse_utils.py
 * (initial_voxelize / point_to_voxel / voxel_to_point), and torch.autograd's walk over them in the backward
 * (latte/tta/tools/latte.py:313).  The reference runs ~250 Python-level ops per pass; module by module that is ~5 ms of
 * host time per pass here as well, which is the whole step on a one-scan batch (BASELINE configs[0]).
 *
 * The caller describes the network ONCE as an op list over numbered feature buffers (host memory, no torch types) and
 * hands over, per batch, the index structures the lt_kmap_* / lt_csr_* / lt_calc_ti_weights calls produced.  One call then
 * issues every kernel of the pass -- pure sequencing of entry points that also exist on their own (lt_conv_bn_fwd/bwd,
 * lt_spconv_stem_fwd, lt_bn_*, lt_voxelize_*, lt_devoxelize_*), so the results are those of the per-layer calls.
 * Feature buffers live in ONE caller-owned arena; the call lays them out itself (keep = 0: a buffer's space is reused
 * after its last reader; keep = 1: every buffer and every conv output is kept for lt_net_backward).
 * Frozen conv / linear weights, BatchNorm on batch statistics (TRAIN.UPDATE_PARAMS == "norm",
 * latte/common/utils/ema_util.py:6-41): the backward returns the BatchNorm gamma / beta gradients only.  No sync. */
#ifndef LATTE_HIP_NET_H
#define LATTE_HIP_NET_H

enum {                /* lt_net_op_t.kind */
  LT_NET_VOX = 1,     /* out[level rows] = mean of the points of every voxel (F.spvoxelize, torchsparse_utils.py:20,23,54) */
  LT_NET_DEVOX = 2,   /* out[points] = trilinear blend of 8 voxel rows (F.spdevoxelize, torchsparse_utils.py:81,93) */
  LT_NET_CONV_BN = 3, /* out = act(bn(conv(in0)) [+ in1]) (basic_blocks.py:17-23,51-75; spvcnn.py:95-111 for relu = 3) */
  LT_NET_CAT = 4      /* out = [in0 | in1] along the channels (torchsparse.cat, spvcnn.py:144,148,156,160) */
};
enum {                /* lt_net_op_t.map of a CONV_BN op: which kernel map / plan of lt_net_maps_t it runs over */
  LT_NET_MAP_K3 = 0,    /* 3^3 stride 1 on `level`                                 (plan_k3[level]) */
  LT_NET_MAP_DOWN = 1,  /* 2^3 stride 2, level -> level + 1                        (plan_down[level]; backward: plan_up) */
  LT_NET_MAP_UP = 2,    /* transposed 2^3 stride 2, level + 1 -> level             (plan_up[level]; backward: plan_down) */
  LT_NET_MAP_DENSE = 3, /* 1x1x1 conv / nn.Linear on the rows of `level` (5 = points): identity plan, K = 1 */
  LT_NET_MAP_STEM = 4   /* the 4-channel first conv, 3^3 on level 0: lt_spconv_stem_fwd, unpacked weights */
};

typedef struct {
  int32_t kind;   /* LT_NET_* */
  int32_t in0;    /* input buffer */
  int32_t in1;    /* CONV_BN: residual / post-add addend; CAT: second operand; -1: none */
  int32_t out;    /* output buffer (written by exactly one op) */
  int32_t layer;  /* CONV_BN: index into the layer table */
  int32_t map;    /* CONV_BN: LT_NET_MAP_* */
  int32_t level;  /* voxel level 0..4 (tensor stride 1, 2, 4, 8, 16) the op works on; 5 = the points (DENSE only) */
  int32_t relu;   /* CONV_BN: lt_bn_apply's relu argument (0, 1, or 3 = relu(bn(x)) + in1) */
} lt_net_op_t;

typedef struct {
  int32_t rows;      /* 0..4: n_vox[rows] rows; 5: n_points rows */
  int32_t channels;  /* multiple of 4 */
} lt_net_buf_t;

typedef struct {       /* conv (or linear) + BatchNorm layer; device pointers */
  const float* w_fwd;  /* lt_spconv_pack_weights mode 0 image (STEM: the unpacked kernel [27,4,Cout]) */
  const float* w_bwd;  /* backward-input image: mode 2 for K3, mode 1 otherwise; NULL: forward only / STEM */
  const float* gamma;
  const float* beta;
  float eps;
  int32_t cin, cout, K;
  int64_t dgamma_off, dbeta_off; /* where lt_net_backward writes the parameter gradients, in floats from `dparams` */
} lt_net_layer_t;

typedef struct {
  const lt_net_op_t* ops;
  int32_t n_ops;
  const lt_net_buf_t* bufs;
  int32_t n_bufs;
  const lt_net_layer_t* layers;
  int32_t n_layers;
  int32_t out_buf; /* the pass's result (kept whatever `keep` says) */
} lt_net_prog_t;

typedef struct {           /* index structures of one batch; device pointers */
  int64_t n_points;
  int64_t n_vox[5];
  const void* plan_k3[5];   /* lt_kmap_plan of the 3^3 map of every level */
  const void* plan_down[4]; /* lt_kmap_plan of the 2^3 stride-2 map level -> level + 1 (its [8, n_vox[level+1]] table) */
  const void* plan_up[4];   /* lt_kmap_plan of lt_kmap_reverse of that map ([8, n_vox[level]]) */
  const void* plan_dense[6]; /* lt_kmap_plan_identity over the rows of level 0..4 / the points; NULL where unused */
  /* point <-> voxel maps per level (NULL where the network has none; SPVCNN uses levels 0, 2 and 4) */
  const int32_t* p2v_perm[5]; /* lt_csr_build of the point -> voxel index: lt_voxelize_csr_fwd */
  const int32_t* p2v_off[5];
  const int32_t* p2v_idx[5];    /* the index itself and the voxel populations: lt_voxelize_bwd */
  const int32_t* p2v_counts[5];
  const int32_t* v2p_idx[5];    /* [n_points, 8] corner rows and trilinear weights: lt_devoxelize_fwd */
  const float* v2p_w[5];
  const int32_t* v2p_cells[5];  /* lt_devoxelize_cells over the level's p2v CSR: lt_devoxelize_cell_bwd (backward only) */
} lt_net_maps_t;

/* Identity plan (K = 1, row i reads row i): what a 1x1x1 conv / nn.Linear needs to run through lt_spconv_plan_fwd*
 * (basic_blocks.py:66-69 shortcut convs, spvcnn.py:95-111 point branches).  plan: lt_kmap_plan_bytes(1, V) bytes. */
int lt_kmap_plan_identity(int64_t V, void* plan, int64_t plan_bytes, lt_stream_t stream);

/* ext[b] != NULL: buffer b is the caller's memory (the input features), never laid out in the arena.
 * lt_net_arena_floats: floats of arena lt_net_forward needs for this program, batch and `keep`. */
int64_t lt_net_arena_floats(const lt_net_prog_t* prog, const lt_net_maps_t* maps, const float* const* ext, int32_t keep);
/* buf_off (HOST, [n_bufs], optional): offset in floats of every buffer in the arena (-1: external, or -- keep = 0 --
 * meaningful only for out_buf, whose space is never reused). */
int lt_net_forward(const lt_net_prog_t* prog, const lt_net_maps_t* maps, const float* const* ext, float* arena,
                   int64_t arena_floats, int32_t keep, int64_t* buf_off, lt_stream_t stream);
/* Backward of a keep = 1 forward over the same program, maps, ext and arena.  gout: gradient of out_buf.  dparams: the
 * BatchNorm gradients land at the layers' dgamma_off / dbeta_off.  garena: scratch of lt_net_grad_arena_floats floats
 * (gradient buffers, reused as the walk proceeds).  Gradients of a buffer's several readers are summed in the order
 * the reversed op list meets them. */
int64_t lt_net_grad_arena_floats(const lt_net_prog_t* prog, const lt_net_maps_t* maps, const float* const* ext);
int lt_net_backward(const lt_net_prog_t* prog, const lt_net_maps_t* maps, const float* const* ext, const float* arena,
                    int64_t arena_floats, const float* gout, float* dparams, float* garena, int64_t garena_floats,
                    lt_stream_t stream);

#endif
