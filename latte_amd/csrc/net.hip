// The SPVCNN backbone pass as one host call (include/latte_hip_net.h): an op list over numbered feature buffers, laid out
// in one caller-owned arena, executed by sequencing the library's own per-layer entry points.  Reference structure:
// latte/models/spvcnn.py:113-165, latte/models/basic_blocks.py:13-76, latte/models/torchsparse_utils.py:11-100.
//
// Why it exists: module by module a pass costs ~5 ms of Python on the host (49 conv+BN composites, 6 point<->voxel
// bridges, 3 point branches), three passes per TTA iteration -- the whole step on a one-scan batch.  Here the host cost
// of a pass is its ~300 kernel launches.
#include "common.h"
#include <map>
#include <vector>
#include "../../include/latte_hip.h"

namespace {

// ---------------------------------------------------------------- small kernels of the executor
__global__ void k_plan_identity(int64_t V, int* __restrict__ order, int* __restrict__ tnbr,
                                unsigned long long* __restrict__ tmask, int* __restrict__ torder) {
  const int64_t tile = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);  // one wave per 64-row tile
  const int lane = threadIdx.x & 63;
  if (tile >= (V + 63) / 64) return;  // wave-uniform
  const int64_t slot = tile * 64 + lane;
  const int v = slot < V ? (int)slot : -1;
  if (slot < V) order[slot] = v;
  tnbr[tile * 64 + lane] = v;
  const unsigned long long m = __ballot(v >= 0);
  if (lane == 0) {
    tmask[tile] = m;
    torder[tile] = (int)tile;
  }
}

// out[v] = [a[v] | b[v]], channel counts in float4 units
__global__ void k_cat2(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ out, int64_t V,
                       int ca4, int cb4) {
  const int c4 = ca4 + cb4;
  const int64_t n = V * c4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = i / c4;
    const int c = (int)(i - v * c4);
    out[i] = c < ca4 ? a[v * ca4 + c] : b[v * cb4 + (c - ca4)];
  }
}

// the backward of k_cat2: ga (+)= g[:, :ca], gb (+)= g[:, ca:]; a NULL destination is skipped
__global__ void k_split2(const float4* __restrict__ g, float4* __restrict__ ga, float4* __restrict__ gb, int64_t V, int ca4,
                         int cb4, int acc_a, int acc_b) {
  const int c4 = ca4 + cb4;
  const int64_t n = V * c4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = i / c4;
    const int c = (int)(i - v * c4);
    const float4 x = g[i];
    float4* d = c < ca4 ? (ga ? ga + v * ca4 + c : nullptr) : (gb ? gb + v * cb4 + (c - ca4) : nullptr);
    if (!d) continue;
    if (c < ca4 ? acc_a : acc_b) {
      float4 o = *d;
      o.x += x.x; o.y += x.y; o.z += x.z; o.w += x.w;
      *d = o;
    } else {
      *d = x;
    }
  }
}

__global__ void k_add4(float4* __restrict__ dst, const float4* __restrict__ src, int64_t n4) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 o = dst[i];
    const float4 x = src[i];
    o.x += x.x; o.y += x.y; o.z += x.z; o.w += x.w;
    dst[i] = o;
  }
}

// ---------------------------------------------------------------- host side
inline int64_t al64(int64_t floats) { return (floats + 63) / 64 * 64; }  // 256-B pieces

inline int64_t rows_of(const lt_net_maps_t* m, int r) { return r == 5 ? m->n_points : m->n_vox[r]; }

struct ConvGeo {
  const void* plan = nullptr;      // forward
  const void* plan_bwd = nullptr;  // backward-input
  int64_t Vin = 0, Vout = 0;
};

int conv_geo(const lt_net_op_t& op, const lt_net_maps_t* m, ConvGeo* g) {
  const int l = op.level;
  switch (op.map) {
    case LT_NET_MAP_K3:
    case LT_NET_MAP_STEM:
      LT_REQUIRE(l >= 0 && l < 5, LT_ERR_ARG, "lt_net: 3^3 conv on level %d", l);
      g->plan = g->plan_bwd = m->plan_k3[l];
      g->Vin = g->Vout = m->n_vox[l];
      break;
    case LT_NET_MAP_DOWN:
      LT_REQUIRE(l >= 0 && l < 4, LT_ERR_ARG, "lt_net: strided conv from level %d", l);
      g->plan = m->plan_down[l];
      g->plan_bwd = m->plan_up[l];
      g->Vin = m->n_vox[l];
      g->Vout = m->n_vox[l + 1];
      break;
    case LT_NET_MAP_UP:
      LT_REQUIRE(l >= 0 && l < 4, LT_ERR_ARG, "lt_net: transposed conv onto level %d", l);
      g->plan = m->plan_up[l];
      g->plan_bwd = m->plan_down[l];
      g->Vin = m->n_vox[l + 1];
      g->Vout = m->n_vox[l];
      break;
    case LT_NET_MAP_DENSE:
      LT_REQUIRE(l >= 0 && l < 6, LT_ERR_ARG, "lt_net: dense layer on rows %d", l);
      g->plan = g->plan_bwd = m->plan_dense[l];
      g->Vin = g->Vout = rows_of(m, l);
      break;
    default:
      LT_REQUIRE(false, LT_ERR_ARG, "lt_net: unknown map kind %d", op.map);
  }
  LT_REQUIRE(g->plan != nullptr, LT_ERR_ARG, "lt_net: the batch carries no plan for map kind %d on level %d", op.map, l);
  return LT_OK;
}

// what every entry point checks first: buffer / layer indices, shapes against the maps
int validate(const lt_net_prog_t* p, const lt_net_maps_t* m) {
  LT_REQUIRE(p && m && p->ops && p->bufs && p->n_ops > 0 && p->n_bufs > 0, LT_ERR_ARG, "lt_net: null program");
  LT_REQUIRE(p->out_buf >= 0 && p->out_buf < p->n_bufs, LT_ERR_ARG, "lt_net: out_buf %d", p->out_buf);
  LT_REQUIRE(m->n_points >= 1, LT_ERR_ARG, "lt_net: no points");
  for (int b = 0; b < p->n_bufs; ++b)
    LT_REQUIRE(p->bufs[b].rows >= 0 && p->bufs[b].rows <= 5 && p->bufs[b].channels >= 4 && p->bufs[b].channels % 4 == 0,
               LT_ERR_ARG, "lt_net: buffer %d (rows %d, channels %d)", b, p->bufs[b].rows, p->bufs[b].channels);
  std::vector<char> defined(p->n_bufs, 0);
  for (int i = 0; i < p->n_ops; ++i) {
    const lt_net_op_t& op = p->ops[i];
    auto ok = [&](int b) { return b >= 0 && b < p->n_bufs; };
    LT_REQUIRE(ok(op.in0) && ok(op.out) && (op.in1 == -1 || ok(op.in1)), LT_ERR_ARG, "lt_net: op %d names a bad buffer", i);
    LT_REQUIRE(!defined[op.out], LT_ERR_ARG, "lt_net: buffer %d is written twice (op %d)", op.out, i);
    defined[op.out] = 1;
    const lt_net_buf_t &bi = p->bufs[op.in0], &bo = p->bufs[op.out];
    switch (op.kind) {
      case LT_NET_VOX:
        LT_REQUIRE(op.level >= 0 && op.level < 5 && m->p2v_perm[op.level] && m->p2v_off[op.level], LT_ERR_ARG,
                   "lt_net: op %d: no point->voxel map on level %d", i, op.level);
        LT_REQUIRE(bi.rows == 5 && bo.rows == op.level && bi.channels == bo.channels, LT_ERR_ARG, "lt_net: op %d: shapes", i);
        break;
      case LT_NET_DEVOX:
        LT_REQUIRE(op.level >= 0 && op.level < 5 && m->v2p_idx[op.level] && m->v2p_w[op.level], LT_ERR_ARG,
                   "lt_net: op %d: no voxel->point map on level %d", i, op.level);
        LT_REQUIRE(bi.rows == op.level && bo.rows == 5 && bi.channels == bo.channels, LT_ERR_ARG, "lt_net: op %d: shapes", i);
        break;
      case LT_NET_CAT:
        LT_REQUIRE(op.in1 >= 0 && p->bufs[op.in1].rows == bi.rows && bo.rows == bi.rows &&
                       bo.channels == bi.channels + p->bufs[op.in1].channels,
                   LT_ERR_ARG, "lt_net: op %d: concat shapes", i);
        break;
      case LT_NET_CONV_BN: {
        LT_REQUIRE(p->layers && op.layer >= 0 && op.layer < p->n_layers, LT_ERR_ARG, "lt_net: op %d: layer %d", i, op.layer);
        const lt_net_layer_t& L = p->layers[op.layer];
        ConvGeo g;
        int rc = conv_geo(op, m, &g);
        if (rc) return rc;
        LT_REQUIRE(L.w_fwd && L.gamma && L.beta, LT_ERR_ARG, "lt_net: layer %d: null parameter", op.layer);
        LT_REQUIRE(bi.channels == L.cin && bo.channels == L.cout, LT_ERR_ARG, "lt_net: op %d: channels %d -> %d, layer %d -> %d",
                   i, bi.channels, bo.channels, L.cin, L.cout);
        LT_REQUIRE(rows_of(m, bi.rows) == g.Vin && rows_of(m, bo.rows) == g.Vout, LT_ERR_ARG, "lt_net: op %d: rows", i);
        LT_REQUIRE(g.Vout >= 2, LT_ERR_ARG, "lt_net: op %d: batch statistics need more than one row", i);
        LT_REQUIRE(op.relu == 0 || op.relu == 1 || (op.relu == 3 && op.in1 >= 0), LT_ERR_ARG, "lt_net: op %d: relu %d", i, op.relu);
        LT_REQUIRE(op.in1 < 0 || (p->bufs[op.in1].rows == bo.rows && p->bufs[op.in1].channels == bo.channels), LT_ERR_ARG,
                   "lt_net: op %d: residual shape", i);
        LT_REQUIRE(op.map == LT_NET_MAP_STEM ? (L.cin == 4 && L.K == 27) : (L.cin % 16 == 0), LT_ERR_ARG,
                   "lt_net: op %d: Cin = %d", i, L.cin);
        LT_REQUIRE(L.cout % 16 == 0 && L.cout <= 1024, LT_ERR_ARG, "lt_net: op %d: Cout = %d", i, L.cout);
        break;
      }
      default:
        LT_REQUIRE(false, LT_ERR_ARG, "lt_net: op %d: unknown kind %d", i, op.kind);
    }
  }
  return LT_OK;
}

// exact-size free lists: the backbone's buffer sizes repeat (a level's rows x a handful of widths)
struct Pool {
  std::multimap<int64_t, int64_t> free_;  // size -> offset
  int64_t top = 0;
  int64_t take(int64_t n) {
    auto it = free_.find(n);
    if (it != free_.end()) {
      const int64_t o = it->second;
      free_.erase(it);
      return o;
    }
    const int64_t o = top;
    top += n;
    return o;
  }
  void give(int64_t off, int64_t n) { free_.emplace(n, off); }
};

struct FwdLayout {
  std::vector<int64_t> off;   // per buffer, floats; -1 = external
  std::vector<int64_t> aux;   // per op: conv output kept for the backward (keep) or -1 (BatchNorm applied in place)
  std::vector<int64_t> stat;  // per op: mean | var
  int64_t ws = 0, ws_bytes = 0, total = 0;
};

int64_t buf_floats(const lt_net_prog_t* p, const lt_net_maps_t* m, int b) {
  return al64(rows_of(m, p->bufs[b].rows) * (int64_t)p->bufs[b].channels);
}

// how many ops read each buffer
std::vector<int> reader_count(const lt_net_prog_t* p) {
  std::vector<int> n(p->n_bufs, 0);
  for (int i = 0; i < p->n_ops; ++i) {
    ++n[p->ops[i].in0];
    if (p->ops[i].in1 >= 0 && p->ops[i].in1 != p->ops[i].in0) ++n[p->ops[i].in1];
  }
  return n;
}

// Devoxelize ops folded into their reader: out = relu(bn(conv(in0))) + devox(v) with the devoxelized tensor read by this
// op alone is issued as conv + statistics, finalize, lt_bn_apply_devox -- the [points, C] addend is formed inside the apply
// kernel from the voxel rows and never exists in memory (2 x 4NC bytes per point branch and pass: ~1 ms of the a2d2 step).
// fused[j] = index of the DEVOX op folded into CONV_BN op j (-1: none); skip[d] = 1 for those DEVOX ops.  Its backward is
// unchanged (the addend's gradient is the output's, see addend_takes_dy).  LT_NET_FUSE_DEVOX=0 switches it off (A/B).
struct DevoxFusion {
  std::vector<int> fused;
  std::vector<char> skip;
};
DevoxFusion devox_fusion(const lt_net_prog_t* p) {
  static const bool on = [] {
    const char* e = getenv("LT_NET_FUSE_DEVOX");
    return !(e && e[0] == '0');
  }();
  DevoxFusion f;
  f.fused.assign(p->n_ops, -1);
  f.skip.assign(p->n_ops, 0);
  if (!on) return f;
  const std::vector<int> readers = reader_count(p);
  std::vector<int> producer(p->n_bufs, -1);
  for (int i = 0; i < p->n_ops; ++i) producer[p->ops[i].out] = i;
  for (int j = 0; j < p->n_ops; ++j) {
    const lt_net_op_t& op = p->ops[j];
    if (op.kind != LT_NET_CONV_BN || op.relu != 3 || op.in1 < 0 || op.in1 == op.in0 || op.map == LT_NET_MAP_STEM) continue;
    const int d = producer[op.in1];
    if (d < 0 || d >= j || p->ops[d].kind != LT_NET_DEVOX || readers[op.in1] != 1 || op.in1 == p->out_buf) continue;
    f.fused[j] = d;
    f.skip[d] = 1;
  }
  return f;
}

int fwd_layout(const lt_net_prog_t* p, const lt_net_maps_t* m, const float* const* ext, bool keep, FwdLayout* L) {
  const int nb = p->n_bufs, no = p->n_ops;
  L->off.assign(nb, -1);
  L->aux.assign(no, -1);
  L->stat.assign(no, -1);
  std::vector<int> last(nb, -1);
  for (int i = 0; i < no; ++i) {
    last[p->ops[i].in0] = i;
    if (p->ops[i].in1 >= 0) last[p->ops[i].in1] = i;
  }
  last[p->out_buf] = no;
  const DevoxFusion fu = devox_fusion(p);
  std::vector<char> virt(nb, 0);  // outputs of folded devoxelize ops: never in memory
  for (int j = 0; j < no; ++j)
    if (fu.fused[j] >= 0) {
      const lt_net_op_t& d = p->ops[fu.fused[j]];
      virt[d.out] = 1;
      if (last[d.in0] < j) last[d.in0] = j;  // the voxel rows are read by the apply kernel of op j
    }
  Pool pool;
  const int64_t shared_stat = keep ? -1 : pool.take(al64(2 * 1024));
  int64_t wsb = 256;
  for (int i = 0; i < no; ++i) {
    const lt_net_op_t& op = p->ops[i];
    LT_REQUIRE(!(ext && ext[op.out]), LT_ERR_ARG, "lt_net: op %d writes the external buffer %d", i, op.out);
    for (int b : {op.in0, op.in1})
      LT_REQUIRE(b < 0 || (ext && ext[b]) || L->off[b] >= 0 || virt[b], LT_ERR_ARG,
                 "lt_net: op %d reads buffer %d before any op wrote it", i, b);
    if (fu.skip[i]) continue;  // folded into its reader: no buffer, and its input stays alive until that reader
    L->off[op.out] = pool.take(buf_floats(p, m, op.out));
    if (op.kind == LT_NET_CONV_BN) {
      const lt_net_layer_t& ly = p->layers[op.layer];
      const int64_t vout = rows_of(m, p->bufs[op.out].rows);
      if (keep) {
        L->aux[i] = pool.take(al64(vout * ly.cout));
        L->stat[i] = pool.take(al64(2 * (int64_t)ly.cout));
      } else {
        L->stat[i] = shared_stat;
      }
      const int64_t w = lt_conv_bn_workspace_bytes(vout, ly.cout);
      if (w > wsb) wsb = w;
    }
    if (!keep) {
      const int ins[2] = {op.in0, op.in1 == op.in0 ? -1 : op.in1};
      for (int b : ins)
        if (b >= 0 && L->off[b] >= 0 && last[b] == i) pool.give(L->off[b], buf_floats(p, m, b));
      if (fu.fused[i] >= 0) {  // ... and the voxel rows of the devoxelize folded into this op
        const int vb = p->ops[fu.fused[i]].in0;
        if (vb != op.in0 && L->off[vb] >= 0 && last[vb] == i) pool.give(L->off[vb], buf_floats(p, m, vb));
      }
      if (last[op.out] < 0) pool.give(L->off[op.out], buf_floats(p, m, op.out));  // an output nobody reads
    }
  }
  L->ws = pool.top;
  L->ws_bytes = wsb;
  L->total = pool.top + al64((wsb + 3) / 4);
  return LT_OK;
}

inline int launch_add(float* dst, const float* src, int64_t floats, hipStream_t s) {
  const int64_t n4 = floats / 4;
  hipLaunchKernelGGL(k_add4, dim3(lt_grid(n4, 256)), dim3(256), 0, s, (float4*)dst, (const float4*)src, n4);
  LT_LAUNCH_CHECK();
  return LT_OK;
}

// which buffers carry a gradient: everything downstream of a BatchNorm (its gamma / beta train); the input features and
// what is computed from them alone do not
std::vector<char> grad_needed(const lt_net_prog_t* p) {
  std::vector<char> need(p->n_bufs, 0);
  for (int i = 0; i < p->n_ops; ++i) {
    const lt_net_op_t& op = p->ops[i];
    need[op.out] = op.kind == LT_NET_CONV_BN || need[op.in0] || (op.in1 >= 0 && need[op.in1]);
  }
  return need;
}

// relu = 3 (out = relu(bn(conv(in0))) + in1): the addend's gradient IS the output's.  When this op is the addend's only
// reader nobody will ever add to that gradient, so it takes over the output's gradient buffer instead of a copy
// (three [points, C] copies per backward otherwise: 0.36 ms of the a2d2 step).
inline bool addend_takes_dy(const lt_net_op_t& op, const std::vector<char>& need, const std::vector<int>& readers) {
  return op.kind == LT_NET_CONV_BN && op.relu == 3 && op.in1 >= 0 && op.in1 != op.in0 && need[op.in1] && readers[op.in1] == 1;
}

struct BwdLayout {
  std::vector<int64_t> goff;  // per buffer: offset of its gradient in garena at the time it is live (-1: none)
  int64_t tmp[3] = {0, 0, 0};
  int64_t ws = 0, ws_bytes = 0, total = 0;
};

// gradient buffers live from the first reader met by the reversed walk to the op that produced the buffer
int bwd_layout(const lt_net_prog_t* p, const lt_net_maps_t* m, BwdLayout* L) {
  const int nb = p->n_bufs, no = p->n_ops;
  L->goff.assign(nb, -1);
  const std::vector<char> need = grad_needed(p);
  const std::vector<int> readers = reader_count(p);
  Pool pool;
  int64_t big = 64, wsb = 256;
  for (int b = 0; b < nb; ++b)
    if (buf_floats(p, m, b) > big) big = buf_floats(p, m, b);
  for (int t = 0; t < 3; ++t) L->tmp[t] = pool.take(big);
  for (int i = no - 1; i >= 0; --i) {
    const lt_net_op_t& op = p->ops[i];
    const bool steal = addend_takes_dy(op, need, readers);
    const int ins[2] = {op.in0, op.in1 == op.in0 ? -1 : op.in1};
    for (int b : ins)
      if (b >= 0 && need[b] && L->goff[b] < 0 && !(steal && b == op.in1)) L->goff[b] = pool.take(buf_floats(p, m, b));
    if (steal && op.out != p->out_buf) {  // the addend's gradient lives on in the output's buffer
      L->goff[op.in1] = L->goff[op.out];
      L->goff[op.out] = -1;
    }
    if (op.out != p->out_buf && L->goff[op.out] >= 0) pool.give(L->goff[op.out], buf_floats(p, m, op.out));
    if (op.kind == LT_NET_CONV_BN) {
      const int64_t w = lt_conv_bn_workspace_bytes(rows_of(m, p->bufs[op.out].rows), p->layers[op.layer].cout);
      if (w > wsb) wsb = w;
    }
  }
  L->ws = pool.top;
  L->ws_bytes = wsb;
  L->total = pool.top + al64((wsb + 3) / 4);
  return LT_OK;
}

}  // namespace

extern "C" int lt_kmap_plan_identity(int64_t V, void* plan, int64_t plan_bytes, hipStream_t stream) {
  LT_REQUIRE(V >= 0 && V < INT_MAX, LT_ERR_ARG, "lt_kmap_plan_identity: V=%lld", (long long)V);
  if (V == 0) return LT_OK;
  LT_REQUIRE(plan && ((uintptr_t)plan & 15) == 0, LT_ERR_ARG, "lt_kmap_plan_identity: plan must be a 16-byte aligned buffer");
  LtPlanView pv = lt_plan_view(plan, 1, V);
  LT_REQUIRE(plan_bytes >= pv.bytes, LT_ERR_WORKSPACE, "lt_kmap_plan_identity: plan needs %lld bytes, got %lld",
             (long long)pv.bytes, (long long)plan_bytes);
  const int64_t T = (V + 63) / 64;
  hipLaunchKernelGGL(k_plan_identity, dim3((unsigned)((T + 3) / 4)), dim3(256), 0, stream, V, pv.order, pv.tnbr, pv.tmask,
                     pv.torder);
  LT_LAUNCH_CHECK();
  return LT_OK;
}

extern "C" int64_t lt_net_arena_floats(const lt_net_prog_t* prog, const lt_net_maps_t* maps, const float* const* ext,
                                       int32_t keep) {
  if (validate(prog, maps)) return -1;
  FwdLayout L;
  if (fwd_layout(prog, maps, ext, keep != 0, &L)) return -1;
  return L.total;
}

extern "C" int lt_net_forward(const lt_net_prog_t* p, const lt_net_maps_t* m, const float* const* ext, float* arena,
                              int64_t arena_floats, int32_t keep, int64_t* buf_off, hipStream_t stream) {
  int rc = validate(p, m);
  if (rc) return rc;
  FwdLayout L;
  rc = fwd_layout(p, m, ext, keep != 0, &L);
  if (rc) return rc;
  LT_REQUIRE(arena && ((uintptr_t)arena & 255) == 0 && arena_floats >= L.total, LT_ERR_WORKSPACE,
             "lt_net_forward: the arena needs %lld floats (256-byte aligned), got %lld", (long long)L.total,
             (long long)arena_floats);
  void* ws = arena + L.ws;
  auto ptr = [&](int b) -> const float* { return (ext && ext[b]) ? ext[b] : arena + L.off[b]; };
  const DevoxFusion fu = devox_fusion(p);
  {
    std::vector<char> virt(p->n_bufs, 0);
    for (int i = 0; i < p->n_ops; ++i)
      if (fu.skip[i]) virt[p->ops[i].out] = 1;
    for (int b = 0; b < p->n_bufs; ++b)
      LT_REQUIRE((ext && ext[b]) || L.off[b] >= 0 || virt[b], LT_ERR_ARG,
                 "lt_net_forward: buffer %d is neither written by an op nor external", b);
  }
  for (int i = 0; i < p->n_ops; ++i) {
    const lt_net_op_t& op = p->ops[i];
    if (fu.skip[i]) continue;  // a devoxelize folded into the apply kernel of its reader
    const float* x = ptr(op.in0);
    float* y = arena + L.off[op.out];
    const int C = p->bufs[op.out].channels;
    switch (op.kind) {
      case LT_NET_VOX:
        rc = lt_voxelize_csr_fwd(x, m->p2v_perm[op.level], m->p2v_off[op.level], y, m->n_vox[op.level], C, stream);
        break;
      case LT_NET_DEVOX:
        rc = lt_devoxelize_fwd(x, m->v2p_idx[op.level], m->v2p_w[op.level], y, m->n_points, C, m->n_vox[op.level], stream);
        break;
      case LT_NET_CAT: {
        const int64_t V = rows_of(m, p->bufs[op.out].rows);
        const int ca4 = p->bufs[op.in0].channels / 4, cb4 = p->bufs[op.in1].channels / 4;
        hipLaunchKernelGGL(k_cat2, dim3(lt_grid(V * (ca4 + cb4), 256)), dim3(256), 0, stream, (const float4*)x,
                           (const float4*)ptr(op.in1), (float4*)y, V, ca4, cb4);
        LT_LAUNCH_CHECK();
        rc = LT_OK;
        break;
      }
      case LT_NET_CONV_BN: {
        const lt_net_layer_t& ly = p->layers[op.layer];
        ConvGeo g;
        rc = conv_geo(op, m, &g);
        if (rc) return rc;
        float* co = keep ? arena + L.aux[i] : y;  // forward only: BatchNorm is applied in place on the conv output
        float* mean = arena + L.stat[i];
        float* var = mean + ly.cout;
        if (fu.fused[i] >= 0) {
          // out = relu(bn(conv(in0))) + devox(voxel rows): the addend is formed inside the apply kernel
          const lt_net_op_t& d = p->ops[fu.fused[i]];
          int64_t nb = 0;
          rc = lt_spconv_plan_fwd_stats_nb(x, ly.w_fwd, g.plan, co, g.Vin, g.Vout, ly.cin, ly.cout, ly.K, ws, &nb, stream);
          if (!rc) rc = lt_bn_finalize_sums_nb(ws, nb, g.Vout, ly.cout, mean, var, stream);
          if (!rc)
            rc = lt_bn_apply_devox(co, mean, var, ly.gamma, ly.beta, ly.eps, ptr(d.in0), m->v2p_idx[d.level], m->v2p_w[d.level],
                                   y, m->n_points, ly.cout, m->n_vox[d.level], stream);
          break;
        }
        const float* res = op.in1 >= 0 ? ptr(op.in1) : nullptr;
        if (op.map == LT_NET_MAP_STEM) {
          rc = lt_spconv_stem_fwd(x, ly.w_fwd, g.plan, co, g.Vin, g.Vout, ly.cin, ly.cout, ly.K, stream);
          if (!rc) rc = lt_bn_stats(co, g.Vout, ly.cout, ws, L.ws_bytes, mean, var, stream);
          if (!rc) rc = lt_bn_apply(co, mean, var, ly.gamma, ly.beta, ly.eps, res, op.relu, y, g.Vout, ly.cout, stream);
        } else {
          rc = lt_conv_bn_fwd(x, ly.w_fwd, g.plan, g.Vin, g.Vout, ly.cin, ly.cout, ly.K, ly.gamma, ly.beta, ly.eps, res,
                              op.relu, ws, L.ws_bytes, co, y, mean, var, stream);
        }
        break;
      }
    }
    if (rc) return rc;
  }
  if (buf_off)
    for (int b = 0; b < p->n_bufs; ++b) buf_off[b] = (ext && ext[b]) ? -1 : L.off[b];
  return LT_OK;
}

extern "C" int64_t lt_net_grad_arena_floats(const lt_net_prog_t* prog, const lt_net_maps_t* maps, const float* const* ext) {
  (void)ext;
  if (validate(prog, maps)) return -1;
  BwdLayout L;
  if (bwd_layout(prog, maps, &L)) return -1;
  return L.total;
}

extern "C" int lt_net_backward(const lt_net_prog_t* p, const lt_net_maps_t* m, const float* const* ext, const float* arena,
                               int64_t arena_floats, const float* gout, float* dparams, float* garena,
                               int64_t garena_floats, hipStream_t stream) {
  int rc = validate(p, m);
  if (rc) return rc;
  FwdLayout F;
  rc = fwd_layout(p, m, ext, true, &F);
  if (rc) return rc;
  BwdLayout B;
  rc = bwd_layout(p, m, &B);
  if (rc) return rc;
  LT_REQUIRE(arena && arena_floats >= F.total, LT_ERR_WORKSPACE, "lt_net_backward: the arena is not the one of a keep = 1 forward");
  LT_REQUIRE(gout && dparams, LT_ERR_ARG, "lt_net_backward: null pointer");
  LT_REQUIRE(garena && ((uintptr_t)garena & 255) == 0 && garena_floats >= B.total, LT_ERR_WORKSPACE,
             "lt_net_backward: the gradient arena needs %lld floats (256-byte aligned), got %lld", (long long)B.total,
             (long long)garena_floats);
  const std::vector<char> need = grad_needed(p);
  const std::vector<int> readers = reader_count(p);
  std::vector<char> written(p->n_bufs, 0);
  std::vector<const float*> gext(p->n_bufs, nullptr);  // a gradient that is the caller's gout itself (never written)
  void* ws = garena + B.ws;
  auto act = [&](int b) -> const float* { return (ext && ext[b]) ? ext[b] : arena + F.off[b]; };
  // the gradient buffers are handed out by replaying bwd_layout's allocation decisions: same pool, same order
  Pool pool;
  std::vector<int64_t> goff(p->n_bufs, -1);
  int64_t tmp_off[3];
  {
    int64_t big = 64;
    for (int b = 0; b < p->n_bufs; ++b)
      if (buf_floats(p, m, b) > big) big = buf_floats(p, m, b);
    for (int t = 0; t < 3; ++t) tmp_off[t] = pool.take(big);
  }
  float* tmp[3] = {garena + tmp_off[0], garena + tmp_off[1], garena + tmp_off[2]};
  auto gptr = [&](int b) -> float* { return garena + goff[b]; };
  for (int i = p->n_ops - 1; i >= 0; --i) {
    const lt_net_op_t& op = p->ops[i];
    const bool steal = addend_takes_dy(op, need, readers);
    const int ins[2] = {op.in0, op.in1 == op.in0 ? -1 : op.in1};
    for (int b : ins)
      if (b >= 0 && need[b] && goff[b] < 0 && !(steal && b == op.in1)) goff[b] = pool.take(buf_floats(p, m, b));
    const bool is_out = op.out == p->out_buf;
    const float* dy = is_out ? gout : (written[op.out] ? (gext[op.out] ? gext[op.out] : gptr(op.out)) : nullptr);
    const int C = p->bufs[op.out].channels;
    const int64_t out_rows = rows_of(m, p->bufs[op.out].rows);
    if (steal) {  // the same hand-over bwd_layout planned: the addend's gradient is the output's buffer from here on
      if (is_out) {
        gext[op.in1] = gout;
      } else {
        gext[op.in1] = gext[op.out];
        goff[op.in1] = goff[op.out];
        goff[op.out] = -1;
      }
      if (dy) written[op.in1] = 1;
    }
    if (dy) {  // no gradient reached this buffer: nothing to propagate (and its parameters get none -- see below)
      switch (op.kind) {
        case LT_NET_VOX:
          if (need[op.in0]) {
            LT_REQUIRE(m->p2v_idx[op.level] && m->p2v_counts[op.level], LT_ERR_ARG,
                       "lt_net_backward: no point->voxel index / counts on level %d", op.level);
            // a second contribution is added in place by the gather itself (the point tensors are the largest of the pass:
            // a temporary and an add would move 3 x 4NC bytes more)
            rc = lt_voxelize_bwd_acc(dy, m->p2v_idx[op.level], m->p2v_counts[op.level], gptr(op.in0), m->n_points, C,
                                     m->n_vox[op.level], (int)written[op.in0], stream);
            written[op.in0] = 1;
          }
          break;
        case LT_NET_DEVOX:
          if (need[op.in0]) {
            LT_REQUIRE(m->v2p_cells[op.level] && m->p2v_perm[op.level] && m->p2v_off[op.level] && m->p2v_idx[op.level],
                       LT_ERR_ARG, "lt_net_backward: no cell table / point->voxel CSR for the voxel->point map on level %d",
                       op.level);
            float* dst = written[op.in0] ? tmp[0] : gptr(op.in0);
            rc = lt_devoxelize_cell_bwd(dy, m->v2p_cells[op.level], m->p2v_perm[op.level], m->p2v_off[op.level],
                                        m->p2v_idx[op.level], m->v2p_idx[op.level], m->v2p_w[op.level], dst, m->n_points,
                                        m->n_vox[op.level], C, stream);
            if (!rc && written[op.in0]) rc = launch_add(gptr(op.in0), dst, m->n_vox[op.level] * (int64_t)C, stream);
            written[op.in0] = 1;
          }
          break;
        case LT_NET_CAT: {
          const int ca4 = p->bufs[op.in0].channels / 4, cb4 = p->bufs[op.in1].channels / 4;
          float* ga = need[op.in0] ? gptr(op.in0) : nullptr;
          float* gb = need[op.in1] ? gptr(op.in1) : nullptr;
          if (ga || gb) {
            hipLaunchKernelGGL(k_split2, dim3(lt_grid(out_rows * (ca4 + cb4), 256)), dim3(256), 0, stream, (const float4*)dy,
                               (float4*)ga, (float4*)gb, out_rows, ca4, cb4, (int)written[op.in0], (int)written[op.in1]);
            LT_LAUNCH_CHECK();
            if (ga) written[op.in0] = 1;
            if (gb) written[op.in1] = 1;
          }
          break;
        }
        case LT_NET_CONV_BN: {
          const lt_net_layer_t& ly = p->layers[op.layer];
          ConvGeo g;
          rc = conv_geo(op, m, &g);
          if (rc) return rc;
          const float* co = arena + F.aux[i];
          const float* mean = arena + F.stat[i];
          const float* var = mean + ly.cout;
          // ReLU mask: from the saved output when a residual went in before the activation, otherwise rebuilt from the conv
          // output (lt_bn_bwd_reduce); relu 3: the addend bypasses the activation, its gradient is dy itself
          const int mode = op.relu == 0 ? 0 : (op.relu == 1 && op.in1 >= 0) ? 1 : 2;
          const float* ysaved = mode == 1 ? act(op.out) : nullptr;
          float* dgamma = dparams + ly.dgamma_off;
          float* dbeta = dparams + ly.dbeta_off;
          const bool need_din = need[op.in0] && op.map != LT_NET_MAP_STEM;
          const bool need_dres = op.in1 >= 0 && op.relu != 3 && need[op.in1];
          float* din = need_din ? (written[op.in0] ? tmp[1] : gptr(op.in0)) : nullptr;
          float* dres = need_dres ? (written[op.in1] ? tmp[2] : gptr(op.in1)) : nullptr;
          if (!need_din && !need_dres) {
            rc = lt_bn_bwd_reduce(dy, co, ysaved, mean, var, ly.gamma, ly.beta, ly.eps, mode, g.Vout, ly.cout, ws, B.ws_bytes,
                                  dbeta, dgamma, stream);
          } else {
            LT_REQUIRE(!need_din || ly.w_bwd, LT_ERR_ARG, "lt_net_backward: layer %d has no backward-input weight image", op.layer);
            rc = lt_conv_bn_bwd(dy, co, ysaved, mean, var, ly.gamma, ly.beta, ly.eps, mode, g.Vout, ly.cout, ws, B.ws_bytes,
                                dgamma, dbeta, dres, need_din ? tmp[0] : nullptr, ly.w_bwd, g.plan_bwd, g.Vin, ly.cin, ly.K, din,
                                stream);
          }
          if (rc) return rc;
          if (need_din) {
            if (written[op.in0]) rc = launch_add(gptr(op.in0), din, g.Vin * (int64_t)ly.cin, stream);
            written[op.in0] = 1;
          }
          if (!rc && need_dres) {
            if (written[op.in1]) rc = launch_add(gptr(op.in1), dres, g.Vout * (int64_t)ly.cout, stream);
            written[op.in1] = 1;
          }
          if (!rc && op.relu == 3 && need[op.in1] && !steal) {
            if (written[op.in1]) {
              rc = launch_add(gptr(op.in1), dy, g.Vout * (int64_t)ly.cout, stream);
            } else {
              LT_CHECK_HIP(hipMemcpyAsync(gptr(op.in1), dy, (size_t)g.Vout * ly.cout * 4, hipMemcpyDeviceToDevice, stream));
            }
            written[op.in1] = 1;
          }
          break;
        }
      }
      if (rc) return rc;
    } else if (op.kind == LT_NET_CONV_BN) {
      const lt_net_layer_t& ly = p->layers[op.layer];
      LT_CHECK_HIP(hipMemsetAsync(dparams + ly.dgamma_off, 0, (size_t)ly.cout * 4, stream));
      LT_CHECK_HIP(hipMemsetAsync(dparams + ly.dbeta_off, 0, (size_t)ly.cout * 4, stream));
    }
    if (!is_out && goff[op.out] >= 0) {
      pool.give(goff[op.out], buf_floats(p, m, op.out));
      goff[op.out] = -1;
    }
  }
  LT_REQUIRE(pool.top <= B.ws, LT_ERR_WORKSPACE, "lt_net_backward: gradient layout overran its plan");
  return LT_OK;
}
