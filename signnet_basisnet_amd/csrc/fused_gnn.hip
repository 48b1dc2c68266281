// fused_gnn.hip — the GINE network that consumes the positional encoding, whole stack in ONE launch.
// Replaces (eval mode) the tail of SetTransformer.forward — `self.out` Linear+BatchNorm on the slot sum
// (sign_net.py:71) — and GNN.forward (Alchemy/sign_net/model.py:36-64, GINESignNetPyG/core/model.py:44-79):
// input encoder, Linear(cat[x, pos]), nl_gnn x [edge encoder, GINEConv (pyg_gnn_wrapper.py:19-28),
// BatchNorm, ReLU, residual], add-pooling over each graph and the 2-layer output encoder.
//
// This stage has few rows (one per node, no eigenvector-slot factor) and a long dependent chain
// (3 + 2*nl_gnn + 2 Linear layers), so it is latency-bound, not throughput-bound.  Mapping: ONE workgroup of
// 8 waves per graph (n <= 64 nodes).  The graph's node rows live in LDS images [64][d_pad+4]; for every
// Linear, wave (row tile rt, group grp) reads its 16 rows from the input image into the MFMA operand layout
// and computes only its share of the output tiles (the output channels are split over the 8/T waves that
// share a row tile, T = ceil(n/16)), then writes them to the output image — one barrier per Linear.  That
// spreads one graph's chain over all 4 SIMDs of a CU instead of one wave.  The GINE neighbour sum
// relu(h_j + e_ji) and the pooling read the same images.
#include "fused_common.hpp"

namespace sn {

#ifdef SN_PROFILE
static __device__ int g_prof_block = 0;
__shared__ unsigned long long s_prof[16];      // per-section cycle sums of thread 0 of the selected block (LDS: cheap to update)
#undef SN_STAMP
#undef SN_ACCUM
#define SN_STAMP(i) do { if ((int)blockIdx.x == g_prof_block && threadIdx.x == 0) g_prof[i] = clock64(); } while (0)
#define SN_ACCUM(i, t0) do { if ((int)blockIdx.x == g_prof_block && threadIdx.x == 0) g_prof[i] += clock64() - (t0); } while (0)
#define SN_T0() const long long sn_t0 = clock64()
#define SN_LACC(i) do { if (threadIdx.x == 0) atomicAdd(&s_prof[i], (unsigned long long)(clock64() - sn_t0)); } while (0)
#else
#define SN_T0() do { } while (0)
#define SN_LACC(i) do { } while (0)
#endif


constexpr int GNN_ROWS = SN_GNN_MAX_NODES;   // 64
constexpr int GNN_WAVES = 8;
constexpr int GNN_EMAX = 192;                // in-edges of one graph staged in LDS
constexpr int GNN_CLS = 16;                  // edge-feature classes per graph whose embeddings stay in LDS for all layers
constexpr int GNN_EEMAX = 96;                // ... of which this many can have their layer embeddings staged too
constexpr int GNN_EEPF = (GNN_EEMAX * 32 + GNN_WAVES * 64 - 1) / (GNN_WAVES * 64);   // float4 per thread (d_pad = 128)

struct GnnStruct {
  const void* x;          // int64 [N, ldx] (discrete) or float [N, F]
  int ldx;
  const void* edge_attr;  // int64 [E, lde] (discrete) or float [E, F_e]
  int lde;
  const float* rho_sum;   // [N, d]
  const int32_t* graph_ptr;
  const int32_t* rowptr;
  const int32_t* col;
  const int32_t* eperm;
  int32_t* status;        // status[3] |= 1 if a graph has more than 64 nodes (host falls back)
  float* y;               // [B, n_out]
  int ee_rows;            // edges whose per-layer embeddings fit the LDS staging area
};

// Weight fragments of ONE output tile of a packed matrix, held in registers.
template <int NTI>
struct WTile { float4 w[NTI]; };

template <int NTI>
__device__ __forceinline__ void wload(WTile<NTI>& p, const float* __restrict__ wp, int nto, int ot, int lane) {
  const __amdgpu_buffer_rsrc_t rs = weight_rsrc(wp, (unsigned)nto * NTI * 1024);
  const int voff = lane * 16;
  const int base = __builtin_amdgcn_readfirstlane(ot * NTI * 1024);
#pragma unroll
  for (int kk = 0; kk < NTI; ++kk) {
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, base + kk * 1024, 0);
    p.w[kk] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
  }
}

// The (output tile, row tile) pairs a wave owns in one Linear: a contiguous range [t_lo, t_hi) of the flattened index
// t = ot*T + rt (T = row tiles of the graph, <= 4) — OUTPUT-TILE MAJOR, so that a wave works on one output tile (two
// at most) for all the graph's row tiles and every weight fragment is fetched by exactly one wave of the workgroup:
// the weight traffic of a Linear is the matrix once per CU (it was once per row tile, and the time to issue those
// loads against the 64 B/clk/CU delivery rate was as long as the MFMAs).
struct TileRange {
  int t_lo, t_hi, T;
  __device__ __forceinline__ bool empty() const { return t_lo >= t_hi; }
  __device__ __forceinline__ void decode(int t, int& ot, int& rt) const { ot = t / T; rt = t - ot * T; }
  __device__ __forceinline__ int first_ot() const { return t_lo / T; }
};

template <int NT>
__device__ __forceinline__ f32x4 mfma_tile32(const WTile<NT>& w, const f32x4 (&in)[NT]) {
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kk = 0; kk < NT; ++kk) {
    if (kk & 1) {
      a1 = mfma16(w.w[kk].x, in[kk][0], a1);
      a1 = mfma16(w.w[kk].y, in[kk][1], a1);
      a1 = mfma16(w.w[kk].z, in[kk][2], a1);
      a1 = mfma16(w.w[kk].w, in[kk][3], a1);
    } else {
      a0 = mfma16(w.w[kk].x, in[kk][0], a0);
      a0 = mfma16(w.w[kk].y, in[kk][1], a0);
      a0 = mfma16(w.w[kk].z, in[kk][2], a0);
      a0 = mfma16(w.w[kk].w, in[kk][3], a0);
    }
  }
  return a0 + a1;
}

// One Linear over the workgroup's rows.  The wave's TileRange is <= 4 (ot, rt) pairs touching <= 2 output tiles.
// Weight fragments ping-pong between `pre` and `alt`: `pre` holds the first output tile's fragments on entry (fetched
// during the previous stage); a second output tile — or, at the end, the NEXT Linear's first tile — is fetched while
// the current one computes.  The row-tile operand is double buffered from LDS.  img: input image [64][LD];
// epi(rt, ot, acc, e0, e1) with e* = ev*[16 ot + 4g ..] (per-channel epilogue vectors, fetched before the MFMAs).
template <int NT, typename Epi>
__device__ __forceinline__ void coop_gemm(WTile<NT>& pre, WTile<NT>& alt, const float* __restrict__ wp, int nto,
                                          const float* img, int LD, TileRange tr, int lane, Epi epi,
                                          const float* __restrict__ next_wp, int next_nto, TileRange next_tr,
                                          const float* __restrict__ ev0 = nullptr, const float* __restrict__ ev1 = nullptr) {
  const int g = lane >> 4;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  if (tr.empty()) {
    if (next_wp && !next_tr.empty()) wload<NT>(pre, next_wp, next_nto, next_tr.first_ot(), lane);   // idle here: keep the chain going
    return;
  }
  int ot0, rt0;
  tr.decode(tr.t_lo, ot0, rt0);
  int otl, rtl;
  tr.decode(tr.t_hi - 1, otl, rtl);
  const bool two_ots = otl != ot0;
  const float* rowbase = img + (lane & 15) * LD + 4 * g;
  f32x4 inA[NT], inB[NT];
#pragma unroll
  for (int kk = 0; kk < NT; ++kk) inA[kk] = lds_ld4(rowbase + rt0 * 16 * LD + 16 * kk);
  // prefetch: second output tile of this Linear, else the next Linear's first tile
  bool next_in_alt = false;
  if (two_ots) wload<NT>(alt, wp, nto, otl, lane);
  else if (next_wp && !next_tr.empty()) { wload<NT>(alt, next_wp, next_nto, next_tr.first_ot(), lane); next_in_alt = true; }
  f32x4 e0 = ev0 ? ld4(ev0 + 16 * ot0 + 4 * g) : z4, e1 = ev1 ? ld4(ev1 + 16 * ot0 + 4 * g) : z4;
  f32x4 f0 = z4, f1 = z4;
  if (two_ots) { f0 = ev0 ? ld4(ev0 + 16 * otl + 4 * g) : z4; f1 = ev1 ? ld4(ev1 + 16 * otl + 4 * g) : z4; }
  bool cur_is_pre = true;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t = tr.t_lo + i;
    if (t < tr.t_hi) {
      int ot, rt;
      tr.decode(t, ot, rt);
      const bool second = ot != ot0;
      if (second && cur_is_pre) {
        cur_is_pre = false;                                   // switch to the second tile (in alt); pre is free again:
        if (next_wp && !next_tr.empty()) wload<NT>(pre, next_wp, next_nto, next_tr.first_ot(), lane);   // next Linear's first tile
      }
      // operand of the NEXT pair (double buffer): its LDS latency hides behind this pair's MFMAs
      f32x4 (&cur)[NT] = (i & 1) ? inB : inA;
      f32x4 (&nxt)[NT] = (i & 1) ? inA : inB;
      if (t + 1 < tr.t_hi) {
        int ot2, rt2;
        tr.decode(t + 1, ot2, rt2);
#pragma unroll
        for (int kk = 0; kk < NT; ++kk) nxt[kk] = lds_ld4(rowbase + rt2 * 16 * LD + 16 * kk);
      }
      const f32x4 acc = cur_is_pre ? mfma_tile32<NT>(pre, cur) : mfma_tile32<NT>(alt, cur);
      epi(rt, ot, acc, second ? f0 : e0, second ? f1 : e1);
    }
  }
  if (next_in_alt) {   // single-tile range: the next Linear's first tile sits in alt — hand it over in pre
#pragma unroll
    for (int kk = 0; kk < NT; ++kk) pre.w[kk] = alt.w[kk];
  }
}

template <int NT>
__global__ __launch_bounds__(GNN_WAVES * 64, 2) void k_gnn_coop(GnnStruct S, sn_gnn_params P) {
  constexpr int D = 16 * NT;
  constexpr int LD = D + 4;
  extern __shared__ __align__(16) float lds[];
  float* X0 = lds;                    // [64][LD]
  float* X1 = lds + GNN_ROWS * LD;
  float* X2 = lds + 2 * GNN_ROWS * LD;
  int* erow = reinterpret_cast<int*>(lds + 3 * GNN_ROWS * LD);   // [65]  CSR row pointers local to the graph
  int* esrc = erow + GNN_ROWS + 4;                               // [GNN_EMAX] local source row of every in-edge
  int* efeat = esrc + GNN_EMAX;                                  // [GNN_EMAX][edge_nf] feature words (int idx / float)
  int* ecls = efeat + GNN_EMAX * (P.n_layers > 0 ? P.edge_nf : 0);   // [GNN_EMAX] feature class of every in-edge
  int* elead = ecls + GNN_EMAX;                                     // [GNN_EMAX] first edge with the same features (scratch)
  int* cedge = elead + GNN_EMAX;                                    // [GNN_CLS]  representative edge of every class
  float* EE = reinterpret_cast<float*>(cedge + GNN_CLS);           // [ee_rows][LD] edge embeddings (per class x layer, or per edge)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, li = lane & 15;
  const int gi = blockIdx.x;
  SN_STAMP(0);
#ifdef SN_PROFILE
  if (threadIdx.x < 16) s_prof[threadIdx.x] = 0;
#endif
#ifdef SN_PROFILE
  if ((int)blockIdx.x == g_prof_block && threadIdx.x == 0) for (int i = 8; i < 20; ++i) g_prof[i] = 0;
  long long pt = 0;
#endif
  const int gs = S.graph_ptr[gi], n = S.graph_ptr[gi + 1] - gs;
  if (n <= 0) return;
  if (n > GNN_ROWS) {
    if (threadIdx.x == 0) atomicOr(&S.status[3], 1);
    return;
  }
  const int e_base = S.rowptr[gs];
  const int ne = S.rowptr[gs + n] - e_base;
  if (ne > GNN_EMAX) {
    if (threadIdx.x == 0) atomicOr(&S.status[3], 2);
    return;
  }
  const int d = P.d;
  const int T = (n + 15) >> 4;                               // row tiles (1..4)
  const int ntile = T * NT;
  const int q = (ntile + GNN_WAVES - 1) / GNN_WAVES;          // pairs per wave (<= T since NT <= 8: at most 2 output tiles)
  TileRange tr;                                               // my share of every node-row Linear (output-tile major)
  tr.T = T;
  tr.t_lo = wave * q < ntile ? wave * q : ntile;
  tr.t_hi = tr.t_lo + q < ntile ? tr.t_lo + q : ntile;
  TileRange hr;                                               // my share of the output encoder (one pooled row tile)
  hr.T = 1;
  hr.t_lo = wave < NT ? wave : NT;
  hr.t_hi = wave < NT ? wave + 1 : NT;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  // Edges of a graph repeat a handful of feature tuples (ZINC: 3 bond types), and an edge's embedding depends on
  // nothing else.  The staging code below groups the graph's edges into classes of identical features; with at
  // most GNN_CLS classes the embeddings of every (layer, class) are built ONCE into LDS (use_tab) and the aggregation
  // reads edge e's embedding as row ecls[e] — no per-edge, per-layer gather.  Otherwise the current layer's
  // per-edge embeddings are staged (use_ee), or gathered directly.
  bool use_tab = false;   // decided after the classes are known
  int ncls = 0;
  bool use_ee = P.n_layers > 0 && ne <= S.ee_rows;
  // embedding of edge k, channels [c, c+4) for layer Lq: DiscreteEncoder sum (elements.py:31-37) or MLP(F_e, d, 1)
  auto edge_embed = [&](const sn_gnn_layer& Lq, int k, int c) -> f32x4 {
    const int EF = P.edge_nf;
    f32x4 ef = zero4;
    if (P.edge_discrete) {
      for (int f = 0; f < EF; ++f) {
        const float* trow = Lq.etab[f] + (int64_t)efeat[k * EF + f] * d;
        if ((d & 3) == 0) { if (c < d) ef += ld4(trow + c); }
        else {
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) if (c + qq < d) ef[qq] += trow[c + qq];
        }
      }
    } else {
      f32x4 acc = zero4;
      for (int f = 0; f < EF; ++f) {
        const float a = __int_as_float(efeat[k * EF + f]);
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) acc[qq] += a * Lq.ew[(c + qq) * EF + f];
      }
      ef = relu4(acc * ld4(Lq.e_scale + c) + ld4(Lq.e_shift + c));
    }
    return ef;
  };
  // one layer's embeddings of all the graph's edges: fetched into registers early, parked in LDS between barriers
  f32x4 eepf[GNN_EEPF];
  auto ee_fetch = [&](int l) {
    if (!use_ee || l >= P.n_layers) return;
    const sn_gnn_layer& Lq = P.layers[l];
#pragma unroll
    for (int i = 0; i < GNN_EEPF; ++i) {
      const int idx = threadIdx.x + i * GNN_WAVES * 64;
      eepf[i] = zero4;
      if (idx < ne * (D / 4)) eepf[i] = edge_embed(Lq, idx / (D / 4), 4 * (idx % (D / 4)));
    }
  };
  auto ee_store = [&]() {
    if (!use_ee) return;
#pragma unroll
    for (int i = 0; i < GNN_EEPF; ++i) {
      const int idx = threadIdx.x + i * GNN_WAVES * 64;
      if (idx < ne * (D / 4)) lds_st4(EE + (idx / (D / 4)) * LD + 4 * (idx % (D / 4)), eepf[i]);
    }
  };

  WTile<NT> pre, alt;
  if (!tr.empty()) wload<NT>(pre, P.rho_out_w, NT, tr.first_ot(), lane);       // in flight while the inputs are staged
  // ---------------------------------------------------------------- per-graph CSR + edge data -> LDS (once)
  {
    const int EF = P.edge_nf;
    for (int k = threadIdx.x; k <= n; k += GNN_WAVES * 64) erow[k] = S.rowptr[gs + k] - e_base;
    for (int k = threadIdx.x; k < ne; k += GNN_WAVES * 64) {
      esrc[k] = S.col[e_base + k] - gs;
      const int eid = S.eperm[e_base + k];
      if (P.n_layers > 0) {
        if (P.edge_discrete) {
          const int64_t* ei = reinterpret_cast<const int64_t*>(S.edge_attr) + (int64_t)eid * S.lde;
          for (int f = 0; f < EF; ++f) efeat[k * EF + f] = (int)ei[f];
        } else {
          const float* ea = reinterpret_cast<const float*>(S.edge_attr) + (int64_t)eid * S.lde;
          for (int f = 0; f < EF; ++f) efeat[k * EF + f] = __float_as_int(ea[f]);
        }
      }
    }
  }
  // ---------------------------------------------------------------- edge-feature classes (see use_tab above)
  if (P.n_layers > 0) {
    const int EF = P.edge_nf;
    __syncthreads();                                   // efeat is complete
    int lead = -1;
    if ((int)threadIdx.x < ne) {
      const int k = threadIdx.x;
      lead = k;                                        // first edge with my feature tuple; with a handful of classes the
      for (int j = 0; j < k; ++j) {                    // scan stops within the first few edges
        bool same = true;
        for (int f = 0; f < EF; ++f) same = same && (efeat[j * EF + f] == efeat[k * EF + f]);
        if (same) { lead = j; break; }
      }
      elead[k] = lead;
    }
    ncls = __syncthreads_count((int)threadIdx.x < ne && lead == (int)threadIdx.x);   // leaders = classes (ne <= 192 < blockDim)
    if ((int)threadIdx.x < ne) {
      int c = 0;
      for (int j = 0; j < lead; ++j) c += (elead[j] == j);
      ecls[threadIdx.x] = c;                           // dense class id = number of leaders before my leader
      if (lead == (int)threadIdx.x && c < GNN_CLS) cedge[c] = lead;
    }
    use_tab = ncls <= GNN_CLS && P.n_layers * ncls <= S.ee_rows;
    if (use_tab) use_ee = false;
    __syncthreads();
    if (use_tab) {   // EE[l * ncls + c][:] = embedding of class c's representative edge in layer l (padded channels: 0)
      for (int i = threadIdx.x; i < P.n_layers * ncls * (D / 4); i += GNN_WAVES * 64) {
        const int rowi = i / (D / 4), ch = 4 * (i % (D / 4));
        const int l = rowi / ncls, c = rowi - l * ncls;
        lds_st4(EE + rowi * LD + ch, edge_embed(P.layers[l], cedge[c], ch));
      }
    }
  }
  // ---------------------------------------------------------------- stage the slot sum (rho output) in X1 (zero padded to D)
  if ((d & 3) == 0) {
    for (int i = threadIdx.x; i < n * (D / 4); i += GNN_WAVES * 64) {
      const int rr = i / (D / 4), c = 4 * (i % (D / 4));
      lds_st4(X1 + rr * LD + c, c < d ? ld4(S.rho_sum + (int64_t)(gs + rr) * d + c) : zero4);
    }
  } else {
    for (int i = threadIdx.x; i < n * D; i += GNN_WAVES * 64) {
      const int rr = i / D, c = i % D;
      X1[rr * LD + c] = c < d ? S.rho_sum[(int64_t)(gs + rr) * d + c] : 0.f;
    }
  }
  // ---------------------------------------------------------------- input encoder -> X0 (model.py:37)
  if (P.node_discrete) {
    for (int t = tr.t_lo; t < tr.t_hi; ++t) {
      int ot, rt;
      tr.decode(t, ot, rt);
      const int row = rt * 16 + li, c = 16 * ot + 4 * g;
      if (row < n) {
        const int64_t* xi = reinterpret_cast<const int64_t*>(S.x) + (int64_t)(gs + row) * S.ldx;
        f32x4 s = zero4;
        for (int f = 0; f < P.node_nf; ++f) {
          const float* trow = P.ntab[f] + xi[f] * d;
          if ((d & 3) == 0) { if (c < d) s += ld4(trow + c); }
          else {
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) if (c + qq < d) s[qq] += trow[c + qq];
          }
        }
        lds_st4(X0 + row * LD + c, s);
      }
    }
  } else {
    // MLP(nfeat, d, 1): Linear(no bias) . BN . ReLU on <= 16 continuous features — VALU, one output tile at a time
    for (int t = tr.t_lo; t < tr.t_hi; ++t) {
      int ot, rt;
      tr.decode(t, ot, rt);
      const int row = rt * 16 + li, c = 16 * ot + 4 * g;
      if (row < n) {
        const float* xr = reinterpret_cast<const float*>(S.x) + (int64_t)(gs + row) * S.ldx;
        f32x4 acc = zero4;
        for (int f = 0; f < P.node_nf; ++f) {
          const float a = xr[f];
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) acc[qq] += a * P.nw[(c + qq) * P.node_nf + f];   // nw: [d_pad, F] row-major
        }
        lds_st4(X0 + row * LD + c, relu4(acc * ld4(P.n_scale + c) + ld4(P.n_shift + c)));
      }
    }
  }
  __syncthreads();
  SN_STAMP(1);
  ee_fetch(0);     // needs efeat (staged above); the loads fly during the three Linears below
  // ---------------------------------------------------------------- pos = BN(W_out . slot_sum): X1 -> X2   (sign_net.py:71)
  coop_gemm<NT>(pre, alt, P.rho_out_w, NT, X1, LD, tr, lane, [&](int rt, int ot, f32x4 acc, f32x4 sc, f32x4 sh) {
    lds_st4(X2 + (rt * 16 + li) * LD + 16 * ot + 4 * g, acc * sc + sh);
  }, P.lin_a, NT, tr, P.rho_scale, P.rho_shift);
  __syncthreads();
  SN_STAMP(20);
  // ---------------------------------------------------------------- h = Linear(cat[x, pos]): X0, X2 -> X1    (model.py:39-40)
  coop_gemm<NT>(pre, alt, P.lin_a, NT, X0, LD, tr, lane, [&](int rt, int ot, f32x4 acc, f32x4, f32x4) {
    lds_st4(X1 + (rt * 16 + li) * LD + 16 * ot + 4 * g, acc);      // readers of X1 (slot sum) passed a barrier
  }, P.lin_b, NT, tr);
  coop_gemm<NT>(pre, alt, P.lin_b, NT, X2, LD, tr, lane, [&](int rt, int ot, f32x4 acc, f32x4 bias, f32x4) {
    float* o = X1 + (rt * 16 + li) * LD + 16 * ot + 4 * g;
    lds_st4(o, (lds_ld4(o) + acc) + bias);
  }, P.n_layers > 0 ? P.layers[0].w1p : P.head_w1, NT, P.n_layers > 0 ? tr : hr, P.lin_bias);
  ee_store();
  __syncthreads();
  SN_STAMP(2);
  // ---------------------------------------------------------------- GINE layers: h lives in X1           (model.py:47-55)
  for (int l = 0; l < P.n_layers; ++l) {
    const sn_gnn_layer& Lp = P.layers[l];
    // u = sum_{j->i} relu(h_j + e_ji) + (1+eps) h_i  for my (row tile, channel tile) pairs: X1 -> X2
#ifdef SN_PROFILE
    pt = clock64();
#endif
    ee_fetch(l + 1);   // next layer's edge embeddings: in flight during this aggregation
    SN_ACCUM(8, pt);
#ifdef SN_PROFILE
    pt = clock64();
#endif
    {
      const float sc = 1.f + *Lp.eps;
#pragma unroll 1
      for (int t = tr.t_lo; t < tr.t_hi; ++t) {
        int ot, rt;
        tr.decode(t, ot, rt);
        const int row = rt * 16 + li, c = 16 * ot + 4 * g;
        if (row >= n) continue;
        f32x4 u = zero4;
        const int e_lo = erow[row], e_hi = erow[row + 1];
        int e = e_lo;
        if (use_tab || use_ee) {
          // the first four in-edges (molecular graphs: all) with predicated, unrolled reads: index reads, then the
          // eight row reads, then the adds in edge order (a missing edge adds +0)
          const int dg = e_hi - e_lo;
          int sr[4], er[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int ei = i < dg ? e_lo + i : 0;
            sr[i] = i < dg ? esrc[ei] : row;
            er[i] = i < dg ? (use_tab ? l * ncls + ecls[ei] : ei) : 0;
          }
          f32x4 hv[4], ev[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) { hv[i] = lds_ld4(X1 + sr[i] * LD + c); ev[i] = lds_ld4(EE + er[i] * LD + c); }
#pragma unroll
          for (int i = 0; i < 4; ++i) u += i < dg ? relu4(hv[i] + ev[i]) : zero4;
          e = e_lo + (dg < 4 ? dg : 4);
        }
        for (; e < e_hi; ++e) {
          const f32x4 ef = use_tab ? lds_ld4(EE + (l * ncls + ecls[e]) * LD + c)
                                   : (use_ee ? lds_ld4(EE + e * LD + c) : edge_embed(Lp, e, c));
          u += relu4(lds_ld4(X1 + esrc[e] * LD + c) + ef);
        }
        {
#pragma clang fp contract(off)
          const f32x4 self = lds_ld4(X1 + row * LD + c) * sc;
          u = u + self;
        }
        lds_st4(X2 + row * LD + c, u);
      }
    }
    __syncthreads();
    SN_ACCUM(9, pt);
#ifdef SN_PROFILE
    pt = clock64();
#endif
    ee_store();        // every wave is done reading this layer's embeddings
    SN_ACCUM(10, pt);
#ifdef SN_PROFILE
    pt = clock64();
#endif
    // nn: Linear . BN . ReLU : X2 -> X0
    coop_gemm<NT>(pre, alt, Lp.w1p, NT, X2, LD, tr, lane, [&](int rt, int ot, f32x4 acc, f32x4 sc, f32x4 sh) {
      lds_st4(X0 + (rt * 16 + li) * LD + 16 * ot + 4 * g, relu4(acc * sc + sh));
    }, Lp.w2p, NT, tr, Lp.bn0_scale, Lp.bn0_shift);
    __syncthreads();
    SN_ACCUM(11, pt);
#ifdef SN_PROFILE
    pt = clock64();
#endif
    // Linear ; BN . ReLU ; + previous_x : X0 -> X1 (my tiles only: nobody else reads them at this point)
    const bool lastl = l + 1 == P.n_layers;
    coop_gemm<NT>(pre, alt, Lp.w2p, NT, X0, LD, tr, lane, [&](int rt, int ot, f32x4 acc, f32x4 sc, f32x4 sh) {
      float* o = X1 + (rt * 16 + li) * LD + 16 * ot + 4 * g;
      lds_st4(o, relu4(acc * sc + sh) + lds_ld4(o));
    }, lastl ? P.head_w1 : P.layers[lastl ? l : l + 1].w1p, NT, lastl ? hr : tr, Lp.bn_scale, Lp.bn_shift);
    __syncthreads();
    SN_ACCUM(12, pt);
  }
  SN_STAMP(3);
  // ---------------------------------------------------------------- add pooling -> row 0 of X2 (rows 1..15 zero)  (model.py:57-61)
  for (int c = threadIdx.x; c < 16 * LD; c += GNN_WAVES * 64) {
    const int rr = c / LD, cc = c - rr * LD;
    float s = 0.f;
    if (rr == 0 && cc < D)
      for (int j = 0; j < n; ++j) s += X1[j * LD + cc];
    X2[c] = s;
  }
  __syncthreads();
  SN_STAMP(4);
  // ---------------------------------------------------------------- output encoder on the pooled row     (model.py:63)
  TileRange h2;
  h2.T = 1;
  h2.t_lo = 0;
  h2.t_hi = wave == 0 ? 1 : 0;
  coop_gemm<NT>(pre, alt, P.head_w1, NT, X2, LD, hr, lane, [&](int rt, int ot, f32x4 acc, f32x4 sc, f32x4 sh) {
    lds_st4(X0 + li * LD + 16 * ot + 4 * g, relu4(acc * sc + sh));
  }, wave == 0 ? P.head_w2 : nullptr, 1, h2, P.head_scale, P.head_shift);
  __syncthreads();
  coop_gemm<NT>(pre, alt, P.head_w2, 1, X0, LD, h2, lane, [&](int rt, int ot, f32x4 acc, f32x4, f32x4) {
    if (li == 0) {
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const int c = 4 * g + qq;
        if (c < P.n_out) S.y[(int64_t)gi * P.n_out + c] = acc[qq] + P.head_b2[c];
      }
    }
  }, nullptr, 0, h2);
  SN_STAMP(5);
#ifdef SN_PROFILE
  if ((int)blockIdx.x == g_prof_block && threadIdx.x == 0) for (int i = 0; i < 16; ++i) g_prof[40 + i] = (long long)s_prof[i];
#endif
#ifdef SN_PROFILE
  if ((int)blockIdx.x == g_prof_block && threadIdx.x == 0) { g_prof[6] = n; g_prof[7] = ne; }
#endif
}

template <int NT>
static int launch_gnn(const GnnStruct& S, const sn_gnn_params& P, int64_t B, hipStream_t st) {
  constexpr int LD = 16 * NT + 4;
  const size_t base = (size_t)(3 * GNN_ROWS * LD) * sizeof(float) + (size_t)(GNN_ROWS + 4 + GNN_EMAX * (3 + (P.n_layers > 0 ? P.edge_nf : 0)) + GNN_CLS) * sizeof(int);
  const size_t lds_cap = 160 * 1024 - 512;     // the kernel also has a few bytes of static LDS (__syncthreads_count)
  const size_t room = base < lds_cap ? lds_cap - base : 0;
  int ee_rows = (int)(room / ((size_t)LD * sizeof(float)));
  if (ee_rows > GNN_EEMAX) ee_rows = GNN_EEMAX;
  GnnStruct S2 = S;
  S2.ee_rows = P.n_layers > 0 ? ee_rows : 0;
  const size_t lds = base + (size_t)S2.ee_rows * LD * sizeof(float);
  static bool init = false;
  if (!init) {
    const size_t lds_max = lds_cap;
    if (lds_max > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_gnn_coop<NT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_max) != hipSuccess)
      return fail(SN_ERR_LAUNCH, "sn_gnn_fused_f32: cannot raise the dynamic LDS limit to %zu", lds_max);
    init = true;
  }
  hipLaunchKernelGGL((k_gnn_coop<NT>), dim3((unsigned)B), dim3(GNN_WAVES * 64), lds, st, S2, P);
  return SN_OK;
}

}  // namespace sn

using namespace sn;

#ifdef SN_PROFILE
extern "C" int sn_prof_read_gnn(long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_prof), sizeof(long long) * 64); }
extern "C" int sn_prof_set_block(int b) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_prof_block), &b, sizeof(int)); }
#endif

extern "C" int sn_gnn_fused_f32(const sn_gnn_params* params, const void* x, int ldx, const void* edge_attr, int lde,
                                const float* rho_sum, const int32_t* graph_ptr, int64_t B, const int32_t* rowptr,
                                const int32_t* col, const int32_t* eperm, int32_t* status, float* y, void* stream) {
  SN_REQUIRE(params && x && rho_sum && graph_ptr && rowptr && status && y && B >= 0, "sn_gnn_fused_f32: null pointer");
  const sn_gnn_params& P = *params;
  SN_REQUIRE(P.d > 0 && P.d <= 128, "sn_gnn_fused_f32: hidden width %d not in (0, 128]", P.d);
  SN_REQUIRE(P.n_layers >= 0 && P.n_layers <= SN_GNN_MAX_LAYERS, "sn_gnn_fused_f32: %d layers unsupported", P.n_layers);
  SN_REQUIRE(P.n_out >= 1 && P.n_out <= 16, "sn_gnn_fused_f32: n_out=%d not in [1,16]", P.n_out);
  SN_REQUIRE(P.node_nf >= 1 && P.node_nf <= (P.node_discrete ? 10 : 16) && ldx >= P.node_nf,
             "sn_gnn_fused_f32: node feature count %d unsupported", P.node_nf);
  SN_REQUIRE(P.n_layers == 0 || (edge_attr && P.edge_nf >= 1 && P.edge_nf <= (P.edge_discrete ? 10 : 16) && lde >= P.edge_nf),
             "sn_gnn_fused_f32: edge feature count %d unsupported", P.edge_nf);
  SN_REQUIRE(P.rho_out_w && P.rho_scale && P.rho_shift && P.lin_a && P.lin_b && P.lin_bias && P.head_w1 && P.head_scale &&
                 P.head_shift && P.head_w2 && P.head_b2,
             "sn_gnn_fused_f32: parameters missing");
  if (P.node_discrete) { for (int f = 0; f < P.node_nf; ++f) SN_REQUIRE(P.ntab[f], "sn_gnn_fused_f32: node table %d missing", f); }
  else SN_REQUIRE(P.nw && P.n_scale && P.n_shift, "sn_gnn_fused_f32: node MLP parameters missing");
  for (int l = 0; l < P.n_layers; ++l) {
    const sn_gnn_layer& L = P.layers[l];
    SN_REQUIRE(L.w1p && L.bn0_scale && L.bn0_shift && L.w2p && L.bn_scale && L.bn_shift && L.eps,
               "sn_gnn_fused_f32: layer %d parameters missing", l);
    if (P.edge_discrete) { for (int f = 0; f < P.edge_nf; ++f) SN_REQUIRE(L.etab[f], "sn_gnn_fused_f32: layer %d edge table %d missing", l, f); }
    else SN_REQUIRE(L.ew && L.e_scale && L.e_shift, "sn_gnn_fused_f32: layer %d edge MLP parameters missing", l);
  }
  if (B == 0) return SN_OK;
  GnnStruct S{x, ldx, edge_attr, lde, rho_sum, graph_ptr, rowptr, col, eperm, status, y, 0};
  hipStream_t st = (hipStream_t)stream;
  int rc = SN_OK;
  switch ((P.d + 15) / 16) {
    case 1: rc = launch_gnn<1>(S, P, B, st); break;
    case 2: rc = launch_gnn<2>(S, P, B, st); break;
    case 3: rc = launch_gnn<3>(S, P, B, st); break;
    case 4: rc = launch_gnn<4>(S, P, B, st); break;
    case 5: rc = launch_gnn<5>(S, P, B, st); break;
    case 6: rc = launch_gnn<6>(S, P, B, st); break;
    case 7: rc = launch_gnn<7>(S, P, B, st); break;
    default: rc = launch_gnn<8>(S, P, B, st); break;
  }
  if (rc != SN_OK) return rc;
  SN_CHECK_LAUNCH("sn_gnn_fused_f32");
  return SN_OK;
}
