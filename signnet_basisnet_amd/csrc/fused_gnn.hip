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

constexpr int GNN_ROWS = SN_GNN_MAX_NODES;   // 64
constexpr int GNN_WAVES = 8;
constexpr int GNN_OTS = 4;                   // max output tiles per wave (NT=8 split over >= 2 groups)
constexpr int GNN_EMAX = 192;                // in-edges of one graph staged in LDS
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

// Weight fragments of two output tiles (ot, ot+1) of one packed matrix, held in registers.
template <int NTI>
struct WPair { float4 w0[NTI], w1[NTI]; };

template <int NTI>
__device__ __forceinline__ void wload(WPair<NTI>& p, const float* __restrict__ wp, int nto, int ot, bool one, bool two,
                                      int lane) {
  const __amdgpu_buffer_rsrc_t rs = weight_rsrc(wp, (unsigned)nto * NTI * 1024);
  const int voff = lane * 16;
  const int base = __builtin_amdgcn_readfirstlane(ot * NTI * 1024);
  if (one) {
#pragma unroll
    for (int kk = 0; kk < NTI; ++kk) {
      u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, base + kk * 1024, 0);
      p.w0[kk] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
    }
  }
  if (two) {
#pragma unroll
    for (int kk = 0; kk < NTI; ++kk) {
      u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, base + (NTI + kk) * 1024, 0);
      p.w1[kk] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
    }
  }
}

// The (row tile, output tile) pairs a wave owns in one Linear: a contiguous range [t_lo, t_hi) of the flattened
// index t = rt*NT + ot.  It touches at most two row tiles ("groups").
struct TileRange {
  int t_lo, t_hi;
  __device__ __forceinline__ bool empty() const { return t_lo >= t_hi; }
  // group k in {0,1}: row tile rt and its output tiles [o_lo, o_hi)
  __device__ __forceinline__ void group(int k, int NT, int& rt, int& o_lo, int& o_hi) const {
    const int rt0 = t_lo / NT;
    if (k == 0) {
      rt = rt0;
      o_lo = t_lo - rt0 * NT;
      o_hi = (t_hi < (rt0 + 1) * NT ? t_hi : (rt0 + 1) * NT) - rt0 * NT;
      if (empty()) o_hi = o_lo;
    } else {
      rt = rt0 + 1;
      o_lo = 0;
      o_hi = t_hi - (rt0 + 1) * NT;
      if (o_hi < 0 || empty()) o_hi = 0;
    }
  }
  __device__ __forceinline__ void first(int NT, int& ot, bool& one, bool& two) const {
    int rt, lo, hi;
    group(0, NT, rt, lo, hi);
    ot = lo;
    one = lo < hi;
    two = lo + 1 < hi;
  }
};

template <int NT>
__device__ __forceinline__ void mfma_pair(const WPair<NT>& w, const f32x4 (&in)[NT], bool two, f32x4& a0, f32x4& a1) {
  a0 = f32x4{0.f, 0.f, 0.f, 0.f};
  a1 = f32x4{0.f, 0.f, 0.f, 0.f};
  if (two) {
#pragma unroll
    for (int kk = 0; kk < NT; ++kk) {
      a0 = mfma16(w.w0[kk].x, in[kk][0], a0);
      a1 = mfma16(w.w1[kk].x, in[kk][0], a1);
      a0 = mfma16(w.w0[kk].y, in[kk][1], a0);
      a1 = mfma16(w.w1[kk].y, in[kk][1], a1);
      a0 = mfma16(w.w0[kk].z, in[kk][2], a0);
      a1 = mfma16(w.w1[kk].z, in[kk][2], a1);
      a0 = mfma16(w.w0[kk].w, in[kk][3], a0);
      a1 = mfma16(w.w1[kk].w, in[kk][3], a1);
    }
  } else {   // a lone tile: split its k-chunks over the two accumulator chains
#pragma unroll
    for (int kk = 0; kk < NT; ++kk) {
      if (kk & 1) {
        a1 = mfma16(w.w0[kk].x, in[kk][0], a1);
        a1 = mfma16(w.w0[kk].y, in[kk][1], a1);
        a1 = mfma16(w.w0[kk].z, in[kk][2], a1);
        a1 = mfma16(w.w0[kk].w, in[kk][3], a1);
      } else {
        a0 = mfma16(w.w0[kk].x, in[kk][0], a0);
        a0 = mfma16(w.w0[kk].y, in[kk][1], a0);
        a0 = mfma16(w.w0[kk].z, in[kk][2], a0);
        a0 = mfma16(w.w0[kk].w, in[kk][3], a0);
      }
    }
    a0 = a0 + a1;
  }
}

// One Linear over the workgroup's rows.  The wave's TileRange (<= 4 tiles) is cut into <= 3 jobs of one or two
// output tiles of one row tile.  Weight fragments ping-pong between `pre` and `alt`: while job j computes, the
// fragments of job j+1 — or, for the last job, of the NEXT Linear's first job — are already in flight, so the L2
// latency overlaps MFMAs, the epilogue, the barrier and the next stage's LDS reads.  On entry `pre` holds job 0's
// fragments; on exit `pre` holds the next Linear's first job.   img: input image [64][LD]; epi(rt, ot, acc).
template <int NT, typename Epi>
__device__ __forceinline__ void coop_gemm(WPair<NT>& pre, WPair<NT>& alt, const float* __restrict__ wp, int nto,
                                          const float* img, int LD, TileRange tr, int lane, Epi epi,
                                          const float* __restrict__ next_wp, int next_nto, TileRange next_tr,
                                          const float* __restrict__ ev0 = nullptr, const float* __restrict__ ev1 = nullptr) {
  const int g = lane >> 4;
  int jrt[3], jot[3];
  bool jtwo[3];
  int nj = 0;
  {
    int t = tr.t_lo;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      jrt[j] = 0; jot[j] = 0; jtwo[j] = false;
      if (t < tr.t_hi) {
        const int rt = t / NT, ot = t - rt * NT;
        const bool two = (t + 1 < tr.t_hi) && (ot + 1 < NT);
        jrt[j] = rt; jot[j] = ot; jtwo[j] = two;
        t += two ? 2 : 1;
        nj = j + 1;
      }
    }
  }
  int nxo = 0;
  bool n1 = false, n2 = false;
  if (next_wp) next_tr.first(NT, nxo, n1, n2);
  f32x4 in[NT];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    if (j < nj) {
      WPair<NT>& cur = (j & 1) ? alt : pre;
      WPair<NT>& oth = (j & 1) ? pre : alt;
      const bool lastj = (j + 1 == nj);
      if (!lastj) wload<NT>(oth, wp, nto, jot[j + 1 < 3 ? j + 1 : 2], true, jtwo[j + 1 < 3 ? j + 1 : 2], lane);
      else if (next_wp) wload<NT>(oth, next_wp, next_nto, nxo, n1, n2, lane);
      if (j == 0 || jrt[j] != jrt[j > 0 ? j - 1 : 0]) {
        const float* rowp = img + (jrt[j] * 16 + (lane & 15)) * LD + 4 * g;
#pragma unroll
        for (int kk = 0; kk < NT; ++kk) in[kk] = lds_ld4(rowp + 16 * kk);
      }
      // per-channel epilogue vectors (folded BatchNorm / bias) are fetched BEFORE the MFMAs so their L2 latency
      // is hidden behind them instead of sitting between the last MFMA and the barrier
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
      const int c0 = 16 * jot[j] + 4 * g;
      const f32x4 e00 = ev0 ? ld4(ev0 + c0) : z4, e10 = ev1 ? ld4(ev1 + c0) : z4;
      const f32x4 e01 = (ev0 && jtwo[j]) ? ld4(ev0 + c0 + 16) : z4, e11 = (ev1 && jtwo[j]) ? ld4(ev1 + c0 + 16) : z4;
      f32x4 a0, a1;
      mfma_pair<NT>(cur, in, jtwo[j], a0, a1);
      epi(jrt[j], jot[j], a0, e00, e10);
      if (jtwo[j]) epi(jrt[j], jot[j] + 1, a1, e01, e11);
      if (lastj && next_wp && (j & 1) == 0) {   // the prefetched fragments sit in `alt`: hand them over in `pre`
#pragma unroll
        for (int kk = 0; kk < NT; ++kk) { pre.w0[kk] = alt.w0[kk]; pre.w1[kk] = alt.w1[kk]; }
      }
    }
  }
  if (nj == 0 && next_wp) wload<NT>(pre, next_wp, next_nto, nxo, n1, n2, lane);   // idle here: keep the chain going
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
  float* EE = reinterpret_cast<float*>(efeat + GNN_EMAX * (P.n_layers > 0 ? P.edge_nf : 0));   // [ee_rows][LD] this layer's edge embeddings
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, li = lane & 15;
  const int gi = blockIdx.x;
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
  const int q = (ntile + GNN_WAVES - 1) / GNN_WAVES;          // tiles per wave (<= NT/2 since T <= 4)
  TileRange tr;                                               // my share of every node-row Linear
  tr.t_lo = wave * q < ntile ? wave * q : ntile;
  tr.t_hi = tr.t_lo + q < ntile ? tr.t_lo + q : ntile;
  TileRange hr;                                               // my share of the output encoder (one pooled row tile)
  hr.t_lo = wave < NT ? wave : NT;
  hr.t_hi = wave < NT ? wave + 1 : NT;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const bool use_ee = P.n_layers > 0 && ne <= S.ee_rows;
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

  WPair<NT> pre, alt;
  {
    int fot; bool f1, f2;
    tr.first(NT, fot, f1, f2);
    wload<NT>(pre, P.rho_out_w, NT, fot, f1, f2, lane);       // in flight while the inputs are staged
  }
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
  // ---------------------------------------------------------------- stage the slot sum (rho output) in X1
  for (int i = threadIdx.x; i < n * d; i += GNN_WAVES * 64) {
    const int rr = i / d, c = i - rr * d;
    X1[rr * LD + c] = S.rho_sum[(int64_t)(gs + rr) * d + c];
  }
  if (d < D) {
    for (int i = threadIdx.x; i < n * (D - d); i += GNN_WAVES * 64) {
      const int rr = i / (D - d), c = d + i - rr * (D - d);
      X1[rr * LD + c] = 0.f;
    }
  }
  // ---------------------------------------------------------------- input encoder -> X0 (model.py:37)
  if (P.node_discrete) {
    for (int t = tr.t_lo; t < tr.t_hi; ++t) {
      const int rt = t / NT, ot = t - rt * NT, row = rt * 16 + li, c = 16 * ot + 4 * g;
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
      const int rt = t / NT, ot = t - rt * NT, row = rt * 16 + li, c = 16 * ot + 4 * g;
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
  ee_fetch(0);     // needs efeat (staged above); the loads fly during the three Linears below
  // ---------------------------------------------------------------- pos = BN(W_out . slot_sum): X1 -> X2   (sign_net.py:71)
  coop_gemm<NT>(pre, alt, P.rho_out_w, NT, X1, LD, tr, lane, [&](int rt, int ot, f32x4 acc, f32x4 sc, f32x4 sh) {
    lds_st4(X2 + (rt * 16 + li) * LD + 16 * ot + 4 * g, acc * sc + sh);
  }, P.lin_a, NT, tr, P.rho_scale, P.rho_shift);
  __syncthreads();
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
  // ---------------------------------------------------------------- GINE layers: h lives in X1           (model.py:47-55)
  for (int l = 0; l < P.n_layers; ++l) {
    const sn_gnn_layer& Lp = P.layers[l];
    // u = sum_{j->i} relu(h_j + e_ji) + (1+eps) h_i  for my (row tile, channel tile) pairs: X1 -> X2
    ee_fetch(l + 1);   // next layer's edge embeddings: in flight during this aggregation
    {
      const float sc = 1.f + *Lp.eps;
#pragma unroll 1
      for (int k = 0; k < 2; ++k) {
        int rt, o_lo, o_hi;
        tr.group(k, NT, rt, o_lo, o_hi);
        const int row = rt * 16 + li;
        if (o_lo >= o_hi || row >= n) continue;
        f32x4 u[GNN_OTS];
#pragma unroll
        for (int o = 0; o < GNN_OTS; ++o) u[o] = zero4;
        for (int e = erow[row]; e < erow[row + 1]; ++e) {
          const float* hsrc = X1 + esrc[e] * LD + 4 * g;
#pragma unroll
          for (int o = 0; o < GNN_OTS; ++o) {
            if (o_lo + o < o_hi) {
              const int c = 16 * (o_lo + o) + 4 * g;
              const f32x4 ef = use_ee ? lds_ld4(EE + e * LD + c) : edge_embed(Lp, e, c);
              u[o] += relu4(lds_ld4(hsrc + 16 * (o_lo + o)) + ef);
            }
          }
        }
#pragma unroll
        for (int o = 0; o < GNN_OTS; ++o) {
          if (o_lo + o < o_hi) {
            const int c = 16 * (o_lo + o) + 4 * g;
            {
#pragma clang fp contract(off)
              const f32x4 self = lds_ld4(X1 + row * LD + c) * sc;
              u[o] = u[o] + self;
            }
            lds_st4(X2 + row * LD + c, u[o]);
          }
        }
      }
    }
    __syncthreads();
    ee_store();        // every wave is done reading this layer's embeddings
    // nn: Linear . BN . ReLU : X2 -> X0
    coop_gemm<NT>(pre, alt, Lp.w1p, NT, X2, LD, tr, lane, [&](int rt, int ot, f32x4 acc, f32x4 sc, f32x4 sh) {
      lds_st4(X0 + (rt * 16 + li) * LD + 16 * ot + 4 * g, relu4(acc * sc + sh));
    }, Lp.w2p, NT, tr, Lp.bn0_scale, Lp.bn0_shift);
    __syncthreads();
    // Linear ; BN . ReLU ; + previous_x : X0 -> X1 (my tiles only: nobody else reads them at this point)
    const bool lastl = l + 1 == P.n_layers;
    coop_gemm<NT>(pre, alt, Lp.w2p, NT, X0, LD, tr, lane, [&](int rt, int ot, f32x4 acc, f32x4 sc, f32x4 sh) {
      float* o = X1 + (rt * 16 + li) * LD + 16 * ot + 4 * g;
      lds_st4(o, relu4(acc * sc + sh) + lds_ld4(o));
    }, lastl ? P.head_w1 : P.layers[lastl ? l : l + 1].w1p, NT, lastl ? hr : tr, Lp.bn_scale, Lp.bn_shift);
    __syncthreads();
  }
  // ---------------------------------------------------------------- add pooling -> row 0 of X2 (rows 1..15 zero)  (model.py:57-61)
  for (int c = threadIdx.x; c < 16 * LD; c += GNN_WAVES * 64) {
    const int rr = c / LD, cc = c - rr * LD;
    float s = 0.f;
    if (rr == 0 && cc < D)
      for (int j = 0; j < n; ++j) s += X1[j * LD + cc];
    X2[c] = s;
  }
  __syncthreads();
  // ---------------------------------------------------------------- output encoder on the pooled row     (model.py:63)
  TileRange h2;
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
}

template <int NT>
static int launch_gnn(const GnnStruct& S, const sn_gnn_params& P, int64_t B, hipStream_t st) {
  constexpr int LD = 16 * NT + 4;
  const size_t base = (size_t)(3 * GNN_ROWS * LD) * sizeof(float) + (size_t)(GNN_ROWS + 4 + GNN_EMAX * (1 + (P.n_layers > 0 ? P.edge_nf : 0))) * sizeof(int);
  const size_t room = base < 160 * 1024 ? 160 * 1024 - base : 0;
  int ee_rows = (int)(room / ((size_t)LD * sizeof(float)));
  if (ee_rows > GNN_EEMAX) ee_rows = GNN_EEMAX;
  GnnStruct S2 = S;
  S2.ee_rows = P.n_layers > 0 ? ee_rows : 0;
  const size_t lds = base + (size_t)S2.ee_rows * LD * sizeof(float);
  static bool init = false;
  if (!init) {
    const size_t lds_max = 160 * 1024;
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
