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

// One Linear over the workgroup's rows.  The wave computes its TileRange, two output tiles at a time (two
// independent accumulator chains; a lone tile splits its k-chunks over the two chains instead).  `pre` must hold
// the fragments of the range's first tile pair (prefetched during the previous stage); once the last MFMAs are
// issued, the fragments of the NEXT Linear's first pair are fetched into `pre`, so that latency overlaps the
// epilogue, the barrier and the next stage's LDS reads.   img: input image [64][LD]; epi(rt, ot, acc).
template <int NT, typename Epi>
__device__ __forceinline__ void coop_gemm(WPair<NT>& pre, const float* __restrict__ wp, int nto, const float* img, int LD,
                                          TileRange tr, int lane, Epi epi, const float* __restrict__ next_wp, int next_nto,
                                          TileRange next_tr) {
  const int g = lane >> 4;
  bool fresh = true;   // `pre` is valid for the first pair only
#pragma unroll 1
  for (int k = 0; k < 2; ++k) {
    int rt, o_lo, o_hi;
    tr.group(k, NT, rt, o_lo, o_hi);
    if (o_lo >= o_hi) continue;
    f32x4 in[NT];
    const float* rowp = img + (rt * 16 + (lane & 15)) * LD + 4 * g;
#pragma unroll
    for (int kk = 0; kk < NT; ++kk) in[kk] = lds_ld4(rowp + 16 * kk);
#pragma unroll 1
    for (int ot = o_lo; ot < o_hi; ot += 2) {
      const bool two = ot + 1 < o_hi;
      if (!fresh) wload<NT>(pre, wp, nto, ot, true, two, lane);
      fresh = false;
      f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
      if (two) {
#pragma unroll
        for (int kk = 0; kk < NT; ++kk) {
          a0 = mfma16(pre.w0[kk].x, in[kk][0], a0);
          a1 = mfma16(pre.w1[kk].x, in[kk][0], a1);
          a0 = mfma16(pre.w0[kk].y, in[kk][1], a0);
          a1 = mfma16(pre.w1[kk].y, in[kk][1], a1);
          a0 = mfma16(pre.w0[kk].z, in[kk][2], a0);
          a1 = mfma16(pre.w1[kk].z, in[kk][2], a1);
          a0 = mfma16(pre.w0[kk].w, in[kk][3], a0);
          a1 = mfma16(pre.w1[kk].w, in[kk][3], a1);
        }
      } else {
#pragma unroll
        for (int kk = 0; kk < NT; ++kk) {
          if (kk & 1) {
            a1 = mfma16(pre.w0[kk].x, in[kk][0], a1);
            a1 = mfma16(pre.w0[kk].y, in[kk][1], a1);
            a1 = mfma16(pre.w0[kk].z, in[kk][2], a1);
            a1 = mfma16(pre.w0[kk].w, in[kk][3], a1);
          } else {
            a0 = mfma16(pre.w0[kk].x, in[kk][0], a0);
            a0 = mfma16(pre.w0[kk].y, in[kk][1], a0);
            a0 = mfma16(pre.w0[kk].z, in[kk][2], a0);
            a0 = mfma16(pre.w0[kk].w, in[kk][3], a0);
          }
        }
        a0 = a0 + a1;
      }
      const bool last = (ot + 2 >= o_hi) && (k == 1 || tr.t_hi <= (tr.t_lo / NT + 1) * NT);
      if (last && next_wp) {
        int nxo; bool n1, n2;
        next_tr.first(NT, nxo, n1, n2);
        wload<NT>(pre, next_wp, next_nto, nxo, n1, n2, lane);
      }
      epi(rt, ot, a0);
      if (two) epi(rt, ot + 1, a1);
    }
  }
  if (tr.empty() && next_wp) {   // idle in this Linear: still prefetch for the next one
    int nxo; bool n1, n2;
    next_tr.first(NT, nxo, n1, n2);
    wload<NT>(pre, next_wp, next_nto, nxo, n1, n2, lane);
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

  WPair<NT> pre;
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
  // ---------------------------------------------------------------- pos = BN(W_out . slot_sum): X1 -> X2   (sign_net.py:71)
  coop_gemm<NT>(pre, P.rho_out_w, NT, X1, LD, tr, lane, [&](int rt, int ot, f32x4 acc) {
    const int c = 16 * ot + 4 * g;
    lds_st4(X2 + (rt * 16 + li) * LD + c, acc * ld4(P.rho_scale + c) + ld4(P.rho_shift + c));
  }, P.lin_a, NT, tr);
  __syncthreads();
  // ---------------------------------------------------------------- h = Linear(cat[x, pos]): X0, X2 -> X1    (model.py:39-40)
  coop_gemm<NT>(pre, P.lin_a, NT, X0, LD, tr, lane, [&](int rt, int ot, f32x4 acc) {
    lds_st4(X1 + (rt * 16 + li) * LD + 16 * ot + 4 * g, acc);      // readers of X1 (slot sum) passed a barrier
  }, P.lin_b, NT, tr);
  coop_gemm<NT>(pre, P.lin_b, NT, X2, LD, tr, lane, [&](int rt, int ot, f32x4 acc) {
    const int c = 16 * ot + 4 * g;
    float* o = X1 + (rt * 16 + li) * LD + c;
    lds_st4(o, (lds_ld4(o) + acc) + ld4(P.lin_bias + c));
  }, P.n_layers > 0 ? P.layers[0].w1p : P.head_w1, NT, P.n_layers > 0 ? tr : hr);
  __syncthreads();
  // ---------------------------------------------------------------- GINE layers: h lives in X1           (model.py:47-55)
  for (int l = 0; l < P.n_layers; ++l) {
    const sn_gnn_layer& Lp = P.layers[l];
    // u = sum_{j->i} relu(h_j + e_ji) + (1+eps) h_i  for my (row tile, channel tile) pairs: X1 -> X2
    {
      const float sc = 1.f + *Lp.eps;
      const int EF = P.edge_nf;
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
              f32x4 ef = zero4;
              if (P.edge_discrete) {
                for (int f = 0; f < EF; ++f) {
                  const float* trow = Lp.etab[f] + (int64_t)efeat[e * EF + f] * d;
                  if ((d & 3) == 0) { if (c < d) ef += ld4(trow + c); }
                  else {
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) if (c + qq < d) ef[qq] += trow[c + qq];
                  }
                }
              } else {
                f32x4 acc = zero4;
                for (int f = 0; f < EF; ++f) {
                  const float a = __int_as_float(efeat[e * EF + f]);
#pragma unroll
                  for (int qq = 0; qq < 4; ++qq) acc[qq] += a * Lp.ew[(c + qq) * EF + f];
                }
                ef = relu4(acc * ld4(Lp.e_scale + c) + ld4(Lp.e_shift + c));
              }
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
    // nn: Linear . BN . ReLU : X2 -> X0
    coop_gemm<NT>(pre, Lp.w1p, NT, X2, LD, tr, lane, [&](int rt, int ot, f32x4 acc) {
      const int c = 16 * ot + 4 * g;
      lds_st4(X0 + (rt * 16 + li) * LD + c, relu4(acc * ld4(Lp.bn0_scale + c) + ld4(Lp.bn0_shift + c)));
    }, Lp.w2p, NT, tr);
    __syncthreads();
    // Linear ; BN . ReLU ; + previous_x : X0 -> X1 (my tiles only: nobody else reads them at this point)
    const bool lastl = l + 1 == P.n_layers;
    coop_gemm<NT>(pre, Lp.w2p, NT, X0, LD, tr, lane, [&](int rt, int ot, f32x4 acc) {
      const int c = 16 * ot + 4 * g;
      float* o = X1 + (rt * 16 + li) * LD + c;
      lds_st4(o, relu4(acc * ld4(Lp.bn_scale + c) + ld4(Lp.bn_shift + c)) + lds_ld4(o));
    }, lastl ? P.head_w1 : P.layers[lastl ? l : l + 1].w1p, NT, lastl ? hr : tr);
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
  coop_gemm<NT>(pre, P.head_w1, NT, X2, LD, hr, lane, [&](int rt, int ot, f32x4 acc) {
    const int c = 16 * ot + 4 * g;
    lds_st4(X0 + li * LD + c, relu4(acc * ld4(P.head_scale + c) + ld4(P.head_shift + c)));
  }, wave == 0 ? P.head_w2 : nullptr, 1, h2);
  __syncthreads();
  coop_gemm<NT>(pre, P.head_w2, 1, X0, LD, h2, lane, [&](int rt, int ot, f32x4 acc) {
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
  const size_t lds = (size_t)(3 * GNN_ROWS * LD) * sizeof(float) + (size_t)(GNN_ROWS + 4 + GNN_EMAX * (1 + (P.n_layers > 0 ? P.edge_nf : 0))) * sizeof(int);
  static bool init = false;
  if (!init) {
    const size_t lds_max = (size_t)(3 * GNN_ROWS * LD) * sizeof(float) + (size_t)(GNN_ROWS + 4 + GNN_EMAX * 17) * sizeof(int);
    if (lds_max > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_gnn_coop<NT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_max) != hipSuccess)
      return fail(SN_ERR_LAUNCH, "sn_gnn_fused_f32: cannot raise the dynamic LDS limit to %zu", lds_max);
    init = true;
  }
  hipLaunchKernelGGL((k_gnn_coop<NT>), dim3((unsigned)B), dim3(GNN_WAVES * 64), lds, st, S, P);
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
  GnnStruct S{x, ldx, edge_attr, lde, rho_sum, graph_ptr, rowptr, col, eperm, status, y};
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
