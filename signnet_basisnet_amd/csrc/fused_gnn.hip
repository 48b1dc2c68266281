// fused_gnn.hip — the GINE network that consumes the positional encoding, whole stack in ONE launch.
// Replaces (eval mode) the tail of SetTransformer.forward — `self.out` Linear+BatchNorm on the slot sum
// (sign_net.py:71) — and GNN.forward (Alchemy/sign_net/model.py:36-64, GINESignNetPyG/core/model.py:44-79):
// input encoder, Linear(cat[x, pos]), nl_gnn x [edge encoder, GINEConv (pyg_gnn_wrapper.py:19-28),
// BatchNorm, ReLU, residual], add-pooling over each graph and the 2-layer output encoder.
//
// Every op is local to one graph, so a workgroup keeps a bin of whole graphs (bins kind 2, 64 rows) on chip:
// node rows live in registers in the MFMA operand layout, the GINE neighbour sum relu(h_j + e_ji) and the
// pooling go through one LDS image, all Linear layers are chained gemm_rows() calls (fp32 MFMA).
#include "fused_common.hpp"

namespace sn {

constexpr int GNN_R = SN_GNN_BIN_ROWS;

struct GnnStruct {
  const void* x;          // int64 [N, ldx] (discrete) or float [N, F]
  int ldx;
  const void* edge_attr;  // int64 [E, lde] (discrete) or float [E, F_e]
  int lde;
  const float* rho_sum;   // [N, d]
  const int32_t* graph_ptr;
  const int32_t* node_graph;
  const int32_t* rowptr;
  const int32_t* col;
  const int32_t* eperm;
  const int32_t* bin_node;
  const int32_t* meta;
  int64_t max_bins;
  float* y;               // [B, n_out]
};

template <int NT>
__device__ __forceinline__ void load_row(f32x4 (&v)[NT], const float* __restrict__ row, int d, int g) {
#pragma unroll
  for (int kk = 0; kk < NT; ++kk) {
    const int c = 16 * kk + 4 * g;
    v[kk] = f32x4{0.f, 0.f, 0.f, 0.f};
    if ((d & 3) == 0) {
      if (c < d) v[kk] = ld4(row + c);
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (c + t < d) v[kk][t] = row[c + t];
    }
  }
}

template <int NT>
__global__ __launch_bounds__(GNN_R * 4, 2) void k_gnn_fused(GnnStruct S, sn_gnn_params P) {
  constexpr int D = 16 * NT;
  constexpr int LD = D + 4;
  extern __shared__ __align__(16) float lds[];
  float* Himg = lds;  // [GNN_R][LD]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = wave * 16 + (lane & 15), g = lane >> 4;
  const int nbins = S.meta[0];
  if (S.meta[1] != 0) return;
  const int d = P.d;

  for (int bin = blockIdx.x; bin < nbins; bin += gridDim.x) {
    const int node = S.bin_node[(int64_t)bin * GNN_R + r];
    const bool valid = node >= 0;
    int gi = 0, gs = 0, n = 0, row0 = 0, e_lo = 0, e_hi = 0;
    if (valid) {
      gi = S.node_graph[node];
      gs = S.graph_ptr[gi];
      n = S.graph_ptr[gi + 1] - gs;
      row0 = r - (node - gs);
      e_lo = S.rowptr[node];
      e_hi = S.rowptr[node + 1];
    }
    float* Hr = Himg + r * LD;
    f32x4 h[NT], u[NT], t[NT];
    // ---------------------------------------------------------------- input encoder (model.py:37)
    if (P.node_discrete) {
#pragma unroll
      for (int kk = 0; kk < NT; ++kk) h[kk] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (valid) {
        const int64_t* xi = reinterpret_cast<const int64_t*>(S.x) + (int64_t)node * S.ldx;
        for (int f = 0; f < P.node_nf; ++f) {
          const float* trow = P.ntab[f] + xi[f] * d;
          f32x4 e[NT];
          load_row<NT>(e, trow, d, g);
#pragma unroll
          for (int kk = 0; kk < NT; ++kk) h[kk] += e[kk];
        }
      }
    } else {
      // MLP(nfeat, d, 1): Linear(no bias) . BN . ReLU       (elements.py:39-69)
      f32x4 xin[1];
      xin[0] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (valid) {
        const float* xr = reinterpret_cast<const float*>(S.x) + (int64_t)node * S.ldx;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (4 * g + q < P.node_nf) xin[0][q] = xr[4 * g + q];
      }
      gemm_rows2<1, NT>(P.nw, xin, lane, [&](int ot, f32x4 acc) {
        const int c = 16 * ot + 4 * g;
        h[ot] = relu4(acc * ld4(P.n_scale + c) + ld4(P.n_shift + c));
      });
    }
    // ---------------------------------------------------------------- pos = BN(W_out . slot_sum)   (sign_net.py:71)
    if (valid) load_row<NT>(u, S.rho_sum + (int64_t)node * d, d, g);
    else {
#pragma unroll
      for (int kk = 0; kk < NT; ++kk) u[kk] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    gemm_rows<NT>(P.rho_out_w, u, lane, [&](int ot, f32x4 acc) {
      const int c = 16 * ot + 4 * g;
      t[ot] = acc * ld4(P.rho_scale + c) + ld4(P.rho_shift + c);
    });
    // ---------------------------------------------------------------- x = Linear(cat[x, pos])       (model.py:39-40)
    gemm_rows<NT>(P.lin_a, h, lane, [&](int ot, f32x4 acc) { u[ot] = acc; });
    gemm_rows<NT>(P.lin_b, t, lane, [&](int ot, f32x4 acc) {
      h[ot] = valid ? (u[ot] + acc) + ld4(P.lin_bias + 16 * ot + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
    });
    // ---------------------------------------------------------------- GINE layers                  (model.py:47-55)
    for (int l = 0; l < P.n_layers; ++l) {
      const sn_gnn_layer& Lp = P.layers[l];
#pragma unroll
      for (int kk = 0; kk < NT; ++kk) lds_st4(Hr + 16 * kk + 4 * g, h[kk]);
      __syncthreads();
#pragma unroll
      for (int kk = 0; kk < NT; ++kk) u[kk] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int e = e_lo; e < e_hi; ++e) {
        const float* nb = Himg + (row0 + S.col[e] - gs) * LD + 4 * g;
        const int eid = S.eperm[e];
        f32x4 ef[NT];
        if (P.edge_discrete) {
#pragma unroll
          for (int kk = 0; kk < NT; ++kk) ef[kk] = f32x4{0.f, 0.f, 0.f, 0.f};
          const int64_t* ei = reinterpret_cast<const int64_t*>(S.edge_attr) + (int64_t)eid * S.lde;
          for (int f = 0; f < P.edge_nf; ++f) {
            f32x4 tr[NT];
            load_row<NT>(tr, Lp.etab[f] + ei[f] * d, d, g);
#pragma unroll
            for (int kk = 0; kk < NT; ++kk) ef[kk] += tr[kk];
          }
        } else {
          // MLP(nfeat_edge, d, 1): relu(BN(W ea))   — F_e <= 16 inputs, done on the VALU per edge
          const float* ea = reinterpret_cast<const float*>(S.edge_attr) + (int64_t)eid * S.lde;
#pragma unroll
          for (int kk = 0; kk < NT; ++kk) {
            const int c = 16 * kk + 4 * g;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int f = 0; f < P.edge_nf; ++f) {
              const float a = ea[f];
#pragma unroll
              for (int q = 0; q < 4; ++q) acc[q] += a * Lp.ew[(c + q) * P.edge_nf + f];   // ew zero-padded to d_pad rows
            }
            ef[kk] = relu4(acc * ld4(Lp.e_scale + c) + ld4(Lp.e_shift + c));
          }
        }
#pragma unroll
        for (int kk = 0; kk < NT; ++kk) u[kk] += relu4(lds_ld4(nb + 16 * kk) + ef[kk]);
      }
      {
#pragma clang fp contract(off)
        const float sc = 1.f + *Lp.eps;
#pragma unroll
        for (int kk = 0; kk < NT; ++kk) {
          const f32x4 self = h[kk] * sc;
          u[kk] = u[kk] + self;
        }
      }
      // nn = Linear . BN . ReLU . Linear ; then BN . ReLU . + previous_x
      gemm_rows<NT>(Lp.w1p, u, lane, [&](int ot, f32x4 acc) {
        const int c = 16 * ot + 4 * g;
        t[ot] = relu4(acc * ld4(Lp.bn0_scale + c) + ld4(Lp.bn0_shift + c));
      });
      gemm_rows<NT>(Lp.w2p, t, lane, [&](int ot, f32x4 acc) {
        const int c = 16 * ot + 4 * g;
        h[ot] = valid ? relu4(acc * ld4(Lp.bn_scale + c) + ld4(Lp.bn_shift + c)) + h[ot] : f32x4{0.f, 0.f, 0.f, 0.f};
      });
      __syncthreads();   // all neighbour reads of this layer's image are done
    }
    // ---------------------------------------------------------------- add-pooling per graph       (model.py:57-61)
#pragma unroll
    for (int kk = 0; kk < NT; ++kk) lds_st4(Hr + 16 * kk + 4 * g, h[kk]);
    __syncthreads();
    const bool first = valid && node == gs;   // first node of its graph: owns the pooled row
    if (first) {
#pragma unroll
      for (int kk = 0; kk < NT; ++kk) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < n; ++j) s += lds_ld4(Himg + (row0 + j) * LD + 16 * kk + 4 * g);
        u[kk] = s;
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < NT; ++kk) u[kk] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();   // the image is rewritten by the next bin
    // ---------------------------------------------------------------- output encoder               (model.py:63)
    // The pooled rows sit at the bin rows of each graph's first node; a wave runs the 2 GEMMs only if it has one.
    if (__ballot(first) != 0ull) {
      gemm_rows<NT>(P.head_w1, u, lane, [&](int ot, f32x4 acc) {
        const int c = 16 * ot + 4 * g;
        t[ot] = relu4(acc * ld4(P.head_scale + c) + ld4(P.head_shift + c));
      });
      gemm_rows2<NT, 1>(P.head_w2, t, lane, [&](int ot, f32x4 acc) {
        if (first) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int c = 4 * g + q;
            if (c < P.n_out) S.y[(int64_t)gi * P.n_out + c] = acc[q] + P.head_b2[c];
          }
        }
      });
    }
  }
}

template <int NT>
static int launch_gnn(const GnnStruct& S, const sn_gnn_params& P, hipStream_t st) {
  constexpr int LD = 16 * NT + 4;
  const size_t lds = (size_t)(GNN_R * LD) * sizeof(float);
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    cus = n > 0 ? n : 256;
  }
  int64_t grid = S.max_bins < (int64_t)2 * cus ? S.max_bins : (int64_t)2 * cus;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL((k_gnn_fused<NT>), dim3((unsigned)grid), dim3(GNN_R * 4), lds, st, S, P);
  return SN_OK;
}

}  // namespace sn

using namespace sn;

extern "C" int sn_gnn_fused_f32(const sn_gnn_params* params, const void* x, int ldx, const void* edge_attr, int lde,
                                const float* rho_sum, const int32_t* graph_ptr, const int32_t* node_graph,
                                const int32_t* rowptr, const int32_t* col, const int32_t* eperm,
                                const int32_t* bin_node, const int32_t* meta, int64_t max_bins, float* y,
                                void* stream) {
  SN_REQUIRE(params && x && rho_sum && graph_ptr && node_graph && rowptr && bin_node && meta && y,
             "sn_gnn_fused_f32: null pointer");
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
  if (max_bins == 0) return SN_OK;
  GnnStruct S{x, ldx, edge_attr, lde, rho_sum, graph_ptr, node_graph, rowptr, col, eperm, bin_node, meta, max_bins, y};
  hipStream_t st = (hipStream_t)stream;
  int rc = SN_OK;
  switch ((P.d + 15) / 16) {
    case 1: rc = launch_gnn<1>(S, P, st); break;
    case 2: rc = launch_gnn<2>(S, P, st); break;
    case 3: rc = launch_gnn<3>(S, P, st); break;
    case 4: rc = launch_gnn<4>(S, P, st); break;
    case 5: rc = launch_gnn<5>(S, P, st); break;
    case 6: rc = launch_gnn<6>(S, P, st); break;
    case 7: rc = launch_gnn<7>(S, P, st); break;
    default: rc = launch_gnn<8>(S, P, st); break;
  }
  if (rc != SN_OK) return rc;
  SN_CHECK_LAUNCH("sn_gnn_fused_f32");
  return SN_OK;
}
