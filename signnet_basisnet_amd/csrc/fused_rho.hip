// fused_rho.hip — the set-transformer rho over the eigenvector-slot axis, all encoder layers + the slot sum
// in ONE launch.  Replaces SetTransformer.forward up to `torch.sum(x, dim=1)` (Alchemy/sign_net/sign_net.py:60-70,
// GINESignNetPyG/core/sign_net.py:64-75) and the TransformerEncoderLayer / MultiHeadAttention /
// PositionwiseFeedForward / MaskedLN stack it calls (model_utils/transformer_module.py:27-127), plus
// (Alchemy only) the eigenvalue encoder `x + pos` (sign_net.py:108,62).
//
// Everything rho does is local to one node's valid slot rows, so a workgroup keeps a bin of whole nodes
// (64 rows; a node's K_g rows padded to a multiple of 16 so a node never straddles a wave's 16-row tile) on chip
// for the entire stack.  With K_g <= 16 and head width a multiple of 16 the attention itself runs on the matrix
// pipe, entirely in registers: S^T = K.Q^T and O^T = V^T.P^T are MFMAs whose operand layouts are exactly the
// accumulator layouts of the q / k projections and of a v projection computed with swapped operands:
//   * rows live in registers in the MFMA operand layout; the six [rows,d]x[d,d] projections per layer are
//     chained gemm_rows() calls (fp32 MFMA) — residuals and LayerNorm run on the accumulators;
//   * attention: q, k, v rows are exchanged through two LDS images; lane (row, head) computes its
//     query's scores against the node's keys (two passes: max, then exp / sum / P.V), exactly
//     softmax(q k^T / sqrt(dk)) restricted to the node's valid slots (transformer_module.py:52-57);
//   * the final sum over slots is done from LDS by each node's first row.
// Bound: fp32 MFMA; flops per valid row and layer: 6 * 2*d*d (+ 4*K*d attention on the VALU).
#include "fused_common.hpp"

namespace sn {

constexpr int RHO_R = SN_RHO_BIN_ROWS;

struct RhoStruct {
  const float* x;        // phi(x)+phi(-x), row = node*K + slot
  const float* eigvals;  // [N] (only with has_pos)
  const int32_t* graph_ptr;
  const int32_t* rho_bin0;  // [B+1] first bin of every graph
  const int32_t* meta;      // [0] nbins, [1] error
  int B;
  int kmax;
  int K;
  float* out_sum;        // [N, d]
};

template <int NT>
__device__ __forceinline__ void masked_layernorm(f32x4 (&v)[NT], const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, float eps, int d, int g, bool valid) {
  float s = 0.f;
#pragma unroll
  for (int kk = 0; kk < NT; ++kk) s += (v[kk][0] + v[kk][1]) + (v[kk][2] + v[kk][3]);   // padded channels hold 0
  const float mean = row_allsum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int kk = 0; kk < NT; ++kk) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float dlt = (16 * kk + 4 * g + t < d) ? v[kk][t] - mean : 0.f;
      q += dlt * dlt;
    }
  }
  const float rstd = 1.0f / sqrtf(row_allsum(q) / (float)d + eps);
#pragma unroll
  for (int kk = 0; kk < NT; ++kk) {
    const int c = 16 * kk + 4 * g;
    const f32x4 ga = ld4(gamma + c), be = ld4(beta + c);   // zero padded -> padded channels stay 0
    f32x4 o;
#pragma unroll
    for (int t = 0; t < 4; ++t) o[t] = valid ? (v[kk][t] - mean) * rstd * ga[t] + be[t] : 0.f;
    v[kk] = o;
  }
}

// gemm with SWAPPED MFMA operands: acc[r] = Y[row = 4g + r][o = 16*ot + (l&15)] — the "channel in lane, rows in
// registers" layout that the P.V product needs for V (same packed weight fragments, same k order).
template <int NT, typename Epi>
__device__ __forceinline__ void gemm_rows_t(const float* __restrict__ wp, const f32x4 (&in)[NT], int lane, Epi epi) {
  const __amdgpu_buffer_rsrc_t rs = weight_rsrc(wp, NT * NT * 1024);
  const int voff = lane * 16;
  float4 wA[NT], wB[NT];
#pragma unroll
  for (int kk = 0; kk < NT; ++kk) wA[kk] = wfrag(rs, voff, kk);
  auto tile = [&](const float4 (&w)[NT]) {
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < NT; kk += 2) {
      a0 = mfma16(in[kk][0], w[kk].x, a0);
      if (kk + 1 < NT) a1 = mfma16(in[kk + 1][0], w[kk + 1].x, a1);
      a0 = mfma16(in[kk][1], w[kk].y, a0);
      if (kk + 1 < NT) a1 = mfma16(in[kk + 1][1], w[kk + 1].y, a1);
      a0 = mfma16(in[kk][2], w[kk].z, a0);
      if (kk + 1 < NT) a1 = mfma16(in[kk + 1][2], w[kk + 1].z, a1);
      a0 = mfma16(in[kk][3], w[kk].w, a0);
      if (kk + 1 < NT) a1 = mfma16(in[kk + 1][3], w[kk + 1].w, a1);
    }
    return a0 + a1;
  };
#pragma unroll
  for (int ot = 0; ot < NT; ot += 2) {
    if (ot + 1 < NT) {
#pragma unroll
      for (int kk = 0; kk < NT; ++kk) wB[kk] = wfrag(rs, voff, (ot + 1) * NT + kk);
    }
    __builtin_amdgcn_sched_barrier(0);
    epi(ot, tile(wA));
    __builtin_amdgcn_sched_barrier(0);
    if (ot + 1 < NT) {
      if (ot + 2 < NT) {
#pragma unroll
        for (int kk = 0; kk < NT; ++kk) wA[kk] = wfrag(rs, voff, (ot + 2) * NT + kk);
      }
      __builtin_amdgcn_sched_barrier(0);
      epi(ot + 1, tile(wB));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

__device__ __forceinline__ float group_allmax(float v) {   // over the 4 lane groups holding one row
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float tile_rowsum(float v) {    // over the 16 rows (lanes l&15) of a tile, same lane group
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  return v + __shfl_xor(v, 8, 64);
}

template <int NT>
__global__ __launch_bounds__(RHO_R * 4, 2) void k_rho_fused(RhoStruct S, sn_rho_params P) {
  constexpr int D = 16 * NT;
  constexpr int LD = D + 4;
  constexpr int DKMAX = (D + 3) / 4;   // heads = 4: dk <= D/4
  extern __shared__ __align__(16) float lds[];
  float* A = lds;                 // [RHO_R][LD]   q, then v          (LDS attention path only)
  float* Bm = lds + RHO_R * LD;   // [RHO_R][LD]   k, then the attention output / slot-sum image
  __shared__ int s_graph;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = wave * 16 + (lane & 15), g = lane >> 4, li = lane & 15;
  const int nbins = S.meta[4];
  if (S.meta[5] != 0) return;
  const int d = P.d, H = P.heads, dk = d / H;
  const float temp = sqrtf((float)dk);

  for (int bin = blockIdx.x; bin < nbins; bin += gridDim.x) {
    // ---------------------------------------------------------------- bin -> graph (bins never mix graphs)
    __syncthreads();
    for (int gq = threadIdx.x; gq < S.B; gq += RHO_R * 4)
      if (S.rho_bin0[gq] <= bin && bin < S.rho_bin0[gq + 1]) s_graph = gq;
    __syncthreads();
    const int gi = s_graph;
    const int gs = S.graph_ptr[gi], n = S.graph_ptr[gi + 1] - gs;
    const int kg = (S.kmax > 0 && n > S.kmax) ? S.kmax : n;    // valid slots of every node of this graph
    const int pad = ((kg + 15) >> 4) << 4;                        // rows reserved per node (tile aligned)
    const int upb = RHO_R / pad;                                   // nodes per bin
    const int q = r / pad, slot = r - q * pad;
    const int u = (bin - S.rho_bin0[gi]) * upb + q;                // node index inside the graph
    const bool unit_ok = q < upb && u < n;                          // rows past upb*pad are padding
    const bool valid = unit_ok && slot < kg;
    const int node = gs + u;
    const int kv = unit_ok ? kg : 0;
    const int u0 = q * pad;                                        // bin row of my node's slot 0
    const bool mfma_attn = (pad == 16) && ((dk & 15) == 0);        // one node == one wave tile: attention in registers
    const bool wave_live = __ballot(valid) != 0ull;
    float* Ar = A + r * LD;
    float* Br = Bm + r * LD;
    // ---------------------------------------------------------------- load x (+ eigenvalue encoding)
    f32x4 x[NT];
#pragma unroll
    for (int kk = 0; kk < NT; ++kk) x[kk] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (valid) {
      const float* xr = S.x + ((int64_t)node * S.K + slot) * d;
#pragma unroll
      for (int kk = 0; kk < NT; ++kk) {
        const int c = 16 * kk + 4 * g;
        if ((d & 3) == 0) {
          if (c < d) x[kk] = ld4(xr + c);
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t)
            if (c + t < d) x[kk][t] = xr[c + t];
        }
      }
      if (P.has_pos) {
        // eigen_encoder = MaskedMLP(1 -> 1 -> d): Linear . BN . ReLU . Linear . BN . ReLU   (sign_net.py:86,108)
        const float ev = S.eigvals[gs + slot];
        const float t0 = fmaxf((ev * P.pe_w1[0]) * P.pe_bn0_scale[0] + P.pe_bn0_shift[0], 0.f);
#pragma unroll
        for (int kk = 0; kk < NT; ++kk) {
          const int c = 16 * kk + 4 * g;
          const f32x4 w2 = ld4(P.pe_w2 + c), s1 = ld4(P.pe_bn1_scale + c), h1 = ld4(P.pe_bn1_shift + c);
          x[kk] += relu4((t0 * w2) * s1 + h1);
        }
      }
    }
    // ---------------------------------------------------------------- encoder layers
    for (int l = 0; l < P.n_layers; ++l) {
      const sn_rho_layer& Lp = P.layers[l];
      f32x4 o[NT], y[NT];
      if (mfma_attn) {
        // ======== attention in registers (K_g <= 16: the node's slots are exactly this wave's 16 rows) ========
        if (wave_live) {
          f32x4 qf[NT], kf[NT], vt[NT];
          gemm_rows<NT>(Lp.wq, x, lane, [&](int ot, f32x4 acc) { qf[ot] = acc / temp; });   // q / sqrt(dk)  (:52)
          gemm_rows<NT>(Lp.wk, x, lane, [&](int ot, f32x4 acc) { kf[ot] = acc; });
          gemm_rows_t<NT>(Lp.wv, x, lane, [&](int ot, f32x4 acc) { vt[ot] = acc; });          // V[key = 4g+r][c = 16ot + li]
          constexpr int CPH = NT / 4 > 0 ? NT / 4 : 1;   // 16-channel chunks per head
#pragma unroll
          for (int h = 0; h < 4; ++h) {
            // S^T[key][query]: lane (query = li, g) holds S[query][key = 4g + r]
            f32x4 sc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int cc = 0; cc < CPH; ++cc) {
              const int kk = h * CPH + cc;
              if (kk < NT) {
                sc = mfma16(kf[kk][0], qf[kk][0], sc);
                sc = mfma16(kf[kk][1], qf[kk][1], sc);
                sc = mfma16(kf[kk][2], qf[kk][2], sc);
                sc = mfma16(kf[kk][3], qf[kk][3], sc);
              }
            }
            float m = -INFINITY;
#pragma unroll
            for (int t = 0; t < 4; ++t) { if (4 * g + t >= kv) sc[t] = -INFINITY; m = fmaxf(m, sc[t]); }
            m = group_allmax(m);
            float z = 0.f;
            f32x4 pr;
#pragma unroll
            for (int t = 0; t < 4; ++t) { pr[t] = (4 * g + t < kv) ? expf(sc[t] - m) : 0.f; z += pr[t]; }
            z = row_allsum(z);
            const float zi = (kv > 0) ? 1.0f / z : 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) pr[t] *= zi;
            // O^T[c][query] = sum_key V[key][c] P[query][key]  -> lane (query, g) holds O[query][16*ot + 4g + r]
#pragma unroll
            for (int cc = 0; cc < CPH; ++cc) {
              const int ot = h * CPH + cc;
              if (ot < NT) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                acc = mfma16(vt[ot][0], pr[0], acc);
                acc = mfma16(vt[ot][1], pr[1], acc);
                acc = mfma16(vt[ot][2], pr[2], acc);
                acc = mfma16(vt[ot][3], pr[3], acc);
                o[ot] = acc;
              }
            }
          }
        }
      } else {
        // ======== attention through LDS (nodes of more than 16 slots span several waves' tiles) ========
        if (wave_live) {
          gemm_rows<NT>(Lp.wq, x, lane, [&](int ot, f32x4 acc) { lds_st4(Ar + 16 * ot + 4 * g, acc); });
          gemm_rows<NT>(Lp.wk, x, lane, [&](int ot, f32x4 acc) { lds_st4(Br + 16 * ot + 4 * g, acc); });
        }
        __syncthreads();
        float qh[DKMAX];
        const int hc = g * dk;
#pragma unroll
        for (int c = 0; c < DKMAX; ++c) qh[c] = (c < dk && wave_live) ? Ar[hc + c] / temp : 0.f;
        // (q rows are written and read by the same wave only: no barrier before A is reused for v)
        if (wave_live) gemm_rows<NT>(Lp.wv, x, lane, [&](int ot, f32x4 acc) { lds_st4(Ar + 16 * ot + 4 * g, acc); });
        __syncthreads();
        float m = -INFINITY;
        float oh[DKMAX];
#pragma unroll
        for (int c = 0; c < DKMAX; ++c) oh[c] = 0.f;
        float z = 0.f;
        if ((dk & 3) == 0) {
          for (int j = 0; j < kv; ++j) {
            const float* kr = Bm + (u0 + j) * LD + hc;
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < DKMAX; c += 4)
              if (c < dk) { const f32x4 k4 = lds_ld4(kr + c); s += qh[c] * k4[0] + qh[c + 1] * k4[1] + qh[c + 2] * k4[2] + qh[c + 3] * k4[3]; }
            m = fmaxf(m, s);
          }
          for (int j = 0; j < kv; ++j) {
            const float* kr = Bm + (u0 + j) * LD + hc;
            const float* vr = A + (u0 + j) * LD + hc;
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < DKMAX; c += 4)
              if (c < dk) { const f32x4 k4 = lds_ld4(kr + c); s += qh[c] * k4[0] + qh[c + 1] * k4[1] + qh[c + 2] * k4[2] + qh[c + 3] * k4[3]; }
            const float pj = expf(s - m);
            z += pj;
#pragma unroll
            for (int c = 0; c < DKMAX; c += 4)
              if (c < dk) { const f32x4 v4 = lds_ld4(vr + c); oh[c] += pj * v4[0]; oh[c + 1] += pj * v4[1]; oh[c + 2] += pj * v4[2]; oh[c + 3] += pj * v4[3]; }
          }
        } else {
          for (int j = 0; j < kv; ++j) {
            const float* kr = Bm + (u0 + j) * LD + hc;
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < DKMAX; ++c)
              if (c < dk) s += qh[c] * kr[c];
            m = fmaxf(m, s);
          }
          for (int j = 0; j < kv; ++j) {
            const float* kr = Bm + (u0 + j) * LD + hc;
            const float* vr = A + (u0 + j) * LD + hc;
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < DKMAX; ++c)
              if (c < dk) s += qh[c] * kr[c];
            const float pj = expf(s - m);
            z += pj;
#pragma unroll
            for (int c = 0; c < DKMAX; ++c)
              if (c < dk) oh[c] += pj * vr[c];
          }
        }
        const float zi = (kv > 0) ? 1.0f / z : 0.f;
        __syncthreads();   // all reads of k (Bm) are done: Bm receives the attention output
#pragma unroll
        for (int c = 0; c < DKMAX; ++c)
          if (c < dk) Br[hc + c] = oh[c] * zi;
        if (H * dk < D) {  // padded channels of the operand image must be 0
          for (int c = H * dk + g; c < D; c += 4) Br[c] = 0.f;
        }
        // (the attention output rows are written and read back by the same wave: no barrier)
#pragma unroll
        for (int kk = 0; kk < NT; ++kk) o[kk] = lds_ld4(Br + 16 * kk + 4 * g);
      }
      if (wave_live) {
        // fc(o) + x -> LayerNorm                                      (transformer_module.py:99-101)
        gemm_rows<NT>(Lp.wfc, o, lane, [&](int ot, f32x4 acc) { y[ot] = acc + x[ot]; });
        masked_layernorm<NT>(y, Lp.ln1_g, Lp.ln1_b, P.ln_eps, d, g, valid);
        // FFN: w2(relu(w1 y + b1)) + b2 + y -> LayerNorm               (transformer_module.py:113-127)
        gemm_rows<NT>(Lp.w1, y, lane, [&](int ot, f32x4 acc) { o[ot] = relu4(acc + ld4(Lp.b1 + 16 * ot + 4 * g)); });
        gemm_rows<NT>(Lp.w2, o, lane, [&](int ot, f32x4 acc) { x[ot] = acc + ld4(Lp.b2 + 16 * ot + 4 * g) + y[ot]; });
        masked_layernorm<NT>(x, Lp.ln2_g, Lp.ln2_b, P.ln_eps, d, g, valid);
      }
    }
    // ---------------------------------------------------------------- sum over the node's slots -> out_sum[node, :]
    if (mfma_attn) {
      if (wave_live) {
        float* orow = S.out_sum + (int64_t)node * d;    // `node` is the same for the 16 rows of the tile
#pragma unroll
        for (int kk = 0; kk < NT; ++kk) {
          f32x4 s;
#pragma unroll
          for (int t = 0; t < 4; ++t) s[t] = tile_rowsum(valid ? x[kk][t] : 0.f);
          const int c = 16 * kk + 4 * g;
          if (li == 0 && unit_ok) {
            if ((d & 3) == 0) {
              if (c < d) *reinterpret_cast<float4*>(orow + c) = make_float4(s[0], s[1], s[2], s[3]);
            } else {
#pragma unroll
              for (int t = 0; t < 4; ++t)
                if (c + t < d) orow[c + t] = s[t];
            }
          }
        }
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < NT; ++kk) lds_st4(Br + 16 * kk + 4 * g, valid ? x[kk] : f32x4{0.f, 0.f, 0.f, 0.f});
      __syncthreads();
      if (valid && slot == 0) {
        float* orow = S.out_sum + (int64_t)node * d;
#pragma unroll
        for (int kk = 0; kk < NT; ++kk) {
          const int c = 16 * kk + 4 * g;
          f32x4 s = {0.f, 0.f, 0.f, 0.f};
          for (int j = 0; j < kv; ++j) s += lds_ld4(Bm + (u0 + j) * LD + c);
          if ((d & 3) == 0) {
            if (c < d) *reinterpret_cast<float4*>(orow + c) = make_float4(s[0], s[1], s[2], s[3]);
          } else {
#pragma unroll
            for (int t = 0; t < 4; ++t)
              if (c + t < d) orow[c + t] = s[t];
          }
        }
      }
    }
  }
}

template <int NT>
static int launch_rho(const RhoStruct& S, const sn_rho_params& P, int64_t bins_bound, hipStream_t st) {
  constexpr int LD = 16 * NT + 4;
  const size_t lds = (size_t)(2 * RHO_R * LD) * sizeof(float);
  static int cus = 0;
  if (cus == 0) {
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_rho_fused<NT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return fail(SN_ERR_LAUNCH, "sn_rho_fused_f32: cannot raise the dynamic LDS limit to %zu", lds);
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    cus = n > 0 ? n : 256;
  }
  int64_t grid = bins_bound < (int64_t)2 * cus ? bins_bound : (int64_t)2 * cus;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL((k_rho_fused<NT>), dim3((unsigned)grid), dim3(RHO_R * 4), lds, st, S, P);
  return SN_OK;
}

}  // namespace sn

using namespace sn;

extern "C" int sn_rho_fused_f32(const sn_rho_params* params, const float* x, const float* eigen_values,
                                const int32_t* graph_ptr, int64_t B, int64_t N, const sn_plan_bins* bins, int kmax,
                                int K, float* out_sum, void* stream) {
  SN_REQUIRE(params && x && graph_ptr && bins && bins->rho_bin0 && bins->meta && out_sum, "sn_rho_fused_f32: null pointer");
  const sn_rho_params& P = *params;
  SN_REQUIRE(P.d > 0 && P.d <= 128, "sn_rho_fused_f32: hidden width %d not in (0, 128]", P.d);
  SN_REQUIRE(P.heads == 4 && P.d % P.heads == 0, "sn_rho_fused_f32: needs 4 heads dividing d (got %d heads, d=%d)", P.heads, P.d);
  SN_REQUIRE(P.n_layers >= 0 && P.n_layers <= SN_RHO_MAX_LAYERS, "sn_rho_fused_f32: %d layers unsupported", P.n_layers);
  SN_REQUIRE(!P.has_pos || (eigen_values && P.pe_w1 && P.pe_bn0_scale && P.pe_bn0_shift && P.pe_w2 && P.pe_bn1_scale && P.pe_bn1_shift),
             "sn_rho_fused_f32: eigenvalue encoder parameters missing");
  for (int l = 0; l < P.n_layers; ++l) {
    const sn_rho_layer& L = P.layers[l];
    SN_REQUIRE(L.wq && L.wk && L.wv && L.wfc && L.ln1_g && L.ln1_b && L.w1 && L.b1 && L.w2 && L.b2 && L.ln2_g && L.ln2_b,
               "sn_rho_fused_f32: layer %d parameters missing", l);
  }
  SN_REQUIRE(K > 0 && B >= 0 && N >= 0 && B < (1ll << 31), "sn_rho_fused_f32: bad sizes");
  if (B == 0 || N == 0) return SN_OK;
  RhoStruct S{x, eigen_values, graph_ptr, bins->rho_bin0, bins->meta, (int)B, kmax, K, out_sum};
  hipStream_t st = (hipStream_t)stream;
  const int64_t bound = N + B;   // every bin holds at least one node
  int rc = SN_OK;
  switch ((P.d + 15) / 16) {
    case 1: rc = launch_rho<1>(S, P, bound, st); break;
    case 2: rc = launch_rho<2>(S, P, bound, st); break;
    case 3: rc = launch_rho<3>(S, P, bound, st); break;
    case 4: rc = launch_rho<4>(S, P, bound, st); break;
    case 5: rc = launch_rho<5>(S, P, bound, st); break;
    case 6: rc = launch_rho<6>(S, P, bound, st); break;
    case 7: rc = launch_rho<7>(S, P, bound, st); break;
    default: rc = launch_rho<8>(S, P, bound, st); break;
  }
  if (rc != SN_OK) return rc;
  SN_CHECK_LAUNCH("sn_rho_fused_f32");
  return SN_OK;
}
