// fused_gated.hip — the GatedGCN network of the DGL tree, every layer and the readout in ONE launch (eval mode).
// Replaces GatedGCNNet.forward from the first GatedGCNLayer on (GraphPrediction/nets/ZINC_graph_regression/gatedgcn_net.py:105-148)
// and GatedGCNLayer.forward (layers/gatedgcn_layer.py:36-81) with batch_norm, no dropout, no graph_norm:
//   Ah, Bh, Dh, Eh = A(h), B(h), D(h), E(h);  Ce = C(e)
//   e_ij = Dh_j + Eh_i + Ce_ij;  sigma = sigmoid(e_ij);  h_i = Ah_i + sum_j sigma_ij Bh_j / (sum_j sigma_ij + 1e-6)
//   h = h_in + relu(BN(h));  e = e_in + relu(BN(e))                                 (residual when the widths agree)
//   readout: mean / sum of h over the graph's nodes, then MLPReadout (layers/mlp_readout_layer.py: d -> d/2 -> d/4 -> 1)
// Like the GINE stage (fused_gnn.hip) this part of the model has few rows and a long dependent chain (16 layers x 3 phases), so it
// is latency-bound; run layer-at-a-time it is ~50 small launches.  Mapping: one workgroup of 4 waves per graph (n <= 64 nodes):
//   * wave w owns node rows 16w..16w+15: their h lives in registers (MFMA operand layout) for the whole network; A/B/D/E are ONE
//     [4*dp, dp] Linear through wg_gemm_split (fp32 on the bf16 matrix pipe, weights streamed through the LDS ring): Ah stays in
//     the owner's registers, Bh / Dh / Eh go to an LDS image Y [64][3*dp] for the gathers;
//   * edges are taken in destination-sorted CSR order, 64 per pass (wave w: 16 edge rows): e rows are read from / written back to
//     global memory by the same lanes (in place, no synchronisation needed), Ce is a second wg_gemm_split whose epilogue forms
//     e_ij, the gate (kept in an LDS image S [edges][dp]) and the new e;
//   * the aggregation walks a node's in-edges in CSR (edge-id) order from S and Y: no atomics, reproducible.
#include "fused_common.hpp"

namespace sn {

constexpr int GG_ROWS = 64;

template <int NT> struct GGCfg {
  static constexpr int D = 16 * NT;
  static constexpr int YLD = 3 * D + 4;                     // Bh | Dh | Eh
  static constexpr int SLD = D + 4;
  // What bounds this kernel (measured with clock64 stamps, profiles/README.md r02): the weight stream.  LDS-DMA delivers ~17 B/clk per
  // CU (a 12 KB chunk per ~700 cycles with the MFMAs and fragment reads removed; same with a 5-deep ring), and a workgroup has only
  // 16-64 rows to spend on every chunk.  Two alternatives were built and measured on the 128-graph, 16-layer, hidden-68 batch: reading
  // the next chunk's fragments during the current chunk's MFMAs (303 us vs 300), and every live wave streaming the matrices straight
  // into a three-chunk register ring with no barrier at all (361 us: each wave then pulls the whole matrix through the CU's L1).
  static constexpr int RING = WRing<NT>::BYTES;
  static constexpr int TAIL = 4 * D * 4 + 1024;              // readout scratch
  static constexpr int EMAX_RAW = (160 * 1024 - RING - GG_ROWS * YLD * 4 - TAIL) / (SLD * 4 + 6);
  static constexpr int EMAX = EMAX_RAW > 192 ? 192 : (EMAX_RAW / 16) * 16;
  static constexpr int BYTES = RING + GG_ROWS * YLD * 4 + EMAX * SLD * 4 + EMAX * 6 + TAIL + 64;
};

struct GatedStruct {
  float* h;               // [N, d]  in: the embedded node features (gatedgcn_net.py:93-103); not written
  float* e;               // [E, d]  in: embedded edge features; updated in place, layer by layer
  const int32_t* graph_ptr;
  const int32_t* rowptr;
  const int32_t* col;
  const int32_t* eperm;
  int32_t* status;        // status[3] |= 1 / 2: a graph with more than 64 nodes / more in-edges than the LDS image holds
  float* y;               // [B]
  int B;
};

__device__ __forceinline__ float gg_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

template <int NT>
__global__ __launch_bounds__(GG_ROWS * 4, 1) void k_gatedgcn_net(GatedStruct S, sn_gatedgcn_params P) {
  using Cfg = GGCfg<NT>;
  constexpr int D = Cfg::D, YLD = Cfg::YLD, SLD = Cfg::SLD, NKB = (NT + 1) / 2;
  using Ring = WRing<NT>;
  extern __shared__ __align__(1024) unsigned char lds_raw[];
  float* Y = reinterpret_cast<float*>(lds_raw + Ring::BYTES);       // [64][YLD]
  float* Sg = Y + GG_ROWS * YLD;                                    // [EMAX][SLD]   sigma of every in-edge
  float* red = Sg + Cfg::EMAX * SLD;                                // [4][D] + MLP scratch
  int32_t* eid = reinterpret_cast<int32_t*>(red + 4 * D + 256);     // [EMAX] global edge id of CSR position p
  unsigned char* srcl = reinterpret_cast<unsigned char*>(eid + Cfg::EMAX);   // [EMAX] local source node
  unsigned char* dstl = srcl + Cfg::EMAX;                           // [EMAX] local destination node
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, g = lane >> 4, r = wave * 16 + li;
  const int d = P.d;
  Ring ring;
  ring.init(lds_raw, wave, lane);
  bool started = false;
  // a discrete feature id outside its embedding table upstream (sn_embedding_sum_f32 raised status[5]; nn.Embedding raises
  // IndexError there): every score of the batch is NaN — nothing computed from a skipped table row is handed back
  const bool upstream_bad = S.status[5] != 0 || S.status[0] != 0;     // (status[0]: sn_batch_plan found the batch malformed)
  for (int gi = blockIdx.x; gi < S.B; gi += gridDim.x) {
    if (upstream_bad) {
      if (threadIdx.x == 0) S.y[gi] = __int_as_float(0x7fc00000);
      continue;
    }
    const int gs = S.graph_ptr[gi], n = S.graph_ptr[gi + 1] - gs;
    const int p0 = S.rowptr[gs], ne = S.rowptr[gs + n] - p0;
    if (n > GG_ROWS || ne > Cfg::EMAX) {          // not evaluable here: flagged, NaN handed back (the host uses the layer path)
      if (threadIdx.x == 0) {
        atomicOr(&S.status[3], n > GG_ROWS ? 1 : 2);
        S.y[gi] = __int_as_float(0x7fc00000);
      }
      continue;
    }
    __syncthreads();     // the previous graph is done with the tables
    for (int i = threadIdx.x; i < n; i += GG_ROWS * 4)
      for (int p = S.rowptr[gs + i] - p0; p < S.rowptr[gs + i + 1] - p0; ++p) dstl[p] = (unsigned char)i;
    for (int p = threadIdx.x; p < ne; p += GG_ROWS * 4) { srcl[p] = (unsigned char)(S.col[p0 + p] - gs); eid[p] = S.eperm[p0 + p]; }
    if (!started) { ring.prologue(P.layers[0].wabde, NT); started = true; }
    __syncthreads();
    SN_PROF_ON(true);
#ifdef SN_PROFILE
    if (blockIdx.x == 0 && threadIdx.x == 0) { for (int i = 1; i <= 7; ++i) g_prof[i] = 0; g_prof[10] = n; g_prof[11] = ne; }
#endif
    SN_STAMP(8);
    const bool valid = r < n;
    const bool wave_live = __ballot(valid) != 0ull;
    const int npass = (ne + GG_ROWS - 1) / GG_ROWS;

    f32x4 h[NT];
    {
      const float* hr = S.h + (valid ? (int64_t)(gs + r) : (int64_t)0) * d;
#pragma unroll
      for (int kk = 0; kk < NT; ++kk) {
        const int c = 16 * kk + 4 * g;
        const f32x4 v = ld4(hr + (c < d ? c : 0));
        h[kk] = (valid && c < d) ? v : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    const int e_lo = valid ? S.rowptr[gs + r] - p0 : 0, e_hi = valid ? S.rowptr[gs + r + 1] - p0 : 0;
#pragma unroll 1
    for (int l = 0; l < P.n_layers; ++l) {
      const sn_gatedgcn_layer& Lp = P.layers[l];
      const void* wnext_layer = (l + 1 < P.n_layers) ? P.layers[l + 1].wabde : P.layers[0].wabde;   // the next graph restarts at layer 0
      // ---------------------------------------------------------------- A | B | D | E
      f32x4 Ah[NT];
#ifdef SN_PROFILE
      long long pt = clock64();
#endif
      {
        Split8 sp[NKB];
        if (wave_live) split_rows<NT>(h, sp);
        float* Yr = Y + r * YLD + 4 * g;
        wg_gemm_split<NT, 4 * NT, false>(ring, Lp.wabde, npass > 0 ? Lp.wc : wnext_layer, wave_live, sp, NoPre(),
                                         [&](int ot, f32x4 acc, f32x4 b, f32x4, f32x4, f32x4) {
                                           const f32x4 v = acc + b;
                                           if (ot < NT) Ah[ot] = v; else lds_st4(Yr + 16 * (ot - NT), v);
                                         });
      }
      SN_ACCUM(1, pt);
#ifdef SN_PROFILE
      pt = clock64();
#endif
      lds_barrier();
      SN_ACCUM(2, pt);
#ifdef SN_PROFILE
      pt = clock64();
      long long pload = 0, pgemm = 0;
#endif
      // ---------------------------------------------------------------- C, the gates and the new edge features
#pragma unroll 1
      for (int ep = 0; ep < npass; ++ep) {
        const int p = ep * GG_ROWS + r;
        const bool ev = p < ne;
        const bool elive = __ballot(ev) != 0ull;
        float* erow = S.e + (int64_t)(ev ? eid[p] : eid[0]) * d;
        f32x4 ein[NT];
#pragma unroll
        for (int kk = 0; kk < NT; ++kk) {
          const int c = 16 * kk + 4 * g;
          const f32x4 v = ld4(erow + (c < d ? c : 0));
          ein[kk] = (ev && c < d) ? v : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#ifdef SN_PROFILE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SN_ACCUM(6, pt);
        long long pt2 = clock64();
#endif
        const float* Ysrc = Y + (ev ? srcl[p] : 0) * YLD + D + 4 * g;         // Dh of the source
        const float* Ydst = Y + (ev ? dstl[p] : 0) * YLD + 2 * D + 4 * g;     // Eh of the destination
        float* Sr = Sg + (ev ? p : 0) * SLD + 4 * g;
        const bool more = ep + 1 < npass;
        Split8 sp[NKB];
        if (elive) split_rows<NT>(ein, sp);
        wg_gemm_split<NT, NT, false>(ring, Lp.wc, more ? Lp.wc : wnext_layer, elive, sp, NoPre(),
                                     [&](int ot, f32x4 acc, f32x4 b, f32x4 es, f32x4 et, f32x4) {
                                       const int c = 16 * ot + 4 * g;
                                       const f32x4 ce = acc + b;
                                       const f32x4 dh = lds_ld4(Ysrc + 16 * ot), eh = lds_ld4(Ydst + 16 * ot);
                                       f32x4 sg, eo;
#pragma unroll
                                       for (int t = 0; t < 4; ++t) {
                                         const float en = (dh[t] + eh[t]) + ce[t];
                                         sg[t] = gg_sigmoid(en);
                                         eo[t] = fmaxf(en * es[t] + et[t], 0.f);
                                       }
                                       if (Lp.residual) eo += ein[ot];
                                       if (ev) {
                                         lds_st4(Sr + 16 * ot, sg);
                                         if (c < d) *reinterpret_cast<float4*>(erow + c) = make_float4(eo[0], eo[1], eo[2], eo[3]);
                                       }
                                     });
#ifdef SN_PROFILE
        SN_ACCUM(7, pt2);
        pt = clock64();
#endif
      }
      SN_ACCUM(3, pt);
#ifdef SN_PROFILE
      pt = clock64();
#endif
      lds_barrier();
      SN_ACCUM(4, pt);
#ifdef SN_PROFILE
      pt = clock64();
#endif
      // ---------------------------------------------------------------- aggregation, BatchNorm, ReLU, residual
      if (valid) {
        f32x4 num[NT], den[NT], hsv[NT], htv[NT];
#pragma unroll
        for (int kk = 0; kk < NT; ++kk) {        // the node BatchNorm of this layer: in flight during the in-edge walk
          hsv[kk] = ld4(Lp.h_scale + 16 * kk + 4 * g);
          htv[kk] = ld4(Lp.h_shift + 16 * kk + 4 * g);
        }
#pragma unroll
        for (int kk = 0; kk < NT; ++kk) { num[kk] = f32x4{0.f, 0.f, 0.f, 0.f}; den[kk] = num[kk]; }
        for (int p = e_lo; p < e_hi; ++p) {
          const float* sr = Sg + p * SLD + 4 * g;
          const float* br = Y + srcl[p] * YLD + 4 * g;
#pragma unroll
          for (int kk = 0; kk < NT; ++kk) {
            const f32x4 s4 = lds_ld4(sr + 16 * kk), b4 = lds_ld4(br + 16 * kk);
#pragma unroll
            for (int t = 0; t < 4; ++t) { num[kk][t] += b4[t] * s4[t]; den[kk][t] += s4[t]; }
          }
        }
#pragma unroll
        for (int kk = 0; kk < NT; ++kk) {
          const f32x4 hs = hsv[kk], ht = htv[kk];
          f32x4 o;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const float hn = Ah[kk][t] + num[kk][t] / (den[kk][t] + 1e-6f);
            o[t] = fmaxf(hn * hs[t] + ht[t], 0.f);
          }
          h[kk] = Lp.residual ? h[kk] + o : o;
        }
      }
      lds_barrier();       // Y and S are free for the next layer
      SN_ACCUM(5, pt);
    }
    SN_STAMP(9);
    // ------------------------------------------------------------------ readout: mean / sum over the nodes, MLPReadout
    {
#pragma unroll
      for (int kk = 0; kk < NT; ++kk) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          float v = valid ? h[kk][t] : 0.f;
          v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);   // the tile's 16 rows, in a fixed order
          if (li == 0) red[wave * D + 16 * kk + 4 * g + t] = v;
        }
      }
      __syncthreads();
      float* v0 = red + 4 * D;          // [<= 128] pooled / hidden vectors of the readout MLP
      float* v1 = v0 + 128;
      if (threadIdx.x < D) {
        const int c = threadIdx.x;
        float s = (red[c] + red[D + c]) + (red[2 * D + c] + red[3 * D + c]);
        if (P.readout_mean) s = s / (float)(n > 0 ? n : 1);
        v0[c] = s;
      }
      __syncthreads();
      const int d1 = P.ro_d1, d2 = P.ro_d2;
      if ((int)threadIdx.x < d1) {
        float a = P.ro_b0[threadIdx.x];
        const float* w = P.ro_w0 + (int64_t)threadIdx.x * P.d_out;
        for (int c = 0; c < P.d_out; ++c) a += w[c] * v0[c];
        v1[threadIdx.x] = fmaxf(a, 0.f);
      }
      __syncthreads();
      if ((int)threadIdx.x < d2) {
        float a = P.ro_b1[threadIdx.x];
        const float* w = P.ro_w1 + (int64_t)threadIdx.x * d1;
        for (int c = 0; c < d1; ++c) a += w[c] * v1[c];
        v0[threadIdx.x] = fmaxf(a, 0.f);
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        float a = P.ro_b2[0];
        for (int c = 0; c < d2; ++c) a += P.ro_w2[c] * v0[c];
        S.y[gi] = a;
      }
    }
  }
  if (started) ring.drain();
}

template <int NT>
static int launch_gated(const GatedStruct& S, const sn_gatedgcn_params& P, hipStream_t st) {
  const size_t lds = (size_t)GGCfg<NT>::BYTES;
  static int cus = 0;
  if (cus == 0) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_gatedgcn_net<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return fail(SN_ERR_LAUNCH, "sn_gatedgcn_fused_f32: cannot raise the dynamic LDS limit to %zu", lds);
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    cus = n > 0 ? n : 256;
  }
  const int grid = S.B < cus ? S.B : cus;
  hipLaunchKernelGGL((k_gatedgcn_net<NT>), dim3((unsigned)grid), dim3(GG_ROWS * 4), lds, st, S, P);
  return SN_OK;
}

}  // namespace sn

using namespace sn;

#ifdef SN_PROFILE
extern "C" int sn_prof_read_gated(long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_prof), sizeof(long long) * 64); }
#endif

extern "C" int sn_gatedgcn_max_edges(int d) {
  const int nt = (d + 15) / 16;
  switch (nt < 3 ? 3 : nt) {
    case 3: return GGCfg<3>::EMAX;
    case 4: return GGCfg<4>::EMAX;
    case 5: return GGCfg<5>::EMAX;
    case 6: return GGCfg<6>::EMAX;
    default: return 0;
  }
}

extern "C" int sn_gatedgcn_fused_f32(const sn_gatedgcn_params* params, const float* h, float* e, const int32_t* graph_ptr, int64_t B,
                                     const int32_t* rowptr, const int32_t* col, const int32_t* eperm, int32_t* status, float* y,
                                     void* stream) {
  SN_REQUIRE(params && h && graph_ptr && rowptr && status && y && B >= 0, "sn_gatedgcn_fused_f32: null pointer");
  const sn_gatedgcn_params& P = *params;
  SN_REQUIRE(P.d >= 4 && P.d <= 96 && (P.d & 3) == 0, "sn_gatedgcn_fused_f32: hidden width %d must be a multiple of 4 in [4, 96]", P.d);
  const int nt = (P.d + 15) / 16 < 3 ? 3 : (P.d + 15) / 16;       // widths below 48 run zero-padded on the 48-wide instantiation
  SN_REQUIRE(P.d_out >= 1 && P.d_out <= 16 * nt, "sn_gatedgcn_fused_f32: bad output width");
  SN_REQUIRE(P.n_layers >= 1 && P.n_layers <= SN_GATED_MAX_LAYERS, "sn_gatedgcn_fused_f32: %d layers unsupported", P.n_layers);
  SN_REQUIRE(P.ro_w0 && P.ro_b0 && P.ro_w1 && P.ro_b1 && P.ro_w2 && P.ro_b2 && P.ro_d1 >= 1 && P.ro_d1 <= 128 && P.ro_d2 >= 1 && P.ro_d2 <= 128,
             "sn_gatedgcn_fused_f32: readout MLP missing or wider than 128");
  for (int l = 0; l < P.n_layers; ++l)
    SN_REQUIRE(P.layers[l].wabde && P.layers[l].wc && P.layers[l].h_scale && P.layers[l].h_shift, "sn_gatedgcn_fused_f32: layer %d parameters missing", l);
  SN_REQUIRE(B <= 0x7fffffff, "sn_gatedgcn_fused_f32: too many graphs");
  if (B == 0) return SN_OK;
  SN_REQUIRE(e && col && eperm, "sn_gatedgcn_fused_f32: null edge arrays");
  GatedStruct S{const_cast<float*>(h), e, graph_ptr, rowptr, col, eperm, status, y, (int)B};
  hipStream_t st = (hipStream_t)stream;
  int rc = SN_OK;
  switch (nt) {
    case 3: rc = launch_gated<3>(S, P, st); break;
    case 4: rc = launch_gated<4>(S, P, st); break;
    case 5: rc = launch_gated<5>(S, P, st); break;
    default: rc = launch_gated<6>(S, P, st); break;
  }
  if (rc != SN_OK) return rc;
  SN_CHECK_LAUNCH("sn_gatedgcn_fused_f32");
  return SN_OK;
}
