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
//     chained wg_gemm_split() calls (fp32 via the bf16 matrix pipe, fused_common.hpp: weights staged once per
//     workgroup in an LDS ring by LDS-DMA) — residuals and LayerNorm run on the accumulators;
//   * attention: q, k, v rows are exchanged through two LDS images; lane (row, head) computes its
//     query's scores against the node's keys (two passes: max, then exp / sum / P.V), exactly
//     softmax(q k^T / sqrt(dk)) restricted to the node's valid slots (transformer_module.py:52-57);
//   * the final sum over slots is done from LDS by each node's first row.
// Bound: matrix pipe (6 bf16 partial products per fp32 product); flops per valid row and layer: 6 * 2*d*d (+ 4*K*d attention).
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
  const int32_t* node_graph;  // [N] graph of every node (register-attention variants: node-major bins)
  int N;
  // the planner's bin member records (sn_plan_bins.phi_bin_mem): with all eigenvectors (kmax == 0) a node's slot rows are as many as
  // its graph has nodes, so phi's slab bins are rho bins too — record (graph, index, offset) = the n slot rows of node `index` of the
  // graph at bin row `offset` (k_rho_wide, COLS)
  const int32_t* bin_mem;
};

// HP (head-padded layout, see sn_rho_params.head_pad): the tile count exceeds ceil(d/16), so every tile from the one holding
// channel d on is (partly) padding and gets its own mask; otherwise only the last tile can be partial.
template <int NT, bool HP = false>
__device__ __forceinline__ void masked_layernorm(f32x4 (&v)[NT], const float* gamma /* LDS */, const float* beta /* LDS */,
                                                 float eps, int d, int g, bool valid) {
  float s = 0.f;
#pragma unroll
  for (int kk = 0; kk < NT; ++kk) s += (v[kk][0] + v[kk][1]) + (v[kk][2] + v[kk][3]);   // padded channels hold 0
  const float mean = row_allsum(s) / (float)d;
  // d % 4 == 0: whole float4 groups are either real channels or padding, and only the last 16-channel tile can hold
  // padding — ONE lane mask instead of one per element (32 of them were precomputed into 64 scalar registers)
  const bool last_ok = 16 * (NT - 1) + 4 * g < d;
  float q = 0.f;
#pragma unroll
  for (int kk = 0; kk < NT; ++kk) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const bool real = HP ? (16 * kk + 4 * g < d) : (kk + 1 < NT || last_ok);
      const float dlt = real ? v[kk][t] - mean : 0.f;
      q += dlt * dlt;
    }
  }
  const float rstd = 1.0f / sqrtf(row_allsum(q) / (float)d + eps);
#pragma unroll
  for (int kk = 0; kk < NT; ++kk) {
    const int c = 16 * kk + 4 * g;
    const f32x4 ga = lds_ld4(gamma + c), be = lds_ld4(beta + c);   // zero padded -> padded channels stay 0
    f32x4 o;
#pragma unroll
    for (int t = 0; t < 4; ++t) o[t] = valid ? (v[kk][t] - mean) * rstd * ga[t] + be[t] : 0.f;
    v[kk] = o;
  }
}

__device__ __forceinline__ float group_allmax(float v) {   // over the 4 lane groups holding one row
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
// sum over the 16 rows (lanes l&15) of a tile, same lane group = one DPP row: four VALU+DPP steps, no LDS crossbar
__device__ __forceinline__ float dpp_add(float v, int ctrl_sel) {
  const int x = __float_as_int(v);
  int y;
  switch (ctrl_sel) {
    case 0: y = __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, true); break;    // quad_perm [1,0,3,2]
    case 1: y = __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, true); break;    // quad_perm [2,3,0,1]
    case 2: y = __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, true); break;   // row_half_mirror
    default: y = __builtin_amdgcn_update_dpp(0, x, 0x140, 0xf, 0xf, true); break;  // row_mirror
  }
  return v + __int_as_float(y);
}
__device__ __forceinline__ float tile_rowsum(float v) {
  v = dpp_add(v, 0);
  v = dpp_add(v, 1);
  v = dpp_add(v, 2);
  return dpp_add(v, 3);
}

// REGATTN: every node has <= 16 valid slots (0 < kmax <= 16) and the head width is a multiple of 16 -> attention in
// registers, no LDS images (the host picks the variant; the other one serves any slot count through LDS).
// ONE: exactly one encoder layer without eigenvalue encoding (GINESignNetPyG: nl_rho is ignored, 1 layer).  The layer's
// input rows are then still in the input buffer when the residual needs them, so they are not kept in registers across
// the q / k / v projections and the attention but re-read right before the output projection.
// NW (round 6; REGATTN only): waves per workgroup = nodes per bin.  The hypothesis: rho is bound by its weight stream — a layer is six
// split-packed [d, d] matrices = 720 KB of LDS-DMA per bin at d = 128, two 4-wave workgroups per CU = two streams; the register-attention
// variants keep their rows in registers, so ONE 8-wave workgroup per CU could hold the same 8 nodes behind ONE stream.  Built (NW = 8),
// bit-identical outputs, and measured SLOWER (61.5 -> 72.9 us on the headline batch): see rho_eight_waves() below.  Default: 4.
template <int NT, bool REGATTN, bool ONE, bool HP = false, int NW = 4>
// (the LDS-attention variants hold two 64-row images beside the weight ring: one workgroup per CU fits, so they may use the whole
//  register file of a SIMD — 512 registers per lane, no private segment)
__global__ __launch_bounds__(64 * NW, REGATTN ? 2 : 1) void k_rho_fused(RhoStruct S, sn_rho_params P) {
  static_assert(NW == 4 || (NW == 8 && REGATTN), "eight waves: the register-attention variants only (no 64-row LDS images)");
  constexpr int NTHR = 64 * NW;
  constexpr int D = 16 * NT;
  constexpr int LD = D + 4;
  constexpr int DKMAX = (D + 3) / 4;   // heads = 4: dk <= D/4
  constexpr int NKB = (NT + 1) / 2;
  using Ring = WRing<NT, NW>;
  extern __shared__ __align__(1024) unsigned char lds_raw[];
  float* A = reinterpret_cast<float*>(lds_raw + Ring::BYTES);   // [RHO_R][LD]   q, then v          (LDS attention path only)
  float* Bm = A + RHO_R * LD;                                     // [RHO_R][LD]   k, then the attention output / slot-sum image
  __shared__ int s_graph;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = wave * 16 + (lane & 15), g = lane >> 4, li = lane & 15;
  const int nbins = REGATTN ? (S.N + NW - 1) / NW : S.meta[4];
  const int d = P.d, H = P.heads, dk = d / H;
  // Round 5: the start of a workgroup used to be six dependent memory round trips (error word -> LayerNorm vectors -> first weight
  // chunk -> node's graph -> its node range -> the rows), ~1 us each, in front of the first MFMA of a kernel of two bins per
  // workgroup.  With node-major bins (REGATTN) nothing but the node range depends on an earlier load: the error word, the first
  // bin's graph id, its rows (node and slot are arithmetic; the rows of slots past the graph's count are read — they exist, x is
  // dense [N, K, d] — and zeroed once the count is known), the LayerNorm vectors and the weight stream are all requested up front.
  f32x4 x[NT];
  int pf_gs = 0, pf_n = 0;                      // node range of the NEXT bin's graph (this wave's node), fetched a bin ahead
  // (pure loads, no selects on the values: a select would make the compiler wait for the rows right here; the masks — slot count,
  //  channel padding — are applied at the top of the bin)
  auto fetch_rows = [&](int node_) {
    int slot_ = lane & 15, g_ = g;
    asm volatile("" : "+v"(slot_), "+v"(g_));
    const bool ok = node_ < S.N && slot_ < S.K;
    const float* xr = S.x + (ok ? ((int64_t)node_ * S.K + slot_) : (int64_t)0) * d;
#pragma unroll
    for (int kk = 0; kk < NT; ++kk) {
      const int c = 16 * kk + 4 * g_;
      const bool inb = HP ? c < d : (kk + 1 < NT || c < d);
      x[kk] = ld4(xr + (inb ? c : 0));
    }
  };
  // graph id -> node range of a wave's node: two dependent SCALAR loads written as such.  (Left to the compiler they are scalar only
  // in front of the first LDS-DMA — behind anything that may write memory a uniform load is no longer provably invariant and becomes a
  // vector load, whose wait (the memory counter is in order) then includes every weight chunk in flight.)  Each load and its wait are
  // ONE asm statement: the compiler does not know an asm's output may still be in flight, it could copy or spill the register before
  // the value has arrived.  Called behind the request for the node's rows, whose latency covers the two scalar round trips.
  auto fetch_graph = [&](int node_) {
    const int32_t* p = S.node_graph + (node_ < S.N ? node_ : 0);
    int gi;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(gi) : "s"(p) : "memory");
    const int32_t* q = S.graph_ptr + gi;
    long long rng;
    asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rng) : "s"(q) : "memory");
    pf_gs = (int)(rng & 0xffffffffll);
    pf_n = (int)(rng >> 32) - pf_gs;
  };
  // (not for the multi-layer variants at d = 128: their row array is live through the whole layer and the kernel sits at 256
  //  registers — with the rows requested a bin ahead 12-20 bytes per lane went to a private segment)
#ifdef SN_RHO_OLDSTART            // (A/B builds: the start as it was until round 5 — six dependent round trips)
  constexpr bool EARLY = false;
#else
  constexpr bool EARLY = REGATTN && (ONE || NT < 8);
#endif
  const int err = S.meta[5];
  const float temp = sqrtf((float)dk);
  const float rtemp = 1.0f / temp;
  { SN_PROF_ON(true); SN_STAMP(13); }
  // LayerNorm gamma / beta of every layer, staged once: read between GEMMs they would otherwise wait (vmcnt 0) behind
  // the weight stream's in-flight LDS-DMA plus their own L2 latency, twice per layer
  float* lnv = reinterpret_cast<float*>(lds_raw + Ring::BYTES) + (REGATTN ? 0 : 2 * RHO_R * LD);   // [n_layers][4][D]
  Ring ring;
  ring.init(lds_raw, wave, lane);
  const void* wfirst = P.n_layers > 0 ? P.layers[0].wq : nullptr;
  const bool stream = NT >= SPLIT_RING && wfirst != nullptr && nbins > (int)blockIdx.x;
  if constexpr (EARLY && ONE && NT >= SPLIT_RING) {
    // everything the first bin needs, in ONE round trip: rows, the layer's four LayerNorm vectors (one float4 per thread; the
    // pointers are kernel arguments), the first RING weight chunks (nobody has touched the ring yet: no barrier in front of the
    // issue), graph id -> node range (the only dependent load).  Straight-line code: behind a branch the compiler can no longer count
    // the loads in flight and waits for all of them (vmcnt(0): the whole ring) before the first use of any.
    if (nbins <= (int)blockIdx.x) return;      // (a workgroup without a bin: nothing requested yet)
    const int node0 = __builtin_amdgcn_readfirstlane((int)blockIdx.x * NW + wave);
    fetch_rows(node0);
    static_assert(4 * (D / 4) <= NTHR, "one float4 per thread covers the four vectors");
    const int li4 = (int)threadIdx.x < 4 * (D / 4) ? (int)threadIdx.x : 0;
    const sn_rho_layer& L0 = P.layers[0];
    const float *p0 = L0.ln1_g, *p1 = L0.ln1_b, *p2 = L0.ln2_g, *p3 = L0.ln2_b;
    asm volatile("" : "+s"(p0), "+s"(p1), "+s"(p2), "+s"(p3));      // (four scalar pointers: as a lane-indexed read of the argument block
                                                                      //  the choice below is a dependent load in front of the vector's)
    const int lv = li4 / (D / 4), lc = 4 * (li4 - lv * (D / 4));
    typedef __attribute__((address_space(1))) const float gfloat_t;      // (laundered pointers are generic: a flat load otherwise)
    gfloat_t* src = (gfloat_t*)(lv == 0 ? p0 : (lv == 1 ? p1 : (lv == 2 ? p2 : p3)));
    const f32x4 lnr = f32x4{src[lc], src[lc + 1], src[lc + 2], src[lc + 3]};
    ring.pos = 0;
#pragma unroll
    for (int c = 0; c < Ring::DEPTH; ++c) ring.issue(wfirst, c, c);
    fetch_graph(node0);
    {
      // (the store as an instruction: for an LDS store the compiler can see it waits for every LDS-DMA in flight — it cannot tell the
      //  ring from the vectors' rows — i.e. for all RING chunks instead of for the one float4 in front of them)
      typedef __attribute__((address_space(3))) float lds_float_t;
      const unsigned la = (unsigned)(size_t)(lds_float_t*)(lnv + 4 * li4);
      asm volatile("ds_write_b128 %0, %1" :: "v"(la), "v"(lnr) : "memory");     // (threads past the vectors repeat thread 0's float4)
    }
    ring.template wait_shares<Ring::DEPTH - 1>();
    lds_barrier();
    if (err != 0) { ring.drain(); return; }
  } else {
    if constexpr (EARLY) {
      if (nbins > (int)blockIdx.x) {
        const int node0 = __builtin_amdgcn_readfirstlane((int)blockIdx.x * NW + wave);
        fetch_rows(node0);
        fetch_graph(node0);
      }
    } else {
      if (err != 0) return;
    }
    for (int i = threadIdx.x; i < P.n_layers * 4 * D; i += NTHR) {
      const int l = i / (4 * D), v = (i / D) & 3, c = i % D;
      const sn_rho_layer& Lq = P.layers[l];
      const float* src = v == 0 ? Lq.ln1_g : (v == 1 ? Lq.ln1_b : (v == 2 ? Lq.ln2_g : Lq.ln2_b));
      lnv[i] = src[c];
    }
    __syncthreads();
    if (stream) ring.prologue(wfirst, NT);
    if constexpr (EARLY) {
      if (err != 0) { ring.drain(); return; }
    }
  }

  for (int bin = blockIdx.x; bin < nbins; bin += gridDim.x) {
    // ---------------------------------------------------------------- bin -> graph (bins never mix graphs)
    SN_PROF_ON(bin == (int)blockIdx.x);
    SN_STAMP(0);
    int gs, n, kg, q, slot, node, kv, u0;
    bool unit_ok;
    if constexpr (REGATTN) {
      // node-major bins: every node owns one 16-row tile (all K_g <= 16), four consecutive nodes of the BATCH per bin, graphs
      // mixed freely — ceil(N/4) bins instead of sum_g ceil(n_g/4) (6 % fewer on ZINC sizes), no bin -> graph search
      q = wave;
      slot = lane & 15;
      node = __builtin_amdgcn_readfirstlane(bin * NW + wave);
      unit_ok = node < S.N;
      if constexpr (EARLY) {
        gs = pf_gs;                                                  // (read at the end of the previous bin / at the kernel's start)
        n = pf_n;
      } else {
        const int gi = unit_ok ? S.node_graph[node] : 0;
        gs = S.graph_ptr[gi];
        n = S.graph_ptr[gi + 1] - gs;
      }
      kg = (S.kmax > 0 && n > S.kmax) ? S.kmax : n;
      kv = unit_ok ? kg : 0;
      u0 = q * 16;
    } else {
      __syncthreads();
      for (int gq = threadIdx.x; gq < S.B; gq += NTHR)
        if (S.rho_bin0[gq] <= bin && bin < S.rho_bin0[gq + 1]) s_graph = gq;
      __syncthreads();
      const int gi = s_graph;
      gs = S.graph_ptr[gi];
      n = S.graph_ptr[gi + 1] - gs;
      kg = (S.kmax > 0 && n > S.kmax) ? S.kmax : n;                // valid slots of every node of this graph
      const int pad = ((kg + 15) >> 4) << 4;                        // rows reserved per node (tile aligned)
      const int upb = RHO_R / pad;                                   // nodes per bin
      q = r / pad;
      slot = r - q * pad;
      const int u = (bin - S.rho_bin0[gi]) * upb + q;                // node index inside the graph
      unit_ok = q < upb && u < n;                                    // rows past upb*pad are padding
      node = gs + u;
      kv = unit_ok ? kg : 0;
      u0 = q * pad;                                                  // bin row of my node's slot 0
    }
    const bool valid = unit_ok && slot < kg;
    constexpr bool mfma_attn = REGATTN;                             // one node == one wave tile: attention in registers
    const bool wave_live = __ballot(valid) != 0ull;
    float* Ar = A + r * LD;
    float* Br = Bm + r * LD;
    SN_STAMP(1);
    // ---------------------------------------------------------------- load x (+ eigenvalue encoding)
    // branch-free (clamped addresses + selects): per-element branches here would turn x[] into a web of phi copies
    auto load_x = [&]() {
      // (slot and g are laundered: otherwise the compiler keeps the 64-bit row / column offsets of every tile — loop-invariant lane
      //  constants — alive across the whole bin loop, and in the 256-register multi-layer variants they were what spilled)
      int slot_ = slot, g_ = g;
      asm volatile("" : "+v"(slot_), "+v"(g_));
      const float* xr = S.x + (valid ? ((int64_t)node * S.K + slot_) : (int64_t)0) * d;
#pragma unroll
      for (int kk = 0; kk < NT; ++kk) {      // d % 4 == 0 (entry-point requirement); only the last tile can be partial
        const int c = 16 * kk + 4 * g_;
        const bool inb = HP ? c < d : (kk + 1 < NT || c < d);
        const f32x4 v = ld4(xr + (inb ? c : 0));
        x[kk] = (valid && inb) ? v : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    };
    if constexpr (EARLY) {
      // (the rows were requested before the node's slot count was known: slots past it read as zero)
#pragma unroll
      for (int kk = 0; kk < NT; ++kk) {
        const bool inb = HP ? (16 * kk + 4 * g < d) : (kk + 1 < NT || 16 * kk + 4 * g < d);
        x[kk] = (valid && inb) ? x[kk] : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    } else {
      load_x();
    }
    if (valid) {
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
    SN_STAMP(12);
#ifdef SN_PROFILE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    SN_STAMP(2);
    // ---------------------------------------------------------------- encoder layers
#pragma unroll 1
    for (int l = 0; l < (ONE ? 1 : P.n_layers); ++l) {
      const sn_rho_layer& Lp = P.layers[l];
      const void* wafter = (l + 1 < P.n_layers) ? P.layers[l + 1].wq : wfirst;   // the stream restarts at layer 0 for the next bin
      f32x4 o[NT];
      Split8 sp[NKB];
      if (wave_live) split_rows<NT>(x, sp);
      if (mfma_attn) {
        // ======== attention in registers (K_g <= 16: the node's slots are exactly this wave's 16 rows) ========
        // q, then k with the scores S^T = K.Q^T accumulated head by head in k's epilogue (k is never stored), softmax,
        // then v (swapped operands: V[key = 4g+r][c = 16ot + li]) with O^T = V^T.P^T in its epilogue (v is never stored).
        constexpr int CPH = NT / 4 > 0 ? NT / 4 : 1;   // 16-channel chunks per head
        f32x4 qf[NT], sc[4];
        wg_gemm_split<NT, NT, false, false, NW>(ring, Lp.wq, Lp.wk, wave_live, sp, NoPre(), [&](int ot, f32x4 acc, f32x4, f32x4, f32x4, f32x4) { qf[ot] = acc * rtemp; });   // q / sqrt(dk)  (:52), as a multiply by the rounded reciprocal (<= 1 ulp)
#pragma unroll
        for (int h = 0; h < 4; ++h) sc[h] = f32x4{0.f, 0.f, 0.f, 0.f};
        wg_gemm_split<NT, NT, false, false, NW>(ring, Lp.wk, Lp.wv, wave_live, sp, NoPre(), [&](int ot, f32x4 kf, f32x4, f32x4, f32x4, f32x4) {
          // lane (query = li, g) accumulates S[query][key = 4g + r] of head ot / CPH
          const int h = ot / CPH < 4 ? ot / CPH : 3;
          sc[h] = mfma16(kf[0], qf[ot][0], sc[h]);
          sc[h] = mfma16(kf[1], qf[ot][1], sc[h]);
          sc[h] = mfma16(kf[2], qf[ot][2], sc[h]);
          sc[h] = mfma16(kf[3], qf[ot][3], sc[h]);
        });
        SN_STAMP(3);
        if (wave_live) {
#pragma unroll
          for (int h = 0; h < 4; ++h) {
            float m = -INFINITY;
#pragma unroll
            for (int t = 0; t < 4; ++t) { if (4 * g + t >= kv) sc[h][t] = -INFINITY; m = fmaxf(m, sc[h][t]); }
            m = group_allmax(m);
            float z = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) { sc[h][t] = (4 * g + t < kv) ? expf(sc[h][t] - m) : 0.f; z += sc[h][t]; }
            z = row_allsum(z);
            const float zi = (kv > 0) ? 1.0f / z : 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) sc[h][t] *= zi;     // P[query][key = 4g + t]
          }
        }
        SN_STAMP(4);
        wg_gemm_split<NT, NT, true, false, NW>(ring, Lp.wv, Lp.wfc, wave_live, sp, NoPre(), [&](int ot, f32x4 vt, f32x4, f32x4, f32x4, f32x4) {
          // O^T[c][query] = sum_key V[key][c] P[query][key]  -> lane (query, g) holds O[query][16*ot + 4g + r]
          const int h = ot / CPH < 4 ? ot / CPH : 3;
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          acc = mfma16(vt[0], sc[h][0], acc);
          acc = mfma16(vt[1], sc[h][1], acc);
          acc = mfma16(vt[2], sc[h][2], acc);
          acc = mfma16(vt[3], sc[h][3], acc);
          o[ot] = acc;
        });
      } else {
        // ======== attention through LDS (nodes of more than 16 slots span several waves' tiles) ========
        wg_gemm_split<NT, NT, false, false, NW>(ring, Lp.wq, Lp.wk, wave_live, sp, NoPre(), [&](int ot, f32x4 acc, f32x4, f32x4, f32x4, f32x4) { lds_st4(Ar + 16 * ot + 4 * g, acc); });
        wg_gemm_split<NT, NT, false, false, NW>(ring, Lp.wk, Lp.wv, wave_live, sp, NoPre(), [&](int ot, f32x4 acc, f32x4, f32x4, f32x4, f32x4) { lds_st4(Br + 16 * ot + 4 * g, acc); });
        lds_barrier();
        float qh[DKMAX];
        const int hc = g * dk;
        constexpr int DKF = D / 4;      // head width when d fills the padded width (d = 128: 32; d = 64: 16)
#pragma unroll
        for (int c = 0; c < DKMAX; ++c) qh[c] = (c < dk && wave_live) ? Ar[hc + c] / temp : 0.f;
        // (q rows are written and read by the same wave only: no barrier before A is reused for v)
        wg_gemm_split<NT, NT, false, false, NW>(ring, Lp.wv, Lp.wfc, wave_live, sp, NoPre(), [&](int ot, f32x4 acc, f32x4, f32x4, f32x4, f32x4) { lds_st4(Ar + 16 * ot + 4 * g, acc); });
        lds_barrier();
        {
        float m = -INFINITY;
        float oh[DKMAX];
#pragma unroll
        for (int c = 0; c < DKMAX; ++c) oh[c] = 0.f;
        float z = 0.f;
        if (dk == DKF) {
          // Compile-time head width: a key's (and value's) DKF floats are NT ds_read_b128 issued back to back and ONE wait, the dot
          // product and the P.V update are straight FMA runs.  (The runtime-width loops below compile to one ds_read + s_waitcnt per
          // 4 channels behind a uniform branch each — ~1.5 k cycles per key and pass; this path made the all-eigenvector rho 2x faster.)
          const float* kb = Bm + u0 * LD + hc;
          const float* vb = A + u0 * LD + hc;
          // one pass over the keys (running maximum, rescaled accumulators): the K rows are read once — this loop is bound by LDS
          // delivery (every lane fetches its own copy of the key / value slice: 16 ds_read_b128 per key and wave)
#pragma unroll 1
          for (int j = 0; j < kv; ++j) {
            f32x4 k4[NT], v4[NT];
#pragma unroll
            for (int i = 0; i < NT; ++i) { k4[i] = lds_ld4(kb + j * LD + 4 * i); v4[i] = lds_ld4(vb + j * LD + 4 * i); }
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
              s0 += qh[4 * i] * k4[i][0] + qh[4 * i + 1] * k4[i][1];
              s1 += qh[4 * i + 2] * k4[i][2] + qh[4 * i + 3] * k4[i][3];
            }
            const float sj = s0 + s1;
            const float mn = fmaxf(m, sj);
            const float corr = expf(m - mn), pj = expf(sj - mn);      // (first key: m = -inf -> corr = 0)
            z = z * corr + pj;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
#pragma unroll
              for (int t = 0; t < 4; ++t) oh[4 * i + t] = oh[4 * i + t] * corr + pj * v4[i][t];
            }
            m = mn;
          }
        } else if ((dk & 3) == 0) {
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
        lds_barrier();   // all reads of k (Bm) are done: Bm receives the attention output
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
      }
      SN_STAMP(5);
      // fc(o) + x -> LayerNorm                                      (transformer_module.py:99-101)
      if (wave_live) split_rows<NT>(o, sp);
      if (ONE) load_x();   // the residual operand, straight from the input buffer (see ONE above)
      wg_gemm_split<NT, NT, false, false, NW>(ring, Lp.wfc, Lp.w1, wave_live, sp, NoPre(), [&](int ot, f32x4 acc, f32x4, f32x4, f32x4, f32x4) { x[ot] = acc + x[ot]; });   // residual in place: one row array for the whole layer
      SN_STAMP(6);
      if (wave_live) {
        int gl = g;
        asm volatile("" : "+v"(gl));      // (the lane's LDS address of the gamma / beta rows is not worth a register across the layer)
        masked_layernorm<NT, HP>(x, lnv + (l * 4 + 0) * D, lnv + (l * 4 + 1) * D, P.ln_eps, d, gl, valid);
        split_rows<NT>(x, sp);
      }
      SN_STAMP(7);
      // FFN: w2(relu(w1 y + b1)) + b2 + y -> LayerNorm               (transformer_module.py:113-127)
      wg_gemm_split<NT, NT, false, true, NW>(ring, Lp.w1, Lp.w2, wave_live, sp, NoPre(), [&](int ot, f32x4 acc, f32x4 b1, f32x4, f32x4, f32x4) { o[ot] = relu4(acc + b1); });
      SN_STAMP(8);
      if (wave_live) split_rows<NT>(o, sp);
#ifdef SN_PROFILE
      asm volatile("" :: "v"(sp[0].h), "v"(sp[NKB - 1].l));
#endif
      SN_STAMP(14);
      wg_gemm_split<NT, NT, false, true, NW>(ring, Lp.w2, wafter, wave_live, sp, NoPre(), [&](int ot, f32x4 acc, f32x4 b2, f32x4, f32x4, f32x4) { x[ot] = acc + b2 + x[ot]; });
      SN_STAMP(9);
      if (wave_live) {
        int gl = g;
        asm volatile("" : "+v"(gl));
        masked_layernorm<NT, HP>(x, lnv + (l * 4 + 2) * D, lnv + (l * 4 + 3) * D, P.ln_eps, d, gl, valid);
      }
      SN_STAMP(10);
    }
    // ---------------------------------------------------------------- sum over the node's slots -> out_sum[node, :]
    if (mfma_attn) {
      if (wave_live) {
        float* orow = S.out_sum + (int64_t)node * d;    // `node` is the same for the 16 rows of the tile
        int g_ = g;
        asm volatile("" : "+v"(g_));                    // (see load_x)
#pragma unroll
        for (int kk = 0; kk < NT; ++kk) {
          f32x4 s;
#pragma unroll
          for (int t = 0; t < 4; ++t) s[t] = tile_rowsum(valid ? x[kk][t] : 0.f);
          const int c = 16 * kk + 4 * g_;
          if (li == 0 && unit_ok && (HP ? c < d : (kk + 1 < NT || c < d))) *reinterpret_cast<float4*>(orow + c) = make_float4(s[0], s[1], s[2], s[3]);
        }
      }
      if (EARLY && bin + (int)gridDim.x < nbins) {                       // the next bin's rows and node range
        const int nnode = __builtin_amdgcn_readfirstlane((bin + (int)gridDim.x) * NW + wave);
        fetch_rows(nnode);
        fetch_graph(nnode);
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
          if (kk + 1 < NT || c < d) *reinterpret_cast<float4*>(orow + c) = make_float4(s[0], s[1], s[2], s[3]);
        }
      }
    }
  }
  { SN_PROF_ON(true); SN_STAMP(11); }
  ring.drain();
}

// =====================================================================================================
// k_rho_wide (round 5): nodes of MORE than 16 slots, d in {64, 128} (whole 16-channel tiles per head).  What differs from
// k_rho_fused<REGATTN = false>:
//   * ONE LDS image instead of two.  q stays in registers (its accumulator layout IS the MFMA operand layout of S^T = K Q^T), k goes
//     through the image, the scores and the softmax live in registers, v then goes through the SAME image and O^T = V^T P^T comes out
//     in the operand layout of the output projection (no attention-output image).  45 KB ring + 33 KB image + 2 KB per layer of
//     LayerNorm vectors = exactly 80 KB for a one-layer net: TWO workgroups per CU (the two-image form ran one, i.e. one wave per SIMD).
//   * Nodes need not start on a 16-row tile.  A wave computes its 16 queries against every key tile of the bin that one of its
//     queries' nodes touches (wave-uniform range) and masks each score by "key row inside my query's node" — so a bin is any set of whole
//     nodes with <= 64 rows in total.
//   * COLS (all eigenvectors, kmax == 0): a node of graph g has n_g slot rows, i.e. rho's units have exactly the shapes of phi's
//     (graph, slot) slabs, and the planner's bin member records (sn_plan_bins.phi_bin_mem: slabs of any graphs packed to <= 64 rows) are
//     reused: record (graph, index) = node `index` of the graph.  On the bench batch (n uniform in 9..37): 1 208 bins at 99 % fill
//     (1 302 at 92 % on phi's columns until the planner packed slabs; 1 750 at 68 % with tile-aligned nodes) — a 33-37-slot node no
//     longer owns a 64-row bin alone.  !COLS (16 < kmax < n): the closed-form per-graph bins (rho_bin0), nodes padded
//     to whole tiles, same kernel body.
//   * the slot sum is a [members x rows] . [rows x channels] product on the fp32 MFMA from the image (exact products by 0 / 1).
// Reference semantics: model_utils/transformer_module.py:27-127 (post-LN encoder layer, softmax over the node's valid slots).
// =====================================================================================================
template <int NT, bool ONE, bool COLS>
__global__ __launch_bounds__(RHO_R * 4, ONE ? 2 : 1) void k_rho_wide(RhoStruct S, sn_rho_params P) {
  static_assert(NT == 4 || NT == 8, "head width 16 or 32");
  constexpr int D = 16 * NT;
  constexpr int LD = D + 4;
  constexpr int NKB = (NT + 1) / 2;
  constexpr int CPH = NT / 4;                                        // 16-channel tiles per head
  using Ring = WRing<NT>;
  extern __shared__ __align__(1024) unsigned char lds_raw[];
  float* IMG = reinterpret_cast<float*>(lds_raw + Ring::BYTES);      // [RHO_R][LD]  k, then v, then the rows to be summed
  float* lnv = IMG + RHO_R * LD;                                     // [n_layers][4][D]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = wave * 16 + (lane & 15), g = lane >> 4, li = lane & 15;
  // (the three plan words at once — as `a != 0 || (COLS && b != 0)` plus the bin count they were three scalar round trips in a row)
  const int nb_raw = COLS ? S.meta[7] : S.meta[4];
  const int m_err = S.meta[5] | (COLS ? S.meta[1] : 0);
  const int nbins = m_err != 0 ? 0 : nb_raw;
  const int d = P.d;
  const float rtemp = 1.0f / sqrtf((float)(d / P.heads));
  Ring ring;
  ring.init(lds_raw, wave, lane);
  const void* wfirst = P.n_layers > 0 ? P.layers[0].wq : nullptr;
  if constexpr (ONE) {
    // one layer: its four LayerNorm vectors (one float4 per thread, scalar pointers) and the first RING weight chunks in ONE round
    // trip, straight-line (see k_rho_fused: behind a branch, or in front of an LDS store the compiler can see, every wait is for
    // all the LDS-DMA in flight)
    static_assert(4 * (D / 4) <= RHO_R * 4, "one float4 per thread covers the four vectors");
    const int li4 = (int)threadIdx.x < 4 * (D / 4) ? (int)threadIdx.x : 0;
    const sn_rho_layer& L0 = P.layers[0];
    const float *p0 = L0.ln1_g, *p1 = L0.ln1_b, *p2 = L0.ln2_g, *p3 = L0.ln2_b;
    asm volatile("" : "+s"(p0), "+s"(p1), "+s"(p2), "+s"(p3));
    const int lv = li4 / (D / 4), lc = 4 * (li4 - lv * (D / 4));
    typedef __attribute__((address_space(1))) const float gfloat_t;
    gfloat_t* src = (gfloat_t*)(lv == 0 ? p0 : (lv == 1 ? p1 : (lv == 2 ? p2 : p3)));
    const f32x4 lnr = f32x4{src[lc], src[lc + 1], src[lc + 2], src[lc + 3]};
    if (nbins <= (int)blockIdx.x) return;      // (a workgroup without a bin, or a plan that failed: the LDS is untouched)
    ring.pos = 0;
#pragma unroll
    for (int c = 0; c < Ring::DEPTH; ++c) ring.issue(wfirst, c, c);
    {
      typedef __attribute__((address_space(3))) float lds_float_t;
      const unsigned la = (unsigned)(size_t)(lds_float_t*)(lnv + 4 * li4);
      asm volatile("ds_write_b128 %0, %1" :: "v"(la), "v"(lnr) : "memory");
    }
    ring.template wait_shares<Ring::DEPTH - 1>();
    lds_barrier();
  } else {
    for (int i = threadIdx.x; i < P.n_layers * 4 * D; i += RHO_R * 4) {
      const int l = i / (4 * D), v = (i / D) & 3, c = i % D;
      const sn_rho_layer& Lq = P.layers[l];
      const float* src = v == 0 ? Lq.ln1_g : (v == 1 ? Lq.ln1_b : (v == 2 ? Lq.ln2_g : Lq.ln2_b));
      lnv[i] = src[c];
    }
    __syncthreads();
    if (wfirst != nullptr && nbins > (int)blockIdx.x) ring.prologue(wfirst, NT);
  }
  float* IMGr = IMG + r * LD;

  for (int bin = blockIdx.x; bin < nbins; bin += gridDim.x) {
    // ---------------------------------------------------------------- bin -> member nodes (wave-uniform: scalar registers)
    int m_off[8], m_len[8], m_node[8], m_gs[8];
    if constexpr (COLS) {
      // (the bin's eight member records as ONE scalar load: left to the compiler — vector loads behind the ring's LDS-DMA, made
      //  uniform word by word — they were four loads each behind the previous one's round trip, in every bin)
      typedef int i32x16 __attribute__((ext_vector_type(16)));
      i32x16 rw;
      {
        const int32_t* rec = S.bin_mem + (int64_t)bin * 16;
        asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rw) : "s"(rec) : "memory");
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int w0 = rw[2 * k], g0 = rw[2 * k + 1];
        const int n = ((w0 >> 25) & 63) + 1;                         // kmax == 0: K_g = n
        m_off[k] = (w0 >> 19) & 63;
        m_len[k] = (w0 >= 0 && n <= S.K) ? n : 0;                    // (n <= K: a caller's K smaller than the graph must not read outside x)
        m_node[k] = g0 + ((w0 >> 13) & 63);
        m_gs[k] = g0;
      }
    } else {
      int lo = 0, hi = S.B;                                          // the graph of the bin: largest g with rho_bin0[g] <= bin
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (S.rho_bin0[mid] <= bin) lo = mid; else hi = mid;
      }
      const int gs = S.graph_ptr[lo], n = S.graph_ptr[lo + 1] - gs;
      const int kg = (S.kmax > 0 && n > S.kmax) ? S.kmax : n;
      const int pad = ((kg + 15) >> 4) << 4;
      const int upb = RHO_R / pad;
      const int ub = (bin - S.rho_bin0[lo]) * upb;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        m_off[k] = k * pad;
        m_len[k] = (k < upb && ub + k < n) ? kg : 0;
        m_node[k] = gs + ub + k;
        m_gs[k] = gs;
      }
    }
    // my row; the key rows my wave's queries attend to
    int u0 = 0, kv = 0, node = 0, gsr = 0;
    int klo = RHO_R, khi = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (r >= m_off[k] && r < m_off[k] + m_len[k]) { u0 = m_off[k]; kv = m_len[k]; node = m_node[k]; gsr = m_gs[k]; }
      if (m_len[k] > 0 && m_off[k] < wave * 16 + 16 && m_off[k] + m_len[k] > wave * 16) {
        klo = min(klo, m_off[k]);
        khi = max(khi, m_off[k] + m_len[k]);
      }
    }
    const int slot = r - u0;
    const bool valid = kv > 0;
    const bool wave_live = khi > 0;
    const int kt_lo = klo >> 4, kt_hi = (khi + 15) >> 4;             // key tiles [kt_lo, kt_hi)
    // ---------------------------------------------------------------- load x (+ eigenvalue encoding)
    f32x4 x[NT];
    auto load_x = [&]() {
      int slot_ = slot, g_ = g;
      asm volatile("" : "+v"(slot_), "+v"(g_));
      const float* xr = S.x + (valid ? ((int64_t)node * S.K + slot_) : (int64_t)0) * D;
#pragma unroll
      for (int kk = 0; kk < NT; ++kk) {
        const f32x4 v = ld4(xr + 16 * kk + 4 * g_);
        x[kk] = valid ? v : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    };
    load_x();
    if (valid && P.has_pos) {
      const float ev = S.eigvals[gsr + slot];
      const float t0 = fmaxf((ev * P.pe_w1[0]) * P.pe_bn0_scale[0] + P.pe_bn0_shift[0], 0.f);
#pragma unroll
      for (int kk = 0; kk < NT; ++kk) {
        const int c = 16 * kk + 4 * g;
        const f32x4 w2 = ld4(P.pe_w2 + c), s1 = ld4(P.pe_bn1_scale + c), h1 = ld4(P.pe_bn1_shift + c);
        x[kk] += relu4((t0 * w2) * s1 + h1);
      }
    }
    // ---------------------------------------------------------------- encoder layers
#pragma unroll 1
    for (int l = 0; l < (ONE ? 1 : P.n_layers); ++l) {
      const sn_rho_layer& Lp = P.layers[l];
      const void* wafter = (l + 1 < P.n_layers) ? P.layers[l + 1].wq : wfirst;
      f32x4 o[NT];
      Split8 sp[NKB];
      if (wave_live) split_rows<NT>(x, sp);
      {
        f32x4 qf[NT];
        wg_gemm_split<NT, NT, false, false>(ring, Lp.wq, Lp.wk, wave_live, sp, NoPre(), [&](int ot, f32x4 acc, f32x4, f32x4, f32x4, f32x4) { qf[ot] = acc * rtemp; });
        wg_gemm_split<NT, NT, false, false>(ring, Lp.wk, Lp.wv, wave_live, sp, NoPre(), [&](int ot, f32x4 acc, f32x4, f32x4, f32x4, f32x4) { lds_st4(IMGr + 16 * ot + 4 * g, acc); });
        lds_barrier();
        // S^T tile = K_tile Q^T on the fp32 MFMA: A[i = key li][k] = k[key][c], B[k][j = query li] = q[query][c], c = 16 ot + 4 g + t
        // (both operands of a lane are values of its own lane row) -> lane (li, g) holds S[query li][key 16 kt + 4 g + t]
        f32x4 sc[4][4];
#pragma unroll
        for (int h = 0; h < 4; ++h)
#pragma unroll
          for (int kt = 0; kt < 4; ++kt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (wave_live && kt >= kt_lo && kt < kt_hi) {
              const float* kr = IMG + (16 * kt + li) * LD + 16 * CPH * h + 4 * g;
#pragma unroll
              for (int j = 0; j < CPH; ++j) {
                const f32x4 kq = lds_ld4(kr + 16 * j);
                acc = mfma16(kq[0], qf[CPH * h + j][0], acc);
                acc = mfma16(kq[1], qf[CPH * h + j][1], acc);
                acc = mfma16(kq[2], qf[CPH * h + j][2], acc);
                acc = mfma16(kq[3], qf[CPH * h + j][3], acc);
              }
            }
            sc[h][kt] = acc;
          }
        // softmax over the keys of my query's node: key row 16 kt + 4 g + t is a key iff it lies in [u0, u0 + kv)
        if (wave_live) {
#pragma unroll
          for (int h = 0; h < 4; ++h) {
            float m = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
              for (int t = 0; t < 4; ++t)
                if ((unsigned)(16 * kt + 4 * g + t - u0) < (unsigned)kv) m = fmaxf(m, sc[h][kt][t]);
            m = group_allmax(m);
            float z = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
              for (int t = 0; t < 4; ++t) {
                const float e = ((unsigned)(16 * kt + 4 * g + t - u0) < (unsigned)kv) ? expf(sc[h][kt][t] - m) : 0.f;
                sc[h][kt][t] = e;
                z += e;
              }
            z = row_allsum(z);
            const float zi = (kv > 0 && z > 0.f) ? 1.0f / z : 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
              for (int t = 0; t < 4; ++t) sc[h][kt][t] *= zi;
          }
        }
        lds_barrier();   // every wave has read its keys: the image receives v
        wg_gemm_split<NT, NT, false, false>(ring, Lp.wv, Lp.wfc, wave_live, sp, NoPre(), [&](int ot, f32x4 acc, f32x4, f32x4, f32x4, f32x4) { lds_st4(IMGr + 16 * ot + 4 * g, acc); });
        lds_barrier();
        // O^T tile = V^T P^T: A[i = channel li][k = key 4 g + s] = v[key][16 ot + li], B[k][j = query li] = P[query][key] (my own
        // register s) -> lane (li, g) holds O[query li][16 ot + 4 g + r]: the operand layout of the output projection
#pragma unroll
        for (int ot = 0; ot < NT; ++ot) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kt = 0; kt < 4; ++kt)
            if (wave_live && kt >= kt_lo && kt < kt_hi) {
              const float* vr = IMG + (16 * kt + 4 * g) * LD + 16 * ot + li;
              acc = mfma16(vr[0], sc[ot / CPH][kt][0], acc);
              acc = mfma16(vr[LD], sc[ot / CPH][kt][1], acc);
              acc = mfma16(vr[2 * LD], sc[ot / CPH][kt][2], acc);
              acc = mfma16(vr[3 * LD], sc[ot / CPH][kt][3], acc);
            }
          o[ot] = acc;
        }
      }
      // fc(o) + x -> LayerNorm                                      (transformer_module.py:99-101)
      if (wave_live) split_rows<NT>(o, sp);
      if (ONE) load_x();   // the residual operand, straight from the input buffer
      wg_gemm_split<NT, NT, false, false>(ring, Lp.wfc, Lp.w1, wave_live, sp, NoPre(), [&](int ot, f32x4 acc, f32x4, f32x4, f32x4, f32x4) { x[ot] = acc + x[ot]; });
      if (wave_live) {
        int gl = g;
        asm volatile("" : "+v"(gl));
        masked_layernorm<NT>(x, lnv + (l * 4 + 0) * D, lnv + (l * 4 + 1) * D, P.ln_eps, d, gl, valid);
        split_rows<NT>(x, sp);
      }
      // FFN: w2(relu(w1 y + b1)) + b2 + y -> LayerNorm               (transformer_module.py:113-127)
      wg_gemm_split<NT, NT, false>(ring, Lp.w1, Lp.w2, wave_live, sp, NoPre(), [&](int ot, f32x4 acc, f32x4 b1, f32x4, f32x4, f32x4) { o[ot] = relu4(acc + b1); });
      if (wave_live) split_rows<NT>(o, sp);
      wg_gemm_split<NT, NT, false>(ring, Lp.w2, wafter, wave_live, sp, NoPre(), [&](int ot, f32x4 acc, f32x4 b2, f32x4, f32x4, f32x4) { x[ot] = acc + b2 + x[ot]; });
      if (wave_live) {
        int gl = g;
        asm volatile("" : "+v"(gl));
        masked_layernorm<NT>(x, lnv + (l * 4 + 2) * D, lnv + (l * 4 + 3) * D, P.ln_eps, d, gl, valid);
      }
    }
    // ---------------------------------------------------------------- sum over every node's slots -> out_sum[node, :]
    // out[member][c] = sum_rows M[member][row] X[row][c], M = 0 / 1 membership: A[i = member li][k = row 4 s + g], B[k][j = channel li].
    // Every wave publishes its rows (zeros for rows without a slot: 0 * stale would be NaN) and owns CPH channel tiles of the sums.
#pragma unroll
    for (int kk = 0; kk < NT; ++kk) lds_st4(IMGr + 16 * kk + 4 * g, valid ? x[kk] : f32x4{0.f, 0.f, 0.f, 0.f});
    lds_barrier();
    {
      int mo = 0, ml = 0;                                            // member li (< 8) of this lane: first row, row count
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (li == k) { mo = m_off[k]; ml = m_len[k]; }
      int nd[4];                                                     // node of member 4 g + t (stores: lane groups 0 and 1 only)
      bool st[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        nd[t] = g == 0 ? m_node[t] : m_node[4 + t];
        st[t] = g < 2 && (g == 0 ? m_len[t] : m_len[4 + t]) > 0;
      }
#pragma unroll
      for (int j = 0; j < CPH; ++j) {
        const int ct = wave * CPH + j;
        const float* xc = IMG + g * LD + 16 * ct + li;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < RHO_R / 4; ++s) {
          const float a = ((unsigned)(4 * s + g - mo) < (unsigned)ml) ? 1.f : 0.f;
          acc = mfma16(a, xc[4 * s * LD], acc);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
          if (st[t]) S.out_sum[(int64_t)nd[t] * D + 16 * ct + li] = acc[t];
      }
    }
  }
  ring.drain();
}

template <int NT, bool ONE, bool COLS>
static int launch_rho_wide(const RhoStruct& S, const sn_rho_params& P, int64_t bins_bound, hipStream_t st) {
  constexpr int LD = 16 * NT + 4;
  const size_t lds_fixed = (size_t)WRing<NT>::BYTES + (size_t)(RHO_R * LD) * sizeof(float);
  const size_t lds_max = lds_fixed + (size_t)SN_RHO_MAX_LAYERS * 4 * 16 * NT * sizeof(float);
  const size_t lds = lds_fixed + (size_t)(P.n_layers > 0 ? P.n_layers : 1) * 4 * 16 * NT * sizeof(float);
  static int cus = 0;
  if (cus == 0) {
    if (lds_max > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_rho_wide<NT, ONE, COLS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_max) != hipSuccess)
      return fail(SN_ERR_LAUNCH, "sn_rho_fused_f32: cannot raise the dynamic LDS limit to %zu", lds_max);
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    cus = n > 0 ? n : 256;
  }
  int64_t grid = bins_bound < (int64_t)2 * cus ? bins_bound : (int64_t)2 * cus;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL((k_rho_wide<NT, ONE, COLS>), dim3((unsigned)grid), dim3(RHO_R * 4), lds, st, S, P);
  return SN_OK;
}

template <int NT, bool REGATTN, bool ONE, bool HP = false, int NW = 4>
static int launch_rho(const RhoStruct& S, const sn_rho_params& P, int64_t bins_bound, hipStream_t st) {
  constexpr int LD = 16 * NT + 4;
  // (the LayerNorm vectors of the layers the net HAS, not of SN_RHO_MAX_LAYERS: 47 KB instead of 61 for the register-attention variant
  //  of a one-layer net — LDS the overlap mode's co-resident kernels can use)
  const size_t lds_fixed = (size_t)WRing<NT, NW>::BYTES + (REGATTN ? 0 : (size_t)(2 * RHO_R * LD) * sizeof(float));
  const size_t lds_max = lds_fixed + (size_t)SN_RHO_MAX_LAYERS * 4 * 16 * NT * sizeof(float);
  const size_t lds = lds_fixed + (size_t)(P.n_layers > 0 ? P.n_layers : 1) * 4 * 16 * NT * sizeof(float);
  static int cus = 0;
  if (cus == 0) {
    if (lds_max > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_rho_fused<NT, REGATTN, ONE, HP, NW>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_max) != hipSuccess)
      return fail(SN_ERR_LAUNCH, "sn_rho_fused_f32: cannot raise the dynamic LDS limit to %zu", lds_max);
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    cus = n > 0 ? n : 256;
  }
  // eight waves per CU either way: two 4-wave workgroups (two weight streams), or one 8-wave workgroup (one stream)
  // (4 waves, two workgroups per CU: with the 47 KB of a one-layer net three or four fit, measured 61.5 against 60.6 us — no gain)
  const int64_t per_cu = NW == 8 ? 1 : 2;
  int64_t grid = bins_bound < per_cu * cus ? bins_bound : per_cu * cus;
#ifdef SN_RHO_GRID1
  grid = cus;
#endif
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL((k_rho_fused<NT, REGATTN, ONE, HP, NW>), dim3((unsigned)grid), dim3(64 * NW), lds, st, S, P);
  return SN_OK;
}

// (A/B switch of the profile scripts: SN_RHO_WAVES=8 in the environment puts the register-attention variants of the common widths on
//  8-wave workgroups.  MEASURED SLOWER and therefore off by default — profiles/scripts/ab_env.sh, one box, alternating: headline rho
//  61.5 -> 72.9 us, Alchemy 203.7 -> 246.8, hidden 64 16.6 -> 19.8: the stream is not what bounds rho; eight waves meeting at every
//  chunk barrier cost more than the second stream they save.)
static bool rho_eight_waves() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("SN_RHO_WAVES"); v = (e && e[0] == '8') ? 1 : 0; }
  return v == 1;
}

template <bool REGATTN, bool ONE>
static int dispatch_rho(int nt, const RhoStruct& S, const sn_rho_params& P, int64_t bound, hipStream_t st) {
  if constexpr (REGATTN) {
    // the common widths on 8-wave workgroups (one weight stream per CU)
    if (rho_eight_waves()) {
      if (nt == 8) return launch_rho<8, true, ONE, false, 8>(S, P, bound, st);
      if (nt == 4) return launch_rho<4, true, ONE, false, 8>(S, P, bound, st);
    }
  }
  switch (nt) {
    case 1: return launch_rho<1, REGATTN, ONE>(S, P, bound, st);
    case 2: return launch_rho<2, REGATTN, ONE>(S, P, bound, st);
    case 3: return launch_rho<3, REGATTN, ONE>(S, P, bound, st);
    case 4: return launch_rho<4, REGATTN, ONE>(S, P, bound, st);
    case 5: return launch_rho<5, REGATTN, ONE>(S, P, bound, st);
    case 6: return launch_rho<6, REGATTN, ONE>(S, P, bound, st);
    case 7: return launch_rho<7, REGATTN, ONE>(S, P, bound, st);
    default: return launch_rho<8, REGATTN, ONE>(S, P, bound, st);
  }
}

}  // namespace sn

using namespace sn;

#ifdef SN_PROFILE
extern "C" int sn_prof_read_rho(long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_prof), sizeof(long long) * 64); }
#endif

extern "C" int sn_rho_fused_f32(const sn_rho_params* params, const float* x, const float* eigen_values,
                                const int32_t* graph_ptr, int64_t B, int64_t N, const sn_plan_bins* bins, int kmax,
                                int K, float* out_sum, void* stream) {
  SN_REQUIRE(params && x && graph_ptr && bins && bins->rho_bin0 && bins->meta && out_sum, "sn_rho_fused_f32: null pointer");
  const sn_rho_params& P = *params;
  SN_REQUIRE(P.d > 0 && P.d <= 128 && (P.d & 3) == 0, "sn_rho_fused_f32: hidden width %d must be a multiple of 4 in (0, 128]", P.d);
  SN_REQUIRE(P.heads == 4 && P.d % P.heads == 0, "sn_rho_fused_f32: needs 4 heads dividing d (got %d heads, d=%d)", P.heads, P.d);
  SN_REQUIRE(P.n_layers >= 0 && P.n_layers <= SN_RHO_MAX_LAYERS, "sn_rho_fused_f32: %d layers unsupported", P.n_layers);
  SN_REQUIRE(!P.has_pos || (eigen_values && P.pe_w1 && P.pe_bn0_scale && P.pe_bn0_shift && P.pe_w2 && P.pe_bn1_scale && P.pe_bn1_shift),
             "sn_rho_fused_f32: eigenvalue encoder parameters missing");
  for (int l = 0; l < P.n_layers; ++l) {
    const sn_rho_layer& L = P.layers[l];
    SN_REQUIRE(L.wq && L.wk && L.wv && L.wfc && L.ln1_g && L.ln1_b && L.w1 && L.w2 && L.ln2_g && L.ln2_b,
               "sn_rho_fused_f32: layer %d parameters missing", l);
  }
  SN_REQUIRE(K > 0 && B >= 0 && N >= 0 && B < (1ll << 31), "sn_rho_fused_f32: bad sizes");
  if (B == 0 || N == 0) return SN_OK;
  SN_REQUIRE(N < (1ll << 31), "sn_rho_fused_f32: too many nodes");
  RhoStruct S{x, eigen_values, graph_ptr, bins->rho_bin0, bins->meta, (int)B, kmax, K, out_sum, bins->node_graph, (int)N,
              bins->phi_bin_mem};
  hipStream_t st = (hipStream_t)stream;
  const int64_t bound = N + B;   // every bin holds at least one node
  // attention in registers when every node has <= 16 slots and the head width is a multiple of 16
  const bool one = P.n_layers == 1 && !P.has_pos;
  // most valid slots any node can have: kmax when set, else the dense slot count K (= the largest graph; all eigenvectors)
  const int kcap = (kmax > 0 && kmax < K) ? kmax : K;
  int rc;
  SN_REQUIRE(kcap > 16 || bins->node_graph, "sn_rho_fused_f32: sn_plan_bins.node_graph is needed when every node has <= 16 slots");
  if (P.head_pad > 0 && kcap <= 16) {
    // head-padded packing: every head occupies head_pad (16 or 32) channels of the q/k/v/attention tensors, so the register
    // attention applies to any d; all weights / vectors are zero-padded to heads*head_pad channels by the caller
    SN_REQUIRE((P.head_pad == 16 || P.head_pad == 32) && P.head_pad >= P.d / P.heads,
               "sn_rho_fused_f32: head_pad must be 16 or 32 and >= d/heads (got %d for d=%d)", P.head_pad, P.d);
    if (rho_eight_waves()) {
      if (P.head_pad == 16) rc = one ? launch_rho<4, true, true, true, 8>(S, P, bound, st) : launch_rho<4, true, false, true, 8>(S, P, bound, st);
      else rc = one ? launch_rho<8, true, true, true, 8>(S, P, bound, st) : launch_rho<8, true, false, true, 8>(S, P, bound, st);
    } else if (P.head_pad == 16) rc = one ? launch_rho<4, true, true, true>(S, P, bound, st) : launch_rho<4, true, false, true>(S, P, bound, st);
    else rc = one ? launch_rho<8, true, true, true>(S, P, bound, st) : launch_rho<8, true, false, true>(S, P, bound, st);
  } else {
    SN_REQUIRE(P.head_pad == 0, "sn_rho_fused_f32: head-padded parameters need <= 16 slots per node (got %d); pass the "
               "natural-layout parameters for this batch", kcap);
    const bool regattn = kcap <= 16 && ((P.d / P.heads) & 15) == 0;
    const int nt = (P.d + 15) / 16;
    if (!regattn && (P.d == 64 || P.d == 128)) {
      // more than 16 slots per node, whole 16-channel tiles per head: one-image kernel, unaligned nodes; with all eigenvectors
      // (kmax == 0) on phi's columns (k_rho_wide)
      const bool cols = kmax == 0 && bins->phi_bin_mem != nullptr;
      if (P.d == 128) {
        if (cols) rc = one ? launch_rho_wide<8, true, true>(S, P, bound, st) : launch_rho_wide<8, false, true>(S, P, bound, st);
        else rc = one ? launch_rho_wide<8, true, false>(S, P, bound, st) : launch_rho_wide<8, false, false>(S, P, bound, st);
      } else {
        if (cols) rc = one ? launch_rho_wide<4, true, true>(S, P, bound, st) : launch_rho_wide<4, false, true>(S, P, bound, st);
        else rc = one ? launch_rho_wide<4, true, false>(S, P, bound, st) : launch_rho_wide<4, false, false>(S, P, bound, st);
      }
    } else
    rc = regattn ? (one ? dispatch_rho<true, true>(nt, S, P, bound, st) : dispatch_rho<true, false>(nt, S, P, bound, st))
                 : (one ? dispatch_rho<false, true>(nt, S, P, bound, st) : dispatch_rho<false, false>(nt, S, P, bound, st));
  }
  if (rc != SN_OK) return rc;
  SN_CHECK_LAUNCH("sn_rho_fused_f32");
  return SN_OK;
}
