// fused_gnn.hip — the GINE network that consumes the positional encoding, whole stack in ONE launch.
// Replaces (eval mode) the tail of SetTransformer.forward — `self.out` Linear+BatchNorm on the slot sum
// (sign_net.py:71) — and GNN.forward (Alchemy/sign_net/model.py:36-64, GINESignNetPyG/core/model.py:44-79):
// input encoder, Linear(cat[x, pos]), nl_gnn x [edge encoder, GINEConv (pyg_gnn_wrapper.py:19-28),
// BatchNorm, ReLU, residual], add-pooling over each graph and the 2-layer output encoder.
//
// This stage has few rows (one per node, no eigenvector-slot factor) and a long dependent chain
// (3 + 2*nl_gnn + 2 Linear layers), so it is latency-bound, not throughput-bound.  Mapping: ONE workgroup of
// 8 waves per graph (n <= 64 nodes) keeps the graph's node rows in LDS for the whole stack:
//   * X1 [64][d_pad+4] fp32 — the layer input h: read by the GINE neighbour sum relu(h_j + e_ji), by the residual and
//     by the pooling;
//   * SA, SB — every Linear's INPUT rows, stored already split into three bf16 planes (the fp32 products are six
//     bf16 partial products, fused_common.hpp §"fp32 GEMMs on the bf16 matrix pipe"): the producer of a value splits
//     it once (5.5 VALU ops) and every consumer wave reads ready-made MFMA operands (ds_read_b128 per plane, K block);
//   * the (output tile, row tile) pairs of a Linear are dealt OUTPUT-TILE MAJOR to the 8 waves, so each weight tile
//     (split-packed, with its epilogue vectors, sn_pack_split_f32) is fetched by exactly one wave — the weight traffic
//     of a Linear is the matrix once per CU; one barrier per Linear;
//   * edges of a graph repeat a handful of feature tuples (ZINC: 3 bond types): their embeddings for every layer are
//     built once into an LDS table indexed by edge class.
//
// Round 4: the same mapping serves two base nets of the DGL tree (GraphPrediction/nets/ZINC_graph_regression), each as ONE launch —
// gnn_graph<NT, MODE>: MODE 1 = gin_net.py (sn_gin_net_fused_f32: no edge term, plain messages, nothing behind the MLP's second Linear,
// a three-Linear readout), MODE 2 = transformer_net.py (sn_transformer_net_fused_f32: hidden 64, every stage a [64, 64] Linear, the edge
// attention one lane per (node, head), the two split images swapping roles per layer).  The mode is a template parameter: the GINE
// instantiations carry none of it.
#include "fused_common.hpp"

namespace sn {

#ifdef SN_PROFILE
static __device__ int g_prof_block = 0;
#undef SN_STAMP
#undef SN_ACCUM
#define SN_STAMP(i) do { if ((int)blockIdx.x == g_prof_block && threadIdx.x == 0) g_prof[i] = clock64(); } while (0)
#define SN_ACCUM(i, t0) do { if ((int)blockIdx.x == g_prof_block && threadIdx.x == 0) g_prof[i] += clock64() - (t0); } while (0)
#endif

constexpr int GNN_ROWS = SN_GNN_MAX_NODES;   // 64
constexpr int GNN_WAVES = 8;
constexpr int GNN_EMAX = 192;                // in-edges of one graph staged in LDS
constexpr int GNN_CLS = 16;                  // edge-feature classes per graph whose embeddings stay in LDS for all layers
constexpr int GNN_EEMAX = 96;                // rows of the edge-embedding area (class x layer table, or per-edge staging)
constexpr int GNN_EEPF = (GNN_EEMAX * 32 + GNN_WAVES * 64 - 1) / (GNN_WAVES * 64);   // float4 per thread (d_pad = 128)

// split image: three bf16 planes [64 rows][256 bytes]; inside a row the k-slots of K block kb and lane group g are one 16-byte chunk
// (8 bf16 = channels 32kb + 16(s>>2) + 4g + (s&3)) -> one ds_read_b128 per plane.  Chunk c = 4 kb + g of row r lives at chunk
// c ^ (r & 15) of the row (round 5): a ds_read_b128 is served in groups of 16 lanes — rows {0-3, 12-15} of lane group g with rows
// 4-11 of lane group g + 1 — and with the rows merely staggered (272-byte stride until round 4) every group had two lanes on one
// bank quad: 2 LDS cycles per group instead of 1, 43 % of the kernel's LDS cycles (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE,
// profiles/r04_pmc_sq_detail.txt); the XOR puts the 16 lanes of every group on 16 different quads (stores: unchanged, 2-way).
constexpr int SP_STRIDE = 256;
constexpr int SP_PLANE = GNN_ROWS * SP_STRIDE;
constexpr int SP_IMAGE = 52224;                      // 3 planes (49152 bytes) + room for the Transformer mode's fp32 Q | K | V rows
static_assert(SP_IMAGE >= 3 * SP_PLANE && SP_IMAGE % 16 == 0, "three planes fit an image");
__device__ __forceinline__ int sp_chunk(int row, int c) { return ((c ^ row) & 15) << 4; }   // byte offset of logical chunk c in its row

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
  int ee_rows;            // rows of the edge-embedding area
  const int32_t* flags_src;   // the batch's flag block [status(8) | bins meta(8)]: read for upstream failures, optionally reported
  int n_flags;
  int32_t* flags_host;
  int rho_ld, rho_w;      // row stride / width of rho_sum (the GINE net: both d; the DGL GIN net: the positional encoding [N, k])
  const void* head_mid;   // DGL GIN net: the middle Linear of MLPReadout (split-packed, (e0, e1) = (1, bias))
  int pool_mean;          // DGL GIN net: readout 'mean' instead of 'sum'
};

// the four channels 16*ot + 4g + t of `row` -> the three planes of a split image (exact 3-way split, fused_common.hpp)
__device__ __forceinline__ void sp_store4(unsigned char* img, int row, int ot, int g, f32x4 v) {
  float h[4], m[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    h[i] = __uint_as_float(__float_as_uint(v[i]) & 0xffff0000u);
    const float r = v[i] - h[i];
    m[i] = __uint_as_float(__float_as_uint(r) & 0xffff0000u);
    l[i] = r - m[i];
  }
  unsigned char* p = img + row * SP_STRIDE + sp_chunk(row, (ot >> 1) * 4 + g) + (ot & 1) * 8;
  *reinterpret_cast<uint2*>(p) = make_uint2(pack_hi16(h[0], h[1]), pack_hi16(h[2], h[3]));
  *reinterpret_cast<uint2*>(p + SP_PLANE) = make_uint2(pack_hi16(m[0], m[1]), pack_hi16(m[2], m[3]));
  *reinterpret_cast<uint2*>(p + 2 * SP_PLANE) = make_uint2(pack_hi16(l[0], l[1]), pack_hi16(l[2], l[3]));
}

// One output tile of a split-packed Linear in registers: NKB x 3 weight fragments + the 3 epilogue vectors.
template <int NKB>
struct WSplit { u32x4 f[NKB * 3]; f32x4 e[SPLIT_EPI]; };

template <int NKB>
__device__ __forceinline__ void wload(WSplit<NKB>& p, const void* wsp, int ot, int lane) {
  constexpr int NFE = 3 * NKB + SPLIT_EPI;
  const __amdgpu_buffer_rsrc_t rs = weight_rsrc(reinterpret_cast<const float*>(wsp), 0x7fffffff);
  const int voff = lane * 16;
  const int base = __builtin_amdgcn_readfirstlane(ot * NFE * 1024);
#pragma unroll
  for (int i = 0; i < 3 * NKB; ++i) p.f[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, base + i * 1024, 0);
#pragma unroll
  for (int j = 0; j < SPLIT_EPI; ++j) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, base + (3 * NKB + j) * 1024, 0);
    p.e[j] = f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
  }
}

// The (output tile, row tile) pairs a wave owns in one Linear: a contiguous range [t_lo, t_hi) of the flattened index
// t = ot*T + rt (T = row tiles of the graph, <= 4) — OUTPUT-TILE MAJOR, so that a wave works on one output tile (two
// at most) for all the graph's row tiles and every weight fragment is fetched by exactly one wave of the workgroup.
struct TileRange {
  int t_lo, t_hi, T;
  __device__ __forceinline__ bool empty() const { return t_lo >= t_hi; }
  __device__ __forceinline__ void decode(int t, int& ot, int& rt) const { ot = t / T; rt = t - ot * T; }
  __device__ __forceinline__ int first_ot() const { return t_lo / T; }
};

template <int NKB>
__device__ __forceinline__ f32x4 mfma_split_tile(const WSplit<NKB>& w, const Split8 (&x)[NKB]) {
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    const u32x4 wh = w.f[3 * kb], wm = w.f[3 * kb + 1], wl = w.f[3 * kb + 2];
    a1 = mfma_bf(wl, x[kb].h, a1);
    a0 = mfma_bf(wm, x[kb].h, a0);
    a1 = mfma_bf(wh, x[kb].l, a1);
    a0 = mfma_bf(wh, x[kb].m, a0);
    a1 = mfma_bf(wm, x[kb].m, a1);
    a0 = mfma_bf(wh, x[kb].h, a0);
  }
  return a0 + a1;
}

// One Linear over the workgroup's rows.  The wave's TileRange is <= 4 (ot, rt) pairs touching <= 2 output tiles.
// Weight tiles ping-pong between `pre` and `alt`: `pre` holds the first output tile on entry (fetched during the
// previous stage); a second output tile — or, at the end, the NEXT Linear's first tile — is fetched while the current
// one computes.  img: split input image; epi(rt, ot, acc, e0, e1, e2) with e* the tile's epilogue vectors.
template <int NKB, typename Epi>
__device__ __forceinline__ void coop_gemm(WSplit<NKB>& pre, WSplit<NKB>& alt, const void* wsp, const unsigned char* img,
                                          TileRange tr, int lane, Epi epi, const void* next_wsp, TileRange next_tr) {
  if (tr.empty()) {
    if (next_wsp && !next_tr.empty()) wload<NKB>(pre, next_wsp, next_tr.first_ot(), lane);   // idle here: keep the chain going
    return;
  }
  // (the range is laundered per call: its decoded (output tile, row tile) pairs and every LDS address derived from them are wave
  //  constants the compiler would otherwise compute once and keep in registers across all the Linears of the kernel — at NT = 7 they
  //  did not fit and 28 of them lived in a private segment)
  asm volatile("" : "+v"(tr.t_lo), "+v"(tr.t_hi), "+v"(tr.T));
  int ot0, rt0;
  tr.decode(tr.t_lo, ot0, rt0);
  int otl, rtl;
  tr.decode(tr.t_hi - 1, otl, rtl);
  const bool two_ots = otl != ot0;
  const unsigned char* rowbase = img + (lane & 15) * SP_STRIDE;
  auto load_rows = [&](int rt, Split8 (&x)[NKB]) {
    const unsigned char* p = rowbase + rt * 16 * SP_STRIDE;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      const unsigned char* pc = p + sp_chunk(lane & 15, kb * 4 + (lane >> 4));
      x[kb].h = *reinterpret_cast<const u32x4*>(pc);
      x[kb].m = *reinterpret_cast<const u32x4*>(pc + SP_PLANE);
      x[kb].l = *reinterpret_cast<const u32x4*>(pc + 2 * SP_PLANE);
    }
  };
#ifdef SN_PROFILE
  long long ct = clock64();
#endif
  Split8 in[NKB];
  load_rows(rt0, in);
#ifdef SN_PROFILE
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  SN_ACCUM(42, ct);
  ct = clock64();
#endif
  // prefetch: second output tile of this Linear, else the next Linear's first tile
  bool next_in_alt = false;
  if (two_ots) wload<NKB>(alt, wsp, otl, lane);
  else if (next_wsp && !next_tr.empty()) { wload<NKB>(alt, next_wsp, next_tr.first_ot(), lane); next_in_alt = true; }
  bool cur_is_pre = true;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t = tr.t_lo + i;
    if (t < tr.t_hi) {
      int ot, rt;
      tr.decode(t, ot, rt);
      if (ot != ot0 && cur_is_pre) {
        cur_is_pre = false;                                   // switch to the second tile (in alt); pre is free again:
        if (next_wsp && !next_tr.empty()) wload<NKB>(pre, next_wsp, next_tr.first_ot(), lane);   // next Linear's first tile
      }
#ifdef SN_PROFILE
      ct = clock64();
#endif
      if (i > 0) load_rows(rt, in);
#ifdef SN_PROFILE
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      SN_ACCUM(42, ct);
      ct = clock64();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // weights of this tile
      SN_ACCUM(41, ct);
      ct = clock64();
      f32x4 accp = cur_is_pre ? mfma_split_tile<NKB>(pre, in) : mfma_split_tile<NKB>(alt, in);
      asm volatile("" :: "v"(accp));
      SN_ACCUM(43, ct);
      ct = clock64();
      if (cur_is_pre) epi(rt, ot, accp, pre.e[0], pre.e[1], pre.e[2]);
      else epi(rt, ot, accp, alt.e[0], alt.e[1], alt.e[2]);
      SN_ACCUM(44, ct);
#else
      if (cur_is_pre) epi(rt, ot, mfma_split_tile<NKB>(pre, in), pre.e[0], pre.e[1], pre.e[2]);
      else epi(rt, ot, mfma_split_tile<NKB>(alt, in), alt.e[0], alt.e[1], alt.e[2]);
#endif
    }
  }
  if (next_in_alt) {   // single-tile range: the next Linear's first tile sits in alt — hand it over in pre
#pragma unroll
    for (int i = 0; i < 3 * NKB; ++i) pre.f[i] = alt.f[i];
#pragma unroll
    for (int j = 0; j < SPLIT_EPI; ++j) pre.e[j] = alt.e[j];
  }
}

// NT >= 7 (d = 112, 128): every wave owns exactly ONE output tile of every Linear, so the two register tiles simply alternate
// from stage to stage and each is refilled — with the tile of the stage AFTER the next — as soon as its last product and epilogue are
// done.  A tile is then on its way for a whole stage (the barrier, the other register tile's Linear, possibly an aggregation) before
// it is needed; with the ping-pong of coop_gemm it is needed one short stage (~1 us of work per wave) after its issue, and every one of
// the 22 stages waited out most of a memory round trip (fetch-size sweep: DESIGN.md §8).
// kb0: first K block of the operand rows (the Transformer mode's FFN 2 reads the two halves of its 128 hidden channels as K blocks 0-1
// and 2-3 of one image).
template <int NKB, typename Epi>
__device__ __forceinline__ void coop_gemm_roll(WSplit<NKB>& cur, const unsigned char* img, TileRange tr, int lane, Epi epi,
                                               const void* nn_wsp, TileRange nn_tr, int kb0 = 0) {
  if (!tr.empty()) {
    asm volatile("" : "+v"(tr.t_lo), "+v"(tr.t_hi), "+v"(tr.T));     // (see coop_gemm)
    const unsigned char* rowbase = img + (lane & 15) * SP_STRIDE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int t = tr.t_lo + i;
      if (t < tr.t_hi) {
        int ot, rt;
        tr.decode(t, ot, rt);
        Split8 in[NKB];
        const unsigned char* p = rowbase + rt * 16 * SP_STRIDE;
        int cx = (lane & 15) ^ (lane >> 4) ^ (kb0 << 2);             // my chunk of K block kb: (4 kb) ^ cx
        asm volatile("" : "+v"(cx));                                   // (per pair: else the chunk addresses of all four pairs are kept live)
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
          const unsigned char* pc = p + (((4 * kb) ^ cx) << 4);
          in[kb].h = *reinterpret_cast<const u32x4*>(pc);
          in[kb].m = *reinterpret_cast<const u32x4*>(pc + SP_PLANE);
          in[kb].l = *reinterpret_cast<const u32x4*>(pc + 2 * SP_PLANE);
        }
        epi(rt, ot, mfma_split_tile<NKB>(cur, in), cur.e[0], cur.e[1], cur.e[2]);
      }
    }
  }
  if (nn_wsp != nullptr && !nn_tr.empty()) wload<NKB>(cur, nn_wsp, nn_tr.first_ot(), lane);
}

// coop_gemm_roll for a graph whose row-tile count is a COMPILE-TIME constant (PAIRS; NT = 8: wave w owns output tile w and the row
// tiles 0 .. PAIRS-1): the refill of `cur` is issued INSIDE the last tile product — K block kb's three weight fragments are dead as
// soon as its six MFMAs are issued, so the next tile's fragments of that K block are requested right there, and the epilogue vectors
// behind the epilogue.  With coop_gemm_roll all 15 loads of every wave sit between the last epilogue and the stage's barrier: 8
// waves x 15 KB through the CU's 64 B/clk vector-memory path, ~1.9 k cycles per stage that nothing overlaps
// (profiles/r04_gnn_stage_stamps.txt: 3.6 k cycles of every stage do not depend on the graph's size).  The refill is unconditional
// (`nn_wsp` / `nn_ot` always name a valid tile: a wave without work in the stage after the next fetches one it will not use), so the
// register tile has ONE definition per call — as a branch the woven form made the compiler copy tiles at every join (900 bytes of
// private segment per lane).
template <int NKB, int PAIRS, typename Epi>
__device__ __forceinline__ void coop_gemm_weave(WSplit<NKB>& cur, const unsigned char* img, int ot, int lane, Epi epi,
                                                const void* nn_wsp, int nn_ot) {
  constexpr int NFE = 3 * NKB + SPLIT_EPI;
  const __amdgpu_buffer_rsrc_t rs = weight_rsrc(reinterpret_cast<const float*>(nn_wsp), 0x7fffffff);
  const int voff = lane * 16;
  const int base = __builtin_amdgcn_readfirstlane(nn_ot * NFE * 1024);
  const unsigned char* rowbase = img + (lane & 15) * SP_STRIDE;
#pragma unroll
  for (int rt = 0; rt < PAIRS; ++rt) {
    Split8 in[NKB];
    const unsigned char* p = rowbase + rt * 16 * SP_STRIDE;
    int cx = (lane & 15) ^ (lane >> 4);
    asm volatile("" : "+v"(cx));                                   // (per pair: see coop_gemm_roll)
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      const unsigned char* pc = p + (((4 * kb) ^ cx) << 4);
      in[kb].h = *reinterpret_cast<const u32x4*>(pc);
      in[kb].m = *reinterpret_cast<const u32x4*>(pc + SP_PLANE);
      in[kb].l = *reinterpret_cast<const u32x4*>(pc + 2 * SP_PLANE);
    }
    if (rt + 1 < PAIRS) {
      epi(rt, ot, mfma_split_tile<NKB>(cur, in), cur.e[0], cur.e[1], cur.e[2]);
      __builtin_amdgcn_sched_barrier(0);                           // pairs stay sequential (registers)
    } else {
      f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        const u32x4 wh = cur.f[3 * kb], wm = cur.f[3 * kb + 1], wl = cur.f[3 * kb + 2];
        a1 = mfma_bf(wl, in[kb].h, a1);
        a0 = mfma_bf(wm, in[kb].h, a0);
        a1 = mfma_bf(wh, in[kb].l, a1);
        a0 = mfma_bf(wh, in[kb].m, a0);
        a1 = mfma_bf(wm, in[kb].m, a1);
        a0 = mfma_bf(wh, in[kb].h, a0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 3; ++j) cur.f[3 * kb + j] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, base + (3 * kb + j) * 1024, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      epi(rt, ot, a0 + a1, cur.e[0], cur.e[1], cur.e[2]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < SPLIT_EPI; ++j) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, base + (3 * NKB + j) * 1024, 0);
        cur.e[j] = f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
      }
    }
  }
}

// DGL = true: the GraphPrediction tree's GIN net on the same mapping (nets/ZINC_graph_regression/gin_net.py:83-126 in eval mode):
//   h = embedding_h[atom] + embedding_p(p)          -> the input stage with lin_a = I, lin_b = embedding_p (bias in e0), "slot sum" = p
//   L x  h = MLP((1 + eps) h_i + sum_{j -> i} h_j)  -> no edge term and no ReLU in the message; Linear . ReLU . [BN folded] . Linear,
//                                                      nothing behind the second Linear (no BatchNorm, ReLU or residual)
//   readout sum / mean, MLPReadout (three Linears)  -> one more head stage (S.head_mid)
// MODE 2: the DGL tree's sparse graph Transformer on the same mapping (transformer_net.py:88-140 with layers/transformer.py:150-312,
// eval mode, d = 64, 8 heads; all stages are [64, 64] Linears — NT = 4):
//   per layer  Q, K, V (fp32 rows parked in the OTHER split image's space) -> one lane per (node, head): the edge attention of
//   sn_edge_attention_f32 over the node's in-edges, E_ij read from the [E, L*d] projection in global memory -> O_h + x, BatchNorm ->
//   FFN 1 (two 64-column halves of the 128 hidden channels) -> FFN 2 (two 64-deep halves, the first one's sums parked in fp32) + x1,
//   BatchNorm.  The two split images swap roles every layer (operand in one, Q | K | V / the next operand in the other).
// The eight stage matrices of layer l are layers[l].etab[0..7] = Q, K, V, O_h, FFN1[:64], FFN1[64:], FFN2[:, :64], FFN2[:, 64:]
// (w1s / w2s repeat etab[0] / etab[1]: what the input stage prefetches).
// TC > 0 (the GINE net at NT = 8 only): the graph's row-tile count as a compile-time constant — the node-row Linears run on
// coop_gemm_weave (the kernel dispatches on ceil(n / 16)).
template <int NT, int MODE = 0, int TC = 0>
__device__ __forceinline__ void gnn_graph(const GnnStruct& S, const sn_gnn_params& P) {
  constexpr bool DGL = MODE != 0, TF = MODE == 2;
  static_assert(TC == 0 || (NT == 8 && MODE == 0), "compile-time row tiles: the GINE net at NT = 8");
  static_assert(!TF || NT == 4, "the Transformer mode is written for d = 64");
  constexpr int D = 16 * NT;
  constexpr int LD = D + 4;
  constexpr int NKB = (NT + 1) / 2;
  extern __shared__ __align__(16) unsigned char lds_raw[];
  unsigned char* SA = lds_raw;                                   // split image: slot sum, then u, then the pooled row
  unsigned char* SB = lds_raw + SP_IMAGE;                        // split image: encoder output, pos, hidden rows
  float* X1 = reinterpret_cast<float*>(lds_raw + 2 * SP_IMAGE);  // [64 + 1][LD] fp32: h; row 64 stays zero (what a missing in-edge reads)
  int* erow = reinterpret_cast<int*>(X1 + (GNN_ROWS + 1) * LD);  // [65]  CSR row pointers local to the graph
  int* esrc = erow + GNN_ROWS + 4;                               // [GNN_EMAX] local source row of every in-edge
  int* efeat = esrc + GNN_EMAX;                                  // [GNN_EMAX][edge_nf] feature words (int idx / float)
  int* ecls = efeat + GNN_EMAX * (P.n_layers > 0 ? P.edge_nf : 0);   // [GNN_EMAX] feature class of every in-edge
  int* elead = ecls + GNN_EMAX;                                     // [GNN_EMAX] first edge with the same features (scratch)
  int* cedge = elead + GNN_EMAX;                                    // [GNN_CLS]  representative edge of every class
  float* EE = reinterpret_cast<float*>(cedge + GNN_CLS);           // [ee_rows][LD] edge embeddings (per class x layer, or per edge)
  float* PART = EE;                                                  // Transformer mode (no edge tables): [64][LD] fp32 sums of FFN 2's first half
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, li = lane & 15;
  const int gi = blockIdx.x;
  SN_STAMP(0);
#ifdef SN_PROFILE
  if ((int)blockIdx.x == g_prof_block && threadIdx.x == 0) for (int i = 8; i < 20; ++i) g_prof[i] = 0;
  long long pt = 0;
#endif
  WSplit<NKB> pre, alt;
  const int gs = S.graph_ptr[gi], n = S.graph_ptr[gi + 1] - gs;
  // a graph that cannot be evaluated gets a NaN output row (never uninitialised memory): see sn_gnn_fused_f32
  auto give_up = [&](int bit) {
    if (threadIdx.x == 0 && bit) atomicOr(&S.status[3], bit);
    if ((int)threadIdx.x < P.n_out) S.y[(int64_t)gi * P.n_out + threadIdx.x] = __uint_as_float(0x7fc00000u);
  };
  // earlier stages of this batch failed (malformed batch: status[0]; phi / rho bins not laid out: meta[1], meta[5])
  if (S.flags_src != nullptr && (S.flags_src[0] != 0 || (S.n_flags >= 16 && (S.flags_src[9] != 0 || S.flags_src[13] != 0)))) { give_up(0); return; }
  if (n <= 0) { give_up(8); return; }          // (a graph without nodes: flagged — the layer path evaluates it as the reference does)
  if (n > GNN_ROWS) { give_up(1); return; }
  const int e_base = S.rowptr[gs];
  const int ne = S.rowptr[gs + n] - e_base;               // (checked below, behind the loads that need `gs` only)
  const int d = P.d;
  // block-wide facts of the prologue: [0] a discrete feature id of this graph lies outside its embedding table, [1] an edge feature
  // value the small class table cannot index, [2] bit v: edge feature value v occurs (one discrete edge feature column)
  __shared__ unsigned s_pro[3];
  if (threadIdx.x == 0) { s_pro[0] = 0u; s_pro[1] = 0u; s_pro[2] = 0u; }
  const int T = TC > 0 ? TC : (n + 15) >> 4;                // row tiles (1..4)
  const int ntile = T * NT;
  TileRange tr;                                               // my share of every node-row Linear (output-tile major)
  tr.T = T;
  if constexpr (GNN_WAVES % NT == 0) {
    // NT divides the wave count: GNN_WAVES / NT waves share an output tile and split its row tiles — every wave stays on ONE output
    // tile for any T (what coop_gemm_roll needs), with at most ceil(T / (GNN_WAVES / NT)) pairs
    constexpr int WPO = GNN_WAVES / NT;
    const int ot = wave / WPO, h = wave % WPO, per = (T + WPO - 1) / WPO;
    tr.t_lo = ot * T + (h * per < T ? h * per : T);
    tr.t_hi = ot * T + ((h + 1) * per < T ? (h + 1) * per : T);
  } else {
    const int q = (ntile + GNN_WAVES - 1) / GNN_WAVES;        // pairs per wave (<= T since NT <= 8: at most 2 output tiles)
    tr.t_lo = wave * q < ntile ? wave * q : ntile;
    tr.t_hi = tr.t_lo + q < ntile ? tr.t_lo + q : ntile;
  }
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
  bool use_ee = !DGL && P.n_layers > 0 && ne <= S.ee_rows;
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
    int t0 = threadIdx.x;
    asm volatile("" : "+v"(t0));      // (per call: the (edge, channel) pair of every slot is a lane constant the compiler would otherwise
                                      //  keep in registers across the layers — with D/4 not a power of two it is not rematerialised)
#pragma unroll
    for (int i = 0; i < GNN_EEPF; ++i) {
      const int idx = t0 + i * GNN_WAVES * 64;
      eepf[i] = zero4;
      if (idx < ne * (D / 4)) eepf[i] = edge_embed(Lq, idx / (D / 4), 4 * (idx % (D / 4)));
    }
  };
  bool tab_pending = false;    // the class table's rows are in registers (eepf), stored by ee_store() behind the input Linears
  auto ee_store = [&]() {
    if (tab_pending) {
      tab_pending = false;
      int t1 = threadIdx.x;
      asm volatile("" : "+v"(t1));
#pragma unroll
      for (int i = 0; i < GNN_EEPF; ++i) {
        const int idx = t1 + i * GNN_WAVES * 64;
        if (idx < P.n_layers * ncls * (D / 4)) lds_st4(EE + (idx / (D / 4)) * LD + 4 * (idx % (D / 4)), eepf[i]);
      }
      return;
    }
    if (!use_ee) return;
    int t0 = threadIdx.x;
    asm volatile("" : "+v"(t0));
#pragma unroll
    for (int i = 0; i < GNN_EEPF; ++i) {
      const int idx = t0 + i * GNN_WAVES * 64;
      if (idx < ne * (D / 4)) lds_st4(EE + (idx / (D / 4)) * LD + 4 * (idx % (D / 4)), eepf[i]);
    }
  };

  // one output tile per wave and Linear -> coop_gemm_roll: NT in {1, 2, 4, 8} by the range split above; NT = 7: the wave's share
  // q = ceil(7 T / 8) of the (tile, row tile) pairs equals T for every T <= 4
  constexpr bool ROLL = NT >= 7 || GNN_WAVES % NT == 0;
  static_assert(GNN_WAVES == 8, "the ROLL condition assumes 8 waves");
  TileRange h2;                               // my share of the last Linear (one tile: wave 0)
  h2.T = 1;
  h2.t_lo = 0;
  h2.t_hi = wave == 0 ? 1 : 0;
  // The graph's own inputs are requested FIRST: the memory counter is in-order, so every later wait for one of them would also wait
  // for whatever was issued before it — the two 15 KB weight tiles below.  One CSR row pointer, one in-edge (source, edge id) and up
  // to four float4 of the slot sum per thread cover the whole graph (n <= 64, ne <= 192 < blockDim).
  static_assert(GNN_EMAX <= GNN_WAVES * 64 && GNN_ROWS < GNN_WAVES * 64 && GNN_ROWS * (D / 4) <= 4 * GNN_WAVES * 64, "one pass of the block covers the graph");
  const int tid = threadIdx.x;
  // Round 5: what needs only the graph's first node — CSR row pointers, the slot sum, the node feature ids of my rows — is requested
  // before the in-edge count is even known; the edge lists follow, then the edge features and the node-table rows (as soon as their
  // ids are there), and only then the two 15 KB weight tiles: every wait below names loads issued before them.  The prologue used to be
  // six dependent round trips (graph_ptr, rowptr, edge lists, edge features, class table | node ids, node table): the node side now
  // runs beside the edge side and the class table's rows land under the two input Linears.
  const int rp_v = tid <= n ? S.rowptr[gs + tid] : 0;
  const int rho_ld = S.rho_ld, rho_w = S.rho_w;
  const bool rs_vec = ((rho_ld | rho_w) & 3) == 0;
  f32x4 rs_v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    rs_v[j] = zero4;
    const int i = tid + j * GNN_WAVES * 64;
    if (rs_vec && i < n * (D / 4)) {
      const int rr = i / (D / 4), c4 = i % (D / 4);
      if (4 * c4 < rho_w) rs_v[j] = ld4(S.rho_sum + (int64_t)(gs + rr) * rho_ld + 4 * c4);
    }
  }
  const bool xid_pref = P.node_discrete && P.node_nf == 1 && (d & 3) == 0;     // one id column: the ids of my (<= 4) rows, up front
  long long xid[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    xid[i] = -1;
    const int t = tr.t_lo + i;
    if (xid_pref && t < tr.t_hi) {
      int ot, rt;
      tr.decode(t, ot, rt);
      if (rt * 16 + li < n) xid[i] = reinterpret_cast<const int64_t*>(S.x)[(int64_t)(gs + rt * 16 + li) * S.ldx];
    }
  }
  if (ne > GNN_EMAX) { give_up(2); return; }
  const int src_v = tid < ne ? S.col[e_base + tid] : 0;
  const int eid_v = tid < ne ? S.eperm[e_base + tid] : 0;
  SN_STAMP(30);
  // ---------------------------------------------------------------- clear the split images (K padding must read as 0)
  for (int i = threadIdx.x; i < 2 * SP_IMAGE / 16; i += GNN_WAVES * 64)
    reinterpret_cast<uint4*>(lds_raw)[i] = make_uint4(0u, 0u, 0u, 0u);
  if ((int)threadIdx.x < LD) { X1[GNN_ROWS * LD + threadIdx.x] = 0.f; EE[S.ee_rows * LD + threadIdx.x] = 0.f; }   // the two zero rows
  lds_barrier();               // (the prologue's flags are initialised)
  SN_STAMP(31);
  // ---------------------------------------------------------------- per-graph CSR + edge data -> LDS (once)
  const bool efast = !DGL && P.n_layers > 0 && P.edge_discrete && P.edge_nf == 1 && (d & 3) == 0;   // classes = the feature values
  int my_ev = 0;
  {
    const int EF = P.edge_nf;
    if (tid <= n) erow[tid] = rp_v - e_base;
    if (tid < ne) {
      const int k = tid;
      esrc[k] = src_v - gs;
      const int eid = eid_v;
      if (TF) ecls[k] = eid;                              // the attention reads E[eid] from global memory
      if (!DGL && P.n_layers > 0) {
        if (P.edge_discrete) {
          const int64_t* ei = reinterpret_cast<const int64_t*>(S.edge_attr) + (int64_t)eid * S.lde;
          for (int f = 0; f < EF; ++f) {
            const int64_t v = ei[f];
            const bool ok = (uint64_t)v < (uint64_t)P.edge_vocab;      // nn.Embedding would raise IndexError: never dereferenced
            efeat[k * EF + f] = ok ? (int)v : 0;
            if (!ok) { atomicOr(&S.status[3], 4); atomicOr(&s_pro[0], 1u); }
            if (efast) {
              my_ev = ok ? (int)v : 0;
              if (my_ev < 32) atomicOr(&s_pro[2], 1u << my_ev); else atomicOr(&s_pro[1], 1u);
            }
          }
        } else {
          const float* ea = reinterpret_cast<const float*>(S.edge_attr) + (int64_t)eid * S.lde;
          for (int f = 0; f < EF; ++f) efeat[k * EF + f] = __float_as_int(ea[f]);
        }
      }
    }
  }
  // the node-table rows of my pairs (their ids were the first loads): in flight during the class work below
  f32x4 nrow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    nrow[i] = zero4;
    const int t = tr.t_lo + i;
    if (xid_pref && t < tr.t_hi && xid[i] != -1) {
      int ot, rt;
      tr.decode(t, ot, rt);
      long long xv = xid[i];
      if ((uint64_t)xv >= (uint64_t)P.node_vocab) { xv = 0; atomicOr(&S.status[3], 4); atomicOr(&s_pro[0], 1u); }
      const int c = 16 * ot + 4 * g;
      if (c < d) nrow[i] = ld4(P.ntab[0] + xv * d + c);
    }
  }
  // (both weight tiles behind every load of the graph's own data; measured and dropped: the first tile right behind the edge lists
  //  — prologue 20.3 k -> 22.5 k cycles — and in front of everything — 23.6 k)
  if (!tr.empty()) {
    wload<NKB>(pre, P.lin_a, tr.first_ot(), lane);                      // in flight while the inputs are staged
    if constexpr (ROLL) wload<NKB>(alt, P.lin_b, tr.first_ot(), lane);
  }
  lds_barrier();               // images cleared, CSR / edge features staged (an LDS-only barrier: the loads above stay in flight)
  SN_STAMP(32);
  // ---------------------------------------------------------------- edge-feature classes (see use_tab above)
  if (efast && s_pro[1] == 0u) {
    // one discrete feature column with small values (ZINC: bond types 1..3): the class of an edge IS its value's rank among the
    // values present — no search over the edges, no further barrier; row (l, c) of the table = layer l's embedding of value c
    const unsigned present = s_pro[2];
    ncls = __popc(present);
    if (tid < ne) ecls[tid] = __popc(present & ((1u << my_ev) - 1u));
    use_tab = ncls <= GNN_CLS && P.n_layers * ncls <= S.ee_rows;
    if (use_tab) {
      use_ee = false;
      tab_pending = true;
      int t0 = threadIdx.x;
      asm volatile("" : "+v"(t0));
#pragma unroll
      for (int i = 0; i < GNN_EEPF; ++i) {
        const int idx = t0 + i * GNN_WAVES * 64;
        eepf[i] = zero4;
        if (idx < P.n_layers * ncls * (D / 4)) {
          const int rowi = idx / (D / 4), ch = 4 * (idx % (D / 4));
          const int l = rowi / ncls, c = rowi - l * ncls;
          unsigned m = present;
          for (int q = 0; q < c; ++q) m &= m - 1u;           // the c-th value present
          if (ch < d) eepf[i] = ld4(P.layers[l].etab[0] + (int64_t)__builtin_ctz(m) * d + ch);
        }
      }
    }
  }
  if (!DGL && P.n_layers > 0 && !tab_pending && !(efast && s_pro[1] == 0u)) {
    const int EF = P.edge_nf;
    // lead = first edge with my feature tuple.  Every thread walks ALL the edges with block-uniform (broadcast) LDS reads and no
    // early exit: the reads pipeline, where a scan that stops at the first match serialises one LDS round trip per candidate and
    // the whole wave waits for the rarest class's first edge.  Leaders of the three edge waves are published as ballots; the dense
    // class id of a leader is the number of leaders before it.
    __shared__ unsigned long long lmask[(GNN_EMAX + 63) / 64];
    int lead = -1;
    if (tid < GNN_EMAX) {                              // whole waves: the ballot below needs every lane of an edge wave
      if (tid < ne) {
        lead = tid;
        if (EF == 1) {
          const int mine = efeat[tid];
#pragma unroll 8
          for (int j = 0; j < ne; ++j) lead = (efeat[j] == mine && j < lead) ? j : lead;
        } else {
          for (int j = 0; j < ne; ++j) {
            bool same = true;
            for (int f = 0; f < EF; ++f) same = same && (efeat[j * EF + f] == efeat[tid * EF + f]);
            lead = (same && j < lead) ? j : lead;
          }
        }
        elead[tid] = lead;
      }
      const unsigned long long m = __ballot(tid < ne && lead == tid);
      if (lane == 0) lmask[tid >> 6] = m;
    }
    __syncthreads();
    ncls = 0;
#pragma unroll
    for (int w = 0; w < (GNN_EMAX + 63) / 64; ++w) ncls += __popcll(lmask[w]);
    if (tid < ne) {
      int c = 0;
#pragma unroll
      for (int w = 0; w < (GNN_EMAX + 63) / 64; ++w) {
        const int below = lead - 64 * w;               // leaders of word w that precede my leader
        const unsigned long long keep = below >= 64 ? ~0ull : (below > 0 ? (1ull << below) - 1 : 0ull);
        c += __popcll(lmask[w] & keep);
      }
      ecls[tid] = c;                                   // dense class id = number of leaders before my leader
      if (lead == tid && c < GNN_CLS) cedge[c] = lead;
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
  SN_STAMP(33);
  // ---------------------------------------------------------------- stage the slot sum (rho output), split, in SA
  if (rs_vec) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = tid + j * GNN_WAVES * 64;
      if (i < n * (D / 4)) {
        const int rr = i / (D / 4), c4 = i % (D / 4);
        sp_store4(SA, rr, c4 >> 2, c4 & 3, rs_v[j]);
      }
    }
  } else {
    for (int i = threadIdx.x; i < n * (D / 4); i += GNN_WAVES * 64) {
      const int rr = i / (D / 4), c4 = i % (D / 4);
      f32x4 v = zero4;
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) if (4 * c4 + qq < rho_w) v[qq] = S.rho_sum[(int64_t)(gs + rr) * rho_ld + 4 * c4 + qq];
      sp_store4(SA, rr, c4 >> 2, c4 & 3, v);
    }
  }
  SN_STAMP(34);
  // ---------------------------------------------------------------- input encoder -> SB (model.py:37)
  if (xid_pref) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int t = tr.t_lo + i;
      if (t < tr.t_hi && xid[i] != -1) {
        int ot, rt;
        tr.decode(t, ot, rt);
        sp_store4(SB, rt * 16 + li, ot, g, nrow[i]);
      }
    }
  } else if (P.node_discrete) {
    for (int t = tr.t_lo; t < tr.t_hi; ++t) {
      int ot, rt;
      tr.decode(t, ot, rt);
      const int row = rt * 16 + li, c = 16 * ot + 4 * g;
      if (row < n) {
        const int64_t* xi = reinterpret_cast<const int64_t*>(S.x) + (int64_t)(gs + row) * S.ldx;
        f32x4 s = zero4;
        for (int f = 0; f < P.node_nf; ++f) {
          int64_t xv = xi[f];
          if ((uint64_t)xv >= (uint64_t)P.node_vocab) { xv = 0; atomicOr(&S.status[3], 4); atomicOr(&s_pro[0], 1u); }    // see the edge features above
          const float* trow = P.ntab[f] + xv * d;
          if ((d & 3) == 0) { if (c < d) s += ld4(trow + c); }
          else {
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) if (c + qq < d) s[qq] += trow[c + qq];
          }
        }
        sp_store4(SB, row, ot, g, s);
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
        sp_store4(SB, row, ot, g, relu4(acc * ld4(P.n_scale + c) + ld4(P.n_shift + c)));
      }
    }
  }
  SN_STAMP(35);
  lds_barrier();                           // inputs staged; LDS-only like every later barrier:
  const int graph_bad = (int)s_pro[0];     // (did anyone see a bad feature id)
  SN_STAMP(1);                             // a __syncthreads() would also drain the weight prefetch in flight (vmcnt(0))
  // (round 5: read here, under the wait for the first weight tile, not in front of layer 0)
  int a_dg[4];
  unsigned a_sr[4], a_er[4];       // four source rows (< 64) / four edge classes or edge ids (< 256), a byte each
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a_dg[i] = -1;
    a_sr[i] = 0u;
    a_er[i] = 0u;
    const int t = tr.t_lo + i;
    if (!TF && t < tr.t_hi && (DGL || use_tab || use_ee)) {
      int ot, rt;
      tr.decode(t, ot, rt);
      const int row = rt * 16 + li;
      if (row < n) {
        const int e_lo = erow[row], dg = erow[row + 1] - e_lo;
        a_dg[i] = dg;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int ei = k < dg ? e_lo + k : 0;
          a_sr[i] |= (unsigned)(k < dg ? esrc[ei] : GNN_ROWS) << (8 * k);              // (a missing in-edge: the zero row of X1 ...
          if (!DGL) a_er[i] |= (unsigned)(k < dg ? (use_tab ? ecls[ei] : ei) : 255) << (8 * k);   //  ... and, marked 255, the zero row of EE)
        }
      }
    }
  }
  ee_fetch(0);     // needs efeat (staged above); the loads fly during the three Linears below
  // ---------------------------------------------------------------- h = Linear(cat[x, pos]) (model.py:39-40), pos = BN(W_out . slot_sum)
  //   (sign_net.py:71).  Order: x part (SB -> X1), pos (SA -> SB, SB being free after a barrier), pos part (SB -> X1 +=).
  const void* first_w = P.n_layers > 0 ? P.layers[0].w1s : P.head_w1;
  const TileRange first_tr = P.n_layers > 0 ? tr : hr;
  {
    // rho.out folded into the pos half of `linear` by the caller (lin_b = W_pos . diag(bn scale) . W_out, bias' = W_pos . bn shift + b):
    // h = lin_a . x + lin_b . slot_sum + bias' — two GEMMs over two images that are both complete; each lane parks its own
    // tiles of the first product in X1 and reads them back itself: no barrier between the two
    auto epi_a = [&](int rt, int ot, f32x4 acc, f32x4, f32x4, f32x4) { lds_st4(X1 + (rt * 16 + li) * LD + 16 * ot + 4 * g, acc); };
    auto epi_b = [&](int rt, int ot, f32x4 acc, f32x4 bias, f32x4, f32x4) {
      float* o = X1 + (rt * 16 + li) * LD + 16 * ot + 4 * g;
      lds_st4(o, (lds_ld4(o) + acc) + bias);
    };
    const bool hasl = P.n_layers > 0;
    if constexpr (TC > 0) coop_gemm_weave<NKB, TC>(pre, SB, wave, lane, epi_a, first_w, first_tr.first_ot());
    else if constexpr (ROLL) coop_gemm_roll<NKB>(pre, SB, tr, lane, epi_a, first_w, first_tr);
    else coop_gemm<NKB>(pre, alt, P.lin_a, SB, tr, lane, epi_a, P.lin_b, tr);
    SN_STAMP(20);
    if constexpr (TC > 0) coop_gemm_weave<NKB, TC>(alt, SA, wave, lane, epi_b, hasl ? P.layers[0].w2s : P.head_w2, hasl ? wave : 0);
    else if constexpr (ROLL) coop_gemm_roll<NKB>(alt, SA, tr, lane, epi_b, hasl ? P.layers[0].w2s : P.head_w2, hasl ? tr : h2);
    else coop_gemm<NKB>(pre, alt, P.lin_b, SA, tr, lane, epi_b, first_w, first_tr);
  }
  ee_store();
  lds_barrier();
  SN_STAMP(2);
  if constexpr (TF) {
    // ---------------------------------------------------------------- graph Transformer layers: h in X1 (fp32) and split in image A
    static_assert(ROLL, "one output tile per wave and Linear");
    for (int t = tr.t_lo; t < tr.t_hi; ++t) {
      int ot, rt;
      tr.decode(t, ot, rt);
      const int row = rt * 16 + li, c = 16 * ot + 4 * g;
      sp_store4(SA, row, ot, g, lds_ld4(X1 + row * LD + c));
    }
    lds_barrier();
    unsigned char* A = SA;
    unsigned char* B = SB;
    const float* Eg = reinterpret_cast<const float*>(S.edge_attr);
    const int an = (int)threadIdx.x >> 3, ah = (int)threadIdx.x & 7;          // the attention's (node, head) of this lane
    const float root = sqrtf(8.f);
    // my node's in-edges do not change from layer to layer: degree and the first four (source row, edge id) pairs are read ONCE, and a
    // layer's E rows of those edges are requested at the layer's entry — three Linear stages before the attention needs them
    int at_lo = 0, at_dg = 0, at_sr[4] = {0, 0, 0, 0};
    const float* at_er[4] = {Eg, Eg, Eg, Eg};
    if (an < n) {
      at_lo = erow[an];
      at_dg = erow[an + 1] - at_lo;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (k < at_dg) {
          at_sr[k] = esrc[at_lo + k];
          at_er[k] = Eg + (int64_t)ecls[at_lo + k] * S.lde + 8 * ah;
        }
      }
    }
    for (int l = 0; l < P.n_layers; ++l) {
      f32x4 pe0[4], pe1[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        pe0[k] = zero4; pe1[k] = zero4;
        if (k < at_dg) { pe0[k] = ld4(at_er[k] + l * D); pe1[k] = ld4(at_er[k] + l * D + 4); }
      }
      const sn_gnn_layer& Lp = P.layers[l];
      const bool lastl = l + 1 == P.n_layers;
      // the stage matrices in launch order, running on into the next layer / the readout (what every stage prefetches two ahead)
      auto seq = [&](int j) -> const void* {
        if (j < 8) return Lp.etab[j];
        if (!lastl) return P.layers[l + 1].etab[j - 8];
        return j == 8 ? P.head_w1 : S.head_mid;
      };
      auto seq_tr = [&](int j) { return (j < 8 || !lastl) ? tr : hr; };
      float* Qi = reinterpret_cast<float*>(B);            // Q | K | V rows, fp32, in the other image's space (3 * 64 * LD floats = one image)
      float* Ki = Qi + GNN_ROWS * LD;
      float* Vi = Ki + GNN_ROWS * LD;
      static_assert((size_t)3 * GNN_ROWS * LD * sizeof(float) <= (size_t)SP_IMAGE, "Q | K | V fit one split image");
      auto epi_q = [&](int rt, int ot, f32x4 acc, f32x4, f32x4, f32x4) { lds_st4(Qi + (rt * 16 + li) * LD + 16 * ot + 4 * g, acc); };
      auto epi_k = [&](int rt, int ot, f32x4 acc, f32x4, f32x4, f32x4) { lds_st4(Ki + (rt * 16 + li) * LD + 16 * ot + 4 * g, acc); };
      auto epi_v = [&](int rt, int ot, f32x4 acc, f32x4, f32x4, f32x4) { lds_st4(Vi + (rt * 16 + li) * LD + 16 * ot + 4 * g, acc); };
      coop_gemm_roll<NKB>(pre, A, tr, lane, epi_q, seq(2), seq_tr(2));
      coop_gemm_roll<NKB>(alt, A, tr, lane, epi_k, seq(3), seq_tr(3));
      coop_gemm_roll<NKB>(pre, A, tr, lane, epi_v, seq(4), seq_tr(4));
      lds_barrier();
      // the edge attention (layers/transformer.py:150-228; arithmetic of k_edge_attention, csrc/dgl_layers.hip): one lane per (node, head)
      if (an < n) {
        const f32x4 q0 = lds_ld4(Qi + an * LD + 8 * ah), q1 = lds_ld4(Qi + an * LD + 8 * ah + 4);
        f32x4 a0 = zero4, a1 = zero4;
        float z = 0.f;
        auto edge = [&](int sr, f32x4 e0, f32x4 e1) {
          const f32x4 k0 = lds_ld4(Ki + sr * LD + 8 * ah), k1 = lds_ld4(Ki + sr * LD + 8 * ah + 4);
          const f32x4 v0 = lds_ld4(Vi + sr * LD + 8 * ah), v1 = lds_ld4(Vi + sr * LD + 8 * ah + 4);
          float sc = 0.f;
#pragma unroll
          for (int c = 0; c < 4; ++c) sc += ((k0[c] * q0[c]) / root) * e0[c];
#pragma unroll
          for (int c = 0; c < 4; ++c) sc += ((k1[c] * q1[c]) / root) * e1[c];
          const float sw = expf(fminf(fmaxf(sc, -5.f), 5.f));
          z += sw;
#pragma unroll
          for (int c = 0; c < 4; ++c) { a0[c] += v0[c] * sw; a1[c] += v1[c] * sw; }
        };
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (k < at_dg) edge(at_sr[k], pe0[k], pe1[k]);              // edge-id order: the first four from the prefetched rows
        for (int e = at_lo + 4; e < at_lo + at_dg; ++e) {              // a node with more in-edges: the rest straight from memory
          const float* er = Eg + (int64_t)ecls[e] * S.lde + l * D + 8 * ah;
          edge(esrc[e], ld4(er), ld4(er + 4));
        }
        const float rz = 1.0f / (z + 1e-6f);
        sp_store4(A, an, ah >> 1, 2 * (ah & 1), a0 * rz);
        sp_store4(A, an, ah >> 1, 2 * (ah & 1) + 1, a1 * rz);
      }
      lds_barrier();
      // x1 = BatchNorm(x + O_h(a)) -> X1 and, split, the other image (Q | K | V are dead)
      auto epi_o = [&](int rt, int ot, f32x4 acc, f32x4 bias, f32x4 sc, f32x4 sh) {
        float* o = X1 + (rt * 16 + li) * LD + 16 * ot + 4 * g;
        f32x4 v = acc + bias;
        v = v + lds_ld4(o);
        v = v * sc + sh;
        lds_st4(o, v);
        sp_store4(B, rt * 16 + li, ot, g, v);
      };
      coop_gemm_roll<NKB>(alt, A, tr, lane, epi_o, seq(5), seq_tr(5));
      lds_barrier();
      // FFN layer 1 in two halves of 64 hidden channels: relu(W x1 + b) -> image A, channels [0, 64) and [64, 128)
      auto epi_f1a = [&](int rt, int ot, f32x4 acc, f32x4 bias, f32x4, f32x4) { sp_store4(A, rt * 16 + li, ot, g, relu4(acc + bias)); };
      auto epi_f1b = [&](int rt, int ot, f32x4 acc, f32x4 bias, f32x4, f32x4) { sp_store4(A, rt * 16 + li, ot + NT, g, relu4(acc + bias)); };
      coop_gemm_roll<NKB>(pre, B, tr, lane, epi_f1a, seq(6), seq_tr(6));
      coop_gemm_roll<NKB>(alt, B, tr, lane, epi_f1b, seq(7), seq_tr(7));
      lds_barrier();
      // FFN layer 2 in two halves of its 128-deep sum: the first half's sums parked in fp32, then x = BatchNorm(x1 + W f + b)
      auto epi_f2a = [&](int rt, int ot, f32x4 acc, f32x4, f32x4, f32x4) { lds_st4(PART + (rt * 16 + li) * LD + 16 * ot + 4 * g, acc); };
      auto epi_f2b = [&](int rt, int ot, f32x4 acc, f32x4 bias, f32x4 sc, f32x4 sh) {
        const int off = (rt * 16 + li) * LD + 16 * ot + 4 * g;
        f32x4 v = (lds_ld4(PART + off) + acc) + bias;
        v = v + lds_ld4(X1 + off);
        v = v * sc + sh;
        lds_st4(X1 + off, v);
        sp_store4(B, rt * 16 + li, ot, g, v);
      };
      coop_gemm_roll<NKB>(pre, A, tr, lane, epi_f2a, seq(8), seq_tr(8));
      coop_gemm_roll<NKB>(alt, A, tr, lane, epi_f2b, seq(9), seq_tr(9), 2);      // K blocks 2, 3 of the hidden rows
      lds_barrier();
      unsigned char* tsw = A; A = B; B = tsw;
    }
  } else {
  // ---------------------------------------------------------------- GINE layers: h lives in X1           (model.py:47-55)
  // The in-edges of my pairs' rows do not change from layer to layer: degree, the first four source rows and their edge classes
  // (or edge ids) are read ONCE — the aggregation of every layer then starts with its row reads instead of two dependent index
  // round trips per pair.
  for (int l = 0; l < P.n_layers; ++l) {
    const sn_gnn_layer& Lp = P.layers[l];
    // u = sum_{j->i} relu(h_j + e_ji) + (1+eps) h_i  for my (channel tile, row tile) pairs: X1 -> SA (split)
#ifdef SN_PROFILE
    pt = clock64();
#endif
    ee_fetch(l + 1);   // next layer's edge embeddings: in flight during this aggregation
    {
      const float sc = 1.f + *Lp.eps;
      if (DGL || use_tab || use_ee) {
        // my (up to four) pairs, one after the other; the in-edge indices of their rows were read once before the layer loop (a_*)
        const int eoff = use_tab ? l * ncls : 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (a_dg[i] >= 0) {
            const int t = tr.t_lo + i;
            int ot, rt;
            tr.decode(t, ot, rt);
            const int row = rt * 16 + li, c = 16 * ot + 4 * g, dg = a_dg[i];
            // the first four in-edges (molecular graphs: all) with unrolled reads: the eight row reads, then the adds in edge order; a
            // missing in-edge reads the two zero rows, relu(0 + 0) = +0 is added: no select on the values (round 5: 16 of a pair's ~95
            // vector instructions were those selects)
            f32x4 hv[4], ev[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int sr = (int)((a_sr[i] >> (8 * k)) & 255u), er = (int)((a_er[i] >> (8 * k)) & 255u);
              hv[k] = lds_ld4(X1 + sr * LD + c);
              if (!DGL) ev[k] = lds_ld4(EE + (er == 255 ? S.ee_rows : er + eoff) * LD + c);
            }
            f32x4 u = zero4;
#pragma unroll
            for (int k = 0; k < 4; ++k) u += DGL ? hv[k] : relu4(hv[k] + ev[k]);
            if (dg > 4) {
              const int e_lo = erow[row], e_hi = e_lo + dg;
              for (int e = e_lo + 4; e < e_hi; ++e) {
                if (DGL) { u += lds_ld4(X1 + esrc[e] * LD + c); continue; }
                const f32x4 ef = use_tab ? lds_ld4(EE + (l * ncls + ecls[e]) * LD + c) : lds_ld4(EE + e * LD + c);
                u += relu4(lds_ld4(X1 + esrc[e] * LD + c) + ef);
              }
            }
            {
#pragma clang fp contract(off)
              const f32x4 self = lds_ld4(X1 + row * LD + c) * sc;
              u = u + self;
            }
            sp_store4(SA, row, ot, g, u);
          }
          __builtin_amdgcn_sched_barrier(0);      // pairs stay sequential: four pairs' row reads at once would not fit the registers
        }
      } else {
#pragma unroll 1
        for (int t = tr.t_lo; t < tr.t_hi; ++t) {
          int ot, rt;
          tr.decode(t, ot, rt);
          const int row = rt * 16 + li, c = 16 * ot + 4 * g;
          if (row >= n) continue;
          f32x4 u = zero4;
          const int e_lo = erow[row], e_hi = erow[row + 1];
          for (int e = e_lo; e < e_hi; ++e) u += relu4(lds_ld4(X1 + esrc[e] * LD + c) + edge_embed(Lp, e, c));
          {
#pragma clang fp contract(off)
            const f32x4 self = lds_ld4(X1 + row * LD + c) * sc;
            u = u + self;
          }
          sp_store4(SA, row, ot, g, u);
        }
      }
    }
    lds_barrier();
    SN_ACCUM(9, pt);
#ifdef SN_PROFILE
    pt = clock64();
#endif
    ee_store();        // every wave is done reading this layer's embeddings
    // nn: Linear . BN . ReLU : SA -> SB
    const bool lastl = l + 1 == P.n_layers;
    auto epi_1 = [&](int rt, int ot, f32x4 acc, f32x4 sc, f32x4 sh, f32x4) { sp_store4(SB, rt * 16 + li, ot, g, relu4(acc * sc + sh)); };
    if constexpr (TC > 0) coop_gemm_weave<NKB, TC>(pre, SA, wave, lane, epi_1, lastl ? P.head_w1 : P.layers[lastl ? l : l + 1].w1s, wave);
    else if constexpr (ROLL) coop_gemm_roll<NKB>(pre, SA, tr, lane, epi_1, lastl ? P.head_w1 : P.layers[lastl ? l : l + 1].w1s, lastl ? hr : tr);
    else coop_gemm<NKB>(pre, alt, Lp.w1s, SA, tr, lane, epi_1, Lp.w2s, tr);
    lds_barrier();
    SN_ACCUM(11, pt);
#ifdef SN_PROFILE
    pt = clock64();
#endif
    // Linear ; BN . ReLU ; + previous_x : SB -> X1 (my tiles only: nobody else reads them at this point)
    auto epi_2 = [&](int rt, int ot, f32x4 acc, f32x4 sc, f32x4 sh, f32x4) {
      float* o = X1 + (rt * 16 + li) * LD + 16 * ot + 4 * g;
      if (DGL) lds_st4(o, acc * sc + sh);           // the MLP's last Linear: nothing behind it
      else lds_st4(o, relu4(acc * sc + sh) + lds_ld4(o));
    };
    if constexpr (TC > 0) coop_gemm_weave<NKB, TC>(alt, SB, wave, lane, epi_2, lastl ? P.head_w2 : P.layers[lastl ? l : l + 1].w2s, lastl ? 0 : wave);
    else if constexpr (ROLL) coop_gemm_roll<NKB>(alt, SB, tr, lane, epi_2, lastl ? (DGL ? S.head_mid : P.head_w2) : P.layers[lastl ? l : l + 1].w2s,
                                            lastl ? (DGL ? hr : h2) : tr);
    else coop_gemm<NKB>(pre, alt, Lp.w2s, SB, tr, lane, epi_2, lastl ? P.head_w1 : P.layers[lastl ? l : l + 1].w1s, lastl ? hr : tr);
    // TC > 0 (NT = 8): wave w owns channel tile w of EVERY row, in this Linear and in the next layer's aggregation alike — the rows the
    // aggregation gathers were written by this very wave (LDS operations of a wave execute in order): no workgroup barrier between
    // the two, only in front of the pooling, which reads all channels
    if (TC > 0 && !lastl) asm volatile("" ::: "memory");
    else lds_barrier();
    SN_ACCUM(12, pt);
    SN_STAMP(24 + l);
  }
  }
  SN_STAMP(3);
  // ---------------------------------------------------------------- add pooling -> row 0 of SA (rows 1..15: zero)   (model.py:57-61)
  // (sixteen interleaved partial sums per channel quad, then their sum in order: 4 + 16 dependent adds instead of n)
  static_assert(16 * (D / 4) <= GNN_WAVES * 64 && (size_t)16 * D * sizeof(float) <= (size_t)(SP_PLANE - 16 * SP_STRIDE), "one pass; the partial sums fit");
  // [16][D], parked in rows 16.. of image A's first plane: the head reads row tile 0 only, and unlike image B's — whose K padding of
  // rows 0-15 the head's second Linear reads and must find zero — nothing there is looked at again
  float* PS = reinterpret_cast<float*>(SA + 16 * SP_STRIDE);
  {
    const int pj = threadIdx.x / (D / 4), pc4 = threadIdx.x % (D / 4);
    if (pj < 16) {
      f32x4 s = zero4;
      for (int r = pj; r < n; r += 16) s += lds_ld4(X1 + r * LD + 4 * pc4);
      lds_st4(PS + pj * D + 4 * pc4, s);
    }
    lds_barrier();
    if (pj < 16) {
      f32x4 s = zero4;
      if (pj == 0) {
#pragma unroll
        for (int j = 0; j < 16; ++j) s += lds_ld4(PS + j * D + 4 * pc4);
        if (DGL && S.pool_mean) s = s / (float)n;
      }
      sp_store4(SA, pj, pc4 >> 2, pc4 & 3, s);
    }
  }
  lds_barrier();
  SN_STAMP(4);
  // ---------------------------------------------------------------- output encoder on the pooled row     (model.py:63)
  auto epi_h1 = [&](int rt, int ot, f32x4 acc, f32x4 sc, f32x4 sh, f32x4) { sp_store4(SB, li, ot, g, relu4(acc * sc + sh)); };
  auto epi_h2 = [&](int rt, int ot, f32x4 acc, f32x4 bias, f32x4, f32x4) {
    if (li == 0) {
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const int c = 4 * g + qq;
        if (c < P.n_out) S.y[(int64_t)gi * P.n_out + c] = graph_bad ? __uint_as_float(0x7fc00000u) : acc[qq] + bias[qq];
      }
    }
  };
  if constexpr (DGL) {
    // MLPReadout (layers/mlp_readout_layer.py): Linear . ReLU . Linear . ReLU . Linear — SA -> SB -> SA -> y
    auto epi_hm = [&](int rt, int ot, f32x4 acc, f32x4 sc, f32x4 sh, f32x4) { sp_store4(SA, li, ot, g, relu4(acc * sc + sh)); };
    if constexpr (ROLL) coop_gemm_roll<NKB>(pre, SA, hr, lane, epi_h1, wave == 0 ? P.head_w2 : nullptr, h2);
    else coop_gemm<NKB>(pre, alt, P.head_w1, SA, hr, lane, epi_h1, S.head_mid, hr);
    lds_barrier();
    if constexpr (ROLL) coop_gemm_roll<NKB>(alt, SB, hr, lane, epi_hm, nullptr, h2);
    else coop_gemm<NKB>(pre, alt, S.head_mid, SB, hr, lane, epi_hm, wave == 0 ? P.head_w2 : nullptr, h2);
    lds_barrier();
    if constexpr (ROLL) coop_gemm_roll<NKB>(pre, SA, h2, lane, epi_h2, nullptr, h2);
    else coop_gemm<NKB>(pre, alt, P.head_w2, SA, h2, lane, epi_h2, nullptr, h2);
  } else {
  if constexpr (ROLL) coop_gemm_roll<NKB>(pre, SA, hr, lane, epi_h1, nullptr, h2);
  else coop_gemm<NKB>(pre, alt, P.head_w1, SA, hr, lane, epi_h1, wave == 0 ? P.head_w2 : nullptr, h2);
  lds_barrier();
  if constexpr (ROLL) coop_gemm_roll<NKB>(alt, SB, h2, lane, epi_h2, nullptr, h2);
  else coop_gemm<NKB>(pre, alt, P.head_w2, SB, h2, lane, epi_h2, nullptr, h2);
  }
  SN_STAMP(5);
#ifdef SN_PROFILE
  if ((int)blockIdx.x == g_prof_block && threadIdx.x == 0) { g_prof[6] = n; g_prof[7] = ne; }
#endif
}

// One workgroup per graph; the last workgroup to finish reports the batch's flags to the host (no separate copy).
template <int NT, int MODE = 0>
__global__ __launch_bounds__(GNN_WAVES * 64, 2) void k_gnn_coop(GnnStruct S, sn_gnn_params P) {
  if constexpr (NT == 8 && MODE == 0) {
    // one instantiation per row-tile count from two on (a graph's nodes: 17-32, 33-48, 49-64; see coop_gemm_weave) — with a fourth
    // one for 1-16 nodes in the same kernel the compiler's allocation ended in 2.3 KB of private segment per lane; such a graph runs
    // the general form (its chain is the shortest of the batch anyway)
    const int n = S.graph_ptr[blockIdx.x + 1] - S.graph_ptr[blockIdx.x];
    switch ((n + 15) >> 4) {
      case 1: gnn_graph<NT, MODE, 1>(S, P); break;
      case 2: gnn_graph<NT, MODE, 2>(S, P); break;
      case 3: gnn_graph<NT, MODE, 3>(S, P); break;
      case 4: gnn_graph<NT, MODE, 4>(S, P); break;
      default:                                          // an empty / oversize graph: a NaN output row and its flag
        if (threadIdx.x == 0) atomicOr(&S.status[3], n > GNN_ROWS ? 1 : 8);
        if ((int)threadIdx.x < P.n_out) S.y[(int64_t)blockIdx.x * P.n_out + threadIdx.x] = __uint_as_float(0x7fc00000u);
        break;
    }
  } else {
    gnn_graph<NT, MODE>(S, P);
  }
  if (S.flags_host != nullptr) {
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();                                              // my flag updates are visible device-wide
      if (atomicAdd(&S.status[4], 1) == (int)gridDim.x - 1) {       // every other workgroup has passed its fence
        __threadfence();
        for (int i = 0; i < S.n_flags; ++i)
          S.flags_host[i] = __hip_atomic_load(&S.flags_src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence_system();
        __hip_atomic_store(&S.flags_host[S.n_flags], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // "ready": a host polling
        __threadfence_system();                                                                        // this word sees the flags
      }
    }
  }
}

template <int NT, int MODE = 0>
static int launch_gnn(const GnnStruct& S, const sn_gnn_params& P, int64_t B, hipStream_t st) {
  constexpr bool DGL = MODE != 0;
  constexpr int LD = 16 * NT + 4;
  const size_t base = (size_t)2 * SP_IMAGE + (size_t)((GNN_ROWS + 1) * LD) * sizeof(float) +
                      (size_t)(GNN_ROWS + 4 + GNN_EMAX * (3 + (P.n_layers > 0 ? P.edge_nf : 0)) + GNN_CLS) * sizeof(int);
  const size_t lds_cap = 160 * 1024 - 512;     // the kernel also has a few bytes of static LDS (__syncthreads_count)
  const size_t room = base < lds_cap ? lds_cap - base : 0;
  int ee_rows = (int)(room / ((size_t)LD * sizeof(float))) - 1;      // (one more row behind them: the zero row a missing in-edge reads)
  if (ee_rows > GNN_EEMAX) ee_rows = GNN_EEMAX;
  if (ee_rows < 0) ee_rows = 0;
  GnnStruct S2 = S;
  S2.ee_rows = MODE == 2 ? GNN_ROWS : ((!DGL && P.n_layers > 0) ? ee_rows : 0);        // (Transformer mode: the fp32 partial-sum image)
  const size_t lds = base + (size_t)(S2.ee_rows + 1) * LD * sizeof(float);
  static bool init = false;
  if (!init) {
    const size_t lds_max = lds_cap;
    if (lds_max > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_gnn_coop<NT, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_max) != hipSuccess)
      return fail(SN_ERR_LAUNCH, "sn_gnn_fused_f32: cannot raise the dynamic LDS limit to %zu", lds_max);
    init = true;
  }
  hipLaunchKernelGGL((k_gnn_coop<NT, MODE>), dim3((unsigned)B), dim3(GNN_WAVES * 64), lds, st, S2, P);
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
                                const int32_t* col, const int32_t* eperm, int32_t* status, float* y,
                                const int32_t* flags_src, int n_flags, int32_t* flags_host, void* stream) {
  SN_REQUIRE(params && x && rho_sum && graph_ptr && rowptr && status && y && B >= 0, "sn_gnn_fused_f32: null pointer");
  const sn_gnn_params& P = *params;
  SN_REQUIRE(P.d > 0 && P.d <= 128, "sn_gnn_fused_f32: hidden width %d not in (0, 128]", P.d);
  SN_REQUIRE(P.n_layers >= 0 && P.n_layers <= SN_GNN_MAX_LAYERS, "sn_gnn_fused_f32: %d layers unsupported", P.n_layers);
  SN_REQUIRE(P.n_out >= 1 && P.n_out <= 16, "sn_gnn_fused_f32: n_out=%d not in [1,16]", P.n_out);
  SN_REQUIRE(P.node_nf >= 1 && P.node_nf <= (P.node_discrete ? 10 : 16) && ldx >= P.node_nf,
             "sn_gnn_fused_f32: node feature count %d unsupported", P.node_nf);
  SN_REQUIRE(P.n_layers == 0 || (edge_attr && P.edge_nf >= 1 && P.edge_nf <= (P.edge_discrete ? 10 : 16) && lde >= P.edge_nf),
             "sn_gnn_fused_f32: edge feature count %d unsupported", P.edge_nf);
  SN_REQUIRE(P.lin_a && P.lin_b && P.head_w1 && P.head_w2, "sn_gnn_fused_f32: parameters missing");
  SN_REQUIRE(P.rho_out_w == nullptr, "sn_gnn_fused_f32: rho_out_w must be NULL — fold rho.out into lin_b (see signnet_hip.h)");
  if (P.node_discrete) { for (int f = 0; f < P.node_nf; ++f) SN_REQUIRE(P.ntab[f], "sn_gnn_fused_f32: node table %d missing", f); }
  else SN_REQUIRE(P.nw && P.n_scale && P.n_shift, "sn_gnn_fused_f32: node MLP parameters missing");
  for (int l = 0; l < P.n_layers; ++l) {
    const sn_gnn_layer& L = P.layers[l];
    SN_REQUIRE(L.w1s && L.w2s && L.eps, "sn_gnn_fused_f32: layer %d parameters missing", l);
    if (P.edge_discrete) { for (int f = 0; f < P.edge_nf; ++f) SN_REQUIRE(L.etab[f], "sn_gnn_fused_f32: layer %d edge table %d missing", l, f); }
    else SN_REQUIRE(L.ew && L.e_scale && L.e_shift, "sn_gnn_fused_f32: layer %d edge MLP parameters missing", l);
  }
  if (B == 0) return SN_OK;
  SN_REQUIRE(!flags_host || (flags_src && n_flags > 0 && n_flags <= 64), "sn_gnn_fused_f32: flag report needs flags_src and 0 < n_flags <= 64");
  SN_REQUIRE(!flags_src || n_flags >= 1, "sn_gnn_fused_f32: flags_src needs n_flags >= 1");
  SN_REQUIRE((!P.node_discrete || P.node_vocab > 0) && (P.n_layers == 0 || !P.edge_discrete || P.edge_vocab > 0),
             "sn_gnn_fused_f32: node_vocab / edge_vocab (rows of the embedding tables) missing");
  GnnStruct S{x, ldx, edge_attr, lde, rho_sum, graph_ptr, rowptr, col, eperm, status, y, 0, flags_src, n_flags, flags_host,
              P.d, P.d, nullptr, 0};
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

// The DGL tree's GIN net (gin_net.py:83-126, eval mode) on the same per-graph stage kernel: see gnn_graph<NT, DGL = true>.
extern "C" int sn_gin_net_fused_f32(const sn_gnn_params* params, const void* head_mid, int pool_mean, const int64_t* atom, const float* p,
                                    int ldp, int kp, const int32_t* graph_ptr, int64_t B, const int32_t* rowptr, const int32_t* col,
                                    const int32_t* eperm, int32_t* status, float* y, const int32_t* flags_src, int n_flags, void* stream) {
  SN_REQUIRE(params && head_mid && atom && p && graph_ptr && rowptr && col && eperm && status && y && B >= 0, "sn_gin_net_fused_f32: null pointer");
  const sn_gnn_params& P = *params;
  SN_REQUIRE(P.d > 0 && P.d <= 128 && (P.d & 3) == 0, "sn_gin_net_fused_f32: padded hidden width %d must be a multiple of 4 in (0, 128]", P.d);
  SN_REQUIRE(P.n_layers >= 1 && P.n_layers <= SN_GNN_MAX_LAYERS, "sn_gin_net_fused_f32: %d layers unsupported", P.n_layers);
  SN_REQUIRE(P.n_out >= 1 && P.n_out <= 16, "sn_gin_net_fused_f32: n_out=%d not in [1,16]", P.n_out);
  SN_REQUIRE(P.node_discrete == 1 && P.node_nf == 1 && P.ntab[0] && P.node_vocab > 0, "sn_gin_net_fused_f32: one atom-type table expected");
  SN_REQUIRE(P.edge_nf == 0, "sn_gin_net_fused_f32: the GIN net has no edge term (edge_nf must be 0)");
  SN_REQUIRE(P.lin_a && P.lin_b && P.head_w1 && P.head_w2 && P.rho_out_w == nullptr, "sn_gin_net_fused_f32: parameters missing");
  SN_REQUIRE(kp >= 1 && kp <= P.d && ldp >= kp, "sn_gin_net_fused_f32: positional encoding width %d unsupported", kp);
  for (int l = 0; l < P.n_layers; ++l) {
    const sn_gnn_layer& L = P.layers[l];
    SN_REQUIRE(L.w1s && L.w2s && L.eps, "sn_gin_net_fused_f32: layer %d parameters missing", l);
  }
  SN_REQUIRE(!flags_src || n_flags >= 1, "sn_gin_net_fused_f32: flags_src needs n_flags >= 1");
  if (B == 0) return SN_OK;
  GnnStruct S{atom, 1, nullptr, 0, p, graph_ptr, rowptr, col, eperm, status, y, 0, flags_src, n_flags, nullptr, ldp, kp, head_mid, pool_mean};
  hipStream_t st = (hipStream_t)stream;
  int rc = SN_OK;
  switch ((P.d + 15) / 16) {          // (the widths the shipped nets pad to; others round up to the next one at pack time)
    case 1: case 2: case 3: case 4: rc = launch_gnn<4, 1>(S, P, B, st); break;
    case 5: case 6: rc = launch_gnn<6, 1>(S, P, B, st); break;
    default: rc = launch_gnn<8, 1>(S, P, B, st); break;
  }
  if (rc != SN_OK) return rc;
  SN_CHECK_LAUNCH("sn_gin_net_fused_f32");
  return SN_OK;
}

// The DGL tree's sparse graph Transformer (transformer_net.py:88-140, eval mode, hidden 64, 8 heads) on the per-graph stage kernel:
// see gnn_graph<4, MODE = 2>.
extern "C" int sn_transformer_net_fused_f32(const sn_gnn_params* params, const void* head_mid, int pool_mean, const int64_t* atom,
                                            const float* p, int ldp, int kp, const float* e_proj, int lde, const int32_t* graph_ptr, int64_t B,
                                            const int32_t* rowptr, const int32_t* col, const int32_t* eperm, int32_t* status, float* y,
                                            const int32_t* flags_src, int n_flags, void* stream) {
  SN_REQUIRE(params && head_mid && atom && p && e_proj && graph_ptr && rowptr && col && eperm && status && y && B >= 0,
             "sn_transformer_net_fused_f32: null pointer");
  const sn_gnn_params& P = *params;
  SN_REQUIRE(P.d == 64, "sn_transformer_net_fused_f32: hidden width %d (the stage kernel is written for 64 = 8 heads of 8)", P.d);
  SN_REQUIRE(P.n_layers >= 1 && P.n_layers <= SN_GNN_MAX_LAYERS, "sn_transformer_net_fused_f32: %d layers unsupported", P.n_layers);
  SN_REQUIRE(P.n_out >= 1 && P.n_out <= 16, "sn_transformer_net_fused_f32: n_out=%d not in [1,16]", P.n_out);
  SN_REQUIRE(P.node_discrete == 1 && P.node_nf == 1 && P.ntab[0] && P.node_vocab > 0, "sn_transformer_net_fused_f32: one atom-type table expected");
  SN_REQUIRE(P.edge_nf == 0, "sn_transformer_net_fused_f32: edge_nf must be 0 (the edge term is e_proj)");
  SN_REQUIRE(P.lin_a && P.lin_b && P.head_w1 && P.head_w2 && P.rho_out_w == nullptr, "sn_transformer_net_fused_f32: parameters missing");
  SN_REQUIRE(kp >= 1 && kp <= P.d && ldp >= kp && lde >= P.n_layers * P.d && (lde & 3) == 0 && (reinterpret_cast<uintptr_t>(e_proj) & 15) == 0,
             "sn_transformer_net_fused_f32: positional encoding / edge projection shapes");
  for (int l = 0; l < P.n_layers; ++l) {
    const sn_gnn_layer& L = P.layers[l];
    for (int j = 0; j < 8; ++j) SN_REQUIRE(L.etab[j], "sn_transformer_net_fused_f32: layer %d stage matrix %d missing", l, j);
    SN_REQUIRE(L.w1s == (const void*)L.etab[0] && L.w2s == (const void*)L.etab[1], "sn_transformer_net_fused_f32: w1s / w2s must repeat etab[0] / etab[1]");
  }
  SN_REQUIRE(!flags_src || n_flags >= 1, "sn_transformer_net_fused_f32: flags_src needs n_flags >= 1");
  if (B == 0) return SN_OK;
  GnnStruct S{atom, 1, e_proj, lde, p, graph_ptr, rowptr, col, eperm, status, y, 0, flags_src, n_flags, nullptr, ldp, kp, head_mid, pool_mean};
  const int rc = launch_gnn<4, 2>(S, P, B, (hipStream_t)stream);
  if (rc != SN_OK) return rc;
  SN_CHECK_LAUNCH("sn_transformer_net_fused_f32");
  return SN_OK;
}
