// fused_phi.hip — phi(x) + phi(-x) for every valid (node, eigenvector slot) row, all layers in ONE launch.
//
// Replaces GNN3d.forward applied to x and -x (Alchemy/sign_net/sign_net.py:28-44,113;
// GINESignNetPyG/core/sign_net.py:30-48,115) in eval mode.  Why one kernel: with BatchNorm folded to
// an affine, every op of every phi layer is local to one (graph, slot, sign) slab of n_graph <= 64
// rows, so the [rows, d] activations never need to touch HBM between layers.
//
// Mapping (SN_PHI_BIN_ROWS = 64 rows per workgroup, 8 waves = two per SIMD, ONE workgroup per CU):
//   * graphs are packed into columns (sn_batch_plan: best-fit-decreasing on the node count, <= 64 rows); bin j of a
//     column holds the slot-j slab of every member graph; wave w < 4 owns bin rows [16w, 16w+16) of phi(+x), wave w+4 the
//     same rows of phi(-x) — the two co-resident waves of a SIMD; a wave whose tile has no rows skips the GEMMs.  Both
//     signs of a bin are resident together: the weights stream through the LDS ring ONCE per bin for all eight waves, and
//     phi(x)+phi(-x) is formed at the end of the bin through the LDS image, so `out` is written exactly once
//     (HBM traffic = the eigenvector scalars in + the [valid rows, d] sum out);
//   * a row tile lives in registers in the MFMA operand layout of common.hpp
//     (lane = (row = l&15, g = l>>4) holds channels 16*kk + 4*g + t), so the accumulators of one GEMM
//     are directly the operand of the next — no LDS round trip between the two Linears of a MaskedMLP;
//   * the GIN neighbour sum goes through LDS: every wave writes its rows to X[sign][row][ch],
//     one barrier, then each lane gathers its CSR neighbours' rows (ds_read_b128) — the reference's
//     [K,E,d] gather + scatter_add (masked_layers.py:75) becomes LDS traffic; the residual `+ previous_x`
//     (sign_net.py:42) is re-read from the same LDS image;
//   * weights: split-bf16 chunks staged once per workgroup in an LDS ring by LDS-DMA (fused_common.hpp).
// Why two single-tile waves per SIMD rather than one wave carrying both signs as two MFMA tiles (tried: 166 us against
// 136 us): a lone wave per SIMD is issue-bound — per 48 MFMAs it also issues ~300 other instructions (operand split,
// epilogue, fragment reads, AGPR moves at 400 registers) and nothing else can fill the matrix pipe meanwhile.
// Bound: bf16 MFMA / 6 (exact three-way split of both operands, six partial products, fp32 accumulate); algorithmic
// flops per valid row: 2 signs * (L-1) layers * 2 GEMMs * 2*d*d.
#include <type_traits>

#include "fused_common.hpp"

namespace sn {

constexpr int PHI_R = SN_PHI_BIN_ROWS;  // rows per bin / workgroup
constexpr int PHI_WAVES = 2 * (PHI_R / 16);   // one wave per (16-row tile, sign)

// From three output tiles on, the GEMMs run on wg_gemm_split_lag (fused_common.hpp): a four-slot weight ring with a look-ahead of
// three chunks, no LDS drain in front of the chunk barrier, and the two sign waves of a SIMD half a tile out of phase at every
// barrier (the + waves take it in front of the tile's last K block, the - waves in front of its first: phi_lag_kb).
// (-DSN_PHI_NOLAG: the one-stream form of rounds 2-3, for A/B runs: profiles/scripts/ab.sh.)
constexpr bool phi_lagged(int nt) {
#ifdef SN_PHI_NOLAG
  return false;
#else
  return nt >= 3;
#endif
}
// (round 5, A/B on two boxes with -DSN_PHI_LAGKB=0 / 1 / 2, the - waves' barrier in front of K block 0 / 1 / 2 of their tile: 0 is 1-2 %
//  faster than round 4's 1 (121.1 against 123.1 us, 110.3 against 111.4), 2 is 2-3 % slower — the two waves of a SIMD a whole tile
//  apart at every barrier instead of half a tile)
#ifdef SN_PHI_LAGKB
constexpr int phi_lag_kb(int nkb) { return SN_PHI_LAGKB < nkb ? SN_PHI_LAGKB : nkb - 1; }
#else
constexpr int phi_lag_kb(int) { return 0; }
#endif

struct PhiStruct {
  const float* ev;
  const int32_t* graph_ptr;
  const int64_t* evoff;
  const int32_t* rowptr;
  const int32_t* col;
  const int32_t* bin_mem;    // [nbins][8][2] member records of every bin (sn_plan_bins.phi_bin_mem): graph | slot << 13 | row offset << 19 |
                             //               (rows - 1) << 25 (-1 = none), first node of the graph
  const int32_t* meta;       // [7] nbins, [1] error
  int64_t max_bins;
  int kmax;
  int K;
  float* out;
  int dense_ld;              // DGL variant: the input is a dense [N, dense_ld] matrix of positional encodings (row = node), not eigenvectors
};

#ifdef SN_TIMELINE   // scratch builds only: a few wall-clock stamps (s_memrealtime, 100 MHz) per workgroup
static __device__ long long g_tl[1024][8];
#define SN_TL(i) do { if (threadIdx.x == 0 && blockIdx.x < 1024) g_tl[blockIdx.x][i] = wall_clock64(); } while (0)
#else
#define SN_TL(i) do { } while (0)
#endif

constexpr int PHI_NBR = 8;   // in-neighbours of a row kept in LDS (more: read from the CSR in global memory)
// Row descriptors of PHI_GB bins at a time: a bin's decode (bin -> member records -> node -> CSR range, eigenvector entry, first
// in-neighbours) is a chain of dependent global loads (five levels through the planner's columns until round 4: ~7 k cycles per
// bin; three since the planner writes per-bin member records); the eight waves of the workgroup decode the next eight bins of the
// workgroup in one pass — one wave per bin — and park the result in LDS.
constexpr int PHI_GB = PHI_WAVES;
constexpr int PHI_DESC_BYTES = 5 * PHI_GB * PHI_R * 4 + PHI_GB * PHI_R + PHI_GB * PHI_R + PHI_GB * PHI_R * PHI_NBR;

// HID1: layer 0 is Linear(1->1).BN.ReLU.Linear(1->d) (GINESignNetPyG) — no [d,d] GEMM in layer 0; else Linear(1->d)...Linear(d->d)
// (Alchemy).  A template parameter so that the variant without the layer-0 GEMM does not carry its registers.
// DGL: the GraphPrediction tree's GIN (layers/gnns.py:81-114 inside deepsigns.py:33-86) with eval-mode BatchNorms folded at pack
// time: per layer  aggregate -> relu(W0 a + b0) -> (W1' h + b1') * s + t   (no ReLU, no residual after the second Linear; the
// BatchNorm that follows the ReLU is folded into W1', b1', the one in front of the next layer is (s, t)); all |kmax| slots of every
// node are evaluated (zero-padded eigenvector columns included), the input is a dense [N, K] matrix, and the output rows are
// P.reserved (= phi_out_dim) wide.  Instantiated separately so that the PyG kernels carry none of it.
template <int NT, bool HID1, bool DGL = false>
__global__ __launch_bounds__(PHI_WAVES * 64, 2) void k_phi_fused(PhiStruct S, sn_phi_params P) {
  constexpr int D = 16 * NT;
  constexpr int LD = D + 4;  // +4 floats: conflict-free ds_write_b128 of 8 consecutive rows
  constexpr int NKB = (NT + 1) / 2;
  constexpr bool LAG = phi_lagged(NT);
  using Ring = WRing<NT, PHI_WAVES, LAG ? LAG_RING : SPLIT_RING>;
  extern __shared__ __align__(1024) unsigned char lds_raw[];
  float* X2 = reinterpret_cast<float*>(lds_raw + Ring::BYTES);  // [2 signs][PHI_R][LD]  x_l
  float* dxs = X2 + 2 * PHI_R * LD;                             // [GB][PHI_R] scalar eigenvector entries (layer 0)
  int* dnode = reinterpret_cast<int*>(dxs + PHI_GB * PHI_R);    // [GB][PHI_R] node of the row, -1 = no row
  int* delo = dnode + PHI_GB * PHI_R;                           // [GB][PHI_R] first in-edge (CSR position)
  int* ddeg = delo + PHI_GB * PHI_R;                            // [GB][PHI_R] in-degree
  int* dgs = ddeg + PHI_GB * PHI_R;                             // [GB][PHI_R] first node of the row's graph
  unsigned char* dslot = reinterpret_cast<unsigned char*>(dgs + PHI_GB * PHI_R);   // [GB][PHI_R] eigenvector slot of the row's slab
  unsigned char* drow0 = dslot + PHI_GB * PHI_R;                // [GB][PHI_R] bin row of the graph's first node
  unsigned char* dnbr = drow0 + PHI_GB * PHI_R;                 // [GB][PHI_R][PHI_NBR] bin rows of the first in-neighbours
  float* l0v = reinterpret_cast<float*>(lds_raw + Ring::BYTES + 2 * PHI_R * LD * sizeof(float) + PHI_DESC_BYTES);   // [4][D] layer-0 vectors
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sg = wave >> 2;                                     // 0: phi(+x), 1: phi(-x)   (wave-uniform)
  float* X = X2 + sg * PHI_R * LD;                              // the image of my sign
  const int r = (wave & 3) * 16 + (lane & 15), g = lane >> 4;
  SN_TL(0);
  // Round 5 (end): the member records of the first group of bins are requested BEFORE the plan's flag words are known (bin ids are
  // arithmetic; the record array has phi_max_bins rows, a row past the bin count is read and ignored) and as FOUR wide loads: written
  // as `rec[2 k]`, `rec[2 k + 1]` inside the member loop they compiled to sixteen single-dword loads, each behind the previous one's
  // round trip and a branch — 2 of the 2.9 us the first decode pass took (profiles/r05_phi_timeline.txt).
  auto load_rec = [&](int dbin_, int4 (&rq_)[4]) {
    int z = 0;
    asm volatile("" : "+v"(z));      // (vector loads on both paths: as scalar loads the sixteen words cost sixteen of the kernel's last scalar registers)
    const int4* rp = reinterpret_cast<const int4*>(S.bin_mem + (int64_t)((int64_t)dbin_ < S.max_bins ? dbin_ : 0) * 16) + z;
    rq_[0] = rp[0]; rq_[1] = rp[1]; rq_[2] = rp[2]; rq_[3] = rp[3];
  };
  int4 rq[4];
  load_rec((int)blockIdx.x + wave * (int)gridDim.x, rq);
  const int nb_raw = S.meta[7];
  const int merr = S.meta[1];   // a graph has more than 64 nodes: no bins here (folded into the bin count: an early `return` in front
                                // of the barriers did not compile — "illegal VGPR to SGPR copy"), the host falls back to the layer path
  // The per-channel vectors of layer 0 (every bin starts with them: four L2 round trips per lane and bin, 3.6 k cycles of a 78 k-cycle
  // bin) are copied into LDS once per workgroup: [w | b | scale | shift][D].  Requested with the records and the flag words — one
  // round trip for all three — and stored BEFORE the weight stream starts (behind it the compiler would wait for every LDS-DMA in
  // flight in front of an LDS store: it cannot tell the ring from these rows).  Straight-line: threads past the vectors repeat
  // thread 0's float4, a missing vector (no bias) reads the first one and stores zeros.
  static_assert(4 * (16 * NT / 4) <= PHI_WAVES * 64, "one float4 per thread covers the four vectors");
  f32x4 l0r;
  bool l0has;
  const int l0i = (int)threadIdx.x < 4 * (D / 4) ? (int)threadIdx.x : 0;
  {
    const float* src[4];
    if (HID1) { src[0] = reinterpret_cast<const float*>(P.l0_w2); src[1] = P.l0_bias2; src[2] = P.l0_bn_scale; src[3] = P.l0_bn_shift; }
    else { src[0] = P.l0_w1; src[1] = nullptr; src[2] = P.l0_bn0_scale; src[3] = P.l0_bn0_shift; }
    const int v = l0i / (D / 4), c = 4 * (l0i - v * (D / 4));
    typedef __attribute__((address_space(1))) const float gfloat_t;      // (a pointer picked by lane is generic: a flat load otherwise)
    const float* sp_ = v == 0 ? src[0] : (v == 1 ? src[1] : (v == 2 ? src[2] : src[3]));
    l0has = sp_ != nullptr;
    gfloat_t* gp_ = (gfloat_t*)(l0has ? sp_ : src[0]);
    l0r = f32x4{gp_[c], gp_[c + 1], gp_[c + 2], gp_[c + 3]};
  }
  const int nbins = merr != 0 ? 0 : nb_raw;
  { SN_PROF_ON(true); SN_STAMP(12); }
#ifdef SN_PROFILE
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    for (int i = 3; i < 64; ++i) g_prof[i] = 0;
    g_prof[12] = clock64();
  }
#endif
  lds_st4(l0v + 4 * l0i, l0has ? l0r : f32x4{0.f, 0.f, 0.f, 0.f});
  Ring ring;
  ring.init(lds_raw, wave, lane);
  // first [d,d] Linear of a (bin, sign) pass — where the weight stream (re)starts
  const void* wfirst = !HID1 ? P.l0_w2 : (P.n_layers > 1 ? P.layers[0].w1s : nullptr);
  if (NT >= SPLIT_RING && wfirst != nullptr && nbins > (int)blockIdx.x) {
    // (LAG: issue only — the first chunks land under the decode pass below, whose closing __syncthreads() waits for them and
    //  publishes them)
    if constexpr (LAG) ring_prologue3_issue(ring, wfirst); else ring.prologue(wfirst, NT);
  }

  SN_TL(1);
#ifdef SN_TIMELINE
  int tl_bin = 0;
#endif
  for (int base = blockIdx.x; base < nbins; base += gridDim.x * PHI_GB) {
    // ------------------------------------------------------------------ decode pass: wave w -> bin base + w * gridDim.x, lane -> row
    if (base != (int)blockIdx.x) __syncthreads();   // the previous group is done with the descriptors
    {
      const int dbin = base + wave * (int)gridDim.x;
      if (base != (int)blockIdx.x) load_rec(dbin, rq);     // (the first group's: requested at the kernel's start)
      if (dbin < nbins) {
        const int rr = lane;
        // (the eight member records: sixteen words already in registers — no load, no branch between the members)
        const int rw[16] = {rq[0].x, rq[0].y, rq[0].z, rq[0].w, rq[1].x, rq[1].y, rq[1].z, rq[1].w,
                            rq[2].x, rq[2].y, rq[2].z, rq[2].w, rq[3].x, rq[3].y, rq[3].z, rq[3].w};
        int node = -1, gs = 0, row0 = 0, gsel = 0, nsel = 0, slot = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int w0 = rw[2 * k], g0 = rw[2 * k + 1];
          const int off = (w0 >> 19) & 63, n = ((w0 >> 25) & 63) + 1, sl = (w0 >> 13) & 63;
          if (w0 >= 0 && rr >= off && rr < off + n && sl < S.K) {      // (sl < K: a caller's K smaller than the batch's slot count must not write outside `out`)
            node = g0 + (rr - off);
            gs = g0;
            row0 = off;
            gsel = w0 & 8191;
            nsel = n;
            slot = sl;
          }
        }
        // level 2: the row's CSR range and its graph's eigenvector block; level 3: the eigenvector entry and the first PHI_NBR
        // in-neighbours — all of a level requested before any of it is used.  (As a loop over the row's degree the neighbour reads
        // compiled to one load, one wait and one byte store per in-edge: three or four more round trips of every first decode pass.)
        int e_lo = 0, e_hi = 0;
        long long eo = 0;
        if (node >= 0) {
          e_lo = S.rowptr[node];
          e_hi = S.rowptr[node + 1];
          if (!DGL) eo = S.evoff[gsel];
        }
        const int deg = e_hi - e_lo;
        // (an unconditional load from a clamped address, the row's validity applied at the store: as `valid ? load : 0` the select —
        //  and with it the wait for the entry — was placed in front of the neighbour reads)
        const float xraw = S.ev[node >= 0 ? (DGL ? (int64_t)node * S.dense_ld + slot : eo + (int64_t)(node - gs) * nsel + slot) : (int64_t)0];
        // (one predicated region, eight loads from addresses clamped to the row's own segment: with one predicate per load the compiler
        //  put a partial wait between them)
        int cv[PHI_NBR];
#pragma unroll
        for (int e = 0; e < PHI_NBR; ++e) cv[e] = gs;
        if (deg > 0) {
          const int32_t* cp = S.col + e_lo;
#pragma unroll
          for (int e = 0; e < PHI_NBR; ++e) cv[e] = cp[e < deg ? e : deg - 1];
        }
        const float xval = node >= 0 ? xraw : 0.f;
        unsigned nlo = 0u, nhi = 0u;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          nlo |= (unsigned)((row0 + cv[e] - gs) & 255) << (8 * e);
          nhi |= (unsigned)((row0 + cv[e + 4] - gs) & 255) << (8 * e);
        }
        static_assert(PHI_NBR == 8, "the neighbour bytes of a row are one 8-byte store");
        int o = wave * PHI_R + rr;
        asm volatile("" : "+v"(o));      // (else the six descriptor addresses are hoisted out of the bin loop and one of them spills)
        dxs[o] = xval; dnode[o] = node; delo[o] = e_lo; ddeg[o] = deg; dgs[o] = gs; drow0[o] = (unsigned char)row0;
        dslot[o] = (unsigned char)slot;
        *reinterpret_cast<uint2*>(dnbr + o * PHI_NBR) = make_uint2(nlo, nhi);
      }
    }
    __syncthreads();
#ifdef SN_TIMELINE
    if (base == (int)blockIdx.x) SN_TL(2);
#endif
#pragma unroll 1
    for (int lb = 0; lb < PHI_GB; ++lb) {
    const int bin = base + lb * (int)gridDim.x;
    if (bin >= nbins) break;
    SN_PROF_ON(bin == (int)blockIdx.x);
    SN_STAMP(0);
#ifdef SN_PROFILE
    long long pt = 0;
#endif
    // ---------------------------------------------------------------- my row, from the descriptors
    const float* xs = dxs + lb * PHI_R;
    const unsigned char* nbr = dnbr + lb * PHI_R * PHI_NBR;
    const int e_lo = delo[lb * PHI_R + r], deg = ddeg[lb * PHI_R + r];
    // (bin row of in-neighbour number e >= PHI_NBR: rare — read from the CSR, with the graph's first node / first bin row from LDS)
    auto far_nbr = [&](int e) { return (int)drow0[lb * PHI_R + r] + S.col[e_lo + e] - dgs[lb * PHI_R + r]; };
    const float xval = xs[r];
    const bool valid = dnode[lb * PHI_R + r] >= 0;
    const bool wave_live = __ballot(valid) != 0ull;   // a 16-row tile without rows skips all MFMAs (keeps barriers + DMA)
    lds_barrier();   // the previous bin is done with X
    float* XR = X + r * LD;                 // my row
    SN_STAMP(1);
    // ---------------------------------------------------------------- layer 0 aggregate (scalar input, sign-free)
    const uint2 nb8 = *reinterpret_cast<const uint2*>(nbr + r * PHI_NBR);   // my first 8 in-neighbours (bin rows), read once
    float a0 = 0.f;
    for (int e = 0; e < deg; ++e) a0 += xs[e < PHI_NBR ? (int)(((e < 4 ? nb8.x : nb8.y) >> (8 * (e & 3))) & 255u) : far_nbr(e)];
    {
#pragma clang fp contract(off)
      const float sc = 1.f + *P.l0_eps;
      const float self = xval * sc;
      a0 = a0 + self;
    }
    SN_STAMP(2);
    // The layers, once for a wave with rows and once for a wave whose tile is empty (it only keeps the workgroup's barriers and
    // its share of the weight stream going): two instantiations under a wave-uniform branch, so that the row state (in, o, sp)
    // is local to the live one — as `if (wave_live)` blocks inside one body it was loop-carried across bins (64 registers).
    auto layers = [&](auto live_c, auto role_c) {
      constexpr bool LIVE = decltype(live_c)::value;
      constexpr int ROLE = decltype(role_c)::value;       // 1: a + wave (chunk barrier late in the tile), 0: a - wave (early)
      constexpr int BARKB = ROLE == 0 ? phi_lag_kb(NKB) : NKB - 1;
      // (compiler fence: without it the layer-0 vectors — loop invariant — are hoisted out of the bin loop into
      //  ~128 VGPRs that then live in scratch for the whole kernel)
      asm volatile("" ::: "memory");
      const float as = sg ? -a0 : a0;          // phi(-x): the aggregate of -x is exactly -(aggregate of x)   (sg: my wave's sign)
      f32x4 in[NT], o[NT];
      Split8 sp[NKB];
#ifdef SN_PROFILE
      pt = clock64();
#endif
      // -------------------------------------------------------------- layer 0
      if (HID1) {
        // Linear(1->1) . BN . ReLU . Linear(1->d) [+b] . BN . ReLU          (core/sign_net.py:20, masked_layers.py:54-64)
        if (LIVE) {
          const float w1 = P.l0_w1[0], s0 = P.l0_bn0_scale[0], h0 = P.l0_bn0_shift[0];
          const float t = fmaxf((as * w1) * s0 + h0, 0.f);
#pragma unroll
          for (int kk = 0; kk < NT; ++kk) {
            const int c = 16 * kk + 4 * g;
            const f32x4 w2 = lds_ld4(l0v + c), b2 = lds_ld4(l0v + D + c), s1 = lds_ld4(l0v + 2 * D + c), h1 = lds_ld4(l0v + 3 * D + c);
            in[kk] = relu4((t * w2 + b2) * s1 + h1);     // (b2: zeros when the Linear has no bias — x + 0 is exact)
          }
        }
      } else {
        // Linear(1->d) . BN . ReLU . Linear(d->d) [+b] . BN . ReLU           (Alchemy sign_net.py:20)
        if (LIVE) {
#pragma unroll
          for (int kk = 0; kk < NT; ++kk) {
            const int c = 16 * kk + 4 * g;
            const f32x4 w1 = lds_ld4(l0v + c), s0 = lds_ld4(l0v + 2 * D + c), h0 = lds_ld4(l0v + 3 * D + c);
            o[kk] = relu4((as * w1) * s0 + h0);
          }
          split_rows<NT>(o, sp);
        }
        const void* nxt = P.n_layers > 1 ? P.layers[0].w1s : wfirst;
        auto epi0 = [&](int ot, f32x4 acc, f32x4 b2, f32x4 s1, f32x4 h1, f32x4) {
          const f32x4 v = (acc + b2) * s1 + h1;
          in[ot] = DGL ? v : relu4(v);
        };
        if constexpr (LAG) wg_gemm_split_lag<NT, NT, BARKB>(ring, P.l0_w2, nxt, LIVE, sp, NoPre(), epi0);
        else wg_gemm_split<NT, NT, false>(ring, P.l0_w2, nxt, LIVE, sp, NoPre(), epi0);
      }
      if (LIVE && !valid) {
#pragma unroll
        for (int kk = 0; kk < NT; ++kk) in[kk] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      SN_ACCUM(3, pt);
      // -------------------------------------------------------------- layers 1 .. L-1
#pragma unroll 1
      for (int l = 1; l < P.n_layers; ++l) {
        const sn_phi_layer& Lp = P.layers[l - 1];
#ifdef SN_PROFILE
        pt = clock64();
#endif
        // publish x_l for the neighbour sums and the residual
        if (LIVE) {
#pragma unroll
          for (int kk = 0; kk < NT; ++kk) lds_st4(XR + 16 * kk + 4 * g, in[kk]);
        }
        lds_barrier();
        SN_ACCUM(4, pt);
#ifdef SN_PROFILE
        pt = clock64();
#endif
        if (LIVE) {
          // GIN aggregate: sum of in-neighbours (edge-id order), then + (1+eps) * self.  The first four neighbours
          // (molecular graphs: all of them) are gathered with predicated, fully unrolled reads so that the LDS
          // latency of one neighbour hides behind the next; a missing neighbour adds +0.
          const float* n0 = X + (deg > 0 ? (int)(nb8.x & 255u) : r) * LD + 4 * g;
          const float* n1 = X + (deg > 1 ? (int)((nb8.x >> 8) & 255u) : r) * LD + 4 * g;
          const float* n2 = X + (deg > 2 ? (int)((nb8.x >> 16) & 255u) : r) * LD + 4 * g;
          const float* n3 = X + (deg > 3 ? (int)(nb8.x >> 24) : r) * LD + 4 * g;
          const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kk = 0; kk < NT; ++kk) {
            const f32x4 v0 = lds_ld4(n0 + 16 * kk), v1 = lds_ld4(n1 + 16 * kk);
            f32x4 a = deg > 0 ? v0 : zero4;
            a += deg > 1 ? v1 : zero4;
            o[kk] = a;
          }
          // third and fourth neighbour: only the lanes that have one read (this loop is bound by LDS delivery, and on molecular
          // graphs most rows have degree <= 2)
          if (deg > 2) {
#pragma unroll
            for (int kk = 0; kk < NT; ++kk) o[kk] += lds_ld4(n2 + 16 * kk);
          }
          if (deg > 3) {
#pragma unroll
            for (int kk = 0; kk < NT; ++kk) o[kk] += lds_ld4(n3 + 16 * kk);
          }
          for (int e = 4; e < deg; ++e) {
            const int nb = e < PHI_NBR ? (int)((nb8.y >> (8 * (e - 4))) & 255u) : far_nbr(e);
            const float* np = X + nb * LD + 4 * g;
#pragma unroll
            for (int kk = 0; kk < NT; ++kk) o[kk] += lds_ld4(np + 16 * kk);
          }
          {
#pragma clang fp contract(off)
            const float sc = 1.f + *Lp.eps;
#pragma unroll
            for (int kk = 0; kk < NT; ++kk) {
              const f32x4 sf = in[kk] * sc;
              o[kk] = o[kk] + sf;
            }
          }
          SN_ACCUM(5, pt);
#ifdef SN_PROFILE
          pt = clock64();
#endif
          split_rows<NT>(o, sp);
          asm volatile("" :: "v"(sp[0].h), "v"(sp[NKB - 1].l));
          SN_ACCUM(6, pt);
        }
#ifdef SN_PROFILE
        pt = clock64();
#endif
        // MaskedMLP: Linear . BN . ReLU . Linear [+b]
        auto epi1 = [&](int ot, f32x4 acc, f32x4 s0, f32x4 h0, f32x4, f32x4) { o[ot] = relu4(acc * s0 + h0); };
        if constexpr (LAG) wg_gemm_split_lag<NT, NT, BARKB>(ring, Lp.w1s, Lp.w2s, LIVE, sp, NoPre(), epi1);
        else wg_gemm_split<NT, NT, false>(ring, Lp.w1s, Lp.w2s, LIVE, sp, NoPre(), epi1);
        SN_ACCUM(7, pt);
#ifdef SN_PROFILE
        pt = clock64();
#endif
        if (LIVE) {
          split_rows<NT>(o, sp);
          asm volatile("" :: "v"(sp[0].h), "v"(sp[NKB - 1].l));
        }
        SN_ACCUM(8, pt);
#ifdef SN_PROFILE
        pt = clock64();
#endif
        const void* nxt = (l + 1 < P.n_layers) ? P.layers[l].w1s : wfirst;   // the next bin restarts the stream here
        // GNN3d: mask . BN . ReLU . + previous_x
        auto pre2 = [&](int ot) { return lds_ld4(XR + 16 * ot + 4 * g); };
        auto epi2 = [&](int ot, f32x4 acc, f32x4 b2, f32x4 s1, f32x4 h1, f32x4 prev) {
          const f32x4 v = (acc + b2) * s1 + h1;
          in[ot] = DGL ? v : relu4(v) + prev;
        };
        if constexpr (LAG) wg_gemm_split_lag<NT, NT, BARKB>(ring, Lp.w2s, nxt, LIVE, sp, pre2, epi2);
        else wg_gemm_split<NT, NT, false>(ring, Lp.w2s, nxt, LIVE, sp, pre2, epi2);
        if (LIVE && !valid) {
#pragma unroll
          for (int kk = 0; kk < NT; ++kk) in[kk] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        SN_ACCUM(9, pt);
#ifdef SN_PROFILE
        pt = clock64();
#endif
        lds_barrier();  // everyone is done reading X before it is overwritten
        SN_ACCUM(10, pt);
      }
#ifdef SN_PROFILE
      pt = clock64();
#endif
      // -------------------------------------------------------------- phi(x) + phi(-x) -> out[node*K + slot, :], written once
      // The two waves of a row tile exchange their results through the LDS image (free now: every wave is past the last
      // layer's gathers — it went through that layer's GEMM barriers); the + wave stores the lower half of the channel tiles
      // of the sum, the - wave the upper half.
      if (LIVE) {
        // (only the channel tiles the partner wave adds and stores: the - wave takes the upper half from the + wave and the other way
        //  round — ds_write_b128 moves 79 B/clk per CU, a full 128 KB image is 1.6 k cycles of every bin)
        constexpr int H = (NT + 1) / 2;
#pragma unroll
        for (int kk = 0; kk < NT; ++kk)
          if (sg ? (kk < H) : (kk >= H)) lds_st4(XR + 16 * kk + 4 * g, in[kk]);
      }
      lds_barrier();
      if (LIVE && valid && !DGL) {
        float* orow = S.out + ((int64_t)dnode[lb * PHI_R + r] * S.K + dslot[lb * PHI_R + r]) * P.d;       // (re-read: not kept live across the layers)
        const float* XO = X2 + (1 - sg) * PHI_R * LD + r * LD;     // the same row in the other sign's image
        constexpr int H = (NT + 1) / 2;
        // (d % 4 == 0 is an entry-point requirement: whole float4 per lane; only the last 16-channel tile can be partial)
#pragma unroll
        for (int kk = 0; kk < NT; ++kk) {
          const bool mine = sg ? (kk >= H) : (kk < H);
          const int c = 16 * kk + 4 * g;
          if (mine && (kk + 1 < NT || c < P.d)) {
            const f32x4 o4 = lds_ld4(XO + c);
            const f32x4 v = sg ? o4 + in[kk] : in[kk] + o4;        // phi(x) + phi(-x), in that order on both waves
            *reinterpret_cast<float4*>(orow + c) = make_float4(v[0], v[1], v[2], v[3]);
          }
        }
      }
      if (LIVE && valid && DGL) {
        const int dout = P.reserved;                               // phi_out_dim: any width <= d (rows are not float4-aligned)
        float* orow = S.out + ((int64_t)dnode[lb * PHI_R + r] * S.K + dslot[lb * PHI_R + r]) * dout;
        const float* XO = X2 + (1 - sg) * PHI_R * LD + r * LD;
        constexpr int H = (NT + 1) / 2;
#pragma unroll
        for (int kk = 0; kk < NT; ++kk) {
          const bool mine = sg ? (kk >= H) : (kk < H);
          const int c = 16 * kk + 4 * g;
          if (mine && c < dout) {
            const f32x4 o4 = lds_ld4(XO + c);
            const f32x4 v = sg ? o4 + in[kk] : in[kk] + o4;
#pragma unroll
            for (int t = 0; t < 4; ++t)
              if (c + t < dout) orow[c + t] = v[t];
          }
        }
      }
      SN_ACCUM(14, pt);
    };
    if constexpr (LAG) {
      // an instantiation per sign: the barrier position is a compile-time property of the wave's stream
      if (sg) { if (wave_live) layers(std::true_type{}, std::integral_constant<int, 0>{}); else layers(std::false_type{}, std::integral_constant<int, 0>{}); }
      else { if (wave_live) layers(std::true_type{}, std::integral_constant<int, 1>{}); else layers(std::false_type{}, std::integral_constant<int, 1>{}); }
    } else {
      if (wave_live) layers(std::true_type{}, std::integral_constant<int, -1>{}); else layers(std::false_type{}, std::integral_constant<int, -1>{});
    }
#ifdef SN_TIMELINE
    if (tl_bin < 4) SN_TL(3 + tl_bin);
    ++tl_bin;
#endif
    SN_STAMP(11);
    }
  }
  { SN_PROF_ON(true); SN_STAMP(13); }
  ring.drain();
  SN_TL(7);
}

template <int NT, bool HID1, bool DGL = false>
static int launch_phi(const PhiStruct& S, const sn_phi_params& P, hipStream_t st) {
  constexpr int LD = 16 * NT + 4;
  const size_t lds = (size_t)WRing<NT, PHI_WAVES, phi_lagged(NT) ? LAG_RING : SPLIT_RING>::BYTES + (size_t)(2 * PHI_R * LD) * sizeof(float) +
                     (size_t)PHI_DESC_BYTES + (size_t)(4 * 16 * NT) * sizeof(float);
  static int cus = 0;  // idempotent one-time setup (same values whichever thread wins)
  if (cus == 0) {
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_phi_fused<NT, HID1, DGL>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return fail(SN_ERR_LAUNCH, "sn_phi_fused_f32: cannot raise the dynamic LDS limit to %zu", lds);
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    cus = n > 0 ? n : 256;
  }
  int64_t grid = S.max_bins < (int64_t)cus ? S.max_bins : (int64_t)cus;   // one 8-wave workgroup per CU (113 KB of LDS)
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL((k_phi_fused<NT, HID1, DGL>), dim3((unsigned)grid), dim3(PHI_WAVES * 64), lds, st, S, P);
  return SN_OK;
}

}  // namespace sn

using namespace sn;

#ifdef SN_TIMELINE
extern "C" int sn_timeline_read_phi(long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tl), sizeof(long long) * 1024 * 8); }
#endif
#ifdef SN_PROFILE
extern "C" int sn_prof_read_phi(long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_prof), sizeof(long long) * 64); }
#endif

extern "C" int sn_phi_fused_f32(const sn_phi_params* params, const float* eigen_vectors, const int32_t* graph_ptr,
                                const int64_t* evoff, const int32_t* rowptr, const int32_t* col,
                                const sn_plan_bins* bins, int kmax, int K, float* out, void* stream) {
  SN_REQUIRE(params && eigen_vectors && graph_ptr && evoff && rowptr && bins && out, "sn_phi_fused_f32: null pointer");
  SN_REQUIRE(bins->phi_bin_mem && bins->meta, "sn_phi_fused_f32: incomplete sn_plan_bins (phi_bin_mem, meta)");
  const sn_phi_params& P = *params;
  SN_REQUIRE(P.d > 0 && P.d <= 128 && (P.d & 3) == 0, "sn_phi_fused_f32: hidden width %d must be a multiple of 4 in (0, 128]", P.d);
  SN_REQUIRE(P.n_layers >= 1 && P.n_layers <= SN_PHI_MAX_LAYERS, "sn_phi_fused_f32: %d layers unsupported", P.n_layers);
  SN_REQUIRE(P.hid0 == 1 || P.hid0 == P.d, "sn_phi_fused_f32: first hidden width must be 1 or d");
  SN_REQUIRE(kmax >= 0, "sn_phi_fused_f32: kmax < 0 (full slots) belongs to sn_deepsigns_phi_f32");
  SN_REQUIRE(P.l0_w1 && P.l0_bn0_scale && P.l0_bn0_shift && P.l0_w2 && P.l0_eps, "sn_phi_fused_f32: layer-0 parameters missing");
  SN_REQUIRE(P.hid0 != 1 || (P.l0_bn_scale && P.l0_bn_shift), "sn_phi_fused_f32: layer-0 BatchNorm vectors missing");
  for (int l = 1; l < P.n_layers; ++l) {
    const sn_phi_layer& L = P.layers[l - 1];
    SN_REQUIRE(L.w1s && L.w2s && L.eps, "sn_phi_fused_f32: layer %d parameters missing", l);
  }
  SN_REQUIRE(K > 0 && bins->phi_max_bins >= 0, "sn_phi_fused_f32: bad K / max_bins");
  if (bins->phi_max_bins == 0) return SN_OK;
  PhiStruct S{eigen_vectors, graph_ptr, evoff, rowptr, col, bins->phi_bin_mem, bins->meta, bins->phi_max_bins, kmax, K, out, 0};
  hipStream_t st = (hipStream_t)stream;
  int rc = SN_OK;
  const int nt = (P.d + 15) / 16;
  if (P.hid0 == 1) {
    switch (nt) {
      case 1: rc = launch_phi<1, true>(S, P, st); break;
      case 2: rc = launch_phi<2, true>(S, P, st); break;
      case 3: rc = launch_phi<3, true>(S, P, st); break;
      case 4: rc = launch_phi<4, true>(S, P, st); break;
      case 5: rc = launch_phi<5, true>(S, P, st); break;
      case 6: rc = launch_phi<6, true>(S, P, st); break;
      case 7: rc = launch_phi<7, true>(S, P, st); break;
      default: rc = launch_phi<8, true>(S, P, st); break;
    }
  } else {
    switch (nt) {
      case 1: rc = launch_phi<1, false>(S, P, st); break;
      case 2: rc = launch_phi<2, false>(S, P, st); break;
      case 3: rc = launch_phi<3, false>(S, P, st); break;
      case 4: rc = launch_phi<4, false>(S, P, st); break;
      case 5: rc = launch_phi<5, false>(S, P, st); break;
      case 6: rc = launch_phi<6, false>(S, P, st); break;
      case 7: rc = launch_phi<7, false>(S, P, st); break;
      default: rc = launch_phi<8, false>(S, P, st); break;
    }
  }
  if (rc != SN_OK) return rc;
  SN_CHECK_LAUNCH("sn_phi_fused_f32");
  return SN_OK;
}

/* The DGL tree's GIN sign-invariant encoder, both signs and their sum, one launch (see sn_deepsigns_phi_f32 in signnet_hip.h). */
extern "C" int sn_deepsigns_phi_f32(const sn_phi_params* params, const float* x, int ldx, const int32_t* graph_ptr,
                                    const int32_t* rowptr, const int32_t* col, const sn_plan_bins* bins, int K, float* out,
                                    void* stream) {
  SN_REQUIRE(params && x && graph_ptr && rowptr && bins && out, "sn_deepsigns_phi_f32: null pointer");
  SN_REQUIRE(bins->phi_bin_mem && bins->meta, "sn_deepsigns_phi_f32: incomplete sn_plan_bins (phi_bin_mem, meta)");
  const sn_phi_params& P = *params;
  SN_REQUIRE(P.d >= 48 && P.d <= 112 && (P.d & 15) == 0, "sn_deepsigns_phi_f32: padded hidden width %d must be a multiple of 16 in [48, 112]", P.d);
  SN_REQUIRE(P.n_layers >= 1 && P.n_layers <= SN_PHI_MAX_LAYERS && P.hid0 == P.d, "sn_deepsigns_phi_f32: bad layer count / hid0");
  SN_REQUIRE(P.reserved >= 1 && P.reserved <= P.d, "sn_deepsigns_phi_f32: output width (params->reserved) %d not in [1, d]", P.reserved);
  SN_REQUIRE(P.l0_w1 && P.l0_bn0_scale && P.l0_bn0_shift && P.l0_w2 && P.l0_eps, "sn_deepsigns_phi_f32: layer-0 parameters missing");
  for (int l = 1; l < P.n_layers; ++l) {
    const sn_phi_layer& L = P.layers[l - 1];
    SN_REQUIRE(L.w1s && L.w2s && L.eps, "sn_deepsigns_phi_f32: layer %d parameters missing", l);
  }
  SN_REQUIRE(K > 0 && K <= 64 && ldx >= K && bins->phi_max_bins >= 0, "sn_deepsigns_phi_f32: bad K / ldx / max_bins");
  if (bins->phi_max_bins == 0) return SN_OK;
  PhiStruct S{x, graph_ptr, nullptr, rowptr, col, bins->phi_bin_mem, bins->meta, bins->phi_max_bins, K, K, out, ldx};
  hipStream_t st = (hipStream_t)stream;
  int rc = SN_OK;
  switch (P.d / 16) {
    case 3: rc = launch_phi<3, false, true>(S, P, st); break;
    case 4: rc = launch_phi<4, false, true>(S, P, st); break;
    case 5: rc = launch_phi<5, false, true>(S, P, st); break;
    case 6: rc = launch_phi<6, false, true>(S, P, st); break;
    default: rc = launch_phi<7, false, true>(S, P, st); break;
  }
  if (rc != SN_OK) return rc;
  SN_CHECK_LAUNCH("sn_deepsigns_phi_f32");
  return SN_OK;
}
