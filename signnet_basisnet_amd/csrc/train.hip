// train.hip — training-step stage kernels (BASELINE configs[3] is this workload).
//
// What the reference gets from torch.autograd over ATen for one "Linear -> BatchNorm1d(train) -> ReLU" link of a MaskedMLP / MLP
// (Alchemy/sign_net/model_utils/masked_layers.py:34-64, GINESignNetPyG/core/model_utils/elements.py:40-69) is ~10 kernels and ~10
// passes over the [rows, d] activations per direction.  Here a link is
//   forward : ONE pass   z = Linear(relu(bn_prev(x)))  with the producer's BatchNorm applied to the operand tile as it is loaded and
//             the batch moments of z taken from the accumulators (k_tlin_fwd), + a one-block finish of the statistics;
//   backward: ONE pass   dz = bn'(dy) formed on load -> dX = dz W (masked by the operand's ReLU, column sums for the producer's
//             BatchNorm backward taken from the accumulators) AND dW = dz^T x_hat, db (k_tlin_bwd), + a one-block finish and one
//             deterministic reduction of the per-workgroup dW partials.
// Rows come in G groups (the phi(+x) / phi(-x) passes share every weight but keep separate batch statistics: two calls of GNN3d,
// sign_net.py:113).  fp32-input MFMA throughout (exact products, fp32 accumulate); no atomics: gradients are bitwise reproducible.
#include "fused_common.hpp"
#include <atomic>

namespace sn {

namespace {

constexpr int TW = 8;                    // waves per workgroup (one workgroup per CU)
constexpr int TROWS = 64;                // rows per round of k_tlin_bwd

__device__ __forceinline__ float t16_sum(float v) {      // sum over the 16 lanes of a DPP row
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true));
  return v;
}
__device__ __forceinline__ f32x4 ldv(const float* __restrict__ p, int c0, int C) {     // 4 consecutive channels, any alignment
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 4; ++r) if (c0 + r < C) v[r] = p[c0 + r];
  return v;
}
__device__ __forceinline__ f32x4 ld4a(const float* __restrict__ p, int c0, int C) {    // 16-byte aligned rows, C % 4 == 0
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (c0 < C) { const float4 t = *reinterpret_cast<const float4*>(p + c0); v = f32x4{t.x, t.y, t.z, t.w}; }
  return v;
}
__device__ __forceinline__ f32x4 lds4(const float* p) {
  const float4 t = *reinterpret_cast<const float4*>(p);
  return f32x4{t.x, t.y, t.z, t.w};
}
__device__ __forceinline__ void st4a(float* __restrict__ p, int c0, int C, f32x4 v) {
  if (c0 < C) *reinterpret_cast<float4*>(p + c0) = make_float4(v[0], v[1], v[2], v[3]);
}
// ---- hand-off of per-workgroup partials to the LAST workgroup of the same launch (the "finish" of a link without a second launch) ----
// Producers publish with write-through (sc1, agent-scope) stores, drain them (vmcnt(0)), meet at the workgroup barrier and take an
// agent-scope ticket; the workgroup that draws the last ticket reads every partial with sc1 loads (served past this CU's L1) and reduces
// them in BLOCK order — the result does not depend on who was last.  No release fence: nothing but the partials has to be visible, and
// a buffer_wbl2 behind a pass that has just dirtied megabytes of L2 is what made the fenced form slower than a second launch.
constexpr int N_TICKETS = 4096;
__device__ unsigned g_tickets[N_TICKETS];        // zero at load; the last arriver of a launch resets its word

__device__ __forceinline__ void st_sc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_sc1(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// (write-through only when a last arriver will read the partial in this launch: an sc1 store drops the line from the XCD's L2, and the
//  finish KERNEL of the unfused path would fetch it from memory instead — k_tbn_finish 6.9 -> 10.6 us when every partial was sc1)
__device__ __forceinline__ void st_pub(float* p, float v, bool wt) { if (wt) st_sc1(p, v); else *p = v; }
__device__ __forceinline__ void st_pub(double* p, double v, bool wt) { if (wt) st_sc1(p, v); else *p = v; }
// 16 bytes with the sc1 policy (aux bit 4): through a buffer descriptor over the whole partial array (offsets < 2 GB)
__device__ __forceinline__ f32x4 ld4_sc1(__amdgpu_buffer_rsrc_t rs, int64_t float_off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(float_off * 4), 0, 16);
  return f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
}
// true in every thread of exactly one workgroup of the launch: the one that arrives last.  `flag`: one LDS word.
__device__ __forceinline__ bool last_arriver(unsigned* ticket, unsigned* flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // my write-through stores have left
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag = (t == gridDim.x - 1) ? 1u : 0u;
    if (t == gridDim.x - 1) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the launch that takes this word next
  }
  __syncthreads();
  return *flag != 0u;
}

__device__ __forceinline__ bool row_ok(int64_t r, int64_t R, const int32_t* __restrict__ nvalid, int K) {
  if (r >= R) return false;
  if (!nvalid) return true;
  if (R <= 0x7fffffffll) {         // (a 64-bit division is ~100 instructions: per row and thread it was most of the pointwise kernels)
    const uint32_t ru = (uint32_t)r, node = ru / (uint32_t)K;
    return (int)(ru - node * (uint32_t)K) < nvalid[node];
  }
  const int64_t node = r / K;
  return (int)(r - node * K) < nvalid[node];
}

// Weight image in the MFMA "A" fragment order of common.hpp, built from the RAW row-major parameter (no pack launch, any alignment):
//   wl[(ot*nk + kk)*64 + lane] = { M[16 ot + (lane&15)][16 kk + 4 (lane>>4) + t] }_t,  M = W (trans = 0: [n_o, n_k]) or W^T
template <bool TRANS>
__device__ __forceinline__ void stage_weight(float4* wl, const float* __restrict__ W, int ldw, int n_o, int n_k, int nto, int ntk) {
  const int total = nto * ntk * 64;
  // A 128 x 128 matrix is 32 image entries per thread.  One entry at a time the staging was a chain of 32 dependent L2 round trips — most
  // of a small launch.  Row-major parameter, 16-byte aligned with ldw % 4 == 0 (the usual case; FlatAdam's views are aligned when the
  // preceding parameters' sizes are multiples of 4): ONE float4 per entry, eight entries in flight; otherwise (and for the transposed
  // image, whose four values sit in four rows) scalar loads, eight entries = 32 loads in flight: ONE round trip for a 128 x 128 matrix.
  if (!TRANS && (reinterpret_cast<uintptr_t>(W) & 15) == 0 && (ldw & 3) == 0 && (n_k & 3) == 0) {
    for (int i0 = threadIdx.x; i0 < total; i0 += 8 * 64 * TW) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * 64 * TW;
        const int ln = i & 63, blk = i >> 6, kk = blk % ntk, ot = blk / ntk;
        const int o = 16 * ot + (ln & 15), k = 16 * kk + 4 * (ln >> 4);
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < total && o < n_o && k < n_k) v[u] = *reinterpret_cast<const float4*>(W + (int64_t)o * ldw + k);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * 64 * TW;
        if (i < total) wl[i] = v[u];
      }
    }
    return;
  }
  for (int i0 = threadIdx.x; i0 < total; i0 += 8 * 64 * TW) {
    float v[8][4];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * 64 * TW;
      const int ln = i & 63, blk = i >> 6, kk = blk % ntk, ot = blk / ntk;
      const int o = 16 * ot + (ln & 15), k = 16 * kk + 4 * (ln >> 4);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        v[u][t] = 0.f;
        if (i < total && o < n_o && k + t < n_k) v[u][t] = TRANS ? W[(int64_t)(k + t) * ldw + o] : W[(int64_t)o * ldw + k + t];
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * 64 * TW;
      if (i < total) wl[i] = make_float4(v[u][0], v[u][1], v[u][2], v[u][3]);
    }
  }
}

// The same image for the split-bf16 matrix path of fused_common.hpp (fp32 products from six bf16 partial products, exact three-way
// operand split): per (output tile, K block of 32) three 1 KB fragments [plane h, m, l][lane][8 bf16], lane (o = 16 ot + (lane&15),
// g = lane>>4) holding the k-slots 32 kb + 16 (s>>2) + 4 g + (s&3).  Built from the RAW row-major parameter (16-byte aligned rows).
template <int NTO, int NKB>
__device__ __forceinline__ void stage_weight_split(u32x4* wl, const float* __restrict__ W, int ldw) {
  constexpr int total = NTO * NKB * 64;
  for (int i0 = threadIdx.x; i0 < total; i0 += 4 * 64 * TW) {
    float4 lo[4], hi[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * 64 * TW;
      const int ln = i & 63, blk = i >> 6, kb = blk % NKB, ot = blk / NKB;
      const float* wr = W + (int64_t)(16 * ot + (ln & 15)) * ldw + 32 * kb + 4 * (ln >> 4);
      lo[u] = make_float4(0.f, 0.f, 0.f, 0.f); hi[u] = lo[u];
      if (i < total) { lo[u] = *reinterpret_cast<const float4*>(wr); hi[u] = *reinterpret_cast<const float4*>(wr + 16); }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * 64 * TW;
      if (i < total) {
        const Split8 sp = split8(f32x4{lo[u].x, lo[u].y, lo[u].z, lo[u].w}, f32x4{hi[u].x, hi[u].y, hi[u].z, hi[u].w});
        const int ln = i & 63, blk = i >> 6;
        u32x4* dst = wl + (blk * 3) * 64 + ln;
        dst[0] = sp.h; dst[64] = sp.m; dst[128] = sp.l;
      }
    }
  }
}

// ... and of the TRANSPOSE (the dX product of the backward link: M = W^T, M[o][k] = W[k][o]): the eight values of an entry sit in
// eight rows of W — scalar loads, two entries = 16 loads in flight per thread.
template <int NTO, int NKB>
__device__ __forceinline__ void stage_weight_split_t(u32x4* wl, const float* __restrict__ W, int ldw) {
  constexpr int total = NTO * NKB * 64;
  for (int i0 = threadIdx.x; i0 < total; i0 += 2 * 64 * TW) {
    float v[2][8];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = i0 + u * 64 * TW;
      const int ln = i & 63, blk = i >> 6, kb = blk % NKB, ot = blk / NKB;
      const float* wc = W + (16 * ot + (ln & 15)) + (int64_t)(32 * kb + 4 * (ln >> 4)) * ldw;
#pragma unroll
      for (int t = 0; t < 8; ++t) v[u][t] = (i < total) ? wc[(int64_t)((t >> 2) * 16 + (t & 3)) * ldw] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = i0 + u * 64 * TW;
      if (i < total) {
        const Split8 sp = split8(f32x4{v[u][0], v[u][1], v[u][2], v[u][3]}, f32x4{v[u][4], v[u][5], v[u][6], v[u][7]});
        const int ln = i & 63, blk = i >> 6;
        u32x4* dst = wl + (blk * 3) * 64 + ln;
        dst[0] = sp.h; dst[64] = sp.m; dst[128] = sp.l;
      }
    }
  }
}

// ============================================================================ forward link
// -DSN_PROFILE (scratch builds only: profiles/scripts/prof_train.sh): cycles per phase of workgroup 0 / wave 0 of the full-width split
// links, read back with sn_prof_read_train().  Slots: [0..9] backward link, [10..19] backward link with a dot_x operand, [20..29]
// forward link with statistics, [30..39] forward link without (each: 8 phase sums, the workgroup's rounds / tiles, R), [40..47] the
// cycle at which each wave of workgroup 0 left the forward tile loop.
#ifdef SN_PROFILE
static __device__ long long g_tprof[64];
#define TP_ON (SPLIT && blockIdx.x == 0 && threadIdx.x == 0)
#define TP_T0() long long tp0 = clock64()
#define TP_ACC(i) do { const long long tpn = clock64(); if (TP_ON) tpv[i] += tpn - tp0; tp0 = tpn; } while (0)
#else
#define TP_T0() do { } while (0)
#define TP_ACC(i) do { } while (0)
#endif
__device__ __forceinline__ void chan(float& na, float& ma, float& qa, float nb, float mb, float qb) {
  if (nb <= 0.f) return;
  const float n = na + nb, d = mb - ma;
  ma += d * (nb / n);
  qa += qb + d * d * (na * nb / n);
  na = n;
}
// (four columns that share the block counts: column by column the arithmetic of chan())
__device__ __forceinline__ void chan4(float& na, f32x4& ma, f32x4& qa, float nb, f32x4 mb, f32x4 qb) {
  float n = na;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    n = na;
    float m = ma[r], q = qa[r];
    chan(n, m, q, nb, mb[r], qb[r]);
    ma[r] = m; qa[r] = q;
  }
  na = n;
}

// The finish of a forward link's BatchNorm (what k_tbn_finish does in a launch of its own), run by the last-arriving workgroup of the
// link: NT threads = 32 column quads x NT/32 lanes; a column's <= nblk partials are merged (Chan) by 16 slices in block order and a
// fixed pairwise tree over the slices — the slicing, the order and every expression of k_tbn_finish (the two are compiled separately:
// where the compiler contracts an FMA in one and not the other the last bit differs; with `#pragma clang fp contract(off)` in both,
// all four launch structures of profiles/scripts/train_ab.sh gave identical gradients, parameters and running statistics).
// C <= 128, C % 4 == 0.  lds: (16*32 + 2*16*128) floats.
struct TFin {
  const float* gamma; const float* beta; float eps, momentum; float* rmean; float* rvar; float* st; float* cnt; unsigned* ticket;
};
constexpr int FIN_LDS_FLOATS = 16 * 32 + 2 * 16 * 128;
template <int NT>
__device__ __forceinline__ void tbn_finish_tail(const float* stat, int nblk, int G, int C, const TFin& f, float* lds) {
  static_assert(NT == 256 || NT == 512, "8 or 16 slice lanes");
  constexpr int SL = 512 / NT;               // slices per thread
  float* ln = lds;                            // [16][32]
  float* lm = lds + 16 * 32;                  // [16][128]
  float* lq = lm + 16 * 128;                  // [16][128]
  const int c4 = threadIdx.x & 31, rl0 = threadIdx.x >> 5;
  const bool col = 4 * c4 < C;
  const int per = (nblk + 15) / 16;
  const int64_t GC = (int64_t)G * C;
  for (int grp = 0; grp < G; ++grp) {
    const float* sg = stat + (int64_t)grp * (2 * (int64_t)nblk * C + nblk);
    const __amdgpu_buffer_rsrc_t rs = weight_rsrc(sg, 0x7fffffff);
#pragma unroll
    for (int sl = 0; sl < SL; ++sl) {
      const int rl = rl0 + sl * (NT / 32);
      const int b0 = rl * per, b1 = b0 + per < nblk ? b0 + per : nblk;
      float n = 0.f;
      f32x4 m = {0.f, 0.f, 0.f, 0.f}, q = m;
      if (col) {
        for (int b = b0; b < b1; b += 8) {        // eight partials in flight
          float nb[8];
          f32x4 mb[8], qb[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            nb[u] = 0.f; mb[u] = f32x4{0.f, 0.f, 0.f, 0.f}; qb[u] = mb[u];
            if (b + u < b1) {
              nb[u] = ld_sc1(sg + 2 * (int64_t)nblk * C + b + u);
              mb[u] = ld4_sc1(rs, (int64_t)(b + u) * C + 4 * c4);
              qb[u] = ld4_sc1(rs, ((int64_t)nblk + b + u) * C + 4 * c4);
            }
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) chan4(n, m, q, nb[u], mb[u], qb[u]);
        }
      }
      ln[rl * 32 + c4] = n;
      *reinterpret_cast<float4*>(lm + rl * 128 + 4 * c4) = make_float4(m[0], m[1], m[2], m[3]);
      *reinterpret_cast<float4*>(lq + rl * 128 + 4 * c4) = make_float4(q[0], q[1], q[2], q[3]);
    }
    __syncthreads();
    for (int step = 8; step >= 1; step >>= 1) {
      if (rl0 < step) {
        float n = ln[rl0 * 32 + c4];
        f32x4 m = lds4(lm + rl0 * 128 + 4 * c4), q = lds4(lq + rl0 * 128 + 4 * c4);
        chan4(n, m, q, ln[(rl0 + step) * 32 + c4], lds4(lm + (rl0 + step) * 128 + 4 * c4), lds4(lq + (rl0 + step) * 128 + 4 * c4));
        ln[rl0 * 32 + c4] = n;
        *reinterpret_cast<float4*>(lm + rl0 * 128 + 4 * c4) = make_float4(m[0], m[1], m[2], m[3]);
        *reinterpret_cast<float4*>(lq + rl0 * 128 + 4 * c4) = make_float4(q[0], q[1], q[2], q[3]);
      }
      __syncthreads();
    }
    if (rl0 == 0 && col) {
      const float n = ln[c4];
      const f32x4 m4 = lds4(lm + 4 * c4), q4 = lds4(lq + 4 * c4);
      float* s = f.st + (int64_t)grp * C;               // st[component][grp][C]
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = 4 * c4 + r;
        const float m = m4[r], q = q4[r];
        const float v = n > 0.f ? q / n : 0.f;
        const float rsd = 1.0f / sqrtf(v + f.eps);
        const float sc = (f.gamma ? f.gamma[c] : 1.f) * rsd;
        s[c] = m; s[GC + c] = v; s[2 * GC + c] = rsd; s[3 * GC + c] = sc; s[4 * GC + c] = (f.beta ? f.beta[c] : 0.f) - m * sc;
        if (c == 0) f.cnt[grp] = n;
        if (f.rmean) {
          const float unb = n > 1.f ? v * (n / (n - 1.f)) : v;
          f.rmean[c] = (1.f - f.momentum) * f.rmean[c] + f.momentum * m;
          f.rvar[c] = (1.f - f.momentum) * f.rvar[c] + f.momentum * unb;
        }
      }
    }
    __syncthreads();
  }
}

struct TLin {
  const float* x; int ldx; int64_t R; int G; int d_in, d_out;
  const float* W; int ldw; const float* bias;
  const int32_t* nvalid; int K;
  const float* in_scale; const float* in_shift; int in_relu; int out_relu;
  float* y; int ldy;
  float* stat;          // per group: [mean nblk*d_out | M2 nblk*d_out | count nblk]
  int nblk;             // workgroups per group
  TFin fin;             // fin.st != NULL: the statistics are finished by the launch's last workgroup
};

// FULL: d_in = 16 NTI and d_out = 16 NTO exactly (the 128-wide links) — the tile counts are compile-time constants, the per-tile guards
// fold away and the MFMA loops are straight-line code the compiler can pipeline the LDS reads of (with the guards every 16-column
// step was its own basic block: read, wait the LDS latency, 8 MFMAs — twice the MFMA time on a one-tile launch).
// SPLIT (round 4; FULL links with 16-byte aligned parameter rows): the products run on the bf16 matrix pipe from the exact three-way
// split of both operands (fused_common.hpp: 6 x v_mfma_f32_16x16x32_bf16 of 16 cycles per 32-deep K block and output tile against
// 8 x v_mfma_f32_16x16x4_f32 of 32) — the weight image holds the three bf16 planes (96 KB instead of 64), the operand tile is split in
// registers after the producer's BatchNorm + ReLU has been applied to it (176 VALU operations per tile for 5 k cycles of matrix pipe saved).
template <int NTI, int NTO, bool STATS, bool FULL, bool SPLIT = false>
__global__ __launch_bounds__(64 * TW, 1) void k_tlin_fwd(TLin a) {
  static_assert(!SPLIT || (FULL && NTI % 2 == 0), "the split path serves the full-width links");
  extern __shared__ __align__(16) unsigned char t_lds[];
  float4* wl = reinterpret_cast<float4*>(t_lds);
  constexpr int NKB = NTI / 2;
  constexpr size_t WIMG = SPLIT ? (size_t)NTO * NKB * 3 * 1024 : (size_t)NTI * NTO * 1024;     // bytes of the weight image
  float* icol = reinterpret_cast<float*>(t_lds + WIMG);     // [2][16*NTI] in_scale | in_shift of my group
  float* piv = icol + 2 * 16 * NTI;                                             // [TW][16*NTO] per-wave pivots of the moment sums
  constexpr int CI = 16 * NTI;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
  const int nti = FULL ? NTI : (a.d_in + 15) >> 4, nto = FULL ? NTO : (a.d_out + 15) >> 4;
  const int grp = blockIdx.x / a.nblk, blk = blockIdx.x - grp * a.nblk;
  const int64_t ntiles_all = (a.R + 15) >> 4;
  const int64_t t_lo = ntiles_all * blk / a.nblk, t_hi = ntiles_all * (blk + 1) / a.nblk;
  const float* xg = a.x + (int64_t)grp * a.R * a.ldx;
  float* yg = a.y + (int64_t)grp * a.R * a.ldy;
  const float* isc = a.in_scale ? a.in_scale + (int64_t)grp * a.d_in : nullptr;
  const float* ish = a.in_scale ? a.in_shift + (int64_t)grp * a.d_in : nullptr;
  // Batch moments of y (STATS): per lane the sums of (v - p) and (v - p)^2 of its rows and 4 columns per output tile, p = the column
  // means of the wave's first tile (kept in LDS) — two VALU operations per value instead of a cross-lane Chan update per tile; the
  // pivot keeps the final M2 = S2 - S1^2/n free of cancellation.  Reduced over the rows once, at the end.
  float rn = 0.f;
  bool have_piv = false;
  // (SPLIT: the sums are reduced over a tile's 16 rows at once (DPP) and kept by ONE lane of the row group per output tile — lane
  //  (lane & 15) == ot — i.e. in 8 registers instead of 64: with the operand's three bf16 planes and a prefetched tile the per-lane
  //  form does not fit 256 registers)
  constexpr int NS = (STATS && !SPLIT) ? NTO : 1;
  f32x4 s1[NS], s2[NS];
#pragma unroll
  for (int ot = 0; ot < NS; ++ot) { s1[ot] = f32x4{0.f, 0.f, 0.f, 0.f}; s2[ot] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  float* mypiv = piv + wave * 16 * NTO;
#ifdef SN_PROFILE
  long long tpv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  TP_T0();
  auto fetch = [&](int64_t tile, bool v, f32x4 (&buf)[NTI]) {
    const float* xr = xg + (tile * 16 + (lane & 15)) * a.ldx;
#pragma unroll
    for (int kk = 0; kk < NTI; ++kk) {
      buf[kk] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (kk < nti && v) buf[kk] = ld4a(xr, 16 * kk + 4 * g, a.d_in);
    }
  };
  f32x4 in[NTI], nx[NTI];
  bool valid = false, nvalid_next = false;
  int64_t tile = t_lo + wave;
  if (tile < t_hi) {
    valid = row_ok(tile * 16 + (lane & 15), a.R, a.nvalid, a.K);
    fetch(tile, valid, in);
  }
  // every global read of the prologue is in flight before the first wait: the tile above, the column constants, the weight image
  static_assert(CI <= 64 * TW, "one column constant per thread");
  float c_sc = 1.f, c_sh = 0.f;
  if (isc && (int)threadIdx.x < a.d_in) { c_sc = isc[threadIdx.x]; c_sh = ish[threadIdx.x]; }
  if constexpr (SPLIT) stage_weight_split<NTO, NKB>(reinterpret_cast<u32x4*>(t_lds), a.W, a.ldw);
  else stage_weight<false>(wl, a.W, a.ldw, a.d_out, a.d_in, nto, nti);
  if (threadIdx.x < CI) { icol[threadIdx.x] = c_sc; icol[CI + threadIdx.x] = c_sh; }
  __syncthreads();
  TP_ACC(0);
#ifdef SN_PROFILE
  const long long tp_loop0 = clock64();
#endif
  for (; tile < t_hi; tile += TW) {
#ifdef SN_PROFILE
    if (TP_ON) tpv[7] += 1;
#endif
    const int64_t row = tile * 16 + (lane & 15);
    const bool inr = row < a.R;
    const bool more = tile + TW < t_hi;
    if (more) {        // the next tile's rows are requested before this tile goes into the matrix pipe
      nvalid_next = row_ok((tile + TW) * 16 + (lane & 15), a.R, a.nvalid, a.K);
      if (!SPLIT) fetch(tile + TW, nvalid_next, nx);     // (SPLIT: into `in` itself, once it has been split — 32 registers less)
    }
    float* yr = yg + row * a.ldy;
    const unsigned long long vb = __ballot(valid);
    if (vb != 0ull) {
      if (isc) {      // the producer's train-mode BatchNorm (+ ReLU), applied to the operand tile
#pragma unroll
        for (int kk = 0; kk < NTI; ++kk) {
          if (kk < nti) {
            const f32x4 sc = lds4(icol + 16 * kk + 4 * g), sh = lds4(icol + CI + 16 * kk + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float v = in[kk][r] * sc[r] + sh[r];
              if (a.in_relu) v = fmaxf(v, 0.f);
              in[kk][r] = valid ? v : 0.f;
            }
          }
        }
      } else if (a.in_relu) {
#pragma unroll
        for (int kk = 0; kk < NTI; ++kk)
#pragma unroll
          for (int r = 0; r < 4; ++r) in[kk][r] = fmaxf(in[kk][r], 0.f);
      }
    }
    TP_ACC(1);
    const float nt = (float)__popcll(vb & 0xffffull);
    const bool first = STATS && vb != 0ull && !have_piv;      // wave-uniform
    auto epilogue = [&](int ot, f32x4 acc) {
      const int o0 = 16 * ot + 4 * g;
      f32x4 v = acc;
      if (!valid) {
        v = f32x4{0.f, 0.f, 0.f, 0.f};
      } else {
        if (a.bias) v += ldv(a.bias, o0, a.d_out);
        if (a.out_relu) v = f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
      }
      if (inr) st4a(yr, o0, a.d_out, v);
      if (STATS) {
        f32x4 pv;
        if (first) {
          const float inv = 1.0f / nt;
#pragma unroll
          for (int r = 0; r < 4; ++r) pv[r] = t16_sum(v[r]) * inv;      // invalid rows hold 0
          if ((lane & 15) == 0) *reinterpret_cast<float4*>(mypiv + o0) = make_float4(pv[0], pv[1], pv[2], pv[3]);
        } else {
          int po = o0;
          if (SPLIT) asm volatile("" : "+v"(po));       // (SPLIT: read here, not hoisted above the tile's MFMAs: 32 registers)
          pv = lds4(mypiv + po);
        }
        if constexpr (SPLIT) {
          const bool mine = (lane & 15) == ot;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float d = valid ? v[r] - pv[r] : 0.f;
            const float t1 = t16_sum(d), t2 = t16_sum(d * d);
            s1[0][r] += mine ? t1 : 0.f;
            s2[0][r] += mine ? t2 : 0.f;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float d = valid ? v[r] - pv[r] : 0.f;
            s1[ot][r] += d;
            s2[ot][r] += d * d;
          }
        }
      }
    };
    if (vb == 0ull) {
      if (inr) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        for (int ot = 0; ot < nto; ++ot) st4a(yr, 16 * ot + 4 * g, a.d_out, z);
      }
    } else if constexpr (SPLIT) {
      Split8 sp[NKB];
      split_rows<NTI>(in, sp);
      asm volatile("" :: "v"(sp[0].h), "v"(sp[NKB - 1].l));
      TP_ACC(2);
      if (more) fetch(tile + TW, nvalid_next, in);
      const u32x4* wb = reinterpret_cast<const u32x4*>(t_lds) + lane;
      auto rdw = [&](int ot, int kb, u32x4 (&f)[3]) {
        const u32x4* q = wb + ((ot * NKB + kb) * 3) * 64;
        f[0] = q[0]; f[1] = q[64]; f[2] = q[128];
      };
      u32x4 fa[3], fb[3];
      rdw(0, 0, fa);
      // (a rolled loop over the output tiles: unrolled, the eight epilogues' addresses, pivots and bias vectors were hoisted above the
      //  tile's MFMAs and the kernel spilled 200 bytes per lane)
#pragma unroll 1
      for (int ot = 0; ot < NTO; ++ot) {
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
        TP_ACC(3);
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
          if (kb + 1 < NKB) rdw(ot, kb + 1, fb);          // the next fragments are read while this K block's six MFMAs run
          else if (ot + 1 < NTO) rdw(ot + 1, 0, fb);
          __builtin_amdgcn_sched_barrier(0);
          a1 = mfma_bf(fa[2], sp[kb].h, a1);
          a0 = mfma_bf(fa[1], sp[kb].h, a0);
          a1 = mfma_bf(fa[0], sp[kb].l, a1);
          a0 = mfma_bf(fa[0], sp[kb].m, a0);
          a1 = mfma_bf(fa[1], sp[kb].m, a1);
          a0 = mfma_bf(fa[0], sp[kb].h, a0);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < 3; ++q) fa[q] = fb[q];
        }
        TP_ACC(5);
        epilogue(ot, a0 + a1);
      }
      if (STATS) { rn += nt; have_piv = true; }
    } else {
#pragma unroll
      for (int ot = 0; ot < NTO; ot += 2) {
        if (ot + 1 < nto) {
          f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
          const float4* w0 = wl + (ot * nti) * 64 + lane;
          const float4* w1 = w0 + nti * 64;
          float4 pn = w0[0], qn = w1[0];        // the next step's fragments are read while this step's 8 MFMAs run
#pragma unroll
          for (int kk = 0; kk < NTI; ++kk) {
            if (kk < nti) {
              const float4 p = pn, q = qn;
              if (kk + 1 < nti) { pn = w0[(kk + 1) * 64]; qn = w1[(kk + 1) * 64]; }
              __builtin_amdgcn_sched_barrier(0);      // (left to itself the scheduler sinks the reads to just before their use)
              acc0 = mfma16(p.x, in[kk][0], acc0);
              acc1 = mfma16(q.x, in[kk][0], acc1);
              acc0 = mfma16(p.y, in[kk][1], acc0);
              acc1 = mfma16(q.y, in[kk][1], acc1);
              acc0 = mfma16(p.z, in[kk][2], acc0);
              acc1 = mfma16(q.z, in[kk][2], acc1);
              acc0 = mfma16(p.w, in[kk][3], acc0);
              acc1 = mfma16(q.w, in[kk][3], acc1);
            }
          }
          epilogue(ot, acc0);
          epilogue(ot + 1, acc1);
        } else if (ot < nto) {
          f32x4 acc0 = {0.f, 0.f, 0.f, 0.f};
          const float4* w0 = wl + (ot * nti) * 64 + lane;
#pragma unroll
          for (int kk = 0; kk < NTI; ++kk) {
            if (kk < nti) {
              const float4 p = w0[kk * 64];
              acc0 = mfma16(p.x, in[kk][0], acc0);
              acc0 = mfma16(p.y, in[kk][1], acc0);
              acc0 = mfma16(p.z, in[kk][2], acc0);
              acc0 = mfma16(p.w, in[kk][3], acc0);
            }
          }
          epilogue(ot, acc0);
        }
      }
      if (STATS) { rn += nt; have_piv = true; }
    }
    TP_ACC(3);
    if (more) {
      if (!SPLIT) {
#pragma unroll
        for (int kk = 0; kk < NTI; ++kk) in[kk] = nx[kk];
      }
      valid = nvalid_next;
    }
  }
  if (STATS) {
    // per-wave (n, mean, M2) from the pivoted sums, then one partial per WORKGROUP: the waves meet in LDS, merged in wave order (Chan)
#ifdef SN_PROFILE
    if (SPLIT && blockIdx.x == 0 && lane == 0) g_tprof[40 + wave] = clock64() - tp_loop0;      // when each wave left the tile loop
#endif
    lds_barrier();          // (not __syncthreads(): the tile stores of y need not have landed for the waves to meet in LDS)
    TP_ACC(4);
    float* sm = reinterpret_cast<float*>(t_lds);          // [TW][mean 16*NTO | M2 16*NTO], counts behind  (the weight image is dead)
    float* sc = sm + TW * 2 * 16 * NTO;
#pragma unroll
    for (int ot = 0; ot < NTO; ++ot) {
      f32x4 pv = {0.f, 0.f, 0.f, 0.f};
      if (have_piv) pv = lds4(mypiv + 16 * ot + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // (SPLIT: lane (lane & 15) == ot of every row group already holds the tile's row-reduced sums: broadcast them to the group)
        const float t1 = SPLIT ? __shfl(s1[0][r], (lane & 48) | ot, 64) : t16_sum(s1[ot][r]);
        const float t2 = SPLIT ? __shfl(s2[0][r], (lane & 48) | ot, 64) : t16_sum(s2[ot][r]);
        if ((lane & 15) == 0) {
          const float inv = rn > 0.f ? 1.0f / rn : 0.f;
          sm[(wave * 2 + 0) * 16 * NTO + 16 * ot + 4 * g + r] = pv[r] + t1 * inv;
          sm[(wave * 2 + 1) * 16 * NTO + 16 * ot + 4 * g + r] = fmaxf(t2 - t1 * t1 * inv, 0.f);
        }
      }
    }
    if (lane == 0) sc[wave] = rn;
    TP_ACC(4);
    lds_barrier();
    float* stg = a.stat + (int64_t)grp * (2 * (int64_t)a.nblk * a.d_out + a.nblk);
    for (int c = threadIdx.x; c < a.d_out; c += 64 * TW) {
      float n = 0.f, m = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < TW; ++w) {
        const float nb = sc[w];
        if (nb > 0.f) {
          const float mb = sm[(w * 2 + 0) * 16 * NTO + c], qb = sm[(w * 2 + 1) * 16 * NTO + c];
          const float nn = n + nb, d = mb - m;
          m += d * (nb / nn);
          q += qb + d * d * (n * nb / nn);
          n = nn;
        }
      }
      st_pub(stg + (int64_t)blk * a.d_out + c, m, a.fin.st != nullptr);
      st_pub(stg + ((int64_t)a.nblk + blk) * a.d_out + c, q, a.fin.st != nullptr);
    }
    if (threadIdx.x == 0) {
      float n = 0.f;
#pragma unroll
      for (int w = 0; w < TW; ++w) n += sc[w];
      st_pub(stg + 2 * (int64_t)a.nblk * a.d_out + blk, n, a.fin.st != nullptr);
    }
    if (a.fin.st) {        // (wave-uniform: a launch argument)
      unsigned* flag = reinterpret_cast<unsigned*>(t_lds + 16 * 1024);          // behind sm / sc (8 KB + 32 B)
      if (last_arriver(a.fin.ticket, flag))
        tbn_finish_tail<64 * TW>(a.stat, a.nblk, a.G, a.d_out, a.fin, reinterpret_cast<float*>(t_lds + 20 * 1024));
    }
  }
  TP_ACC(4);
#ifdef SN_PROFILE
  if (TP_ON) {
    const int o = STATS ? 20 : 30;
    for (int i = 0; i < 8; ++i) g_tprof[o + i] = tpv[i];
    g_tprof[o + 8] = t_hi - t_lo;
    g_tprof[o + 9] = a.R;
  }
#endif
}

// The same link for FEW rows (a batch of small graphs: 2 950 nodes = 185 row tiles).  The persistent kernel above gives such a launch one
// tile per wave on a quarter of the SIMDs and a serial chain of 8 output tiles x (32 MFMAs + epilogue) behind a 64 KB weight staging:
// 13 us of kernel for 3 us of arithmetic.  Here a workgroup is ONE 16-row tile and a wave ONE 16-column output tile: every wave reads
// its 16 weight rows (8 KB, L2) and the tile's rows straight into registers (no LDS, no barrier), runs its 32 MFMAs in two accumulator
// chains and writes its 16 output columns and their per-tile moments (mean, M2, count: the finish kernel merges them, Chan).
// grid = G * ntiles (nblk = ntiles), 64 * TW threads; d_out <= 16 * TW.
template <int NTI, bool STATS>
__global__ __launch_bounds__(64 * TW) void k_tlin_fwd_tile(TLin a) {
  const int lane = threadIdx.x & 63, ot = threadIdx.x >> 6, g = lane >> 4, lr = lane & 15;
  const int nti = (a.d_in + 15) >> 4, nto = (a.d_out + 15) >> 4;
  const bool fin = STATS && a.fin.st != nullptr;
  if (ot >= nto && !fin) return;          // (with the finish in this launch every wave stays: the last workgroup needs all of its lanes)
  const bool act = ot < nto;
  const int grp = blockIdx.x / a.nblk, blk = blockIdx.x - grp * a.nblk;
  const int64_t row = (int64_t)blk * 16 + lr;
  const bool inr = row < a.R && act, valid = act && row_ok(row, a.R, a.nvalid, a.K);
  const float* xr = a.x + ((int64_t)grp * a.R + row) * a.ldx;
  const int o = 16 * ot + lr;
  const float* wr = a.W + (int64_t)o * a.ldw;
  const bool wvec = (reinterpret_cast<uintptr_t>(a.W) & 15) == 0 && (a.ldw & 3) == 0 && (a.d_in & 3) == 0;
  const float* isc = a.in_scale ? a.in_scale + (int64_t)grp * a.d_in : nullptr;
  const float* ish = a.in_scale ? a.in_shift + (int64_t)grp * a.d_in : nullptr;
  f32x4 in[NTI], wf[NTI], sc[NTI], sh[NTI];
#pragma unroll
  for (int kk = 0; kk < NTI; ++kk) {
    const int c0 = 16 * kk + 4 * g;
    in[kk] = f32x4{0.f, 0.f, 0.f, 0.f};
    wf[kk] = in[kk];
    if (kk < nti) {
      if (valid) in[kk] = ld4a(xr, c0, a.d_in);
      if (o < a.d_out) wf[kk] = wvec ? ld4a(wr, c0, a.d_in) : ldv(wr, c0, a.d_in);
      if (isc) { sc[kk] = ld4a(isc, c0, a.d_in); sh[kk] = ld4a(ish, c0, a.d_in); }
    }
  }
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
  for (int kk = 0; kk < NTI; ++kk) {
    if (kk < nti) {
      f32x4 v = in[kk];
      if (isc) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float t = v[r] * sc[kk][r] + sh[kk][r];
          if (a.in_relu) t = fmaxf(t, 0.f);
          v[r] = valid ? t : 0.f;
        }
      } else if (a.in_relu) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
      }
      if (kk & 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc1 = mfma16(wf[kk][r], v[r], acc1);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc0 = mfma16(wf[kk][r], v[r], acc0);
      }
    }
  }
  const int o0 = 16 * ot + 4 * g;
  f32x4 v = acc0 + acc1;
  if (!valid) {
    v = f32x4{0.f, 0.f, 0.f, 0.f};
  } else {
    if (a.bias) v += ldv(a.bias, o0, a.d_out);
    if (a.out_relu) v = f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
  }
  if (inr) st4a(a.y + ((int64_t)grp * a.R + row) * a.ldy, o0, a.d_out, v);
  if (STATS) {
    const float nt = (float)__popcll(__ballot(valid) & 0xffffull);
    const float inv = nt > 0.f ? 1.0f / nt : 0.f;
    float* stg = a.stat + (int64_t)grp * (2 * (int64_t)a.nblk * a.d_out + a.nblk);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float m = t16_sum(v[r]) * inv;                 // invalid rows hold 0
      const float d = valid ? v[r] - m : 0.f;
      const float q = t16_sum(d * d);
      if (lr == 0 && act && o0 + r < a.d_out) {
        st_pub(stg + (int64_t)blk * a.d_out + o0 + r, m, fin);
        st_pub(stg + ((int64_t)a.nblk + blk) * a.d_out + o0 + r, q, fin);
      }
    }
    if (threadIdx.x == 0) st_pub(stg + 2 * (int64_t)a.nblk * a.d_out + blk, nt, fin);
    if (fin) {
      __shared__ __align__(16) float fin_lds[FIN_LDS_FLOATS + 4];
      if (last_arriver(a.fin.ticket, reinterpret_cast<unsigned*>(fin_lds + FIN_LDS_FLOATS)))
        tbn_finish_tail<64 * TW>(a.stat, a.nblk, a.G, a.d_out, a.fin, fin_lds);
    }
  }
}

// Finish of the train-mode BatchNorm(s) of one forward link: merges the per-workgroup moments of every group (Chan), writes the
// state the consumers and the backward read — st[0..4][grp][C] = mean, var (biased), rstd, scale = gamma*rstd, shift = beta - mean*scale;
// cnt[grp] — and applies the running-statistics updates in group order (two sequential calls of the module in the reference).
// grid cdiv(C,16), 256 threads = 16 columns x 16 lanes.
__global__ __launch_bounds__(256) void k_tbn_finish(const float* __restrict__ stat, int nblk, int G, int C, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, float eps, float momentum, float* __restrict__ rmean,
                                                    float* __restrict__ rvar, float* __restrict__ st, float* __restrict__ cnt) {
  __shared__ float ln[16][17], lm[16][17], lq[16][17];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  const int per = (nblk + 15) / 16, b0 = rl * per, b1 = b0 + per < nblk ? b0 + per : nblk;
  // (round 6: the kernel is nothing but latency — 6.9-7.6 us in the trace, 20 launches a step.  The column's affine parameters and running
  //  statistics are requested up front instead of behind the merge tree, and a lane's <= 16 partials of a group in ONE round trip from
  //  clamped addresses — straight-line loads, the count of a slot past the slice reads as 0 — instead of 8 + 4 behind a predicate each.)
  const bool fin = rl == 0 && c < C;
  float g_c = 1.f, be_c = 0.f, rm_c = 0.f, rv_c = 0.f;
  if (fin) {
    if (gamma) g_c = gamma[c];
    if (beta) be_c = beta[c];
    if (rmean) { rm_c = rmean[c]; rv_c = rvar[c]; }
  }
  for (int grp = 0; grp < G; ++grp) {
    const float* sg = stat + (int64_t)grp * (2 * (int64_t)nblk * C + nblk);
    float n = 0.f, m = 0.f, q = 0.f;
    if (c < C && b0 < b1) {
      if (per <= 16) {
        float nb[16], mb[16], qb[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int b = b0 + u < b1 ? b0 + u : b1 - 1;
          nb[u] = sg[2 * (int64_t)nblk * C + b]; mb[u] = sg[(int64_t)b * C + c]; qb[u] = sg[((int64_t)nblk + b) * C + c];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) chan(n, m, q, b0 + u < b1 ? nb[u] : 0.f, mb[u], qb[u]);
      } else {
        for (int b = b0; b < b1; b += 8) {        // eight partials in flight (one at a time: a chain of L2 round trips per column)
          float nb[8], mb[8], qb[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            nb[u] = 0.f; mb[u] = 0.f; qb[u] = 0.f;
            if (b + u < b1) { nb[u] = sg[2 * (int64_t)nblk * C + b + u]; mb[u] = sg[(int64_t)(b + u) * C + c]; qb[u] = sg[((int64_t)nblk + b + u) * C + c]; }
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) chan(n, m, q, nb[u], mb[u], qb[u]);
        }
      }
    }
    ln[rl][cl] = n; lm[rl][cl] = m; lq[rl][cl] = q;
    __syncthreads();
    for (int step = 8; step >= 1; step >>= 1) {
      if (rl < step) {
        chan(n, m, q, ln[rl + step][cl], lm[rl + step][cl], lq[rl + step][cl]);
        ln[rl][cl] = n; lm[rl][cl] = m; lq[rl][cl] = q;
      }
      __syncthreads();
    }
    if (fin) {
      const float v = n > 0.f ? q / n : 0.f;
      const float rs = 1.0f / sqrtf(v + eps);
      const float sc = g_c * rs;
      float* s = st + (int64_t)grp * C;               // st[component][grp][C]: every component is one contiguous [G][C] block
      const int64_t GC = (int64_t)G * C;
      s[c] = m; s[GC + c] = v; s[2 * GC + c] = rs; s[3 * GC + c] = sc; s[4 * GC + c] = be_c - m * sc;
      if (c == 0) cnt[grp] = n;
      if (rmean) {                                    // group order: two sequential calls of the module
        const float unb = n > 1.f ? v * (n / (n - 1.f)) : v;
        rm_c = (1.f - momentum) * rm_c + momentum * m;
        rv_c = (1.f - momentum) * rv_c + momentum * unb;
      }
    }
    __syncthreads();
  }
  if (fin && rmean) { rmean[c] = rm_c; rvar[c] = rv_c; }
}

// ============================================================================ backward link
// dz (this Linear's output gradient) is formed on load:  g = dy * [ms*zo + mt > 0]  (the ReLU behind this Linear's BatchNorm; ms NULL:
// g = dy), dz = A*g - B - C*zo (the BatchNorm backward with the column constants of k_tbn_bwd_finish; A NULL: dz = g).
// x_hat (the Linear's operand) is x, or relu?(xs*x + xt) when the operand was the producer's BatchNorm applied on load (xs != NULL).
// Outputs: gx = (dz W) * [x_hat > 0 if xrelu] — the gradient at the producer's BatchNorm output, masked by its ReLU;
//          sums[grp][blk][0][c] = sum_rows gx, sums[..][1][c] = sum_rows gx * (x - xmu)  (xmu != NULL: for the producer's BatchNorm backward);
//          dwp[blk] = sum_rows dz^T x_hat  (+ db behind it): per-workgroup partials, all groups together (shared weights).
// The finish of a BatchNorm backward (k_tbn_bwd_finish / tbn_bwd_finish_block in a launch of their own) by the last-arriving workgroup
// of the kernel that produced the column-sum partials: 16 slices per column in block order, the fixed pairwise tree, the same expressions.
// NT threads = 32 column quads x NT/32 lanes, C <= 128, C % 4 == 0.  lds: 2*16*128 floats.
struct TBFin {
  const float* st; const float* cnt; const float* gamma; float* coef; float* dgamma; float* dbeta; int acc;    // coef != NULL: BatchNorm finish
  float* dot_out;                                                                                              // != NULL: eps finish
  unsigned* ticket;
};
constexpr int BFIN_LDS_FLOATS = 2 * 16 * 128;
template <int NT>
__device__ __forceinline__ void tbn_bwd_finish_tail(const float* sums, int nblk, int G, int C, const TBFin& f, float* lds) {
  static_assert(NT == 256 || NT == 512, "8 or 16 slice lanes");
  constexpr int SL = 512 / NT;
  float* l1 = lds;                  // [16][128]
  float* l2 = lds + 16 * 128;
  const int c4 = threadIdx.x & 31, rl0 = threadIdx.x >> 5;
  const bool col = 4 * c4 < C;
  const int per = (nblk + 15) / 16;
  const int64_t GC = (int64_t)G * C;
  f32x4 dg = {0.f, 0.f, 0.f, 0.f}, db = dg;
  for (int grp = 0; grp < G; ++grp) {
    const float* S = sums + (int64_t)grp * nblk * 2 * C;
    const __amdgpu_buffer_rsrc_t rs = weight_rsrc(S, 0x7fffffff);
#pragma unroll
    for (int sl = 0; sl < SL; ++sl) {
      const int rl = rl0 + sl * (NT / 32);
      const int b0 = rl * per, b1 = b0 + per < nblk ? b0 + per : nblk;
      f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1;
      if (col) {
        for (int b = b0; b < b1; b += 8) {        // eight partials in flight
          f32x4 v1[8], v2[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            v1[u] = f32x4{0.f, 0.f, 0.f, 0.f}; v2[u] = v1[u];
            if (b + u < b1) { v1[u] = ld4_sc1(rs, (int64_t)(b + u) * 2 * C + 4 * c4); v2[u] = ld4_sc1(rs, (int64_t)(b + u) * 2 * C + C + 4 * c4); }
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) { s1 += v1[u]; s2 += v2[u]; }
        }
      }
      *reinterpret_cast<float4*>(l1 + rl * 128 + 4 * c4) = make_float4(s1[0], s1[1], s1[2], s1[3]);
      *reinterpret_cast<float4*>(l2 + rl * 128 + 4 * c4) = make_float4(s2[0], s2[1], s2[2], s2[3]);
    }
    __syncthreads();
    for (int step = 8; step >= 1; step >>= 1) {
      if (rl0 < step) {
        const f32x4 s1 = lds4(l1 + rl0 * 128 + 4 * c4) + lds4(l1 + (rl0 + step) * 128 + 4 * c4);
        const f32x4 s2 = lds4(l2 + rl0 * 128 + 4 * c4) + lds4(l2 + (rl0 + step) * 128 + 4 * c4);
        *reinterpret_cast<float4*>(l1 + rl0 * 128 + 4 * c4) = make_float4(s1[0], s1[1], s1[2], s1[3]);
        *reinterpret_cast<float4*>(l2 + rl0 * 128 + 4 * c4) = make_float4(s2[0], s2[1], s2[2], s2[3]);
      }
      __syncthreads();
    }
    if (rl0 == 0 && col) {
      const f32x4 t1 = lds4(l1 + 4 * c4), t2 = lds4(l2 + 4 * c4);
      const float* s = f.st + (int64_t)grp * C;
      const float n = f.cnt[grp];
      float* o = f.coef + (int64_t)grp * C;              // coef[0..2][grp][C]
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = 4 * c4 + r;
        const float s1 = t1[r], s2 = t2[r];
        const float mu = s[c], rsd = s[2 * GC + c];
        const float A = (f.gamma ? f.gamma[c] : 1.f) * rsd;
        const float m1 = n > 0.f ? s1 / n : 0.f, m2 = n > 0.f ? rsd * s2 / n : 0.f;
        o[c] = A;
        o[GC + c] = A * (m1 - m2 * rsd * mu);
        o[2 * GC + c] = A * m2 * rsd;
        db[r] += s1;
        dg[r] += rsd * s2;
      }
    }
    __syncthreads();
  }
  if (rl0 == 0 && col) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = 4 * c4 + r;
      if (f.dgamma) f.dgamma[c] = (f.acc ? f.dgamma[c] : 0.f) + dg[r];
      if (f.dbeta) f.dbeta[c] = (f.acc ? f.dbeta[c] : 0.f) + db[r];
    }
  }
}
// ... and of the eps gradient (tdot_finish_block): threads 0..255 add the n float64 partials, out[0] += the sum.  red: 4 doubles in LDS.
__device__ __forceinline__ void tdot_finish_tail(const double* part, int n, float* out, double* red) {
  const int tid = threadIdx.x;
  double t = 0.0;
  if (tid < 256) {
    for (int i = tid; i < n; i += 256) t += ld_sc1(part + i);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
    if ((tid & 63) == 0) red[tid >> 6] = t;
  }
  __syncthreads();
  if (tid == 0) out[0] = out[0] + (float)((red[0] + red[1]) + (red[2] + red[3]));
}

// Round 6, the consumer-side form of the same finish: the backward link that READS the coefficients merges the column-sum partials of
// its own BatchNorm in its prologue — every workgroup for its own group, in parallel, under the weight staging — instead of a finish
// launch (k_tbn_bwd_finish / the finish blocks of k_tpost) in front of it: the finish step is removed, not moved.  Same slicing, order and
// expressions as tbn_bwd_finish_block.  coef a | b | c of the workgroup's group go to LDS (`ocol`: [3][CO]); workgroup 0 also walks the
// other groups and adds d gamma / d beta (summed over the groups in group order).  NT threads, C <= 128, C % 4 == 0; lds: 2*16*128 floats.
struct TBMerge {
  const float* sums; int nblk; const float* st; const float* cnt; const float* gamma; float* dgamma; float* dbeta; int acc;
};
template <int NT>
__device__ __forceinline__ void tbn_bwd_merge(const TBMerge& f, int G, int C, int mygrp, bool all_groups, float* lds, float* ocol, int CO) {
  static_assert(NT == 512, "16 slice lanes");
  float* l1 = lds;                  // [16][128]
  float* l2 = lds + 16 * 128;
  const int c4 = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const bool col = 4 * c4 < C;
  const int nblk = f.nblk;
  const int per = (nblk + 15) / 16;
  const int b0 = rl * per, b1 = b0 + per < nblk ? b0 + per : nblk;
  const int64_t GC = (int64_t)G * C;
  f32x4 dg = {0.f, 0.f, 0.f, 0.f}, db = dg;
  for (int grp = all_groups ? 0 : mygrp; grp < (all_groups ? G : mygrp + 1); ++grp) {
    const float* S = f.sums + (int64_t)grp * nblk * 2 * C;
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1;
    if (col) {
      for (int b = b0; b < b1; b += 8) {        // eight partials in flight
        f32x4 v1[8], v2[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          v1[u] = f32x4{0.f, 0.f, 0.f, 0.f}; v2[u] = v1[u];
          if (b + u < b1) { v1[u] = ld4a(S + (int64_t)(b + u) * 2 * C, 4 * c4, C); v2[u] = ld4a(S + (int64_t)(b + u) * 2 * C + C, 4 * c4, C); }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { s1 += v1[u]; s2 += v2[u]; }
      }
    }
    *reinterpret_cast<float4*>(l1 + rl * 128 + 4 * c4) = make_float4(s1[0], s1[1], s1[2], s1[3]);
    *reinterpret_cast<float4*>(l2 + rl * 128 + 4 * c4) = make_float4(s2[0], s2[1], s2[2], s2[3]);
    __syncthreads();
    for (int step = 8; step >= 1; step >>= 1) {
      if (rl < step) {
        const f32x4 t1 = lds4(l1 + rl * 128 + 4 * c4) + lds4(l1 + (rl + step) * 128 + 4 * c4);
        const f32x4 t2 = lds4(l2 + rl * 128 + 4 * c4) + lds4(l2 + (rl + step) * 128 + 4 * c4);
        *reinterpret_cast<float4*>(l1 + rl * 128 + 4 * c4) = make_float4(t1[0], t1[1], t1[2], t1[3]);
        *reinterpret_cast<float4*>(l2 + rl * 128 + 4 * c4) = make_float4(t2[0], t2[1], t2[2], t2[3]);
      }
      __syncthreads();
    }
    if (rl == 0 && col) {
      const f32x4 t1 = lds4(l1 + 4 * c4), t2 = lds4(l2 + 4 * c4);
      const float* s = f.st + (int64_t)grp * C;
      const float n = f.cnt[grp];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = 4 * c4 + r;
        const float mu = s[c], rsd = s[2 * GC + c];
        const float A = (f.gamma ? f.gamma[c] : 1.f) * rsd;
        const float m1 = n > 0.f ? t1[r] / n : 0.f, m2 = n > 0.f ? rsd * t2[r] / n : 0.f;
        if (grp == mygrp) {
          ocol[c] = A;
          ocol[CO + c] = A * (m1 - m2 * rsd * mu);
          ocol[2 * CO + c] = A * m2 * rsd;
        }
        db[r] += t1[r];
        dg[r] += rsd * t2[r];
      }
    } else if (rl == 0 && grp == mygrp) {        // padding columns of the last tile: the identity (their dz must stay finite: 0 * NaN)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = 4 * c4 + r;
        if (c < CO) { ocol[c] = 1.f; ocol[CO + c] = 0.f; ocol[2 * CO + c] = 0.f; }
      }
    }
    __syncthreads();
  }
  if (all_groups && rl == 0 && col) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = 4 * c4 + r;
      if (f.dgamma) f.dgamma[c] = (f.acc ? f.dgamma[c] : 0.f) + dg[r];
      if (f.dbeta) f.dbeta[c] = (f.acc ? f.dbeta[c] : 0.f) + db[r];
    }
  }
}

struct TBwd {
  int64_t R; int G; const int32_t* nvalid; int K; int d_in, d_out;
  const float* dy; int lddy; const float* zo; int ldzo;
  const float* cA; const float* cB; const float* cC; const float* ms; const float* mt;     // [G][d_out]
  const float* x; int ldx; const float* xs; const float* xt; int xrelu; const float* xmu;   // [G][d_in]
  const float* W; int ldw;
  float* gx; int ldgx; float* sums; float* dwp; int want_db;
  int gx_acc;                                     // gx += (several Linears share one operand: q, k, v of the attention)
  const float* dotx; int lddot; double* dotp;      // dotp[blockIdx] = sum over my rows of gx . dotx   (the GIN / GINE eps gradient)
  int nblk;             // workgroups per group
  TBFin fin;            // fin.ticket != NULL: the producer's BatchNorm-backward finish / the eps finish by the launch's last workgroup
  TBMerge mg;           // mg.sums != NULL: coef a | b | c are merged from these column-sum partials in the prologue (cA / cB / cC unused)
};

__host__ __device__ constexpr int stage_ld(int ntiles) { return ((16 * ntiles + 63) / 64) * 64 + 16; }   // row stride = 16 mod 64 banks

// RT = row tiles per round (a round = 16 RT rows; the 8 waves are RT row tiles x 8/RT column parts): 4 for the persistent launch, 1 for
// FEW rows (a batch of small graphs) — there a workgroup per CU gets one 64-row round, 64 MFMAs deep per phase and wave on a fifth of
// the chip; with 16-row rounds the same rows are 4x the workgroups, each 4x shallower (the weight image is re-staged per workgroup
// from L2 either way, and the dW partials — one 64 KB image per workgroup — grow 4x: 12 MB at 2 950 rows, read once by the reduce).
// SPLIT (round 4, the full-width links on 32-row rounds): the dX product on the bf16 matrix pipe from the exact three-way split of dz
// and W^T (fused_common.hpp) — a wave's 16 dz rows are split once per round (176 VALU operations) and serve its output tiles at
// 4 x 6 MFMAs of 16 cycles per tile instead of 32 of 32.  The W^T image holds three bf16 planes (96 KB), which is why the rounds are
// 32 rows: 64-row staging images no longer fit beside it.  The dW product (K = rows: its operands would have to be split per
// MFMA step, more VALU work than the matrix pipe saves) stays on the fp32-input MFMA.
template <int NTI, int NTO, bool FULL, int RT, bool SPLIT = false>
__global__ __launch_bounds__(64 * TW, 1) void k_tlin_bwd(TBwd a) {
  static_assert(!SPLIT || (FULL && NTO % 2 == 0), "the split path serves the full-width links");
  constexpr int LDO = stage_ld(NTO), LDI = stage_ld(NTI);
  constexpr int TR = 16 * RT, NP = TW / RT;
  constexpr int NKBO = NTO / 2;
  constexpr size_t WIMG = SPLIT ? (size_t)NTI * NKBO * 3 * 1024 : (size_t)NTI * NTO * 1024;     // bytes of the W^T image
  extern __shared__ __align__(16) unsigned char t_lds[];
  const int nti = FULL ? NTI : (a.d_in + 15) >> 4, nto = FULL ? NTO : (a.d_out + 15) >> 4;
  float4* wl = reinterpret_cast<float4*>(t_lds);                          // W^T image: [ot2 < nti][kk < nto][64] float4
  float* dzs = reinterpret_cast<float*>(t_lds + WIMG);                    // [TR][LDO]
  float* xsg = dzs + TR * LDO;                                          // [TR][LDI]  raw x (x_hat is re-formed at each use)
  float* red = xsg + TR * LDI;                                          // [3 sums][4 row tiles][16*NTI] running column sums of gx
  float* xcol = red + 3 * 4 * 16 * NTI;                                    // [3][16*NTI] x_scale | x_shift | x_mean of my group
  float* ocol = xcol + 3 * 16 * NTI;                                       // [5][16*NTO] coef a | b | c | mask scale | mask shift
  constexpr int CI = 16 * NTI, CO = 16 * NTO;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, lr = lane & 15;
  const int rt = wave % RT, part = wave / RT;
  const int grp = blockIdx.x / a.nblk, blk = blockIdx.x - grp * a.nblk;
  const int64_t nrounds = (a.R + TR - 1) / TR;
  const int64_t r_lo = nrounds * blk / a.nblk, r_hi = nrounds * (blk + 1) / a.nblk;
  const int64_t goff = (int64_t)grp * a.R;
  const float* cA = a.cA ? a.cA + (int64_t)grp * a.d_out : nullptr;
  const float* cB = a.cA ? a.cB + (int64_t)grp * a.d_out : nullptr;
  const float* cC = a.cA ? a.cC + (int64_t)grp * a.d_out : nullptr;
  const float* ms = a.ms ? a.ms + (int64_t)grp * a.d_out : nullptr;
  const float* mt = a.ms ? a.mt + (int64_t)grp * a.d_out : nullptr;
  const float* xs = a.xs ? a.xs + (int64_t)grp * a.d_in : nullptr;
  const float* xt = a.xs ? a.xt + (int64_t)grp * a.d_in : nullptr;
  const float* xmu = a.xmu ? a.xmu + (int64_t)grp * a.d_in : nullptr;
  const bool want_dx = a.gx != nullptr;
#ifdef SN_PROFILE
  long long tpv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  TP_T0();
  // dW accumulators of this wave: output tile `wave` (16 dz columns) x every operand tile
  f32x4 dw[NTI];
#pragma unroll
  for (int it = 0; it < NTI; ++it) dw[it] = f32x4{0.f, 0.f, 0.f, 0.f};
  float dbacc = 0.f;
  double dots = 0.0;
  // column sums of gx over my rows (a wave's 16 rows x its column tiles), all rounds
  f32x4 cs1[(NTI + TW / RT - 1) / (TW / RT)], cs2[(NTI + TW / RT - 1) / (TW / RT)];
#pragma unroll
  for (int j = 0; j < (NTI + TW / RT - 1) / (TW / RT); ++j) { cs1[j] = f32x4{0.f, 0.f, 0.f, 0.f}; cs2[j] = cs1[j]; }
  // The raw rows of round r+1 are requested (into registers) right after round r's tiles are published, so the HBM latency runs
  // under the round's 256 MFMAs per wave; a wave loads the column tiles kk = part, part + NP, ... of its 16 rows.
  constexpr int HO = (NTO + NP - 1) / NP, HI = (NTI + NP - 1) / NP;
  f32x4 pdy[HO], pz[HO], px[HI];
  // the rows of dot_x / of the gradient being accumulated into, fetched with the round's other rows (consumed in phase 2: copied in
  // front of the next request).  Only where a wave has two column tiles: with four the copies spill, and those read them in place.
  constexpr bool PREQ = HI <= 2;
  f32x4 pq[PREQ ? HI : 1], pg[PREQ ? HI : 1];
  bool pvalid = false;
  auto request = [&](int64_t round) {
    const int64_t row = round * TR + 16 * rt + lr;
    pvalid = row_ok(row, a.R, a.nvalid, a.K);
    const float* dyr = a.dy + (goff + row) * a.lddy;
    const float* zr = a.zo ? a.zo + (goff + row) * a.ldzo : nullptr;
    const float* xr = a.x + (goff + row) * a.ldx;
#pragma unroll
    for (int j = 0; j < HO; ++j) {
      const int c0 = 16 * (NP * j + part) + 4 * g;
      pdy[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      pz[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (pvalid && NP * j + part < nto) {
        pdy[j] = ld4a(dyr, c0, a.d_out);
        if (zr) pz[j] = ld4a(zr, c0, a.d_out);
      }
    }
#pragma unroll
    for (int j = 0; j < HI; ++j) {
      px[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (pvalid && NP * j + part < nti) px[j] = ld4a(xr, 16 * (NP * j + part) + 4 * g, a.d_in);
    }
    if constexpr (PREQ) {
      if (a.dotx) {
#pragma unroll
        for (int j = 0; j < HI; ++j) {
          pq[j] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (pvalid && NP * j + part < nti) pq[j] = ld4a(a.dotx + (goff + row) * a.lddot, 16 * (NP * j + part) + 4 * g, a.d_in);
        }
      }
      if (a.gx_acc && a.gx) {
#pragma unroll
        for (int j = 0; j < HI; ++j) {
          pg[j] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (row < a.R && NP * j + part < nti) pg[j] = ld4a(a.gx + (goff + row) * a.ldgx, 16 * (NP * j + part) + 4 * g, a.d_in);
        }
      }
    }
  };
  // Prologue: every global read is in flight before the first wait — the first round's rows, the column constants (one per thread,
  // held in registers), the weight image — then the LDS writes.  (One after the other these were five L2 round trips, ~6 us a launch.)
  request(r_lo);
  static_assert(CI <= 64 * TW && CO <= 64 * TW, "one column constant per thread");
  const bool merged = a.mg.sums != nullptr;
  if (merged)        // (the staging images are idle until the first round: the merge tree uses their LDS; its barriers come before any other LDS write)
    tbn_bwd_merge<64 * TW>(a.mg, a.G, a.d_out, grp, blockIdx.x == 0, dzs, ocol, CO);
  float c_x[3] = {1.f, 0.f, 0.f}, c_o[5] = {1.f, 0.f, 0.f, 0.f, 1.f};
  {
    const int i = threadIdx.x;
    if (xs && i < a.d_in) { c_x[0] = xs[i]; c_x[1] = xt[i]; }
    if (xmu && i < a.d_in) c_x[2] = xmu[i];
    if (cA && i < a.d_out) { c_o[0] = cA[i]; c_o[1] = cB[i]; c_o[2] = cC[i]; }
    if (ms && i < a.d_out) { c_o[3] = ms[i]; c_o[4] = mt[i]; }
  }
  if (want_dx) {
    if constexpr (SPLIT) stage_weight_split_t<NTI, NKBO>(reinterpret_cast<u32x4*>(t_lds), a.W, a.ldw);
    else stage_weight<true>(wl, a.W, a.ldw, a.d_in, a.d_out, nti, nto);
  }
  if (threadIdx.x < CI) {
#pragma unroll
    for (int j = 0; j < 3; ++j) xcol[j * CI + threadIdx.x] = c_x[j];
  }
  if (threadIdx.x < CO) {
    if (!merged) {                                                                          // (merged: a | b | c are in place)
#pragma unroll
      for (int j = 0; j < 3; ++j) ocol[j * CO + threadIdx.x] = c_o[j];
    }
#pragma unroll
    for (int j = 3; j < 5; ++j) ocol[j * CO + threadIdx.x] = c_o[j];
  }
  // running column sums of gx: one LDS slot per (row tile, column), owned by one lane of one wave (kept out of the register file:
  // with them the kernel spilled)
  for (int i = threadIdx.x; i < 3 * 4 * 16 * NTI; i += 64 * TW) red[i] = 0.f;
  __syncthreads();          // the column constants (and the weight image) are published before phase 1 reads them
  TP_ACC(0);
  for (int64_t round = r_lo; round < r_hi; ++round) {
    const int64_t row = round * TR + 16 * rt + lr;     // my row within the group (phases 1, 2)
    const bool valid = pvalid;
    // ---------------------------------------------------------------- phase 1: dz and x of the round -> LDS
    {
      float* dst = dzs + (16 * rt + lr) * LDO;
#pragma unroll
      for (int j = 0; j < HO; ++j) {
        const int kk = NP * j + part;
        if (kk < nto) {
          const int c0 = 16 * kk + 4 * g;
          f32x4 v = pdy[j];
          if (valid && a.zo) {
            const f32x4 z = pz[j];
            if (ms) {
              const f32x4 m0 = lds4(ocol + 3 * CO + c0), m1 = lds4(ocol + 4 * CO + c0);
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = (z[r] * m0[r] + m1[r] > 0.f) ? v[r] : 0.f;
            }
            if (cA || merged) {
              const f32x4 A = lds4(ocol + c0), B = lds4(ocol + CO + c0), Cc = lds4(ocol + 2 * CO + c0);
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = (A[r] * v[r] - B[r]) - Cc[r] * z[r];
            }
          }
          *reinterpret_cast<float4*>(dst + c0) = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
      float* dsx = xsg + (16 * rt + lr) * LDI;
#pragma unroll
      for (int j = 0; j < HI; ++j) {
        const int kk = NP * j + part;
        if (kk < nti) *reinterpret_cast<float4*>(dsx + 16 * kk + 4 * g) = make_float4(px[j][0], px[j][1], px[j][2], px[j][3]);
      }
    }
    TP_ACC(1);
    __syncthreads();
    TP_ACC(2);
    f32x4 cq[PREQ ? HI : 1], cg[PREQ ? HI : 1];
    if constexpr (PREQ) {
#pragma unroll
      for (int j = 0; j < HI; ++j) { cq[j] = pq[j]; cg[j] = pg[j]; }
    }
    if (round + 1 < r_hi) request(round + 1);
    // ---------------------------------------------------------------- phase 2: gx = (dz W) * mask, column sums
    if (want_dx) {
      f32x4 fr[NTO];
      const float* src = dzs + (16 * rt + lr) * LDO;
#pragma unroll
      for (int kk = 0; kk < NTO; ++kk) {
        fr[kk] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (kk < nto) { const float4 t = *reinterpret_cast<const float4*>(src + 16 * kk + 4 * g); fr[kk] = f32x4{t.x, t.y, t.z, t.w}; }
      }
      float* gr = a.gx + (goff + row) * a.ldgx;
      const float* xrow = xsg + (16 * rt + lr) * LDI;
      Split8 dsp[SPLIT ? NKBO : 1];
      if constexpr (SPLIT) {
        split_rows<NTO>(fr, dsp);
        asm volatile("" :: "v"(dsp[0].h), "v"(dsp[NKBO - 1].l));
      }
#pragma unroll
      for (int j = 0; j < HI; ++j) {
        const int ot = NP * j + part;
        if (ot < nti) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          if constexpr (SPLIT) {
            const u32x4* wb = reinterpret_cast<const u32x4*>(t_lds) + lane + (ot * NKBO * 3) * 64;
            f32x4 a1 = {0.f, 0.f, 0.f, 0.f};
            u32x4 fa[3] = {wb[0], wb[64], wb[128]}, fb[3];
#pragma unroll
            for (int kb = 0; kb < NKBO; ++kb) {
              if (kb + 1 < NKBO) { const u32x4* q = wb + ((kb + 1) * 3) * 64; fb[0] = q[0]; fb[1] = q[64]; fb[2] = q[128]; }
              __builtin_amdgcn_sched_barrier(0);
              a1 = mfma_bf(fa[2], dsp[kb].h, a1);
              acc = mfma_bf(fa[1], dsp[kb].h, acc);
              a1 = mfma_bf(fa[0], dsp[kb].l, a1);
              acc = mfma_bf(fa[0], dsp[kb].m, acc);
              a1 = mfma_bf(fa[1], dsp[kb].m, a1);
              acc = mfma_bf(fa[0], dsp[kb].h, acc);
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int q = 0; q < 3; ++q) fa[q] = fb[q];
            }
            acc = acc + a1;
          } else {
          const float4* w0 = wl + (ot * nto) * 64 + lane;
          float4 pn = w0[0];
#pragma unroll
          for (int kk = 0; kk < NTO; ++kk) {
            if (kk < nto) {
              const float4 p = pn;
              if (kk + 1 < nto) pn = w0[(kk + 1) * 64];
              __builtin_amdgcn_sched_barrier(0);
              acc = mfma16(p.x, fr[kk][0], acc);
              acc = mfma16(p.y, fr[kk][1], acc);
              acc = mfma16(p.z, fr[kk][2], acc);
              acc = mfma16(p.w, fr[kk][3], acc);
            }
          }
          }
          const int c0 = 16 * ot + 4 * g;
          const float4 xq = *reinterpret_cast<const float4*>(xrow + c0);
          const f32x4 xv = {xq.x, xq.y, xq.z, xq.w};
          f32x4 v = acc;
          if (!valid) v = f32x4{0.f, 0.f, 0.f, 0.f};
          else if (a.xrelu) {
            if (xs) {
              const f32x4 sc = lds4(xcol + c0), sh = lds4(xcol + CI + c0);
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = (xv[r] * sc[r] + sh[r] > 0.f) ? v[r] : 0.f;
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = xv[r] > 0.f ? v[r] : 0.f;
            }
          }
          if (a.gx_acc && row < a.R) {
            if constexpr (PREQ) v += cg[PREQ ? j : 0]; else v += ld4a(gr, c0, a.d_in);
          }
          if (row < a.R) st4a(gr, c0, a.d_in, v);
          if (a.dotx) {      // a cancelling scalar sum over all rows and columns: float64, one accumulator per lane over all rounds
            f32x4 q = {0.f, 0.f, 0.f, 0.f};
            if constexpr (PREQ) q = cq[PREQ ? j : 0];
            else { if (valid) q = ld4a(a.dotx + (goff + row) * a.lddot, c0, a.d_in); }
#pragma unroll
            for (int r = 0; r < 4; ++r) dots += (double)(v[r] * q[r]);
          }
          if (xmu) {        // my rows' share of the column sums, in registers over all rounds; across the 16 rows once, after the loop
            const f32x4 mu = lds4(xcol + 2 * CI + c0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              cs1[j][r] += v[r];
              cs2[j][r] += v[r] * (xv[r] - mu[r]);
            }
          }
        }
      }
    }
    TP_ACC(3);
    // ---------------------------------------------------------------- phase 3: dW[tile `wave`] += dz^T x_hat over the round's rows
    if (wave < nto && a.dwp) {
      float xsc[NTI], xsh[NTI];       // x_hat's column constants of my dW lanes, out of the step loop (16 LDS reads per step otherwise)
#pragma unroll
      for (int it = 0; it < NTI; ++it) { xsc[it] = xcol[16 * it + lr]; xsh[it] = xcol[CI + 16 * it + lr]; }
#pragma unroll 1     // (unrolled by 2 the LDS reads of both steps are hoisted and the kernel spills; a hand-made prefetch of the next
                     //  step's operands measured slower)
      for (int q = 0; q < TR / 4; ++q) {
        const int rl = 4 * q + g;
        const float av = dzs[rl * LDO + 16 * wave + lr];
        dbacc += av;
        const float* xr = xsg + rl * LDI + lr;
        float bv[NTI];
#pragma unroll
        for (int it = 0; it < NTI; ++it) bv[it] = (it < nti) ? xr[16 * it] : 0.f;
        __builtin_amdgcn_sched_barrier(0);      // all of the step's LDS reads in flight, ONE wait (the scheduler paired each read with
                                                // its two MFMAs: four LDS round trips per step, ~2x the MFMA time)
#pragma unroll
        for (int it = 0; it < NTI; ++it) {
          if (it < nti) {
            float b = bv[it];
            if (xs) { b = b * xsc[it] + xsh[it]; if (a.xrelu) b = fmaxf(b, 0.f); }
            dw[it] = mfma16(av, b, dw[it]);
          }
        }
      }
    }
    TP_ACC(4);
    __syncthreads();
    TP_ACC(5);
  }
  if (want_dx && xmu) {       // the waves' column sums meet in LDS: slot (row tile, column) has ONE owner
#pragma unroll
    for (int j = 0; j < HI; ++j) {
      const int ot = NP * j + part;
      if (ot < nti) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float a1 = t16_sum(cs1[j][r]), a2 = t16_sum(cs2[j][r]);
          if (lr == 0) {
            red[(0 * 4 + rt) * 16 * NTI + 16 * ot + 4 * g + r] = a1;
            red[(1 * 4 + rt) * 16 * NTI + 16 * ot + 4 * g + r] = a2;
          }
        }
      }
    }
    __syncthreads();
  }
  // ------------------------------------------------------------------ partial results of the workgroup
  if (a.dwp && wave < nto) {
    float* P = a.dwp + (int64_t)blockIdx.x * ((int64_t)a.d_out * a.d_in + (a.want_db ? a.d_out : 0));
#pragma unroll
    for (int it = 0; it < NTI; ++it) {
      if (it < nti) {
        const int i = 16 * it + lr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = 16 * wave + 4 * g + r;
          if (o < a.d_out && i < a.d_in) P[(int64_t)o * a.d_in + i] = dw[it][r];
        }
      }
    }
    if (a.want_db) {
      float s = dbacc;
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      const int o = 16 * wave + lr;
      if (g == 0 && o < a.d_out) P[(int64_t)a.d_out * a.d_in + o] = s;
    }
  }
  if (want_dx && a.dotx) {
    double t = dots;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
    __shared__ double wsum[TW];
    if (lane == 0) wsum[wave] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
      double tt = 0.0;
#pragma unroll
      for (int w = 0; w < TW; ++w) tt += wsum[w];
      st_pub(a.dotp + blockIdx.x, tt, a.fin.ticket != nullptr);
    }
  }
  if (want_dx && xmu) {
    // the four row tiles' sums -> one partial per workgroup   (published by the barrier above)
    float* S = a.sums + (int64_t)blockIdx.x * 2 * a.d_in;
    for (int i = threadIdx.x; i < 2 * a.d_in; i += 64 * TW) {
      const int w = i / a.d_in, c = i - w * a.d_in;
      const float* p = red + (w * 4) * 16 * NTI + c;
      st_pub(S + i, (p[0] + p[16 * NTI]) + (p[2 * 16 * NTI] + p[3 * 16 * NTI]), a.fin.ticket != nullptr);
    }
  }
  if (a.fin.ticket) {          // (a launch argument: wave-uniform)
    __shared__ unsigned s_last;
    if (last_arriver(a.fin.ticket, &s_last)) {      // (its barriers also retire every read of `red` and of the weight image)
      if (a.fin.coef) tbn_bwd_finish_tail<64 * TW>(a.sums, a.nblk, a.G, a.d_in, a.fin, reinterpret_cast<float*>(t_lds));
      if (a.fin.dot_out) tdot_finish_tail(a.dotp, a.nblk * a.G, a.fin.dot_out, reinterpret_cast<double*>(t_lds));
    }
  }
  TP_ACC(6);
#ifdef SN_PROFILE
  if (TP_ON) {
    const int o = a.dotx ? 10 : 0;
    for (int i = 0; i < 8; ++i) g_tprof[o + i] = tpv[i];
    g_tprof[o + 8] = r_hi - r_lo;
    g_tprof[o + 9] = a.R;
  }
#endif
}


// Column sums for a BatchNorm backward whose upstream gradient comes from somewhere else than k_tlin_bwd (the last BatchNorm of a
// stack: its output feeds an aggregation / a residual): sums[grp][blk][0][c] = sum g, [1][c] = sum g * (z - mu), g = dy * [ms*z + mt > 0].
__global__ __launch_bounds__(256) void k_tbn_bwd_sums(const float* __restrict__ dy, int lddy, const float* __restrict__ z, int ldz, int64_t R,
                                                      int G, int C, const int32_t* __restrict__ nvalid, int K, const float* __restrict__ st,
                                                      int relu, int nblk, float* sums, TBFin fin) {
  const int grp = blockIdx.x / nblk, blk = blockIdx.x - grp * nblk;
  const int C4 = C >> 2, cg = threadIdx.x % C4, rg = threadIdx.x / C4, nrg = 256 / C4;      // C % 4 == 0, C <= 1024
  const int64_t r_lo = R * blk / nblk, r_hi = R * (blk + 1) / nblk;
  const float* s = st + (int64_t)grp * C;
  const int64_t GC = (int64_t)G * C;
  __shared__ __align__(16) float sh[BFIN_LDS_FLOATS + 4];       // the row groups' sums; then the finish of the last workgroup + its flag
  float (*red)[256][4] = reinterpret_cast<float (*)[256][4]>(sh);
  f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = a1;
  if (rg < nrg) {
    const f32x4 mu = ld4a(s, 4 * cg, C), sc = ld4a(s + 3 * GC, 4 * cg, C), sh = ld4a(s + 4 * GC, 4 * cg, C);
    for (int64_t r = r_lo + rg; r < r_hi; r += nrg) {
      if (!row_ok(r, R, nvalid, K)) continue;
      const f32x4 d = ld4a(dy + ((int64_t)grp * R + r) * lddy, 4 * cg, C), zz = ld4a(z + ((int64_t)grp * R + r) * ldz, 4 * cg, C);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float gv = (!relu || zz[t] * sc[t] + sh[t] > 0.f) ? d[t] : 0.f;
        a1[t] += gv;
        a2[t] += gv * (zz[t] - mu[t]);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) { red[0][threadIdx.x][t] = a1[t]; red[1][threadIdx.x][t] = a2[t]; }
  __syncthreads();
  float* S = sums + (int64_t)blockIdx.x * 2 * C;
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    const int w = i / C, c = i - w * C;
    float acc = 0.f;
    for (int q = 0; q < nrg; ++q) acc += red[w][q * C4 + (c >> 2)][c & 3];
    if (fin.ticket) st_sc1(S + i, acc); else S[i] = acc;
  }
  if (fin.ticket) {
    if (last_arriver(fin.ticket, reinterpret_cast<unsigned*>(sh + BFIN_LDS_FLOATS))) tbn_bwd_finish_tail<256>(sums, nblk, G, C, fin, sh);
  }
}

// Finish of a BatchNorm backward: from the column-sum partials of every group, the column constants of
//   dz = A*g - B - C*z,   A = gamma*rstd,  B = A*(m1 - m2*rstd*mu),  C = A*m2*rstd,   m1 = sum g / n,  m2 = rstd * sum g (z-mu) / n
// (coef[0..2][grp][C]) and the affine gradients d beta += sum g, d gamma += rstd * sum g (z - mu) summed over the groups (the
// reference applies ONE module to both sign passes).  One thread per column; partials added in block order (deterministic).
__device__ __forceinline__ void tbn_bwd_finish_block(int bid, int tid, const float* __restrict__ sums, int nblk, int G, int C, const float* __restrict__ st,
                                                        const float* __restrict__ cnt, const float* __restrict__ gamma, float* __restrict__ coef,
                                                        float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate) {
  // 16 columns x 16 lanes per block: a lane adds its slice of the block partials in order, then a fixed pairwise tree
  // (round 6: the group's statistics / count / gamma requested with the partials instead of behind the tree, and a lane's <= 16
  //  partials in one round trip from clamped addresses: see k_tbn_finish)
  __shared__ float l1[16][17], l2[16][17];
  const int cl = tid & 15, rl = tid >> 4;
  const int c = bid * 16 + cl;
  const int per = (nblk + 15) / 16, b0 = rl * per, b1 = b0 + per < nblk ? b0 + per : nblk;
  const int64_t GC = (int64_t)G * C;
  const bool fin = rl == 0 && c < C;
  const float g_c = (fin && gamma) ? gamma[c] : 1.f;
  float dg = 0.f, db = 0.f;
  for (int grp = 0; grp < G; ++grp) {
    const float* S = sums + (int64_t)grp * nblk * 2 * C;
    float mu = 0.f, rs = 0.f, n = 0.f;
    if (fin) { const float* s = st + (int64_t)grp * C; mu = s[c]; rs = s[2 * GC + c]; n = cnt[grp]; }
    float s1 = 0.f, s2 = 0.f;
    if (c < C && b0 < b1) {
      if (per <= 16) {
        float v1[16], v2[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int b = b0 + u < b1 ? b0 + u : b1 - 1;
          v1[u] = S[(int64_t)b * 2 * C + c]; v2[u] = S[(int64_t)b * 2 * C + C + c];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) { s1 += b0 + u < b1 ? v1[u] : 0.f; s2 += b0 + u < b1 ? v2[u] : 0.f; }
      } else {
        for (int b = b0; b < b1; b += 8) {        // eight partials in flight
          float v1[8], v2[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            v1[u] = 0.f; v2[u] = 0.f;
            if (b + u < b1) { v1[u] = S[(int64_t)(b + u) * 2 * C + c]; v2[u] = S[(int64_t)(b + u) * 2 * C + C + c]; }
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) { s1 += v1[u]; s2 += v2[u]; }
        }
      }
    }
    l1[rl][cl] = s1; l2[rl][cl] = s2;
    __syncthreads();
    for (int step = 8; step >= 1; step >>= 1) {
      if (rl < step) {
        s1 += l1[rl + step][cl]; s2 += l2[rl + step][cl];
        l1[rl][cl] = s1; l2[rl][cl] = s2;
      }
      __syncthreads();
    }
    if (fin) {
      const float A = g_c * rs;
      const float m1 = n > 0.f ? s1 / n : 0.f, m2 = n > 0.f ? rs * s2 / n : 0.f;
      float* o = coef + (int64_t)grp * C;              // coef[0..2][grp][C]
      o[c] = A;
      o[GC + c] = A * (m1 - m2 * rs * mu);
      o[2 * GC + c] = A * m2 * rs;
      db += s1;
      dg += rs * s2;
    }
    __syncthreads();
  }
  if (fin) {
    if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + dg;
    if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + db;
  }
}

// out[i] (+)= sum_b part[b*stride + i]   (partials added in block order)
__global__ __launch_bounds__(256) void k_tbn_bwd_finish(const float* __restrict__ sums, int nblk, int G, int C, const float* __restrict__ st,
                                                        const float* __restrict__ cnt, const float* __restrict__ gamma, float* __restrict__ coef,
                                                        float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate) {
  tbn_bwd_finish_block(blockIdx.x, threadIdx.x, sums, nblk, G, C, st, cnt, gamma, coef, dgamma, dbeta, accumulate);
}
__device__ __forceinline__ void tsum_parts_block(int bid, int tid, const float* __restrict__ part, int nparts, int64_t stride, int64_t n,
                                                     float* __restrict__ out, int accumulate) {
  // 64 outputs per block, 16 lanes per output: lane q adds the partials b = q, q + 16, ... in order, eight loads in flight (with 4 lanes
  // and 4 loads in flight the 256 partials of a large link were 16 dependent round trips: 6.6 us x 26 launches a step), then the 16
  // lanes in order
  __shared__ float red[16][64];
  const int c = tid & 63, q = tid >> 6;
  const int64_t i = (int64_t)bid * 64 + c;
  float acc = 0.f;
  if (i < n) {
    const float* src = part + i;
    int b = q;
    for (; b + 7 * 16 < nparts; b += 8 * 16) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(b + 16 * u) * stride];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; b < nparts; b += 16) acc += src[(int64_t)b * stride];
  }
  red[q][c] = acc;
  __syncthreads();
  if (q == 0 && i < n) {
    float t = red[0][c];
#pragma unroll
    for (int w = 1; w < 16; ++w) t += red[w][c];
    out[i] = (accumulate ? out[i] : 0.f) + t;
  }
}
__global__ __launch_bounds__(1024) void k_tsum_parts(const float* __restrict__ part, int nparts, int64_t stride, int64_t n,
                                                     float* __restrict__ out, int accumulate) {
  tsum_parts_block(blockIdx.x, threadIdx.x, part, nparts, stride, n, out, accumulate);
}

// ... for a table of jobs in one launch (the dW / db partials of every backward link of a step): block -> job by the prefix of block counts
struct TJobs {
  const float* part[SN_TRAIN_MAX_REDUCE_JOBS]; float* out[SN_TRAIN_MAX_REDUCE_JOBS]; int64_t stride[SN_TRAIN_MAX_REDUCE_JOBS];
  int64_t n[SN_TRAIN_MAX_REDUCE_JOBS]; int nparts[SN_TRAIN_MAX_REDUCE_JOBS]; int first[SN_TRAIN_MAX_REDUCE_JOBS + 1]; int acc_mask_lo, acc_mask_hi;
  int njobs;
};
__global__ __launch_bounds__(1024) void k_treduce_jobs(TJobs J) {
  int j = 0;
  while (j + 1 < J.njobs && (int)blockIdx.x >= J.first[j + 1]) ++j;          // (uniform: <= 64 scalar compares)
  const int acc = j < 32 ? (J.acc_mask_lo >> j) & 1 : (J.acc_mask_hi >> (j - 32)) & 1;
  tsum_parts_block((int)blockIdx.x - J.first[j], threadIdx.x, J.part[j], J.nparts[j], J.stride[j], J.n[j], J.out[j], acc);
}

// y = [relu](z * scale[g] + shift[g]) [+ res] on valid rows, 0 elsewhere (the last BatchNorm of a stack, whose output is materialised:
// it feeds an aggregation and the next layer's residual).  One float4 per thread.
__global__ __launch_bounds__(256) void k_tbn_apply(const float* __restrict__ z, int ldz, int64_t R, int G, int C, const int32_t* __restrict__ nvalid,
                                                   int K, const float* __restrict__ st, int relu, const float* __restrict__ res, int ldr,
                                                   float* __restrict__ y, int ldy) {
  const int C4 = C >> 2;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)G * R * C4) return;
  const int64_t row = idx / C4;
  const int c0 = 4 * (int)(idx - row * C4);
  const int grp = (int)(row / R);
  const int64_t r = row - (int64_t)grp * R;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (row_ok(r, R, nvalid, K)) {
    const float* s = st + (int64_t)grp * C;
    const int64_t GC = (int64_t)G * C;
    const f32x4 zz = ld4a(z + row * ldz, c0, C), sc = ld4a(s + 3 * GC, c0, C), sh = ld4a(s + 4 * GC, c0, C);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float a = zz[t] * sc[t] + sh[t];
      if (relu) a = fmaxf(a, 0.f);
      v[t] = a;
    }
    if (res) v += ld4a(res + row * ldr, c0, C);
  }
  st4a(y + row * ldy, c0, C, v);
}

// =====================================================================================================================
// The 1 -> 1 -> d MaskedMLP in front of a GNN3d whose input is a scalar per (node, slot) row — GINESignNetPyG's first phi layer and
// its eigen_encoder2 (core/sign_net.py:20-22, 90-112: Linear(1,1).BN.ReLU.Linear(1,d)[.BN.ReLU]) — in closed form.
//   z_a = w1 a;  h = relu(bn_a(z_a));  z_b[c] = w2[c] h + b2[c];  y[c] = relu(bn_b(z_b)[c])
// Every column of z_b is an affine image of the SAME scalar h, so its batch statistics follow from the moments of h
// (mean_c = w2_c m_h + b2_c, var_c = w2_c^2 v_h) and y[row][c] = relu(p_c (h - m_h) + beta_c), p_c = gamma_c w2_c / sqrt(w2_c^2 v_h + eps):
// the forward is ONE write of [G, M, d] (no [M, d] intermediate is ever read), the backward ONE read of dy: with g = dy [y > 0],
//   d gamma_c = (w2_c/s_c) S2_c,  d beta_c = S1_c,  S1_c = sum g,  S2_c = sum g (h - m_h),  s_c = sqrt(w2_c^2 v_h + eps)
//   d w2_c = (gamma_c/s_c) (S2_c - m2_c (w2_c/s_c) n v_h),  m1_c = S1_c/n,  m2_c = (w2_c/s_c) S2_c / n
//   d h[row] = sum_c p_c g[row][c] - C1 - (h - m_h) C2,  C1 = sum_c p_c m1_c,  C2 = sum_c p_c (w2_c/s_c) m2_c
// and the scalar chain (ReLU, one-channel BatchNorm, w1) runs on [G, M] scalars.  Rows: G groups of M (group 1 optionally -a: phi(-x)).
struct SMlp {
  const float* a; int64_t M; int G; int negate1; const int32_t* nvalid; int K; int d;
  const float* w1; const float* ga; const float* ba; float eps_a;
  const float* w2; const float* b2; const float* gb; const float* bb; float eps_b; int relu_b;
  double* sst;          // [G][8] (float64: the scalar chain is kept exact to fp64 rounding — its biases are multiplied by the row count):
                        //   n, mu_a, rstd_a, scale_a (= gamma_a rstd_a), shift_a, m_h, v_h, -
  float* cst;           // [G][2][d]: p_c, s_c
};
// The scalar passes have global, sequential reductions (mean before variance, a's statistics before h's).  They run as a few short
// multi-workgroup launches over row blocks: a block computes the exact local (n, mean, M2) of its rows — two passes over rows it has
// just read — and the next kernel's blocks each merge the <= 256 block partials themselves (Chan), so no launch is a serial chain over
// the rows (a single-workgroup version took 115 us forward and 275 us backward for 47 200 rows).
constexpr int SM_T = 256;
constexpr int SM_MAXB = 256;

__device__ __forceinline__ float block_sum(float v, float* red) {      // all threads get the sum; red: SM_T/64 floats
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = 0.f;
  for (int w = 0; w < SM_T / 64; ++w) s += red[w];
  return s;
}
__device__ __forceinline__ double block_sum_d(double v, double* red) {      // the scalar chain's sums: float64 (a handful per block)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
  for (int w = 0; w < SM_T / 64; ++w) s += red[w];
  return s;
}
__device__ __forceinline__ int smlp_blocks(int64_t M) {
  const int64_t want = (M + SM_T - 1) / SM_T;
  return (int)(want < 1 ? 1 : (want < SM_MAXB ? want : SM_MAXB));
}
// local moments of f(row) over the valid rows [lo, hi) of this block: out = (n, mean, M2)
// raw sums (n, sum f, sum f^2) of f(row) over the valid rows [lo, hi) of this block, in float64: with 53 bits the raw second moment
// loses nothing to cancellation, and merging block partials is a plain sum (a block-parallel one: <= SM_T partials)
template <typename F>
__device__ __forceinline__ void local_sums(const SMlp& p, int64_t lo, int64_t hi, double* red, F&& f, double& n, double& s1, double& s2) {
  double cn = 0.0, cs = 0.0, cq = 0.0;
  for (int64_t r = lo + threadIdx.x; r < hi; r += SM_T)
    if (row_ok(r, p.M, p.nvalid, p.K)) { const double v = f(r); cn += 1.0; cs += v; cq += v * v; }
  n = block_sum_d(cn, red);
  s1 = block_sum_d(cs, red);
  s2 = block_sum_d(cq, red);
}
// (n, mean, biased variance) from the nb <= SM_T block partials
__device__ __forceinline__ void merge_sums(const double* __restrict__ part, int nb, double* red, double& n, double& mean, double& var) {
  const bool in = (int)threadIdx.x < nb;
  n = block_sum_d(in ? part[3 * threadIdx.x] : 0.0, red);
  const double s1 = block_sum_d(in ? part[3 * threadIdx.x + 1] : 0.0, red);
  const double s2 = block_sum_d(in ? part[3 * threadIdx.x + 2] : 0.0, red);
  mean = n > 0.0 ? s1 / n : 0.0;
  var = n > 0.0 ? fmax(s2 / n - mean * mean, 0.0) : 0.0;
}

__global__ __launch_bounds__(SM_T) void k_smlp_s1(SMlp p, double* __restrict__ p1) {
  __shared__ double red[SM_T / 64];
  const int nb = gridDim.x;
  const int64_t lo = p.M * blockIdx.x / nb, hi = p.M * (blockIdx.x + 1) / nb;
  double n, s1, s2;
  local_sums(p, lo, hi, red, [&](int64_t r) { return (double)p.a[r]; }, n, s1, s2);
  if (threadIdx.x == 0) { p1[3 * blockIdx.x] = n; p1[3 * blockIdx.x + 1] = s1; p1[3 * blockIdx.x + 2] = s2; }
}
// statistics of z_a = (+-w1) a from the moments of a; then the block moments of h per group
__global__ __launch_bounds__(SM_T) void k_smlp_s2(SMlp p, const double* __restrict__ p1, double* __restrict__ p2, float mom_a, float* rm_a,
                                                  float* rv_a) {
  __shared__ double red[SM_T / 64];
  const int nb = gridDim.x;
  const int64_t lo = p.M * blockIdx.x / nb, hi = p.M * (blockIdx.x + 1) / nb;
  double n, ma, va;
  merge_sums(p1, nb, red, n, ma, va);
  const double w1 = p.w1[0];
  const double var = w1 * w1 * va;
  const double rstd = 1.0 / sqrt(var + (double)p.eps_a);
  const double gam = (p.ga ? p.ga[0] : 1.f), bet = (p.ba ? p.ba[0] : 0.f);
  for (int grp = 0; grp < p.G; ++grp) {
    const double sw = (grp == 1 && p.negate1) ? -w1 : w1;
    const double mu = sw * ma, sc = gam * rstd, sh = bet - mu * sc;
    double hn, hm, hq;
    local_sums(p, lo, hi, red, [&](int64_t r) { return fmax(sw * (double)p.a[r] * sc + sh, 0.0); }, hn, hm, hq);
    if (threadIdx.x == 0) {
      double* o = p2 + ((int64_t)grp * nb + blockIdx.x) * 3;
      o[0] = hn; o[1] = hm; o[2] = hq;
      if (blockIdx.x == 0) {
        double* s = p.sst + grp * 8;
        s[0] = n; s[1] = mu; s[2] = rstd; s[3] = sc; s[4] = sh;
        if (rm_a) {       // group order: two sequential calls of the module
          const double unb = n > 1.0 ? var * (n / (n - 1.0)) : var;
          rm_a[0] = (float)((1.0 - mom_a) * rm_a[0] + mom_a * mu);
          rv_a[0] = (float)((1.0 - mom_a) * rv_a[0] + mom_a * unb);
        }
      }
    }
  }
}
// one workgroup, a thread per column: m_h, v_h of every group, the column constants, the running statistics of bn_b
__global__ __launch_bounds__(SM_T) void k_smlp_s3(SMlp p, const double* __restrict__ p2, int nb, float mom_b, float* rm_b, float* rv_b) {
  __shared__ double red[SM_T / 64];
  for (int grp = 0; grp < p.G; ++grp) {
    double n, mh, vh;
    merge_sums(p2 + (int64_t)grp * nb * 3, nb, red, n, mh, vh);
    if (threadIdx.x == 0) { p.sst[grp * 8 + 5] = mh; p.sst[grp * 8 + 6] = vh; p.sst[grp * 8 + 7] = 0.0; }
    for (int c = threadIdx.x; c < p.d; c += SM_T) {
      const double w2 = p.w2[c];
      const double vc = w2 * w2 * vh;
      const double s_c = sqrt(vc + (double)p.eps_b);
      p.cst[(grp * 2 + 0) * p.d + c] = (float)((p.gb ? p.gb[c] : 1.f) * w2 / s_c);
      p.cst[(grp * 2 + 1) * p.d + c] = (float)s_c;
      if (rm_b) {
        const double unb = n > 1.0 ? vc * (n / (n - 1.0)) : vc;
        rm_b[c] = (float)((1.0 - mom_b) * rm_b[c] + mom_b * (w2 * mh + (p.b2 ? p.b2[c] : 0.f)));
        rv_b[c] = (float)((1.0 - mom_b) * rv_b[c] + mom_b * unb);
      }
    }
  }
}

// y[grp][row][c] = [relu](p_c (h - m_h) + beta_c), 0 on invalid rows; one float4 per thread
__global__ __launch_bounds__(256) void k_smlp_apply(SMlp p, float* __restrict__ y) {
  const int C4 = p.d >> 2;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)p.G * p.M * C4) return;
  int64_t row;
  int grp;
  if ((int64_t)p.G * p.M * C4 <= 0x7fffffffll) {          // 32-bit index arithmetic when it fits
    const uint32_t rw = (uint32_t)idx / (uint32_t)C4;
    row = rw;
    grp = (int)(rw / (uint32_t)p.M);
  } else {
    row = idx / C4;
    grp = (int)(row / p.M);
  }
  const int c0 = 4 * (int)(idx - row * C4);
  const int64_t r = row - (int64_t)grp * p.M;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (row_ok(r, p.M, p.nvalid, p.K)) {
    const double* s = p.sst + grp * 8;
    const float sg = (grp == 1 && p.negate1) ? -1.f : 1.f;
    const float hc = fmaxf(sg * p.w1[0] * p.a[r] * (float)s[3] + (float)s[4], 0.f) - (float)s[5];
    const f32x4 pc = ld4a(p.cst + (grp * 2) * p.d, c0, p.d), be = p.bb ? ldv(p.bb, c0, p.d) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float pre = pc[t] * hc + be[t];
      v[t] = p.relu_b ? fmaxf(pre, 0.f) : pre;
    }
  }
  st4a(y + row * p.d, c0, p.d, v);
}

// backward pass over dy: column partials part[grp][blk][0][c] = sum g, [1][c] = sum g (h - m_h), row sums t[grp][row] = sum_c p_c g
__global__ __launch_bounds__(256) void k_smlp_bwd_pass(SMlp p, const float* __restrict__ dy, int nblk, float* __restrict__ part,
                                                       float* __restrict__ trow) {
  const int grp = blockIdx.x / nblk, blk = blockIdx.x - grp * nblk;
  const int C4 = p.d >> 2, cg = threadIdx.x % C4, rg = threadIdx.x / C4, nrg = 256 / C4;
  const int64_t r_lo = p.M * blk / nblk, r_hi = p.M * (blk + 1) / nblk;
  __shared__ float red[2][256][4];
  __shared__ float rsum[256];
  const float s3 = (float)p.sst[grp * 8 + 3], s4 = (float)p.sst[grp * 8 + 4], s5 = (float)p.sst[grp * 8 + 5];
  const float sg = (grp == 1 && p.negate1) ? -1.f : 1.f, w1 = p.w1[0];
  const bool pow2 = (C4 & (C4 - 1)) == 0 && C4 <= 64;
  f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = a1;
  f32x4 pc = {0.f, 0.f, 0.f, 0.f}, be = pc;
  if (rg < nrg) {
    pc = ld4a(p.cst + (grp * 2) * p.d, 4 * cg, p.d);
    if (p.bb) be = ldv(p.bb, 4 * cg, p.d);
  }
  for (int64_t r0 = r_lo; r0 < r_hi; r0 += nrg) {       // uniform trip count: the row sums go through LDS
    const int64_t r = r0 + rg;
    float tsum = 0.f;
    if (rg < nrg && r < r_hi && row_ok(r, p.M, p.nvalid, p.K)) {
      const float hc = fmaxf(sg * w1 * p.a[r] * s3 + s4, 0.f) - s5;
      const f32x4 dv = ld4a(dy + ((int64_t)grp * p.M + r) * p.d, 4 * cg, p.d);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float gv = (!p.relu_b || pc[t] * hc + be[t] > 0.f) ? dv[t] : 0.f;
        a1[t] += gv;
        a2[t] += gv * hc;
        tsum += pc[t] * gv;
      }
    }
    if (pow2) {         // a row's C4 lanes are consecutive lanes of one wave: the row sum is a butterfly, no barrier in the loop
      for (int off = C4 >> 1; off >= 1; off >>= 1) tsum += __shfl_xor(tsum, off, 64);
      if (cg == 0 && rg < nrg && r < r_hi) trow[(int64_t)grp * p.M + r] = tsum;
      continue;
    }
    rsum[threadIdx.x] = tsum;
    __syncthreads();
    if (threadIdx.x < nrg && r0 + threadIdx.x < r_hi) {
      float acc = 0.f;
      for (int q = 0; q < C4; ++q) acc += rsum[threadIdx.x * C4 + q];
      trow[(int64_t)grp * p.M + r0 + threadIdx.x] = acc;
    }
    __syncthreads();
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) { red[0][threadIdx.x][t] = a1[t]; red[1][threadIdx.x][t] = a2[t]; }
  __syncthreads();
  float* S = part + (int64_t)blockIdx.x * 2 * p.d;
  for (int i = threadIdx.x; i < 2 * p.d; i += 256) {
    const int w = i / p.d, c = i - w * p.d;
    float acc = 0.f;
    for (int q = 0; q < nrg; ++q) acc += red[w][q * C4 + (c >> 2)][c & 3];
    S[i] = acc;
  }
}

// column finish (16 columns x 16 lanes per block): the closed-form gradients of w2 / gamma_b / beta_b, summed over the groups in order, and
// the per-column terms of C1, C2: colc[grp][0][c] = p_c m1_c, colc[grp][1][c] = p_c (w2_c/s_c) m2_c
__global__ __launch_bounds__(256) void k_smlp_b2(SMlp p, const float* __restrict__ part, int nblk, float* dw2, float* dgb, float* dbb,
                                                 double* __restrict__ colc, int accumulate) {
  __shared__ double l1[16][17], l2[16][17];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  const int per = (nblk + 15) / 16, b0 = rl * per, b1 = b0 + per < nblk ? b0 + per : nblk;
  double g_dgb = 0.0, g_dbb = 0.0, g_dw2 = 0.0;
  for (int grp = 0; grp < p.G; ++grp) {
    const float* P = part + (int64_t)grp * nblk * 2 * p.d;
    double S1 = 0.0, S2 = 0.0;
    if (c < p.d)
      for (int b = b0; b < b1; ++b) { S1 += (double)P[(int64_t)b * 2 * p.d + c]; S2 += (double)P[(int64_t)b * 2 * p.d + p.d + c]; }
    l1[rl][cl] = S1; l2[rl][cl] = S2;
    __syncthreads();
    for (int step = 8; step >= 1; step >>= 1) {
      if (rl < step) {
        S1 += l1[rl + step][cl]; S2 += l2[rl + step][cl];
        l1[rl][cl] = S1; l2[rl][cl] = S2;
      }
      __syncthreads();
    }
    if (rl == 0 && c < p.d) {
      const double* s = p.sst + grp * 8;
      const double n = s[0], vh = s[6];
      const double w2 = p.w2[c], gam = p.gb ? p.gb[c] : 1.f;
      const double s_c = sqrt(w2 * w2 * vh + (double)p.eps_b), pc = gam * w2 / s_c;
      const double u = w2 / s_c;                       // y_hat = u (h - m_h)
      const double m1 = n > 0.0 ? S1 / n : 0.0, m2 = n > 0.0 ? u * S2 / n : 0.0;
      g_dgb += u * S2;
      g_dbb += S1;
      g_dw2 += (gam / s_c) * (S2 - m2 * u * n * vh);
      colc[(grp * 2 + 0) * p.d + c] = pc * m1;
      colc[(grp * 2 + 1) * p.d + c] = pc * u * m2;
    }
    __syncthreads();
  }
  if (rl == 0 && c < p.d) {
    if (dgb) dgb[c] = (accumulate ? dgb[c] : 0.f) + (float)g_dgb;
    if (dbb) dbb[c] = (accumulate ? dbb[c] : 0.f) + (float)g_dbb;
    if (dw2) dw2[c] = (accumulate ? dw2[c] : 0.f) + (float)g_dw2;
  }
}
struct SRow { double za, pre, gav; };
__device__ __forceinline__ SRow smlp_row(const SMlp& p, const float* __restrict__ trow, int grp, int64_t r, double sw, const double* s,
                                         double C1, double C2) {
  SRow o;
  o.za = sw * (double)p.a[r];
  o.pre = o.za * s[3] + s[4];
  const double dh = (double)trow[(int64_t)grp * p.M + r] - C1 - (fmax(o.pre, 0.0) - s[5]) * C2;
  o.gav = o.pre > 0.0 ? dh : 0.0;
  return o;
}
__device__ __forceinline__ void smlp_c12(const SMlp& p, const double* __restrict__ colc, int grp, double* red, double& C1, double& C2) {
  double c1 = 0.0, c2 = 0.0;
  for (int c = threadIdx.x; c < p.d; c += SM_T) { c1 += colc[(grp * 2) * p.d + c]; c2 += colc[(grp * 2 + 1) * p.d + c]; }
  C1 = block_sum_d(c1, red);
  C2 = block_sum_d(c2, red);
}
// row blocks: sums of the gradient at the one-channel BatchNorm's output: p3[grp][blk] = (sum g_a, sum g_a z_hat)
__global__ __launch_bounds__(SM_T) void k_smlp_b3(SMlp p, const float* __restrict__ trow, const double* __restrict__ colc, double* __restrict__ p3) {
  __shared__ double red[SM_T / 64];
  const int nb = gridDim.x;
  const int64_t lo = p.M * blockIdx.x / nb, hi = p.M * (blockIdx.x + 1) / nb;
  const double w1 = p.w1[0];
  for (int grp = 0; grp < p.G; ++grp) {
    double C1, C2;
    smlp_c12(p, colc, grp, red, C1, C2);
    const double* s = p.sst + grp * 8;
    const double sw = (grp == 1 && p.negate1) ? -w1 : w1;
    double sga = 0.0, sgz = 0.0;
    for (int64_t r = lo + threadIdx.x; r < hi; r += SM_T)
      if (row_ok(r, p.M, p.nvalid, p.K)) {
        const SRow q = smlp_row(p, trow, grp, r, sw, s, C1, C2);
        sga += q.gav;
        sgz += q.gav * (q.za - s[1]) * s[2];
      }
    const double A = block_sum_d(sga, red), Z = block_sum_d(sgz, red);
    if (threadIdx.x == 0) { p3[((int64_t)grp * nb + blockIdx.x) * 2] = A; p3[((int64_t)grp * nb + blockIdx.x) * 2 + 1] = Z; }
  }
}
// row blocks: d z_a per row -> da (both groups, with their signs) and the block sums of d z_a * a:  p4[grp][blk]
__global__ __launch_bounds__(SM_T) void k_smlp_b4(SMlp p, const float* __restrict__ trow, const double* __restrict__ colc,
                                                  const double* __restrict__ p3, double* __restrict__ p4, float* __restrict__ da) {
  __shared__ double red[SM_T / 64];
  const int nb = gridDim.x;
  const int64_t lo = p.M * blockIdx.x / nb, hi = p.M * (blockIdx.x + 1) / nb;
  const double w1 = p.w1[0];
  for (int grp = 0; grp < p.G; ++grp) {
    double C1, C2;
    smlp_c12(p, colc, grp, red, C1, C2);
    const bool inb = (int)threadIdx.x < nb;       // nb <= SM_T
    const double A = block_sum_d(inb ? p3[((int64_t)grp * nb + threadIdx.x) * 2] : 0.0, red);
    const double Z = block_sum_d(inb ? p3[((int64_t)grp * nb + threadIdx.x) * 2 + 1] : 0.0, red);
    const double* s = p.sst + grp * 8;
    const double n = s[0];
    const double ma = n > 0.0 ? A / n : 0.0, mz = n > 0.0 ? Z / n : 0.0;
    const double sgn = (grp == 1 && p.negate1) ? -1.0 : 1.0;
    const double sw = sgn * w1;
    double sw1 = 0.0;
    for (int64_t r = lo + threadIdx.x; r < hi; r += SM_T) {
      double dav = 0.0;
      if (row_ok(r, p.M, p.nvalid, p.K)) {
        const SRow q = smlp_row(p, trow, grp, r, sw, s, C1, C2);
        const double dza = s[3] * (q.gav - ma - (q.za - s[1]) * s[2] * mz);
        sw1 += dza * sgn * (double)p.a[r];
        dav = dza * sw;
      }
      if (da) da[r] = (grp == 0 ? 0.f : da[r]) + (float)dav;
    }
    const double W = block_sum_d(sw1, red);
    if (threadIdx.x == 0) p4[(int64_t)grp * nb + blockIdx.x] = W;
  }
}
__global__ __launch_bounds__(SM_T) void k_smlp_b5(int G, int nb, const double* __restrict__ p3, const double* __restrict__ p4, float* dw1, float* dga,
                                                  float* dba, int accumulate) {
  __shared__ double red[SM_T / 64];
  double w = 0.0, a = 0.0, z = 0.0;
  for (int i = threadIdx.x; i < G * nb; i += SM_T) { w += p4[i]; a += p3[2 * i]; z += p3[2 * i + 1]; }
  w = block_sum_d(w, red); a = block_sum_d(a, red); z = block_sum_d(z, red);
  if (threadIdx.x != 0) return;
  if (dw1) dw1[0] = (accumulate ? dw1[0] : 0.f) + (float)w;
  if (dga) dga[0] = (accumulate ? dga[0] : 0.f) + (float)z;
  if (dba) dba[0] = (accumulate ? dba[0] : 0.f) + (float)a;
}

int train_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
    cus = n > 0 ? n : 256;
  }
  return cus;
}

template <typename KFn>
int raise_lds(KFn fn, size_t lds, const char* who) {
  if (lds <= 64 * 1024) return SN_OK;
  // once per kernel is enough, and nothing but launches may run while a stream is being captured into a HIP graph
  static const void* seen[16];
  static size_t seen_lds[16];
  static int nseen = 0;
  const void* key = reinterpret_cast<const void*>(fn);
  for (int i = 0; i < nseen; ++i)
    if (seen[i] == key && seen_lds[i] >= lds) return SN_OK;
  if (nseen < 16) { seen[nseen] = key; seen_lds[nseen] = lds; ++nseen; }
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return fail(SN_ERR_LAUNCH, "%s: cannot raise the dynamic LDS limit to %zu", who, lds);
  return SN_OK;
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// one arrival word per launch, round-robin over N_TICKETS (the last arriver resets its word: a word is reused N_TICKETS launches later)
unsigned* take_ticket() {
  static unsigned* base[32] = {};
  static std::atomic<unsigned> next{0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return nullptr;
  if (!base[dev]) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_tickets)) != hipSuccess) return nullptr;
    base[dev] = static_cast<unsigned*>(p);
  }
  return base[dev] + (next.fetch_add(1u) % (unsigned)N_TICKETS);
}

}  // namespace
}  // namespace sn
#ifdef SN_PROFILE
extern "C" int sn_prof_read_train(long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(sn::g_tprof), sizeof(long long) * 64); }
#endif

using namespace sn;

// workgroups per group of the forward link (= moment partials per group) / of the backward link (= dW, column-sum partials per group)
// Few rows: one workgroup per 16-row tile (k_tlin_fwd_tile); otherwise the persistent kernel, a workgroup per CU, >= 4 tiles each.
constexpr int64_t TILE_MODE_MAX = 512;          // row tiles, all groups together
// (A/B switch of the profile scripts: SN_TRAIN_SPLIT=0 in the environment keeps the links on the fp32-input MFMA)
static bool train_split_enabled() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("SN_TRAIN_SPLIT"); v = (e && e[0] == '0') ? 0 : 1; }
  return v == 1;
}
static bool tile_mode(int64_t R, int G) { return cdiv(R > 0 ? R : 1, 16) * (G < 1 ? 1 : G) <= TILE_MODE_MAX; }
extern "C" int sn_train_linear_blocks(int64_t R, int G) {
  if (G < 1) G = 1;
  if (tile_mode(R, G)) return (int)cdiv(R > 0 ? R : 1, 16);
  const int64_t want = cdiv(cdiv(R > 0 ? R : 1, 16), 4);
  int64_t cap = train_cus() / G;
  if (cap < 1) cap = 1;
  return (int)(want < cap ? want : cap);
}
// Few rows: 16-row rounds, one per workgroup (measured at 2 950 rows: 15.6 us; 32-row rounds 17.0; the persistent 64-row rounds 23.4)
static int bwd_small_rt() { return 1; }
extern "C" int sn_train_linear_bwd_blocks(int64_t R, int G) {
  if (G < 1) G = 1;
  if (tile_mode(R, G)) return (int)cdiv(R > 0 ? R : 1, 16 * bwd_small_rt());       // one round per workgroup
  const int64_t want = cdiv(R > 0 ? R : 1, TROWS);
  int64_t cap = train_cus() / G;
  if (cap < 1) cap = 1;
  return (int)(want < cap ? want : cap);
}

extern "C" int sn_train_linear_f32(const sn_train_linear_args* args, void* stream) {
  SN_REQUIRE(args, "sn_train_linear_f32: null arguments");
  const sn_train_linear_args& p = *args;
  SN_REQUIRE(p.x && p.W && p.y && p.R >= 0 && p.G >= 1 && p.d_in > 0 && p.d_out > 0, "sn_train_linear_f32: bad arguments");
  SN_REQUIRE(p.d_in <= 128 && p.d_out <= 128 && p.d_in % 4 == 0 && p.d_out % 4 == 0,
             "sn_train_linear_f32: widths (%d -> %d) must be multiples of 4 up to 128", p.d_in, p.d_out);
  SN_REQUIRE(p.ldx >= p.d_in && p.ldy >= p.d_out && p.ldx % 4 == 0 && p.ldy % 4 == 0 && al16(p.x) && al16(p.y) && p.ldw >= p.d_in,
             "sn_train_linear_f32: rows must be 16-byte aligned");
  SN_REQUIRE(!p.nvalid || p.K > 0, "sn_train_linear_f32: nvalid needs K > 0");
  SN_REQUIRE((p.in_scale == nullptr) == (p.in_shift == nullptr) && (!p.in_scale || (al16(p.in_scale) && al16(p.in_shift))),
             "sn_train_linear_f32: in_scale / in_shift go together, 16-byte aligned");
  SN_REQUIRE(!p.fin_state || (p.stat_part && p.fin_count && al16(p.stat_part) && (p.fin_running_mean != nullptr) == (p.fin_running_var != nullptr)),
             "sn_train_linear_f32: the fused finish needs stat_part (16-byte aligned), fin_count and running_mean / running_var together");
  if (p.R == 0) {
    // no rows: no launch; the moment partials a BatchNorm finish would read are defined (count 0), never left uninitialised
    if (p.stat_part) {
      const int nb0 = sn_train_linear_blocks(0, p.G);
      if (hipMemsetAsync(p.stat_part, 0, sizeof(float) * (size_t)p.G * (2 * (size_t)nb0 * p.d_out + nb0), (hipStream_t)stream) != hipSuccess)
        return fail(SN_ERR_LAUNCH, "sn_train_linear_f32: memset of the moment partials failed");
    }
    return SN_OK;
  }
  const int nblk = sn_train_linear_blocks(p.R, p.G);
  TLin a{p.x, p.ldx, p.R, p.G, p.d_in, p.d_out, p.W, p.ldw, p.bias, p.nvalid, p.K, p.in_scale, p.in_shift, p.in_relu, p.out_relu,
         p.y, p.ldy, p.stat_part, nblk, TFin{}};
  if (p.fin_state) {
    a.fin = TFin{p.fin_gamma, p.fin_beta, p.fin_eps, p.fin_momentum, p.fin_running_mean, p.fin_running_var, p.fin_state, p.fin_count,
                 take_ticket()};
    SN_REQUIRE(a.fin.ticket, "sn_train_linear_f32: no arrival ticket");
  }
  const int nti = (p.d_in + 15) / 16, nto = (p.d_out + 15) / 16;
  // (the full-width links with aligned parameter rows: split-bf16 matrix path, three bf16 planes in the image)
  const bool split = p.d_in == 128 && p.d_out == 128 && (reinterpret_cast<uintptr_t>(p.W) & 15) == 0 && (p.ldw & 3) == 0 &&
                     !tile_mode(p.R, p.G) && train_split_enabled();
  const size_t lds = (size_t)(split ? 8 * 4 * 3 : 8 * 8) * 1024 + (size_t)(2 * 16 * 8 + TW * 16 * 8) * sizeof(float);    // weight image + in_scale | in_shift + pivots
  hipStream_t st = (hipStream_t)stream;
  int rc;
  const bool full = p.d_in == 128 && p.d_out == 128;
#define SN_TLIN_FWD(STATS, FULL)                                                                                             \
  do {                                                                                                                        \
    if ((rc = raise_lds(k_tlin_fwd<8, 8, STATS, FULL>, lds, "sn_train_linear_f32")) != SN_OK) return rc;                      \
    hipLaunchKernelGGL((k_tlin_fwd<8, 8, STATS, FULL>), dim3((unsigned)(nblk * p.G)), dim3(64 * TW), lds, st, a);             \
  } while (0)
  if (tile_mode(p.R, p.G)) {
    if (p.stat_part) hipLaunchKernelGGL((k_tlin_fwd_tile<8, true>), dim3((unsigned)(nblk * p.G)), dim3(64 * TW), 0, st, a);
    else hipLaunchKernelGGL((k_tlin_fwd_tile<8, false>), dim3((unsigned)(nblk * p.G)), dim3(64 * TW), 0, st, a);
  } else if (split) {
    if (p.stat_part) {
      if ((rc = raise_lds(k_tlin_fwd<8, 8, true, true, true>, lds, "sn_train_linear_f32")) != SN_OK) return rc;
      hipLaunchKernelGGL((k_tlin_fwd<8, 8, true, true, true>), dim3((unsigned)(nblk * p.G)), dim3(64 * TW), lds, st, a);
    } else {
      if ((rc = raise_lds(k_tlin_fwd<8, 8, false, true, true>, lds, "sn_train_linear_f32")) != SN_OK) return rc;
      hipLaunchKernelGGL((k_tlin_fwd<8, 8, false, true, true>), dim3((unsigned)(nblk * p.G)), dim3(64 * TW), lds, st, a);
    }
  } else if (p.stat_part) { if (full) SN_TLIN_FWD(true, true); else SN_TLIN_FWD(true, false); }
  else { if (full) SN_TLIN_FWD(false, true); else SN_TLIN_FWD(false, false); }
#undef SN_TLIN_FWD
  SN_CHECK_LAUNCH("sn_train_linear_f32");
  return SN_OK;
}

extern "C" int sn_train_bn_finish_f32(const float* stat_part, int nblk, int G, int C, const float* gamma, const float* beta, float eps,
                                      float momentum, float* running_mean, float* running_var, float* state, float* count, void* stream) {
  SN_REQUIRE(stat_part && state && count && nblk >= 1 && G >= 1 && C > 0, "sn_train_bn_finish_f32: bad arguments");
  SN_REQUIRE((running_mean != nullptr) == (running_var != nullptr), "sn_train_bn_finish_f32: running_mean / running_var go together");
  hipLaunchKernelGGL(k_tbn_finish, dim3((unsigned)cdiv(C, 16)), dim3(256), 0, (hipStream_t)stream, stat_part, nblk, G, C, gamma, beta, eps,
                     momentum, running_mean, running_var, state, count);
  SN_CHECK_LAUNCH("sn_train_bn_finish_f32");
  return SN_OK;
}

extern "C" int64_t sn_train_linear_bwd_part_floats(int64_t R, int G, int d_in, int d_out) {
  return (int64_t)sn_train_linear_bwd_blocks(R, G) * G * ((int64_t)d_in * d_out + d_out);
}

extern "C" int sn_train_linear_bwd_f32(const sn_train_linear_bwd_args* args, void* stream) {
  SN_REQUIRE(args, "sn_train_linear_bwd_f32: null arguments");
  const sn_train_linear_bwd_args& p = *args;
  SN_REQUIRE(p.dy && p.x && p.W && p.R >= 0 && p.G >= 1 && p.d_in > 0 && p.d_out > 0, "sn_train_linear_bwd_f32: bad arguments");
  SN_REQUIRE(p.d_in <= 128 && p.d_out <= 128 && p.d_in % 4 == 0 && p.d_out % 4 == 0,
             "sn_train_linear_bwd_f32: widths (%d -> %d) must be multiples of 4 up to 128", p.d_in, p.d_out);
  SN_REQUIRE(p.lddy >= p.d_out && p.ldx >= p.d_in && p.lddy % 4 == 0 && p.ldx % 4 == 0 && al16(p.dy) && al16(p.x) && p.ldw >= p.d_in,
             "sn_train_linear_bwd_f32: rows must be 16-byte aligned");
  SN_REQUIRE(!p.zo || (p.ldzo >= p.d_out && p.ldzo % 4 == 0 && al16(p.zo)), "sn_train_linear_bwd_f32: zo rows must be 16-byte aligned");
  SN_REQUIRE(!p.gx || (p.ldgx >= p.d_in && p.ldgx % 4 == 0 && al16(p.gx)), "sn_train_linear_bwd_f32: gx rows must be 16-byte aligned");
  SN_REQUIRE((!p.coef_a && !p.coef_b && !p.coef_c) || (p.coef_a && p.coef_b && p.coef_c && p.zo), "sn_train_linear_bwd_f32: coef_a/b/c need zo");
  SN_REQUIRE((!p.mask_scale && !p.mask_shift) || (p.mask_scale && p.mask_shift && p.zo), "sn_train_linear_bwd_f32: mask_scale/shift need zo");
  SN_REQUIRE((p.x_scale == nullptr) == (p.x_shift == nullptr), "sn_train_linear_bwd_f32: x_scale / x_shift go together");
  SN_REQUIRE(!p.x_mean || (p.gx && p.sums_part), "sn_train_linear_bwd_f32: x_mean needs gx and sums_part");
  SN_REQUIRE(!p.gx_accumulate || (p.gx && !p.x_mean && !p.dot_x), "sn_train_linear_bwd_f32: gx_accumulate excludes the column sums");
  SN_REQUIRE(!p.nvalid || p.K > 0, "sn_train_linear_bwd_f32: nvalid needs K > 0");
  SN_REQUIRE(!p.dot_x || (p.dot_part && p.gx && p.lddot >= p.d_in && p.lddot % 4 == 0 && al16(p.dot_x)),
             "sn_train_linear_bwd_f32: dot_x needs gx, dot_part and 16-byte aligned rows");
  for (const float* v : {p.coef_a, p.coef_b, p.coef_c, p.mask_scale, p.mask_shift, p.x_scale, p.x_shift, p.x_mean})
    SN_REQUIRE(!v || al16(v), "sn_train_linear_bwd_f32: column vectors must be 16-byte aligned");
  if (p.R == 0) return SN_OK;
  const int nblk = sn_train_linear_bwd_blocks(p.R, p.G);
  TBwd a{p.R, p.G, p.nvalid, p.K, p.d_in, p.d_out, p.dy, p.lddy, p.zo, p.ldzo, p.coef_a, p.coef_b, p.coef_c, p.mask_scale, p.mask_shift,
         p.x, p.ldx, p.x_scale, p.x_shift, p.x_relu, p.x_mean, p.W, p.ldw, p.gx, p.ldgx, p.sums_part, p.dw_part, p.want_db,
         p.gx_accumulate, p.dot_x, p.lddot, p.dot_part, nblk, TBFin{}, TBMerge{}};
  if (p.merge_sums) {
    SN_REQUIRE(!p.coef_a && p.zo && p.merge_nblk >= 1 && p.merge_state && p.merge_count && al16(p.merge_sums),
               "sn_train_linear_bwd_f32: merge_sums excludes coef_a/b/c and needs zo, merge_nblk, merge_state, merge_count (sums 16-byte aligned)");
    a.mg = TBMerge{p.merge_sums, p.merge_nblk, p.merge_state, p.merge_count, p.merge_gamma, p.merge_dgamma, p.merge_dbeta, p.merge_accumulate};
  }
  if (p.fin_coef || p.fin_dot_out) {
    SN_REQUIRE(!p.fin_coef || (p.x_mean && p.sums_part && al16(p.sums_part) && p.fin_state && p.fin_count),
               "sn_train_linear_bwd_f32: the fused BatchNorm finish needs x_mean, sums_part (16-byte aligned), fin_state and fin_count");
    SN_REQUIRE(!p.fin_dot_out || (p.dot_x && p.dot_part), "sn_train_linear_bwd_f32: the fused eps finish needs dot_x / dot_part");
    a.fin = TBFin{p.fin_state, p.fin_count, p.fin_gamma, p.fin_coef, p.fin_dgamma, p.fin_dbeta, p.fin_accumulate, p.fin_dot_out, take_ticket()};
    SN_REQUIRE(a.fin.ticket, "sn_train_linear_bwd_f32: no arrival ticket");
  }
  auto lds_of = [](int rt, bool split = false) {
    return (size_t)(split ? 8 * 4 * 3 : 8 * 8) * 1024 + (size_t)16 * rt * (stage_ld(8) + stage_ld(8)) * sizeof(float) +
           (size_t)(3 * 4 + 3 + 5) * 16 * 8 * sizeof(float);
  };
  const bool full = p.d_in == 128 && p.d_out == 128;
  const int rt = tile_mode(p.R, p.G) ? bwd_small_rt() : 4;
  int rc;
  if (full && rt == 4 && p.gx && train_split_enabled()) {       // the large full-width links: dX on the split-bf16 path, 32-row rounds
    if ((rc = raise_lds(k_tlin_bwd<8, 8, true, 2, true>, lds_of(2, true), "sn_train_linear_bwd_f32")) != SN_OK) return rc;
    hipLaunchKernelGGL((k_tlin_bwd<8, 8, true, 2, true>), dim3((unsigned)(nblk * p.G)), dim3(64 * TW), lds_of(2, true), (hipStream_t)stream, a);
    SN_CHECK_LAUNCH("sn_train_linear_bwd_f32");
    return SN_OK;
  }
#define SN_TLIN_BWD(FULL, RT)                                                                                                    \
  do {                                                                                                                            \
    if ((rc = raise_lds(k_tlin_bwd<8, 8, FULL, RT>, lds_of(RT), "sn_train_linear_bwd_f32")) != SN_OK) return rc;                  \
    hipLaunchKernelGGL((k_tlin_bwd<8, 8, FULL, RT>), dim3((unsigned)(nblk * p.G)), dim3(64 * TW), lds_of(RT), (hipStream_t)stream, a); \
  } while (0)
  if (rt == 1) { if (full) SN_TLIN_BWD(true, 1); else SN_TLIN_BWD(false, 1); }
  else { if (full) SN_TLIN_BWD(true, 4); else SN_TLIN_BWD(false, 4); }
#undef SN_TLIN_BWD
  SN_CHECK_LAUNCH("sn_train_linear_bwd_f32");
  return SN_OK;
}

extern "C" int sn_train_bn_bwd_blocks(int64_t R, int G) {
  if (G < 1) G = 1;
  const int64_t want = cdiv(R > 0 ? R : 1, tile_mode(R, G) ? 32 : 128);      // (few rows: the pass is a chain of row loads per thread)
  int64_t cap = 2 * train_cus() / G;
  if (cap < 1) cap = 1;
  return (int)(want < cap ? want : cap);
}

extern "C" int sn_train_bn_bwd_sums_f32(const float* dy, int lddy, const float* z, int ldz, int64_t R, int G, int C, const int32_t* nvalid,
                                        int K, const float* state, int relu, float* sums_part, void* stream) {
  SN_REQUIRE(dy && z && state && sums_part && R >= 0 && G >= 1 && C > 0 && C % 4 == 0 && C <= 1024, "sn_train_bn_bwd_sums_f32: bad arguments");
  SN_REQUIRE(lddy >= C && ldz >= C && lddy % 4 == 0 && ldz % 4 == 0 && al16(dy) && al16(z) && al16(state),
             "sn_train_bn_bwd_sums_f32: rows must be 16-byte aligned");
  SN_REQUIRE(!nvalid || K > 0, "sn_train_bn_bwd_sums_f32: nvalid needs K > 0");
  const int nblk = sn_train_bn_bwd_blocks(R, G);
  hipLaunchKernelGGL(k_tbn_bwd_sums, dim3((unsigned)(nblk * G)), dim3(256), 0, (hipStream_t)stream, dy, lddy, z, ldz, R, G, C, nvalid, K, state,
                     relu, nblk, sums_part, TBFin{});
  SN_CHECK_LAUNCH("sn_train_bn_bwd_sums_f32");
  return SN_OK;
}

extern "C" int sn_train_bn_bwd_f32(const float* dy, int lddy, const float* z, int ldz, int64_t R, int G, int C, const int32_t* nvalid, int K,
                                   const float* state, const float* count, int relu, const float* gamma, float* sums_part, float* coef,
                                   float* dgamma, float* dbeta, int accumulate, void* stream) {
  SN_REQUIRE(dy && z && state && count && sums_part && coef && R >= 0 && G >= 1 && C > 0 && C % 4 == 0 && C <= 128,
             "sn_train_bn_bwd_f32: bad arguments (C: a multiple of 4 up to 128)");
  SN_REQUIRE(lddy >= C && ldz >= C && lddy % 4 == 0 && ldz % 4 == 0 && al16(dy) && al16(z) && al16(state) && al16(sums_part),
             "sn_train_bn_bwd_f32: rows must be 16-byte aligned");
  SN_REQUIRE(!nvalid || K > 0, "sn_train_bn_bwd_f32: nvalid needs K > 0");
  const int nblk = sn_train_bn_bwd_blocks(R, G);
  const TBFin fin{state, count, gamma, coef, dgamma, dbeta, accumulate, nullptr, take_ticket()};
  SN_REQUIRE(fin.ticket, "sn_train_bn_bwd_f32: no arrival ticket");
  hipLaunchKernelGGL(k_tbn_bwd_sums, dim3((unsigned)(nblk * G)), dim3(256), 0, (hipStream_t)stream, dy, lddy, z, ldz, R, G, C, nvalid, K, state,
                     relu, nblk, sums_part, fin);
  SN_CHECK_LAUNCH("sn_train_bn_bwd_f32");
  return SN_OK;
}

extern "C" int sn_train_bn_bwd_finish_f32(const float* sums_part, int nblk, int G, int C, const float* state, const float* count,
                                          const float* gamma, float* coef, float* dgamma, float* dbeta, int accumulate, void* stream) {
  SN_REQUIRE(sums_part && state && count && coef && nblk >= 1 && G >= 1 && C > 0, "sn_train_bn_bwd_finish_f32: bad arguments");
  hipLaunchKernelGGL(k_tbn_bwd_finish, dim3((unsigned)cdiv(C, 16)), dim3(256), 0, (hipStream_t)stream, sums_part, nblk, G, C, state, count,
                     gamma, coef, dgamma, dbeta, accumulate);
  SN_CHECK_LAUNCH("sn_train_bn_bwd_finish_f32");
  return SN_OK;
}

extern "C" int sn_train_bn_apply_f32(const float* z, int ldz, int64_t R, int G, int C, const int32_t* nvalid, int K, const float* state,
                                     int relu, const float* residual, int ldr, float* y, int ldy, void* stream) {
  SN_REQUIRE(z && state && y && R >= 0 && G >= 1 && C > 0 && C % 4 == 0, "sn_train_bn_apply_f32: bad arguments");
  SN_REQUIRE(ldz >= C && ldy >= C && ldz % 4 == 0 && ldy % 4 == 0 && al16(z) && al16(y) && al16(state) &&
                 (!residual || (ldr >= C && ldr % 4 == 0 && al16(residual))),
             "sn_train_bn_apply_f32: rows must be 16-byte aligned");
  SN_REQUIRE(!nvalid || K > 0, "sn_train_bn_apply_f32: nvalid needs K > 0");
  const int64_t n = (int64_t)G * R * (C / 4);
  if (n == 0) return SN_OK;
  hipLaunchKernelGGL(k_tbn_apply, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, z, ldz, R, G, C, nvalid, K, state, relu,
                     residual, ldr, y, ldy);
  SN_CHECK_LAUNCH("sn_train_bn_apply_f32");
  return SN_OK;
}

// out[0] (+)= (float) sum of n float64 partials — the eps gradient of an aggregation (sn_train_linear_bwd_f32's dot_part), one launch
__device__ __forceinline__ void tdot_finish_block(int tid, const double* __restrict__ part, int n, float* __restrict__ out, int accumulate) {
  __shared__ double red[4];
  double t = 0.0;
  for (int i = tid; i < n; i += 256) t += part[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
  if ((tid & 63) == 0) red[tid >> 6] = t;
  __syncthreads();
  if (tid == 0) out[0] = (accumulate ? out[0] : 0.f) + (float)((red[0] + red[1]) + (red[2] + red[3]));
}
__global__ __launch_bounds__(256) void k_tdot_finish(const double* __restrict__ part, int n, float* __restrict__ out, int accumulate) {
  tdot_finish_block(threadIdx.x, part, n, out, accumulate);
}

// What follows a backward link, in ONE launch: the dW (and db) reduction of the link's per-workgroup partials INTO the parameters'
// gradients, the BatchNorm-backward finish of the producer link (its column sums came out of the same kernel) and the eps-gradient
// finish — three tiny kernels that each sat at the ~5 us dependent-launch floor, ~21 launches a step.  A block does one job.
struct TPost {
  const float* part; int nparts; int64_t stride; int64_t n_w; float* out_w; int64_t n_b; float* out_b;
  const float* sums; int nblk; int G; int C; const float* st; const float* cnt; const float* gamma; float* coef; float* dgamma; float* dbeta;
  int acc_f;
  const double* dpart; int dn; float* dout;
  int nbw, nbb, nbf;
};
__global__ __launch_bounds__(1024) void k_tpost(TPost p) {
  int b = blockIdx.x;
  const int tid = threadIdx.x;
  if (b < p.nbw) { tsum_parts_block(b, tid, p.part, p.nparts, p.stride, p.n_w, p.out_w, 1); return; }
  b -= p.nbw;
  if (b < p.nbb) { tsum_parts_block(b, tid, p.part + p.n_w, p.nparts, p.stride, p.n_b, p.out_b, 1); return; }
  b -= p.nbb;
  if (tid >= 256) return;                 // (the two finishes are 256-thread jobs: the other waves leave before any barrier)
  if (b < p.nbf) { tbn_bwd_finish_block(b, tid, p.sums, p.nblk, p.G, p.C, p.st, p.cnt, p.gamma, p.coef, p.dgamma, p.dbeta, p.acc_f); return; }
  tdot_finish_block(tid, p.dpart, p.dn, p.dout, 1);
}
// ... for a table of eps gradients in one launch (a block per job): the dot finishes of every aggregation of a step, at the end of backward
struct TDotJobs { const double* part[SN_TRAIN_MAX_REDUCE_JOBS]; float* out[SN_TRAIN_MAX_REDUCE_JOBS]; int n[SN_TRAIN_MAX_REDUCE_JOBS]; };
__global__ __launch_bounds__(256) void k_tdot_jobs(TDotJobs J) {
  tdot_finish_block(threadIdx.x, J.part[blockIdx.x], J.n[blockIdx.x], J.out[blockIdx.x], 1);
}
extern "C" int sn_train_dot_jobs_f64(const sn_train_dot_job* jobs, int njobs, void* stream) {
  SN_REQUIRE(jobs && njobs >= 1 && njobs <= SN_TRAIN_MAX_REDUCE_JOBS, "sn_train_dot_jobs_f64: 1..%d jobs", SN_TRAIN_MAX_REDUCE_JOBS);
  TDotJobs J{};
  for (int j = 0; j < njobs; ++j) {
    SN_REQUIRE(jobs[j].part && jobs[j].out && jobs[j].n >= 1, "sn_train_dot_jobs_f64: bad job %d", j);
    J.part[j] = jobs[j].part; J.out[j] = jobs[j].out; J.n[j] = jobs[j].n;
  }
  hipLaunchKernelGGL(k_tdot_jobs, dim3((unsigned)njobs), dim3(256), 0, (hipStream_t)stream, J);
  SN_CHECK_LAUNCH("sn_train_dot_jobs_f64");
  return SN_OK;
}

extern "C" int sn_train_dot_finish_f64(const double* part, int n, float* out, int accumulate, void* stream) {
  SN_REQUIRE(part && out && n >= 1, "sn_train_dot_finish_f64: bad arguments");
  hipLaunchKernelGGL(k_tdot_finish, dim3(1), dim3(256), 0, (hipStream_t)stream, part, n, out, accumulate);
  SN_CHECK_LAUNCH("sn_train_dot_finish_f64");
  return SN_OK;
}

extern "C" int sn_train_reduce_parts_f32(const float* part, int nparts, int64_t stride, int64_t n, float* out, int accumulate, void* stream) {
  SN_REQUIRE(part && out && nparts >= 1 && n >= 0 && stride >= n, "sn_train_reduce_parts_f32: bad arguments");
  if (n == 0) return SN_OK;
  hipLaunchKernelGGL(k_tsum_parts, dim3((unsigned)cdiv(n, 64)), dim3(1024), 0, (hipStream_t)stream, part, nparts, stride, n, out, accumulate);
  SN_CHECK_LAUNCH("sn_train_reduce_parts_f32");
  return SN_OK;
}

extern "C" int sn_train_reduce_jobs_f32(const sn_train_reduce_job* jobs, int njobs, void* stream) {
  SN_REQUIRE(jobs && njobs >= 1 && njobs <= SN_TRAIN_MAX_REDUCE_JOBS, "sn_train_reduce_jobs_f32: 1..%d jobs", SN_TRAIN_MAX_REDUCE_JOBS);
  TJobs J{};
  int nb = 0, k = 0;
  for (int j = 0; j < njobs; ++j) {
    const sn_train_reduce_job& q = jobs[j];
    SN_REQUIRE(q.part && q.out && q.nparts >= 1 && q.n >= 0 && q.stride >= q.n, "sn_train_reduce_jobs_f32: bad job %d", j);
    if (q.n == 0) continue;
    J.part[k] = q.part; J.out[k] = q.out; J.stride[k] = q.stride; J.n[k] = q.n; J.nparts[k] = q.nparts; J.first[k] = nb;
    if (q.accumulate) { if (k < 32) J.acc_mask_lo |= 1 << k; else J.acc_mask_hi |= 1 << (k - 32); }
    nb += (int)cdiv(q.n, 64);
    ++k;
  }
  if (k == 0) return SN_OK;
  J.first[k] = nb;
  J.njobs = k;
  hipLaunchKernelGGL(k_treduce_jobs, dim3((unsigned)nb), dim3(1024), 0, (hipStream_t)stream, J);
  SN_CHECK_LAUNCH("sn_train_reduce_jobs_f32");
  return SN_OK;
}

extern "C" int sn_train_post_link_f32(const sn_train_post_args* args, void* stream) {
  SN_REQUIRE(args, "sn_train_post_link_f32: null arguments");
  const sn_train_post_args& q = *args;
  SN_REQUIRE(q.dw_part && q.nparts >= 1 && q.n_w > 0 && q.dw_out && q.stride >= q.n_w + q.n_b && (q.n_b == 0 || q.db_out),
             "sn_train_post_link_f32: the dW reduction is mandatory (accumulating into dw_out / db_out)");
  SN_REQUIRE(!q.sums_part || (q.state && q.count && q.coef && q.nblk >= 1 && q.G >= 1 && q.C > 0), "sn_train_post_link_f32: bad BatchNorm finish arguments");
  SN_REQUIRE(!q.dot_part || (q.dot_out && q.dot_n >= 1), "sn_train_post_link_f32: bad dot finish arguments");
  TPost p{q.dw_part, q.nparts, q.stride, q.n_w, q.dw_out, q.n_b, q.db_out, q.sums_part, q.nblk, q.G, q.C, q.state, q.count, q.gamma, q.coef,
          q.dgamma, q.dbeta, q.accumulate_bn, q.dot_part, q.dot_n, q.dot_out, 0, 0, 0};
  p.nbw = (int)cdiv(q.n_w, 64);
  p.nbb = q.n_b > 0 ? (int)cdiv(q.n_b, 64) : 0;
  p.nbf = q.sums_part ? (int)cdiv(q.C, 16) : 0;
  const int nbd = q.dot_part ? 1 : 0;
  hipLaunchKernelGGL(k_tpost, dim3((unsigned)(p.nbw + p.nbb + p.nbf + nbd)), dim3(1024), 0, (hipStream_t)stream, p);
  SN_CHECK_LAUNCH("sn_train_post_link_f32");
  return SN_OK;
}

static int smlp_check(const sn_train_scalar_mlp_args* args, const char* who) {
  SN_REQUIRE(args, "%s: null arguments", who);
  const sn_train_scalar_mlp_args& p = *args;
  SN_REQUIRE(p.a && p.w1 && p.w2 && p.scalar_state && p.column_state && p.M >= 0 && p.G >= 1 && p.G <= 2 && p.d > 0 && p.d % 4 == 0 && p.d <= 1024,
             "%s: bad arguments", who);
  SN_REQUIRE(!p.nvalid || p.K > 0, "%s: nvalid needs K > 0", who);
  SN_REQUIRE(al16(p.column_state), "%s: column_state must be 16-byte aligned", who);
  return SN_OK;
}
static SMlp smlp_of(const sn_train_scalar_mlp_args& p) {
  return SMlp{p.a, p.M, p.G, p.negate_second, p.nvalid, p.K, p.d, p.w1, p.gamma_a, p.beta_a, p.eps_a, p.w2, p.b2, p.gamma_b, p.beta_b,
              p.eps_b, p.relu_b, p.scalar_state, p.column_state};
}

static int smlp_nb(int64_t M) {
  const int64_t want = (M + SM_T - 1) / SM_T;
  return (int)(want < 1 ? 1 : (want < SM_MAXB ? want : SM_MAXB));
}
extern "C" int64_t sn_train_scalar_mlp_work_doubles(int64_t M, int G, int d) {
  const int64_t nb = smlp_nb(M);
  if (G < 1) G = 1;
  return 3 * nb + 3 * (int64_t)G * nb + 2 * (int64_t)G * d + 2 * (int64_t)G * nb + (int64_t)G * nb + 16;
}

extern "C" int sn_train_scalar_mlp_stats_f32(const sn_train_scalar_mlp_args* args, float momentum_a, float* running_mean_a,
                                             float* running_var_a, float momentum_b, float* running_mean_b, float* running_var_b,
                                             double* work, void* stream) {
  int rc = smlp_check(args, "sn_train_scalar_mlp_stats_f32");
  if (rc != SN_OK) return rc;
  SN_REQUIRE(work, "sn_train_scalar_mlp_stats_f32: work buffer missing");
  SN_REQUIRE((running_mean_a != nullptr) == (running_var_a != nullptr) && (running_mean_b != nullptr) == (running_var_b != nullptr),
             "sn_train_scalar_mlp_stats_f32: running_mean / running_var go together");
  const int nb = smlp_nb(args->M);
  double* p1 = work;
  double* p2 = work + 3 * nb;
  hipStream_t st = (hipStream_t)stream;
  const SMlp p = smlp_of(*args);
  hipLaunchKernelGGL(k_smlp_s1, dim3(nb), dim3(SM_T), 0, st, p, p1);
  hipLaunchKernelGGL(k_smlp_s2, dim3(nb), dim3(SM_T), 0, st, p, p1, p2, momentum_a, running_mean_a, running_var_a);
  hipLaunchKernelGGL(k_smlp_s3, dim3(1), dim3(SM_T), 0, st, p, p2, nb, momentum_b, running_mean_b, running_var_b);
  SN_CHECK_LAUNCH("sn_train_scalar_mlp_stats_f32");
  return SN_OK;
}

extern "C" int sn_train_scalar_mlp_apply_f32(const sn_train_scalar_mlp_args* args, float* y, void* stream) {
  int rc = smlp_check(args, "sn_train_scalar_mlp_apply_f32");
  if (rc != SN_OK) return rc;
  SN_REQUIRE(y && al16(y), "sn_train_scalar_mlp_apply_f32: y must be 16-byte aligned");
  const int64_t n = (int64_t)args->G * args->M * (args->d / 4);
  if (n == 0) return SN_OK;
  hipLaunchKernelGGL(k_smlp_apply, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, smlp_of(*args), y);
  SN_CHECK_LAUNCH("sn_train_scalar_mlp_apply_f32");
  return SN_OK;
}

extern "C" int sn_train_scalar_mlp_bwd_f32(const sn_train_scalar_mlp_args* args, const float* dy, float* part, float* row_sums, float* dw1,
                                           float* dgamma_a, float* dbeta_a, float* dw2, float* dgamma_b, float* dbeta_b, float* da,
                                           int accumulate, double* work, void* stream) {
  int rc = smlp_check(args, "sn_train_scalar_mlp_bwd_f32");
  if (rc != SN_OK) return rc;
  SN_REQUIRE(dy && part && row_sums && work && al16(dy) && args->d <= 1024, "sn_train_scalar_mlp_bwd_f32: bad arguments");
  if (args->M == 0) return SN_OK;
  const int nblk = sn_train_bn_bwd_blocks(args->M, args->G);
  const int nb = smlp_nb(args->M);
  const int G = args->G, d = args->d;
  double* colc = work;
  double* p3 = colc + 2 * (int64_t)G * d;
  double* p4 = p3 + 2 * (int64_t)G * nb;
  hipStream_t st = (hipStream_t)stream;
  const SMlp p = smlp_of(*args);
  hipLaunchKernelGGL(k_smlp_bwd_pass, dim3((unsigned)(nblk * G)), dim3(256), 0, st, p, dy, nblk, part, row_sums);
  hipLaunchKernelGGL(k_smlp_b2, dim3((unsigned)cdiv(d, 16)), dim3(256), 0, st, p, part, nblk, dw2, dgamma_b, dbeta_b, colc, accumulate);
  hipLaunchKernelGGL(k_smlp_b3, dim3(nb), dim3(SM_T), 0, st, p, row_sums, colc, p3);
  hipLaunchKernelGGL(k_smlp_b4, dim3(nb), dim3(SM_T), 0, st, p, row_sums, colc, p3, p4, da);
  hipLaunchKernelGGL(k_smlp_b5, dim3(1), dim3(SM_T), 0, st, G, nb, p3, p4, dw1, dgamma_a, dbeta_a, accumulate);
  SN_CHECK_LAUNCH("sn_train_scalar_mlp_bwd_f32");
  return SN_OK;
}
