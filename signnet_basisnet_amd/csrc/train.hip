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
#include "common.hpp"
#include <stdlib.h>

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
__device__ __forceinline__ bool row_ok(int64_t r, int64_t R, const int32_t* __restrict__ nvalid, int K) {
  if (r >= R) return false;
  if (!nvalid) return true;
  const int64_t node = r / K;
  return (int)(r - node * K) < nvalid[node];
}

// Weight image in the MFMA "A" fragment order of common.hpp, built from the RAW row-major parameter (no pack launch, any alignment):
//   wl[(ot*nk + kk)*64 + lane] = { M[16 ot + (lane&15)][16 kk + 4 (lane>>4) + t] }_t,  M = W (trans = 0: [n_o, n_k]) or W^T
template <bool TRANS>
__device__ __forceinline__ void stage_weight(float4* wl, const float* __restrict__ W, int ldw, int n_o, int n_k, int nto, int ntk) {
  const int total = nto * ntk * 64;
  // four entries (16 scalar loads) in flight per thread: one entry at a time the staging of a 128 x 128 matrix was a chain of 32
  // dependent L2 round trips — most of a small launch
  for (int i0 = threadIdx.x; i0 < total; i0 += 4 * 64 * TW) {
    float v[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
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
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * 64 * TW;
      if (i < total) wl[i] = make_float4(v[u][0], v[u][1], v[u][2], v[u][3]);
    }
  }
}

// ============================================================================ forward link
struct TLin {
  const float* x; int ldx; int64_t R; int G; int d_in, d_out;
  const float* W; int ldw; const float* bias;
  const int32_t* nvalid; int K;
  const float* in_scale; const float* in_shift; int in_relu; int out_relu;
  float* y; int ldy;
  float* stat;          // per group: [mean nblk*d_out | M2 nblk*d_out | count nblk]
  int nblk;             // workgroups per group
};

template <int NTI, int NTO, bool STATS>
__global__ __launch_bounds__(64 * TW, 1) void k_tlin_fwd(TLin a) {
  extern __shared__ __align__(16) unsigned char t_lds[];
  float4* wl = reinterpret_cast<float4*>(t_lds);
  float* icol = reinterpret_cast<float*>(t_lds) + (size_t)NTI * NTO * 256;     // [2][16*NTI] in_scale | in_shift of my group
  float* piv = icol + 2 * 16 * NTI;                                             // [TW][16*NTO] per-wave pivots of the moment sums
  constexpr int CI = 16 * NTI;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
  const int nti = (a.d_in + 15) >> 4, nto = (a.d_out + 15) >> 4;
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
  f32x4 s1[STATS ? NTO : 1], s2[STATS ? NTO : 1];
  if (STATS) {
#pragma unroll
    for (int ot = 0; ot < NTO; ++ot) { s1[ot] = f32x4{0.f, 0.f, 0.f, 0.f}; s2[ot] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  }
  float* mypiv = piv + wave * 16 * NTO;
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
  stage_weight<false>(wl, a.W, a.ldw, a.d_out, a.d_in, nto, nti);
  for (int i = threadIdx.x; i < CI; i += 64 * TW) {
    icol[i] = (isc && i < a.d_in) ? isc[i] : 1.f;
    icol[CI + i] = (isc && i < a.d_in) ? ish[i] : 0.f;
  }
  __syncthreads();
  for (; tile < t_hi; tile += TW) {
    const int64_t row = tile * 16 + (lane & 15);
    const bool inr = row < a.R;
    const bool more = tile + TW < t_hi;
    if (more) {        // the next tile's rows are requested before this tile goes into the matrix pipe
      nvalid_next = row_ok((tile + TW) * 16 + (lane & 15), a.R, a.nvalid, a.K);
      fetch(tile + TW, nvalid_next, nx);
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
          pv = lds4(mypiv + o0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = valid ? v[r] - pv[r] : 0.f;
          s1[ot][r] += d;
          s2[ot][r] += d * d;
        }
      }
    };
    if (vb == 0ull) {
      if (inr) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        for (int ot = 0; ot < nto; ++ot) st4a(yr, 16 * ot + 4 * g, a.d_out, z);
      }
    } else {
#pragma unroll
      for (int ot = 0; ot < NTO; ot += 2) {
        if (ot + 1 < nto) {
          f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
          const float4* w0 = wl + (ot * nti) * 64 + lane;
          const float4* w1 = w0 + nti * 64;
#pragma unroll
          for (int kk = 0; kk < NTI; ++kk) {
            if (kk < nti) {
              const float4 p = w0[kk * 64], q = w1[kk * 64];
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
    if (more) {
#pragma unroll
      for (int kk = 0; kk < NTI; ++kk) in[kk] = nx[kk];
      valid = nvalid_next;
    }
  }
  if (STATS) {
    // per-wave (n, mean, M2) from the pivoted sums, then one partial per WORKGROUP: the waves meet in LDS, merged in wave order (Chan)
    __syncthreads();
    float* sm = reinterpret_cast<float*>(t_lds);          // [TW][mean 16*NTO | M2 16*NTO], counts behind  (the weight image is dead)
    float* sc = sm + TW * 2 * 16 * NTO;
#pragma unroll
    for (int ot = 0; ot < NTO; ++ot) {
      f32x4 pv = {0.f, 0.f, 0.f, 0.f};
      if (have_piv) pv = lds4(mypiv + 16 * ot + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float t1 = t16_sum(s1[ot][r]), t2 = t16_sum(s2[ot][r]);
        if ((lane & 15) == 0) {
          const float inv = rn > 0.f ? 1.0f / rn : 0.f;
          sm[(wave * 2 + 0) * 16 * NTO + 16 * ot + 4 * g + r] = pv[r] + t1 * inv;
          sm[(wave * 2 + 1) * 16 * NTO + 16 * ot + 4 * g + r] = fmaxf(t2 - t1 * t1 * inv, 0.f);
        }
      }
    }
    if (lane == 0) sc[wave] = rn;
    __syncthreads();
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
      stg[(int64_t)blk * a.d_out + c] = m;
      stg[((int64_t)a.nblk + blk) * a.d_out + c] = q;
    }
    if (threadIdx.x == 0) {
      float n = 0.f;
#pragma unroll
      for (int w = 0; w < TW; ++w) n += sc[w];
      stg[2 * (int64_t)a.nblk * a.d_out + blk] = n;
    }
  }
}

// Finish of the train-mode BatchNorm(s) of one forward link: merges the per-workgroup moments of every group (Chan), writes the
// state the consumers and the backward read — st[0..4][grp][C] = mean, var (biased), rstd, scale = gamma*rstd, shift = beta - mean*scale;
// cnt[grp] — and applies the running-statistics updates in group order (two sequential calls of the module in the reference).
// grid cdiv(C,16), 256 threads = 16 columns x 16 lanes.
__device__ __forceinline__ void chan(float& na, float& ma, float& qa, float nb, float mb, float qb) {
  if (nb <= 0.f) return;
  const float n = na + nb, d = mb - ma;
  ma += d * (nb / n);
  qa += qb + d * d * (na * nb / n);
  na = n;
}
__global__ __launch_bounds__(256) void k_tbn_finish(const float* __restrict__ stat, int nblk, int G, int C, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, float eps, float momentum, float* __restrict__ rmean,
                                                    float* __restrict__ rvar, float* __restrict__ st, float* __restrict__ cnt) {
  __shared__ float ln[16][17], lm[16][17], lq[16][17];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  const int per = (nblk + 15) / 16, b0 = rl * per, b1 = b0 + per < nblk ? b0 + per : nblk;
  for (int grp = 0; grp < G; ++grp) {
    const float* sg = stat + (int64_t)grp * (2 * (int64_t)nblk * C + nblk);
    float n = 0.f, m = 0.f, q = 0.f;
    if (c < C)
      for (int b = b0; b < b1; ++b) chan(n, m, q, sg[2 * (int64_t)nblk * C + b], sg[(int64_t)b * C + c], sg[((int64_t)nblk + b) * C + c]);
    ln[rl][cl] = n; lm[rl][cl] = m; lq[rl][cl] = q;
    __syncthreads();
    for (int step = 8; step >= 1; step >>= 1) {
      if (rl < step) {
        chan(n, m, q, ln[rl + step][cl], lm[rl + step][cl], lq[rl + step][cl]);
        ln[rl][cl] = n; lm[rl][cl] = m; lq[rl][cl] = q;
      }
      __syncthreads();
    }
    if (rl == 0 && c < C) {
      const float v = n > 0.f ? q / n : 0.f;
      const float rs = 1.0f / sqrtf(v + eps);
      const float sc = (gamma ? gamma[c] : 1.f) * rs;
      float* s = st + (int64_t)grp * C;               // st[component][grp][C]: every component is one contiguous [G][C] block
      const int64_t GC = (int64_t)G * C;
      s[c] = m; s[GC + c] = v; s[2 * GC + c] = rs; s[3 * GC + c] = sc; s[4 * GC + c] = (beta ? beta[c] : 0.f) - m * sc;
      if (c == 0) cnt[grp] = n;
      if (rmean) {
        const float unb = n > 1.f ? v * (n / (n - 1.f)) : v;
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * m;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * unb;
      }
    }
    __syncthreads();
  }
}

// ============================================================================ backward link
// dz (this Linear's output gradient) is formed on load:  g = dy * [ms*zo + mt > 0]  (the ReLU behind this Linear's BatchNorm; ms NULL:
// g = dy), dz = A*g - B - C*zo (the BatchNorm backward with the column constants of k_tbn_bwd_finish; A NULL: dz = g).
// x_hat (the Linear's operand) is x, or relu?(xs*x + xt) when the operand was the producer's BatchNorm applied on load (xs != NULL).
// Outputs: gx = (dz W) * [x_hat > 0 if xrelu] — the gradient at the producer's BatchNorm output, masked by its ReLU;
//          sums[grp][blk][0][c] = sum_rows gx, sums[..][1][c] = sum_rows gx * (x - xmu)  (xmu != NULL: for the producer's BatchNorm backward);
//          dwp[blk] = sum_rows dz^T x_hat  (+ db behind it): per-workgroup partials, all groups together (shared weights).
struct TBwd {
  int64_t R; int G; const int32_t* nvalid; int K; int d_in, d_out;
  const float* dy; int lddy; const float* zo; int ldzo;
  const float* cA; const float* cB; const float* cC; const float* ms; const float* mt;     // [G][d_out]
  const float* x; int ldx; const float* xs; const float* xt; int xrelu; const float* xmu;   // [G][d_in]
  const float* W; int ldw;
  float* gx; int ldgx; float* sums; float* dwp; int want_db;
  int nblk;             // workgroups per group
  int dbg;
};

__host__ __device__ constexpr int stage_ld(int ntiles) { return ((16 * ntiles + 63) / 64) * 64 + 16; }   // row stride = 16 mod 64 banks

template <int NTI, int NTO>
__global__ __launch_bounds__(64 * TW, 1) void k_tlin_bwd(TBwd a) {
  constexpr int LDO = stage_ld(NTO), LDI = stage_ld(NTI);
  extern __shared__ __align__(16) unsigned char t_lds[];
  const int nti = (a.d_in + 15) >> 4, nto = (a.d_out + 15) >> 4;
  float4* wl = reinterpret_cast<float4*>(t_lds);                          // W^T image: [ot2 < nti][kk < nto][64] float4
  float* dzs = reinterpret_cast<float*>(t_lds) + (size_t)NTI * NTO * 256;  // [TROWS][LDO]
  float* xsg = dzs + TROWS * LDO;                                          // [TROWS][LDI]  raw x (x_hat is re-formed at each use)
  float* red = xsg + TROWS * LDI;                                          // [2 sums][4 row tiles][16*NTI] running column sums of gx
  float* xcol = red + 2 * 4 * 16 * NTI;                                    // [3][16*NTI] x_scale | x_shift | x_mean of my group
  float* ocol = xcol + 3 * 16 * NTI;                                       // [5][16*NTO] coef a | b | c | mask scale | mask shift
  constexpr int CI = 16 * NTI, CO = 16 * NTO;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, lr = lane & 15;
  const int rt = wave & 3, half = wave >> 2;
  const int grp = blockIdx.x / a.nblk, blk = blockIdx.x - grp * a.nblk;
  const int64_t nrounds = (a.R + TROWS - 1) / TROWS;
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
  if (want_dx) stage_weight<true>(wl, a.W, a.ldw, a.d_in, a.d_out, nti, nto);
  // dW accumulators of this wave: output tile `wave` (16 dz columns) x every operand tile; column constants of x_hat for my dW lanes
  f32x4 dw[NTI];
#pragma unroll
  for (int it = 0; it < NTI; ++it) dw[it] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < 16 * NTI; i += 64 * TW) {
    xcol[i] = (xs && i < a.d_in) ? xs[i] : 1.f;
    xcol[CI + i] = (xs && i < a.d_in) ? xt[i] : 0.f;
    xcol[2 * CI + i] = (xmu && i < a.d_in) ? xmu[i] : 0.f;
  }
  for (int i = threadIdx.x; i < 16 * NTO; i += 64 * TW) {     // (global / L1 round trips per tile and round cost ~20 us per launch)
    const bool in = i < a.d_out;
    ocol[i] = (cA && in) ? cA[i] : 1.f;
    ocol[CO + i] = (cA && in) ? cB[i] : 0.f;
    ocol[2 * CO + i] = (cA && in) ? cC[i] : 0.f;
    ocol[3 * CO + i] = (ms && in) ? ms[i] : 0.f;
    ocol[4 * CO + i] = (ms && in) ? mt[i] : 1.f;
  }
  float dbacc = 0.f;
  // running column sums of gx: one LDS slot per (row tile, column), owned by one lane of one wave (kept out of the register file:
  // with them the kernel spilled)
  for (int i = threadIdx.x; i < 2 * 4 * 16 * NTI; i += 64 * TW) red[i] = 0.f;

  // The raw rows of round r+1 are requested (into registers) right after round r's tiles are published, so the HBM latency runs
  // under the round's 256 MFMAs per wave; a wave loads the column tiles kk = half, half+2, ... of its 16 rows.
  constexpr int HO = (NTO + 1) / 2, HI = (NTI + 1) / 2;
  f32x4 pdy[HO], pz[HO], px[HI];
  bool pvalid = false;
  auto request = [&](int64_t round) {
    const int64_t row = round * TROWS + 16 * rt + lr;
    pvalid = row_ok(row, a.R, a.nvalid, a.K);
    const float* dyr = a.dy + (goff + row) * a.lddy;
    const float* zr = a.zo ? a.zo + (goff + row) * a.ldzo : nullptr;
    const float* xr = a.x + (goff + row) * a.ldx;
#pragma unroll
    for (int j = 0; j < HO; ++j) {
      const int c0 = 16 * (2 * j + half) + 4 * g;
      pdy[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      pz[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (pvalid && 2 * j + half < nto) {
        pdy[j] = ld4a(dyr, c0, a.d_out);
        if (zr) pz[j] = ld4a(zr, c0, a.d_out);
      }
    }
#pragma unroll
    for (int j = 0; j < HI; ++j) {
      px[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (pvalid && 2 * j + half < nti) px[j] = ld4a(xr, 16 * (2 * j + half) + 4 * g, a.d_in);
    }
  };
  request(r_lo);
  __syncthreads();          // the column constants (and the weight image) are published before phase 1 reads them
  for (int64_t round = r_lo; round < r_hi; ++round) {
    const int64_t row = round * TROWS + 16 * rt + lr;     // my row within the group (phases 1, 2)
    const bool valid = pvalid;
    // ---------------------------------------------------------------- phase 1: dz and x of the round -> LDS
    {
      float* dst = dzs + (16 * rt + lr) * LDO;
#pragma unroll
      for (int j = 0; j < HO; ++j) {
        const int kk = 2 * j + half;
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
            if (cA) {
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
        const int kk = 2 * j + half;
        if (kk < nti) *reinterpret_cast<float4*>(dsx + 16 * kk + 4 * g) = make_float4(px[j][0], px[j][1], px[j][2], px[j][3]);
      }
    }
    __syncthreads();
    if (round + 1 < r_hi && !(a.dbg & 4)) request(round + 1);
    // ---------------------------------------------------------------- phase 2: gx = (dz W) * mask, column sums
    if (want_dx && !(a.dbg & 1)) {
      f32x4 fr[NTO];
      const float* src = dzs + (16 * rt + lr) * LDO;
#pragma unroll
      for (int kk = 0; kk < NTO; ++kk) {
        fr[kk] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (kk < nto) { const float4 t = *reinterpret_cast<const float4*>(src + 16 * kk + 4 * g); fr[kk] = f32x4{t.x, t.y, t.z, t.w}; }
      }
      float* gr = a.gx + (goff + row) * a.ldgx;
      const float* xrow = xsg + (16 * rt + lr) * LDI;
#pragma unroll
      for (int j = 0; j < (NTI + 1) / 2; ++j) {
        const int ot = 2 * j + half;
        if (ot < nti) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          const float4* w0 = wl + (ot * nto) * 64 + lane;
#pragma unroll
          for (int kk = 0; kk < NTO; ++kk) {
            if (kk < nto) {
              const float4 p = w0[kk * 64];
              acc = mfma16(p.x, fr[kk][0], acc);
              acc = mfma16(p.y, fr[kk][1], acc);
              acc = mfma16(p.z, fr[kk][2], acc);
              acc = mfma16(p.w, fr[kk][3], acc);
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
          if (row < a.R) st4a(gr, c0, a.d_in, v);
          if (xmu) {
            const f32x4 mu = lds4(xcol + 2 * CI + c0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float a1 = t16_sum(v[r]), a2 = t16_sum(v[r] * (xv[r] - mu[r]));
              if (lr == 0) {
                red[(0 * 4 + rt) * 16 * NTI + c0 + r] += a1;
                red[(1 * 4 + rt) * 16 * NTI + c0 + r] += a2;
              }
            }
          }
        }
      }
    }
    // ---------------------------------------------------------------- phase 3: dW[tile `wave`] += dz^T x_hat over the round's rows
    if (wave < nto && a.dwp && !(a.dbg & 2)) {
#pragma unroll 1     // (unrolled by 2 the LDS reads of both steps are hoisted and the kernel spills)
      for (int q = 0; q < TROWS / 4; ++q) {
        const int rl = 4 * q + g;
        const float av = dzs[rl * LDO + 16 * wave + lr];
        dbacc += av;
        const float* xr = xsg + rl * LDI + lr;
#pragma unroll
        for (int it = 0; it < NTI; ++it) {
          if (it < nti) {
            float bv = xr[16 * it];
            if (xs) { bv = bv * xcol[16 * it + lr] + xcol[CI + 16 * it + lr]; if (a.xrelu) bv = fmaxf(bv, 0.f); }
            dw[it] = mfma16(av, bv, dw[it]);
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
  if (want_dx && xmu) {
    // the four row tiles' sums -> one partial per workgroup   (the last round's barrier has published them)
    float* S = a.sums + (int64_t)blockIdx.x * 2 * a.d_in;
    for (int i = threadIdx.x; i < 2 * a.d_in; i += 64 * TW) {
      const int w = i / a.d_in, c = i - w * a.d_in;
      const float* p = red + (w * 4) * 16 * NTI + c;
      S[i] = (p[0] + p[16 * NTI]) + (p[2 * 16 * NTI] + p[3 * 16 * NTI]);
    }
  }
}

// Column sums for a BatchNorm backward whose upstream gradient comes from somewhere else than k_tlin_bwd (the last BatchNorm of a
// stack: its output feeds an aggregation / a residual): sums[grp][blk][0][c] = sum g, [1][c] = sum g * (z - mu), g = dy * [ms*z + mt > 0].
__global__ __launch_bounds__(256) void k_tbn_bwd_sums(const float* __restrict__ dy, int lddy, const float* __restrict__ z, int ldz, int64_t R,
                                                      int G, int C, const int32_t* __restrict__ nvalid, int K, const float* __restrict__ st,
                                                      int relu, int nblk, float* __restrict__ sums) {
  const int grp = blockIdx.x / nblk, blk = blockIdx.x - grp * nblk;
  const int C4 = C >> 2, cg = threadIdx.x % C4, rg = threadIdx.x / C4, nrg = 256 / C4;      // C % 4 == 0, C <= 1024
  const int64_t r_lo = R * blk / nblk, r_hi = R * (blk + 1) / nblk;
  const float* s = st + (int64_t)grp * C;
  const int64_t GC = (int64_t)G * C;
  __shared__ float red[2][256][4];
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
    S[i] = acc;
  }
}

// Finish of a BatchNorm backward: from the column-sum partials of every group, the column constants of
//   dz = A*g - B - C*z,   A = gamma*rstd,  B = A*(m1 - m2*rstd*mu),  C = A*m2*rstd,   m1 = sum g / n,  m2 = rstd * sum g (z-mu) / n
// (coef[0..2][grp][C]) and the affine gradients d beta += sum g, d gamma += rstd * sum g (z - mu) summed over the groups (the
// reference applies ONE module to both sign passes).  One thread per column; partials added in block order (deterministic).
__global__ __launch_bounds__(256) void k_tbn_bwd_finish(const float* __restrict__ sums, int nblk, int G, int C, const float* __restrict__ st,
                                                        const float* __restrict__ cnt, const float* __restrict__ gamma, float* __restrict__ coef,
                                                        float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate) {
  // 16 columns x 16 lanes per block: a lane adds its slice of the block partials in order, then a fixed pairwise tree
  __shared__ float l1[16][17], l2[16][17];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  const int per = (nblk + 15) / 16, b0 = rl * per, b1 = b0 + per < nblk ? b0 + per : nblk;
  const int64_t GC = (int64_t)G * C;
  float dg = 0.f, db = 0.f;
  for (int grp = 0; grp < G; ++grp) {
    const float* S = sums + (int64_t)grp * nblk * 2 * C;
    float s1 = 0.f, s2 = 0.f;
    if (c < C)
      for (int b = b0; b < b1; ++b) { s1 += S[(int64_t)b * 2 * C + c]; s2 += S[(int64_t)b * 2 * C + C + c]; }
    l1[rl][cl] = s1; l2[rl][cl] = s2;
    __syncthreads();
    for (int step = 8; step >= 1; step >>= 1) {
      if (rl < step) {
        s1 += l1[rl + step][cl]; s2 += l2[rl + step][cl];
        l1[rl][cl] = s1; l2[rl][cl] = s2;
      }
      __syncthreads();
    }
    if (rl == 0 && c < C) {
      const float* s = st + (int64_t)grp * C;
      const float mu = s[c], rs = s[2 * GC + c], n = cnt[grp];
      const float A = (gamma ? gamma[c] : 1.f) * rs;
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
  if (rl == 0 && c < C) {
    if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + dg;
    if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + db;
  }
}

// out[i] (+)= sum_b part[b*stride + i]   (partials added in block order)
__global__ __launch_bounds__(256) void k_tsum_parts(const float* __restrict__ part, int nparts, int64_t stride, int64_t n,
                                                    float* __restrict__ out, int accumulate) {
  // 64 outputs per block, 4 lanes per output: lane q adds the partials b = q, q+4, ... in order (independent loads), then q = 0..3 in order
  __shared__ float red[4][64];
  const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + c;
  float acc = 0.f;
  if (i < n) {
    int b = q;
    for (; b + 12 < nparts; b += 16) {
      const float v0 = part[(int64_t)b * stride + i], v1 = part[(int64_t)(b + 4) * stride + i], v2 = part[(int64_t)(b + 8) * stride + i],
                  v3 = part[(int64_t)(b + 12) * stride + i];
      acc = (((acc + v0) + v1) + v2) + v3;
    }
    for (; b < nparts; b += 4) acc += part[(int64_t)b * stride + i];
  }
  red[q][c] = acc;
  __syncthreads();
  if (q == 0 && i < n) out[i] = (accumulate ? out[i] : 0.f) + (((red[0][c] + red[1][c]) + red[2][c]) + red[3][c]);
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
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return fail(SN_ERR_LAUNCH, "%s: cannot raise the dynamic LDS limit to %zu", who, lds);
  return SN_OK;
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace
}  // namespace sn

using namespace sn;

// workgroups per group of the forward link (= moment partials per group) / of the backward link (= dW, column-sum partials per group)
extern "C" int sn_train_linear_blocks(int64_t R, int G) {
  if (G < 1) G = 1;
  const int64_t want = cdiv(cdiv(R > 0 ? R : 1, 16), 4);
  int64_t cap = train_cus() / G;
  if (cap < 1) cap = 1;
  return (int)(want < cap ? want : cap);
}
extern "C" int sn_train_linear_bwd_blocks(int64_t R, int G) {
  if (G < 1) G = 1;
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
  if (p.R == 0) return SN_OK;
  const int nblk = sn_train_linear_blocks(p.R, p.G);
  TLin a{p.x, p.ldx, p.R, p.G, p.d_in, p.d_out, p.W, p.ldw, p.bias, p.nvalid, p.K, p.in_scale, p.in_shift, p.in_relu, p.out_relu,
         p.y, p.ldy, p.stat_part, nblk};
  const int nti = (p.d_in + 15) / 16, nto = (p.d_out + 15) / 16;
  const size_t lds = (size_t)8 * 8 * 1024 + (size_t)(2 * 16 * 8 + TW * 16 * 8) * sizeof(float);    // weight image + in_scale | in_shift + pivots
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (p.stat_part) {
    if ((rc = raise_lds(k_tlin_fwd<8, 8, true>, lds, "sn_train_linear_f32")) != SN_OK) return rc;
    hipLaunchKernelGGL((k_tlin_fwd<8, 8, true>), dim3((unsigned)(nblk * p.G)), dim3(64 * TW), lds, st, a);
  } else {
    if ((rc = raise_lds(k_tlin_fwd<8, 8, false>, lds, "sn_train_linear_f32")) != SN_OK) return rc;
    hipLaunchKernelGGL((k_tlin_fwd<8, 8, false>), dim3((unsigned)(nblk * p.G)), dim3(64 * TW), lds, st, a);
  }
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
  SN_REQUIRE(!p.nvalid || p.K > 0, "sn_train_linear_bwd_f32: nvalid needs K > 0");
  for (const float* v : {p.coef_a, p.coef_b, p.coef_c, p.mask_scale, p.mask_shift, p.x_scale, p.x_shift, p.x_mean})
    SN_REQUIRE(!v || al16(v), "sn_train_linear_bwd_f32: column vectors must be 16-byte aligned");
  if (p.R == 0) return SN_OK;
  const int nblk = sn_train_linear_bwd_blocks(p.R, p.G);
  TBwd a{p.R, p.G, p.nvalid, p.K, p.d_in, p.d_out, p.dy, p.lddy, p.zo, p.ldzo, p.coef_a, p.coef_b, p.coef_c, p.mask_scale, p.mask_shift,
         p.x, p.ldx, p.x_scale, p.x_shift, p.x_relu, p.x_mean, p.W, p.ldw, p.gx, p.ldgx, p.sums_part, p.dw_part, p.want_db, nblk,
         getenv("SN_TRAIN_DBG") ? atoi(getenv("SN_TRAIN_DBG")) : 0};
  constexpr size_t lds = (size_t)8 * 8 * 1024 + (size_t)TROWS * (stage_ld(8) + stage_ld(8)) * sizeof(float) + (size_t)(2 * 4 + 3 + 5) * 16 * 8 * sizeof(float);
  int rc;
  if ((rc = raise_lds(k_tlin_bwd<8, 8>, lds, "sn_train_linear_bwd_f32")) != SN_OK) return rc;
  hipLaunchKernelGGL((k_tlin_bwd<8, 8>), dim3((unsigned)(nblk * p.G)), dim3(64 * TW), lds, (hipStream_t)stream, a);
  SN_CHECK_LAUNCH("sn_train_linear_bwd_f32");
  return SN_OK;
}

extern "C" int sn_train_bn_bwd_blocks(int64_t R, int G) {
  if (G < 1) G = 1;
  const int64_t want = cdiv(R > 0 ? R : 1, 128);
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
                     relu, nblk, sums_part);
  SN_CHECK_LAUNCH("sn_train_bn_bwd_sums_f32");
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

extern "C" int sn_train_reduce_parts_f32(const float* part, int nparts, int64_t stride, int64_t n, float* out, int accumulate, void* stream) {
  SN_REQUIRE(part && out && nparts >= 1 && n >= 0 && stride >= n, "sn_train_reduce_parts_f32: bad arguments");
  if (n == 0) return SN_OK;
  hipLaunchKernelGGL(k_tsum_parts, dim3((unsigned)cdiv(n, 64)), dim3(256), 0, (hipStream_t)stream, part, nparts, stride, n, out, accumulate);
  SN_CHECK_LAUNCH("sn_train_reduce_parts_f32");
  return SN_OK;
}
