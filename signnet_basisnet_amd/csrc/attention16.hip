// attention16.hip — per-node set attention over K <= 16 eigenvector slots on the fp32 matrix pipe (forward and adjoint).
//
// The layer-at-a-time / training path's ScaledDotProductAttention (transformer_module.py:44-58 inside MultiHeadAttention :60-102) for
// the BASELINE "k = 16" configs: per (node, head) S = (q/sqrt(dk)) k^T over the node's valid slots, P = softmax(S) [* dropout mask],
// o = P v.  One WAVE per (node, head): the 16 x 16 score tile is one accumulator of v_mfma_f32_16x16x4_f32, and with the operand roles
// chosen below no register transpose is needed in the forward:
//   S^T  = K Q^T      A[i = key][k] = k[key][c],  B[k][j = query] = q[query][c]   -> lane (lr, g) holds S[query = lr][key = 4g + r]
//   O    = P V        A[i = query][k = key 4g + s] = P[lr][4g + s] (its own register s),  B[key 4g + s][j = c] = v[key][c]
// (the k index of an MFMA step may be any permutation as long as both operands use the same one: lane group g supplies channel
//  (dk/4) g + s in S^T, and key 4g + s in P V).  The adjoint needs P^T and dS^T for dV and dK: two 16 x 16 transposes through LDS.
// The scalar kernels they replace (ops.hip k_set_attention, backward.hip k_set_attention_bwd) took 120 us / 250 us for the headline
// batch (2 950 nodes x 4 heads); they remain for K > 16 and head widths that are not a multiple of 16.
#include "common.hpp"

namespace sn {
namespace {

template <int DK>
struct Rows {            // a [16, DK] head slice of a [N, K, D] tensor: lane (lr, g) keeps channels (DK/4) g .. (DK/4) g + DK/4 - 1 of row lr
  float v[DK / 4];
  __device__ __forceinline__ void load(const float* __restrict__ base, int D, int lr, int g, bool live) {
#pragma unroll
    for (int i = 0; i < DK / 4; ++i) v[i] = 0.f;
    if (live) {
      const float* p = base + (int64_t)lr * D + (DK / 4) * g;
#pragma unroll
      for (int i = 0; i < DK / 16; ++i) {
        const float4 t = *reinterpret_cast<const float4*>(p + 4 * i);
        v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
      }
    }
  }
};

// D[key 4g+r][query lr] = sum_c a[key][c] b[query][c]
template <int DK>
__device__ __forceinline__ f32x4 score_t(const Rows<DK>& a_keys, const Rows<DK>& b_queries) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < DK / 4; ++s) acc = mfma16(a_keys.v[s], b_queries.v[s], acc);
  return acc;
}
__device__ __forceinline__ float gmax(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float gsum(float v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); }

// softmax over the valid keys of query lr; p[r] = P[query lr][key 4g + r] (0 for invalid keys / queries)
__device__ __forceinline__ f32x4 softmax_keys(f32x4 s, int kv, int lr, int g) {
  float m = -INFINITY;
#pragma unroll
  for (int r = 0; r < 4; ++r) if (4 * g + r < kv) m = fmaxf(m, s[r]);
  m = gmax(m);
  f32x4 e;
  float z = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) { e[r] = (4 * g + r < kv) ? expf(s[r] - m) : 0.f; z += e[r]; }
  z = gsum(z);
  const float inv = (lr < kv && z > 0.f) ? 1.0f / z : 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) e[r] *= inv;
  return e;
}
// out[row 4g + r][c = 16 t + lr] = sum_{key 4g'+s} a[row? ...]: D[i = lr-row of A][j = column lr of B]; A from registers areg[s] (k-slot 4g+s),
// B[k = row 4g + s][j = 16 t + lr] read from global `bmat` (rows of a [16, DK] slice)
template <int DK>
__device__ __forceinline__ void times_rows(const f32x4 areg, const float* __restrict__ bmat, int D, int nrows, int lr, int g, f32x4 (&out)[DK / 16]) {
#pragma unroll
  for (int t = 0; t < DK / 16; ++t) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int row = 4 * g + s;
      const float b = row < nrows ? bmat[(int64_t)row * D + 16 * t + lr] : 0.f;
      acc = mfma16(areg[s], b, acc);
    }
    out[t] = acc;
  }
}
template <int DK>
__device__ __forceinline__ void store_rows(float* __restrict__ dst, int D, int K, int lr, int g, const f32x4 (&val)[DK / 16], float scale) {
#pragma unroll
  for (int t = 0; t < DK / 16; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (4 * g + r < K) dst[(int64_t)(4 * g + r) * D + 16 * t + lr] = val[t][r] * scale;
}

template <int DK>
__global__ __launch_bounds__(256) void k_attn16_fwd(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                    int64_t NH, int K, int H, const int32_t* __restrict__ nvalid,
                                                    const float* __restrict__ pmask, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, lr = lane & 15, g = lane >> 4;
  const int64_t nh = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (nh >= NH) return;
  const int64_t node = nh / H;
  const int h = (int)(nh - node * H);
  const int D = H * DK;
  const int kv = nvalid ? (nvalid[node] < K ? nvalid[node] : K) : K;
  const int64_t base = node * K * D + (int64_t)h * DK;
  Rows<DK> rq, rk;
  rq.load(q + base, D, lr, g, lr < kv);
  rk.load(k + base, D, lr, g, lr < kv);
  f32x4 s = score_t<DK>(rk, rq);
  const float inv_t = 1.0f / sqrtf((float)DK);
#pragma unroll
  for (int r = 0; r < 4; ++r) s[r] *= inv_t;
  f32x4 p = softmax_keys(s, kv, lr, g);
  if (pmask && lr < K) {
    const float* pm = pmask + (nh * K + lr) * K;
#pragma unroll
    for (int r = 0; r < 4; ++r) if (4 * g + r < K) p[r] *= pm[4 * g + r];
  }
  f32x4 o[DK / 16];
  times_rows<DK>(p, v + base, D, kv, lr, g, o);
  store_rows<DK>(out + base, D, K, lr, g, o, 1.0f);
}

template <int DK>
__global__ __launch_bounds__(256) void k_attn16_bwd(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                    const float* __restrict__ dout, int64_t NH, int K, int H,
                                                    const int32_t* __restrict__ nvalid, const float* __restrict__ pmask,
                                                    float* __restrict__ dq, float* __restrict__ dkk, float* __restrict__ dv) {
  __shared__ float tile[4][2][16][17];
  const int lane = threadIdx.x & 63, lr = lane & 15, g = lane >> 4, w = threadIdx.x >> 6;
  const int64_t nh = (int64_t)blockIdx.x * 4 + w;
  if (nh >= NH) return;
  const int64_t node = nh / H;
  const int h = (int)(nh - node * H);
  const int D = H * DK;
  const int kv = nvalid ? (nvalid[node] < K ? nvalid[node] : K) : K;
  const int64_t base = node * K * D + (int64_t)h * DK;
  const float inv_t = 1.0f / sqrtf((float)DK);
  Rows<DK> rq, rk, rv, rg;
  rq.load(q + base, D, lr, g, lr < kv);
  rk.load(k + base, D, lr, g, lr < kv);
  rv.load(v + base, D, lr, g, lr < kv);
  rg.load(dout + base, D, lr, g, lr < kv);
  f32x4 s = score_t<DK>(rk, rq);
#pragma unroll
  for (int r = 0; r < 4; ++r) s[r] *= inv_t;
  const f32x4 p = softmax_keys(s, kv, lr, g);                  // P[query lr][key 4g + r]
  f32x4 dp = score_t<DK>(rv, rg);                              // dP'[query lr][key 4g + r] = sum_c dO[query][c] v[key][c]
  f32x4 pm = {1.f, 1.f, 1.f, 1.f};
  if (pmask && lr < K) {
    const float* m = pmask + (nh * K + lr) * K;
#pragma unroll
    for (int r = 0; r < 4; ++r) if (4 * g + r < K) pm[r] = m[4 * g + r];
  }
  float dot = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) { dp[r] *= pm[r]; dot += p[r] * dp[r]; }
  dot = gsum(dot);
  f32x4 ds, pd;
#pragma unroll
  for (int r = 0; r < 4; ++r) { ds[r] = p[r] * (dp[r] - dot); pd[r] = p[r] * pm[r]; }
  // dQ[query][c] = sum_key dS[query][key] k[key][c] / sqrt(dk)
  f32x4 o[DK / 16];
  times_rows<DK>(ds, k + base, D, kv, lr, g, o);
  store_rows<DK>(dq + base, D, K, lr, g, o, inv_t);
  // transposes: lane (lr = key, g) needs X[query 4g + s][key lr]
  float (*T0)[17] = tile[w][0];
  float (*T1)[17] = tile[w][1];
#pragma unroll
  for (int r = 0; r < 4; ++r) { T0[lr][4 * g + r] = pd[r]; T1[lr][4 * g + r] = ds[r]; }
  __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): the wave's own LDS writes have landed (single-wave tile: no barrier)
  f32x4 pt, dst;
#pragma unroll
  for (int r = 0; r < 4; ++r) { pt[r] = T0[4 * g + r][lr]; dst[r] = T1[4 * g + r][lr]; }
  // dV[key][c] = sum_query P'[query][key] dO[query][c];   dK[key][c] = sum_query dS[query][key] q[query][c] / sqrt(dk)
  times_rows<DK>(pt, dout + base, D, kv, lr, g, o);
  store_rows<DK>(dv + base, D, K, lr, g, o, 1.0f);
  times_rows<DK>(dst, q + base, D, kv, lr, g, o);
  store_rows<DK>(dkk + base, D, K, lr, g, o, inv_t);
}

inline bool a16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// returns true when the shape takes the matrix-pipe path (the caller falls back to the scalar kernels otherwise)
bool attention16_forward(const float* q, const float* k, const float* v, int64_t N, int K, int heads, int dk, const int32_t* nvalid,
                         const float* prob_mask, float* out, hipStream_t st) {
  if (K > 16 || (dk != 16 && dk != 32 && dk != 64) || !a16(q) || !a16(k) || !a16(v)) return false;
  const int64_t NH = N * heads;
  const dim3 grid((unsigned)cdiv(NH, 4));
  if (dk == 16) hipLaunchKernelGGL(k_attn16_fwd<16>, grid, dim3(256), 0, st, q, k, v, NH, K, heads, nvalid, prob_mask, out);
  else if (dk == 32) hipLaunchKernelGGL(k_attn16_fwd<32>, grid, dim3(256), 0, st, q, k, v, NH, K, heads, nvalid, prob_mask, out);
  else hipLaunchKernelGGL(k_attn16_fwd<64>, grid, dim3(256), 0, st, q, k, v, NH, K, heads, nvalid, prob_mask, out);
  return true;
}
bool attention16_backward(const float* q, const float* k, const float* v, const float* dout, int64_t N, int K, int heads, int dk,
                          const int32_t* nvalid, const float* prob_mask, float* dq, float* dkk, float* dv, hipStream_t st) {
  if (K > 16 || (dk != 16 && dk != 32 && dk != 64) || !a16(q) || !a16(k) || !a16(v) || !a16(dout)) return false;
  const int64_t NH = N * heads;
  const dim3 grid((unsigned)cdiv(NH, 4));
  if (dk == 16) hipLaunchKernelGGL(k_attn16_bwd<16>, grid, dim3(256), 0, st, q, k, v, dout, NH, K, heads, nvalid, prob_mask, dq, dkk, dv);
  else if (dk == 32) hipLaunchKernelGGL(k_attn16_bwd<32>, grid, dim3(256), 0, st, q, k, v, dout, NH, K, heads, nvalid, prob_mask, dq, dkk, dv);
  else hipLaunchKernelGGL(k_attn16_bwd<64>, grid, dim3(256), 0, st, q, k, v, dout, NH, K, heads, nvalid, prob_mask, dq, dkk, dv);
  return true;
}

}  // namespace sn
