// dense_attention.hip — multi-head softmax attention over a whole sequence, forward and backward.
//
// Replaces the attention inside torch.nn.TransformerEncoderLayer as LearningFilters/models.py:115-135 (`Transformer`) uses it:
// nn.MultiheadAttention(batch_first) on sequences of the graph's N = 1024 nodes, 4 heads of width 3-8 — scores softmax(Q K^T / sqrt(dk)) V
// per head.  The head width is tiny, so this is not an MFMA problem: one thread per (sequence, head, query) keeps its query row and the
// running output in registers and streams the keys / values of its (sequence, head) through LDS tiles (every thread of the workgroup
// reads the same LDS word at the same time: broadcast, no bank conflicts), with an online softmax (running maximum and sum).  The
// forward also writes the row's log-sum-exp, from which the backward recomputes the probabilities:
//   dV_j = sum_i p_ij dO_i,  dS_ij = p_ij (dO_i . V_j - delta_i), delta_i = dO_i . O_i,  dQ_i = scale sum_j dS_ij K_j,  dK_j = scale sum_i dS_ij Q_i
// in two passes (one thread per query for dQ, one thread per key for dK / dV): no atomics, reproducible.
#include "common.hpp"

namespace sn {

constexpr int DA_T = 128;     // keys (or queries) per LDS tile
constexpr int DA_MAXD = 32;   // head width limit (registers per thread)

// q, k, v, o: [Bt, L, H*dk] row-major; lse: [Bt, H, L].  grid (ceil(L/256), H, Bt), 256 threads.
template <int DK>
__global__ __launch_bounds__(256) void k_dense_attn_fwd(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                        int L, int H, int dk, float scale, float* __restrict__ o, float* __restrict__ lse) {
  __shared__ float Ks[DA_T][DK + 1], Vs[DA_T][DK + 1];
  const int h = blockIdx.y, b = blockIdx.z, i = blockIdx.x * 256 + threadIdx.x;
  const int d = H * dk;
  const float* base_q = q + ((int64_t)b * L) * d + h * dk;
  const float* base_k = k + ((int64_t)b * L) * d + h * dk;
  const float* base_v = v + ((int64_t)b * L) * d + h * dk;
  float qr[DK], acc[DK];
#pragma unroll
  for (int c = 0; c < DK; ++c) { qr[c] = (i < L && c < dk) ? base_q[(int64_t)i * d + c] * scale : 0.f; acc[c] = 0.f; }
  float m = -INFINITY, l = 0.f;
  for (int j0 = 0; j0 < L; j0 += DA_T) {
    __syncthreads();
    for (int e = threadIdx.x; e < DA_T * dk; e += 256) {
      const int jj = e / dk, c = e - jj * dk;
      const bool ok = j0 + jj < L;
      Ks[jj][c] = ok ? base_k[(int64_t)(j0 + jj) * d + c] : 0.f;
      Vs[jj][c] = ok ? base_v[(int64_t)(j0 + jj) * d + c] : 0.f;
    }
    __syncthreads();
    const int nj = L - j0 < DA_T ? L - j0 : DA_T;
    for (int jj = 0; jj < nj; ++jj) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < DK; ++c)
        if (c < dk) s += qr[c] * Ks[jj][c];
      const float mn = fmaxf(m, s);
      const float corr = expf(m - mn), p = expf(s - mn);
      l = l * corr + p;
#pragma unroll
      for (int c = 0; c < DK; ++c)
        if (c < dk) acc[c] = acc[c] * corr + p * Vs[jj][c];
      m = mn;
    }
  }
  if (i < L) {
    const float r = 1.0f / l;
    float* orow = o + ((int64_t)b * L + i) * d + h * dk;
#pragma unroll
    for (int c = 0; c < DK; ++c)
      if (c < dk) orow[c] = acc[c] * r;
    lse[((int64_t)b * H + h) * L + i] = m + logf(l);
  }
}

// dQ (one thread per query) and delta_i = dO_i . O_i (written for the second pass)
template <int DK>
__global__ __launch_bounds__(256) void k_dense_attn_bwd_q(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                          const float* __restrict__ o, const float* __restrict__ dout,
                                                          const float* __restrict__ lse, int L, int H, int dk, float scale,
                                                          float* __restrict__ dq, float* __restrict__ delta) {
  __shared__ float Ks[DA_T][DK + 1], Vs[DA_T][DK + 1];
  const int h = blockIdx.y, b = blockIdx.z, i = blockIdx.x * 256 + threadIdx.x;
  const int d = H * dk;
  const int64_t row = ((int64_t)b * L + i) * d + h * dk;
  const float* base_k = k + ((int64_t)b * L) * d + h * dk;
  const float* base_v = v + ((int64_t)b * L) * d + h * dk;
  float qr[DK], go[DK], acc[DK];
  float dl = 0.f;
#pragma unroll
  for (int c = 0; c < DK; ++c) {
    const bool ok = i < L && c < dk;
    qr[c] = ok ? q[row + c] * scale : 0.f;
    go[c] = ok ? dout[row + c] : 0.f;
    dl += ok ? go[c] * o[row + c] : 0.f;
    acc[c] = 0.f;
  }
  const float li = i < L ? lse[((int64_t)b * H + h) * L + i] : 0.f;
  for (int j0 = 0; j0 < L; j0 += DA_T) {
    __syncthreads();
    for (int e = threadIdx.x; e < DA_T * dk; e += 256) {
      const int jj = e / dk, c = e - jj * dk;
      const bool ok = j0 + jj < L;
      Ks[jj][c] = ok ? base_k[(int64_t)(j0 + jj) * d + c] : 0.f;
      Vs[jj][c] = ok ? base_v[(int64_t)(j0 + jj) * d + c] : 0.f;
    }
    __syncthreads();
    const int nj = L - j0 < DA_T ? L - j0 : DA_T;
    for (int jj = 0; jj < nj; ++jj) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int c = 0; c < DK; ++c)
        if (c < dk) { s += qr[c] * Ks[jj][c]; dp += go[c] * Vs[jj][c]; }
      const float ds = expf(s - li) * (dp - dl);
#pragma unroll
      for (int c = 0; c < DK; ++c)
        if (c < dk) acc[c] += ds * Ks[jj][c];
    }
  }
  if (i < L) {
#pragma unroll
    for (int c = 0; c < DK; ++c)
      if (c < dk) dq[row + c] = acc[c] * scale;
    delta[((int64_t)b * H + h) * L + i] = dl;
  }
}

// dK, dV (one thread per key): streams the queries, their dO rows, lse and delta through LDS
template <int DK>
__global__ __launch_bounds__(256) void k_dense_attn_bwd_kv(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                           const float* __restrict__ dout, const float* __restrict__ lse,
                                                           const float* __restrict__ delta, int L, int H, int dk, float scale,
                                                           float* __restrict__ dkk, float* __restrict__ dv) {
  __shared__ float Qs[DA_T][DK + 1], Gs[DA_T][DK + 1], Ls[DA_T], Ds[DA_T];
  const int h = blockIdx.y, b = blockIdx.z, j = blockIdx.x * 256 + threadIdx.x;
  const int d = H * dk;
  const int64_t row = ((int64_t)b * L + j) * d + h * dk;
  const float* base_q = q + ((int64_t)b * L) * d + h * dk;
  const float* base_g = dout + ((int64_t)b * L) * d + h * dk;
  float kr[DK], vr[DK], ak[DK], av[DK];
#pragma unroll
  for (int c = 0; c < DK; ++c) {
    const bool ok = j < L && c < dk;
    kr[c] = ok ? k[row + c] : 0.f;
    vr[c] = ok ? v[row + c] : 0.f;
    ak[c] = 0.f; av[c] = 0.f;
  }
  for (int i0 = 0; i0 < L; i0 += DA_T) {
    __syncthreads();
    for (int e = threadIdx.x; e < DA_T * dk; e += 256) {
      const int ii = e / dk, c = e - ii * dk;
      const bool ok = i0 + ii < L;
      Qs[ii][c] = ok ? base_q[(int64_t)(i0 + ii) * d + c] * scale : 0.f;
      Gs[ii][c] = ok ? base_g[(int64_t)(i0 + ii) * d + c] : 0.f;
    }
    for (int e = threadIdx.x; e < DA_T; e += 256) {
      const bool ok = i0 + e < L;
      Ls[e] = ok ? lse[((int64_t)b * H + h) * L + i0 + e] : 0.f;
      Ds[e] = ok ? delta[((int64_t)b * H + h) * L + i0 + e] : 0.f;
    }
    __syncthreads();
    const int ni = L - i0 < DA_T ? L - i0 : DA_T;
    for (int ii = 0; ii < ni; ++ii) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int c = 0; c < DK; ++c)
        if (c < dk) { s += Qs[ii][c] * kr[c]; dp += Gs[ii][c] * vr[c]; }
      const float p = expf(s - Ls[ii]);
      const float ds = p * (dp - Ds[ii]);
#pragma unroll
      for (int c = 0; c < DK; ++c)
        if (c < dk) { av[c] += p * Gs[ii][c]; ak[c] += ds * Qs[ii][c]; }       // Qs already carries the 1/sqrt(dk) factor
    }
  }
  if (j < L) {
#pragma unroll
    for (int c = 0; c < DK; ++c)
      if (c < dk) { dkk[row + c] = ak[c]; dv[row + c] = av[c]; }
  }
}

}  // namespace sn

using namespace sn;

// the head width rounded up to a compiled register-array size
#define SN_DA_DISPATCH(dk, ...)                                     \
  do {                                                              \
    if ((dk) <= 4) { constexpr int DKT = 4; __VA_ARGS__; }          \
    else if ((dk) <= 8) { constexpr int DKT = 8; __VA_ARGS__; }     \
    else if ((dk) <= 16) { constexpr int DKT = 16; __VA_ARGS__; }   \
    else { constexpr int DKT = 32; __VA_ARGS__; }                   \
  } while (0)

extern "C" int sn_dense_attention_f32(const float* q, const float* k, const float* v, int64_t Bt, int L, int heads, int dk, float* out,
                                      float* lse, void* stream) {
  SN_REQUIRE(q && k && v && out && lse && Bt >= 0 && L >= 1 && heads >= 1, "sn_dense_attention_f32: bad arguments");
  SN_REQUIRE(dk >= 1 && dk <= DA_MAXD, "sn_dense_attention_f32: head width %d not in [1, %d]", dk, DA_MAXD);
  SN_REQUIRE(Bt <= 65535 && heads <= 65535, "sn_dense_attention_f32: too many sequences / heads for one launch");
  if (Bt == 0) return SN_OK;
  const dim3 grid((unsigned)cdiv(L, 256), (unsigned)heads, (unsigned)Bt);
  const float scale = 1.0f / sqrtf((float)dk);
  SN_DA_DISPATCH(dk, hipLaunchKernelGGL(k_dense_attn_fwd<DKT>, grid, dim3(256), 0, (hipStream_t)stream, q, k, v, L, heads, dk, scale, out, lse));
  SN_CHECK_LAUNCH("sn_dense_attention_f32");
  return SN_OK;
}

extern "C" int sn_dense_attention_bwd_f32(const float* q, const float* k, const float* v, const float* out, const float* lse,
                                          const float* dout, int64_t Bt, int L, int heads, int dk, float* dq, float* dk_out, float* dv,
                                          float* delta, void* stream) {
  SN_REQUIRE(q && k && v && out && lse && dout && dq && dk_out && dv && delta && Bt >= 0 && L >= 1 && heads >= 1,
             "sn_dense_attention_bwd_f32: bad arguments");
  SN_REQUIRE(dk >= 1 && dk <= DA_MAXD && Bt <= 65535 && heads <= 65535, "sn_dense_attention_bwd_f32: bad sizes");
  if (Bt == 0) return SN_OK;
  const float scale = 1.0f / sqrtf((float)dk);
  const dim3 grid((unsigned)cdiv(L, 256), (unsigned)heads, (unsigned)Bt);
  SN_DA_DISPATCH(dk, hipLaunchKernelGGL(k_dense_attn_bwd_q<DKT>, grid, dim3(256), 0, (hipStream_t)stream, q, k, v, out, dout, lse, L, heads, dk,
                                        scale, dq, delta));
  SN_DA_DISPATCH(dk, hipLaunchKernelGGL(k_dense_attn_bwd_kv<DKT>, grid, dim3(256), 0, (hipStream_t)stream, q, k, v, dout, lse, delta, L, heads,
                                        dk, scale, dk_out, dv));
  SN_CHECK_LAUNCH("sn_dense_attention_bwd_f32");
  return SN_OK;
}
