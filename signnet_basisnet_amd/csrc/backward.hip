// backward.hip — gradients of the layer-at-a-time train path (backward + optimiser step).
//
// The reference gets these from torch.autograd over ATen / PyG / torch_scatter kernels (loss.backward() at
// Alchemy/main_alchemy.py:108, GINESignNetPyG/core/train.py:62-63).  Here every forward entry point of
// signnet_hip.h has its hand-written adjoint; the Python side (signnet_basisnet_amd/autograd.py) only wires them
// into torch.autograd.Function objects.  Same conventions as the forward: fp32 row matrices, a row r = node*K +
// slot is valid iff slot < nvalid[node], invalid rows carry zero gradient.
#include "common.hpp"

namespace sn {
namespace {

// rows are < 2^31 everywhere (checked on the host): 32-bit division
__device__ __forceinline__ bool row_ok(const int32_t* __restrict__ nvalid, int K, int64_t r) {
  if (!nvalid) return true;
  const unsigned node = (unsigned)r / (unsigned)K;
  return (int)((unsigned)r - node * (unsigned)K) < nvalid[node];
}
__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ============================================================================ dW = dy^T x, db = sum dy   (fp32 MFMA)
// grid (row chunks, groups of 64 outputs, groups of 128 inputs); 4 waves, wave w owns output tile 4*og + w and all
// 8 input tiles of the group: acc[t][r] = dW[16(4og+w) + 4(l>>4) + r][128 ig + 16 t + (l&15)].
constexpr int WG_OC = 64, WG_IC = 128, WG_LDO = WG_OC + 16, WG_LDI = WG_IC + 16;
__global__ __launch_bounds__(256) void k_wgrad(const float* __restrict__ x, int ldx, const float* __restrict__ dy, int ldy,
                                               int64_t R, int d_in, int d_out, const int32_t* __restrict__ nvalid, int K,
                                               int64_t rows_per_block, float* __restrict__ part /* [chunk][pstride] */,
                                               int64_t pstride, int want_b /* bias partial at part[chunk][d_out*d_in ..] */, int vec) {
  __shared__ __attribute__((aligned(16))) float xs[16 * WG_LDI];
  __shared__ __attribute__((aligned(16))) float ds[16 * WG_LDO];
  __shared__ int okrow[16];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int oc0 = blockIdx.y * WG_OC, ic0 = blockIdx.z * WG_IC;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = (r0 + rows_per_block < R) ? r0 + rows_per_block : R;
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  for (int64_t rb = r0; rb < r1; rb += 16) {
    __syncthreads();
    if (t < 16) okrow[t] = (rb + t < r1 && row_ok(nvalid, K, rb + t)) ? 1 : 0;
    __syncthreads();
    if (vec) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {                       // 16 rows x 32 float4
        const int rr = (t >> 5) + 8 * j, c = (t & 31) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (okrow[rr] && ic0 + c < d_in) v = *reinterpret_cast<const float4*>(x + (rb + rr) * ldx + ic0 + c);
        *reinterpret_cast<float4*>(&xs[rr * WG_LDI + c]) = v;
      }
      {
        const int rr = t >> 4, c = (t & 15) * 4;          // 16 rows x 16 float4
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (okrow[rr] && oc0 + c < d_out) v = *reinterpret_cast<const float4*>(dy + (rb + rr) * ldy + oc0 + c);
        *reinterpret_cast<float4*>(&ds[rr * WG_LDO + c]) = v;
      }
    } else {
      for (int i = t; i < 16 * WG_IC; i += 256) {
        const int rr = i / WG_IC, c = i - rr * WG_IC;
        xs[rr * WG_LDI + c] = (okrow[rr] && ic0 + c < d_in) ? x[(rb + rr) * ldx + ic0 + c] : 0.f;
      }
      for (int i = t; i < 16 * WG_OC; i += 256) {
        const int rr = i / WG_OC, c = i - rr * WG_OC;
        ds[rr * WG_LDO + c] = (okrow[rr] && oc0 + c < d_out) ? dy[(rb + rr) * ldy + oc0 + c] : 0.f;
      }
    }
    __syncthreads();
    if (t < WG_OC) {
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) bsum += ds[rr * WG_LDO + t];
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int row = 4 * kk + (lane >> 4);
      const float a = ds[row * WG_LDO + 16 * w + (lane & 15)];
#pragma unroll
      for (int tt = 0; tt < 8; ++tt) acc[tt] = mfma16(a, xs[row * WG_LDI + 16 * tt + (lane & 15)], acc[tt]);
    }
  }
  float* p = part + (int64_t)blockIdx.x * pstride;
#pragma unroll
  for (int tt = 0; tt < 8; ++tt) {
    const int ic = ic0 + 16 * tt + (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int oc = oc0 + 16 * w + 4 * (lane >> 4) + r;
      if (oc < d_out && ic < d_in) p[(int64_t)oc * d_in + ic] = acc[tt][r];
    }
  }
  if (want_b && blockIdx.z == 0 && t < WG_OC && oc0 + t < d_out) p[(int64_t)d_out * d_in + oc0 + t] = bsum;
}
// out[i] = sum_b part[b][i]   (blockIdx.y splits the partials; four independent accumulators keep loads in flight)
__global__ __launch_bounds__(256) void k_sum_parts(const float* __restrict__ part, int nblk, int64_t n, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int per = (nblk + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = (b0 + per < nblk) ? b0 + per : nblk;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int b = b0;
  for (; b + 3 < b1; b += 4) {
    s0 += part[(int64_t)b * n + i];
    s1 += part[(int64_t)(b + 1) * n + i];
    s2 += part[(int64_t)(b + 2) * n + i];
    s3 += part[(int64_t)(b + 3) * n + i];
  }
  for (; b < b1; ++b) s0 += part[(int64_t)b * n + i];
  out[(int64_t)blockIdx.y * n + i] = (s0 + s1) + (s2 + s3);
}
// out[i] = sum_b part[b*stride + i]  (single stage; the fallback when dW / db are not one buffer)
__global__ __launch_bounds__(256) void k_sum_strided(const float* __restrict__ part, int nblk, int64_t stride, int64_t n,
                                                     float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int b = 0; b < nblk; ++b) s += part[(int64_t)b * stride + i];
  out[i] = s;
}
// deterministic two-stage reduction of nblk partial vectors of n floats; tmp: float[16 * n] when nblk > 32
static inline void sum_parts(const float* part, int nblk, int64_t n, float* out, float* tmp, hipStream_t st) {
  const dim3 blk(256);
  const unsigned gx = (unsigned)((n + 255) / 256);
  if (nblk > 32 && tmp != nullptr) {
    hipLaunchKernelGGL(k_sum_parts, dim3(gx, 16), blk, 0, st, part, nblk, n, tmp);
    hipLaunchKernelGGL(k_sum_parts, dim3(gx, 1), blk, 0, st, (const float*)tmp, 16, n, out);
  } else {
    hipLaunchKernelGGL(k_sum_parts, dim3(gx, 1), blk, 0, st, part, nblk, n, out);
  }
}

// ============================================================================ train-mode BatchNorm (+ReLU) backward
// forward: a = scale[c]*z + shift[c] (scale = gamma*rstd, shift = beta - mean*scale), y = relu?(a) (+ residual).
// g = dy * [a > 0];  s1 = sum g, s2 = sum g*xhat (xhat = (z-mean)*rstd);  dz = scale*(g - s1/M - xhat*s2/M).
__global__ __launch_bounds__(256) void k_bn_bwd_partial(const float* __restrict__ z, int ldz, const float* __restrict__ dy,
                                                        int ldd, int64_t R, int C, const int32_t* __restrict__ nvalid, int K,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        int relu, int64_t rows_per_block, float* __restrict__ part) {
  __shared__ float red[2][4][64];
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = (r0 + rows_per_block < R) ? r0 + rows_per_block : R;
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int c = c0 + cl;
    float s1 = 0.f, s2 = 0.f;
    if (c < C) {
      const float m = mean[c], rs = rstd[c], sc = scale[c], sh = shift[c];
      for (int64_t r = r0 + rl; r < r1; r += 4) {
        if (!row_ok(nvalid, K, r)) continue;
        const float zv = z[r * ldz + c];
        float g = dy[r * ldd + c];
        if (relu && !(zv * sc + sh > 0.f)) g = 0.f;
        s1 += g;
        s2 += g * ((zv - m) * rs);
      }
    }
    red[0][rl][cl] = s1;
    red[1][rl][cl] = s2;
    __syncthreads();
    if (rl == 0 && c < C) {
      part[(int64_t)blockIdx.x * 2 * C + c] = red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl];
      part[(int64_t)blockIdx.x * 2 * C + C + c] = red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl];
    }
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void k_bn_bwd_apply(const float* __restrict__ z, int ldz, const float* __restrict__ dy, int ldd,
                                                      int64_t R, int C, const int32_t* __restrict__ nvalid, int K,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      const float* __restrict__ scale, const float* __restrict__ shift,
                                                      int relu, const float* __restrict__ sums, const float* __restrict__ count,
                                                      float* __restrict__ dz, int ldo) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= R * C) return;
  const int64_t r = (unsigned)idx / (unsigned)C;            // R*C < 2^32 (checked on the host)
  const int c = (int)(idx - r * C);
  float v = 0.f;
  if (row_ok(nvalid, K, r)) {
    const float zv = z[r * ldz + c], sc = scale[c];
    float g = dy[r * ldd + c];
    if (relu && !(zv * sc + shift[c] > 0.f)) g = 0.f;
    const float M = *count, inv = M > 0.f ? 1.0f / M : 0.f;
    const float xh = (zv - mean[c]) * rstd[c];
    v = sc * (g - sums[c] * inv - xh * sums[C + c] * inv);
  }
  dz[r * ldo + c] = v;
}
// dx = dy * [y > 0] on valid rows (plain ReLU epilogue without a norm)
__global__ __launch_bounds__(256) void k_relu_bwd(const float* __restrict__ y, const float* __restrict__ dy, int64_t R, int C,
                                                  const int32_t* __restrict__ nvalid, int K, float* __restrict__ dx) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= R * C) return;
  const int64_t r = idx / C;
  dx[idx] = (row_ok(nvalid, K, r) && y[idx] > 0.f) ? dy[idx] : 0.f;
}

// ============================================================================ masked LayerNorm backward
// forward (k_layernorm): u = x + res, y = (u - mean_r)*rstd_r*gamma + beta.  One wave per row, LN_ROWS rows per wave.
constexpr int LN_ROWS = 16;
__global__ __launch_bounds__(256) void k_layernorm_bwd(const float* __restrict__ x, const float* __restrict__ res,
                                                       const float* __restrict__ dy, int64_t R, int C,
                                                       const float* __restrict__ gamma, float eps,
                                                       const int32_t* __restrict__ nvalid, int K, float* __restrict__ du,
                                                       float* __restrict__ part /* [nblk][2C] */) {
  extern __shared__ float sm[];          // [4][2C]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float* mine = sm + (size_t)w * 2 * C;
  for (int c = lane; c < 2 * C; c += 64) mine[c] = 0.f;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + w) * LN_ROWS;
  for (int i = 0; i < LN_ROWS; ++i) {
    const int64_t row = row0 + i;
    if (row >= R) break;
    float* dr = du + row * C;
    if (!row_ok(nvalid, K, row)) {
      for (int c = lane; c < C; c += 64) dr[c] = 0.f;
      continue;
    }
    const float* xr = x + row * C;
    const float* rr = res ? res + row * C : nullptr;
    const float* gr = dy + row * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += xr[c] + (rr ? rr[c] : 0.f);
    const float mean = wsum(s) / (float)C;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) { const float d = xr[c] + (rr ? rr[c] : 0.f) - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(wsum(q) / (float)C + eps);
    float m1 = 0.f, m2 = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float xh = (xr[c] + (rr ? rr[c] : 0.f) - mean) * rstd, h = gr[c] * gamma[c];
      m1 += h;
      m2 += h * xh;
      mine[c] += gr[c] * xh;            // d gamma
      mine[C + c] += gr[c];             // d beta
    }
    m1 = wsum(m1) / (float)C;
    m2 = wsum(m2) / (float)C;
    for (int c = lane; c < C; c += 64) {
      const float xh = (xr[c] + (rr ? rr[c] : 0.f) - mean) * rstd;
      dr[c] = rstd * (gr[c] * gamma[c] - m1 - xh * m2);
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * C; c += 256)
    part[(int64_t)blockIdx.x * 2 * C + c] = sm[c] + sm[2 * C + c] + sm[4 * C + c] + sm[6 * C + c];
}

// The same for C = 4 * LPR with LPR (lanes per row) a power of two <= 64 and 16-byte aligned rows: a row lives in the registers of LPR
// lanes (one float4 each of u = x + res and of dy: every operand is read ONCE — the kernel above re-reads the row four times with scalar
// loads), row reductions are xor-shuffles inside the LPR lanes, and a thread keeps the d gamma / d beta partials of its own four columns
// in registers over the LNV_ROWS rows of its workgroup.
constexpr int LNV_ROWS = 64;       // = 4 * LN_ROWS: the scratch sized for the scalar kernel's partials fits
template <int LPR>
__global__ __launch_bounds__(256) void k_layernorm_bwd_v4(const float* __restrict__ x, const float* __restrict__ res,
                                                          const float* __restrict__ dy, int64_t R, const float* __restrict__ gamma, float eps,
                                                          const int32_t* __restrict__ nvalid, int K, float* __restrict__ du,
                                                          float* __restrict__ part /* [nblk][2C] */) {
  constexpr int C = 4 * LPR, RPP = 256 / LPR;          // rows per pass
  __shared__ float4 accg[256], accb[256];
  const int cv = threadIdx.x % LPR, rl = threadIdx.x / LPR;
  const int64_t row0 = (int64_t)blockIdx.x * LNV_ROWS;
  const float4 gm = make_float4(gamma[4 * cv], gamma[4 * cv + 1], gamma[4 * cv + 2], gamma[4 * cv + 3]);     // (a parameter: any alignment)
  float4 ag = make_float4(0.f, 0.f, 0.f, 0.f), ab = ag;
  auto rsum = [](float v) {
#pragma unroll
    for (int off = LPR / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
  };
  for (int i = 0; i < LNV_ROWS / RPP; ++i) {
    const int64_t row = row0 + (int64_t)i * RPP + rl;
    if (row >= R) continue;                      // (uniform within the LPR lanes of a row: the shuffles below stay inside them)
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row_ok(nvalid, K, row)) {
      float4 u = *reinterpret_cast<const float4*>(x + row * C + 4 * cv);
      if (res) { const float4 r4 = *reinterpret_cast<const float4*>(res + row * C + 4 * cv); u.x += r4.x; u.y += r4.y; u.z += r4.z; u.w += r4.w; }
      const float4 g4 = *reinterpret_cast<const float4*>(dy + row * C + 4 * cv);
      const float mean = rsum((u.x + u.y) + (u.z + u.w)) * (1.0f / C);
      const float d0 = u.x - mean, d1 = u.y - mean, d2 = u.z - mean, d3 = u.w - mean;
      const float rstd = 1.0f / sqrtf(rsum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) * (1.0f / C) + eps);
      const float x0 = d0 * rstd, x1 = d1 * rstd, x2 = d2 * rstd, x3 = d3 * rstd;
      const float h0 = g4.x * gm.x, h1 = g4.y * gm.y, h2 = g4.z * gm.z, h3 = g4.w * gm.w;
      const float m1 = rsum((h0 + h1) + (h2 + h3)) * (1.0f / C);
      const float m2 = rsum((h0 * x0 + h1 * x1) + (h2 * x2 + h3 * x3)) * (1.0f / C);
      ag.x += g4.x * x0; ag.y += g4.y * x1; ag.z += g4.z * x2; ag.w += g4.w * x3;
      ab.x += g4.x; ab.y += g4.y; ab.z += g4.z; ab.w += g4.w;
      o = make_float4(rstd * (h0 - m1 - x0 * m2), rstd * (h1 - m1 - x1 * m2), rstd * (h2 - m1 - x2 * m2), rstd * (h3 - m1 - x3 * m2));
    }
    *reinterpret_cast<float4*>(du + row * C + 4 * cv) = o;
  }
  accg[threadIdx.x] = ag;
  accb[threadIdx.x] = ab;
  __syncthreads();
  if (threadIdx.x < LPR) {          // the RPP row lanes of a column group, in lane order
    float4 tg = accg[threadIdx.x], tb = accb[threadIdx.x];
    for (int q = 1; q < RPP; ++q) {
      const float4 a = accg[q * LPR + threadIdx.x], b = accb[q * LPR + threadIdx.x];
      tg.x += a.x; tg.y += a.y; tg.z += a.z; tg.w += a.w;
      tb.x += b.x; tb.y += b.y; tb.z += b.z; tb.w += b.w;
    }
    float* P = part + (int64_t)blockIdx.x * 2 * C;
    *reinterpret_cast<float4*>(P + 4 * threadIdx.x) = tg;
    *reinterpret_cast<float4*>(P + C + 4 * threadIdx.x) = tb;
  }
}

// out[i] (+)= sum_b part[b * stride + i], i < n: 64 outputs per block, 4 lanes per output, partials added in a fixed order
// out[y][i] (+)= sum_b part[b * stride + y * yoff + i]: 64 columns x 16 row lanes per block, a lane adds its partials b = q, q + 16, ... in
// order with eight loads in flight (four lanes and one load at a time, 738 partials of a 47 200-row LayerNorm were a 45 us chain),
// then the 16 lanes in order.  blockIdx.y picks the output (dgamma | dbeta in one launch).
__global__ __launch_bounds__(1024) void k_cols_reduce(const float* __restrict__ part, int nparts, int64_t stride, int n, int64_t yoff,
                                                      float* __restrict__ out0, float* __restrict__ out1, int accumulate) {
  __shared__ float red[16][64];
  const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + c;
  const float* src = part + (int64_t)blockIdx.y * yoff + i;
  float* out = blockIdx.y ? out1 : out0;
  float acc = 0.f;
  if (i < n) {
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

// ============================================================================ per-node set attention backward
// One wave per (node, head), mirroring k_set_attention: P = softmax(q k^T / sqrt(dk)) over the valid slots, o = P v.
__global__ __launch_bounds__(64) void k_set_attention_bwd(const float* __restrict__ q, const float* __restrict__ k,
                                                          const float* __restrict__ v, const float* __restrict__ dout, int K,
                                                          int H, int dk, const int32_t* __restrict__ nvalid,
                                                          const float* __restrict__ pmask, float* __restrict__ dq,
                                                          float* __restrict__ dkk, float* __restrict__ dv) {
  extern __shared__ float sm[];
  const int node = blockIdx.x / H, h = blockIdx.x - node * H;
  const int lane = threadIdx.x;
  const int kv = nvalid ? nvalid[node] : K;
  const int D = H * dk;
  float* sq = sm;                // [K][dk]  q / sqrt(dk)
  float* sk = sq + K * dk;
  float* sv = sk + K * dk;
  float* sg = sv + K * dk;       // [K][dk]  d out
  float* sp = sg + K * dk;       // [K][K+1] P
  float* sd = sp + K * (K + 1);  // [K][K+1] dS
  const float temp = sqrtf((float)dk);
  const int64_t base = (int64_t)node * K * D + (int64_t)h * dk;
  for (int i = lane; i < kv * dk; i += 64) {
    const int r = i / dk, c = i - r * dk;
    const int64_t o = base + (int64_t)r * D + c;
    sq[i] = q[o] / temp;
    sk[i] = k[o];
    sv[i] = v[o];
    sg[i] = dout[o];
  }
  __syncthreads();
  for (int i = lane; i < kv * kv; i += 64) {
    const int a = i / kv, b = i - a * kv;
    float s = 0.f, d = 0.f;
    for (int c = 0; c < dk; ++c) { s += sq[a * dk + c] * sk[b * dk + c]; d += sg[a * dk + c] * sv[b * dk + c]; }
    sp[a * (K + 1) + b] = s;
    sd[a * (K + 1) + b] = d;       // dP
  }
  __syncthreads();
  for (int a = lane; a < kv; a += 64) {
    float m = -INFINITY;
    for (int b = 0; b < kv; ++b) m = fmaxf(m, sp[a * (K + 1) + b]);
    float zs = 0.f;
    for (int b = 0; b < kv; ++b) { const float e = expf(sp[a * (K + 1) + b] - m); sp[a * (K + 1) + b] = e; zs += e; }
    // with attention dropout the forward used P' = P*m (m = 0 or 1/(1-p)): dP = dP'*m, and dV below needs P'
    const float* pm = pmask ? pmask + ((int64_t)blockIdx.x * K + a) * K : nullptr;
    float dot = 0.f;
    for (int b = 0; b < kv; ++b) {
      const float p = sp[a * (K + 1) + b] / zs, m = pm ? pm[b] : 1.0f, dp = sd[a * (K + 1) + b] * m;
      sp[a * (K + 1) + b] = p;
      sd[a * (K + 1) + b] = dp;
      dot += p * dp;
    }
    for (int b = 0; b < kv; ++b) {
      const float p = sp[a * (K + 1) + b];
      sd[a * (K + 1) + b] = p * (sd[a * (K + 1) + b] - dot);     // dS
      sp[a * (K + 1) + b] = p * (pm ? pm[b] : 1.0f);             // P' for dV
    }
  }
  __syncthreads();
  for (int i = lane; i < K * dk; i += 64) {
    const int a = i / dk, c = i - a * dk;
    float gq = 0.f, gk = 0.f, gv = 0.f;
    if (a < kv)
      for (int b = 0; b < kv; ++b) {
        gq += sd[a * (K + 1) + b] * sk[b * dk + c];
        gk += sd[b * (K + 1) + a] * sq[b * dk + c];
        gv += sp[b * (K + 1) + a] * sg[b * dk + c];
      }
    const int64_t o = base + (int64_t)a * D + c;
    dq[o] = gq / temp;
    dkk[o] = gk;
    dv[o] = gv;
  }
}

// ============================================================================ GINE aggregation backward
// forward: out_i = (1+eps) h_i + sum_{e: j->i} relu(h_j + ee_e).  Per SOURCE node j over its out-edges (reverse CSR:
// rcol = destination, rperm = edge id):  dh_j = (1+eps) g_j + sum_e [h_j + ee_e > 0] g_i ;  dee_e = [..] g_i.
__global__ __launch_bounds__(256) void k_gine_bwd(const float* __restrict__ h, const float* __restrict__ ee,
                                                  const float* __restrict__ g, int64_t N, int C,
                                                  const int32_t* __restrict__ rrow, const int32_t* __restrict__ rcol,
                                                  const int32_t* __restrict__ rperm, const float* __restrict__ eps,
                                                  float* __restrict__ dh, float* __restrict__ dee, const float* __restrict__ plus) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= N * C) return;
  const int64_t j = idx / C;
  const int c = (int)(idx - j * C);
  const float hv = h[idx];
  float acc = (1.0f + (eps ? *eps : 0.f)) * g[idx];
  const int lo = rrow[j], hi = rrow[j + 1];
  for (int s0 = lo; s0 < hi; s0 += 4) {        // four out-edges in flight: ids, then their ee / g rows (one at a time the pass was a
    int64_t e[4], d[4];                        // chain of dependent L2 round trips: 11.6 us for 3 MB)
#pragma unroll
    for (int u = 0; u < 4; ++u) { e[u] = 0; d[u] = 0; if (s0 + u < hi) { e[u] = rperm[s0 + u]; d[u] = rcol[s0 + u]; } }
    float ev[4], gv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { ev[u] = 0.f; gv[u] = 0.f; if (s0 + u < hi) { ev[u] = ee[e[u] * C + c]; gv[u] = g[d[u] * C + c]; } }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (s0 + u < hi) {
        const float gi = (hv + ev[u] > 0.f) ? gv[u] : 0.f;
        acc += gi;
        dee[e[u] * C + c] = gi;
      }
    }
  }
  dh[idx] = plus ? acc + plus[idx] : acc;      // plus: the gradient of the residual branch of h, added in the same pass
}

// ============================================================================ broadcasts / scatters / reductions
// dx[n,k,:] = g[n,:] for k < nvalid[n], else 0      (adjoint of k_slot_sum over the valid slots)
__global__ __launch_bounds__(256) void k_slot_bcast(const float* __restrict__ g, int64_t N, int K, int C,
                                                    const int32_t* __restrict__ nvalid, float* __restrict__ dx) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= N * K * C) return;
  const int64_t r = idx / C;
  const int c = (int)(idx - r * C);
  const int64_t n = r / K;
  dx[idx] = row_ok(nvalid, K, r) ? g[n * C + c] : 0.f;
}
// dx[i,:] = g[graph(i),:] (* 1/n_g for mean pooling)
__global__ __launch_bounds__(256) void k_segment_bcast(const float* __restrict__ g, int C, const int32_t* __restrict__ graph_ptr,
                                                       int mode, float* __restrict__ dx) {
  const int b = blockIdx.x;
  const int lo = graph_ptr[b], hi = graph_ptr[b + 1];
  const float w = (mode == 1 && hi > lo) ? 1.0f / (float)(hi - lo) : 1.0f;
  for (int64_t i = threadIdx.x; i < (int64_t)(hi - lo) * C; i += 256) {
    const int c = (int)(i % C);
    dx[(int64_t)lo * C + i] = g[(int64_t)b * C + c] * w;
  }
}
// dT_f[v, :] += sum over the rows r with idx[r,f] == v of g[r, :] — deterministic (no atomics: nn.Embedding's own backward is not) —
// for L gradient planes g[l] at once (the L layers' edge encoders of a GINE stack share their index column: one pair of launches
// per step instead of one per layer).
//   chunk : a workgroup takes 64 consecutive rows and one plane: the rows' DISTINCT table rows (a row leads if no earlier row of the
//           chunk has its id) get a slot each; a thread owns one channel and — privately — that channel's column of an LDS table
//           [slot][C]: tab[slot[r]][c] += g[r][c] in row order (no sorting, no cross-thread traffic)   -> part[l][chunk][slot][C],
//           vals[chunk][slot], nvals[chunk]
//   gather: workgroup (v, l) looks v up in every chunk's list (<= 64 ids per chunk), and — if it occurs at all — adds the matching
//           partial rows in chunk order (float4 columns x row lanes, four loads in flight, the lanes added in order).
// Work is proportional to the rows, not to rows x table size (a 500-row table of which ZINC uses 4 to 28 rows).  The first version
// (256-row chunks sorted by id, a launch pair per table and layer) cost 26 us per call at 6 372 edges, 7 calls a training step.
// An index outside [0, V) contributes nothing and raises bit 0 of *status.
constexpr int EMB_ROWS = 64;
struct EmbPtrs { float* p[16]; };
__global__ __launch_bounds__(256) void k_embedding_bwd_chunk(const int64_t* __restrict__ idx, int ldi, int f, int64_t R, int64_t V, int C,
                                                             const float* __restrict__ g, int64_t g_plane, float* __restrict__ part,
                                                             int64_t part_plane, int32_t* __restrict__ vals, int32_t* __restrict__ nvals,
                                                             int32_t* __restrict__ status) {
  extern __shared__ float eb_tab[];          // [EMB_ROWS][C]
  __shared__ int32_t ids[EMB_ROWS], lead[EMB_ROWS], slot[EMB_ROWS];
  __shared__ int32_t nlead_s;
  const int tid = threadIdx.x, l = blockIdx.y;
  const int64_t r0 = (int64_t)blockIdx.x * EMB_ROWS;
  const int nr = (int)(R - r0 < EMB_ROWS ? R - r0 : EMB_ROWS);
  bool bad = false;
  int id = -1;
  if (tid < EMB_ROWS) {
    const int64_t id64 = tid < nr ? idx[(r0 + tid) * ldi + f] : -1;
    bad = tid < nr && (uint64_t)id64 >= (uint64_t)V;
    id = (tid < nr && !bad) ? (int)id64 : -1;
    ids[tid] = id;
  }
  __syncthreads();
  if (tid < EMB_ROWS) {                      // (wave 0)
    bool leader = id >= 0;
    for (int j = 0; j < tid && leader; ++j) leader = ids[j] != id;
    const unsigned long long lm = __ballot(leader);
    if (leader) lead[__popcll(lm & ((1ull << tid) - 1ull))] = id;
    if (tid == 0) nlead_s = __popcll(lm);
  }
  __syncthreads();
  const int nlead = nlead_s;
  if (tid < EMB_ROWS) {
    int my = -1;
    for (int sidx = 0; sidx < nlead; ++sidx) my = (lead[sidx] == id) ? sidx : my;      // (ids are distinct among the leaders)
    slot[tid] = id < 0 ? -1 : my;
  }
  for (int i = tid; i < nlead * C; i += blockDim.x) eb_tab[i] = 0.f;
  __syncthreads();
  const float* gl = g + (int64_t)l * g_plane + r0 * C;
  float* P = part + (int64_t)l * part_plane + (int64_t)blockIdx.x * EMB_ROWS * C;
  for (int c = tid; c < C; c += blockDim.x) {          // my channel: nobody else touches column c of the table
    for (int j0 = 0; j0 < nr; j0 += 16) {
      float gv[16]; int sl[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        sl[u] = j0 + u < nr ? slot[j0 + u] : -1;
        gv[u] = sl[u] >= 0 ? gl[(int64_t)(j0 + u) * C + c] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (sl[u] >= 0) eb_tab[sl[u] * C + c] += gv[u];
    }
    for (int sidx = 0; sidx < nlead; ++sidx) P[(int64_t)sidx * C + c] = eb_tab[sidx * C + c];
  }
  if (l == 0) {
    if (tid < nlead) vals[(int64_t)blockIdx.x * EMB_ROWS + tid] = lead[tid];
    if (tid == 0) nvals[blockIdx.x] = nlead;
  }
  if (status != nullptr && __syncthreads_or(bad) && tid == 0) atomicOr(status, 1);
}

__global__ __launch_bounds__(256) void k_embedding_bwd_gather(const float* __restrict__ part, int64_t part_plane,
                                                              const int32_t* __restrict__ vals, const int32_t* __restrict__ nvals,
                                                              int nchunks, int C, EmbPtrs dts) {
  __shared__ int32_t hit[1024], list[1024];
  __shared__ int32_t nlist_s;
  __shared__ float4 acc4[256];
  const int v = blockIdx.x, tid = threadIdx.x;
  float* dT = dts.p[blockIdx.y];
  const float* pl = part + (int64_t)blockIdx.y * part_plane;
  for (int c0 = 0; c0 < nchunks; c0 += 1024) {          // (1024 chunks = 65 536 rows per pass)
    const int nc = nchunks - c0 < 1024 ? nchunks - c0 : 1024;
    for (int j = tid; j < nc; j += 256) {
      const int n = nvals[c0 + j];
      const int32_t* vl = vals + (int64_t)(c0 + j) * EMB_ROWS;
      int sidx = -1;
      for (int q0 = 0; q0 < n; q0 += 8) {                // eight ids in flight
        int t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = q0 + u < n ? vl[q0 + u] : -1;
#pragma unroll
        for (int u = 0; u < 8; ++u) sidx = (t[u] == v) ? q0 + u : sidx;
      }
      hit[j] = sidx;
    }
    __syncthreads();
    if (tid < 64) {                                      // the chunks that hold v, in chunk order (wave 0)
      int cnt = 0;
      for (int base = 0; base < nc; base += 64) {
        const int h = base + tid < nc ? hit[base + tid] : -1;
        const unsigned long long m = __ballot(h >= 0);
        if (h >= 0) list[cnt + __popcll(m & ((1ull << tid) - 1ull))] = (c0 + base + tid) * EMB_ROWS + h;
        cnt += __popcll(m);
      }
      if (tid == 0) nlist_s = cnt;
    }
    __syncthreads();
    const int nl = nlist_s;
    if (nl > 0) {
      const int C4 = C >> 2;
      if ((C & 3) == 0 && C4 <= 256) {
        const int nql = 256 / C4, cv = tid % C4, ql = tid / C4;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ql < nql) {
          int e = ql;
          for (; e + 3 * nql < nl; e += 4 * nql) {
            const float4 x0 = *reinterpret_cast<const float4*>(pl + (int64_t)list[e] * C + 4 * cv);
            const float4 x1 = *reinterpret_cast<const float4*>(pl + (int64_t)list[e + nql] * C + 4 * cv);
            const float4 x2 = *reinterpret_cast<const float4*>(pl + (int64_t)list[e + 2 * nql] * C + 4 * cv);
            const float4 x3 = *reinterpret_cast<const float4*>(pl + (int64_t)list[e + 3 * nql] * C + 4 * cv);
            a.x = (((a.x + x0.x) + x1.x) + x2.x) + x3.x; a.y = (((a.y + x0.y) + x1.y) + x2.y) + x3.y;
            a.z = (((a.z + x0.z) + x1.z) + x2.z) + x3.z; a.w = (((a.w + x0.w) + x1.w) + x2.w) + x3.w;
          }
          for (; e < nl; e += nql) {
            const float4 x = *reinterpret_cast<const float4*>(pl + (int64_t)list[e] * C + 4 * cv);
            a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w;
          }
        }
        acc4[tid] = a;
        __syncthreads();
        if (tid < C4) {
          float4 t = acc4[tid];
          for (int q = 1; q < nql; ++q) { const float4 u = acc4[q * C4 + tid]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
          float* o = dT + (int64_t)v * C + 4 * tid;
          o[0] += t.x; o[1] += t.y; o[2] += t.z; o[3] += t.w;
        }
      } else {
        for (int c = tid; c < C; c += 256) {
          float a = 0.f;
          for (int e = 0; e < nl; ++e) a += pl[(int64_t)list[e] * C + c];
          dT[(int64_t)v * C + c] += a;
        }
      }
    }
    __syncthreads();
  }
}
// out[0] = sum_i a[i]*b[i]   (two stages, deterministic)
__global__ __launch_bounds__(256) void k_dot_partial(const float* __restrict__ a, const float* __restrict__ b, int64_t n,
                                                     float* __restrict__ part) {
  __shared__ float ws[4];
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += a[i] * b[i];
  s = wsum(s);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// ============================================================================ Adam (torch.optim.Adam, no amsgrad)
__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps,
                                              float wd, float bc1, float bc2_sqrt, float gscale) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float gi = g[i] * gscale;
  if (wd != 0.f) gi += wd * p[i];
  const float mi = b1 * m[i] + (1.0f - b1) * gi;
  const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  p[i] -= (lr / bc1) * (mi / denom);
}

}  // namespace
}  // namespace sn

namespace sn {
bool attention16_backward(const float* q, const float* k, const float* v, const float* dout, int64_t N, int K, int heads, int dk,
                          const int32_t* nvalid, const float* prob_mask, float* dq, float* dkk, float* dv, hipStream_t st);   // attention16.hip
}
using namespace sn;

static inline int64_t wgrad_rows_per_block(int64_t R) {
  int64_t rpb = 128;                       // >= 2 workgroups per CU at the bench's 47 k rows; partials stay a few MB
  while (cdiv(R, rpb) > 512) rpb *= 2;
  return rpb;
}
extern "C" int64_t sn_linear_wgrad_scratch_floats(int64_t R, int d_in, int d_out) {
  const int64_t nblk = cdiv(R > 0 ? R : 1, wgrad_rows_per_block(R));
  return (nblk + 16) * ((int64_t)d_in * d_out + d_out);
}
extern "C" int sn_linear_wgrad_f32(const float* x, int ldx, const float* dy, int ldy, int64_t R, int d_in, int d_out,
                                   const int32_t* nvalid, int K, float* dW, float* db, float* scratch, void* stream) {
  SN_REQUIRE(x && dy && dW && scratch && R >= 0 && d_in > 0 && d_out > 0, "sn_linear_wgrad_f32: bad arguments");
  SN_REQUIRE(ldx >= d_in && ldy >= d_out, "sn_linear_wgrad_f32: leading dimension too small");
  SN_REQUIRE(!nvalid || K > 0, "sn_linear_wgrad_f32: nvalid needs K > 0");
  hipStream_t st = (hipStream_t)stream;
  const int64_t rpb = wgrad_rows_per_block(R);
  const int nblk = (int)cdiv(R > 0 ? R : 1, rpb);
  // one partial row per chunk: [d_out*d_in weight sums | d_out bias sums]; when db directly follows dW in memory (the Python
  // wrapper allocates them as one buffer) a single two-stage reduction produces both
  const int64_t n = (int64_t)d_in * d_out, pstride = n + d_out;
  float* part = scratch;
  float* tmp = scratch + (int64_t)nblk * pstride;
  dim3 grid((unsigned)nblk, (unsigned)cdiv(d_out, WG_OC), (unsigned)cdiv(d_in, WG_IC));
  const int vec = (d_in % 4 == 0 && d_out % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 &&
                   ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0) ? 1 : 0;
  SN_REQUIRE(R < (1ll << 31), "sn_linear_wgrad_f32: too many rows");
  hipLaunchKernelGGL(k_wgrad, grid, dim3(256), 0, st, x, ldx, dy, ldy, R, d_in, d_out, nvalid, K, rpb, part, pstride, db ? 1 : 0, vec);
  SN_CHECK_LAUNCH("k_wgrad");
  if (db == dW + n) {
    sum_parts(part, nblk, pstride, dW, tmp, st);
  } else {
    // strided partials -> two reductions (k_sum_parts takes a dense [nblk][n] block: reduce the weight part through tmp rows)
    hipLaunchKernelGGL(k_sum_strided, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, (const float*)part, nblk, pstride, n, dW);
    if (db) hipLaunchKernelGGL(k_sum_strided, dim3((unsigned)cdiv(d_out, 256)), dim3(256), 0, st, (const float*)(part + n), nblk, pstride,
                               (int64_t)d_out, db);
  }
  SN_CHECK_LAUNCH("k_sum_parts");
  return SN_OK;
}

static inline int bn_bwd_blocks(int64_t R) {
  const int64_t b = cdiv(R > 0 ? R : 1, 64);
  return (int)(b > 2048 ? 2048 : b);
}
extern "C" int64_t sn_bn_act_bwd_scratch_floats(int64_t R, int C) { return (int64_t)(bn_bwd_blocks(R) + 16) * 2 * C; }
extern "C" int sn_bn_act_bwd_f32(const float* z, int ldz, const float* dy, int ldd, int64_t R, int C, const int32_t* nvalid, int K,
                                 const float* mean, const float* rstd, const float* scale, const float* shift, int relu,
                                 const float* count, float* sums /* [2C]: d beta, d (gamma) */, float* dz, int ldo,
                                 float* scratch /* [sn_bn_act_bwd_scratch_floats(R, C)] */, void* stream) {
  SN_REQUIRE(z && dy && mean && rstd && scale && shift && count && sums && dz && scratch && R >= 0 && C > 0,
             "sn_bn_act_bwd_f32: bad arguments");
  SN_REQUIRE(ldz >= C && ldd >= C && ldo >= C, "sn_bn_act_bwd_f32: leading dimension too small");
  SN_REQUIRE(!nvalid || K > 0, "sn_bn_act_bwd_f32: nvalid needs K > 0");
  if (R == 0) return SN_OK;
  SN_REQUIRE(R * C < (1ll << 32) && R < (1ll << 31), "sn_bn_act_bwd_f32: R*C too large");
  hipStream_t st = (hipStream_t)stream;
  const int nblk = bn_bwd_blocks(R);
  const int64_t rpb = cdiv(R, nblk);
  hipLaunchKernelGGL(k_bn_bwd_partial, dim3((unsigned)nblk), dim3(256), 0, st, z, ldz, dy, ldd, R, C, nvalid, K, mean, rstd, scale,
                     shift, relu, rpb, scratch);
  sum_parts(scratch, nblk, (int64_t)2 * C, sums, scratch + (int64_t)nblk * 2 * C, st);
  hipLaunchKernelGGL(k_bn_bwd_apply, dim3((unsigned)cdiv(R * C, 256)), dim3(256), 0, st, z, ldz, dy, ldd, R, C, nvalid, K, mean,
                     rstd, scale, shift, relu, sums, count, dz, ldo);
  SN_CHECK_LAUNCH("sn_bn_act_bwd_f32");
  return SN_OK;
}

extern "C" int sn_relu_bwd_f32(const float* y, const float* dy, int64_t R, int C, const int32_t* nvalid, int K, float* dx,
                               void* stream) {
  SN_REQUIRE(y && dy && dx && R >= 0 && C > 0 && (!nvalid || K > 0), "sn_relu_bwd_f32: bad arguments");
  if (R == 0) return SN_OK;
  hipLaunchKernelGGL(k_relu_bwd, dim3((unsigned)cdiv(R * C, 256)), dim3(256), 0, (hipStream_t)stream, y, dy, R, C, nvalid, K, dx);
  SN_CHECK_LAUNCH("sn_relu_bwd_f32");
  return SN_OK;
}

extern "C" int64_t sn_layernorm_bwd_scratch_floats(int64_t R, int C) { return (cdiv(R > 0 ? R : 1, 4 * LN_ROWS) + 17) * 2 * C; }
static int layernorm_bwd_impl(const float* x, const float* residual, const float* dy, int64_t R, int C,
                              const float* gamma, float eps, const int32_t* nvalid, int K, float* du,
                              float* dgamma, float* dbeta /* contiguous pair is NOT required */, float* scratch, int accumulate,
                              void* stream) {
  SN_REQUIRE(x && dy && gamma && du && dgamma && dbeta && scratch && R >= 0 && C > 0 && (!nvalid || K > 0),
             "sn_masked_layernorm_bwd_f32: bad arguments");
  SN_REQUIRE((size_t)8 * C * sizeof(float) <= 64 * 1024, "sn_masked_layernorm_bwd_f32: C too large");
  hipStream_t st = (hipStream_t)stream;
  int nblk = (int)cdiv(R > 0 ? R : 1, 4 * LN_ROWS);
  auto a16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  const bool vec = (C == 32 || C == 64 || C == 128 || C == 256) && a16(x) && a16(dy) && a16(du) && (!residual || a16(residual));
  if (vec) {
    nblk = (int)cdiv(R > 0 ? R : 1, LNV_ROWS);                 // (fewer, larger partials: fits the scratch sized for 4 * LN_ROWS rows per block)
    if (C == 32) hipLaunchKernelGGL(k_layernorm_bwd_v4<8>, dim3((unsigned)nblk), dim3(256), 0, st, x, residual, dy, R, gamma, eps, nvalid, K, du, scratch);
    else if (C == 64) hipLaunchKernelGGL(k_layernorm_bwd_v4<16>, dim3((unsigned)nblk), dim3(256), 0, st, x, residual, dy, R, gamma, eps, nvalid, K, du, scratch);
    else if (C == 128) hipLaunchKernelGGL(k_layernorm_bwd_v4<32>, dim3((unsigned)nblk), dim3(256), 0, st, x, residual, dy, R, gamma, eps, nvalid, K, du, scratch);
    else hipLaunchKernelGGL(k_layernorm_bwd_v4<64>, dim3((unsigned)nblk), dim3(256), 0, st, x, residual, dy, R, gamma, eps, nvalid, K, du, scratch);
  } else
  hipLaunchKernelGGL(k_layernorm_bwd, dim3((unsigned)nblk), dim3(256), (size_t)8 * C * sizeof(float), st, x, residual, dy, R, C,
                     gamma, eps, nvalid, K, du, scratch);
  // scratch rows are [d gamma (C) | d beta (C)] per block: added in block order straight into the two outputs
  hipLaunchKernelGGL(k_cols_reduce, dim3((unsigned)cdiv(C, 64), 2), dim3(1024), 0, st, (const float*)scratch, nblk, (int64_t)2 * C, C, (int64_t)C,
                     dgamma, dbeta, accumulate);
  SN_CHECK_LAUNCH("sn_masked_layernorm_bwd_f32");
  return SN_OK;
}

extern "C" int sn_masked_layernorm_bwd_f32(const float* x, const float* residual, const float* dy, int64_t R, int C,
                                           const float* gamma, float eps, const int32_t* nvalid, int K, float* du,
                                           float* dgamma, float* dbeta /* contiguous pair is NOT required */, float* scratch,
                                           void* stream) {
  return layernorm_bwd_impl(x, residual, dy, R, C, gamma, eps, nvalid, K, du, dgamma, dbeta, scratch, 0, stream);
}
/* the same with d gamma / d beta ADDED to the given buffers (a parameter's .grad): no separate accumulation launch */
extern "C" int sn_masked_layernorm_bwd_acc_f32(const float* x, const float* residual, const float* dy, int64_t R, int C,
                                               const float* gamma, float eps, const int32_t* nvalid, int K, float* du,
                                               float* dgamma, float* dbeta, float* scratch, void* stream) {
  return layernorm_bwd_impl(x, residual, dy, R, C, gamma, eps, nvalid, K, du, dgamma, dbeta, scratch, 1, stream);
}

extern "C" int sn_set_attention_bwd_f32(const float* q, const float* k, const float* v, const float* dout, int64_t N, int K,
                                        int heads, int dk, const int32_t* nvalid, const float* prob_mask, float* dq,
                                        float* dk_out, float* dv, void* stream) {
  SN_REQUIRE(q && k && v && dout && dq && dk_out && dv && N >= 0 && K > 0 && heads > 0 && dk > 0,
             "sn_set_attention_bwd_f32: bad arguments");
  if (N == 0) return SN_OK;
  if (sn::attention16_backward(q, k, v, dout, N, K, heads, dk, nvalid, prob_mask, dq, dk_out, dv, (hipStream_t)stream)) {   // K <= 16
    SN_CHECK_LAUNCH("sn_set_attention_bwd_f32");
    return SN_OK;
  }
  const size_t lds = ((size_t)4 * K * dk + (size_t)2 * K * (K + 1)) * sizeof(float);
  SN_REQUIRE(lds <= 160 * 1024, "sn_set_attention_bwd_f32: K=%d dk=%d needs %zu B of LDS (> 160 KiB)", K, dk, lds);
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(k_set_attention_bwd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
          hipSuccess)
    return fail(SN_ERR_LAUNCH, "sn_set_attention_bwd_f32: cannot raise the dynamic LDS limit");
  hipLaunchKernelGGL(k_set_attention_bwd, dim3((unsigned)(N * heads)), dim3(64), lds, (hipStream_t)stream, q, k, v, dout, K,
                     heads, dk, nvalid, prob_mask, dq, dk_out, dv);
  SN_CHECK_LAUNCH("sn_set_attention_bwd_f32");
  return SN_OK;
}

extern "C" int sn_gine_aggregate_bwd_f32(const float* h, const float* ee, const float* g, int64_t N, int C,
                                         const int32_t* rev_rowptr, const int32_t* rev_col, const int32_t* rev_eperm,
                                         const float* eps, float* dh, float* dee, void* stream) {
  SN_REQUIRE(h && ee && g && rev_rowptr && dh && dee && N >= 0 && C > 0, "sn_gine_aggregate_bwd_f32: bad arguments");
  if (N == 0) return SN_OK;
  hipLaunchKernelGGL(k_gine_bwd, dim3((unsigned)cdiv(N * C, 256)), dim3(256), 0, (hipStream_t)stream, h, ee, g, N, C, rev_rowptr,
                     rev_col, rev_eperm, eps, dh, dee, (const float*)nullptr);
  SN_CHECK_LAUNCH("sn_gine_aggregate_bwd_f32");
  return SN_OK;
}

extern "C" int sn_gine_aggregate_bwd_add_f32(const float* h, const float* ee, const float* g, const float* plus, int64_t N, int C,
                                             const int32_t* rev_rowptr, const int32_t* rev_col, const int32_t* rev_eperm, const float* eps,
                                             float* dh, float* dee, void* stream) {
  SN_REQUIRE(h && ee && g && plus && rev_rowptr && dh && dee && N >= 0 && C > 0, "sn_gine_aggregate_bwd_add_f32: bad arguments");
  if (N == 0) return SN_OK;
  hipLaunchKernelGGL(k_gine_bwd, dim3((unsigned)cdiv(N * C, 256)), dim3(256), 0, (hipStream_t)stream, h, ee, g, N, C, rev_rowptr,
                     rev_col, rev_eperm, eps, dh, dee, plus);
  SN_CHECK_LAUNCH("sn_gine_aggregate_bwd_add_f32");
  return SN_OK;
}

extern "C" int sn_slot_broadcast_f32(const float* g, int64_t N, int K, int C, const int32_t* nvalid, float* dx, void* stream) {
  SN_REQUIRE(g && dx && N >= 0 && K > 0 && C > 0, "sn_slot_broadcast_f32: bad arguments");
  if (N == 0) return SN_OK;
  hipLaunchKernelGGL(k_slot_bcast, dim3((unsigned)cdiv(N * K * C, 256)), dim3(256), 0, (hipStream_t)stream, g, N, K, C, nvalid, dx);
  SN_CHECK_LAUNCH("sn_slot_broadcast_f32");
  return SN_OK;
}

extern "C" int sn_segment_broadcast_f32(const float* g, int64_t B, int C, const int32_t* graph_ptr, int mode, float* dx,
                                        void* stream) {
  SN_REQUIRE(g && dx && graph_ptr && B >= 0 && C > 0 && (mode == 0 || mode == 1), "sn_segment_broadcast_f32: bad arguments");
  if (B == 0) return SN_OK;
  hipLaunchKernelGGL(k_segment_bcast, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, g, C, graph_ptr, mode, dx);
  SN_CHECK_LAUNCH("sn_segment_broadcast_f32");
  return SN_OK;
}

extern "C" int64_t sn_embedding_bwd_layers_scratch_floats(int64_t R, int L, int C) {
  const int64_t nchunks = cdiv(R > 0 ? R : 1, EMB_ROWS);
  return (int64_t)(L > 0 ? L : 1) * nchunks * EMB_ROWS * C + nchunks * EMB_ROWS + nchunks + 16;      // partial rows per plane | ids per chunk | counts
}
extern "C" int64_t sn_embedding_bwd_scratch_floats(int64_t R, int nf, const int64_t* table_rows, int C) {
  (void)nf; (void)table_rows;
  return sn_embedding_bwd_layers_scratch_floats(R, 1, C);
}

extern "C" int sn_embedding_sum_bwd_layers_f32(const int64_t* idx, int ldi, int nf, int64_t R, int L, float* const* dtables,
                                               const int64_t* table_rows, int C, const float* g, int32_t* status, float* scratch,
                                               void* stream) {
  SN_REQUIRE(idx && dtables && table_rows && g && scratch && nf > 0 && nf <= 10 && ldi >= nf && C > 0 && C <= 512 && R >= 0 && L >= 1 && L <= 16,
             "sn_embedding_sum_bwd_layers_f32: bad arguments (C <= 512, 1 <= L <= 16)");
  if (R == 0) return SN_OK;
  const int nchunks = (int)cdiv(R, EMB_ROWS);
  const int64_t plane = (int64_t)nchunks * EMB_ROWS * C;
  float* part = scratch;
  int32_t* vals = reinterpret_cast<int32_t*>(scratch + (int64_t)L * plane);
  int32_t* nvals = vals + (int64_t)nchunks * EMB_ROWS;
  const size_t lds = (size_t)EMB_ROWS * C * sizeof(float);
  if (lds > 64 * 1024) {
    static bool raised = false;
    if (!raised) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_embedding_bwd_chunk), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048) != hipSuccess)
        return fail(SN_ERR_LAUNCH, "sn_embedding_sum_bwd_layers_f32: cannot raise the dynamic LDS limit");
      raised = true;
    }
  }
  const unsigned bs = (unsigned)(C >= 256 ? 256 : ((C + 63) / 64) * 64);
  for (int f = 0; f < nf; ++f) {
    EmbPtrs dts;
    for (int l = 0; l < 16; ++l) dts.p[l] = l < L ? dtables[(int64_t)l * nf + f] : nullptr;
    for (int l = 0; l < L; ++l) SN_REQUIRE(dts.p[l], "sn_embedding_sum_bwd_layers_f32: table %d of plane %d missing", f, l);
    SN_REQUIRE(table_rows[f] > 0 && table_rows[f] <= 65535, "sn_embedding_sum_bwd_layers_f32: table %d empty or > 65535 rows", f);
    const int64_t V = table_rows[f];
    hipLaunchKernelGGL(k_embedding_bwd_chunk, dim3((unsigned)nchunks, (unsigned)L), dim3(bs), lds, (hipStream_t)stream, idx, ldi, f, R, V, C, g,
                       R * (int64_t)C, part, plane, vals, nvals, status);
    hipLaunchKernelGGL(k_embedding_bwd_gather, dim3((unsigned)V, (unsigned)L), dim3(256), 0, (hipStream_t)stream, (const float*)part, plane,
                       (const int32_t*)vals, (const int32_t*)nvals, nchunks, C, dts);
  }
  SN_CHECK_LAUNCH("sn_embedding_sum_bwd_layers_f32");
  return SN_OK;
}

extern "C" int sn_embedding_sum_bwd_f32(const int64_t* idx, int ldi, int nf, int64_t R, float* const* dtables,
                                        const int64_t* table_rows, int C, const float* g, int32_t* status, float* scratch,
                                        void* stream) {
  SN_REQUIRE(idx && dtables && table_rows && g && scratch && nf > 0 && nf <= 10 && ldi >= nf && C > 0 && C <= 512 && R >= 0,
             "sn_embedding_sum_bwd_f32: bad arguments (C <= 512)");
  for (int f = 0; f < nf; ++f)
    SN_REQUIRE(dtables[f] && table_rows[f] > 0 && table_rows[f] <= 65535, "sn_embedding_sum_bwd_f32: table %d missing, empty or > 65535 rows", f);
  return sn_embedding_sum_bwd_layers_f32(idx, ldi, nf, R, 1, dtables, table_rows, C, g, status, scratch, stream);
}

extern "C" int sn_dot_f32(const float* a, const float* b, int64_t n, float* out, float* scratch /* [256] */, void* stream) {
  SN_REQUIRE(a && b && out && scratch && n >= 0, "sn_dot_f32: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int nblk = (int)(n > 0 ? (cdiv(n, 256) < 256 ? cdiv(n, 256) : 256) : 1);
  hipLaunchKernelGGL(k_dot_partial, dim3((unsigned)nblk), dim3(256), 0, st, a, b, n, scratch);
  hipLaunchKernelGGL(k_sum_parts, dim3(1, 1), dim3(256), 0, st, (const float*)scratch, nblk, (int64_t)1, out);
  SN_CHECK_LAUNCH("sn_dot_f32");
  return SN_OK;
}

extern "C" int sn_adam_step_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                                float eps, float weight_decay, int step, float grad_scale, void* stream) {
  SN_REQUIRE(p && g && m && v && n >= 0 && step >= 1, "sn_adam_step_f32: bad arguments");
  if (n == 0) return SN_OK;
  const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
  hipLaunchKernelGGL(k_adam, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps,
                     weight_decay, bc1, sqrtf(bc2), grad_scale);
  SN_CHECK_LAUNCH("sn_adam_step_f32");
  return SN_OK;
}
