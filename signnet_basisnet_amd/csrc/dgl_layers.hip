// dgl_layers.hip — message passing of the two remaining DGL base networks that consume the sign-invariant positional encoding
// (SURVEY.md §8 row f3): PNA (GraphPrediction/layers/pna_layer.py + pna_utils.py) and the sparse graph Transformer
// (GraphPrediction/layers/transformer.py).  Both reduce per-edge quantities over each node's in-edges; the destination-sorted
// CSR of sn_batch_plan (in-edges in edge-id order, the order DGL's reduce sees them) is the iteration space, so there are no
// atomics and the sums are reproducible.  HBM-bound gathers over small feature rows; one thread per (node, channel) /
// (node, head).
#include "common.hpp"

namespace sn {

// PNATower.reduce_func_for_h (pna_layer.py:50-56) with aggregators 'mean max min std' (pna_utils.py:13-36, EPS 1e-5) and scalers
// 'identity amplification attenuation' (:68-81): out[n, C0 + (3*s... ] — layout = cat_scalers(cat_aggregators):
//   out[n, off + (4*s + a)*C + c],  a in {mean, max, min, std},  s in {1, log(D+1)/avg_log, avg_log/log(D+1)}
// plus (optional) the node's own row in the first C columns — the tower's `torch.cat([h, g.ndata['h']], dim=1)` (:69).
// A node without in-edges gets zeros (DGL does not call the reduce function for it).
__global__ __launch_bounds__(256) void k_pna_aggregate(const float* __restrict__ msg, int ldm, const float* __restrict__ hself, int ldh,
                                                       int C, int64_t N, const int32_t* __restrict__ rowptr,
                                                       const int32_t* __restrict__ eperm, float avg_log, float* __restrict__ out,
                                                       int ldo) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= N * C) return;
  const int64_t n = i / C;
  const int c = (int)(i - n * C);
  const int lo = rowptr[n], hi = rowptr[n + 1];
  float s1 = 0.f, s2 = 0.f, mx = -INFINITY, mn = INFINITY;
  for (int e = lo; e < hi; ++e) {
    const float v = msg[(int64_t)eperm[e] * ldm + c];
    s1 += v;
    s2 += v * v;
    mx = fmaxf(mx, v);
    mn = fminf(mn, v);
  }
  float* o = out + n * ldo;
  int off = 0;
  if (hself != nullptr) { o[c] = hself[n * ldh + c]; off = C; }
  const int D = hi - lo;
  if (D == 0) {
#pragma unroll
    for (int k = 0; k < 12; ++k) o[off + k * C + c] = 0.f;
    return;
  }
  const float inv = 1.0f / (float)D;
  const float mean = s1 * inv;
  const float var = fmaxf(s2 * inv - mean * mean, 0.f);          // torch.relu(E[x^2] - E[x]^2)
  const float sd = sqrtf(var + 1e-5f);
  const float logd = logf((float)D + 1.0f);
  const float amp = logd / avg_log, att = avg_log / logd;
  const float a[4] = {mean, mx, mn, sd};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    o[off + k * C + c] = a[k];
    o[off + (4 + k) * C + c] = a[k] * amp;
    o[off + (8 + k) * C + c] = a[k] * att;
  }
}

// The same reduction with the message formed on the fly:  msg(j -> n, edge e) = Ps[j] + Pd[n] + Qe[e]  — PNATower.pretrans_edges
// (pna_layer.py:38-44) is a Linear over cat[h_src, h_dst, e], i.e. W_s h_src + W_d h_dst + (W_e e + b): the two node terms are
// computed once per NODE and the [E, 2 C_in + C_e] gather / concatenation and its E-row Linear disappear.  All towers of a layer in one
// launch (their channels side by side: C = in_dim).
__global__ __launch_bounds__(256) void k_pna_aggregate_gather(const float* __restrict__ Ps, int ldps, const float* __restrict__ Pd, int ldpd,
                                                              const float* __restrict__ Qe, int ldq, const float* __restrict__ hself, int ldh,
                                                              int C, int64_t N, const int32_t* __restrict__ rowptr,
                                                              const int32_t* __restrict__ col, const int32_t* __restrict__ eperm,
                                                              float avg_log, float* __restrict__ out, int ldo, int tower) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= N * C) return;
  const int64_t n = i / C;
  const int c = (int)(i - n * C);
  const int lo = rowptr[n], hi = rowptr[n + 1];
  const float pd = Pd[n * ldpd + c];
  // output column of (block j, channel c): j = 0 the node's own row, j = 1 + k the k-th scaled aggregate.  tower = 0: block-major
  // [j][C]; tower = it > 0: TOWER-major — tower t = c / it owns the 13 it contiguous columns [t][j][it] (a grouped Linear reads them)
  const int t_ = tower > 0 ? c / tower : 0, ct = tower > 0 ? c - t_ * tower : c;
  const int base = tower > 0 ? t_ * 13 * tower + ct : c, jstride = tower > 0 ? tower : C;
  float s1 = 0.f, s2 = 0.f, mx = -INFINITY, mn = INFINITY;
  for (int e0 = lo; e0 < hi; e0 += 4) {          // four in-edges in flight
    float a[4], q[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u] = 0.f; q[u] = 0.f;
      if (e0 + u < hi) { a[u] = Ps[(int64_t)col[e0 + u] * ldps + c]; q[u] = Qe[(int64_t)eperm[e0 + u] * ldq + c]; }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (e0 + u < hi) {
        const float v = (a[u] + pd) + q[u];
        s1 += v;
        s2 += v * v;
        mx = fmaxf(mx, v);
        mn = fminf(mn, v);
      }
    }
  }
  float* o = out + n * ldo + base;
  int off = 0;
  if (hself != nullptr) { o[0] = hself[n * ldh + c]; off = jstride; }
  const int D = hi - lo;
  if (D == 0) {
#pragma unroll
    for (int k = 0; k < 12; ++k) o[off + k * jstride] = 0.f;
    return;
  }
  const float inv = 1.0f / (float)D;
  const float mean = s1 * inv;
  const float var = fmaxf(s2 * inv - mean * mean, 0.f);
  const float sd = sqrtf(var + 1e-5f);
  const float logd = logf((float)D + 1.0f);
  const float amp = logd / avg_log, att = avg_log / logd;
  const float ag[4] = {mean, mx, mn, sd};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    o[off + k * jstride] = ag[k];
    o[off + (4 + k) * jstride] = ag[k] * amp;
    o[off + (8 + k) * jstride] = ag[k] * att;
  }
}

// y[:, g*dout .. ] = ((x[:, g*din ..] W_g^T + b_g) * rowscale[row]) * scale + shift  for G independent column groups (a block-diagonal
// Linear: the towers' posttrans Linears of a PNA layer over the tower-major aggregation output, with graph_norm's snorm_n and the
// folded BatchNorm as the epilogue; pna_layer.py:69-79).  A wave takes (16-row tile, group) pairs; the group's weight fragments
// (dout <= 16, din <= 256: <= 16 float4 per lane, straight from the row-major [G][dout][din] array) are loaded per pair from L2.
__global__ __launch_bounds__(256) void k_grouped_linear(const float* __restrict__ x, int ldx, int64_t R, int G, int din, int dout,
                                                        const float* __restrict__ W, const float* __restrict__ bias,
                                                        const float* __restrict__ rowscale, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, float* __restrict__ y, int ldy) {
  const int lane = threadIdx.x & 63, lr = lane & 15, g = lane >> 4;
  const int64_t ntiles = (R + 15) >> 4;
  const int64_t task = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (task >= ntiles * G) return;
  const int grp = (int)(task % G);
  const int64_t tile = task / G;
  const int64_t row = tile * 16 + lr;
  const bool inr = row < R;
  const int nk = (din + 15) >> 4;
  const float* xr = x + row * ldx + (int64_t)grp * din;
  const float* wr = W + ((int64_t)grp * dout + lr) * din;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int kk = 0; kk < nk; ++kk) {
    const int k0 = 16 * kk + 4 * g;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
    if (k0 < din) {           // din % 4 == 0
      if (lr < dout) { const float4 t = *reinterpret_cast<const float4*>(wr + k0); a = f32x4{t.x, t.y, t.z, t.w}; }
      if (inr) { const float4 t = *reinterpret_cast<const float4*>(xr + k0); b = f32x4{t.x, t.y, t.z, t.w}; }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) acc = mfma16(a[t], b[t], acc);
  }
  if (!inr) return;
  const float rs = rowscale ? rowscale[row] : 1.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int oc = 4 * g + r;
    if (oc < dout) {
      const int c = grp * dout + oc;
      float v = acc[r] + (bias ? bias[c] : 0.f);
      v *= rs;
      if (scale) v = v * scale[c] + shift[c];
      y[row * ldy + c] = v;
    }
  }
}

// MultiHeadAttentionLayer.propagate_attention (transformer.py:150-195, full_graph False, edge features): per in-edge (j -> i, id e)
// and head h:  score = sum_c K[j,h,c] * Q[i,h,c] / sqrt(dk) * E[e,h,c];  s = exp(clamp(score, -5, 5));
// out[i,h,:] = sum_e s * V[j,h,:] / (sum_e s + 1e-6).  One thread per (node, head); dk <= 32.
// (ldq / lde: row strides of Q, K, V and of Ee in floats — heads*dk for separate matrices, 3*heads*dk / L*heads*dk for the column
//  blocks of a fused projection: sn_edge_attention_strided_f32)
__global__ __launch_bounds__(256) void k_edge_attention(const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
                                                        int ldq, const float* __restrict__ Ee, int lde, int64_t N, int H, int dk,
                                                        const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                        const int32_t* __restrict__ eperm, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= N * H) return;
  const int64_t n = i / H;
  const int h = (int)(i - n * H);
  const int d = H * dk;
  const float root = sqrtf((float)dk);
  float q[32], acc[32];
#pragma unroll
  for (int c = 0; c < 32; ++c) { q[c] = c < dk ? Q[n * ldq + h * dk + c] : 0.f; acc[c] = 0.f; }
  float z = 0.f;
  for (int e = rowptr[n]; e < rowptr[n + 1]; ++e) {
    const int64_t j = col[e], eid = eperm[e];
    const float* kr = K + j * ldq + h * dk;
    const float* er = Ee + eid * lde + h * dk;
    const float* vr = V + j * ldq + h * dk;
    float sc = 0.f;
#pragma unroll
    for (int c = 0; c < 32; ++c)
      if (c < dk) sc += ((kr[c] * q[c]) / root) * er[c];              // src_dot_dst, scaling, imp_exp_attn — in the reference's order
    const float s = expf(fminf(fmaxf(sc, -5.f), 5.f));
    z += s;
#pragma unroll
    for (int c = 0; c < 32; ++c)
      if (c < dk) acc[c] += vr[c] * s;
  }
  const float r = 1.0f / (z + 1e-6f);
#pragma unroll
  for (int c = 0; c < 32; ++c)
    if (c < dk) out[n * d + h * dk + c] = acc[c] * r;
}

// y = act((x * rowscale[r]) * scale[c] + shift[c]) + residual    (any of rowscale / scale+shift / residual may be absent)
// act: 0 none, 1 ReLU, 2 LeakyReLU(slope), 3 ReLU after the residual add (the DeepSets / IGN `relu(x1 + x2)`).  Covers PNA's graph_norm (h * snorm_n) + BatchNorm and its mixing FCLayer's LeakyReLU.
__global__ __launch_bounds__(256) void k_pointwise(const float* __restrict__ x, int ldx, int64_t R, int C, const float* __restrict__ rowscale,
                                                   const float* __restrict__ scale, const float* __restrict__ shift, int act, float slope,
                                                   const float* __restrict__ res, int ldr, float* __restrict__ y, int ldy) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= R * C) return;
  const int64_t r = i / C;
  const int c = (int)(i - r * C);
  float v = x[r * ldx + c];
  if (rowscale) v *= rowscale[r];
  if (scale) v = v * scale[c] + shift[c];
  if (act == 1) v = fmaxf(v, 0.f);
  else if (act == 2) v = v > 0.f ? v : v * slope;
  if (res) v += res[r * ldr + c];
  if (act == 3) v = fmaxf(v, 0.f);
  y[r * ldy + c] = v;
}

// dgl.nn.pytorch.GATConv as gat_net.py:62-66 builds it (feat_drop = attn_drop = 0, negative_slope 0.2, no residual, bias, ReLU):
//   feat = fc(h) [N, H, C];  el_j = feat_j . attn_l[h], er_i = feat_i . attn_r[h];  e_ij = leaky_relu(el_j + er_i, 0.2) over in-edges j -> i;
//   a = edge_softmax (max-subtracted);  out[i,h,:] = act(sum_j a_ij feat[j,h,:] + bias[h,:]).
// One thread per (node, head), C <= 64: the in-edge rows are walked twice (scores / maximum, then weights and the sum) in edge-id
// order; a node without in-edges gets act(bias) (DGL raises for such graphs unless allow_zero_in_degree).  lse (optional, [N, H]) keeps
// max + log(sum) for the backward.
__global__ __launch_bounds__(256) void k_gat_aggregate(const float* __restrict__ feat, const float* __restrict__ attn_l,
                                                       const float* __restrict__ attn_r, const float* __restrict__ bias, int64_t N, int H,
                                                       int C, float slope, int relu, const int32_t* __restrict__ rowptr,
                                                       const int32_t* __restrict__ col, float* __restrict__ out, float* __restrict__ lse) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= N * H) return;
  const int64_t n = i / H;
  const int h = (int)(i - n * H);
  const int d = H * C;
  const float* al = attn_l + h * C;
  const float* ar = attn_r + h * C;
  float er = 0.f;
  {
    const float* fi = feat + n * d + h * C;
    for (int c = 0; c < C; ++c) er += fi[c] * ar[c];
  }
  const int lo = rowptr[n], hi = rowptr[n + 1];
  float m = -INFINITY;
  for (int e = lo; e < hi; ++e) {
    const float* fj = feat + (int64_t)col[e] * d + h * C;
    float el = 0.f;
    for (int c = 0; c < C; ++c) el += fj[c] * al[c];
    float s = el + er;
    s = s > 0.f ? s : s * slope;
    m = fmaxf(m, s);
  }
  float acc[64];
#pragma unroll
  for (int c = 0; c < 64; ++c) acc[c] = 0.f;
  float z = 0.f;
  for (int e = lo; e < hi; ++e) {
    const float* fj = feat + (int64_t)col[e] * d + h * C;
    float el = 0.f;
    for (int c = 0; c < C; ++c) el += fj[c] * al[c];
    float s = el + er;
    s = s > 0.f ? s : s * slope;
    const float w = expf(s - m);
    z += w;
#pragma unroll
    for (int c = 0; c < 64; ++c)
      if (c < C) acc[c] += w * fj[c];
  }
  const float r = hi > lo ? 1.0f / z : 0.f;
  float* o = out + n * d + h * C;
#pragma unroll
  for (int c = 0; c < 64; ++c)
    if (c < C) {
      float v = acc[c] * r + (bias ? bias[h * C + c] : 0.f);
      if (relu) v = fmaxf(v, 0.f);
      o[c] = v;
    }
  if (lse) lse[i] = hi > lo ? m + logf(z) : 0.f;
}

// ---------------------------------------------------------------------------------------------------------------- adjoints
// (SURVEY.md §8 f1 for the f3 nets: the reference gets these from torch.autograd through DGL's message passing.)  All by CSR walks
// with one owner per output element: no atomics, bitwise reproducible.

// out[n, :] = sum over the CSR range of n of g[eperm[p], :]: the adjoint of gathering node rows onto edges (h[dst] with the
// destination-sorted plan, h[src] with the plan of the flipped edge list).
__global__ __launch_bounds__(256) void k_edge_rows_sum(const float* __restrict__ g, int ldg, int C, int64_t N, const int32_t* __restrict__ rowptr,
                                                       const int32_t* __restrict__ eperm, float* __restrict__ out, int ldo) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= N * C) return;
  const int64_t n = i / C;
  const int c = (int)(i - n * C);
  float a = 0.f;
  for (int p = rowptr[n]; p < rowptr[n + 1]; ++p) a += g[(int64_t)eperm[p] * ldg + c];
  out[n * ldo + c] = a;
}

// adjoint of k_pna_aggregate: dmsg [E, C] (every message row belongs to exactly one destination) and dself [N, C]
__global__ __launch_bounds__(256) void k_pna_aggregate_bwd(const float* __restrict__ msg, int ldm, int C, int64_t N,
                                                           const int32_t* __restrict__ rowptr, const int32_t* __restrict__ eperm,
                                                           float avg_log, const float* __restrict__ dout, int ldo, int has_self,
                                                           float* __restrict__ dmsg, float* __restrict__ dself) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= N * C) return;
  const int64_t n = i / C;
  const int c = (int)(i - n * C);
  const int lo = rowptr[n], hi = rowptr[n + 1];
  const float* go = dout + n * ldo;
  int off = 0;
  if (has_self) { dself[n * C + c] = go[c]; off = C; }
  const int D = hi - lo;
  if (D == 0) return;
  float s1 = 0.f, s2 = 0.f, mx = -INFINITY, mn = INFINITY;
  int imx = lo, imn = lo;
  for (int e = lo; e < hi; ++e) {
    const float v = msg[(int64_t)eperm[e] * ldm + c];
    s1 += v;
    s2 += v * v;
    if (v > mx) { mx = v; imx = e; }            // the first maximum / minimum in edge order takes the gradient (torch.max / min)
    if (v < mn) { mn = v; imn = e; }
  }
  const float inv = 1.0f / (float)D;
  const float mean = s1 * inv;
  const float vraw = s2 * inv - mean * mean;
  const float sd = sqrtf(fmaxf(vraw, 0.f) + 1e-5f);
  const float logd = logf((float)D + 1.0f);
  const float amp = logd / avg_log, att = avg_log / logd;
  float da[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) da[k] = go[off + k * C + c] + go[off + (4 + k) * C + c] * amp + go[off + (8 + k) * C + c] * att;
  const float dvar = vraw > 0.f ? da[3] / (2.0f * sd) : 0.f;       // relu(E[x^2] - E[x]^2) passes no gradient where it clamps
  for (int e = lo; e < hi; ++e) {
    const int64_t row = eperm[e];
    const float v = msg[row * ldm + c];
    float d = da[0] * inv + dvar * 2.0f * inv * (v - mean);
    if (e == imx) d += da[1];
    if (e == imn) d += da[2];
    dmsg[row * C + c] = d;
  }
}

// The same with a WAVE per (node, head): lane c owns channel c (C <= 64), the attention logits el = feat_j . attn_l, er = feat_i . attn_r
// are wave reductions, the two passes over the in-edges (maximum, then weights and the weighted sum) read coalesced rows.  One THREAD
// per (node, head) walking 59 channels serially was 176 us per layer on the shipped GAT (11 800 threads on a 256-CU part): 62 % of
// the net's forward.
__global__ __launch_bounds__(256) void k_gat_aggregate_wave(const float* __restrict__ feat, const float* __restrict__ attn_l,
                                                            const float* __restrict__ attn_r, const float* __restrict__ bias, int64_t N,
                                                            int H, int C, float slope, int relu, const int32_t* __restrict__ rowptr,
                                                            const int32_t* __restrict__ col, float* __restrict__ out,
                                                            float* __restrict__ lse) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= N * H) return;
  const int64_t n = i / H;
  const int h = (int)(i - n * H);
  const int d = H * C;
  const bool on = lane < C;
  const float al = on ? attn_l[h * C + lane] : 0.f, ar = on ? attn_r[h * C + lane] : 0.f;
  auto wsum = [](float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
  };
  const float er = wsum(on ? feat[n * d + h * C + lane] * ar : 0.f);
  const int lo = rowptr[n], hi = rowptr[n + 1];
  float m = -INFINITY;
  for (int e = lo; e < hi; ++e) {
    const float fj = on ? feat[(int64_t)col[e] * d + h * C + lane] : 0.f;
    float sc = wsum(fj * al) + er;
    sc = sc > 0.f ? sc : sc * slope;
    m = fmaxf(m, sc);
  }
  float acc = 0.f, z = 0.f;
  for (int e = lo; e < hi; ++e) {
    const float fj = on ? feat[(int64_t)col[e] * d + h * C + lane] : 0.f;
    float sc = wsum(fj * al) + er;
    sc = sc > 0.f ? sc : sc * slope;
    const float w = expf(sc - m);
    z += w;
    acc += w * fj;
  }
  if (on) {
    float v = (hi > lo ? acc / z : 0.f) + (bias ? bias[h * C + lane] : 0.f);
    if (relu) v = fmaxf(v, 0.f);
    out[n * d + h * C + lane] = v;
  }
  if (lse && lane == 0) lse[i] = hi > lo ? m + logf(z) : 0.f;
}

// adjoint of k_gat_aggregate, destination side — one thread per (node, head): with go = dout * [out > 0] and a_e = exp(s_e - lse),
//   d s_e = a_e (go . f_src(e) - go . (out - bias)),   d pre_e = d s_e * leaky'(pre_e),   d er_n = sum_e d pre_e.
// Writes go [N, H*C] (the bias gradient's rows, and what the source side needs), a_e and d pre_e per (edge id, head), d er [N, H].
__global__ __launch_bounds__(256) void k_gat_bwd_dst(const float* __restrict__ feat, const float* __restrict__ attn_l,
                                                     const float* __restrict__ attn_r, const float* __restrict__ bias,
                                                     const float* __restrict__ out, const float* __restrict__ lse,
                                                     const float* __restrict__ dout, int64_t N, int H, int C, float slope, int relu,
                                                     const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                     const int32_t* __restrict__ eperm, float* __restrict__ gob, float* __restrict__ alpha,
                                                     float* __restrict__ dpre, float* __restrict__ der) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= N * H) return;
  const int64_t n = i / H;
  const int h = (int)(i - n * H);
  const int d = H * C;
  const float* al = attn_l + h * C;
  const float* ar = attn_r + h * C;
  const float* fi = feat + n * d + h * C;
  float go[64];
  float gdo = 0.f, er = 0.f;
#pragma unroll
  for (int c = 0; c < 64; ++c) {
    go[c] = 0.f;
    if (c < C) {
      const float o = out[n * d + h * C + c];
      const float g = (!relu || o > 0.f) ? dout[n * d + h * C + c] : 0.f;
      go[c] = g;
      gob[n * d + h * C + c] = g;
      gdo += g * (o - (bias ? bias[h * C + c] : 0.f));
      er += fi[c] * ar[c];
    }
  }
  const float L = lse[n * H + h];
  float sum = 0.f;
  for (int e = rowptr[n]; e < rowptr[n + 1]; ++e) {
    const int64_t eid = eperm[e];
    const float* fj = feat + (int64_t)col[e] * d + h * C;
    float el = 0.f, gv = 0.f;
#pragma unroll
    for (int c = 0; c < 64; ++c)
      if (c < C) { el += fj[c] * al[c]; gv += go[c] * fj[c]; }
    const float pre = el + er;
    const float sc = pre > 0.f ? pre : pre * slope;
    const float a = expf(sc - L);
    const float dp = a * (gv - gdo) * (pre > 0.f ? 1.0f : slope);
    alpha[eid * H + h] = a;
    dpre[eid * H + h] = dp;
    sum += dp;
  }
  der[n * H + h] = sum;
}

// source side: d feat[j] = sum_{e: j -> i} a_e go_i + (sum_e d pre_e) attn_l + d er_j attn_r; d el [N, H] is kept for the attn_l gradient
__global__ __launch_bounds__(256) void k_gat_bwd_src(const float* __restrict__ attn_l, const float* __restrict__ attn_r,
                                                     const float* __restrict__ gob, const float* __restrict__ alpha,
                                                     const float* __restrict__ dpre, const float* __restrict__ der, int64_t N, int H, int C,
                                                     const int32_t* __restrict__ rev_rowptr, const int32_t* __restrict__ rev_col,
                                                     const int32_t* __restrict__ rev_eperm, float* __restrict__ dfeat,
                                                     float* __restrict__ del) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= N * H) return;
  const int64_t n = i / H;
  const int h = (int)(i - n * H);
  const int d = H * C;
  float acc[64];
#pragma unroll
  for (int c = 0; c < 64; ++c) acc[c] = 0.f;
  float sum = 0.f;
  for (int e = rev_rowptr[n]; e < rev_rowptr[n + 1]; ++e) {
    const int64_t eid = rev_eperm[e];
    const float* g = gob + (int64_t)rev_col[e] * d + h * C;
    const float a = alpha[eid * H + h];
    sum += dpre[eid * H + h];
#pragma unroll
    for (int c = 0; c < 64; ++c)
      if (c < C) acc[c] += a * g[c];
  }
  const float r = der[n * H + h];
  del[n * H + h] = sum;
#pragma unroll
  for (int c = 0; c < 64; ++c)
    if (c < C) dfeat[n * d + h * C + c] = acc[c] + sum * attn_l[h * C + c] + r * attn_r[h * C + c];
}

// adjoint of k_edge_attention, destination side: dQ [N, H*dk], dE [E, H*dk] and the per-(edge, head) scalars the source side needs —
// wv = s / (z + 1e-6) and dsc = d(score) — one thread per (node, head)
__global__ __launch_bounds__(256) void k_edge_attention_bwd_dst(const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
                                                                const float* __restrict__ Ee, const float* __restrict__ out,
                                                                const float* __restrict__ dout, int64_t N, int H, int dk,
                                                                const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                                const int32_t* __restrict__ eperm, float* __restrict__ dQ,
                                                                float* __restrict__ dE, float* __restrict__ wv, float* __restrict__ dsc) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= N * H) return;
  const int64_t n = i / H;
  const int h = (int)(i - n * H);
  const int d = H * dk;
  const float root = sqrtf((float)dk);
  float q[32], go[32], aq[32];
  float gdo = 0.f;
#pragma unroll
  for (int c = 0; c < 32; ++c) {
    const bool ok = c < dk;
    q[c] = ok ? Q[n * d + h * dk + c] : 0.f;
    go[c] = ok ? dout[n * d + h * dk + c] : 0.f;
    gdo += ok ? go[c] * out[n * d + h * dk + c] : 0.f;
    aq[c] = 0.f;
  }
  float z = 0.f;
  for (int e = rowptr[n]; e < rowptr[n + 1]; ++e) {
    const int64_t j = col[e], eid = eperm[e];
    const float* kr = K + j * d + h * dk;
    const float* er = Ee + eid * d + h * dk;
    float sc = 0.f;
#pragma unroll
    for (int c = 0; c < 32; ++c)
      if (c < dk) sc += ((kr[c] * q[c]) / root) * er[c];
    z += expf(fminf(fmaxf(sc, -5.f), 5.f));
  }
  const float r = 1.0f / (z + 1e-6f);
  for (int e = rowptr[n]; e < rowptr[n + 1]; ++e) {
    const int64_t j = col[e], eid = eperm[e];
    const float* kr = K + j * d + h * dk;
    const float* er = Ee + eid * d + h * dk;
    const float* vr = V + j * d + h * dk;
    float sc = 0.f, gv = 0.f;
#pragma unroll
    for (int c = 0; c < 32; ++c)
      if (c < dk) { sc += ((kr[c] * q[c]) / root) * er[c]; gv += go[c] * vr[c]; }
    const bool inside = sc > -5.f && sc < 5.f;
    const float s = expf(fminf(fmaxf(sc, -5.f), 5.f));
    const float ds = r * (gv - gdo);                              // d s_e
    const float dscore = inside ? ds * s : 0.f;
    wv[eid * H + h] = s * r;
    dsc[eid * H + h] = dscore;
    float* der = dE + eid * d + h * dk;
#pragma unroll
    for (int c = 0; c < 32; ++c)
      if (c < dk) {
        const float kq = kr[c] / root;
        aq[c] += dscore * kq * er[c];
        der[c] = dscore * kq * q[c];
      }
  }
#pragma unroll
  for (int c = 0; c < 32; ++c)
    if (c < dk) dQ[n * d + h * dk + c] = aq[c];
}
// source side over the reverse CSR (rcol = destination node, rperm = edge id): dK[j] = sum_out dsc * Q[dst] * E[e] / sqrt(dk),
// dV[j] = sum_out wv * dout[dst]
__global__ __launch_bounds__(256) void k_edge_attention_bwd_src(const float* __restrict__ Q, const float* __restrict__ Ee, const float* __restrict__ dout,
                                                                const float* __restrict__ wv, const float* __restrict__ dsc, int64_t N,
                                                                int H, int dk, const int32_t* __restrict__ rrow, const int32_t* __restrict__ rcol,
                                                                const int32_t* __restrict__ rperm, float* __restrict__ dK, float* __restrict__ dV) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= N * H) return;
  const int64_t n = i / H;
  const int h = (int)(i - n * H);
  const int d = H * dk;
  const float root = sqrtf((float)dk);
  float ak[32], av[32];
#pragma unroll
  for (int c = 0; c < 32; ++c) { ak[c] = 0.f; av[c] = 0.f; }
  for (int e = rrow[n]; e < rrow[n + 1]; ++e) {
    const int64_t t = rcol[e], eid = rperm[e];
    const float w = wv[eid * H + h], ds = dsc[eid * H + h];
    const float* qr = Q + t * d + h * dk;
    const float* er = Ee + eid * d + h * dk;
    const float* gr = dout + t * d + h * dk;
#pragma unroll
    for (int c = 0; c < 32; ++c)
      if (c < dk) { ak[c] += ds * (qr[c] / root) * er[c]; av[c] += w * gr[c]; }
  }
#pragma unroll
  for (int c = 0; c < 32; ++c)
    if (c < dk) { dK[n * d + h * dk + c] = ak[c]; dV[n * d + h * dk + c] = av[c]; }
}

// dx = dy * act'(x): act 1 ReLU, 2 LeakyReLU(slope) (x = the PRE-activation), times an optional row scale
__global__ __launch_bounds__(256) void k_act_bwd(const float* __restrict__ x, const float* __restrict__ dy, int64_t R, int C,
                                                 const float* __restrict__ rowscale, int act, float slope, float* __restrict__ dx) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= R * C) return;
  float g = dy[i];
  if (act == 1) g = x[i] > 0.f ? g : 0.f;
  else if (act == 2) g = x[i] > 0.f ? g : g * slope;
  if (rowscale) g *= rowscale[i / C];
  dx[i] = g;
}

}  // namespace sn

using namespace sn;

extern "C" int sn_pna_aggregate_f32(const float* msg, int ldm, const float* hself, int ldh, int C, int64_t N, const int32_t* rowptr,
                                    const int32_t* eperm, float avg_log, float* out, int ldo, void* stream) {
  SN_REQUIRE(msg && rowptr && eperm && out && C > 0 && N >= 0 && ldm >= C && avg_log > 0.f, "sn_pna_aggregate_f32: bad arguments");
  SN_REQUIRE(ldo >= (hself ? 13 : 12) * C && (!hself || ldh >= C), "sn_pna_aggregate_f32: output rows too narrow");
  if (N == 0) return SN_OK;
  hipLaunchKernelGGL(k_pna_aggregate, dim3((unsigned)cdiv(N * C, 256)), dim3(256), 0, (hipStream_t)stream, msg, ldm, hself, ldh, C, N,
                     rowptr, eperm, avg_log, out, ldo);
  SN_CHECK_LAUNCH("sn_pna_aggregate_f32");
  return SN_OK;
}

extern "C" int sn_pna_aggregate_gather_f32(const float* Ps, int ldps, const float* Pd, int ldpd, const float* Qe, int ldq, const float* hself,
                                           int ldh, int C, int64_t N, const int32_t* rowptr, const int32_t* col, const int32_t* eperm,
                                           float avg_log, float* out, int ldo, int tower_width, void* stream) {
  SN_REQUIRE(Ps && Pd && Qe && rowptr && col && eperm && out && C > 0 && N >= 0 && ldps >= C && ldpd >= C && ldq >= C && avg_log > 0.f,
             "sn_pna_aggregate_gather_f32: bad arguments");
  SN_REQUIRE(ldo >= (hself ? 13 : 12) * C && (!hself || ldh >= C), "sn_pna_aggregate_gather_f32: output rows too narrow");
  SN_REQUIRE(tower_width == 0 || (tower_width > 0 && C % tower_width == 0 && hself), "sn_pna_aggregate_gather_f32: tower_width must divide C (and needs hself)");
  if (N == 0) return SN_OK;
  hipLaunchKernelGGL(k_pna_aggregate_gather, dim3((unsigned)cdiv(N * C, 256)), dim3(256), 0, (hipStream_t)stream, Ps, ldps, Pd, ldpd, Qe, ldq,
                     hself, ldh, C, N, rowptr, col, eperm, avg_log, out, ldo, tower_width);
  SN_CHECK_LAUNCH("sn_pna_aggregate_gather_f32");
  return SN_OK;
}

extern "C" int sn_grouped_linear_f32(const float* x, int ldx, int64_t R, int G, int din, int dout, const float* W, const float* bias,
                                     const float* rowscale, const float* scale, const float* shift, float* y, int ldy, void* stream) {
  SN_REQUIRE(x && W && y && R >= 0 && G >= 1 && din >= 4 && din <= 256 && din % 4 == 0 && dout >= 1 && dout <= 16,
             "sn_grouped_linear_f32: bad arguments (din a multiple of 4 up to 256, dout <= 16)");
  SN_REQUIRE(ldx >= G * din && ldx % 4 == 0 && ldy >= G * dout && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0,
             "sn_grouped_linear_f32: rows must be 16-byte aligned");
  SN_REQUIRE((scale != nullptr) == (shift != nullptr), "sn_grouped_linear_f32: scale / shift go together");
  if (R == 0) return SN_OK;
  const int64_t tasks = cdiv(R, 16) * G;
  hipLaunchKernelGGL(k_grouped_linear, dim3((unsigned)cdiv(tasks, 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, R, G, din, dout, W, bias, rowscale,
                     scale, shift, y, ldy);
  SN_CHECK_LAUNCH("sn_grouped_linear_f32");
  return SN_OK;
}

extern "C" int sn_edge_attention_f32(const float* Q, const float* K, const float* V, const float* Ee, int64_t N, int heads, int dk,
                                     const int32_t* rowptr, const int32_t* col, const int32_t* eperm, float* out, void* stream) {
  SN_REQUIRE(Q && K && V && Ee && rowptr && col && eperm && out && N >= 0 && heads > 0, "sn_edge_attention_f32: bad arguments");
  SN_REQUIRE(dk >= 1 && dk <= 32, "sn_edge_attention_f32: head width %d not in [1, 32]", dk);
  if (N == 0) return SN_OK;
  hipLaunchKernelGGL(k_edge_attention, dim3((unsigned)cdiv(N * heads, 256)), dim3(256), 0, (hipStream_t)stream, Q, K, V, heads * dk, Ee,
                     heads * dk, N, heads, dk, rowptr, col, eperm, out);
  SN_CHECK_LAUNCH("sn_edge_attention_f32");
  return SN_OK;
}

extern "C" int sn_edge_attention_strided_f32(const float* Q, const float* K, const float* V, int ldq, const float* Ee, int lde, int64_t N,
                                             int heads, int dk, const int32_t* rowptr, const int32_t* col, const int32_t* eperm, float* out,
                                             void* stream) {
  SN_REQUIRE(Q && K && V && Ee && rowptr && col && eperm && out && N >= 0 && heads > 0, "sn_edge_attention_strided_f32: bad arguments");
  SN_REQUIRE(dk >= 1 && dk <= 32, "sn_edge_attention_strided_f32: head width %d not in [1, 32]", dk);
  SN_REQUIRE(ldq >= heads * dk && lde >= heads * dk, "sn_edge_attention_strided_f32: row strides shorter than heads * dk");
  if (N == 0) return SN_OK;
  hipLaunchKernelGGL(k_edge_attention, dim3((unsigned)cdiv(N * heads, 256)), dim3(256), 0, (hipStream_t)stream, Q, K, V, ldq, Ee, lde, N, heads,
                     dk, rowptr, col, eperm, out);
  SN_CHECK_LAUNCH("sn_edge_attention_strided_f32");
  return SN_OK;
}

extern "C" int sn_pointwise_f32(const float* x, int ldx, int64_t R, int C, const float* rowscale, const float* scale, const float* shift,
                                int act, float slope, const float* residual, int ldr, float* y, int ldy, void* stream) {
  SN_REQUIRE(x && y && C > 0 && R >= 0 && ldx >= C && ldy >= C && act >= 0 && act <= 3, "sn_pointwise_f32: bad arguments");
  SN_REQUIRE((scale == nullptr) == (shift == nullptr), "sn_pointwise_f32: scale and shift come together");
  SN_REQUIRE(!residual || ldr >= C, "sn_pointwise_f32: residual rows too narrow");
  if (R == 0) return SN_OK;
  hipLaunchKernelGGL(k_pointwise, dim3((unsigned)cdiv(R * C, 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, R, C, rowscale, scale, shift,
                     act, slope, residual, ldr, y, ldy);
  SN_CHECK_LAUNCH("sn_pointwise_f32");
  return SN_OK;
}

extern "C" int sn_edge_rows_sum_f32(const float* g, int ldg, int C, int64_t N, const int32_t* rowptr, const int32_t* eperm, float* out, int ldo,
                                    void* stream) {
  SN_REQUIRE(g && rowptr && eperm && out && C > 0 && N >= 0 && ldg >= C && ldo >= C, "sn_edge_rows_sum_f32: bad arguments");
  if (N == 0) return SN_OK;
  hipLaunchKernelGGL(k_edge_rows_sum, dim3((unsigned)cdiv(N * C, 256)), dim3(256), 0, (hipStream_t)stream, g, ldg, C, N, rowptr, eperm, out, ldo);
  SN_CHECK_LAUNCH("sn_edge_rows_sum_f32");
  return SN_OK;
}

extern "C" int sn_pna_aggregate_bwd_f32(const float* msg, int ldm, int C, int64_t N, const int32_t* rowptr, const int32_t* eperm, float avg_log,
                                        const float* dout, int ldo, float* dmsg, float* dself, void* stream) {
  SN_REQUIRE(msg && rowptr && eperm && dout && dmsg && C > 0 && N >= 0 && ldm >= C && avg_log > 0.f, "sn_pna_aggregate_bwd_f32: bad arguments");
  SN_REQUIRE(ldo >= (dself ? 13 : 12) * C, "sn_pna_aggregate_bwd_f32: gradient rows too narrow");
  if (N == 0) return SN_OK;
  hipLaunchKernelGGL(k_pna_aggregate_bwd, dim3((unsigned)cdiv(N * C, 256)), dim3(256), 0, (hipStream_t)stream, msg, ldm, C, N, rowptr, eperm,
                     avg_log, dout, ldo, dself ? 1 : 0, dmsg, dself);
  SN_CHECK_LAUNCH("sn_pna_aggregate_bwd_f32");
  return SN_OK;
}

extern "C" int sn_edge_attention_bwd_f32(const float* Q, const float* K, const float* V, const float* Ee, const float* out, const float* dout,
                                         int64_t N, int64_t E, int heads, int dk, const int32_t* rowptr, const int32_t* col,
                                         const int32_t* eperm, const int32_t* rev_rowptr, const int32_t* rev_col, const int32_t* rev_eperm,
                                         float* dQ, float* dK, float* dV, float* dE, float* scratch /* [2*E*heads] */, void* stream) {
  SN_REQUIRE(Q && K && V && Ee && out && dout && rowptr && rev_rowptr && dQ && dK && dV && dE && scratch && N >= 0 && E >= 0 && heads > 0,
             "sn_edge_attention_bwd_f32: bad arguments");
  SN_REQUIRE(dk >= 1 && dk <= 32, "sn_edge_attention_bwd_f32: head width %d not in [1, 32]", dk);
  if (N == 0) return SN_OK;
  SN_REQUIRE(E == 0 || (col && eperm && rev_col && rev_eperm), "sn_edge_attention_bwd_f32: null edge arrays");
  hipStream_t st = (hipStream_t)stream;
  float* wv = scratch;
  float* dsc = scratch + E * heads;
  const dim3 grid((unsigned)cdiv(N * heads, 256));
  hipLaunchKernelGGL(k_edge_attention_bwd_dst, grid, dim3(256), 0, st, Q, K, V, Ee, out, dout, N, heads, dk, rowptr, col, eperm, dQ, dE, wv, dsc);
  hipLaunchKernelGGL(k_edge_attention_bwd_src, grid, dim3(256), 0, st, Q, Ee, dout, (const float*)wv, (const float*)dsc, N, heads, dk, rev_rowptr,
                     rev_col, rev_eperm, dK, dV);
  SN_CHECK_LAUNCH("sn_edge_attention_bwd_f32");
  return SN_OK;
}

extern "C" int sn_act_bwd_f32(const float* x, const float* dy, int64_t R, int C, const float* rowscale, int act, float slope, float* dx,
                              void* stream) {
  SN_REQUIRE(dy && dx && R >= 0 && C > 0 && act >= 0 && act <= 2 && (act == 0 || x), "sn_act_bwd_f32: bad arguments");
  if (R == 0) return SN_OK;
  hipLaunchKernelGGL(k_act_bwd, dim3((unsigned)cdiv(R * C, 256)), dim3(256), 0, (hipStream_t)stream, x, dy, R, C, rowscale, act, slope, dx);
  SN_CHECK_LAUNCH("sn_act_bwd_f32");
  return SN_OK;
}

extern "C" int sn_gat_aggregate_f32(const float* feat, const float* attn_l, const float* attn_r, const float* bias, int64_t N, int heads, int C,
                                    float negative_slope, int relu, const int32_t* rowptr, const int32_t* col, float* out, float* lse,
                                    void* stream) {
  SN_REQUIRE(feat && attn_l && attn_r && rowptr && out && N >= 0 && heads > 0, "sn_gat_aggregate_f32: bad arguments");
  SN_REQUIRE(C >= 1 && C <= 64, "sn_gat_aggregate_f32: head width %d not in [1, 64]", C);
  if (N == 0) return SN_OK;
  hipLaunchKernelGGL(k_gat_aggregate_wave, dim3((unsigned)cdiv(N * heads, 4)), dim3(256), 0, (hipStream_t)stream, feat, attn_l, attn_r, bias, N,
                     heads, C, negative_slope, relu, rowptr, col, out, lse);
  SN_CHECK_LAUNCH("sn_gat_aggregate_f32");
  return SN_OK;
}

extern "C" int sn_gat_aggregate_bwd_f32(const float* feat, const float* attn_l, const float* attn_r, const float* bias, const float* out,
                                        const float* lse, const float* dout, int64_t N, int64_t E, int heads, int C, float negative_slope,
                                        int relu, const int32_t* rowptr, const int32_t* col, const int32_t* eperm, const int32_t* rev_rowptr,
                                        const int32_t* rev_col, const int32_t* rev_eperm, float* dfeat, float* dbias_rows, float* d_el,
                                        float* d_er, float* scratch, void* stream) {
  SN_REQUIRE(feat && attn_l && attn_r && out && lse && dout && rowptr && rev_rowptr && dfeat && dbias_rows && d_el && d_er && scratch &&
                 N >= 0 && E >= 0 && heads > 0,
             "sn_gat_aggregate_bwd_f32: bad arguments");
  SN_REQUIRE(C >= 1 && C <= 64, "sn_gat_aggregate_bwd_f32: head width %d not in [1, 64]", C);
  if (N == 0) return SN_OK;
  SN_REQUIRE(E == 0 || (col && eperm && rev_col && rev_eperm), "sn_gat_aggregate_bwd_f32: null edge arrays");
  hipStream_t st = (hipStream_t)stream;
  float* alpha = scratch;
  float* dpre = scratch + E * heads;
  const dim3 grid((unsigned)cdiv(N * heads, 256));
  hipLaunchKernelGGL(k_gat_bwd_dst, grid, dim3(256), 0, st, feat, attn_l, attn_r, bias, out, lse, dout, N, heads, C, negative_slope, relu, rowptr,
                     col, eperm, dbias_rows, alpha, dpre, d_er);
  hipLaunchKernelGGL(k_gat_bwd_src, grid, dim3(256), 0, st, attn_l, attn_r, (const float*)dbias_rows, (const float*)alpha, (const float*)dpre,
                     (const float*)d_er, N, heads, C, rev_rowptr, rev_col, rev_eperm, dfeat, d_el);
  SN_CHECK_LAUNCH("sn_gat_aggregate_bwd_f32");
  return SN_OK;
}
