// ign_mlp.hip — everything of IGN2to1.forward AFTER the 2->1 contractions (LearningFilters/ign.py:29-39), eval mode, in ONE launch.
//
//   o [b, n, 5]  (contractions_2_to_1 of the projector stack: sn_ign_contract_2to1_f32 / sn_ign_contract_eigvecs_f32)
//   h0 = bn0(relu(W0 o + b0))                                   layer_2_to_1 (:88-128) as a Linear over the 5 basis channels
//   h1 = bn1(relu(W1a h0 + W1b mean_n(h0) + b1))                layer_1_to_1 (:174-214): identity block + mean block
//   h2 = bn2(relu(W2a h1 + W2b mean_n(h1) + b2))
//   y  = fc2(relu(fc1(h2)))        written TRANSPOSED: y[b, out, n]   (ign.py:36-39: x.transpose(2,1) ... back)
//
// One workgroup per matrix (8 waves); a wave owns every eighth 16-row tile and keeps its rows' H channels in registers for the whole
// network in the MFMA operand layout, so the four H x H Linears are 4 (H = 16) or 16 (H = 32) v_mfma_f32_16x16x4_f32 per tile with the
// weight fragments loaded once per layer; the only cross-row quantity, the per-matrix column mean, is reduced over the workgroup (DPP
// row sums, the 8 waves through LDS).  The layer-at-a-time path moved the [b*n, H] activations through HBM nine times per
// multiplicity group (0.5 ms of the 1.15 ms BasisNet forward on the 32x32 grid); this reads o and writes y.
// (Two VALU formulations were measured first: weights as LDS broadcasts — 256 ds_read_b128 per row and layer, LDS-delivery bound,
// 250 us per launch on average — and weights through uniform global loads, which the compiler does not scalarise: 1.2 ms.)
#include "common.hpp"

namespace sn {
namespace {

constexpr int IGN_T = 512;           // threads per workgroup (8 waves: 256 VGPRs per lane for the rows' channels)

struct IgnMlp {
  const float* o; int64_t b; int n; int O;
  const float *w0, *b0, *s0, *t0;                    // [H,5], [H] x3
  const float *w1a, *w1b, *b1, *s1, *t1;             // [H,H] x2, [H] x3
  const float *w2a, *w2b, *b2, *s2, *t2;
  const float *f1w, *f1b, *f2w, *f2b;                // [H,H], [H], [O,H], [O]
  float* y;
};

__device__ __forceinline__ float row16_sum(float v) {      // sum over the 16 lanes of a DPP row (the 16 rows of a tile)
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true));
  return v;
}
__device__ __forceinline__ f32x4 ldg4(const float* q) { const float4 t = *reinterpret_cast<const float4*>(q); return f32x4{t.x, t.y, t.z, t.w}; }
__device__ __forceinline__ f32x4 relu4(f32x4 v) { return f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)}; }

// Layout (the library's MFMA convention, common.hpp): a wave works on 16-row tiles; lane (lr = lane & 15, g = lane >> 4) holds row lr's
// channels 16 kk + 4 g + t of a tile as f32x4 in[kk]; a weight fragment for output tile ot is W[16 ot + lr][16 kk + 4 g + t] — a float4
// straight from the row-major parameter; the accumulator of v_mfma_f32_16x16x4_f32 comes back in the operand layout, so the layers chain
// in registers.  TPW tiles per wave (tile wave + 8 i), NT = H / 16.
template <int H, int TPW>
__global__ __launch_bounds__(IGN_T) void k_ign_mlp(IgnMlp p) {
  constexpr int NT = H / 16, NW = IGN_T / 64;
  __shared__ float RED[NW][H];
  __shared__ float MEAN[H], BB[H];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, g = lane >> 4;
  const int64_t mat = blockIdx.x;
  const int n = p.n;
  f32x4 h[TPW][NT];
  bool valid[TPW];
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

  // ---- layer 0: Linear(5 -> H) over the contraction basis (k padded to 16), ReLU, BatchNorm
  {
    f32x4 w0[NT], b0[NT], s0[NT], t0[NT];
#pragma unroll
    for (int ot = 0; ot < NT; ++ot) {
      const float* wr = p.w0 + (16 * ot + lr) * 5;
      w0[ot] = g == 0 ? f32x4{wr[0], wr[1], wr[2], wr[3]} : (g == 1 ? f32x4{wr[4], 0.f, 0.f, 0.f} : zero);
      b0[ot] = ldg4(p.b0 + 16 * ot + 4 * g); s0[ot] = ldg4(p.s0 + 16 * ot + 4 * g); t0[ot] = ldg4(p.t0 + 16 * ot + 4 * g);
    }
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      const int row = (wave + NW * i) * 16 + lr;
      valid[i] = row < n;
      f32x4 in = zero;
      if (valid[i]) {
        const float* op = p.o + (mat * n + row) * 5;
        if (g == 0) in = f32x4{op[0], op[1], op[2], op[3]};
        else if (g == 1) in = f32x4{op[4], 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int ot = 0; ot < NT; ++ot) {
        f32x4 acc = zero;
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = mfma16(w0[ot][t], in[t], acc);
        const f32x4 v = relu4(acc + b0[ot]) * s0[ot] + t0[ot];
        h[i][ot] = valid[i] ? v : zero;
      }
    }
  }
  // column means of h over the matrix's n rows, then the block bias  wb . mean -> BB
  auto mean_and_block_bias = [&](const float* __restrict__ wb) {
    __syncthreads();                  // (the previous layer is done with BB)
#pragma unroll
    for (int ot = 0; ot < NT; ++ot) {
      f32x4 s = zero;
#pragma unroll
      for (int i = 0; i < TPW; ++i) s += h[i][ot];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float c = row16_sum(s[r]);
        if (lr == 0) RED[wave][16 * ot + 4 * g + r] = c;
      }
    }
    __syncthreads();
    if (tid < H) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) s += RED[w][tid];
      MEAN[tid] = s / (float)n;
    }
    __syncthreads();
    if (tid < H) {
      float s = 0.f;
      for (int d = 0; d < H; ++d) s += wb[tid * H + d] * MEAN[d];
      BB[tid] = s;
    }
    __syncthreads();
  };
  // h <- [bn](relu(wa h + bias [+ BB])) on every tile of the wave
  auto layer = [&](const float* __restrict__ wa, const float* __restrict__ bias, const float* __restrict__ sc,
                   const float* __restrict__ sh, bool block) {
    f32x4 wf[NT][NT], bv[NT], sv[NT], tv[NT];
#pragma unroll
    for (int ot = 0; ot < NT; ++ot) {
#pragma unroll
      for (int kk = 0; kk < NT; ++kk) wf[ot][kk] = ldg4(wa + (16 * ot + lr) * H + 16 * kk + 4 * g);
      bv[ot] = ldg4(bias + 16 * ot + 4 * g);
      if (block) bv[ot] += f32x4{BB[16 * ot + 4 * g], BB[16 * ot + 4 * g + 1], BB[16 * ot + 4 * g + 2], BB[16 * ot + 4 * g + 3]};
      sv[ot] = sc ? ldg4(sc + 16 * ot + 4 * g) : f32x4{1.f, 1.f, 1.f, 1.f};
      tv[ot] = sc ? ldg4(sh + 16 * ot + 4 * g) : zero;
    }
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      f32x4 out[NT];
#pragma unroll
      for (int ot = 0; ot < NT; ++ot) {
        f32x4 acc = zero;
#pragma unroll
        for (int kk = 0; kk < NT; ++kk)
#pragma unroll
          for (int t = 0; t < 4; ++t) acc = mfma16(wf[ot][kk][t], h[i][kk][t], acc);
        const f32x4 v = relu4(acc + bv[ot]) * sv[ot] + tv[ot];
        out[ot] = valid[i] ? v : zero;
      }
#pragma unroll
      for (int ot = 0; ot < NT; ++ot) h[i][ot] = out[ot];
    }
  };
  mean_and_block_bias(p.w1b);
  layer(p.w1a, p.b1, p.s1, p.t1, true);
  mean_and_block_bias(p.w2b);
  layer(p.w2a, p.b2, p.s2, p.t2, true);
  layer(p.f1w, p.f1b, nullptr, nullptr, false);
  // ---- fc2 -> y[b, O, n]   (output channels 16 ot + 4 g + r of row lr: for a fixed r the 16 lanes of a group write 16 consecutive rows)
  const int O = p.O;
  for (int ot = 0; 16 * ot < O; ++ot) {
    f32x4 wf[NT];
    const int oc = 16 * ot + lr;
#pragma unroll
    for (int kk = 0; kk < NT; ++kk) wf[kk] = oc < O ? ldg4(p.f2w + oc * H + 16 * kk + 4 * g) : zero;
    f32x4 bv = zero;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (p.f2b && 16 * ot + 4 * g + r < O) bv[r] = p.f2b[16 * ot + 4 * g + r];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      f32x4 acc = zero;
#pragma unroll
      for (int kk = 0; kk < NT; ++kk)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = mfma16(wf[kk][t], h[i][kk][t], acc);
      acc += bv;
      if (valid[i]) {
        const int row = (wave + NW * i) * 16 + lr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = 16 * ot + 4 * g + r;
          if (o < O) p.y[(mat * O + o) * n + row] = acc[r];
        }
      }
    }
  }
}

template <int H, int TPW>
int launch(const IgnMlp& p, hipStream_t st) {
  hipLaunchKernelGGL((k_ign_mlp<H, TPW>), dim3((unsigned)p.b), dim3(IGN_T), 0, st, p);
  return SN_OK;
}
template <int H>
int launch_h(const IgnMlp& p, hipStream_t st) {
  const int tpw = ((p.n + 15) / 16 + IGN_T / 64 - 1) / (IGN_T / 64);
  if (tpw <= 1) return launch<H, 1>(p, st);
  if (tpw <= 2) return launch<H, 2>(p, st);
  if (tpw <= 4) return launch<H, 4>(p, st);
  return launch<H, 8>(p, st);
}

}  // namespace
}  // namespace sn

using namespace sn;

extern "C" int sn_ign_mlp_supported(int n, int H, int O) {
  return (H == 16 || H == 32) && n >= 1 && n <= 1024 && O >= 1 && O <= 32;
}

extern "C" int sn_ign_mlp_f32(const float* o, int64_t b, int n, int H, int O, const sn_ign_mlp_params* P, float* y, void* stream) {
  SN_REQUIRE(o && P && y && b >= 0, "sn_ign_mlp_f32: bad arguments");
  SN_REQUIRE(sn_ign_mlp_supported(n, H, O), "sn_ign_mlp_f32: hidden width 16 or 32, n <= 1024 rows per matrix, <= 32 output channels");
  const float* req[] = {P->w0, P->b0, P->s0, P->t0, P->w1a, P->w1b, P->b1, P->s1, P->t1, P->w2a, P->w2b, P->b2, P->s2, P->t2, P->fc1_w,
                        P->fc1_b, P->fc2_w};
  for (const float* q : req) SN_REQUIRE(q && (reinterpret_cast<uintptr_t>(q) & 15) == 0, "sn_ign_mlp_f32: missing parameter");
  if (b == 0) return SN_OK;
  IgnMlp p{o, b, n, O, P->w0, P->b0, P->s0, P->t0, P->w1a, P->w1b, P->b1, P->s1, P->t1, P->w2a, P->w2b, P->b2, P->s2, P->t2,
           P->fc1_w, P->fc1_b, P->fc2_w, P->fc2_b, y};
  const int rc = H == 16 ? launch_h<16>(p, (hipStream_t)stream) : launch_h<32>(p, (hipStream_t)stream);
  if (rc != SN_OK) return rc;
  SN_CHECK_LAUNCH("sn_ign_mlp_f32");
  return SN_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// EqDeepSetsEncoder (LearningFilters/models.py:58-113) behind its first Linear, ONE set (b = 1), eval / no-grad value, in one launch:
//   h = z (the first layer's lin1(x) + lin2(mean x), pre-activation);   for every further layer i:
//   h = relu(h);  [h = BatchNorm(h) with the statistics of the n rows (track_running_stats = False: always batch statistics)];
//   h = W1_i h + b1_i + W2_i mean_n(h) + b2_i          (no ReLU / BatchNorm after the last layer)
// One workgroup of 1024 threads; the [n, width] activations stay in LDS (n * width <= 16384); column means / variances are block
// reductions (two passes: mean, then sum (x - mean)^2, as sn_masked_colstats_f32 does).  The layer-at-a-time
// path spent ~10 launches of 5-20 us per layer on a [1024, 10] matrix.
namespace sn {
namespace {
constexpr int DS_T = 1024, DS_W = 32, DS_BUF = 16384;      // threads; widest layer; floats per activation buffer (n * width)

__global__ __launch_bounds__(DS_T) void k_deepsets_tail(const float* __restrict__ z, int n, sn_deepsets_tail_params P, float* __restrict__ y) {
  // the activations [n][width] live in LDS (two buffers); every step is a flat loop over its elements — (row, channel) pairs — so the
  // small widths (10 in the LearningFilters model) cost what they are, not a padded 32 x 32 tile per row
  extern __shared__ __align__(16) float ds_lds[];
  float* A = ds_lds;
  float* Bf = ds_lds + DS_BUF;
  __shared__ float WA[DS_W * DS_W], WB[DS_W * DS_W], V[6][DS_W], RED[32][DS_W + 1];
  const int tid = threadIdx.x;
  int din = P.width[0];
  const float inv_n = 1.0f / (float)n;
  if (P.split0) {
    // z = x [W1 ; W2]^T + [b1 ; b2]  ([n, 2 width0], ONE GEMM of the caller): the first layer is z[:, :w] + mean_rows(z[:, w:]) — the mean of
    // the set commutes with lin2, so the [n, F] column mean of the (wide) input is never formed
    const int c = tid & 31, rl = tid >> 5;
    float sm = 0.f;
    if (c < din)
      for (int r = rl; r < n; r += 32) sm += z[(int64_t)r * 2 * din + din + c];
    RED[rl][c] = sm;
    __syncthreads();
    if (tid < DS_W) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 32; ++q) t += RED[q][tid];
      V[0][tid] = t * inv_n;
    }
    __syncthreads();
    for (int i = tid; i < n * din; i += DS_T) { const int r = i / din, cc = i - r * din; A[i] = z[(int64_t)r * 2 * din + cc] + V[0][cc]; }
  } else {
    for (int i = tid; i < n * din; i += DS_T) A[i] = z[i];
  }
  // V[slot][c] = sum over the rows of f(A[r][c])   (32 row lanes per column, then a fixed-order fold)
  auto colsum = [&](int slot, int width, auto f) {
    __syncthreads();
    const int c = tid & 31, rl = tid >> 5;
    float s = 0.f;
    if (c < width)
      for (int r = rl; r < n; r += 32) s += f(A[r * width + c], c);
    RED[rl][c] = s;
    __syncthreads();
    if (tid < DS_W) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 32; ++q) t += RED[q][tid];
      V[slot][tid] = t;
    }
    __syncthreads();
  };
  for (int i = 1; i < P.n_layers; ++i) {
    const int dout = P.width[i];
    __syncthreads();
    for (int k = tid; k < dout * din; k += DS_T) { WA[k] = P.w1[i][k]; WB[k] = P.w2[i][k]; }
    if (tid < DS_W) {
      V[3][tid] = tid < dout ? P.b1[i][tid] + P.b2[i][tid] : 0.f;
      V[4][tid] = (P.use_bn && tid < din) ? P.gamma[i - 1][tid] : 1.f;
      V[5][tid] = (P.use_bn && tid < din) ? P.beta[i - 1][tid] : 0.f;
    }
    for (int k = tid; k < n * din; k += DS_T) A[k] = fmaxf(A[k], 0.f);
    colsum(0, din, [](float v, int) { return v; });                                     // V[0] = sum h
    if (P.use_bn) {
      colsum(1, din, [&](float v, int c) { const float d = v - V[0][c] * inv_n; return d * d; });      // V[1] = sum (h - mean)^2
      if (tid < DS_W && tid < din) {            // fold to scale | shift (in V[4] | V[5]); the normalised mean is then the shift + ...
        const float mean = V[0][tid] * inv_n, var = V[1][tid] * inv_n;
        const float sc = V[4][tid] / sqrtf(var + P.eps);
        V[5][tid] = V[5][tid] - mean * sc;
        V[4][tid] = sc;
      }
      __syncthreads();
      for (int k = tid; k < n * din; k += DS_T) { const int c = k % din; A[k] = A[k] * V[4][c] + V[5][c]; }
      colsum(0, din, [](float v, int) { return v; });                                   // V[0] = sum of the normalised rows
    }
    if (tid < DS_W) {                                                                   // V[2] = W2 mean + b1 + b2
      float s = V[3][tid];
      if (tid < dout)
        for (int d = 0; d < din; ++d) s += WB[tid * din + d] * (V[0][d] * inv_n);
      V[2][tid] = s;
    }
    __syncthreads();
    float* dst = (i == P.n_layers - 1) ? y : Bf;          // the last layer goes straight to global memory (it may be the widest)
    for (int k = tid; k < n * dout; k += DS_T) {
      const int r = k / dout, o = k - r * dout;
      float a = V[2][o];
      const float* hr = A + r * din;
      const float* wr = WA + o * din;
      for (int d = 0; d < din; ++d) a += wr[d] * hr[d];
      dst[k] = a;
    }
    __syncthreads();
    float* t = A; A = Bf; Bf = t;
    din = dout;
  }
}
}  // namespace
}  // namespace sn

extern "C" int sn_deepsets_tail_f32(const float* z, int n, const sn_deepsets_tail_params* P, float* y, void* stream) {
  SN_REQUIRE(z && P && y && n >= 1, "sn_deepsets_tail_f32: bad arguments");
  SN_REQUIRE(P->n_layers >= 2 && P->n_layers <= SN_DEEPSETS_MAX_LAYERS, "sn_deepsets_tail_f32: 2 .. %d layers", SN_DEEPSETS_MAX_LAYERS);
  for (int i = 0; i < P->n_layers; ++i) SN_REQUIRE(P->width[i] >= 1 && P->width[i] <= sn::DS_W, "sn_deepsets_tail_f32: layer widths up to 32");
  for (int i = 1; i < P->n_layers; ++i)
    SN_REQUIRE(P->w1[i] && P->w2[i] && P->b1[i] && P->b2[i] && (!P->use_bn || (P->gamma[i - 1] && P->beta[i - 1])), "sn_deepsets_tail_f32: missing parameter");
  int wmax = 0;
  for (int i = 0; i + 1 < P->n_layers; ++i) wmax = P->width[i] > wmax ? P->width[i] : wmax;
  SN_REQUIRE((int64_t)n * wmax <= sn::DS_BUF, "sn_deepsets_tail_f32: n * widest layer but the last must be <= 16384");
  const size_t lds = (size_t)2 * sn::DS_BUF * sizeof(float);
  static bool raised = false;
  if (!raised) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(sn::k_deepsets_tail), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return sn::fail(SN_ERR_LAUNCH, "sn_deepsets_tail_f32: cannot raise the dynamic LDS limit");
    raised = true;
  }
  hipLaunchKernelGGL(sn::k_deepsets_tail, dim3(1), dim3(sn::DS_T), lds, (hipStream_t)stream, z, n, *P, y);
  SN_CHECK_LAUNCH("sn_deepsets_tail_f32");
  return SN_OK;
}
