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
