// ops.hip — the layer-at-a-time kernels behind the C ABI (include/signnet_hip.h).
// One kernel per op class the reference's forward inherits from PyG / torch_scatter / DGL / ATen
// (SURVEY.md §2.1).  The fused whole-stage kernels (fused_*.hip) reuse the same GEMM convention.
#include "common.hpp"

namespace sn {

// ============================================================================ weight packing
// trans: W is the [d_in, d_out] matrix whose TRANSPOSE is packed (the input-gradient Linear dX = dY W reads the forward weight in place)
__global__ void k_pack_weight(const float* __restrict__ W, int d_out, int d_in, int ldw, int nto, int nti,
                              float* __restrict__ Wp, int trans) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t total = (int64_t)nto * nti * 256;
  if (idx >= total) return;
  int t = idx & 3;
  int lane = (idx >> 2) & 63;
  int64_t blk = idx >> 8;
  int kk = (int)(blk % nti), ot = (int)(blk / nti);
  int o = 16 * ot + (lane & 15);
  int k = 16 * kk + 4 * (lane >> 4) + t;
  Wp[idx] = (o < d_out && k < d_in) ? (trans ? W[(int64_t)k * ldw + o] : W[(int64_t)o * ldw + k]) : 0.f;
}

// ============================================================================ split-packed linear (bf16 x 3 + epilogue vectors)
// Layout consumed by wg_gemm_split (fused_common.hpp): per 16-output tile ot one chunk of
//   [kb < NKB][plane h,m,l][lane][8 bf16]   with k-slot s of lane (o = lane&15, g = lane>>4) = channel 32kb + 16(s>>2) + 4g + (s&3)
//   [e < 3][lane][4 f32] = vec_e[16 ot + 4g + t]
__global__ void k_pack_split(const float* __restrict__ W, int d_out, int d_in, int ldw, const float* __restrict__ e0,
                             const float* __restrict__ e1, const float* __restrict__ e2, int nto, int nkb,
                             unsigned char* __restrict__ dst) {
  const int nfe = 3 * nkb + SN_SPLIT_EPI;
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // one thread per (chunk, fragment, lane)
  if (idx >= (int64_t)nto * nfe * 64) return;
  const int lane = idx & 63;
  const int fr = (int)((idx >> 6) % nfe), ot = (int)((idx >> 6) / nfe);
  unsigned char* out = dst + ((int64_t)ot * nfe + fr) * 1024 + lane * 16;
  const int g = lane >> 4;
  if (fr < 3 * nkb) {
    const int kb = fr / 3, plane = fr - 3 * kb;
    const int o = 16 * ot + (lane & 15);
    unsigned short v[8];
    for (int s = 0; s < 8; ++s) {
      const int ch = 32 * kb + 16 * (s >> 2) + 4 * g + (s & 3);
      const float w = (o < d_out && ch < d_in) ? W[(int64_t)o * ldw + ch] : 0.f;
      const float h = __uint_as_float(__float_as_uint(w) & 0xffff0000u);
      const float r = w - h;
      const float m = __uint_as_float(__float_as_uint(r) & 0xffff0000u);
      const float l = r - m;
      const float pick = plane == 0 ? h : (plane == 1 ? m : l);
      v[s] = (unsigned short)(__float_as_uint(pick) >> 16);
    }
    uint4 q;
    q.x = v[0] | ((unsigned)v[1] << 16);
    q.y = v[2] | ((unsigned)v[3] << 16);
    q.z = v[4] | ((unsigned)v[5] << 16);
    q.w = v[6] | ((unsigned)v[7] << 16);
    *reinterpret_cast<uint4*>(out) = q;
  } else {
    const int e = fr - 3 * nkb;
    const float* vec = e == 0 ? e0 : (e == 1 ? e1 : e2);
    float4 q;
    float* qq = reinterpret_cast<float*>(&q);
    for (int t = 0; t < 4; ++t) {
      const int c = 16 * ot + 4 * g + t;
      qq[t] = (vec && c < d_out) ? vec[c] : 0.f;
    }
    *reinterpret_cast<float4*>(out) = q;
  }
}

// running_mean / running_var update of a train-mode BatchNorm1d forward (momentum m, unbiased variance),
// from the batch statistics of sn_masked_colstats_f32 (biased variance, row count on the device).
__global__ void k_bn_running_update(const float* __restrict__ mean, const float* __restrict__ var, const float* __restrict__ count,
                                    float momentum, int C, float* __restrict__ rmean, float* __restrict__ rvar) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float n = count[0];
  const float unb = n > 1.f ? var[c] * (n / (n - 1.f)) : var[c];
  rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean[c];
  rvar[c] = (1.f - momentum) * rvar[c] + momentum * unb;
}

// ============================================================================ BatchNorm(eval) folding
// scale = weight / sqrt(running_var + eps), shift = bias - running_mean * scale; zero padded to Cp.
__global__ void k_bn_fold(const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ rm,
                          const float* __restrict__ rv, float eps, int C, int Cp, float* __restrict__ scale,
                          float* __restrict__ shift) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= Cp) return;
  float sc = 0.f, sh = 0.f;
  if (c < C) {
    sc = (w ? w[c] : 1.f) / sqrtf(rv[c] + eps);
    sh = (b ? b[c] : 0.f) - rm[c] * sc;
  }
  scale[c] = sc;
  shift[c] = sh;
}

// ============================================================================ masked linear
struct LinArgs {
  const float* x; int ldx; int64_t R; int d_in;
  const float4* wp; int nti; int d_out; int nto;
  const float* bias; const int32_t* nvalid; int K; int flags;
  const float* scale; const float* shift; const float* res; int ldr;
  float* y; int ldy;
  const float* bbias = nullptr; int64_t bb_rows = 0; int ldbb = 0;     // SN_EPI_BLOCK_BIAS: + bbias[row / bb_rows][:], with the bias
};

// sum over the 16 lanes of a DPP row (the 16 rows of a tile for one lane group): four VALU+DPP steps
__device__ __forceinline__ float tile16_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));   // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true));   // row_mirror
  return v;
}

template <bool VEC>
__device__ __forceinline__ f32x4 load4(const float* __restrict__ p, int c0, int C) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (VEC) {
    if (c0 < C) { float4 t = *reinterpret_cast<const float4*>(p + c0); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) if (c0 + r < C) v[r] = p[c0 + r];
  }
  return v;
}
template <bool VEC>
__device__ __forceinline__ void store4(float* __restrict__ p, int c0, int C, f32x4 v) {
  if (VEC) {
    if (c0 < C) *reinterpret_cast<float4*>(p + c0) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) if (c0 + r < C) p[c0 + r] = v[r];
  }
}

// XV: x rows 16-byte aligned with d_in % 4 == 0.  YV: every per-channel vector and y/res rows
// 16-byte aligned with d_out % 4 == 0.
template <bool XV, bool YV>
__global__ __launch_bounds__(256) void k_linear(LinArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  if (tile * 16 >= a.R) return;  // wave-uniform
  const int64_t row = tile * 16 + (lane & 15);
  const int g = lane >> 4;
  const bool inr = row < a.R;
  bool valid = inr;
  if (inr && a.nvalid) {
    int64_t node = row / a.K;
    valid = (int)(row - node * a.K) < a.nvalid[node];
  }
  const float* xr = a.x + row * a.ldx;
  float* yr = a.y + row * a.ldy;
  // few rows: the output tiles are dealt to gridDim.y workgroups per row block (a [2 950, 236] x [236, 236] Linear was 47 workgroups
  // walking 15 output tiles each: 82 us on a fifth of the chip)
  const int otc = (a.nto + (int)gridDim.y - 1) / (int)gridDim.y;
  const int ot_lo = (int)blockIdx.y * otc, ot_hi = ot_lo + otc < a.nto ? ot_lo + otc : a.nto;
  if (__ballot(valid) == 0ull) {  // nothing valid in this tile: zeros
    if (inr) {
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      for (int ot = ot_lo; ot < ot_hi; ++ot) store4<YV>(yr, 16 * ot + 4 * g, a.d_out, z);
    }
    return;
  }
  f32x4 in[8];
  const bool single = a.nti <= 8;
  if (single) {
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      in[kk] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (kk < a.nti && valid) in[kk] = load4<XV>(xr, 16 * kk + 4 * g, a.d_in);
    }
  }
  for (int ot = ot_lo; ot < ot_hi; ++ot) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float4* wo = a.wp + (int64_t)ot * a.nti * 64;
    for (int kc = 0; kc < a.nti; kc += 8) {
      if (!single) {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          in[kk] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (kc + kk < a.nti && valid) in[kk] = load4<XV>(xr, 16 * (kc + kk) + 4 * g, a.d_in);
        }
      }
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        if (kc + kk < a.nti) {
          float4 w = wo[(kc + kk) * 64 + lane];
          acc = mfma16(w.x, in[kk][0], acc);
          acc = mfma16(w.y, in[kk][1], acc);
          acc = mfma16(w.z, in[kk][2], acc);
          acc = mfma16(w.w, in[kk][3], acc);
        }
      }
    }
    // ---- epilogue on Y[row][o0 .. o0+3]
    const int o0 = 16 * ot + 4 * g;
    f32x4 v = acc;
    if (!valid) {
      v = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
      if (a.flags & SN_EPI_BIAS) v += load4<YV>(a.bias, o0, a.d_out);
      if (a.flags & SN_EPI_BLOCK_BIAS) v += load4<YV>(a.bbias + (row / a.bb_rows) * a.ldbb, o0, a.d_out);
      if (a.flags & SN_EPI_RESIDUAL_PRE) v += load4<YV>(a.res + row * a.ldr, o0, a.d_out);
      if (a.flags & SN_EPI_RELU_PRE) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
      }
      if (a.flags & SN_EPI_AFFINE) {
        f32x4 sc = load4<YV>(a.scale, o0, a.d_out), sh = load4<YV>(a.shift, o0, a.d_out);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = v[r] * sc[r] + sh[r];
      }
      if (a.flags & SN_EPI_RELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
      }
      if (a.flags & SN_EPI_LEAKY) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : v[r] * 0.01f;
      }
      if (a.flags & SN_EPI_RESIDUAL) v += load4<YV>(a.res + row * a.ldr, o0, a.d_out);
    }
    if (inr) store4<YV>(yr, o0, a.d_out, v);
  }
}

// Deep and narrow (d_in >= 512, d_out <= 64) on few rows — e.g. the first layer of LearningFilters' DeepSets rho, [1024, 2048] x [2048, 20]:
// k_linear gives every wave a 16-row tile and the whole K loop (128 dependent 16-wide steps, 32 workgroups on the chip: 71 us).  Here
// the four waves of a workgroup share ONE 16-row tile and split the K tiles round-robin; the partial accumulators meet in LDS and wave 0
// adds them in wave order (deterministic) and runs k_linear's epilogue.  grid = row tiles.
template <bool XV, bool YV>
__global__ __launch_bounds__(256) void k_linear_ksplit(LinArgs a) {
  __shared__ float4 part[4][4][64];                  // [wave][output tile][lane]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
  const int64_t row = (int64_t)blockIdx.x * 16 + (lane & 15);
  const bool inr = row < a.R;
  bool valid = inr;
  if (inr && a.nvalid) {
    const int64_t node = row / a.K;
    valid = (int)(row - node * a.K) < a.nvalid[node];
  }
  const float* xr = a.x + row * a.ldx;
  f32x4 acc[4];
#pragma unroll
  for (int ot = 0; ot < 4; ++ot) acc[ot] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int kk = wave; kk < a.nti; kk += 4) {
    f32x4 in = {0.f, 0.f, 0.f, 0.f};
    if (valid) in = load4<XV>(xr, 16 * kk + 4 * g, a.d_in);
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) {
      if (ot < a.nto) {
        const float4 w = a.wp[((int64_t)ot * a.nti + kk) * 64 + lane];
        acc[ot] = mfma16(w.x, in[0], acc[ot]);
        acc[ot] = mfma16(w.y, in[1], acc[ot]);
        acc[ot] = mfma16(w.z, in[2], acc[ot]);
        acc[ot] = mfma16(w.w, in[3], acc[ot]);
      }
    }
  }
#pragma unroll
  for (int ot = 0; ot < 4; ++ot) part[wave][ot][lane] = make_float4(acc[ot][0], acc[ot][1], acc[ot][2], acc[ot][3]);
  __syncthreads();
  if (wave != 0) return;
  float* yr = a.y + row * a.ldy;
  for (int ot = 0; ot < a.nto; ++ot) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 4; ++w) { const float4 t = part[w][ot][lane]; v += f32x4{t.x, t.y, t.z, t.w}; }
    const int o0 = 16 * ot + 4 * g;
    if (!valid) {
      v = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
      if (a.flags & SN_EPI_BIAS) v += load4<YV>(a.bias, o0, a.d_out);
      if (a.flags & SN_EPI_BLOCK_BIAS) v += load4<YV>(a.bbias + (row / a.bb_rows) * a.ldbb, o0, a.d_out);
      if (a.flags & SN_EPI_RESIDUAL_PRE) v += load4<YV>(a.res + row * a.ldr, o0, a.d_out);
      if (a.flags & SN_EPI_RELU_PRE) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
      }
      if (a.flags & SN_EPI_AFFINE) {
        const f32x4 sc = load4<YV>(a.scale, o0, a.d_out), sh = load4<YV>(a.shift, o0, a.d_out);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = v[r] * sc[r] + sh[r];
      }
      if (a.flags & SN_EPI_RELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
      }
      if (a.flags & SN_EPI_LEAKY) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : v[r] * 0.01f;
      }
      if (a.flags & SN_EPI_RESIDUAL) v += load4<YV>(a.res + row * a.ldr, o0, a.d_out);
    }
    if (inr) store4<YV>(yr, o0, a.d_out, v);
  }
}

// Large row counts: the packed weight is staged ONCE per workgroup in LDS (k_linear fetches it from L2 per wave and per output
// tile: at 47 k rows x 128 x 128 the fp32 matrix pipe was a third busy) and every wave walks a grid-stride sequence of 16-row tiles
// with the whole tile's operand in registers; two output tiles are accumulated at a time (two independent MFMA chains).  The
// products, their order and the epilogue are those of k_linear: results are bit-identical.  Vector path only (XV && YV), nti <= NTI.
// STATS: per-workgroup column moments (count, mean, M2) of the OUTPUT rows the workgroup produced, for the train-mode BatchNorm that
// follows a Linear (k_bn_train_finish1 merges them): the separate pass over z disappears.  stat: [nblk][C] means, [nblk][C] M2, [nblk]
// counts.
constexpr int LIN_W = 8;      // waves per workgroup of k_linear_lds (one workgroup per CU: two waves per SIMD)
template <int NTI, int NTO, bool STATS>
__global__ __launch_bounds__(64 * LIN_W, 1) void k_linear_lds(LinArgs a, int64_t ntiles, float* __restrict__ stat) {
  extern __shared__ __align__(16) unsigned char lin_lds[];
  float4* wl = reinterpret_cast<float4*>(lin_lds);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nw4 = a.nto * a.nti * 64;
  const int g = lane >> 4;
  // a contiguous, balanced range of tiles per workgroup, dealt round-robin to its waves (waves w and w + 4 share a SIMD)
  const int64_t t_lo = ntiles * blockIdx.x / gridDim.x, t_hi = ntiles * (blockIdx.x + 1) / gridDim.x;
  const int64_t gw = t_lo + wave, nwv = LIN_W;
  ntiles = t_hi;
  // running moments of this wave's rows (STATS): every lane keeps the statistics of its own 4 columns per output tile
  float rn = 0.f;
  f32x4 rmean[STATS ? NTO : 1], rm2[STATS ? NTO : 1];
  if (STATS) {
#pragma unroll
    for (int ot = 0; ot < NTO; ++ot) { rmean[ot] = f32x4{0.f, 0.f, 0.f, 0.f}; rm2[ot] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  }
  // the next tile's operand rows are fetched while the current tile is in the matrix pipe (register double buffer)
  auto row_valid = [&](int64_t row) {
    bool v = row < a.R;
    if (v && a.nvalid) {
      const int64_t node = row / a.K;
      v = (int)(row - node * a.K) < a.nvalid[node];
    }
    return v;
  };
  auto fetch = [&](int64_t tile, bool v, f32x4 (&buf)[NTI]) {
    const float* xr = a.x + (tile * 16 + (lane & 15)) * a.ldx;
#pragma unroll
    for (int kk = 0; kk < NTI; ++kk) {
      buf[kk] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (kk < a.nti && v) buf[kk] = load4<true>(xr, 16 * kk + 4 * g, a.d_in);
    }
  };
  constexpr bool PF = !STATS;                  // (the operand, its copy and the running moments do not fit in 256 registers together)
  f32x4 in[NTI], nx[PF ? NTI : 1];
  bool valid = false, nvalid_next = false;
  if (gw < ntiles) {                      // the first tile's rows are on their way while the weight is staged
    valid = row_valid(gw * 16 + (lane & 15));
    fetch(gw, valid, in);
  }
  for (int i = threadIdx.x; i < nw4; i += 64 * LIN_W) wl[i] = a.wp[i];
  __syncthreads();
  for (int64_t tile = gw; tile < ntiles; tile += nwv) {
    const int64_t row = tile * 16 + (lane & 15);
    const bool inr = row < a.R;
    const bool more = tile + nwv < ntiles;
    if (more) {
      nvalid_next = row_valid((tile + nwv) * 16 + (lane & 15));
      if constexpr (PF) fetch(tile + nwv, nvalid_next, nx);
    }
    float* yr = a.y + row * a.ldy;
    const unsigned long long vb = __ballot(valid);
    if (vb == 0ull) {  // nothing valid in this tile: zeros
      if (inr) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        for (int ot = 0; ot < a.nto; ++ot) store4<true>(yr, 16 * ot + 4 * g, a.d_out, z);
      }
      if (more) {
        if constexpr (PF) {
#pragma unroll
          for (int kk = 0; kk < NTI; ++kk) in[kk] = nx[kk];
        } else {
          fetch(tile + nwv, nvalid_next, in);
        }
        valid = nvalid_next;
      }
      continue;
    }
    const float nt = (float)__popcll(vb & 0xffffull);                  // valid rows of the tile (STATS)
    float inv_nt = 0.f, wa = 0.f, wb = 0.f;
    if (STATS) {
      const float n = rn + nt;
      inv_nt = 1.0f / nt;                                              // nt >= 1 here
      wb = nt / n;                                                     // Chan's update: mean += d * nb/n, M2 += M2b + d^2 * na*nb/n
      wa = rn * wb;
    }
    auto epilogue = [&](int ot, f32x4 acc) {
      const int o0 = 16 * ot + 4 * g;
      f32x4 v = acc;
      if (!valid) {
        v = f32x4{0.f, 0.f, 0.f, 0.f};
      } else {
        if (a.flags & SN_EPI_BIAS) v += load4<true>(a.bias, o0, a.d_out);
        if (a.flags & SN_EPI_BLOCK_BIAS) v += load4<true>(a.bbias + (row / a.bb_rows) * a.ldbb, o0, a.d_out);
        if (a.flags & SN_EPI_RESIDUAL_PRE) v += load4<true>(a.res + row * a.ldr, o0, a.d_out);
        if (a.flags & SN_EPI_RELU_PRE) v = f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
        if (a.flags & SN_EPI_AFFINE) {
          const f32x4 sc = load4<true>(a.scale, o0, a.d_out), sh = load4<true>(a.shift, o0, a.d_out);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = v[r] * sc[r] + sh[r];
        }
        if (a.flags & SN_EPI_RELU) v = f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
        if (a.flags & SN_EPI_LEAKY) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : v[r] * 0.01f;
        }
        if (a.flags & SN_EPI_RESIDUAL) v += load4<true>(a.res + row * a.ldr, o0, a.d_out);
      }
      if (inr) store4<true>(yr, o0, a.d_out, v);
      if (STATS) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float mt = tile16_sum(v[r]) * inv_nt;                  // tile mean of the column (invalid rows hold 0)
          const float d = valid ? v[r] - mt : 0.f;
          const float qt = tile16_sum(d * d);
          const float dm = mt - rmean[ot][r];
          rmean[ot][r] += dm * wb;
          rm2[ot][r] += qt + dm * dm * wa;
        }
      }
    };
#pragma unroll
    for (int ot = 0; ot < NTO; ot += 2) {          // fully unrolled (compile-time register indices); guards are wave-uniform
      if (ot + 1 < a.nto) {
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
        const float4* w0 = wl + (ot * a.nti) * 64 + lane;
        const float4* w1 = w0 + a.nti * 64;
#pragma unroll
        for (int kk = 0; kk < NTI; ++kk) {
          if (kk < a.nti) {
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
      } else if (ot < a.nto) {
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f};
        const float4* w0 = wl + (ot * a.nti) * 64 + lane;
#pragma unroll
        for (int kk = 0; kk < NTI; ++kk) {
          if (kk < a.nti) {
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
    if (STATS) rn += nt;
    if (more) {
      if constexpr (PF) {
#pragma unroll
        for (int kk = 0; kk < NTI; ++kk) in[kk] = nx[kk];
      } else {
        fetch(tile + nwv, nvalid_next, in);
      }
      valid = nvalid_next;
    }
  }
  if (STATS) {
    // one partial per WORKGROUP: the waves' moments meet in LDS (the weight image is dead by now) and are merged in wave order
    __syncthreads();
    float* sm = reinterpret_cast<float*>(lin_lds);          // [LIN_W waves][mean 16*NTO | M2 16*NTO], counts behind
    float* sc = sm + LIN_W * 2 * 16 * NTO;
    if ((lane & 15) == 0) {
#pragma unroll
      for (int ot = 0; ot < NTO; ++ot) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sm[(wave * 2 + 0) * 16 * NTO + 16 * ot + 4 * g + r] = rmean[ot][r];
          sm[(wave * 2 + 1) * 16 * NTO + 16 * ot + 4 * g + r] = rm2[ot][r];
        }
      }
    }
    if (lane == 0) sc[wave] = rn;
    __syncthreads();
    const int nblk = gridDim.x;
    for (int c = threadIdx.x; c < a.d_out; c += 64 * LIN_W) {
      float n = 0.f, m = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < LIN_W; ++w) {
        const float nb = sc[w];
        if (nb > 0.f) {
          const float mb = sm[(w * 2 + 0) * 16 * NTO + c], qb = sm[(w * 2 + 1) * 16 * NTO + c];
          const float nn = n + nb, d = mb - m;
          m += d * (nb / nn);
          q += qb + d * d * (n * nb / nn);
          n = nn;
        }
      }
      stat[(int64_t)blockIdx.x * a.d_out + c] = m;
      stat[((int64_t)nblk + blockIdx.x) * a.d_out + c] = q;
    }
    if (threadIdx.x == 0) {
      float n = 0.f;
#pragma unroll
      for (int w = 0; w < LIN_W; ++w) n += sc[w];       // integers < 2^24: exact in any order
      stat[2 * (int64_t)nblk * a.d_out + blockIdx.x] = n;
    }
  }
}

template <typename VT>
__device__ __forceinline__ VT vzero();
template <>
__device__ __forceinline__ float vzero<float>() { return 0.f; }
template <>
__device__ __forceinline__ float4 vzero<float4>() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// acc + x*s with the product rounded separately (no FMA contraction), so the result is bit-identical
// to the reference's `out += (1 + eps) * x` evaluated as two fp32 ops.
__device__ __forceinline__ float self_term(float acc, float x, float s) {
#pragma clang fp contract(off)
  float p = x * s;
  return acc + p;
}
__device__ __forceinline__ float4 self_term(float4 acc, float4 x, float s) {
  return make_float4(self_term(acc.x, x.x, s), self_term(acc.y, x.y, s), self_term(acc.z, x.z, s),
                     self_term(acc.w, x.w, s));
}

// ============================================================================ GIN aggregate (gather)
// out[i,f] = sum_{e in in(i)} x[col[e], f]  +  (1+eps) * x[i,f]      (neighbours first, in edge-id
// order, then the self term: the order PyG's propagate + `out += (1+eps)*x_r` produces)
#ifndef SN_GIN_UW
#define SN_GIN_UW 2
#endif
#ifndef SN_GIN_KU
#define SN_GIN_KU 2
#endif
#ifndef SN_GIN_UN
#define SN_GIN_UN 2
#endif
// A workgroup owns 256 / CW groups of U consecutive nodes x CW vector columns (CW = 256: one group, node ids block-uniform, so
// the CSR reads are scalar loads; narrower rows: several groups side by side, so that no thread idles).  The feature reads of a
// thread's U nodes are issued together, neighbour k of every node in one go — with one node per workgroup the kernel was a chain
// of dependent round trips (row pointer -> column -> row) with 4 KB in flight per workgroup and delivered 4.2 TB/s on [N, 2048]
// (1.8 TB/s on [N, 128], where 7 of 8 threads had no column).  The additions stay in edge order per node, self term last.
// consecutive workgroups that share an XCD (its L2 serves the neighbour reads): 128 nodes of wide rows, 512 of narrow ones
// (measured: 32 / 128 / 512 nodes -> 5.13 / 5.18 / 4.59 TB/s on [N, 2048], 4.2 / 4.5 / 4.6 TB/s on [N, 128])
constexpr int gin_xcd_chunk(int U, int CW) {
  const int nodes = CW == 256 ? 128 : 512, per_wg = U * (256 / CW);
  return nodes / per_wg > 0 ? nodes / per_wg : 1;
}
template <typename VT, int U, int CW>
__global__ __launch_bounds__(256) void k_gin_gather(const VT* __restrict__ x, VT* __restrict__ out, int64_t N,
                                                    int FV, int P, const int32_t* __restrict__ rowptr,
                                                    const int32_t* __restrict__ col,
                                                    const float* __restrict__ eps, int negate, const VT* __restrict__ plus) {
  constexpr int NSUB = 256 / CW;
  constexpr int KU = SN_GIN_KU;                  // neighbours of a node in flight
  constexpr int CHUNK = gin_xcd_chunk(U, CW);
  const int64_t L = xcd_remap(blockIdx.x, CHUNK * P);
  const int64_t grp = L / P;
  const int part = (int)(L - grp * P);
  const int sub = CW == 256 ? 0 : (int)threadIdx.x / CW;
  const int64_t n0 = (grp * NSUB + sub) * U;
  const int f = part * 256 + (CW == 256 ? (int)threadIdx.x : (int)threadIdx.x % CW);
  if (n0 >= N || f >= FV) return;
  const float sc = 1.f + (eps ? *eps : 0.f);
  VT zero = vzero<VT>();
  int lo[U], deg[U], kmax = 0;
  VT self[U], acc[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const bool live = n0 + u < N;
    lo[u] = live ? rowptr[n0 + u] : 0;
    deg[u] = live ? rowptr[n0 + u + 1] - lo[u] : 0;
    kmax = deg[u] > kmax ? deg[u] : kmax;
    self[u] = zero;
    if (live) self[u] = x[(n0 + u) * FV + f];
    acc[u] = zero;
  }
  for (int k0 = 0; k0 < kmax; k0 += KU) {
    VT v[U][KU];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int j = 0; j < KU; ++j) {
        v[u][j] = zero;
        if (k0 + j < deg[u]) v[u][j] = x[(int64_t)col[lo[u] + k0 + j] * FV + f];
      }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int j = 0; j < KU; ++j)
        if (k0 + j < deg[u]) acc[u] = acc[u] + v[u][j];          // (guarded: -0 + 0 would not be -0)
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (n0 + u < N) {
      VT r = self_term(acc[u], self[u], sc);
      if (negate) r = zero - r;
      if (plus) r = r + plus[(n0 + u) * FV + f];      // the adjoint's other branch (a residual's gradient), added in the same pass
      out[(n0 + u) * FV + f] = r;
    }
  }
}

template <typename VT, int U, int CW>
static void launch_gin_gather(const VT* x, VT* out, int64_t N, int FV, const int32_t* rowptr, const int32_t* col, const float* eps,
                              int negate, hipStream_t st, const VT* plus = nullptr) {
  constexpr int NSUB = 256 / CW;
  constexpr int CHUNK = gin_xcd_chunk(U, CW);
  const int P = CW == 256 ? (int)cdiv(FV, 256) : 1;
  const int64_t grp = (int64_t)8 * CHUNK * P;
  const int64_t nblk = cdiv(cdiv(N, (int64_t)U * NSUB) * P, grp) * grp;
  hipLaunchKernelGGL((k_gin_gather<VT, U, CW>), dim3((unsigned)nblk), dim3(256), 0, st, x, out, N, FV, P, rowptr, col, eps, negate, plus);
}
template <typename VT>
static void dispatch_gin_gather(const VT* x, VT* out, int64_t N, int FV, const int32_t* rowptr, const int32_t* col, const float* eps,
                                int negate, hipStream_t st, const VT* plus = nullptr) {
  if (FV > 128) launch_gin_gather<VT, SN_GIN_UW, 256>(x, out, N, FV, rowptr, col, eps, negate, st, plus);
  else if (FV > 64) launch_gin_gather<VT, SN_GIN_UN, 128>(x, out, N, FV, rowptr, col, eps, negate, st, plus);
  else if (FV > 32) launch_gin_gather<VT, SN_GIN_UN, 64>(x, out, N, FV, rowptr, col, eps, negate, st, plus);
  else launch_gin_gather<VT, SN_GIN_UN, 32>(x, out, N, FV, rowptr, col, eps, negate, st, plus);
}

// ============================================================================ GIN aggregate (LDS slab)
// One workgroup per (graph, chunk of CH vector columns).  The graph's rows of this chunk and its
// CSR slice are staged in LDS; every feature row is read from HBM exactly once (coalesced), every
// output row written once.  Graphs that do not fit the LDS budget fall back to the gather form.
template <typename VT>
__global__ __launch_bounds__(256) void k_gin_slab(const VT* __restrict__ x, VT* __restrict__ out, int FV, int CH,
                                                  int nchunk, const int32_t* __restrict__ graph_ptr,
                                                  const int32_t* __restrict__ rowptr,
                                                  const int32_t* __restrict__ col, const float* __restrict__ eps,
                                                  int negate, int lds_bytes) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int g = blockIdx.x / nchunk, c = blockIdx.x - g * nchunk;
  const int gs = graph_ptr[g], n = graph_ptr[g + 1] - gs;
  if (n <= 0) return;
  const int c0 = c * CH;
  const int cw = (FV - c0) < CH ? (FV - c0) : CH;  // vector columns in this chunk
  const int e0 = rowptr[gs], ne = rowptr[gs + n] - e0;
  const float sc = 1.f + (eps ? *eps : 0.f);
  const int64_t need = (int64_t)n * cw * sizeof(VT) + (int64_t)(n + 1 + ne) * 4;
  const int tid = threadIdx.x;
  if (need <= lds_bytes) {
    VT* slab = reinterpret_cast<VT*>(smem);
    int* lrow = reinterpret_cast<int*>(smem + (size_t)n * cw * sizeof(VT));
    int* lcol = lrow + n + 1;
    for (int i = tid; i < n * cw; i += 256) {
      int r = i / cw, f = i - r * cw;
      slab[i] = x[(int64_t)(gs + r) * FV + c0 + f];
    }
    for (int i = tid; i <= n; i += 256) lrow[i] = rowptr[gs + i] - e0;
    for (int i = tid; i < ne; i += 256) lcol[i] = col[e0 + i] - gs;
    __syncthreads();
    for (int i = tid; i < n * cw; i += 256) {
      int r = i / cw, f = i - r * cw;
      VT self = slab[i];
      const VT zero = vzero<VT>();
      VT acc = zero;
      for (int e = lrow[r]; e < lrow[r + 1]; ++e) acc = acc + slab[lcol[e] * cw + f];
      acc = self_term(acc, self, sc);
      out[(int64_t)(gs + r) * FV + c0 + f] = negate ? zero - acc : acc;
    }
  } else {
    for (int i = tid; i < n * cw; i += 256) {
      int r = i / cw, f = i - r * cw;
      VT self = x[(int64_t)(gs + r) * FV + c0 + f];
      const VT zero = vzero<VT>();
      VT acc = zero;
      for (int e = rowptr[gs + r]; e < rowptr[gs + r + 1]; ++e) acc = acc + x[(int64_t)col[e] * FV + c0 + f];
      acc = self_term(acc, self, sc);
      out[(int64_t)(gs + r) * FV + c0 + f] = negate ? zero - acc : acc;
    }
  }
}

// ============================================================================ GINE aggregate
template <typename VT>
__device__ __forceinline__ VT vrelu(VT v);
template <>
__device__ __forceinline__ float vrelu<float>(float v) { return fmaxf(v, 0.f); }
template <>
__device__ __forceinline__ float4 vrelu<float4>(float4 v) {
  return make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
}

#ifndef SN_GINE_XCD_NODES
#define SN_GINE_XCD_NODES 128
#endif
#ifndef SN_GINE_U
#define SN_GINE_U 1
#endif
#ifndef SN_GINE_KU
#define SN_GINE_KU 4
#endif
constexpr int GINE_U = SN_GINE_U, GINE_KU = SN_GINE_KU;
// A thread owns one vector column of GINE_U consecutive nodes and walks their in-edges GINE_KU at a time: the (source, edge id)
// pairs of all of them are read together, then the 2 x GINE_U x GINE_KU feature rows — one node and one edge per step was a chain
// of dependent round trips.  Per node the additions stay in edge order, self term last.
template <typename VT>
__global__ __launch_bounds__(256) void k_gine_gather(const VT* __restrict__ x, const VT* __restrict__ ea,
                                                     VT* __restrict__ out, int64_t N, int CV,
                                                     const int32_t* __restrict__ rowptr,
                                                     const int32_t* __restrict__ col,
                                                     const int32_t* __restrict__ eperm,
                                                     const float* __restrict__ eps, int xcd_chunk) {
  // runs of xcd_chunk consecutive workgroups (128 / 512 consecutive nodes) share an XCD, whose L2 then serves the source rows
  const int64_t idx = xcd_remap(blockIdx.x, xcd_chunk) * 256 + threadIdx.x;
  const int64_t ngrp = (N + GINE_U - 1) / GINE_U;
  if (idx >= ngrp * CV) return;
  const int64_t n0 = (idx / CV) * GINE_U;
  const int f = (int)(idx % CV);
  const float sc = 1.f + (eps ? *eps : 0.f);
  VT zero = vzero<VT>();
  int lo[GINE_U], deg[GINE_U], kmax = 0;
  VT self[GINE_U], acc[GINE_U];
#pragma unroll
  for (int u = 0; u < GINE_U; ++u) {
    const bool live = n0 + u < N;
    lo[u] = live ? rowptr[n0 + u] : 0;
    deg[u] = live ? rowptr[n0 + u + 1] - lo[u] : 0;
    kmax = deg[u] > kmax ? deg[u] : kmax;
    self[u] = zero;
    if (live) self[u] = x[(n0 + u) * CV + f];
    acc[u] = zero;
  }
  for (int k0 = 0; k0 < kmax; k0 += GINE_KU) {
    int c[GINE_U][GINE_KU], p[GINE_U][GINE_KU];
    VT xv[GINE_U][GINE_KU], ev[GINE_U][GINE_KU];
#pragma unroll
    for (int u = 0; u < GINE_U; ++u)
#pragma unroll
      for (int j = 0; j < GINE_KU; ++j) {
        c[u][j] = 0; p[u][j] = 0;
        if (k0 + j < deg[u]) { c[u][j] = col[lo[u] + k0 + j]; p[u][j] = eperm[lo[u] + k0 + j]; }
      }
#pragma unroll
    for (int u = 0; u < GINE_U; ++u)
#pragma unroll
      for (int j = 0; j < GINE_KU; ++j) {
        xv[u][j] = zero; ev[u][j] = zero;
        if (k0 + j < deg[u]) { xv[u][j] = x[(int64_t)c[u][j] * CV + f]; ev[u][j] = ea[(int64_t)p[u][j] * CV + f]; }
      }
#pragma unroll
    for (int u = 0; u < GINE_U; ++u)
#pragma unroll
      for (int j = 0; j < GINE_KU; ++j)
        if (k0 + j < deg[u]) acc[u] = acc[u] + vrelu<VT>(xv[u][j] + ev[u][j]);
  }
#pragma unroll
  for (int u = 0; u < GINE_U; ++u)
    if (n0 + u < N) out[(n0 + u) * CV + f] = self_term(acc[u], self[u], sc);
}

// ============================================================================ masked column statistics
// pass 0: per-block partial sums of x over valid rows; pass 1: partial sums of (x-mean)^2.
// A block = 4 row lanes x 64 column lanes over <= rows_per_block rows (coalesced 256-byte row segments, four rows in flight).
__global__ __launch_bounds__(256) void k_colstats_partial(const float* __restrict__ x, int ldx, int64_t R, int C,
                                                          const int32_t* __restrict__ nvalid, int K,
                                                          const float* __restrict__ mean, int pass,
                                                          int64_t rows_per_block, float* __restrict__ part,
                                                          float* __restrict__ cnt_part) {
  __shared__ float red[4][64];
  __shared__ int cred[4];
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = (r0 + rows_per_block < R) ? r0 + rows_per_block : R;
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int c = c0 + cl;
    const float m = (pass && c < C) ? mean[c] : 0.f;
    float s = 0.f;
    int cnt = 0;
    for (int64_t r = r0 + rl; r < r1; r += 4) {
      bool ok = true;
      if (nvalid) { unsigned node = (unsigned)r / (unsigned)K; ok = (int)((unsigned)r - node * (unsigned)K) < nvalid[node]; }   // rows < 2^31
      if (ok) {
        ++cnt;
        if (c < C) { const float v = x[r * ldx + c] - m; s += pass ? v * v : v; }
      }
    }
    red[rl][cl] = s;
    if (cl == 0) cred[rl] = cnt;
    __syncthreads();
    if (rl == 0 && c < C) part[(int64_t)blockIdx.x * C + c] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
    if (threadIdx.x == 0 && c0 == 0 && cnt_part) cnt_part[blockIdx.x] = (float)(cred[0] + cred[1] + cred[2] + cred[3]);
    __syncthreads();
  }
}
// sums a slice of the partial rows: out[y][c] = sum_{b in slice y} part[b][c]  (c == C: the per-block counts)
__global__ __launch_bounds__(256) void k_colstats_reduce(const float* __restrict__ part, const float* __restrict__ cnt_part,
                                                         int nblk, int C, float* __restrict__ out, float* __restrict__ out_cnt) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c > C || (c == C && !cnt_part)) return;
  const int per = (nblk + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = (b0 + per < nblk) ? b0 + per : nblk;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int b = b0;
  if (c < C) {
    for (; b + 3 < b1; b += 4) {
      s0 += part[(int64_t)b * C + c];
      s1 += part[(int64_t)(b + 1) * C + c];
      s2 += part[(int64_t)(b + 2) * C + c];
      s3 += part[(int64_t)(b + 3) * C + c];
    }
    for (; b < b1; ++b) s0 += part[(int64_t)b * C + c];
    out[(int64_t)blockIdx.y * C + c] = (s0 + s1) + (s2 + s3);
  } else {
    for (; b < b1; ++b) s0 += cnt_part[b];
    out_cnt[blockIdx.y] = s0;
  }
}
__global__ __launch_bounds__(256) void k_colstats_final(const float* __restrict__ part, const float* __restrict__ cnt_part,
                                                        int nblk, int C, float* __restrict__ outv,
                                                        float* __restrict__ count, int pass) {
  __shared__ float tot;
  if (threadIdx.x == 0) {
    float t = 0.f;
    if (pass == 0) { for (int b = 0; b < nblk; ++b) t += cnt_part[b]; *count = t; }
    else t = *count;
    tot = t;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float s = 0.f;
    for (int b = 0; b < nblk; ++b) s += part[(int64_t)b * C + c];
    outv[c] = tot > 0.f ? s / tot : 0.f;
  }
}

// One-pass column moments for the train-mode BatchNorm: every block writes (count, mean, M2) of its rows per column — sums of
// d = x - s and d^2 around a shift s taken from the block's first valid row (a sample of the column, so the subtraction
// S2 - S1^2/n loses at most a small constant factor) — and k_bn_train_finish1 merges the blocks with Chan's pairwise update.
// x is read ONCE (the two-pass form read it twice and took six launches).
__global__ __launch_bounds__(256) void k_colstats_moments(const float* __restrict__ x, int ldx, int64_t R, int C,
                                                          const int32_t* __restrict__ nvalid, int K, int64_t rows_per_block,
                                                          float* __restrict__ pmean, float* __restrict__ pm2,
                                                          float* __restrict__ pcnt) {
  __shared__ float red1[4][64], red2[4][64];
  __shared__ int cred[4];
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = (r0 + rows_per_block < R) ? r0 + rows_per_block : R;
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  int64_t rs = r0;                                             // first valid row of the block (uniform)
  if (nvalid && r0 < r1) {
    const unsigned node = (unsigned)r0 / (unsigned)K;
    if ((int)((unsigned)r0 - node * (unsigned)K) >= nvalid[node]) {
      int64_t nd = (int64_t)node + 1;
      while (nd * K < r1 && nvalid[nd] <= 0) ++nd;
      rs = nd * K;
    }
  }
  const bool any = rs < r1;
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int c = c0 + cl;
    const float sh = (any && c < C) ? x[rs * ldx + c] : 0.f;
    float s1 = 0.f, s2 = 0.f;
    int cnt = 0;
    for (int64_t r = r0 + rl; r < r1; r += 4) {
      bool ok = true;
      if (nvalid) { unsigned node = (unsigned)r / (unsigned)K; ok = (int)((unsigned)r - node * (unsigned)K) < nvalid[node]; }
      if (ok) {
        ++cnt;
        if (c < C) { const float d = x[r * ldx + c] - sh; s1 += d; s2 += d * d; }
      }
    }
    red1[rl][cl] = s1;
    red2[rl][cl] = s2;
    if (cl == 0) cred[rl] = cnt;
    __syncthreads();
    if (rl == 0 && c < C) {
      const float n = (float)(cred[0] + cred[1] + cred[2] + cred[3]);
      const float t1 = (red1[0][cl] + red1[1][cl]) + (red1[2][cl] + red1[3][cl]);
      const float t2 = (red2[0][cl] + red2[1][cl]) + (red2[2][cl] + red2[3][cl]);
      pmean[(int64_t)blockIdx.x * C + c] = n > 0.f ? sh + t1 / n : 0.f;
      pm2[(int64_t)blockIdx.x * C + c] = n > 0.f ? fmaxf(t2 - t1 * t1 / n, 0.f) : 0.f;
    }
    if (threadIdx.x == 0 && c0 == 0) pcnt[blockIdx.x] = (float)(cred[0] + cred[1] + cred[2] + cred[3]);
    __syncthreads();
  }
}
__device__ __forceinline__ void chan_merge(float& na, float& ma, float& qa, float nb, float mb, float qb) {
  if (nb <= 0.f) return;
  const float n = na + nb, d = mb - ma;
  ma += d * (nb / n);
  qa += qb + d * d * (na * nb / n);
  na = n;
}
// merges the per-block moments (16 lanes per column, then a tree across the lanes) and finishes the BatchNorm: mean, biased variance,
// rstd, folded (scale, shift), count, running statistics.  grid cdiv(C, 16), 256 threads = 16 columns x 16 lanes.  (With 4 lanes per
// column the merge of the 1024 partial blocks of a million-row input — the LearningFilters epochs — was an 88 us serial chain.)
template <int LP>   // lanes per column: 16 (few partials) or 64 (the thousands of per-wave partials of k_linear_lds)
__global__ __launch_bounds__(256) void k_bn_train_finish1(const float* __restrict__ pmean, const float* __restrict__ pm2,
                                                          const float* __restrict__ pcnt, int nblk, int C,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                          float momentum, float* __restrict__ mean, float* __restrict__ var,
                                                          float* __restrict__ rstd, float* __restrict__ scale, float* __restrict__ shift,
                                                          float* __restrict__ count, float* __restrict__ rmean, float* __restrict__ rvar) {
  constexpr int COLS = 256 / LP;
  __shared__ float ln[LP][COLS + 1], lm[LP][COLS + 1], lq[LP][COLS + 1];
  const int cl = threadIdx.x % COLS, rl = threadIdx.x / COLS;
  const int c = blockIdx.x * COLS + cl;
  const int per = (nblk + LP - 1) / LP, b0 = rl * per, b1 = b0 + per < nblk ? b0 + per : nblk;
  float n = 0.f, m = 0.f, q = 0.f;
  if (c < C)
    for (int b = b0; b < b1; ++b) chan_merge(n, m, q, pcnt[b], pmean[(int64_t)b * C + c], pm2[(int64_t)b * C + c]);
  ln[rl][cl] = n; lm[rl][cl] = m; lq[rl][cl] = q;
  __syncthreads();
  for (int step = LP / 2; step >= 1; step >>= 1) {     // pairwise tree in a fixed order
    if (rl < step) {
      chan_merge(n, m, q, ln[rl + step][cl], lm[rl + step][cl], lq[rl + step][cl]);
      ln[rl][cl] = n; lm[rl][cl] = m; lq[rl][cl] = q;
    }
    __syncthreads();
  }
  if (rl != 0 || c >= C) return;
  const float v = n > 0.f ? q / n : 0.f;
  const float rs = 1.0f / sqrtf(v + eps);
  const float sc = (gamma ? gamma[c] : 1.f) * rs;
  mean[c] = m;
  var[c] = v;
  rstd[c] = rs;
  scale[c] = sc;
  shift[c] = (beta ? beta[c] : 0.f) - m * sc;
  if (c == 0) count[0] = n;
  if (rmean) {
    const float unb = n > 1.f ? v * (n / (n - 1.f)) : v;
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * m;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * unb;
  }
}

// Second-pass finish of a train-mode BatchNorm in ONE launch: variance from the partial sums of (x-mean)^2, then
// rstd, the folded (scale, shift) and the running-statistics update (what k_colstats_final + two k_bn_fold + k_bn_running_update did).
__global__ __launch_bounds__(256) void k_bn_train_finish(const float* __restrict__ part, int nblk, int C, const float* __restrict__ count,
                                                         const float* __restrict__ mean, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float eps, float momentum,
                                                         float* __restrict__ var, float* __restrict__ rstd,
                                                         float* __restrict__ scale, float* __restrict__ shift,
                                                         float* __restrict__ rmean, float* __restrict__ rvar) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float n = count[0];
  float s = 0.f;
  for (int b = 0; b < nblk; ++b) s += part[(int64_t)b * C + c];
  const float v = n > 0.f ? s / n : 0.f;
  const float rs = 1.0f / sqrtf(v + eps);
  const float sc = (gamma ? gamma[c] : 1.f) * rs;
  var[c] = v;
  rstd[c] = rs;
  scale[c] = sc;
  shift[c] = (beta ? beta[c] : 0.f) - mean[c] * sc;
  if (rmean) {
    const float unb = n > 1.f ? v * (n / (n - 1.f)) : v;
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean[c];
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * unb;
  }
}

// ============================================================================ masked affine (un-fused BN apply)
__global__ __launch_bounds__(256) void k_affine(const float* __restrict__ x, int ldx, int64_t R, int C,
                                                const int32_t* __restrict__ nvalid, int K, int flags,
                                                const float* __restrict__ scale, const float* __restrict__ shift,
                                                const float* __restrict__ res, int ldr, float* __restrict__ y,
                                                int ldy) {
  int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= R * C) return;
  int64_t r = idx / C;
  int c = (int)(idx - r * C);
  bool ok = true;
  if (nvalid) { unsigned node = (unsigned)r / (unsigned)K; ok = (int)((unsigned)r - node * (unsigned)K) < nvalid[node]; }   // rows < 2^31
  float v = 0.f;
  if (ok) {
    v = x[r * ldx + c];
    if (flags & SN_EPI_RELU_PRE) v = fmaxf(v, 0.f);
    if (flags & SN_EPI_AFFINE) v = v * scale[c] + shift[c];
    if (flags & SN_EPI_RELU) v = fmaxf(v, 0.f);
    if (flags & SN_EPI_RESIDUAL) v += res[r * ldr + c];
  }
  y[r * ldy + c] = v;
}

// ============================================================================ masked LayerNorm
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__global__ __launch_bounds__(256) void k_layernorm(const float* __restrict__ x, const float* __restrict__ res,
                                                   int64_t R, int C, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, float eps,
                                                   const int32_t* __restrict__ nvalid, int K,
                                                   float* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  bool ok = true;
  if (nvalid) { unsigned node = (unsigned)row / (unsigned)K; ok = (int)((unsigned)row - node * (unsigned)K) < nvalid[node]; }
  const float* xr = x + row * C;
  const float* rr = res ? res + row * C : nullptr;
  float* yr = y + row * C;
  if (!ok) {
    for (int c = lane; c < C; c += 64) yr[c] = 0.f;
    return;
  }
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += xr[c] + (rr ? rr[c] : 0.f);
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
  for (int c = lane; c < C; c += 64) { float d = xr[c] + (rr ? rr[c] : 0.f) - mean; q += d * d; }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
  for (int c = lane; c < C; c += 64) {
    float v = xr[c] + (rr ? rr[c] : 0.f);
    yr[c] = (v - mean) * rstd * gamma[c] + beta[c];
  }
}

// ============================================================================ per-node set attention
// One wave per (node, head).  q,k,v rows are [N*K, H*dk]; head h owns columns [h*dk, (h+1)*dk).
__global__ __launch_bounds__(64) void k_set_attention(const float* __restrict__ q, const float* __restrict__ k,
                                                      const float* __restrict__ v, int K, int H, int dk,
                                                      const int32_t* __restrict__ nvalid, const float* __restrict__ pmask,
                                                      float* __restrict__ out) {
  extern __shared__ float sm[];
  const int node = blockIdx.x / H, h = blockIdx.x - node * H;
  const int lane = threadIdx.x;
  const int kv = nvalid ? nvalid[node] : K;
  const int D = H * dk;
  float* sq = sm;                // [K][dk]   (q / sqrt(dk))
  float* sk = sq + K * dk;       // [K][dk]
  float* sv = sk + K * dk;       // [K][dk]
  float* sp = sv + K * dk;       // [K][K+1]
  const float temp = sqrtf((float)dk);
  const int64_t base = (int64_t)node * K * D + (int64_t)h * dk;
  for (int i = lane; i < kv * dk; i += 64) {
    int r = i / dk, c = i - r * dk;
    int64_t o = base + (int64_t)r * D + c;
    sq[i] = q[o] / temp;
    sk[i] = k[o];
    sv[i] = v[o];
  }
  __syncthreads();
  for (int i = lane; i < kv * kv; i += 64) {
    int a = i / kv, b = i - a * kv;
    float s = 0.f;
    for (int c = 0; c < dk; ++c) s += sq[a * dk + c] * sk[b * dk + c];
    sp[a * (K + 1) + b] = s;
  }
  __syncthreads();
  for (int a = lane; a < kv; a += 64) {
    float m = -INFINITY;
    for (int b = 0; b < kv; ++b) m = fmaxf(m, sp[a * (K + 1) + b]);
    float z = 0.f;
    for (int b = 0; b < kv; ++b) { float e = expf(sp[a * (K + 1) + b] - m); sp[a * (K + 1) + b] = e; z += e; }
    // attention dropout (transformer_module.py:55): the caller's mask holds 0 or 1/(1-p) per (node, head, query, key)
    const float* pm = pmask ? pmask + ((int64_t)blockIdx.x * K + a) * K : nullptr;
    for (int b = 0; b < kv; ++b) sp[a * (K + 1) + b] = sp[a * (K + 1) + b] / z * (pm ? pm[b] : 1.0f);
  }
  __syncthreads();
  for (int i = lane; i < K * dk; i += 64) {
    int a = i / dk, c = i - a * dk;
    float s = 0.f;
    if (a < kv)
      for (int b = 0; b < kv; ++b) s += sp[a * (K + 1) + b] * sv[b * dk + c];
    out[base + (int64_t)a * D + c] = s;
  }
}

// ============================================================================ small reductions / gathers
__global__ __launch_bounds__(256) void k_slot_sum(const float* __restrict__ x, int64_t N, int K, int C,
                                                  float* __restrict__ out) {
  int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= N * C) return;
  int64_t n = idx / C;
  int c = (int)(idx - n * C);
  const float* p = x + n * K * C + c;
  float s = 0.f;
  for (int j = 0; j < K; ++j) s += p[(int64_t)j * C];
  out[idx] = s;
}

struct TablePtrs { const float* t[10]; int64_t rows[10]; };

// An index outside [0, rows) (nn.Embedding raises IndexError there) is never dereferenced: it contributes 0 and raises bit 0 of *status.
__global__ __launch_bounds__(256) void k_embedding_sum(const int64_t* __restrict__ idx, int ldi, int nf, int64_t R,
                                                       TablePtrs tp, int C, float* __restrict__ out, int32_t* __restrict__ status) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= R * C) return;
  int64_t r = i / C;
  int c = (int)(i - r * C);
  float s = 0.f;
  bool bad = false;
  for (int f = 0; f < nf; ++f) {
    const int64_t id = idx[r * ldi + f];
    if ((uint64_t)id < (uint64_t)tp.rows[f]) s += tp.t[f][id * C + c];
    else bad = true;
  }
  out[i] = s;
  if (bad && c == 0 && status != nullptr) atomicOr(status, 1);
}

// L encoders over ONE index block (the per-layer edge encoders of a GINE stack): out[l][r][:] = sum_f tables[l][f][idx[r, f]], one launch.
struct LayerTablePtrs { const float* t[16][4]; int64_t rows[4]; };
__global__ __launch_bounds__(256) void k_embedding_sum_layers(const int64_t* __restrict__ idx, int ldi, int nf, int64_t R,
                                                              LayerTablePtrs tp, int C, float* __restrict__ out, int32_t* __restrict__ status) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= R * C) return;
  const int l = blockIdx.y;
  const int64_t r = i / C;
  const int c = (int)(i - r * C);
  float s = 0.f;
  bool bad = false;
  for (int f = 0; f < nf; ++f) {
    const int64_t id = idx[r * ldi + f];
    if ((uint64_t)id < (uint64_t)tp.rows[f]) s += tp.t[l][f][id * C + c];
    else bad = true;
  }
  out[(int64_t)l * R * C + i] = s;
  if (bad && c == 0 && l == 0 && status != nullptr) atomicOr(status, 1);
}

// One workgroup per segment; 256 threads = RL row lanes x CW column lanes (CW = the power of two >= min(C, 256)), folded in LDS.
__global__ __launch_bounds__(256) void k_segment_pool(const float* __restrict__ x, int C,
                                                      const int32_t* __restrict__ graph_ptr, int mode,
                                                      float* __restrict__ out) {
  __shared__ float red[256];
  const int g = blockIdx.x;
  const int lo = graph_ptr[g], hi = graph_ptr[g + 1];
  int CW = 1;
  while (CW < C && CW < 256) CW <<= 1;
  const int RL = 256 / CW, cl = threadIdx.x & (CW - 1), rl = threadIdx.x / CW;
  const float scale = (mode == 1) ? 1.0f / (float)(hi - lo > 0 ? hi - lo : 1) : 1.0f;
  for (int c0 = blockIdx.y * CW; c0 < C; c0 += gridDim.y * CW) {      // wide rows: column chunks spread over blockIdx.y
    const int c = c0 + cl;
    float s0 = 0.f, s1 = 0.f;
    if (c < C) {
      int i = lo + rl;
      for (; i + 7 * RL < hi; i += 8 * RL) {         // eight independent loads in flight per thread (long segments: IGN's n rows)
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = x[(int64_t)(i + u * RL) * C + c];
        s0 += (v[0] + v[2]) + (v[4] + v[6]);
        s1 += (v[1] + v[3]) + (v[5] + v[7]);
      }
      for (; i < hi; i += RL) s0 += x[(int64_t)i * C + c];
    }
    red[threadIdx.x] = s0 + s1;
    __syncthreads();
    for (int o = 128; o >= CW; o >>= 1) {
      if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (rl == 0 && c < C) out[(int64_t)g * C + c] = red[cl] * scale;
    __syncthreads();
  }
}

}  // namespace sn

namespace sn {
bool attention16_forward(const float* q, const float* k, const float* v, int64_t N, int K, int heads, int dk, const int32_t* nvalid,
                         const float* prob_mask, float* out, hipStream_t st);     // attention16.hip
}
using namespace sn;

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int sn_version(void) { return SN_ABI_VERSION; }
extern "C" const char* sn_last_error(void) { return sn::err_buf(); }

extern "C" int sn_device_info(int* cu_count, int* lds_bytes_per_cu, int* clock_khz) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return fail(SN_ERR_LAUNCH, "sn_device_info: no HIP device");
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) return fail(SN_ERR_LAUNCH, "sn_device_info: query failed");
  if (cu_count) *cu_count = p.multiProcessorCount;
  if (lds_bytes_per_cu) *lds_bytes_per_cu = (int)p.maxSharedMemoryPerMultiProcessor;
  if (clock_khz) *clock_khz = p.clockRate;
  return SN_OK;
}

// Clock probe: one wave spins for `cycles` shader cycles (s_memtime ticks = shader cycles) and reports what it counted; the caller
// brackets the launch with events: cycles / elapsed = the clock the part actually runs at under this launch pattern (a bench line
// that carries it lets a reader tell a slower BOX from a slower kernel).
namespace sn { namespace {
__global__ void k_clock_probe(long long cycles, long long* out) {
  const long long t0 = clock64();
  long long t = t0;
  while (t - t0 < cycles) t = clock64();
  if (threadIdx.x == 0) out[0] = t - t0;
}
} }
extern "C" int sn_clock_probe(int64_t cycles, int64_t* out_cycles, void* stream) {
  SN_REQUIRE(out_cycles && cycles > 0 && cycles <= (1ll << 34), "sn_clock_probe: bad arguments");
  hipLaunchKernelGGL(sn::k_clock_probe, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long)cycles, reinterpret_cast<long long*>(out_cycles));
  SN_CHECK_LAUNCH("sn_clock_probe");
  return SN_OK;
}

extern "C" int64_t sn_packed_weight_floats(int d_out, int d_in) {
  return (int64_t)cdiv(d_out, 16) * cdiv(d_in, 16) * 256;
}

extern "C" int sn_pack_weight_f32(const float* W, int d_out, int d_in, int ldw, float* Wp, void* stream) {
  SN_REQUIRE(W && Wp && d_out > 0 && d_in > 0 && ldw >= d_in, "sn_pack_weight_f32: bad arguments");
  int nto = (int)cdiv(d_out, 16), nti = (int)cdiv(d_in, 16);
  int64_t total = (int64_t)nto * nti * 256;
  hipLaunchKernelGGL(k_pack_weight, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, W, d_out,
                     d_in, ldw, nto, nti, Wp, 0);
  SN_CHECK_LAUNCH("sn_pack_weight_f32");
  return SN_OK;
}

extern "C" int sn_pack_weight_t_f32(const float* W, int rows, int cols, int ldw, float* Wp, void* stream) {
  SN_REQUIRE(W && Wp && rows > 0 && cols > 0 && ldw >= cols, "sn_pack_weight_t_f32: bad arguments");
  const int d_out = cols, d_in = rows;                 // the packed matrix is W^T: [cols, rows]
  int nto = (int)cdiv(d_out, 16), nti = (int)cdiv(d_in, 16);
  int64_t total = (int64_t)nto * nti * 256;
  hipLaunchKernelGGL(k_pack_weight, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, W, d_out,
                     d_in, ldw, nto, nti, Wp, 1);
  SN_CHECK_LAUNCH("sn_pack_weight_t_f32");
  return SN_OK;
}

extern "C" int64_t sn_split_packed_bytes(int d_out, int d_in) {
  if (d_out <= 0 || d_in <= 0) return 0;
  return cdiv(d_out, 16) * (3 * cdiv(d_in, 32) + SN_SPLIT_EPI) * 1024;
}

extern "C" int sn_pack_split_f32(const float* W, int d_out, int d_in, int ldw, const float* e0, const float* e1,
                                 const float* e2, void* Wsp, void* stream) {
  SN_REQUIRE(W && Wsp && d_out > 0 && d_in > 0 && ldw >= d_in, "sn_pack_split_f32: bad arguments");
  SN_REQUIRE(al16(Wsp), "sn_pack_split_f32: destination must be 16-byte aligned");
  const int nto = (int)cdiv(d_out, 16), nkb = (int)cdiv(d_in, 32);
  const int64_t total = (int64_t)nto * (3 * nkb + SN_SPLIT_EPI) * 64;
  hipLaunchKernelGGL(k_pack_split, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, W, d_out, d_in,
                     ldw, e0, e1, e2, nto, nkb, reinterpret_cast<unsigned char*>(Wsp));
  SN_CHECK_LAUNCH("sn_pack_split_f32");
  return SN_OK;
}

extern "C" int sn_bn_fold_f32(const float* weight, const float* bias, const float* running_mean,
                              const float* running_var, float eps, int C, int C_pad, float* scale, float* shift,
                              void* stream) {
  SN_REQUIRE(running_mean && running_var && scale && shift && C > 0 && C_pad >= C, "sn_bn_fold_f32: bad arguments");
  hipLaunchKernelGGL(k_bn_fold, dim3((unsigned)cdiv(C_pad, 128)), dim3(128), 0, (hipStream_t)stream, weight, bias,
                     running_mean, running_var, eps, C, C_pad, scale, shift);
  SN_CHECK_LAUNCH("sn_bn_fold_f32");
  return SN_OK;
}

extern "C" int sn_bn_running_update_f32(const float* mean, const float* var, const float* count, float momentum, int C,
                                        float* running_mean, float* running_var, void* stream) {
  SN_REQUIRE(mean && var && count && running_mean && running_var && C > 0, "sn_bn_running_update_f32: bad arguments");
  hipLaunchKernelGGL(k_bn_running_update, dim3((unsigned)cdiv(C, 128)), dim3(128), 0, (hipStream_t)stream, mean, var, count,
                     momentum, C, running_mean, running_var);
  SN_CHECK_LAUNCH("sn_bn_running_update_f32");
  return SN_OK;
}

// the LDS-resident-weight form also shortens the latency chain of small launches (128 rows: 26 -> 12 us); below two row tiles the
// per-wave kernel has less to stage
constexpr int64_t LIN_LDS_MIN_ROWS = 32;
static int lin_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    cus = n > 0 ? n : 256;
  }
  return cus;
}
// workgroups (= moment partials with STATS) a k_linear_lds launch over R rows uses; 0: the shape does not take that path
static int64_t lin_lds_blocks(int64_t R, int d_in, int d_out) {
  const int nti = (int)cdiv(d_in, 16), nto = (int)cdiv(d_out, 16);
  if (nti > 16 || nto > 16 || (size_t)nti * nto * 1024 > 128 * 1024) return 0;
  const int64_t ntiles = cdiv(R, 16), want = cdiv(ntiles, 4), cap = lin_cus();     // small inputs: one tile per SIMD, spread over the CUs
  return want < cap ? want : cap;
}
template <int NTI, int NTO>
static int launch_linear_lds_t(const LinArgs& a, float* stat, int64_t blocks, size_t lds, hipStream_t st) {
  constexpr bool CAN_STAT = NTO <= 8 && NTI <= 8;   // (wider shapes: the moments do not fit the register file beside the operand)
  const void* fn = reinterpret_cast<const void*>(k_linear_lds<NTI, NTO, false>);
  if constexpr (CAN_STAT) { if (stat) fn = reinterpret_cast<const void*>(k_linear_lds<NTI, NTO, true>); }
  else if (stat) return fail(SN_ERR_ARG, "linear + statistics: d_out > 128");
  static bool raised[2] = {false, false};
  if (lds > 64 * 1024 && !raised[stat ? 1 : 0]) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess)
      return fail(SN_ERR_LAUNCH, "sn_masked_linear_f32: cannot raise the dynamic LDS limit");
    raised[stat ? 1 : 0] = true;
  }
  const int64_t ntiles = cdiv(a.R, 16);
  if constexpr (CAN_STAT) {
    if (stat) {
      hipLaunchKernelGGL((k_linear_lds<NTI, NTO, true>), dim3((unsigned)blocks), dim3(64 * LIN_W), lds, st, a, ntiles, stat);
      return SN_OK;
    }
  }
  hipLaunchKernelGGL((k_linear_lds<NTI, NTO, false>), dim3((unsigned)blocks), dim3(64 * LIN_W), lds, st, a, ntiles, stat);
  return SN_OK;
}
// returns 1 when the shape does not take the LDS path (caller falls back), SN_OK / an error otherwise
static int launch_linear_lds(const LinArgs& a, float* stat, hipStream_t st) {
  const int64_t blocks = lin_lds_blocks(a.R, a.d_in, a.d_out);
  if (blocks == 0) return 1;
  size_t lds = (size_t)a.nti * a.nto * 1024;
  const size_t xch = (size_t)(LIN_W * 2 * 16 * 8 + LIN_W) * sizeof(float);                      // the workgroup's moment exchange
  if (stat && lds < xch) lds = xch;
  if (a.nti <= 8 && a.nto <= 8) return launch_linear_lds_t<8, 8>(a, stat, blocks, lds, st);
  if (a.nti <= 16 && a.nto <= 8) return launch_linear_lds_t<16, 8>(a, stat, blocks, lds, st);
  if (a.nti <= 8) return launch_linear_lds_t<8, 16>(a, stat, blocks, lds, st);
  return launch_linear_lds_t<16, 16>(a, stat, blocks, lds, st);
}

static int masked_linear_impl(const float* x, int ldx, int64_t R, int d_in, const float* Wp, int d_out,
                              const float* bias, const int32_t* nvalid, int K, int flags,
                              const float* scale, const float* shift, const float* residual, int ldr,
                              float* y, int ldy, const float* bbias, int64_t bb_rows, int ldbb, void* stream) {
  SN_REQUIRE(x && Wp && y && R >= 0 && d_in > 0 && d_out > 0, "sn_masked_linear_f32: bad arguments");
  SN_REQUIRE(!(flags & SN_EPI_BLOCK_BIAS) || (bbias && bb_rows > 0 && ldbb >= d_out), "sn_masked_linear_blockbias_f32: block bias missing");
  SN_REQUIRE(ldx >= d_in && ldy >= d_out, "sn_masked_linear_f32: leading dimension too small");
  SN_REQUIRE((flags & (SN_EPI_RELU | SN_EPI_LEAKY)) != (SN_EPI_RELU | SN_EPI_LEAKY), "sn_masked_linear_f32: RELU and LEAKY are exclusive");
  SN_REQUIRE(!(flags & SN_EPI_BIAS) || bias, "sn_masked_linear_f32: BIAS without bias");
  SN_REQUIRE(!(flags & SN_EPI_AFFINE) || (scale && shift), "sn_masked_linear_f32: AFFINE without scale/shift");
  SN_REQUIRE(!(flags & (SN_EPI_RESIDUAL | SN_EPI_RESIDUAL_PRE)) || (residual && ldr >= d_out), "sn_masked_linear_f32: RESIDUAL without residual");
  SN_REQUIRE((flags & (SN_EPI_RESIDUAL | SN_EPI_RESIDUAL_PRE)) != (SN_EPI_RESIDUAL | SN_EPI_RESIDUAL_PRE),
             "sn_masked_linear_f32: the residual goes in front of the affine or behind it, not both");
  SN_REQUIRE(!nvalid || K > 0, "sn_masked_linear_f32: nvalid needs K > 0");
  SN_REQUIRE(al16(Wp), "sn_masked_linear_f32: Wp must be 16-byte aligned");
  if (R == 0) return SN_OK;
  LinArgs a{x, ldx, R, d_in, reinterpret_cast<const float4*>(Wp), (int)cdiv(d_in, 16), d_out, (int)cdiv(d_out, 16),
            bias, nvalid, K, flags, scale, shift, residual, ldr, y, ldy, bbias, bb_rows, ldbb};
  const bool xv = (d_in % 4 == 0) && (ldx % 4 == 0) && al16(x);
  const bool yv = (d_out % 4 == 0) && (ldy % 4 == 0) && al16(y) && (!bias || al16(bias)) && (!scale || al16(scale)) &&
                  (!shift || al16(shift)) && (!residual || (al16(residual) && ldr % 4 == 0)) && (!bbias || (al16(bbias) && ldbb % 4 == 0));
  dim3 grid((unsigned)cdiv(R, 64)), block(256);
  {
    const int64_t gx = cdiv(R, 64), nto = cdiv(d_out, 16);
    int64_t ny = gx >= 512 ? 1 : cdiv(512, gx);
    grid.y = (unsigned)(ny < nto ? ny : nto);
  }
  hipStream_t st = (hipStream_t)stream;
  if (xv && yv && R >= LIN_LDS_MIN_ROWS) {
    const int rc = launch_linear_lds(a, nullptr, st);
    if (rc != 1) { if (rc != SN_OK) return rc; SN_CHECK_LAUNCH("sn_masked_linear_f32"); return SN_OK; }
  }
  if (a.nti >= 32 && a.nto <= 4 && R <= 8192) {        // deep, narrow, few rows: split K over the waves of a workgroup
    const dim3 gk((unsigned)cdiv(R, 16));
    if (xv && yv) hipLaunchKernelGGL((k_linear_ksplit<true, true>), gk, block, 0, st, a);
    else if (xv) hipLaunchKernelGGL((k_linear_ksplit<true, false>), gk, block, 0, st, a);
    else if (yv) hipLaunchKernelGGL((k_linear_ksplit<false, true>), gk, block, 0, st, a);
    else hipLaunchKernelGGL((k_linear_ksplit<false, false>), gk, block, 0, st, a);
    SN_CHECK_LAUNCH("sn_masked_linear_f32");
    return SN_OK;
  }
  if (xv && yv) hipLaunchKernelGGL((k_linear<true, true>), grid, block, 0, st, a);
  else if (xv) hipLaunchKernelGGL((k_linear<true, false>), grid, block, 0, st, a);
  else if (yv) hipLaunchKernelGGL((k_linear<false, true>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((k_linear<false, false>), grid, block, 0, st, a);
  SN_CHECK_LAUNCH("sn_masked_linear_f32");
  return SN_OK;
}

extern "C" int sn_masked_linear_f32(const float* x, int ldx, int64_t R, int d_in, const float* Wp, int d_out,
                                    const float* bias, const int32_t* nvalid, int K, int flags,
                                    const float* scale, const float* shift, const float* residual, int ldr,
                                    float* y, int ldy, void* stream) {
  SN_REQUIRE(!(flags & SN_EPI_BLOCK_BIAS), "sn_masked_linear_f32: SN_EPI_BLOCK_BIAS needs sn_masked_linear_blockbias_f32");
  return masked_linear_impl(x, ldx, R, d_in, Wp, d_out, bias, nvalid, K, flags, scale, shift, residual, ldr, y, ldy, nullptr, 0, 0, stream);
}

extern "C" int sn_masked_linear_blockbias_f32(const float* x, int ldx, int64_t R, int d_in, const float* Wp, int d_out,
                                              const float* bias, const float* block_bias, int64_t rows_per_block, int ldbb, int flags,
                                              const float* scale, const float* shift, float* y, int ldy, void* stream) {
  SN_REQUIRE(x && Wp && y && block_bias && rows_per_block > 0 && !(flags & SN_EPI_RESIDUAL), "sn_masked_linear_blockbias_f32: bad arguments");
  return masked_linear_impl(x, ldx, R, d_in, Wp, d_out, bias, nullptr, 0, flags | SN_EPI_BLOCK_BIAS, scale, shift, nullptr, 0, y, ldy,
                            block_bias, rows_per_block, ldbb, stream);
}

extern "C" int sn_gin_aggregate_f32(const float* x, float* out, int64_t N, int F, const int32_t* rowptr,
                                    const int32_t* col, const float* eps, int negate, void* stream) {
  SN_REQUIRE(x && out && rowptr && N >= 0 && F > 0, "sn_gin_aggregate_f32: bad arguments");
  SN_REQUIRE(x != out, "sn_gin_aggregate_f32: in-place aggregation is not supported");
  if (N == 0) return SN_OK;
  hipStream_t st = (hipStream_t)stream;
  if (F % 4 == 0 && al16(x) && al16(out)) {
    dispatch_gin_gather<float4>(reinterpret_cast<const float4*>(x), reinterpret_cast<float4*>(out), N, F / 4, rowptr, col, eps, negate, st);
  } else {
    dispatch_gin_gather<float>(x, out, N, F, rowptr, col, eps, negate, st);
  }
  SN_CHECK_LAUNCH("sn_gin_aggregate_f32");
  return SN_OK;
}

/* out = aggregate(x) + plus: the adjoint of a GIN aggregation whose input also feeds a residual (x -> aggregate, x -> + y): the residual's
 * gradient is added in the same pass instead of a separate elementwise launch.  F % 4 == 0, 16-byte aligned rows. */
extern "C" int sn_gin_aggregate_add_f32(const float* x, const float* plus, float* out, int64_t N, int F, const int32_t* rowptr,
                                        const int32_t* col, const float* eps, void* stream) {
  SN_REQUIRE(x && plus && out && rowptr && N >= 0 && F > 0 && F % 4 == 0 && al16(x) && al16(out) && al16(plus),
             "sn_gin_aggregate_add_f32: bad arguments (F % 4 == 0, 16-byte aligned rows)");
  SN_REQUIRE(x != out, "sn_gin_aggregate_add_f32: in-place aggregation is not supported");
  if (N == 0) return SN_OK;
  dispatch_gin_gather<float4>(reinterpret_cast<const float4*>(x), reinterpret_cast<float4*>(out), N, F / 4, rowptr, col, eps, 0,
                              (hipStream_t)stream, reinterpret_cast<const float4*>(plus));
  SN_CHECK_LAUNCH("sn_gin_aggregate_add_f32");
  return SN_OK;
}

extern "C" int sn_gin_aggregate_slab_f32(const float* x, float* out, int64_t N, int F, int64_t B,
                                         const int32_t* graph_ptr, const int32_t* rowptr, const int32_t* col,
                                         const float* eps, int negate, void* stream) {
  SN_REQUIRE(x && out && rowptr && graph_ptr && N >= 0 && F > 0 && B >= 0, "sn_gin_aggregate_slab_f32: bad arguments");
  SN_REQUIRE(x != out, "sn_gin_aggregate_slab_f32: in-place aggregation is not supported");
  if (N == 0 || B == 0) return SN_OK;
  hipStream_t st = (hipStream_t)stream;
  const int lds = 64 * 1024;  // 2 workgroups per CU
  if (F % 4 == 0 && al16(x) && al16(out)) {
    int FV = F / 4;
    int CH = FV < 64 ? FV : 64;
    int nchunk = (int)cdiv(FV, CH);
    hipLaunchKernelGGL((k_gin_slab<float4>), dim3((unsigned)(B * nchunk)), dim3(256), lds, st,
                       reinterpret_cast<const float4*>(x), reinterpret_cast<float4*>(out), FV, CH, nchunk, graph_ptr,
                       rowptr, col, eps, negate, lds);
  } else {
    int CH = F < 256 ? F : 256;
    int nchunk = (int)cdiv(F, CH);
    hipLaunchKernelGGL((k_gin_slab<float>), dim3((unsigned)(B * nchunk)), dim3(256), lds, st, x, out, F, CH, nchunk,
                       graph_ptr, rowptr, col, eps, negate, lds);
  }
  SN_CHECK_LAUNCH("sn_gin_aggregate_slab_f32");
  return SN_OK;
}

extern "C" int sn_gine_aggregate_f32(const float* x, const float* ea, float* out, int64_t N, int C,
                                     const int32_t* rowptr, const int32_t* col, const int32_t* eperm,
                                     const float* eps, void* stream) {
  SN_REQUIRE(x && ea && out && rowptr && N >= 0 && C > 0, "sn_gine_aggregate_f32: bad arguments");
  SN_REQUIRE(x != out, "sn_gine_aggregate_f32: in-place aggregation is not supported");
  if (N == 0) return SN_OK;
  hipStream_t st = (hipStream_t)stream;
  auto grid = [&](int cols, int& chunk) {       // workgroups, padded to whole XCD rounds (the kernel drops the tail)
    const int64_t nodes = cols >= 64 ? SN_GINE_XCD_NODES : 4 * SN_GINE_XCD_NODES;
    chunk = (int)std::max<int64_t>(1, nodes * cols / (256 * GINE_U));
    const int64_t round = (int64_t)8 * chunk;
    return dim3((unsigned)(cdiv(cdiv(cdiv(N, GINE_U) * cols, 256), round) * round));
  };
  int chunk = 1;
  if (C % 4 == 0 && al16(x) && al16(ea) && al16(out)) {
    const int CV = C / 4;
    const dim3 g = grid(CV, chunk);
    hipLaunchKernelGGL((k_gine_gather<float4>), g, dim3(256), 0, st,
                       reinterpret_cast<const float4*>(x), reinterpret_cast<const float4*>(ea),
                       reinterpret_cast<float4*>(out), N, CV, rowptr, col, eperm, eps, chunk);
  } else {
    const dim3 g = grid(C, chunk);
    hipLaunchKernelGGL((k_gine_gather<float>), g, dim3(256), 0, st, x, ea, out, N, C, rowptr, col, eperm, eps, chunk);
  }
  SN_CHECK_LAUNCH("sn_gine_aggregate_f32");
  return SN_OK;
}

constexpr int CS_SPLIT = 16;      // second-level partial rows of the column-statistics reduction
extern "C" int sn_colstats_blocks(int64_t R) {
  int64_t b = cdiv(R, 64);
  b = b < 1 ? 1 : (b > 2048 ? 2048 : b);
  return (int)(b + CS_SPLIT);      // + room for the second-level partials (callers size scratch as nblocks*(C+1))
}

extern "C" int sn_masked_colstats_f32(const float* x, int ldx, int64_t R, int C, const int32_t* nvalid, int K,
                                      float* mean, float* var, float* count, float* scratch, void* stream) {
  SN_REQUIRE(x && mean && var && count && scratch && C > 0 && R >= 0 && ldx >= C, "sn_masked_colstats_f32: bad arguments");
  SN_REQUIRE(!nvalid || K > 0, "sn_masked_colstats_f32: nvalid needs K > 0");
  SN_REQUIRE(R < (1ll << 31), "sn_masked_colstats_f32: too many rows");
  hipStream_t st = (hipStream_t)stream;
  const int ntot = sn_colstats_blocks(R), nblk = ntot - CS_SPLIT;
  int64_t rpb = cdiv(R > 0 ? R : 1, nblk);
  float* part = scratch;                              // [nblk][C] then [CS_SPLIT][C]
  float* part2 = scratch + (int64_t)nblk * C;
  float* cnt = scratch + (int64_t)ntot * C;           // [nblk] then [CS_SPLIT]
  float* cnt2 = cnt + nblk;
  const bool two = nblk > 2 * CS_SPLIT;
  const dim3 rgrid((unsigned)cdiv(C + 1, 256), CS_SPLIT);
  hipLaunchKernelGGL(k_colstats_partial, dim3(nblk), dim3(256), 0, st, x, ldx, R, C, nvalid, K, (const float*)nullptr, 0,
                     rpb, part, cnt);
  if (two) hipLaunchKernelGGL(k_colstats_reduce, rgrid, dim3(256), 0, st, (const float*)part, (const float*)cnt, nblk, C, part2, cnt2);
  hipLaunchKernelGGL(k_colstats_final, dim3(1), dim3(256), 0, st, two ? part2 : part, two ? cnt2 : cnt, two ? CS_SPLIT : nblk, C,
                     mean, count, 0);
  hipLaunchKernelGGL(k_colstats_partial, dim3(nblk), dim3(256), 0, st, x, ldx, R, C, nvalid, K, (const float*)mean, 1,
                     rpb, part, (float*)nullptr);
  if (two) hipLaunchKernelGGL(k_colstats_reduce, rgrid, dim3(256), 0, st, (const float*)part, (const float*)nullptr, nblk, C, part2,
                              (float*)nullptr);
  hipLaunchKernelGGL(k_colstats_final, dim3(1), dim3(256), 0, st, two ? part2 : part, cnt, two ? CS_SPLIT : nblk, C, var, count, 1);
  SN_CHECK_LAUNCH("sn_masked_colstats_f32");
  return SN_OK;
}

static void launch_bn_finish(const float* pmean, const float* pm2, const float* pcnt, int nblk, int C, const float* gamma, const float* beta,
                             float eps, float momentum, float* mean, float* var, float* rstd, float* scale, float* shift, float* count,
                             float* running_mean, float* running_var, hipStream_t st) {
  if (nblk > 256)
    hipLaunchKernelGGL(k_bn_train_finish1<64>, dim3((unsigned)cdiv(C, 4)), dim3(256), 0, st, pmean, pm2, pcnt, nblk, C, gamma, beta, eps,
                       momentum, mean, var, rstd, scale, shift, count, running_mean, running_var);
  else
    hipLaunchKernelGGL(k_bn_train_finish1<16>, dim3((unsigned)cdiv(C, 16)), dim3(256), 0, st, pmean, pm2, pcnt, nblk, C, gamma, beta, eps,
                       momentum, mean, var, rstd, scale, shift, count, running_mean, running_var);
}

extern "C" int sn_bn_train_stats_f32(const float* x, int ldx, int64_t R, int C, const int32_t* nvalid, int K, const float* gamma,
                                     const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                                     float* mean, float* var, float* rstd, float* scale, float* shift, float* count,
                                     float* scratch, void* stream) {
  SN_REQUIRE(x && mean && var && rstd && scale && shift && count && scratch && C > 0 && R >= 0 && ldx >= C,
             "sn_bn_train_stats_f32: bad arguments");
  SN_REQUIRE(!nvalid || K > 0, "sn_bn_train_stats_f32: nvalid needs K > 0");
  SN_REQUIRE((running_mean != nullptr) == (running_var != nullptr), "sn_bn_train_stats_f32: running_mean / running_var go together");
  SN_REQUIRE(R < (1ll << 31), "sn_bn_train_stats_f32: too many rows");
  hipStream_t st = (hipStream_t)stream;
  // scratch is sized for the two-pass form ((nblk + CS_SPLIT) * (C + 1) floats): half as many blocks, two moment planes + counts
  const int nfull = sn_colstats_blocks(R) - CS_SPLIT;
  const int nblk = nfull > 1 ? nfull / 2 : 1;
  const int64_t rpb = cdiv(R > 0 ? R : 1, nblk);
  float* pmean = scratch;
  float* pm2 = scratch + (int64_t)nblk * C;
  float* pcnt = scratch + (int64_t)2 * nblk * C;
  hipLaunchKernelGGL(k_colstats_moments, dim3(nblk), dim3(256), 0, st, x, ldx, R, C, nvalid, K, rpb, pmean, pm2, pcnt);
  launch_bn_finish(pmean, pm2, pcnt, nblk, C, gamma, beta, eps, momentum, mean, var, rstd, scale, shift, count, running_mean, running_var, st);
  SN_CHECK_LAUNCH("sn_bn_train_stats_f32");
  return SN_OK;
}

extern "C" int64_t sn_linear_bn_scratch_floats(int64_t R, int d_in, int d_out) {
  const int64_t a = (int64_t)sn_colstats_blocks(R) * (d_out + 1);
  const int64_t w = (d_out <= 128 && d_in <= 128) ? lin_lds_blocks(R, d_in, d_out) : 0;
  const int64_t b = w * (2 * (int64_t)d_out + 1);
  return a > b ? a : b;
}

extern "C" int sn_linear_bn_train_f32(const float* x, int ldx, int64_t R, int d_in, const float* Wp, int d_out, const float* bias,
                                      const int32_t* nvalid, int K, float* z, int ldz, const float* gamma, const float* beta, float eps,
                                      float momentum, float* running_mean, float* running_var, float* mean, float* var, float* rstd,
                                      float* scale, float* shift, float* count, float* scratch, void* stream) {
  SN_REQUIRE(x && Wp && z && mean && var && rstd && scale && shift && count && scratch && R >= 0 && d_in > 0 && d_out > 0,
             "sn_linear_bn_train_f32: bad arguments");
  SN_REQUIRE(ldx >= d_in && ldz >= d_out, "sn_linear_bn_train_f32: leading dimension too small");
  SN_REQUIRE(!nvalid || K > 0, "sn_linear_bn_train_f32: nvalid needs K > 0");
  SN_REQUIRE((running_mean != nullptr) == (running_var != nullptr), "sn_linear_bn_train_f32: running_mean / running_var go together");
  SN_REQUIRE(al16(Wp), "sn_linear_bn_train_f32: Wp must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const bool vec = (d_in % 4 == 0) && (ldx % 4 == 0) && al16(x) && (d_out % 4 == 0) && (ldz % 4 == 0) && al16(z) && (!bias || al16(bias));
  const int64_t nblk = (vec && d_out <= 128 && d_in <= 128 && R >= LIN_LDS_MIN_ROWS) ? lin_lds_blocks(R, d_in, d_out) : 0;
  if (nblk == 0) {     // small / unaligned / wide: the Linear, then the one-pass statistics
    const int rc = sn_masked_linear_f32(x, ldx, R, d_in, Wp, d_out, bias, nvalid, K, bias ? SN_EPI_BIAS : 0, nullptr, nullptr, nullptr, 0, z,
                                        ldz, stream);
    if (rc != SN_OK) return rc;
    return sn_bn_train_stats_f32(z, ldz, R, d_out, nvalid, K, gamma, beta, eps, momentum, running_mean, running_var, mean, var, rstd, scale,
                                 shift, count, scratch, stream);
  }
  LinArgs a{x, ldx, R, d_in, reinterpret_cast<const float4*>(Wp), (int)cdiv(d_in, 16), d_out, (int)cdiv(d_out, 16),
            bias, nvalid, K, bias ? SN_EPI_BIAS : 0, nullptr, nullptr, nullptr, 0, z, ldz};
  const int rc = launch_linear_lds(a, scratch, st);
  if (rc != SN_OK) return rc == 1 ? fail(SN_ERR_LAUNCH, "sn_linear_bn_train_f32: internal path selection") : rc;
  const float* pmean = scratch;
  const float* pm2 = scratch + nblk * d_out;
  const float* pcnt = scratch + 2 * nblk * d_out;
  launch_bn_finish(pmean, pm2, pcnt, (int)nblk, d_out, gamma, beta, eps, momentum, mean, var, rstd, scale, shift, count, running_mean,
                   running_var, st);
  SN_CHECK_LAUNCH("sn_linear_bn_train_f32");
  return SN_OK;
}

extern "C" int sn_masked_affine_f32(const float* x, int ldx, int64_t R, int C, const int32_t* nvalid, int K,
                                    int flags, const float* scale, const float* shift, const float* residual,
                                    int ldr, float* y, int ldy, void* stream) {
  SN_REQUIRE(x && y && C > 0 && R >= 0 && ldx >= C && ldy >= C, "sn_masked_affine_f32: bad arguments");
  SN_REQUIRE(!(flags & SN_EPI_AFFINE) || (scale && shift), "sn_masked_affine_f32: AFFINE without scale/shift");
  SN_REQUIRE(!(flags & SN_EPI_RESIDUAL) || (residual && ldr >= C), "sn_masked_affine_f32: RESIDUAL without residual");
  SN_REQUIRE(!nvalid || K > 0, "sn_masked_affine_f32: nvalid needs K > 0");
  if (R == 0) return SN_OK;
  hipLaunchKernelGGL(k_affine, dim3((unsigned)cdiv(R * C, 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, R, C,
                     nvalid, K, flags, scale, shift, residual, ldr, y, ldy);
  SN_CHECK_LAUNCH("sn_masked_affine_f32");
  return SN_OK;
}

extern "C" int sn_masked_layernorm_f32(const float* x, const float* residual, int64_t R, int C, const float* gamma,
                                       const float* beta, float eps, const int32_t* nvalid, int K, float* y,
                                       void* stream) {
  SN_REQUIRE(x && y && gamma && beta && C > 0 && R >= 0, "sn_masked_layernorm_f32: bad arguments");
  SN_REQUIRE(!nvalid || K > 0, "sn_masked_layernorm_f32: nvalid needs K > 0");
  if (R == 0) return SN_OK;
  hipLaunchKernelGGL(k_layernorm, dim3((unsigned)cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, x, residual, R, C,
                     gamma, beta, eps, nvalid, K, y);
  SN_CHECK_LAUNCH("sn_masked_layernorm_f32");
  return SN_OK;
}

extern "C" int sn_set_attention_f32(const float* q, const float* k, const float* v, int64_t N, int K, int heads,
                                    int dk, const int32_t* nvalid, const float* prob_mask, float* out, void* stream) {
  SN_REQUIRE(q && k && v && out && N >= 0 && K > 0 && heads > 0 && dk > 0, "sn_set_attention_f32: bad arguments");
  size_t lds = ((size_t)3 * K * dk + (size_t)K * (K + 1)) * sizeof(float);
  SN_REQUIRE(lds <= 160 * 1024, "sn_set_attention_f32: K=%d dk=%d needs %zu B of LDS (> 160 KiB)", K, dk, lds);
  if (N == 0) return SN_OK;
  if (sn::attention16_forward(q, k, v, N, K, heads, dk, nvalid, prob_mask, out, (hipStream_t)stream)) {     // K <= 16: matrix pipe
    SN_CHECK_LAUNCH("sn_set_attention_f32");
    return SN_OK;
  }
  if (lds > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_set_attention), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return fail(SN_ERR_LAUNCH, "sn_set_attention_f32: cannot raise dynamic LDS limit");
  }
  hipLaunchKernelGGL(k_set_attention, dim3((unsigned)(N * heads)), dim3(64), lds, (hipStream_t)stream, q, k, v, K, heads,
                     dk, nvalid, prob_mask, out);
  SN_CHECK_LAUNCH("sn_set_attention_f32");
  return SN_OK;
}

namespace sn {
__global__ __launch_bounds__(256) void k_keep_mask(float* __restrict__ u, int64_t n, float p, float scale) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < n) {
    float4 v = *reinterpret_cast<float4*>(u + i);
    v.x = v.x >= p ? scale : 0.f; v.y = v.y >= p ? scale : 0.f; v.z = v.z >= p ? scale : 0.f; v.w = v.w >= p ? scale : 0.f;
    *reinterpret_cast<float4*>(u + i) = v;
  } else {
    for (int64_t j = i; j < n; ++j) u[j] = u[j] >= p ? scale : 0.f;
  }
}
}  // namespace sn

extern "C" int sn_keep_mask_f32(float* u, int64_t n, float p, float scale, void* stream) {
  SN_REQUIRE(u && n >= 0 && (reinterpret_cast<uintptr_t>(u) & 15) == 0, "sn_keep_mask_f32: bad arguments");
  if (n == 0) return SN_OK;
  hipLaunchKernelGGL(sn::k_keep_mask, dim3((unsigned)sn::cdiv(sn::cdiv(n, 4), 256)), dim3(256), 0, (hipStream_t)stream, u, n, p, scale);
  SN_CHECK_LAUNCH("sn_keep_mask_f32");
  return SN_OK;
}

extern "C" int sn_slot_sum_f32(const float* x, int64_t N, int K, int C, float* out, void* stream) {
  SN_REQUIRE(x && out && N >= 0 && K > 0 && C > 0, "sn_slot_sum_f32: bad arguments");
  if (N == 0) return SN_OK;
  hipLaunchKernelGGL(k_slot_sum, dim3((unsigned)cdiv(N * C, 256)), dim3(256), 0, (hipStream_t)stream, x, N, K, C, out);
  SN_CHECK_LAUNCH("sn_slot_sum_f32");
  return SN_OK;
}

extern "C" int sn_embedding_sum_f32(const int64_t* idx, int ldi, int nf, int64_t R, const float* const* tables,
                                    const int64_t* table_rows, int C, float* out, int32_t* status, void* stream) {
  SN_REQUIRE(idx && tables && table_rows && out && nf > 0 && nf <= 10 && ldi >= nf && C > 0 && R >= 0,
             "sn_embedding_sum_f32: bad arguments");
  if (R == 0) return SN_OK;
  TablePtrs tp;
  for (int f = 0; f < 10; ++f) { tp.t[f] = f < nf ? tables[f] : nullptr; tp.rows[f] = f < nf ? table_rows[f] : 0; }
  for (int f = 0; f < nf; ++f) SN_REQUIRE(tp.t[f] && tp.rows[f] > 0, "sn_embedding_sum_f32: table %d missing or empty", f);
  hipLaunchKernelGGL(k_embedding_sum, dim3((unsigned)cdiv(R * C, 256)), dim3(256), 0, (hipStream_t)stream, idx, ldi, nf,
                     R, tp, C, out, status);
  SN_CHECK_LAUNCH("sn_embedding_sum_f32");
  return SN_OK;
}

extern "C" int sn_embedding_sum_layers_f32(const int64_t* idx, int ldi, int nf, int64_t R, int L, const float* const* tables,
                                           const int64_t* table_rows, int C, float* out, int32_t* status, void* stream) {
  SN_REQUIRE(idx && tables && table_rows && out && nf > 0 && nf <= 4 && ldi >= nf && C > 0 && R >= 0 && L >= 1 && L <= 16,
             "sn_embedding_sum_layers_f32: bad arguments (nf <= 4, L <= 16)");
  if (R == 0) return SN_OK;
  LayerTablePtrs tp;
  for (int f = 0; f < 4; ++f) tp.rows[f] = f < nf ? table_rows[f] : 0;
  for (int l = 0; l < 16; ++l)
    for (int f = 0; f < 4; ++f) tp.t[l][f] = (l < L && f < nf) ? tables[l * nf + f] : nullptr;
  for (int l = 0; l < L; ++l)
    for (int f = 0; f < nf; ++f) SN_REQUIRE(tp.t[l][f] && tp.rows[f] > 0, "sn_embedding_sum_layers_f32: table %d of layer %d missing or empty", f, l);
  hipLaunchKernelGGL(k_embedding_sum_layers, dim3((unsigned)cdiv(R * C, 256), (unsigned)L), dim3(256), 0, (hipStream_t)stream, idx, ldi, nf, R, tp,
                     C, out, status);
  SN_CHECK_LAUNCH("sn_embedding_sum_layers_f32");
  return SN_OK;
}

extern "C" int sn_segment_pool_f32(const float* x, int64_t B, int C, const int32_t* graph_ptr, int mode, float* out,
                                   void* stream) {
  SN_REQUIRE(x && out && graph_ptr && B >= 0 && C > 0 && (mode == 0 || mode == 1), "sn_segment_pool_f32: bad arguments");
  if (B == 0) return SN_OK;
  hipLaunchKernelGGL(k_segment_pool, dim3((unsigned)B, (unsigned)cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, x, C, graph_ptr, mode, out);
  SN_CHECK_LAUNCH("sn_segment_pool_f32");
  return SN_OK;
}

// ============================================================================ IGN 2->1 contractions (BasisNet)
// Replaces contractions_2_to_1 (LearningFilters/ign.py:344-374, normalization 'inf') for X [b, n, n]:
//   ops[b, i, :] = [ X_ii, tr(X)/n, rowsum_i/n, colsum_i/n, sum(X)/n^2 ]        (row-major [b, n, 5])
// HBM-bound: every X element is read exactly once (4*b*n^2 bytes).  Stage A: one workgroup per (matrix,
// 64- or 128-row strip) — row sums finished in the block, column sums as per-strip partials (deterministic, no float
// atomics).  Stage B: one workgroup per matrix folds the partials.
namespace sn {
// rows per workgroup: 128 for n >= 512, 64 below (the scratch is sized for 64).  A workgroup pays a fixed price — start-up, eight
// barriers of the column fold, the partials' store — per strip: with 32-row strips the 2 GB launch of the grid ran at 4.5 TB/s,
// with 64 / 128 rows 4.9-5.3 / 5.1-5.4 (64 x 1024^2: 3.4 / 4.2 / 4.8 TB/s; 2048 x 256^2: 4.4 / 4.5-4.7 / 4.2-4.3).
constexpr int IGN_STRIP_MIN = 64;
__host__ __device__ constexpr int ign_strip(int n) { return n >= 512 ? 128 : IGN_STRIP_MIN; }

__global__ __launch_bounds__(256) void k_ign_rowcol(const float* __restrict__ X, int n, int nstrips,
                                                    float* __restrict__ rowsum /* [b,n] */, float* __restrict__ diag /* [b,n] */,
                                                    float* __restrict__ colpart /* [b,nstrips,n] */, int strip) {
  __shared__ float red[4];
  const int b = blockIdx.x / nstrips, st = blockIdx.x - b * nstrips;
  const int r0 = st * strip, r1 = (r0 + strip < n) ? r0 + strip : n;
  const float* Xb = X + (int64_t)b * n * n;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int c0 = 0; c0 < n; c0 += 256 * 4) {          // column panels of 1024
    float ca[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r = r0; r < r1; ++r) {
      float rs = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = c0 + k * 256 + tid;
        if (c < n) {
          const float v = Xb[(int64_t)r * n + c];
          ca[k] += v;
          rs += v;
          if (c == r) diag[(int64_t)b * n + r] = v;
        }
      }
      rs = wave_sum(rs);
      if (lane == 0) red[wave] = rs;
      __syncthreads();
      if (tid == 0) {
        const float t = (red[0] + red[1]) + (red[2] + red[3]);
        if (c0 == 0) rowsum[(int64_t)b * n + r] = t; else rowsum[(int64_t)b * n + r] += t;
      }
      __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = c0 + k * 256 + tid;
      if (c < n) colpart[((int64_t)b * nstrips + st) * n + c] = ca[k];
    }
  }
}

// n % 4 == 0 and 16-byte aligned rows: a wave owns whole rows (float4 per lane, 4 per 1024-column panel), row sums are wave
// reductions and the column partials stay in registers until one LDS fold per panel — no barrier inside the row loop, so
// the loads of several rows are in flight per wave (the scalar kernel above synchronises the block twice per row).
__global__ __launch_bounds__(256) void k_ign_rowcol_v4(const float* __restrict__ X, int n, int nstrips,
                                                       float* __restrict__ rowsum, float* __restrict__ diag,
                                                       float* __restrict__ colpart, int strip) {
  __shared__ float4 fold[4][256];
  const int b = blockIdx.x / nstrips, st = blockIdx.x - b * nstrips;
  const int r0 = st * strip, r1 = (r0 + strip < n) ? r0 + strip : n;
  const float* Xb = X + (int64_t)b * n * n;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int c0 = 0; c0 < n; c0 += 1024) {
    float4 ca[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) ca[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int rr = r0 + wave; rr < r1; rr += 8) {          // two rows (rr, rr + 4) per iteration: eight float4 loads in flight
      float4 v[2][4];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int r = rr + 4 * u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int c = c0 + j * 256 + lane * 4;
          v[u][j] = (c < n && r < r1) ? *reinterpret_cast<const float4*>(Xb + (int64_t)r * n + c) : make_float4(0.f, 0.f, 0.f, 0.f);
          // (nontemporal loads were measured slower here: 731 vs 600 us for the 2 GB launch)
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int r = rr + 4 * u;
        float rs = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          rs += (v[u][j].x + v[u][j].y) + (v[u][j].z + v[u][j].w);
          ca[j].x += v[u][j].x; ca[j].y += v[u][j].y; ca[j].z += v[u][j].z; ca[j].w += v[u][j].w;
        }
        rs = wave_sum(rs);
        if (r < r1) {
          const int dc = r - c0;                       // the diagonal element of this row, if it lies in the panel
          if (dc >= 0 && dc < 1024 && ((dc & 255) >> 2) == lane) {
            // (static indices only: `v[u][dc >> 8]` made the compiler keep all of v[][] in scratch memory — every loaded
            //  element was written to and read back from private memory, WRITE_SIZE = FETCH_SIZE in the counters)
            float dv = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              if ((dc >> 8) == j) {
                const int e = dc & 3;
                dv = e == 0 ? v[u][j].x : (e == 1 ? v[u][j].y : (e == 2 ? v[u][j].z : v[u][j].w));
              }
            }
            diag[(int64_t)b * n + r] = dv;
          }
          if (lane == 0) {
            if (c0 == 0) rowsum[(int64_t)b * n + r] = rs; else rowsum[(int64_t)b * n + r] += rs;
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      __syncthreads();
      fold[wave][lane + 64 * 0] = ca[j];             // [wave][lane]: 64 float4 used per wave
      __syncthreads();
      if (wave == 0) {
        const int c = c0 + j * 256 + lane * 4;
        if (c < n) {
          const float4 a = fold[0][lane], bq = fold[1][lane], cq = fold[2][lane], dq = fold[3][lane];
          float4 o;
          o.x = (a.x + bq.x) + (cq.x + dq.x); o.y = (a.y + bq.y) + (cq.y + dq.y);
          o.z = (a.z + bq.z) + (cq.z + dq.z); o.w = (a.w + bq.w) + (cq.w + dq.w);
          *reinterpret_cast<float4*>(colpart + ((int64_t)b * nstrips + st) * n + c) = o;
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_ign_finish(const float* __restrict__ rowsum, const float* __restrict__ diag,
                                                    const float* __restrict__ colpart, int n, int nstrips,
                                                    float* __restrict__ ops /* [b,n,5] */) {
  __shared__ float red[2][4];
  __shared__ float tot[2];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float tsum = 0.f, dsum = 0.f;
  for (int c = tid; c < n; c += 256) {
    float cs = 0.f;
    for (int s = 0; s < nstrips; ++s) cs += colpart[((int64_t)b * nstrips + s) * n + c];
    ops[((int64_t)b * n + c) * 5 + 3] = cs / (float)n;
    tsum += cs;
    dsum += diag[(int64_t)b * n + c];
  }
  tsum = wave_sum(tsum);
  dsum = wave_sum(dsum);
  if (lane == 0) { red[0][wave] = tsum; red[1][wave] = dsum; }
  __syncthreads();
  if (tid == 0) {
    tot[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    tot[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
  __syncthreads();
  const float total = tot[0] / ((float)n * (float)n), tr = tot[1] / (float)n;
  for (int c = tid; c < n; c += 256) {
    float* o = ops + ((int64_t)b * n + c) * 5;
    o[0] = diag[(int64_t)b * n + c];
    o[1] = tr;
    o[2] = rowsum[(int64_t)b * n + c] / (float)n;
    o[4] = total;
  }
}
}  // namespace sn

extern "C" int64_t sn_ign_contract_scratch_floats(int64_t b, int n) {
  return b * (2 * (int64_t)n + sn::cdiv(n, sn::IGN_STRIP_MIN) * (int64_t)n);
}

extern "C" int sn_ign_contract_2to1_f32(const float* X, int64_t b, int n, float* ops_out, float* scratch, void* stream) {
  SN_REQUIRE(X && ops_out && scratch && b >= 0 && n > 0, "sn_ign_contract_2to1_f32: bad arguments");
  if (b == 0) return SN_OK;
  // 128-row strips from n = 512 on — unless that leaves most of the chip without a workgroup (one 1024 x 1024 projector: 8 strips):
  // then the 64-row strips the scratch is sized for
  int strip = sn::ign_strip(n);
  if (b * sn::cdiv(n, strip) < 256) strip = sn::IGN_STRIP_MIN;
  const int nstrips = (int)sn::cdiv(n, strip);
  SN_REQUIRE(b * nstrips < (1ll << 31), "sn_ign_contract_2to1_f32: too many workgroups");
  float* rowsum = scratch;
  float* diag = scratch + b * n;
  float* colpart = scratch + 2 * b * n;
  hipStream_t st = (hipStream_t)stream;
  if (n % 4 == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0 && (reinterpret_cast<uintptr_t>(scratch) & 15) == 0)
    hipLaunchKernelGGL(sn::k_ign_rowcol_v4, dim3((unsigned)(b * nstrips)), dim3(256), 0, st, X, n, nstrips, rowsum, diag, colpart, strip);
  else
    hipLaunchKernelGGL(sn::k_ign_rowcol, dim3((unsigned)(b * nstrips)), dim3(256), 0, st, X, n, nstrips, rowsum, diag, colpart, strip);
  hipLaunchKernelGGL(sn::k_ign_finish, dim3((unsigned)b), dim3(256), 0, st, rowsum, diag, colpart, n, nstrips, ops_out);
  SN_CHECK_LAUNCH("sn_ign_contract_2to1_f32");
  return SN_OK;
}
