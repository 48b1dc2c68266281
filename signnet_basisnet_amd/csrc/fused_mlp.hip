// fused_mlp.hip — a whole MLP (Linear -> ReLU -> [eval BatchNorm] ... -> Linear) over the rows of a matrix in ONE launch, optionally
// fed by the masked sum over a node's eigenvector slots.  This is rho of the DGL tree's sign-invariant networks:
//   GINDeepSigns       rho = MLP(k * phi_out -> hidden -> ... -> k) on x.reshape(N, -1)               (layers/deepsigns.py:42-51)
//   MaskedGINDeepSigns rho = MLP(phi_out -> hidden -> ... -> k) on sum_{slot < n_graph} x[:, slot, :] (layers/deepsigns.py:62-86)
// with MLP = layers/mlp.py:37-56 (Linear, activation, BatchNorm per hidden layer; the last Linear plain).  In eval mode every
// BatchNorm follows a ReLU and precedes a Linear, so it is folded into that Linear's weights at pack time: the chain is
// relu(W0 x + b0), relu(W1' h + b1'), ..., W_L' h + b_L'.
// A workgroup keeps 64 rows on chip for the whole chain: activations live in registers in the MFMA operand layout, every Linear is a
// wg_gemm_split() (fp32 through the bf16 matrix pipe, weights streamed once per workgroup pass through the LDS ring).  All widths are
// zero-padded to 16*NT by the caller (sn_pack_split_f32 of the padded matrix, e0 = bias).
#include "fused_common.hpp"

namespace sn {

constexpr int MLP_R = 64;

struct MlpStruct {
  const float* x;
  int64_t R;
  int ldx, d_in;
  const int32_t* nvalid;   // != NULL: row r of the chain input = sum_{s < nvalid[r]} x[(r*K + s)*ldx + 0..d_in)
  int K;
  float* y;
  int ldy, d_out;
  int n_layers;
  const void* w[SN_MLP_MAX_LAYERS];
};

template <int NT>
__global__ __launch_bounds__(MLP_R * 4, 2) void k_mlp_chain(MlpStruct S) {
  constexpr int NKB = (NT + 1) / 2;
  using Ring = WRing<NT>;
  extern __shared__ __align__(1024) unsigned char lds_raw[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = wave * 16 + (lane & 15), g = lane >> 4;
  const int64_t nbins = (S.R + MLP_R - 1) / MLP_R;
  Ring ring;
  ring.init(lds_raw, wave, lane);
  if (nbins > (int64_t)blockIdx.x) ring.prologue(S.w[0], NT);
  for (int64_t bin = blockIdx.x; bin < nbins; bin += gridDim.x) {
    const int64_t row = bin * MLP_R + r;
    const bool valid = row < S.R;
    const bool wave_live = __ballot(valid) != 0ull;
    f32x4 in[NT];
#pragma unroll
    for (int kk = 0; kk < NT; ++kk) in[kk] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool vec4 = ((S.d_in | S.ldx) & 3) == 0;          // whole float4 groups, 16-byte aligned rows
    if (valid) {
      const int ns = S.nvalid ? S.nvalid[row] : 1;
      const float* base = S.x + (S.nvalid ? row * S.K : row) * (int64_t)S.ldx;
      for (int s = 0; s < ns; ++s) {
        const float* xr = base + (int64_t)s * S.ldx;
        if (vec4) {
#pragma unroll
          for (int kk = 0; kk < NT; ++kk) {
            const int c = 16 * kk + 4 * g;
            if (c < S.d_in) in[kk] += ld4(xr + c);
          }
        } else {
#pragma unroll
          for (int kk = 0; kk < NT; ++kk) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const int c = 16 * kk + 4 * g + t;
              if (c < S.d_in) in[kk][t] += xr[c];
            }
          }
        }
      }
    }
#pragma unroll 1
    for (int l = 0; l < S.n_layers; ++l) {
      Split8 sp[NKB];
      if (wave_live) split_rows<NT>(in, sp);
      const bool last = l + 1 == S.n_layers;
      const void* nxt = last ? S.w[0] : S.w[l + 1];            // the next bin restarts the stream at the first matrix
      wg_gemm_split<NT, NT, false>(ring, S.w[l], nxt, wave_live, sp, NoPre(), [&](int ot, f32x4 acc, f32x4 b, f32x4, f32x4, f32x4) {
        const f32x4 v = acc + b;
        in[ot] = last ? v : relu4(v);
      });
    }
    if (valid) {
      float* yr = S.y + row * (int64_t)S.ldy;
#pragma unroll
      for (int kk = 0; kk < NT; ++kk) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int c = 16 * kk + 4 * g + t;
          if (c < S.d_out) yr[c] = in[kk][t];
        }
      }
    }
  }
  ring.drain();
}

template <int NT>
static int launch_mlp(const MlpStruct& S, hipStream_t st) {
  const size_t lds = (size_t)WRing<NT>::BYTES;
  static int cus = 0;
  if (cus == 0) {
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_mlp_chain<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return fail(SN_ERR_LAUNCH, "sn_mlp_chain_f32: cannot raise the dynamic LDS limit to %zu", lds);
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    cus = n > 0 ? n : 256;
  }
  const int64_t nbins = (S.R + MLP_R - 1) / MLP_R;
  const int64_t cap = 2 * (int64_t)cus;
  const int64_t grid = nbins < cap ? nbins : cap;
  hipLaunchKernelGGL((k_mlp_chain<NT>), dim3((unsigned)grid), dim3(MLP_R * 4), lds, st, S);
  return SN_OK;
}

}  // namespace sn

using namespace sn;

extern "C" int sn_mlp_chain_f32(const float* x, int ldx, int64_t R, int d_in, const int32_t* nvalid, int K, const void* const* weights,
                                int n_layers, int d_pad, float* y, int ldy, int d_out, void* stream) {
  SN_REQUIRE(x && weights && y && R >= 0 && d_in >= 1 && d_out >= 1 && ldx >= d_in && ldy >= d_out, "sn_mlp_chain_f32: bad arguments");
  SN_REQUIRE(n_layers >= 1 && n_layers <= SN_MLP_MAX_LAYERS, "sn_mlp_chain_f32: %d layers unsupported (max %d)", n_layers, SN_MLP_MAX_LAYERS);
  SN_REQUIRE(d_pad >= 48 && d_pad <= 128 && (d_pad & 15) == 0 && d_in <= d_pad && d_out <= d_pad,
             "sn_mlp_chain_f32: padded width %d must be a multiple of 16 in [48, 128] covering d_in / d_out", d_pad);
  SN_REQUIRE(!nvalid || K > 0, "sn_mlp_chain_f32: nvalid needs K > 0");
  if (R == 0) return SN_OK;
  MlpStruct S{x, R, ldx, d_in, nvalid, K, y, ldy, d_out, n_layers, {}};
  for (int l = 0; l < n_layers; ++l) {
    SN_REQUIRE(weights[l], "sn_mlp_chain_f32: weight %d missing", l);
    S.w[l] = weights[l];
  }
  int rc = SN_OK;
  hipStream_t st = (hipStream_t)stream;
  switch (d_pad / 16) {
    case 3: rc = launch_mlp<3>(S, st); break;
    case 4: rc = launch_mlp<4>(S, st); break;
    case 5: rc = launch_mlp<5>(S, st); break;
    case 6: rc = launch_mlp<6>(S, st); break;
    case 7: rc = launch_mlp<7>(S, st); break;
    default: rc = launch_mlp<8>(S, st); break;
  }
  if (rc != SN_OK) return rc;
  SN_CHECK_LAUNCH("sn_mlp_chain_f32");
  return SN_OK;
}
