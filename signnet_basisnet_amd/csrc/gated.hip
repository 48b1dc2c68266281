// gated.hip — GatedGCN's edge-gated aggregation and its adjoint.
//
// Replaces the DGL message passing of GatedGCNLayer.forward (GraphPrediction/layers/gatedgcn_layer.py:51-56):
//   apply_edges(u_add_v('Dh','Eh')) ; e = DEh + Ce ; sigma = sigmoid(e)
//   update_all(u_mul_e('Bh','sigma'), sum) ; update_all(copy_e('sigma'), sum) ; h = Ah + sum_sigma_h / (sum_sigma + 1e-6)
// in ONE pass over the destination-sorted CSR of the batch plan: a thread owns (destination node, channel), walks the
// node's in-edges (edge-id order -> deterministic sums), writes each edge's new feature row once and the node's new
// feature.  HBM-bound gather; algorithmic bytes 4*d*(3E + 4N) (Ce in, e out, Bh/Dh gathered per edge; Ah, Eh in, h, den out).
// The adjoint is two passes without atomics: by destination (d e, d Eh, d Ah) and by source over the reverse CSR (d Dh, d Bh).
#include "common.hpp"

namespace sn {
namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// ldn: row stride of the four node arrays (they may be column blocks of ONE [N, 4C] GEMM output).  Optional fused eval-mode
// epilogue (hs/ht, es/et = folded BatchNorm scale/shift of bn_node_h / bn_node_e; hres / eres = residual inputs):
//   h_out = [hres +] relu(hs*h + ht),  e_out = [eres +] relu(es*e + et)        (gatedgcn_layer.py:64-72)
struct GatedEpi { const float *hs, *ht, *es, *et, *hres, *eres; };
__global__ __launch_bounds__(256) void k_gated_fwd(const float* __restrict__ Ah, const float* __restrict__ Bh,
                                                   const float* __restrict__ Dh, const float* __restrict__ Eh, int ldn,
                                                   const float* __restrict__ Ce, int64_t N, int C,
                                                   const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                   const int32_t* __restrict__ eperm, float* __restrict__ h_out,
                                                   float* __restrict__ e_out, float* __restrict__ den_out, GatedEpi ep) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= N * C) return;
  const int64_t i = idx / C;
  const int c = (int)(idx - i * C);
  const float eh = Eh[i * ldn + c];
  const bool fuse = ep.hs != nullptr;
  const float es = fuse ? ep.es[c] : 1.f, et = fuse ? ep.et[c] : 0.f;
  float num = 0.f, den = 0.f;
  for (int s = rowptr[i]; s < rowptr[i + 1]; ++s) {
    const int64_t j = col[s], e = eperm[s];
    const float en = Dh[j * ldn + c] + eh + Ce[e * C + c];
    const float sg = sigmoidf_(en);
    float eo = en;
    if (fuse) {
      eo = fmaxf(en * es + et, 0.f);
      if (ep.eres) eo += ep.eres[e * C + c];
    }
    e_out[e * C + c] = eo;
    num += Bh[j * ldn + c] * sg;
    den += sg;
  }
  float h = Ah[i * ldn + c] + num / (den + 1e-6f);
  if (fuse) {
    h = fmaxf(h * ep.hs[c] + ep.ht[c], 0.f);
    if (ep.hres) h += ep.hres[idx];
  }
  h_out[idx] = h;
  if (den_out) den_out[idx] = den;
}

// by destination: g = dh / (den+eps); d den = -dh * (h - Ah) / (den+eps)   [since num/(den+eps) = h - Ah]
//   d sigma_e = g * Bh_src + d den ;  d e_e = de_e + d sigma_e * sigma_e (1 - sigma_e) ;  d Eh_i = sum_in d e_e ;  d Ah = dh
__global__ __launch_bounds__(256) void k_gated_bwd_dst(const float* __restrict__ Ah, const float* __restrict__ Bh,
                                                       const float* __restrict__ e_new, const float* __restrict__ h_new,
                                                       const float* __restrict__ den, const float* __restrict__ dh,
                                                       const float* __restrict__ de, int64_t N, int C,
                                                       const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                       const int32_t* __restrict__ eperm, float* __restrict__ dE,
                                                       float* __restrict__ de_new /* [E,C]: d(Dh_src + Eh_dst + Ce) */,
                                                       float* __restrict__ gnum /* [N,C]: dh/(den+eps), for the source pass */) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= N * C) return;
  const int64_t i = idx / C;
  const int c = (int)(idx - i * C);
  const float inv = 1.0f / (den[idx] + 1e-6f);
  const float g = dh[idx] * inv;
  const float dden = -g * (h_new[idx] - Ah[idx]);
  float acc = 0.f;
  for (int s = rowptr[i]; s < rowptr[i + 1]; ++s) {
    const int64_t j = col[s], e = eperm[s];
    const float sg = sigmoidf_(e_new[e * C + c]);
    const float dsg = g * Bh[j * C + c] + dden;
    const float d = (de ? de[e * C + c] : 0.f) + dsg * sg * (1.0f - sg);
    de_new[e * C + c] = d;
    acc += d;
  }
  dE[idx] = acc;
  gnum[idx] = g;
}
// by source over the reverse CSR (rcol = destination, rperm = edge id): d Dh_j = sum_out d e_e ; d Bh_j = sum_out sigma_e * g_dst
__global__ __launch_bounds__(256) void k_gated_bwd_src(const float* __restrict__ e_new, const float* __restrict__ de_new,
                                                       const float* __restrict__ gnum, int64_t N, int C,
                                                       const int32_t* __restrict__ rrow, const int32_t* __restrict__ rcol,
                                                       const int32_t* __restrict__ rperm, float* __restrict__ dD,
                                                       float* __restrict__ dB) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= N * C) return;
  const int64_t j = idx / C;
  const int c = (int)(idx - j * C);
  float ad = 0.f, ab = 0.f;
  for (int s = rrow[j]; s < rrow[j + 1]; ++s) {
    const int64_t i = rcol[s], e = rperm[s];
    ad += de_new[e * C + c];
    ab += sigmoidf_(e_new[e * C + c]) * gnum[i * C + c];
  }
  dD[idx] = ad;
  dB[idx] = ab;
}

}  // namespace
}  // namespace sn

using namespace sn;

extern "C" int sn_gated_aggregate_f32(const float* Ah, const float* Bh, const float* Dh, const float* Eh, int ldn, const float* Ce,
                                      int64_t N, int C, const int32_t* rowptr, const int32_t* col, const int32_t* eperm,
                                      float* h_out, float* e_out, float* den_out, const float* h_scale, const float* h_shift,
                                      const float* e_scale, const float* e_shift, const float* h_res, const float* e_res,
                                      void* stream) {
  SN_REQUIRE(Ah && Bh && Dh && Eh && Ce && rowptr && col && eperm && h_out && e_out && N >= 0 && C > 0 && ldn >= C,
             "sn_gated_aggregate_f32: bad arguments");
  SN_REQUIRE((h_scale != nullptr) == (h_shift != nullptr) && (h_scale != nullptr) == (e_scale != nullptr) &&
             (e_scale != nullptr) == (e_shift != nullptr), "sn_gated_aggregate_f32: the fused epilogue needs all four scale/shift vectors");
  SN_REQUIRE(h_scale || (!h_res && !e_res), "sn_gated_aggregate_f32: residuals only with the fused epilogue");
  SN_REQUIRE(!h_scale || !den_out, "sn_gated_aggregate_f32: the fused (eval) epilogue and den_out (training) exclude each other");
  if (N == 0) return SN_OK;
  GatedEpi ep{h_scale, h_shift, e_scale, e_shift, h_res, e_res};
  hipLaunchKernelGGL(k_gated_fwd, dim3((unsigned)cdiv(N * C, 256)), dim3(256), 0, (hipStream_t)stream, Ah, Bh, Dh, Eh, ldn, Ce, N, C,
                     rowptr, col, eperm, h_out, e_out, den_out, ep);
  SN_CHECK_LAUNCH("sn_gated_aggregate_f32");
  return SN_OK;
}

extern "C" int sn_gated_aggregate_bwd_f32(const float* Ah, const float* Bh, const float* e_new, const float* h_new, const float* den,
                                          const float* dh, const float* de /* may be NULL */, int64_t N, int C,
                                          const int32_t* rowptr, const int32_t* col, const int32_t* eperm,
                                          const int32_t* rev_rowptr, const int32_t* rev_col, const int32_t* rev_eperm,
                                          float* dB, float* dD, float* dE, float* de_new, float* scratch /* [N*C] */, void* stream) {
  SN_REQUIRE(Ah && Bh && e_new && h_new && den && dh && rowptr && rev_rowptr && dB && dD && dE && de_new && scratch && N >= 0 && C > 0,
             "sn_gated_aggregate_bwd_f32: bad arguments");
  if (N == 0) return SN_OK;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)cdiv(N * C, 256)), blk(256);
  hipLaunchKernelGGL(k_gated_bwd_dst, grid, blk, 0, st, Ah, Bh, e_new, h_new, den, dh, de, N, C, rowptr, col, eperm, dE, de_new, scratch);
  hipLaunchKernelGGL(k_gated_bwd_src, grid, blk, 0, st, e_new, (const float*)de_new, (const float*)scratch, N, C, rev_rowptr, rev_col,
                     rev_eperm, dD, dB);
  SN_CHECK_LAUNCH("sn_gated_aggregate_bwd_f32");
  return SN_OK;
}
