// evd.hip — batched Laplacian eigendecomposition on the device (the step before the hot path).
//
// Replaces, for a whole collated batch at once, the reference's per-sample host transform
//   EVDTransform / EVD_Laplacian   Alchemy/sign_net/transform.py:7-23, GINESignNetPyG/core/transform.py:7-26
//     (to_undirected -> get_laplacian(norm) -> dense -> torch.linalg.eigh, ascending)
//   lap_positional_encoding        GraphPrediction/data/molecules.py:148-181
//     (I - D^-1/2 A D^-1/2 -> eig -> sort -> columns 1..k, zero padded)
//
// Method: parallel one-sided (Hestenes) Jacobi in registers.  A graph of n <= NR nodes is owned by NR lanes of a
// wave (NR = 16 / 32 / 64: 4 / 2 / 1 graphs per wave); lane j holds column j of G = L*V and of V (NR registers
// each).  A step rotates n/2 disjoint column pairs at once (round-robin tournament): a lane pulls its partner's
// column through ds_bpermute, forms the three dot products in registers, and applies its half of the rotation in
// the Rutishauser form x' = x -/+ s (y +/- tau x) (keeps V orthogonal to fp32 rounding: the plain c/s form drifts,
// because c rounds to 1 for small angles while s does not).  Converged when a whole sweep rotates nothing;
// eigenvalues are the Rayleigh quotients v_j . g_j, ranked in registers, and the columns are stored in ascending
// order in the reference's wire format (eigen_values [n], eigen_vectors [n*n] row-major V[node, eig]).
// Both Laplacians are positive semi-definite, so V diagonalising G^T G = L^2 diagonalises L.
// The eigenvectors' signs (and the basis inside a repeated eigenvalue) are arbitrary, as they are in LAPACK.
#include "common.hpp"

namespace sn {
namespace {

constexpr int EVD_MAX_N = 64;
constexpr int EVD_MAX_SWEEPS = 16;
constexpr float EVD_TOL = 5e-7f;        // rotate while |g_i.g_j| > tol * |g_i||g_j|
constexpr float EVD_NOCONV_TOL = 1e-5f; // a rotation this large in the last allowed sweep = not converged
constexpr float EVD_ZERO = 1e-6f;       // columns below zero * max column norm are null-space columns

// status bits (status[0])
constexpr int EVD_ST_CROSS = 1;         // an edge leaves its graph / node id out of range
constexpr int EVD_ST_OVERSIZE = 2;      // a graph has more than 64 nodes (its outputs are not written)
constexpr int EVD_ST_NOCONV = 4;        // a graph did not converge in EVD_MAX_SWEEPS sweeps
constexpr int EVD_ST_SPACE = 8;         // eigen_vectors buffer too small

// ---- offsets of the n_b x n_b blocks + size classes (one workgroup)
__global__ __launch_bounds__(256) void k_evd_prep(const int32_t* __restrict__ graph_ptr, int B, int64_t total,
                                                    int64_t* __restrict__ evoff, int32_t* __restrict__ cls_list,
                                                    int32_t* __restrict__ cls_count, int32_t* __restrict__ status) {
  __shared__ int64_t wsum[4];
  __shared__ int64_t carry_s;
  __shared__ int cnt[3];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid < 3) cnt[tid] = 0;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  int flags = 0;
  for (int g0 = 0; g0 < B; g0 += 256) {
    const int g = g0 + tid;
    int n = 0;
    if (g < B) n = graph_ptr[g + 1] - graph_ptr[g];
    if (n < 0) { n = 0; flags |= EVD_ST_CROSS; }
    int64_t v = (int64_t)n * n, incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      int64_t t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int64_t pre = carry_s;
    for (int i = 0; i < w; ++i) pre += wsum[i];
    if (g < B) {
      evoff[g] = pre + incl - v;
      if (n > EVD_MAX_N) flags |= EVD_ST_OVERSIZE;
      else if (n > 0) {
        const int c = n <= 16 ? 0 : (n <= 32 ? 1 : 2);
        const int pos = atomicAdd(&cnt[c], 1);
        cls_list[(int64_t)c * B + pos] = g;
      }
    }
    __syncthreads();
    if (tid == 255) carry_s = pre + incl;
    __syncthreads();
  }
  if (tid == 0) {
    evoff[B] = carry_s;
    if (carry_s > total) flags |= EVD_ST_SPACE;
  }
  if (tid < 3) cls_count[tid] = cnt[tid];
  if (flags) atomicOr(&status[0], flags);
}

// ---- dense adjacency (undirected closure, self loops dropped, duplicates coalesced) into the eigenvector blocks
__global__ __launch_bounds__(256) void k_evd_scatter(const int64_t* __restrict__ edge_index, int64_t E, int64_t N,
                                                       const int32_t* __restrict__ graph_ptr, int B,
                                                       const int64_t* __restrict__ evoff, int64_t total,
                                                       float* __restrict__ vec, int32_t* __restrict__ status) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  const int64_t s = edge_index[e], d = edge_index[E + e];
  if (s == d) return;
  if (s < 0 || s >= N || d < 0 || d >= N) { atomicOr(&status[0], EVD_ST_CROSS); return; }
  int lo = 0, hi = B;                     // graph of s: last g with graph_ptr[g] <= s
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (graph_ptr[mid] <= s) lo = mid; else hi = mid;
  }
  const int n0 = graph_ptr[lo], n1 = graph_ptr[lo + 1], n = n1 - n0;
  if (d < n0 || d >= n1 || s >= n1) { atomicOr(&status[0], EVD_ST_CROSS); return; }
  if (n > EVD_MAX_N) return;
  const int64_t off = evoff[lo];
  if (off + (int64_t)n * n > total) return;
  const int a = (int)(s - n0), b = (int)(d - n0);
  vec[off + (int64_t)a * n + b] = 1.0f;
  vec[off + (int64_t)b * n + a] = 1.0f;
}

__device__ __forceinline__ float bperm(int lane, float v) {
  return __int_as_float(__builtin_amdgcn_ds_bpermute(lane << 2, __float_as_int(v)));
}

struct EvdArgs {
  const int32_t* graph_ptr;
  const int64_t* evoff;
  const int32_t* cls_list;     // [3][B]
  const int32_t* cls_count;    // [3]
  float* val;                  // [N]
  float* vec;                  // adjacency in, eigenvectors out
  float* pos_enc;              // [N, k] or null
  int32_t* status;
  int64_t total;
  int norm, k, skip;
};

// rows in blocks of 8 under a wave-uniform bound: rows >= the largest n of the wave's graphs are zero in G and V
// (real columns never mix with the padding columns), so they are skipped without dynamic register indexing
#define EVD_ROWS(...)                                   \
  _Pragma("unroll") for (int rb = 0; rb < NR / 8; ++rb) \
    if (rb * 8 < nmax) {                                \
      _Pragma("unroll") for (int q = 0; q < 8; ++q) {   \
        const int r = rb * 8 + q;                       \
        __VA_ARGS__                                     \
      }                                                 \
    }

template <int NR>
__device__ __forceinline__ void evd_jacobi(const EvdArgs& a, const int32_t* cls_list, int count, int blk) {
  constexpr int GPW = 64 / NR;
  const int lane = threadIdx.x, j = lane & (NR - 1), base = lane & ~(NR - 1);
  const int slot = blk * GPW + lane / NR;
  const bool live = slot < count;
  int n = 0, n0 = 0;
  int64_t off = 0;
  if (live) {
    const int g = cls_list[slot];
    n0 = a.graph_ptr[g];
    n = a.graph_ptr[g + 1] - n0;
    off = a.evoff[g];
    if (off + (int64_t)n * n > a.total) n = 0;
  }
  const bool col = j < n;
  const int m = (n + 1) & ~1, mm1 = m - 1;
  int mmax = m;
#pragma unroll
  for (int o = NR; o < 64; o <<= 1) mmax = max(mmax, __shfl_xor(mmax, o));
  mmax = __builtin_amdgcn_readfirstlane(mmax);
  const int nmax = mmax;                   // rows / columns >= nmax are padding in every graph of this wave
  float G[NR], V[NR], T[NR];
  // adjacency column j (= row j), degrees, Laplacian
  float deg = 0.f;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    G[r] = (col && r < n) ? a.vec[off + (int64_t)r * n + j] : 0.f;
    deg += G[r];
    V[r] = (r == j) ? 1.f : 0.f;
    T[r] = 0.f;
  }
  if (a.norm == 1) {       // get_laplacian(normalization='sym'): I - D^-1/2 A D^-1/2, 1/sqrt(0) -> 0, unit diagonal everywhere
    const float dis = deg > 0.f ? 1.0f / sqrtf(deg) : 0.f;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const float dr = bperm(base + r, dis);
      G[r] = -(G[r] * dr) * dis;
      if (r == j && col) G[r] = 1.f;
    }
  } else {                 // normalization=None: D - A
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      G[r] = -G[r];
      if (r == j) G[r] = deg;
    }
  }
  float amax = 0.f;
#pragma unroll
  for (int r = 0; r < NR; ++r) amax = fmaf(G[r], G[r], amax);
#pragma unroll
  for (int o = 1; o < NR; o <<= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
  const float athr = EVD_ZERO * EVD_ZERO * amax;

  bool pending = false;
  int sweep = 0;
  for (; sweep < EVD_MAX_SWEEPS; ++sweep) {
    bool rotated = false;
    pending = false;
    for (int step = 0; step < mmax - 1; ++step) {
      // round-robin partner inside the graph's m columns (m even): i+j = step (mod m-1), the fixed point meets m-1
      const bool active = j < m && step < mm1;
      int p = step - j;
      if (p < 0) p += mm1;
      if (p == j) p = mm1;
      if (j == mm1) p = (step & 1) ? (step + mm1) >> 1 : step >> 1;
      if (!active) p = j;
      const int pl = base + p;
      EVD_ROWS(T[r] = bperm(pl, G[r]);)
      float al0 = 0.f, al1 = 0.f, ga0 = 0.f, ga1 = 0.f;
      EVD_ROWS(if (q & 1) { al1 = fmaf(G[r], G[r], al1); ga1 = fmaf(G[r], T[r], ga1); }
               else { al0 = fmaf(G[r], G[r], al0); ga0 = fmaf(G[r], T[r], ga0); })
      const float alpha = al0 + al1, gamma = ga0 + ga1;
      const float beta = bperm(pl, alpha);
      const bool first = j < p;
      const float lo = first ? alpha : beta, hi = first ? beta : alpha;     // norm^2 of the lower / the higher column
      const bool rot = active && gamma * gamma > (EVD_TOL * EVD_TOL) * (lo * hi) && fminf(lo, hi) > athr;
      if (__ballot(rot) == 0ull) continue;
      rotated |= rot;
      pending |= rot && gamma * gamma > (EVD_NOCONV_TOL * EVD_NOCONV_TOL) * (lo * hi);
      float s = 0.f, tau = 0.f;
      if (rot) {
        const float zeta = (hi - lo) * __builtin_amdgcn_rcpf(2.0f * gamma);
        float t = __builtin_amdgcn_rcpf(fabsf(zeta) + __builtin_amdgcn_sqrtf(fmaf(zeta, zeta, 1.0f)));
        t = zeta < 0.f ? -t : t;
        const float c = __builtin_amdgcn_rsqf(fmaf(t, t, 1.0f));
        s = c * t;
        tau = s * __builtin_amdgcn_rcpf(1.0f + c);
        if (first) { s = -s; tau = -tau; }
      }
      // lower column: x - s (y + tau x);  higher column: x + s (y - tau x)   (signs folded into s, tau above)
      EVD_ROWS(G[r] = fmaf(s, fmaf(-tau, G[r], T[r]), G[r]);)
      EVD_ROWS(T[r] = bperm(pl, V[r]);)
      EVD_ROWS(V[r] = fmaf(s, fmaf(-tau, V[r], T[r]), V[r]);)
    }
    if (__ballot(rotated) == 0ull) break;
  }
  if (__ballot(pending) != 0ull && lane == 0) atomicOr(&a.status[0], EVD_ST_NOCONV);
  if (lane == 0) atomicMax(&a.status[1], sweep + 1);          // most sweeps any wave needed (diagnostic)

  // Rayleigh quotients, ascending rank (ties by column index), stores
  float lam = 0.f;
#pragma unroll
  for (int r = 0; r < NR; ++r) lam = fmaf(V[r], G[r], lam);
  int rank = 0;
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const float li = bperm(base + i, lam);
    rank += (i < n && (li < lam || (li == lam && i < j))) ? 1 : 0;
  }
  if (!col) return;
  a.val[n0 + rank] = lam;
#pragma unroll
  for (int r = 0; r < NR; ++r)
    if (r < n) a.vec[off + (int64_t)r * n + rank] = V[r];
  if (a.pos_enc != nullptr) {
    const int c = rank - a.skip;
    if (c >= 0 && c < a.k) {
#pragma unroll
      for (int r = 0; r < NR; ++r)
        if (r < n) a.pos_enc[(int64_t)(n0 + r) * a.k + c] = V[r];
    }
  }
}
#undef EVD_ROWS

// one launch for the three size classes, largest first (their chains are the longest): block ranges from the class counts
__global__ __launch_bounds__(64) void k_evd_jacobi(EvdArgs a, int B) {
  const int c16 = a.cls_count[0], c32 = a.cls_count[1], c64 = a.cls_count[2];
  int b = blockIdx.x;
  if (b < c64) return evd_jacobi<64>(a, a.cls_list + 2 * (int64_t)B, c64, b);
  b -= c64;
  const int b32 = (c32 + 1) >> 1;
  if (b < b32) return evd_jacobi<32>(a, a.cls_list + B, c32, b);
  b -= b32;
  if (b < ((c16 + 3) >> 2)) evd_jacobi<16>(a, a.cls_list, c16, b);
}

}  // namespace
}  // namespace sn

extern "C" int64_t sn_evd_work_ints(int64_t B) { return 3 * B + 8; }

extern "C" int sn_laplacian_evd_f32(const int64_t* edge_index, int64_t E, const int32_t* graph_ptr, int64_t B, int64_t N,
                                    int norm, int64_t* evoff, float* eigen_values, float* eigen_vectors, int64_t total,
                                    float* pos_enc, int k, int skip, int32_t* work, int32_t* status, void* stream) {
  using namespace sn;
  SN_REQUIRE(graph_ptr && evoff && eigen_values && eigen_vectors && work && status, "sn_laplacian_evd_f32: null pointer");
  SN_REQUIRE(E == 0 || edge_index, "sn_laplacian_evd_f32: null edge_index with E = %lld", (long long)E);
  SN_REQUIRE(B >= 0 && N >= 0 && E >= 0 && total >= 0 && B < (1ll << 31) && N < (1ll << 31), "sn_laplacian_evd_f32: bad sizes");
  SN_REQUIRE(norm == 0 || norm == 1, "sn_laplacian_evd_f32: norm must be 0 (None: D - A) or 1 ('sym'), got %d", norm);
  SN_REQUIRE(pos_enc == nullptr || (k > 0 && skip >= 0), "sn_laplacian_evd_f32: pos_enc needs k > 0, skip >= 0");
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(status, 0, 4 * sizeof(int32_t), st);
  if (e == hipSuccess && total) e = hipMemsetAsync(eigen_vectors, 0, (size_t)total * sizeof(float), st);
  if (e == hipSuccess && pos_enc && N) e = hipMemsetAsync(pos_enc, 0, (size_t)N * k * sizeof(float), st);
  if (e != hipSuccess) return fail(SN_ERR_LAUNCH, "sn_laplacian_evd_f32: memset: %s", hipGetErrorString(e));
  if (B == 0) return SN_OK;
  int32_t* cls_list = work;
  int32_t* cls_count = work + 3 * B;
  hipLaunchKernelGGL(k_evd_prep, dim3(1), dim3(256), 0, st, graph_ptr, (int)B, total, evoff, cls_list, cls_count, status);
  SN_CHECK_LAUNCH("k_evd_prep");
  if (E) {
    hipLaunchKernelGGL(k_evd_scatter, dim3((unsigned)cdiv(E, 256)), dim3(256), 0, st, edge_index, E, N, graph_ptr, (int)B,
                       evoff, total, eigen_vectors, status);
    SN_CHECK_LAUNCH("k_evd_scatter");
  }
  EvdArgs a{graph_ptr, evoff, cls_list, cls_count, eigen_values, eigen_vectors, pos_enc, status, total, norm, k, skip};
  // a + b + c = B graphs in the three classes need at most ceil(a/4) + ceil(b/2) + c <= B + 2 blocks
  hipLaunchKernelGGL(k_evd_jacobi, dim3((unsigned)(B + 2)), dim3(64), 0, st, a, (int)B);
  SN_CHECK_LAUNCH("k_evd_jacobi");
  return SN_OK;
}
