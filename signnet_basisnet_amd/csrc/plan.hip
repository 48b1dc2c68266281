// plan.hip — per-batch structure: graph_ptr / nvalid / dst-sorted CSR, and eigen-data packing.
// Replaces the reference's to_dense_EVD bookkeeping (Alchemy/sign_net/transform.py:26-61),
// the mask construction of SignNet.forward (sign_net.py:100-102) and PyG's per-call COO gather
// (torch_geometric MessagePassing) — see include/signnet_hip.h.
#include "common.hpp"

namespace sn {

// status words: [0] error bits, [1] max nodes per graph, [2] max in-degree, [3] reserved
enum { ST_ERR = 0, ST_NMAX = 1, ST_DEGMAX = 2 };
enum { ERR_UNSORTED = 1, ERR_GRAPH_ID = 2, ERR_EDGE_RANGE = 4, ERR_EDGE_CROSS = 8 };

// K1: per node — graph id, graph boundaries; zero the in-degree counters.
__global__ void k_plan_nodes(const int64_t* __restrict__ batch, int64_t N, int64_t B,
                             int32_t* __restrict__ graph_ptr, int32_t* __restrict__ node_graph,
                             int32_t* __restrict__ deg, int32_t* __restrict__ status) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) graph_ptr[B] = (int32_t)N;
  if (i >= N) return;
  int64_t g = batch[i];
  deg[i] = 0;
  if (g < 0 || g >= B) {
    atomicOr(&status[ST_ERR], ERR_GRAPH_ID);
    node_graph[i] = 0;
    return;
  }
  node_graph[i] = (int32_t)g;
  int64_t gp = (i == 0) ? -1 : batch[i - 1];
  if (gp > g) atomicOr(&status[ST_ERR], ERR_UNSORTED);
  if (gp != g) {
    // every graph id in (gp, g] starts here (ids in between are empty graphs)
    for (int64_t t = (gp < 0 ? 0 : gp + 1); t <= g; ++t) graph_ptr[t] = (int32_t)i;
  }
  if (i == N - 1)
    for (int64_t t = g + 1; t < B; ++t) graph_ptr[t] = (int32_t)N;
}

// K2: per edge — validate, count in-degree.
__global__ void k_plan_degree(const int64_t* __restrict__ ei, int64_t E, int64_t N,
                              const int32_t* __restrict__ node_graph, int32_t* __restrict__ deg,
                              int32_t* __restrict__ status) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  int64_t s = ei[e], d = ei[E + e];
  if (s < 0 || s >= N || d < 0 || d >= N) {
    atomicOr(&status[ST_ERR], ERR_EDGE_RANGE);
    return;
  }
  if (node_graph[s] != node_graph[d]) atomicOr(&status[ST_ERR], ERR_EDGE_CROSS);
  atomicAdd(&deg[d], 1);
}

// K3: one workgroup — exclusive scans: deg -> rowptr (and cursor copy), n_b^2 -> evoff; nvalid; maxima.
__global__ __launch_bounds__(1024) void k_plan_scan(int64_t N, int64_t B, int kmax,
                                                    const int32_t* __restrict__ graph_ptr,
                                                    const int32_t* __restrict__ node_graph,
                                                    int32_t* __restrict__ deg /* in: degree, out: cursor=rowptr */,
                                                    int32_t* __restrict__ rowptr,
                                                    int32_t* __restrict__ nvalid,
                                                    int64_t* __restrict__ evoff,
                                                    int32_t* __restrict__ status) {
  __shared__ long long part[1024];
  __shared__ long long carry_s;
  __shared__ int maxs[2];
  const int T = blockDim.x, t = threadIdx.x;
  if (t == 0) { carry_s = 0; maxs[0] = 0; maxs[1] = 0; }
  __syncthreads();
  // ---- rowptr = exclusive scan of deg (chunked: each thread owns a contiguous run)
  {
    int64_t per = (N + T - 1) / T;
    int64_t lo = (int64_t)t * per, hi = lo + per < N ? lo + per : N;
    long long s = 0;
    int dmax = 0;
    for (int64_t i = lo; i < hi; ++i) { s += deg[i]; dmax = dmax > deg[i] ? dmax : deg[i]; }
    part[t] = s;
    atomicMax(&maxs[1], dmax);
    __syncthreads();
    // Hillis-Steele inclusive scan over the T partials
    for (int off = 1; off < T; off <<= 1) {
      long long v = (t >= off) ? part[t - off] : 0;
      __syncthreads();
      part[t] += v;
      __syncthreads();
    }
    long long run = part[t] - s;  // exclusive prefix of this thread's run
    for (int64_t i = lo; i < hi; ++i) {
      int dgi = deg[i];
      rowptr[i] = (int32_t)run;
      deg[i] = (int32_t)run;  // fill cursor
      run += dgi;
    }
    if (t == T - 1) rowptr[N] = (int32_t)part[T - 1];
    __syncthreads();
  }
  // ---- evoff = exclusive scan of n_b^2 over graphs
  {
    int64_t per = (B + T - 1) / T;
    int64_t lo = (int64_t)t * per, hi = lo + per < B ? lo + per : B;
    long long s = 0;
    int nmax = 0;
    for (int64_t b = lo; b < hi; ++b) {
      long long n = graph_ptr[b + 1] - graph_ptr[b];
      s += n * n;
      nmax = nmax > (int)n ? nmax : (int)n;
    }
    part[t] = s;
    atomicMax(&maxs[0], nmax);
    __syncthreads();
    for (int off = 1; off < T; off <<= 1) {
      long long v = (t >= off) ? part[t - off] : 0;
      __syncthreads();
      part[t] += v;
      __syncthreads();
    }
    long long run = part[t] - s;
    for (int64_t b = lo; b < hi; ++b) {
      long long n = graph_ptr[b + 1] - graph_ptr[b];
      evoff[b] = run;
      run += n * n;
    }
    if (t == T - 1) evoff[B] = part[T - 1];
    __syncthreads();
  }
  // ---- nvalid per node
  for (int64_t i = t; i < N; i += T) {
    int g = node_graph[i];
    int n = graph_ptr[g + 1] - graph_ptr[g];
    nvalid[i] = (kmax > 0 && n > kmax) ? kmax : n;
  }
  if (t == 0) { status[ST_NMAX] = maxs[0]; status[ST_DEGMAX] = maxs[1]; }
}

// K4: per edge — scatter into its destination's segment (arrival order is arbitrary here ...)
__global__ void k_plan_fill(const int64_t* __restrict__ ei, int64_t E, int64_t N,
                            int32_t* __restrict__ cursor, int32_t* __restrict__ col,
                            int32_t* __restrict__ eperm) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  int64_t s = ei[e], d = ei[E + e];
  if (s < 0 || s >= N || d < 0 || d >= N) return;
  int p = atomicAdd(&cursor[d], 1);
  col[p] = (int32_t)s;
  eperm[p] = (int32_t)e;
}

// K5: per node — ... so sort each segment by edge id (insertion sort; molecular degrees are <= 4).
// The in-edge order, and therefore the fp32 summation order of every aggregation, is deterministic.
__global__ void k_plan_sort(int64_t N, const int32_t* __restrict__ rowptr, int32_t* __restrict__ col,
                            int32_t* __restrict__ eperm) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  int lo = rowptr[i], hi = rowptr[i + 1];
  for (int a = lo + 1; a < hi; ++a) {
    int ke = eperm[a], kc = col[a];
    int b = a - 1;
    while (b >= lo && eperm[b] > ke) {
      eperm[b + 1] = eperm[b];
      col[b + 1] = col[b];
      --b;
    }
    eperm[b + 1] = ke;
    col[b + 1] = kc;
  }
}

// Eigen-data packing (to_dense_list_EVD, transform.py:52-61).
__global__ void k_pack_eig(const float* __restrict__ ev, const float* __restrict__ es,
                           const int32_t* __restrict__ graph_ptr, const int32_t* __restrict__ node_graph,
                           const int32_t* __restrict__ nvalid, const int64_t* __restrict__ evoff,
                           int64_t N, int K, float* __restrict__ x0, float* __restrict__ s0) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * K) return;
  int64_t i = idx / K;
  int j = (int)(idx - i * K);
  int g = node_graph[i];
  int gs = graph_ptr[g];
  int n = graph_ptr[g + 1] - gs;
  bool ok = j < nvalid[i];
  float v = 0.f, s = 0.f;
  if (ok) {
    v = ev[evoff[g] + (int64_t)(i - gs) * n + j];
    if (s0) s = es[gs + j];
  }
  x0[idx] = v;
  if (s0) s0[idx] = s;
}

}  // namespace sn

using namespace sn;

extern "C" int sn_batch_plan(const int64_t* batch, int64_t N, int64_t B, const int64_t* edge_index,
                             int64_t E, int kmax, int32_t* graph_ptr, int32_t* node_graph,
                             int32_t* nvalid, int64_t* evoff, int32_t* rowptr, int32_t* col,
                             int32_t* eperm, int32_t* status, int32_t* scratch, void* stream) {
  SN_REQUIRE(N >= 0 && B >= 0 && E >= 0, "sn_batch_plan: negative size");
  SN_REQUIRE(N < (1ll << 31) && E < (1ll << 31), "sn_batch_plan: N/E exceed int32");
  SN_REQUIRE(graph_ptr && node_graph && nvalid && evoff && rowptr && status && scratch,
             "sn_batch_plan: null output");
  SN_REQUIRE(N == 0 || batch, "sn_batch_plan: null batch");
  SN_REQUIRE(E == 0 || (edge_index && col && eperm), "sn_batch_plan: null edge arrays");
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(status, 0, 4 * sizeof(int32_t), st) != hipSuccess)
    return fail(SN_ERR_LAUNCH, "sn_batch_plan: memset failed");
  int32_t* deg = scratch;
  const int T = 256;
  hipLaunchKernelGGL(k_plan_nodes, dim3((unsigned)cdiv(N > 0 ? N : 1, T)), dim3(T), 0, st, batch, N, B,
                     graph_ptr, node_graph, deg, status);
  if (E > 0)
    hipLaunchKernelGGL(k_plan_degree, dim3((unsigned)cdiv(E, T)), dim3(T), 0, st, edge_index, E, N,
                       node_graph, deg, status);
  hipLaunchKernelGGL(k_plan_scan, dim3(1), dim3(1024), 0, st, N, B, kmax, graph_ptr, node_graph, deg,
                     rowptr, nvalid, evoff, status);
  if (E > 0) {
    hipLaunchKernelGGL(k_plan_fill, dim3((unsigned)cdiv(E, T)), dim3(T), 0, st, edge_index, E, N, deg,
                       col, eperm);
    hipLaunchKernelGGL(k_plan_sort, dim3((unsigned)cdiv(N, T)), dim3(T), 0, st, N, rowptr, col, eperm);
  }
  SN_CHECK_LAUNCH("sn_batch_plan");
  return SN_OK;
}

extern "C" int sn_pack_eig_f32(const float* eigen_vectors, const float* eigen_values,
                               const int32_t* graph_ptr, const int32_t* node_graph,
                               const int32_t* nvalid, const int64_t* evoff, int64_t N, int K, float* x0,
                               float* s0, void* stream) {
  SN_REQUIRE(K > 0 && N >= 0, "sn_pack_eig_f32: bad sizes");
  SN_REQUIRE(x0 && graph_ptr && node_graph && nvalid && evoff, "sn_pack_eig_f32: null pointer");
  SN_REQUIRE(!s0 || eigen_values, "sn_pack_eig_f32: s0 requested without eigen_values");
  if (N == 0) return SN_OK;
  hipLaunchKernelGGL(k_pack_eig, dim3((unsigned)cdiv(N * K, 256)), dim3(256), 0, (hipStream_t)stream,
                     eigen_vectors, eigen_values, graph_ptr, node_graph, nvalid, evoff, N, K, x0, s0);
  SN_CHECK_LAUNCH("sn_pack_eig_f32");
  return SN_OK;
}
