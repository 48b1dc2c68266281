// plan.hip — per-batch structure: graph_ptr / nvalid / dst-sorted CSR, the work bins of the fused stages, and
// eigen-data packing.  Replaces the reference's to_dense_EVD bookkeeping (Alchemy/sign_net/transform.py:26-61),
// the mask construction of SignNet.forward (sign_net.py:100-102) and PyG's per-call COO gather
// (torch_geometric MessagePassing) — see include/signnet_hip.h.  Everything runs on the device; the host never
// learns a graph size.
#include "common.hpp"

namespace sn {

// status words: [0] error bits, [1] max nodes per graph, [2] max in-degree, [3] set by the fused kernels
enum { ST_ERR = 0, ST_NMAX = 1, ST_DEGMAX = 2 };
enum { ERR_UNSORTED = 1, ERR_GRAPH_ID = 2, ERR_EDGE_RANGE = 4, ERR_EDGE_CROSS = 8 };

constexpr int PLAN_T = 1024;          // threads of the single-workgroup stages
constexpr int BINS_BMAX = 6144;       // graphs the bin planner handles (LDS-resident working set)

struct BinsDev {  // device view of sn_plan_bins
  int32_t* phi_bin_col;
  long long phi_max_bins;
  int32_t* phi_col_bin0;
  int32_t* phi_col_mem;
  int32_t* phi_col_off;
  int32_t* rho_bin0;
  int32_t* meta;
  int32_t* phi_bin_mem;   // [phi_max_bins][16] member records of every bin (see sn_plan_bins), or null
};

// Early report of the batch's flags to pinned host memory (sn_plan_early): every workgroup of the one-launch plan writes the words it
// owns and then its own "done" word — no counter to zero, no extra launch.
struct EarlyDev {
  const int64_t* node_ids;
  long long n_node_ids, node_vocab;
  const int64_t* edge_ids;
  long long n_edge_ids, edge_vocab;
  int max_graph_edges;
  int32_t* host;          // null: no report
};
enum { EH_ERR = 0, EH_NMAX = 1, EH_DEGMAX = 2, EH_EDGES = 3, EH_PHI = 4, EH_RHO = 5, EH_IDS = 6, EH_DONE = 8 };
__device__ __forceinline__ void early_put(int32_t* host, int i, int v) {
  __hip_atomic_store(&host[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// (thread 0 of a block, after its early_put()s)
__device__ __forceinline__ void early_done(int32_t* host, int block) {
  __threadfence_system();
  __hip_atomic_store(&host[EH_DONE + block], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  __threadfence_system();
}

// exclusive scan of one int per thread over a PLAN_T-thread workgroup; wsum: LDS int[32]; returns the exclusive
// prefix, *total gets the grand total (valid for every thread).
__device__ __forceinline__ int block_exscan(int v, int* wsum, int t, int* total) {
  const int lane = t & 63, w = t >> 6;
  int x = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int y = __shfl_up(x, off, 64);
    if (lane >= off) x += y;
  }
  if (lane == 63) wsum[w] = x;
  __syncthreads();
  if (w == 0) {
    int s = (lane < PLAN_T / 64) ? wsum[lane] : 0;
    int inc = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int y = __shfl_up(inc, off, 64);
      if (lane >= off) inc += y;
    }
    if (lane < PLAN_T / 64) wsum[lane] = inc - s;
    if (lane == PLAN_T / 64 - 1) wsum[31] = inc;
  }
  __syncthreads();
  const int r = x - v + wsum[w];
  *total = wsum[31];
  __syncthreads();
  return r;
}

// the same over 64-bit values (several packed counters in one pass); wsum64: LDS long long[32]
__device__ __forceinline__ long long block_exscan64(long long v, long long* wsum64, int t, long long* total) {
  const int lane = t & 63, w = t >> 6;
  long long x = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const long long y = __shfl_up(x, off, 64);
    if (lane >= off) x += y;
  }
  if (lane == 63) wsum64[w] = x;
  __syncthreads();
  if (w == 0) {
    long long s = (lane < PLAN_T / 64) ? wsum64[lane] : 0;
    long long inc = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const long long y = __shfl_up(inc, off, 64);
      if (lane >= off) inc += y;
    }
    if (lane < PLAN_T / 64) wsum64[lane] = inc - s;
    if (lane == PLAN_T / 64 - 1) wsum64[31] = inc;
  }
  __syncthreads();
  const long long r = x - v + wsum64[w];
  *total = wsum64[31];
  return r;
}

// kmax > 0: at most kmax eigenvector slots per node; 0: all n of them; kmax < 0 ("full slots", the DGL tree's dense [N, K] positional
// encodings): the valid-slot count is still min(n, |kmax|), but the phi work bins cover all |kmax| slots of every graph (zero-padded
// eigenvectors go through GINDeepSigns like any other column, deepsigns.py:45-51).
__device__ __forceinline__ int slots_of(int n, int kmax) { const int k = kmax < 0 ? -kmax : kmax; return (k > 0 && n > k) ? k : n; }
__device__ __forceinline__ int phi_slots_of(int n, int kmax) { return kmax < 0 ? -kmax : slots_of(n, kmax); }

// ---------------------------------------------------------------------------- work bins (one workgroup)
// phi: a unit is one (graph, slot) slab of n rows; graphs are packed into COLUMNS by best-fit-decreasing on n
//      (capacity 64 rows, at most 8 graphs): bin j of a column holds slot j of every member graph, so a column
//      of height max_g K_g covers all slabs of its graphs and every bin is (nearly) full.
// rho: a unit is one node's K_g slot rows, padded to p = 16*ceil(K_g/16) so that a unit never straddles a 16-row
//      tile; 64/p units per bin, bins never mix graphs -> closed form, no sequential pass.
// gp: graph_ptr in LDS ([B+1]); lds: int scratch [5*B + 3*66 + 32 + 4].
#ifdef SN_PROFILE
static __device__ long long g_pprof[64];
#define PL_STAMP(i) do { if (threadIdx.x == 0) g_pprof[(i) + 16 * blockIdx.x] = clock64(); } while (0)
#define PL_STAMP_T(i, thr) do { if (threadIdx.x == (thr)) g_pprof[(i) + 16 * blockIdx.x] = clock64(); } while (0)
#else
#define PL_STAMP(i) do { } while (0)
#define PL_STAMP_T(i, thr) do { } while (0)
#endif

// rho bins (closed form per graph + a prefix): independent of the phi columns — its own workgroup in the single-launch plan
__device__ void plan_rho_block(const int* gp, int B, int kmax, BinsDev bd, int* lds, int32_t* early_host = nullptr) {
  const int t = threadIdx.x;
  int* nbv = lds;               // [B]   rho bins per graph
  int* wsum = lds + B;          // [32]
  __shared__ int s_rerr;
  if (t == 0) s_rerr = 0;
  __syncthreads();
  int rows = 0, prows = 0;      // rho rows (meta[6]); phi rows (meta[2]: graphs of <= 64 nodes x their phi slots — the column planner's
                                //  block leaves this total to this one, which is off the critical path)
  const int per = (B + PLAN_T - 1) / PLAN_T;
  const int lo = t * per, hi = (lo + per < B) ? lo + per : B;
  int mine = 0;
  for (int g = lo; g < hi; ++g) {
    const int n = gp[g + 1] - gp[g];
    const int K = slots_of(n, kmax);
    int nb = 0;
    if (n > 0 && n <= 64) prows += n * phi_slots_of(n, kmax);
    if (n > 0) {
      if (K > 64) atomicOr(&s_rerr, 2);
      else {
        const int p = ((K + 15) >> 4) << 4;
        const int upb = 64 / p;                 // 4, 2, 1, 1
        nb = (n + upb - 1) / upb;
        rows += n * K;
      }
    }
    nbv[g] = nb;
    mine += nb;
  }
  int total;
  int run = block_exscan(mine, wsum, t, &total);
  for (int g = lo; g < hi; ++g) { bd.rho_bin0[g] = run; run += nbv[g]; }
  int rtot, ptot;
  block_exscan(rows, wsum, t, &rtot);
  block_exscan(prows, wsum, t, &ptot);
  __syncthreads();
  if (t == 0) {
    bd.rho_bin0[B] = total;
    bd.meta[2] = ptot;
    bd.meta[4] = total;
    bd.meta[5] = (s_rerr & 2);
    bd.meta[6] = rtot;
    if (early_host != nullptr) early_put(early_host, EH_RHO, s_rerr & 2);
  }
}

// vec with lane `lane` replaced by the wave-uniform `val` (v_writelane_b32: value and lane select are scalar operands)
__device__ __forceinline__ int writelane(int vec, int val, int lane) {
  // (gfx9 encodes one scalar register per VALU instruction: the lane select goes through M0)
  asm("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(vec) : "s"(val), "s"(lane) : "m0");
  return vec;
}

// Slab-level packing of the all-eigenvector mode (kmax == 0; `pat` != null).  There a graph of n nodes is n slabs of n rows (phi: one
// per eigenvector; rho: one per node, n slot rows each), and column packing leaves the bins above a column's shorter members part
// empty (bench batch, n uniform in 9..37: 1 302 bins at 92 % fill, six rounds of 256 workgroups).  Packing the SLABS — any slabs
// of any graphs, best-fit-decreasing on the size classes, <= 8 per bin — reaches 98-99 % (1 208 bins, five rounds).  The chain
// walks bin PATTERNS, not slabs: with c_s = s * (graphs of size s) slabs per class the greedy choice repeats until a class of the
// pattern runs out, so one link covers r = min_s floor(left_s / copies_s) identical bins and every link retires a class (or leaves
// it fewer slabs than the pattern took): a few dozen links on a ZINC batch, at most PAT_MAX (from PAT_MAX - 64 links on the
// remaining classes are emitted one slab per bin, which needs at most 64 more).  A pattern is expanded into its bins' member
// records in parallel afterwards: slab q of class s is (graph = q / s-th of the class in id order, index = q % s).
constexpr int PAT_MAX = 160;
constexpr int PAT_INTS = (PAT_MAX + 1) + PAT_MAX + 2 * 8 * PAT_MAX;
constexpr int PLAN_SLAB_BMAX = 4096;   // five-launch plan: slab-level packing up to this many graphs (the pattern table must fit the LDS beside 6 B ints)   // first bin [PAT_MAX+1] | members [PAT_MAX] | member words | slab bases

__device__ void plan_bins_block(const int* gp, int B, int kmax, BinsDev bd, int* lds, int* pat, int32_t* early_host = nullptr) {
  const int t = threadIdx.x;
  const bool slab = pat != nullptr && kmax == 0 && bd.phi_bin_mem != nullptr;
  int* pat_first = pat;                          // [PAT_MAX + 1]
  int* pat_nm = pat + (PAT_MAX + 1);             // [PAT_MAX]
  int* pat_mem = pat_nm + PAT_MAX;               // [PAT_MAX][8]  runs: class | row offset << 7 | copies << 13
  int* pat_base = pat_mem + 8 * PAT_MAX;         // [PAT_MAX][8]  first slab of the run's class this pattern takes
  __shared__ int s_npat, s_slab_bins;
  int* bucket = lds;            // [B]   graph ids grouped by size, ascending id inside a group
  int* nbv = lds + B;           // [B]   (record scratch of the column packing)
  int* hist = lds + 2 * B;      // [66]
  int* bstart = hist + 66;      // [66]
  int* wsum = bstart + 2 * 66;  // [32]  (66 ints behind bstart are spare)
  __shared__ int s_err, s_nrec;
  __shared__ long long wsum64[32];
  if (t < 66) hist[t] = 0;
  if (t == 0) { s_err = 0; s_nrec = 0; }
  __syncthreads();
  PL_STAMP(3);
  // ---- phi: group graphs by size
  for (int g = t; g < B; g += PLAN_T) {
    const int n = gp[g + 1] - gp[g];
    if (n > 64) atomicOr(&s_err, 1);
    else if (n > 0) atomicAdd(&hist[n], 1);
    else atomicOr(&s_err, 8);                      // (a graph without nodes: no bins; told to the early report only)
  }
  __syncthreads();
  if (t < 64) {   // bstart = exclusive prefix of hist[0..65] (wave scan; hist[0] = 0: empty graphs are not binned)
    const int h = hist[t];
    int x = h;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int y = __shfl_up(x, off, 64);
      if (t >= off) x += y;
    }
    bstart[t] = x - h;
    if (t == 63) { bstart[64] = x; bstart[65] = x + hist[64]; }
  }
  __syncthreads();
  // ---- phi: best-fit-decreasing over the size classes.  Wave 0 walks the items with its state in registers: lane s-1 holds the
  //      remaining count of size class s; the loop-carried chain is scalar arithmetic, one readlane and two writelanes per item, and
  //      all it emits per placed graph is ONE word (class | first-of-column flag | graphs of the class still unplaced).  Everything
  //      else — rank inside the class, column, member index, row offset, the columns' first bins — follows from that sequence by
  //      workgroup-wide scans afterwards (the chain used to keep those records itself: ~335 cycles per graph, the longest phase of
  //      the plan).
  int* rec = nbv;                              // [B] reuse (rho is done with nbv): the chain's word of the r-th placed graph
  int* rec_pre = lds + 2 * B + 3 * 66 + 32;    // [B] exclusive prefix of the classes (rows) over the records   (extra [3*B] ints)
  int* col_start = rec_pre + B;                // [B] record index of every column's first member
  int* col_bin = col_start + B;                // [B + 1] first bin of every column
  PL_STAMP(4);
  const int rank_t0 = 128;                         // first thread of the id-ranking waves (wave 0: the column chain, wave 1: the slab chain)
  if (t >= 64 && t < 128) {
    if (slab) {
    // ---- wave 1, all-eigenvector mode: the slab-level pattern chain (see above).  State in registers: lane s-1 holds class s's
    //      unplaced / placed slab counts; the members of the pattern being built sit in lanes 0..7 of two registers and go to LDS
    //      with one store per pattern; the run length is a float-reciprocal quotient (corrected to the exact floor) reduced over
    //      the pattern's classes by a ballot walk (<= 8 set bits) — no LDS round trip inside a link.
    const int lane = t - 64;
    int cnt = hist[lane + 1] * (lane + 1);          // slabs of class s = lane + 1 not yet placed
    int used = 0;                                   // ... and placed
    unsigned long long avail = __ballot(cnt > 0);
    int np = 0, nb = 0;
    while (avail) {
      int mult = 0;                                 // copies of my class in this pattern
      int pm = -1, pb = 0;                          // lanes 0..7: the pattern's runs (class | row offset << 7 | copies << 13) / their first slabs
      int cap = 64, nm = 0, nrun = 0;
      int r = 0x7fffffff;                           // run length so far: min over the runs with ONE copy of their class (no division)
      bool multi = false;                           // a run with several copies of its class: its quotient is taken below
      const bool single = np >= PAT_MAX - 64;
      unsigned long long m = avail;
      do {
        const int cls = 64 - __clzll(m);            // the largest class with slabs left that fits the rows left
        const int left = __builtin_amdgcn_readlane(cnt, cls - 1);
        int lim = left < 8 - nm ? left : 8 - nm;    // as many copies of it as fit (a run)
        if (single) lim = 1;
        int j = 1, capj = cap - cls;
        while (j < lim && capj >= cls) { ++j; capj -= cls; }
        pm = writelane(pm, cls | ((64 - cap) << 7) | (j << 13), nrun);
        pb = writelane(pb, __builtin_amdgcn_readlane(used, cls - 1), nrun);
        mult = writelane(mult, j, cls - 1);
        if (j == 1) r = left < r ? left : r; else multi = true;
        cap = capj;
        nm += j;
        ++nrun;
        const int limit = cap < cls - 1 ? cap : cls - 1;       // next: the largest class below this one that still fits
        m = (limit > 0 && nm < 8 && !single) ? (avail & ((1ull << limit) - 1ull)) : 0ull;
      } while (m);
      if (multi) {
        // floor(cnt / mult) of the classes taken several times   (cnt < 2^24: the reciprocal quotient is within one of it)
        int q = 0x7fffffff;
        if (mult > 1) {
          q = (int)((float)cnt * __builtin_amdgcn_rcpf((float)mult));
          q += ((q + 1) * mult <= cnt) ? 1 : 0;
          q -= (q * mult > cnt) ? 1 : 0;
        }
        for (unsigned long long pmask = __ballot(mult > 1); pmask; pmask &= pmask - 1) {
          const int qc = __builtin_amdgcn_readlane(q, __builtin_ctzll(pmask));
          r = qc < r ? qc : r;
        }
      }                                            // r >= 1: every class of the pattern had its copies left
      cnt -= r * mult;
      used += r * mult;
      if (lane < 8) { pat_mem[np * 8 + lane] = pm; pat_base[np * 8 + lane] = pb; }
      if (lane == 0) { pat_first[np] = nb; pat_nm[np] = nrun; }
      nb += r;
      ++np;
      avail = __ballot(cnt > 0);
    }
    if (lane == 0) { pat_first[np] = nb; s_npat = np; s_slab_bins = nb; }
    PL_STAMP_T(8, 64);
    }
  } else if (t >= rank_t0) {
    // ---- meanwhile, on the other waves: the graphs of every size class in ascending id order (`bucket`; the packer only needs the
    //      class counts, the ids are resolved after it).  Rank inside the class = number of earlier graphs of the same size (LDS
    //      broadcast reads): deterministic without a sort, and off the critical path whatever it costs.
    for (int g = t - rank_t0; g < B; g += PLAN_T - rank_t0) {
      const int n = gp[g + 1] - gp[g];
      if (n > 0 && n <= 64) {
        int rank = 0, prev = gp[0];
        for (int q = 0; q < g; ++q) {
          const int nx = gp[q + 1];
          rank += (nx - prev == n) ? 1 : 0;
          prev = nx;
        }
        bucket[bstart[n] + rank] = g;
      }
    }
  } else if (slab) {
    if (t == 0) s_nrec = 0;                         // (slab mode lays out no columns: the stage kernels walk the member records only)
  } else {
    const int lane = t;
    int cnt = hist[lane + 1];                       // class s = lane + 1
    unsigned long long avail = __ballot(cnt > 0);
    // (round 5, end: a link's record goes straight to LDS — a store the chain never waits for — instead of into a lane of a register
    //  that is flushed every 64 links (mask, M0, writelane, compare, branch: 7 instructions of the ~30 per link); the availability
    //  update is a branch that is taken once per class, not a select chain on every link)
    int nrec = 0;
    while (avail) {
      int cls = 64 - __clzll(avail);
      int cap = 64, members = 0, first = 1 << 7;
      while (true) {
        const int left = __builtin_amdgcn_readlane(cnt, cls - 1) - 1;
        cnt = writelane(cnt, left, cls - 1);
        if (__builtin_expect(left == 0, 0)) {
          asm volatile("" : "+s"(avail));                                   // (keeps this a branch)
          avail &= ~(1ull << (cls - 1));
        }
        rec[nrec] = cls | first | (left << 8);
        ++nrec;
        first = 0;
        ++members;
        cap -= cls;
        if (members >= 8 || cap <= 0) break;
        const unsigned long long m = avail & ((1ull << cap) - 1ull);      // 0 < cap < 64
        if (!m) break;
        cls = 64 - __clzll(m);
      }
    }
    if (lane == 0) s_nrec = nrec;
    PL_STAMP(9);
  }
  __syncthreads();
  PL_STAMP(5);
  {
    // records -> (column, member, row offset, graph id); columns -> first bins.  ONE scan of three packed counters per PLAN_T records
    // (rows placed: bits 0-23, columns opened: bits 24-39, bins of the opened columns: bits 40-63), running carries between rounds.
    // Results stay in LDS until every barrier is behind us: a workgroup barrier also drains the outstanding global stores.
    const int nrec = s_nrec;
    long long carry = 0;
    for (int r0 = 0; r0 < nrec; r0 += PLAN_T) {
      const int r = r0 + t;
      const int w = r < nrec ? rec[r] : 0;
      const int cls = w & 127, first = (w >> 7) & 1;
      const int h = first ? phi_slots_of(cls, kmax) : 0;
      long long tot;
      const long long ex = block_exscan64((long long)cls | ((long long)first << 24) | ((long long)h << 40), wsum64, t, &tot) + carry;
      if (r < nrec) {
        rec_pre[r] = (int)(ex & 0xffffff);
        if (first) {
          const int c = (int)((ex >> 24) & 0xffff);
          col_start[c] = r;
          col_bin[c] = (int)(ex >> 40);
        }
      }
      carry += tot;
    }
    const int ncol = (int)((carry >> 24) & 0xffff), nbins = (int)(carry >> 40);
    if (t == 0) col_bin[ncol] = nbins;
    __syncthreads();
    // ---- write-out (no barrier from here on)
    const int rbins = slab ? s_slab_bins : nbins;        // bins of the member records (what the stage kernels walk)
    const bool over = nbins > bd.phi_max_bins || rbins > bd.phi_max_bins;
    if (t == 0) {
      bd.meta[0] = rbins;
      bd.meta[1] = (s_err & 1) | (over ? 4 : 0);
      bd.meta[3] = ncol;
      bd.meta[7] = bd.phi_bin_mem != nullptr ? rbins : 0;
      if (early_host != nullptr) { early_put(early_host, EH_PHI, (s_err & 9) | (over ? 4 : 0)); early_done(early_host, 1); }
    }
    PL_STAMP(6);
    // ---- member records of every bin: word pair (graph | index << 13 | row offset << 19 | (rows - 1) << 25, first node of the graph),
    //      -1 = no member.  index = the eigenvector slot (phi) / the node (rho, all-eigenvector mode) of the slab.
    if (bd.phi_bin_mem != nullptr && !over) {
      if (slab) {
        const int npat = s_npat;
        for (int b = t; b < rbins; b += PLAN_T) {
          int lo = 0, hi = npat;                           // largest p with pat_first[p] <= b
          while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (pat_first[mid] <= b) lo = mid; else hi = mid;
          }
          const int rep = b - pat_first[lo], nrun = pat_nm[lo];
          int w[16];
#pragma unroll
          for (int k = 0; k < 16; ++k) w[k] = (k & 1) ? 0 : -1;
          int mi = 0;                                      // member of the bin (<= 8 over all runs)
          for (int k = 0; k < nrun; ++k) {
            const int pw = pat_mem[lo * 8 + k];
            const int cls = pw & 127, off = (pw >> 7) & 63, j = (pw >> 13) & 15;
            const int q0 = pat_base[lo * 8 + k] + rep * j;
            for (int c = 0; c < j; ++c, ++mi) {
              const int q = q0 + c;
              const int rank = q / cls, idx = q - rank * cls;
              const int g = bucket[bstart[cls] + rank];
              const int w0 = g | (idx << 13) | ((off + c * cls) << 19) | ((cls - 1) << 25), w1 = gp[g];
#pragma unroll
              for (int u = 0; u < 8; ++u)
                if (u == mi) { w[2 * u] = w0; w[2 * u + 1] = w1; }
            }
          }
          int4* dst = reinterpret_cast<int4*>(bd.phi_bin_mem + (size_t)b * 16);
#pragma unroll
          for (int k = 0; k < 4; ++k) dst[k] = make_int4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
        }
      } else {
        for (int b = t; b < nbins; b += PLAN_T) {
          int lo = 0, hi = ncol;                           // largest c with col_bin[c] <= b
          while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (col_bin[mid] <= b) lo = mid; else hi = mid;
          }
          const int slot = b - col_bin[lo];
          const int cs = col_start[lo], ce = (lo + 1 < ncol) ? col_start[lo + 1] : nrec;
          int w[16];
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            w[2 * k] = -1; w[2 * k + 1] = 0;
            if (cs + k < ce) {
              const int rw = rec[cs + k];
              const int cls = rw & 127, left = rw >> 8;
              if (slot < phi_slots_of(cls, kmax)) {
                const int g = bucket[bstart[cls] + hist[cls] - 1 - left];
                w[2 * k] = g | (slot << 13) | ((rec_pre[cs + k] - rec_pre[cs]) << 19) | ((cls - 1) << 25);
                w[2 * k + 1] = gp[g];
              }
            }
          }
          int4* dst = reinterpret_cast<int4*>(bd.phi_bin_mem + (size_t)b * 16);
#pragma unroll
          for (int k = 0; k < 4; ++k) dst[k] = make_int4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
        }
      }
    }
    PL_STAMP(7);
    if (bd.phi_bin_col == nullptr) return;            // the caller wants the member records only (what the stage kernels walk)
    for (int c = t; c <= ncol; c += PLAN_T) bd.phi_col_bin0[c] = col_bin[c];
    // member records: the column of record r is the number of column starts <= r, found by bisection over col_start; the last member
    // of a column also closes its unused slots
    for (int r = t; r < nrec; r += PLAN_T) {
      const int w = rec[r];
      const int cls = w & 127, left = w >> 8;
      int lo = 0, hi = ncol;                         // largest c with col_start[c] <= r
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (col_start[mid] <= r) lo = mid; else hi = mid;
      }
      const int cs = col_start[lo];
      const int rank = hist[cls] - 1 - left;          // graphs of the class are taken in bucket order
      const int g = bucket[bstart[cls] + rank];
      const int mem = r - cs;
      const int slot = lo * 8 + mem;
      bd.phi_col_mem[slot] = g;
      bd.phi_col_off[slot] = rec_pre[r] - rec_pre[cs];
      if (r + 1 == nrec || ((rec[r + 1] >> 7) & 1)) {
        for (int m = mem + 1; m < 8; ++m) { bd.phi_col_mem[lo * 8 + m] = -1; bd.phi_col_off[lo * 8 + m] = 0; }
      }
    }
    if (!over) {
      for (int c = t; c < ncol; c += PLAN_T) {
        const int lo = col_bin[c], hi = col_bin[c + 1];
        for (int j = lo; j < hi; ++j) bd.phi_bin_col[j] = c;
      }
    }
  }
}

// graph_ptr from a sorted batch vector, into LDS: gp[k] = first node of graph k (empty graphs included), gp[B] = N.
__device__ void lds_graph_ptr(const int64_t* __restrict__ batch, int N, int B, int* gp, int* err, int* ng = nullptr) {
  const int t = threadIdx.x;
  for (int i = t; i <= B; i += PLAN_T) gp[i] = N;
  __syncthreads();
  for (int i = t; i < N; i += PLAN_T) {
    const long long g = batch[i];
    const long long gprev = (i == 0) ? -1 : batch[i - 1];
    if (ng) ng[i] = (g < 0 || g >= B) ? -1 : (int)g;
    if (g < 0 || g >= B) { if (err) atomicOr(err, ERR_GRAPH_ID); continue; }
    if (gprev > g && err) atomicOr(err, ERR_UNSORTED);
    if (gprev < g) {
      const long long lo = gprev < -1 ? 0 : gprev + 1;
      for (long long k = lo; k <= g; ++k) gp[k] = i;
    }
  }
  __syncthreads();
}

// ============================================================================ small batches: ONE launch, two workgroups
// Block 0: nodes, CSR (LDS atomics + per-segment sort), scans, write-out.  Block 1: the work bins.
// Limits: N <= 4096, E <= 12288, B <= 1024 (the reference's batches: 128-256 molecules).
constexpr int PS_NMAX = 4096, PS_EMAX = 12288, PS_BMAX = 1024;
constexpr int PS_BINS_INTS = 5 * PS_BMAX + 3 * 66 + 32 + 8;     // plan_bins_block's scratch; the pattern table follows it



__global__ __launch_bounds__(PLAN_T) void k_plan_small(const int64_t* __restrict__ batch, int N, int B,
                                                       const int64_t* __restrict__ ei, int E, int kmax,
                                                       int32_t* __restrict__ graph_ptr, int32_t* __restrict__ node_graph,
                                                       int32_t* __restrict__ nvalid, int64_t* __restrict__ evoff,
                                                       int32_t* __restrict__ rowptr, int32_t* __restrict__ col,
                                                       int32_t* __restrict__ eperm, int32_t* __restrict__ status, BinsDev bd,
                                                       int do_bins, EarlyDev early) {
  extern __shared__ int sm[];
  const int t = threadIdx.x;
  PL_STAMP(0);
  if (blockIdx.x == 1) {
    if (!do_bins) return;
    int* gp = sm;                      // [B+1]
    lds_graph_ptr(batch, N, B, gp, nullptr);
    PL_STAMP(1);
    plan_bins_block(gp, B, kmax, bd, sm + (PS_BMAX + 4), sm + (PS_BMAX + 4) + PS_BINS_INTS, early.host);
    PL_STAMP(2);
    return;
  }
  if (blockIdx.x == 3) {
    // fourth workgroup (sn_plan_early with feature ids): every discrete feature id against the rows of its embedding tables — what
    // nn.Embedding would raise IndexError for (model_utils/elements.py:21-37); the fused GINE stage never dereferences such an id,
    // this is the same verdict a forward earlier
    int bad = 0;
    for (long long i = t; i < early.n_node_ids; i += PLAN_T) bad |= ((unsigned long long)early.node_ids[i] >= (unsigned long long)early.node_vocab) ? 1 : 0;
    for (long long i = t; i < early.n_edge_ids; i += PLAN_T) bad |= ((unsigned long long)early.edge_ids[i] >= (unsigned long long)early.edge_vocab) ? 1 : 0;
    bad = __syncthreads_or(bad);
    if (t == 0) { early_put(early.host, EH_IDS, bad ? 1 : 0); early_done(early.host, 3); }
    return;
  }
  if (blockIdx.x == 2) {               // rho bins: third workgroup (off the critical path of the column packing)
    int* gp = sm;
    lds_graph_ptr(batch, N, B, gp, nullptr);
    plan_rho_block(gp, B, kmax, bd, sm + (PS_BMAX + 4), early.host);
    // evoff = exclusive scan of n^2 and the largest graph: nothing of the CSR block depends on them, so they are taken here
    {
      int* wsum2 = sm + (PS_BMAX + 4) + B;     // plan_rho_block's scan scratch
      __shared__ int s_nmax2;
      if (t == 0) s_nmax2 = 0;
      __syncthreads();
      const int perb = (B + PLAN_T - 1) / PLAN_T;
      const int blo = t * perb, bhi = (blo + perb < B) ? blo + perb : B;
      int q = 0, nmax = 0;
      for (int g = blo; g < bhi; ++g) { const int n = gp[g + 1] - gp[g]; q += n * n; nmax = nmax > n ? nmax : n; }
      for (int off = 32; off > 0; off >>= 1) nmax = max(nmax, __shfl_xor(nmax, off, 64));
      if ((t & 63) == 0) atomicMax(&s_nmax2, nmax);
      int qtot;
      int qrun = block_exscan(q, wsum2, t, &qtot);
      for (int g = blo; g < bhi; ++g) { const int n = gp[g + 1] - gp[g]; evoff[g] = qrun; qrun += n * n; }
      if (t == 0) {
        evoff[B] = qtot; status[ST_NMAX] = s_nmax2;
        if (early.host != nullptr) { early_put(early.host, EH_NMAX, s_nmax2); early_done(early.host, 2); }
      }
    }
    return;
  }
  int* gp = sm;                        // [B+1]
  int* deg = gp + (PS_BMAX + 4);       // [N]  in-degree
  int* rp = deg + PS_NMAX;             // [N+1]
  int* lfill = rp + (PS_NMAX + 4);     // [E]  one packed key per in-edge in arrival order: edge id << 12 | source
  int* lperm = lfill + PS_EMAX;        // [E]  ... sorted by edge id inside every node's segment
  int* wsum = lperm + PS_EMAX;         // [32]
  int* ng = wsum + 32;                 // [N]  graph id of every node (edge validation without global gathers)
  __shared__ int s_err, s_nmax, s_dmax, s_big;
  if (t == 0) { s_err = 0; s_nmax = 0; s_dmax = 0; s_big = 0; }
  // Round 5: every global read of this workgroup is requested up front (the batch vector and both rows of edge_index: <= 4 nodes and
  // <= 12 edges per thread, coalesced) — until round 4 the edge list was read twice (degree pass, fill pass) behind the batch vector,
  // three exposed memory round trips of ~3 k cycles each — and every in-edge keeps its arrival position from the degree pass's LDS
  // atomic, so the fill is a plain store and the per-segment sort a rank count (reads of a segment pipeline; the per-node sorting
  // networks walked three nodes per thread with dependent LDS round trips: 12 k cycles).
  constexpr int NPT = PS_NMAX / PLAN_T, EPT = PS_EMAX / PLAN_T;
  long long bcur[NPT], bprev[NPT];
#pragma unroll
  for (int k = 0; k < NPT; ++k) {
    const int i = t + k * PLAN_T;
    bcur[k] = i < N ? batch[i] : 0;
    bprev[k] = (i < N && i > 0) ? batch[i - 1] : -1;
  }
  int es[EPT], ed[EPT];                // source / target of my edges (-1: out of range)
  {
    long long sv[EPT], dv[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int e = t + k * PLAN_T;
      sv[k] = e < E ? ei[e] : 0;
      dv[k] = e < E ? ei[(long long)E + e] : 0;
    }
    for (int i = t; i < N; i += PLAN_T) deg[i] = 0;
    for (int i = t; i <= B; i += PLAN_T) gp[i] = N;
    __syncthreads();
    // graph_ptr (first node of every graph, empty graphs included) and the nodes' graph ids
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
      const int i = t + k * PLAN_T;
      if (i < N) {
        const long long g = bcur[k], gprev = bprev[k];
        const bool ok = g >= 0 && g < B;
        ng[i] = ok ? (int)g : -1;
        if (!ok) atomicOr(&s_err, ERR_GRAPH_ID);
        else {
          if (gprev > g) atomicOr(&s_err, ERR_UNSORTED);
          if (gprev < g) {
            const long long lo = gprev < -1 ? 0 : gprev + 1;
            for (long long q = lo; q <= g; ++q) gp[q] = i;
          }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int e = t + k * PLAN_T;
      const bool bad = e < E && (sv[k] < 0 || sv[k] >= N || dv[k] < 0 || dv[k] >= N);
      if (bad) atomicOr(&s_err, ERR_EDGE_RANGE);
      es[k] = (e < E && !bad) ? (int)sv[k] : -1;
      ed[k] = (e < E && !bad) ? (int)dv[k] : -1;
    }
  }
  __syncthreads();
  PL_STAMP(1);
  // ---- in-degrees (LDS atomics: the value returned is the edge's arrival position in its segment) + edge validation
  int epos[EPT];
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    epos[k] = 0;
    if (ed[k] >= 0) {
      if (ng[es[k]] != ng[ed[k]]) atomicOr(&s_err, ERR_EDGE_CROSS);
      epos[k] = atomicAdd(&deg[ed[k]], 1);
    }
  }
  __syncthreads();
  PL_STAMP(2);
  // ---- rowptr = exclusive scan of deg ; evoff = exclusive scan of n^2
  {
    const int per = (N + PLAN_T - 1) / PLAN_T;
    const int lo = t * per, hi = (lo + per < N) ? lo + per : N;
    int s = 0, dmax = 0;
    for (int i = lo; i < hi; ++i) { s += deg[i]; dmax = dmax > deg[i] ? dmax : deg[i]; }
    for (int off = 32; off > 0; off >>= 1) dmax = max(dmax, __shfl_xor(dmax, off, 64));
    if ((t & 63) == 0) atomicMax(&s_dmax, dmax);
    int total;
    int run = block_exscan(s, wsum, t, &total);
    for (int i = lo; i < hi; ++i) { rp[i] = run; run += deg[i]; }
    if (t == 0) rp[N] = total;
    if (!do_bins) {     // (with the bin planner the third workgroup takes evoff and the largest graph)
      const int perb = (B + PLAN_T - 1) / PLAN_T;
      const int blo = t * perb, bhi = (blo + perb < B) ? blo + perb : B;
      int q = 0, nmax = 0;
      for (int g = blo; g < bhi; ++g) { const int n = gp[g + 1] - gp[g]; q += n * n; nmax = nmax > n ? nmax : n; }
      for (int off = 32; off > 0; off >>= 1) nmax = max(nmax, __shfl_xor(nmax, off, 64));
      if ((t & 63) == 0) atomicMax(&s_nmax, nmax);
      int qtot;
      int qrun = block_exscan(q, wsum, t, &qtot);
      for (int g = blo; g < bhi; ++g) { const int n = gp[g + 1] - gp[g]; evoff[g] = qrun; qrun += n * n; }
      if (t == 0) evoff[B] = qtot;
    }
  }
  __syncthreads();
  PL_STAMP(3);
  // ---- the CSR segments in arrival order (plain stores: every in-edge kept its position) ...
#pragma unroll
  for (int k = 0; k < EPT; ++k)
    if (ed[k] >= 0) lfill[rp[ed[k]] + epos[k]] = ((t + k * PLAN_T) << 12) | es[k];   // edge id (< 2^14: E <= 12288) above the source node (< 2^12)
  if (early.host != nullptr && early.max_graph_edges > 0) {
    // (early report) a graph with more in-edges than the fused GINE stage stages in LDS: its rows of the CSR are contiguous
    for (int g = t; g < B; g += PLAN_T)
      if (rp[gp[g + 1]] - rp[gp[g]] > early.max_graph_edges) atomicOr(&s_big, 1);
  }
  __syncthreads();
  PL_STAMP(4);
  // ---- ... then by edge id: an in-edge's place is the number of smaller keys in its segment (the keys are distinct)
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    if (ed[k] >= 0) {
      const int key = ((t + k * PLAN_T) << 12) | es[k];
      const int lo = rp[ed[k]], hi = rp[ed[k] + 1];
      int rank = 0;
      for (int q = lo; q < hi; ++q) rank += lfill[q] < key ? 1 : 0;
      lperm[lo + rank] = key;
    }
  }
  __syncthreads();
  PL_STAMP(5);
  // ---- write out
  for (int i = t; i <= N; i += PLAN_T) rowptr[i] = rp[i];
  for (int i = t; i < E; i += PLAN_T) { const int k = lperm[i]; col[i] = k & 4095; eperm[i] = k >> 12; }
  for (int i = t; i <= B; i += PLAN_T) graph_ptr[i] = gp[i];
  for (int i = t; i < N; i += PLAN_T) {
    const int g = ng[i];
    int nv = 0, gi = 0;
    if (g >= 0) { gi = g; nv = slots_of(gp[g + 1] - gp[g], kmax); }
    node_graph[i] = gi;
    nvalid[i] = nv;
  }
  if (t == 0) { status[ST_ERR] = s_err; if (!do_bins) status[ST_NMAX] = s_nmax; status[ST_DEGMAX] = s_dmax; status[3] = 0; }
  if (t >= 4 && t < 8) status[t] = 0;
  if (early.host != nullptr) {
    if (t == 0) {
      early_put(early.host, EH_ERR, s_err);
      early_put(early.host, EH_DEGMAX, s_dmax);
      early_put(early.host, EH_EDGES, s_big);
      if (!do_bins) { early_put(early.host, EH_NMAX, s_nmax); early_put(early.host, EH_PHI, 0); early_put(early.host, EH_RHO, 0); }
      early_done(early.host, 0);
      if (!do_bins) { early_done(early.host, 1); early_done(early.host, 2); }
      if (early.node_ids == nullptr && early.edge_ids == nullptr) { early_put(early.host, EH_IDS, 0); early_done(early.host, 3); }
    }
  }
  PL_STAMP(6);
}

// ============================================================================ general path: five launches
// K1: per node — graph id, graph boundaries; zero the in-degree counters and the status words.
__global__ void k_plan_nodes(const int64_t* __restrict__ batch, int64_t N, int64_t B,
                             int32_t* __restrict__ graph_ptr, int32_t* __restrict__ node_graph,
                             int32_t* __restrict__ deg, int32_t* __restrict__ status) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 8) status[i] = 0;
  if (i == 0) graph_ptr[B] = (int32_t)N;
  if (i >= N) return;
  int64_t g = batch[i];
  deg[i] = 0;
  if (g < 0 || g >= B) { node_graph[i] = 0; return; }   // reported by k_plan_degree
  node_graph[i] = (int32_t)g;
  int64_t gp = (i == 0) ? -1 : batch[i - 1];
  if (gp < -1) gp = -1;
  if (gp < g)
    for (int64_t t = gp + 1; t <= g; ++t) graph_ptr[t] = (int32_t)i;   // ids in between are empty graphs
  if (i == N - 1)
    for (int64_t t = g + 1; t < B; ++t) graph_ptr[t] = (int32_t)N;
}

// K2: validation (per node and per edge) and in-degree counting.
__global__ void k_plan_degree(const int64_t* __restrict__ batch, const int64_t* __restrict__ ei, int64_t E,
                              int64_t N, int64_t B, int32_t* __restrict__ deg, int32_t* __restrict__ status) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < N) {
    int64_t g = batch[t];
    if (g < 0 || g >= B) atomicOr(&status[ST_ERR], ERR_GRAPH_ID);
    if (t > 0 && batch[t - 1] > g) atomicOr(&status[ST_ERR], ERR_UNSORTED);
  }
  if (t >= E) return;
  int64_t s = ei[t], d = ei[E + t];
  if (s < 0 || s >= N || d < 0 || d >= N) {
    atomicOr(&status[ST_ERR], ERR_EDGE_RANGE);
    return;
  }
  if (batch[s] != batch[d]) atomicOr(&status[ST_ERR], ERR_EDGE_CROSS);
  atomicAdd(&deg[d], 1);
}

// K3: block 0 — exclusive scans: deg -> rowptr (and cursor copy), n_b^2 -> evoff; nvalid; maxima.
//     block 1 — the work bins (graph_ptr staged in LDS).
__global__ __launch_bounds__(PLAN_T) void k_plan_scan(int64_t N, int64_t B, int kmax,
                                                      const int32_t* __restrict__ graph_ptr,
                                                      const int32_t* __restrict__ node_graph,
                                                      int32_t* __restrict__ deg /* in: degree, out: cursor=rowptr */,
                                                      int32_t* __restrict__ rowptr, int32_t* __restrict__ nvalid,
                                                      int64_t* __restrict__ evoff, int32_t* __restrict__ status, BinsDev bd) {
  extern __shared__ int sm[];
  const int T = blockDim.x, t = threadIdx.x;
  if (blockIdx.x == 1) {
    int* gp = sm;
    for (int64_t i = t; i <= B; i += T) gp[i] = graph_ptr[i];
    __syncthreads();
    plan_rho_block(gp, (int)B, kmax, bd, sm + B + 4);
    __syncthreads();
    // (the slab chain's pattern table follows the block's scratch when the launch made room for it: up to PLAN_SLAB_BMAX graphs)
    plan_bins_block(gp, (int)B, kmax, bd, sm + B + 4, (kmax == 0 && B <= PLAN_SLAB_BMAX) ? sm + B + 4 + 5 * (int)B + 3 * 66 + 32 + 8 : nullptr);
    return;
  }
  __shared__ long long part[PLAN_T];
  __shared__ int maxs[2];
  if (t == 0) { maxs[0] = 0; maxs[1] = 0; }
  __syncthreads();
  {
    int64_t per = (N + T - 1) / T;
    int64_t lo = (int64_t)t * per, hi = lo + per < N ? lo + per : N;
    long long s = 0;
    int dmax = 0;
    for (int64_t i = lo; i < hi; ++i) { s += deg[i]; dmax = dmax > deg[i] ? dmax : deg[i]; }
    part[t] = s;
    atomicMax(&maxs[1], dmax);
    __syncthreads();
    for (int off = 1; off < T; off <<= 1) {   // Hillis-Steele inclusive scan over the T partials
      long long v = (t >= off) ? part[t - off] : 0;
      __syncthreads();
      part[t] += v;
      __syncthreads();
    }
    long long run = part[t] - s;
    for (int64_t i = lo; i < hi; ++i) {
      int dgi = deg[i];
      rowptr[i] = (int32_t)run;
      deg[i] = (int32_t)run;
      run += dgi;
    }
    if (t == T - 1) rowptr[N] = (int32_t)part[T - 1];
    __syncthreads();
  }
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
  for (int64_t i = t; i < N; i += T) {
    int g = node_graph[i];
    nvalid[i] = slots_of(graph_ptr[g + 1] - graph_ptr[g], kmax);
  }
  if (t == 0) { status[ST_NMAX] = maxs[0]; status[ST_DEGMAX] = maxs[1]; }
}

// K4: per edge — scatter into its destination's segment (arrival order arbitrary, fixed by k_plan_sort).
__global__ __launch_bounds__(256) void k_plan_fill(const int64_t* __restrict__ ei, int64_t E, int64_t N,
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

// K5: per node — sort each CSR segment by edge id (insertion sort; molecular degrees are <= 4).
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

static bool plan_one_launch(int64_t N, int64_t E, int64_t B) { return N > 0 && N <= PS_NMAX && E <= PS_EMAX && B <= PS_BMAX; }

extern "C" int sn_batch_plan_early_supported(int64_t N, int64_t E, int64_t B) { return plan_one_launch(N, E, B) ? 1 : 0; }

extern "C" int sn_batch_plan_ex(const int64_t* batch, int64_t N, int64_t B, const int64_t* edge_index,
                                int64_t E, int kmax, int32_t* graph_ptr, int32_t* node_graph,
                                int32_t* nvalid, int64_t* evoff, int32_t* rowptr, int32_t* col,
                                int32_t* eperm, int32_t* status, const sn_plan_bins* bins, int32_t* scratch,
                                const sn_plan_early* early, void* stream) {
  SN_REQUIRE(N >= 0 && B >= 0 && E >= 0, "sn_batch_plan: negative size");
  SN_REQUIRE(N < (1ll << 31) && E < (1ll << 31), "sn_batch_plan: N/E exceed int32");
  SN_REQUIRE(graph_ptr && node_graph && nvalid && evoff && rowptr && status && scratch,
             "sn_batch_plan: null output");
  SN_REQUIRE(N == 0 || batch, "sn_batch_plan: null batch");
  SN_REQUIRE(E == 0 || (edge_index && col && eperm), "sn_batch_plan: null edge arrays");
  BinsDev bd{};
  const bool do_bins = bins != nullptr;
  if (do_bins) {
    SN_REQUIRE(bins->rho_bin0 && bins->meta && bins->phi_max_bins >= 0, "sn_batch_plan: incomplete sn_plan_bins");
    const int ncolp = (bins->phi_bin_col != nullptr) + (bins->phi_col_bin0 != nullptr) + (bins->phi_col_mem != nullptr) + (bins->phi_col_off != nullptr);
    SN_REQUIRE(ncolp == 4 || (ncolp == 0 && bins->phi_bin_mem), "sn_batch_plan: give all four column arrays of sn_plan_bins, or none of them and phi_bin_mem");
    SN_REQUIRE(B <= BINS_BMAX, "sn_batch_plan: work bins support at most %d graphs per batch (got %lld)", BINS_BMAX, (long long)B);
    SN_REQUIRE(!bins->phi_bin_mem || (reinterpret_cast<uintptr_t>(bins->phi_bin_mem) & 15) == 0, "sn_batch_plan: phi_bin_mem must be 16-byte aligned");
    bd = BinsDev{bins->phi_bin_col, bins->phi_max_bins, bins->phi_col_bin0, bins->phi_col_mem, bins->phi_col_off,
                 bins->rho_bin0, bins->meta, bins->phi_bin_mem};
  }
  EarlyDev ed{};
  if (early != nullptr) {
    SN_REQUIRE(early->host, "sn_batch_plan_ex: sn_plan_early.host (pinned int32[16]) missing");
    SN_REQUIRE(plan_one_launch(N, E, B), "sn_batch_plan_ex: the early report needs the one-launch plan (N <= %d, E <= %d, B <= %d): ask "
               "sn_batch_plan_early_supported first", PS_NMAX, PS_EMAX, PS_BMAX);
    SN_REQUIRE((!early->node_ids && !early->edge_ids) || do_bins, "sn_batch_plan_ex: the feature-id check runs beside the bin planner (bins must be given)");
    SN_REQUIRE((!early->node_ids || (early->n_node_ids >= 0 && early->node_vocab > 0)) && (!early->edge_ids || (early->n_edge_ids >= 0 && early->edge_vocab > 0)),
               "sn_batch_plan_ex: feature-id counts / table rows");
    ed = EarlyDev{early->node_ids, early->node_ids ? early->n_node_ids : 0, early->node_vocab, early->edge_ids,
                  early->edge_ids ? early->n_edge_ids : 0, early->edge_vocab, early->max_graph_edges, early->host};
  }
  hipStream_t st = (hipStream_t)stream;
  if (plan_one_launch(N, E, B)) {
    const size_t lds0 = (size_t)((PS_BMAX + 4) + PS_NMAX + (PS_NMAX + 4) + 2 * PS_EMAX + 32 + PS_NMAX) * sizeof(int);
    const size_t lds1 = (size_t)((PS_BMAX + 4) + PS_BINS_INTS + PAT_INTS) * sizeof(int);
    const size_t lds = lds0 > lds1 ? lds0 : lds1;
    static bool init = false;
    if (!init) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_plan_small), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds) != hipSuccess)
        return fail(SN_ERR_LAUNCH, "sn_batch_plan: cannot raise the dynamic LDS limit to %zu", lds);
      init = true;
    }
    const bool ids = ed.host != nullptr && (ed.node_ids != nullptr || ed.edge_ids != nullptr);
    hipLaunchKernelGGL(k_plan_small, dim3(do_bins ? (ids ? 4 : 3) : 1), dim3(PLAN_T), lds, st, batch, (int)N, (int)B, edge_index, (int)E,
                       kmax, graph_ptr, node_graph, nvalid, evoff, rowptr, col, eperm, status, bd, do_bins ? 1 : 0, ed);
    SN_CHECK_LAUNCH("sn_batch_plan");
    return SN_OK;
  }
  int32_t* deg = scratch;  // [N]
  const int T = 256;
  hipLaunchKernelGGL(k_plan_nodes, dim3((unsigned)cdiv(N > 0 ? N : 1, T)), dim3(T), 0, st, batch, N, B, graph_ptr, node_graph, deg,
                     status);
  const int64_t ne = N > E ? N : E;
  hipLaunchKernelGGL(k_plan_degree, dim3((unsigned)cdiv(ne > 0 ? ne : 1, T)), dim3(T), 0, st, batch, edge_index, E, N, B, deg,
                     status);
  const size_t lds3 = do_bins ? (size_t)((B + 4) + 5 * B + 3 * 66 + 32 + 8 + ((kmax == 0 && B <= PLAN_SLAB_BMAX) ? PAT_INTS : 0)) * sizeof(int) : 0;
  if (lds3 > 48 * 1024) {
    static bool init3 = false;
    if (!init3) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_plan_scan), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)((size_t)(6 * BINS_BMAX + 4 + 3 * 66 + 32 + 8) * sizeof(int))) != hipSuccess)
        return fail(SN_ERR_LAUNCH, "sn_batch_plan: cannot raise the dynamic LDS limit");
      init3 = true;
    }
  }
  hipLaunchKernelGGL(k_plan_scan, dim3(do_bins ? 2 : 1), dim3(PLAN_T), lds3, st, N, B, kmax, graph_ptr, node_graph, deg, rowptr,
                     nvalid, evoff, status, bd);
  if (E > 0) {
    hipLaunchKernelGGL(k_plan_fill, dim3((unsigned)cdiv(E, T)), dim3(T), 0, st, edge_index, E, N, deg, col, eperm);
    hipLaunchKernelGGL(k_plan_sort, dim3((unsigned)cdiv(N, T)), dim3(T), 0, st, N, rowptr, col, eperm);
  }
  SN_CHECK_LAUNCH("sn_batch_plan");
  return SN_OK;
}

extern "C" int sn_batch_plan(const int64_t* batch, int64_t N, int64_t B, const int64_t* edge_index,
                             int64_t E, int kmax, int32_t* graph_ptr, int32_t* node_graph,
                             int32_t* nvalid, int64_t* evoff, int32_t* rowptr, int32_t* col,
                             int32_t* eperm, int32_t* status, const sn_plan_bins* bins, int32_t* scratch,
                             void* stream) {
  return sn_batch_plan_ex(batch, N, B, edge_index, E, kmax, graph_ptr, node_graph, nvalid, evoff, rowptr, col, eperm, status, bins,
                          scratch, nullptr, stream);
}

#ifdef SN_PROFILE
extern "C" int sn_prof_read_plan(long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_pprof), sizeof(long long) * 64); }
#endif

extern "C" int64_t sn_phi_bins_bound(int64_t B, int kmax) {
  // every column has height <= min(kmax, 64); at worst one graph per column
  const int k = kmax < 0 ? -kmax : kmax;          // (full-slot mode: exactly |kmax| bins per column)
  const int64_t h = kmax < 0 ? k : ((k > 0 && k < 64) ? k : 64);
  return B * h + 1;
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

// The CSR of two disjoint copies of a batch (nodes N .. 2N-1 = the second copy): what the training step's stacked phi(+x) / phi(-x)
// aggregation walks (train_stage.py).  Index plumbing — it replaced eight torch launches (add, cat) per step.
namespace sn {
__global__ __launch_bounds__(256) void k_plan_double(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, int64_t N, int64_t E,
                                                     int32_t* __restrict__ rowptr2, int32_t* __restrict__ col2) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i <= N) {
    const int32_t r = rowptr[i];
    rowptr2[i] = r;
    if (i > 0) rowptr2[N + i] = r + (int32_t)E;
  }
  if (i < E) {
    const int32_t c = col[i];
    col2[i] = c;
    col2[E + i] = c + (int32_t)N;
  }
}
}  // namespace sn

extern "C" int sn_plan_double_i32(const int32_t* rowptr, const int32_t* col, int64_t N, int64_t E, int32_t* rowptr2, int32_t* col2, void* stream) {
  SN_REQUIRE(rowptr && rowptr2 && N >= 0 && E >= 0 && (E == 0 || (col && col2)), "sn_plan_double_i32: bad arguments");
  SN_REQUIRE(2 * N < (1ll << 31) && 2 * E < (1ll << 31), "sn_plan_double_i32: the doubled graph does not fit 32-bit indices");
  const int64_t work = (N + 1 > E ? N + 1 : E);
  hipLaunchKernelGGL(sn::k_plan_double, dim3((unsigned)sn::cdiv(work, 256)), dim3(256), 0, (hipStream_t)stream, rowptr, col, N, E, rowptr2, col2);
  SN_CHECK_LAUNCH("sn_plan_double_i32");
  return SN_OK;
}
