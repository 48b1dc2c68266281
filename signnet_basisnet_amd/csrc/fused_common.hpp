// fused_common.hpp — helpers shared by the whole-stage kernels (fused_phi / fused_rho / fused_gnn): the split-bf16
// GEMM primitives (operand split, LDS weight ring, workgroup GEMM) and small LDS / reduction helpers.
#pragma once
#include "common.hpp"

namespace sn {

// Phase timeline for kernel tuning (scratch builds with -DSN_PROFILE only; never in the shipped library).
#ifdef SN_PROFILE
static __device__ long long g_prof[64];
#define SN_STAMP(i) do { if (sn_prof_on && blockIdx.x == 0 && threadIdx.x == 0) g_prof[i] = clock64(); } while (0)
#define SN_ACCUM(i, t0) do { if (sn_prof_on && blockIdx.x == 0 && threadIdx.x == 0) g_prof[i] += clock64() - (t0); } while (0)
#define SN_PROF_ON(cond) const bool sn_prof_on = (cond)
#else
#define SN_STAMP(i) do { } while (0)
#define SN_ACCUM(i, t0) do { } while (0)
#define SN_PROF_ON(cond) do { } while (0)
#endif

__device__ __forceinline__ f32x4 ld4(const float* __restrict__ p) {
  float4 t = *reinterpret_cast<const float4*>(p);
  return f32x4{t.x, t.y, t.z, t.w};
}
__device__ __forceinline__ f32x4 lds_ld4(const float* p) {
  float4 t = *reinterpret_cast<const float4*>(p);
  return f32x4{t.x, t.y, t.z, t.w};
}
__device__ __forceinline__ void lds_st4(float* p, f32x4 v) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ f32x4 relu4(f32x4 v) {
  return f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Weight fragments are fetched with buffer loads: descriptor in SGPRs (wave-uniform base), one VGPR of
// per-lane offset (lane*16) and an immediate/SGPR fragment offset — no 64-bit address VGPR per fragment.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t weight_rsrc(const float* p, unsigned bytes) {
  unsigned long long a = reinterpret_cast<unsigned long long>(p);
  unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
  unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  void* q = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, bytes, 0x00020000);
}

// =====================================================================================================
// fp32 GEMMs on the bf16 matrix pipe: three-way significand split (six partial products).
//
// An fp32 value is split EXACTLY into three bf16 pieces h + m + l (8 significand bits each, truncation); of the
// nine partial products of x*w the six with weight >= 2^-16 (hh, hm, mh, hl, lh, mm) are evaluated as
// v_mfma_f32_16x16x32_bf16 (exact products, fp32 accumulate); the three dropped ones are <= 2^-23 |x||w| — the
// size of one fp32 rounding.  16x16x32 bf16 issues in ~17 cycles/SIMD for 8192 MACs against 8 x 32 cycles for
// the same MACs on v_mfma_f32_16x16x4_f32: 2.5x the fp32 matrix peak at fp32 accuracy.
//
// Layouts (transposed product as in common.hpp; a K block is 32 channels = two 16-channel operand chunks):
//   activation operand of K block kb: lane (row = l&15, g = l>>4) holds 8 bf16 k-slots s = 0..7 = channels
//     32*kb + 16*(s>>2) + 4*g + (s&3)  — exactly in[2kb][0..3], in[2kb+1][0..3] of the fp32 operand layout, so the
//     accumulators of one GEMM split in place into the operand of the next;
//   packed weights ("split-packed linear", sn_pack_split_f32): per 16-output tile ot one CHUNK of
//     NF = 3*NKB weight fragments [kb][plane h,m,l][lane][8 bf16] (1 KiB each) followed by SN_SPLIT_EPI = 3
//     epilogue fragments [e][lane][4 f32] = vec_e[16*ot + 4*(lane>>4) + t] (bias / folded-BatchNorm vectors).
//
// Weight stream: all 4 waves of a workgroup consume the same fragments, so the chunks are staged ONCE per
// workgroup in an LDS ring (SPLIT_RING chunks) by LDS-DMA (global_load_lds_dwordx4: no staging VGPRs), chunk
// c+RING issued as soon as every wave has read chunk c; fragments go LDS -> VGPR one K block ahead of their MFMAs.
// Sync (MI355X guide, "Read a staged buffer one phase AFTER the wait that retires it"): each wave waits its own
// DMA of chunk c+1 with a counted vmcnt, then the workgroup barrier publishes it; ds_reads of chunk c+1 are issued
// only after that barrier.  The stream continues across consecutive GEMMs (`wnext`).
// =====================================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(3))) char lds_char_t;

constexpr int SPLIT_RING = 3;
static_assert(SPLIT_RING == 3, "slot_of() is x mod 3");
constexpr int SPLIT_EPI = SN_SPLIT_EPI;

struct Split8 { u32x4 h, m, l; };

__device__ __forceinline__ unsigned pack_hi16(float a, float b) {   // bf16 bits of a | bf16 bits of b << 16
  return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
__device__ __forceinline__ Split8 split8(f32x4 a, f32x4 b) {
  const float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  float h[8], m[8], l[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    h[i] = __uint_as_float(__float_as_uint(x[i]) & 0xffff0000u);
    const float r = x[i] - h[i];                       // exact
    m[i] = __uint_as_float(__float_as_uint(r) & 0xffff0000u);
    l[i] = r - m[i];                                   // exact, <= 8 significant bits
  }
  Split8 s;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    s.h[i] = pack_hi16(h[2 * i], h[2 * i + 1]);
    s.m[i] = pack_hi16(m[2 * i], m[2 * i + 1]);
    s.l[i] = pack_hi16(l[2 * i], l[2 * i + 1]);
  }
  return s;
}
// half of split8: the four values `a` become k-slots 4*half .. 4*half+3 of the operand block `s` (same arithmetic as split8)
__device__ __forceinline__ void split4(f32x4 a, Split8& s, int half) {
  float h[4], m[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    h[i] = __uint_as_float(__float_as_uint(a[i]) & 0xffff0000u);
    const float r = a[i] - h[i];                       // exact
    m[i] = __uint_as_float(__float_as_uint(r) & 0xffff0000u);
    l[i] = r - m[i];                                   // exact
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    s.h[2 * half + i] = pack_hi16(h[2 * i], h[2 * i + 1]);
    s.m[2 * half + i] = pack_hi16(m[2 * i], m[2 * i + 1]);
    s.l[2 * half + i] = pack_hi16(l[2 * i], l[2 * i + 1]);
  }
}
template <int NT>
__device__ __forceinline__ void split_rows(const f32x4 (&in)[NT], Split8 (&xs)[(NT + 1) / 2]) {
#pragma unroll
  for (int kb = 0; kb < (NT + 1) / 2; ++kb)
    xs[kb] = split8(in[2 * kb], (2 * kb + 1 < NT) ? in[2 * kb + 1] : f32x4{0.f, 0.f, 0.f, 0.f});
}
__device__ __forceinline__ f32x4 mfma_bf(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// workgroup barrier for LDS traffic only: does not drain in-flight LDS-DMA (a __syncthreads() would)
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// NW: waves of the workgroup (all of them share the DMA of every chunk).  RING: chunks of LDS the stream cycles through — RING - 1
// chunks are in flight while one is being consumed; the latency-bound per-graph kernels (few rows, long chains) use a deeper ring
// than the throughput kernels, whose default of 3 leaves the LDS to the activation images.
template <int NT, int NW = 4, int RING = SPLIT_RING>
struct WRing {
  static constexpr int NKB = (NT + 1) / 2;
  static constexpr int NF = 3 * NKB;                 // weight fragments per chunk
  static constexpr int NFE = NF + SPLIT_EPI;         // + epilogue fragments
  static constexpr int CHUNK = NFE * 1024;           // bytes
  static constexpr int BYTES = RING * CHUNK;
  static constexpr int DEPTH = RING;
  static constexpr int LPC = (NFE + NW - 1) / NW;    // DMA instructions every wave issues per chunk
  lds_char_t* base;   // LDS
  int pos;            // ring slot of chunk 0 of the current GEMM (wave-uniform)
  int wave, lane;

  __device__ __forceinline__ void init(void* lds_base, int wave_, int lane_) {
    base = (lds_char_t*)lds_base; pos = 0; wave = wave_; lane = lane_;
  }
  // wait until at most N of my chunk shares are still in flight
  template <int N>
  __device__ __forceinline__ void wait_shares() const { wait_vmcnt<N * LPC>(); }
  __device__ __forceinline__ int slot_of(int c) const {
    const int x = pos + c;
    if (RING == 3) return x - 3 * ((x * 43) >> 7);       // x mod 3, x < 128
    if (RING == 4) return x & 3;
    return x % RING;
  }
  // this wave's share of the DMA of chunk `c` of packed matrix `w` into ring slot `slot`.  Buffer form: the matrix
  // base lives in an SGPR descriptor, the chunk / fragment offset in an SGPR, lane*16 in one VGPR — no 64-bit
  // per-lane address per DMA (which the compiler would precompute for every chunk of every matrix and spill).
  __device__ __forceinline__ void issue(const void* w, int c, int slot) const {
    const __amdgpu_buffer_rsrc_t rs = weight_rsrc(reinterpret_cast<const float*>(w), 0x7fffffff);
    // (the wave's fragment offset is laundered through an empty asm: otherwise the compiler precomputes the LDS
    //  address and the buffer offset of every (slot, fragment) pair of every GEMM as loop invariants and the ~40
    //  scalar registers they occupy spill — and a kernel with a private segment pays ~6 us at launch)
    int wo = wave * 1024;
    asm volatile("" : "+s"(wo));
    lds_char_t* dst = base + slot * CHUNK + wo;
    const int src = c * CHUNK + wo;
#pragma unroll
    for (int f = 0; f < LPC; ++f) {
      int fo = NW * 1024 * f;
      if (NW * f + NW - 1 >= NFE) {      // branch-free tail: the surplus waves re-stage the last fragment
        const int last = (NFE - 1) * 1024 - wo;
        fo = fo < last ? fo : last;
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(dst + fo), 16, lane * 16, src + fo, 0, 0);
    }
  }
  // start of the weight stream (once per kernel when the GEMMs chain, else once per GEMM): chunks 0..RING-1 of `w`
  // issued, chunk 0 visible to every wave.  All waves call it.
  __device__ __forceinline__ void prologue(const void* w, int nchunks) {
    lds_barrier();   // nobody still reads the ring
    pos = 0;
#pragma unroll
    for (int c = 0; c < RING; ++c)
      if (c < nchunks) issue(w, c, c);
    if (nchunks >= RING) wait_shares<RING - 1>(); else wait_vmcnt<0>();
    lds_barrier();
  }
  // before the kernel exits: no LDS-DMA of this wave may still be in flight (the LDS would be handed to another workgroup)
  __device__ __forceinline__ void drain() const { wait_vmcnt<0>(); }
};

struct WFrag { u32x4 h, m, l; };

// acc(ot) = W[16 outputs of tile ot] . x  for ot < NTO, consumed by epi(ot, acc, e0, e1, e2, pv) (e*: the chunk's
// epilogue vectors in the accumulator layout; pv = pre(ot), an f32x4 the caller wants fetched BEFORE the tile's
// MFMAs, e.g. the residual from LDS).  EVERY wave of the workgroup must call it (barriers, DMA shares);
// `live` = this wave has rows (a dead wave only keeps the stream going).  `wnext` (never null): the matrix whose
// first chunks are staged behind this one's — the next wg_gemm_split() of the workgroup must be on `wnext`; the
// kernel starts the stream with WRing::prologue(first matrix) and ends with WRing::drain().  SWAP: operands exchanged -> acc[r] = Y[row = 4g + r][out = 16*ot + (l&15)].
template <int NT, int NTO, bool SWAP, bool EPIV = true, int NW = 4, int RING = SPLIT_RING, typename Pre, typename Epi>
__device__ __forceinline__ void wg_gemm_split(WRing<NT, NW, RING>& ring, const void* w, const void* wnext, bool live,
                                              const Split8 (&xs)[(NT + 1) / 2], Pre pre, Epi epi) {
  using R = WRing<NT, NW, RING>;
  constexpr int NKB = R::NKB;
  constexpr bool CHAIN = NTO >= RING;     // the stream runs on into the next matrix; else: one prologue per GEMM
  if (!CHAIN) ring.prologue(w, NTO);
  typedef __attribute__((address_space(3))) const u32x4 lds_u32x4;
  typedef __attribute__((address_space(3))) const f32x4 lds_f32x4;
  int sl[RING];
#pragma unroll
  for (int i = 0; i < RING; ++i) sl[i] = ring.slot_of(i);
  auto slot = [&](int c) { return sl[c % RING]; };   // c is a compile-time constant
  auto stage = [&](int ot) {   // after the barrier of step ot: refill the slot of chunk ot with chunk ot + RING
    if (CHAIN) {
      const int pc = ot + RING;
      if (pc < NTO) ring.issue(w, pc, slot(ot)); else ring.issue(wnext, pc - NTO, slot(ot));
    }
  };
  if (live) {
    int ln16 = ring.lane * 16;
    asm volatile("" : "+v"(ln16));   // per call: keeps the per-lane fragment addresses of every GEMM from being hoisted and held
    const lds_char_t* lbase = ring.base + ln16;
    auto rd = [&](int c, int kb) {
      const lds_char_t* p = lbase + slot(c) * R::CHUNK + kb * 3072;
      WFrag f;
      f.h = *(lds_u32x4*)(p);
      f.m = *(lds_u32x4*)(p + 1024);
      f.l = *(lds_u32x4*)(p + 2048);
      return f;
    };
    auto mm = [&](const WFrag& f, const Split8& x, f32x4& a0, f32x4& a1) {
      if (!SWAP) {
        a1 = mfma_bf(f.l, x.h, a1);
        a0 = mfma_bf(f.m, x.h, a0);
        a1 = mfma_bf(f.h, x.l, a1);
        a0 = mfma_bf(f.h, x.m, a0);
        a1 = mfma_bf(f.m, x.m, a1);
        a0 = mfma_bf(f.h, x.h, a0);
      } else {
        a1 = mfma_bf(x.h, f.l, a1);
        a0 = mfma_bf(x.h, f.m, a0);
        a1 = mfma_bf(x.l, f.h, a1);
        a0 = mfma_bf(x.m, f.h, a0);
        a1 = mfma_bf(x.m, f.m, a1);
        a0 = mfma_bf(x.h, f.h, a0);
      }
    };
    WFrag fa = rd(0, 0), fb;
#pragma unroll
    for (int ot = 0; ot < NTO; ++ot) {
      f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
      // epilogue operands first: their LDS latency hides behind this tile's MFMAs
      f32x4 e0 = {0.f, 0.f, 0.f, 0.f}, e1 = e0, e2 = e0;
      if (EPIV) {     // EPIV = false: the Linear has no epilogue vectors (their fragments are still staged, never read)
        const lds_char_t* pe = lbase + slot(ot) * R::CHUNK + R::NF * 1024;
        e0 = *(lds_f32x4*)(pe); e1 = *(lds_f32x4*)(pe + 1024); e2 = *(lds_f32x4*)(pe + 2048);
      }
      const f32x4 pv = pre(ot);
#pragma unroll
      for (int kb = 0; kb + 1 < NKB; ++kb) {
        fb = rd(ot, kb + 1);
        __builtin_amdgcn_sched_barrier(0);
        mm(fa, xs[kb], a0, a1);
        __builtin_amdgcn_sched_barrier(0);
        fa = fb;
      }
      if (CHAIN) {
        // all my reads of chunk ot are complete and my share of chunk ot+1 has landed -> barrier -> refill the slot
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        ring.template wait_shares<RING - 2>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        stage(ot);
      }
      if (ot + 1 < NTO) fb = rd(ot + 1, 0);
      __builtin_amdgcn_sched_barrier(0);
      mm(fa, xs[NKB - 1], a0, a1);
      __builtin_amdgcn_sched_barrier(0);
      epi(ot, a0 + a1, e0, e1, e2, pv);
      fa = fb;
    }
  } else if (CHAIN) {
#pragma unroll
    for (int ot = 0; ot < NTO; ++ot) {
      ring.template wait_shares<RING - 2>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      stage(ot);
    }
  }
  ring.pos = ring.slot_of(NTO);
}

// =====================================================================================================
// wg_gemm_split for the 8-wave phi workgroup (round 4): a FOUR-slot ring for the same look-ahead of three chunks.  With three
// slots the chunk barrier of tile ot had to certify "every read of chunk ot has returned" (s_waitcnt lgkmcnt(0) in front of it) so
// that the slot could be refilled right behind it; with four the refill goes into the slot of chunk ot-1, which every wave left a
// whole tile ago — a fragment read is retired by the MFMA that consumes it — so nothing but the wave's own DMA share is waited for
// at the barrier, and a wave may take the barrier at ANY K block of its tile (BAR_KB):
//     + waves (BAR_KB = NKB-1):  [kb0 kb1 kb2 | wait, barrier, DMA issue | kb3, epilogue]
//     - waves (BAR_KB = 1):      [kb0 | wait, barrier, DMA issue | kb1 kb2 kb3, epilogue]
// i.e. the two sign waves of a SIMD are half a tile out of phase at every barrier: one is in the middle of its tile's MFMAs while
// the other waits, issues and applies its epilogue.  Measured on the headline batch (same box, HIP events): one-stream form with
// three slots 119.1 us; four slots, no LDS drain, both barriers at NKB-1: 114.6; with the half-tile lag: 112.3-113.0.
// What was measured on the way and dropped (all slower than the one-stream form): strict ping-pong with two
// barriers per tile (L | M segments alternating between the wave groups, 130 us: the L segment — 4 waves x 16 KB of ds_read_b128
// behind the DMA issue — is a ~500-cycle serial chain, longer than the 384-cycle MFMA segment it was meant to hide behind), one
// barrier per tile with complementary segment ORDER (M,L / L,M: 133 us — a wave's own L + M chain is what bounds the interval, not
// the pipe), s_setprio in any combination (no effect on MFMA arbitration between the two waves of a SIMD), each accumulator as one
// back-to-back MFMA chain instead of two alternating ones (no effect), half of the waves issuing the whole DMA (116-120 us).
// Protocol:
//   * at barrier ot (every wave calls it once per tile) chunk ot+1 becomes visible (every wave waited for its share of it right
//     before) and chunk ot+3 is issued into the slot of chunk ot-1;
//   * the first fragment of chunk ot+1 is fetched during the tile's last K block, i.e. behind barrier ot for either BAR_KB;
//   * entry state (ring_prologue3, or the previous GEMM's `wnext`): chunk 0 visible, chunks 1 and 2 in flight.
// =====================================================================================================
constexpr int LAG_RING = 4;

template <int NT, int NW>
__device__ __forceinline__ void ring_prologue3(WRing<NT, NW, LAG_RING>& ring, const void* w) {
  lds_barrier();   // nobody still reads the ring
  ring.pos = 0;
  ring.issue(w, 0, 0);
  ring.issue(w, 1, 1);
  ring.issue(w, 2, 2);
  ring.template wait_shares<2>();
  lds_barrier();
}

// the same without the wait: for a caller whose next workgroup-wide __syncthreads() (which drains the DMA) comes before the first GEMM
template <int NT, int NW>
__device__ __forceinline__ void ring_prologue3_issue(WRing<NT, NW, LAG_RING>& ring, const void* w) {
  lds_barrier();
  ring.pos = 0;
  ring.issue(w, 0, 0);
  ring.issue(w, 1, 1);
  ring.issue(w, 2, 2);
}

template <int NT, int NTO, int BAR_KB, int NW, typename Pre, typename Epi>
__device__ __forceinline__ void wg_gemm_split_lag(WRing<NT, NW, LAG_RING>& ring, const void* w, const void* wnext, bool live,
                                                  const Split8 (&xs)[(NT + 1) / 2], Pre pre, Epi epi) {
  using R = WRing<NT, NW, LAG_RING>;
  constexpr int NKB = R::NKB;
  static_assert(NTO >= 3 && BAR_KB >= 0 && BAR_KB < NKB, "look-ahead of three chunks; the barrier sits in front of a K block of the tile");
  typedef __attribute__((address_space(3))) const u32x4 lds_u32x4;
  typedef __attribute__((address_space(3))) const f32x4 lds_f32x4;
  int sl[LAG_RING];
#pragma unroll
  for (int i = 0; i < LAG_RING; ++i) sl[i] = ring.slot_of(i);
  auto slot = [&](int c) { return sl[c % LAG_RING]; };   // c is a compile-time constant
  auto sync_stage = [&](int ot) {   // barrier ot: chunk ot + 1 published, chunk ot + 3 -> the slot of chunk ot - 1
    ring.template wait_shares<1>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const int pc = ot + 3;
    if (pc < NTO) ring.issue(w, pc, slot(pc)); else ring.issue(wnext, pc - NTO, slot(pc));
  };
  if (live) {
    int ln16 = ring.lane * 16;
    asm volatile("" : "+v"(ln16));   // per call: keeps the per-lane fragment addresses of every GEMM from being hoisted and held
    const lds_char_t* lbase = ring.base + ln16;
    auto rd = [&](int c, int kb) {
      const lds_char_t* p = lbase + slot(c) * R::CHUNK + kb * 3072;
      WFrag f;
      f.h = *(lds_u32x4*)(p);
      f.m = *(lds_u32x4*)(p + 1024);
      f.l = *(lds_u32x4*)(p + 2048);
      return f;
    };
    WFrag fa = rd(0, 0), fb;
#pragma unroll
    for (int ot = 0; ot < NTO; ++ot) {
      f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
      // epilogue operands first: their LDS latency hides behind this tile's MFMAs
      const lds_char_t* pe = lbase + slot(ot) * R::CHUNK + R::NF * 1024;
      const f32x4 e0 = *(lds_f32x4*)(pe), e1 = *(lds_f32x4*)(pe + 1024), e2 = *(lds_f32x4*)(pe + 2048);
      const f32x4 pv = pre(ot);
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        if (kb == BAR_KB) sync_stage(ot);
        if (kb + 1 < NKB) fb = rd(ot, kb + 1);
        else if (ot + 1 < NTO) fb = rd(ot + 1, 0);     // (behind barrier ot, which published chunk ot + 1)
        __builtin_amdgcn_sched_barrier(0);
        a1 = mfma_bf(fa.l, xs[kb].h, a1);
        a0 = mfma_bf(fa.m, xs[kb].h, a0);
        a1 = mfma_bf(fa.h, xs[kb].l, a1);
        a0 = mfma_bf(fa.h, xs[kb].m, a0);
        a1 = mfma_bf(fa.m, xs[kb].m, a1);
        a0 = mfma_bf(fa.h, xs[kb].h, a0);
        __builtin_amdgcn_sched_barrier(0);
        fa = fb;
      }
      epi(ot, a0 + a1, e0, e1, e2, pv);
    }
  } else {
#pragma unroll
    for (int ot = 0; ot < NTO; ++ot) sync_stage(ot);
  }
  ring.pos = ring.slot_of(NTO);
}

// sum over the 4 lane groups (lanes l, l^16, l^32, l^48) that hold one activation row
__device__ __forceinline__ float row_allsum(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

// wg_gemm_split without a prefetch hook
struct NoPre { __device__ __forceinline__ f32x4 operator()(int) const { return f32x4{0.f, 0.f, 0.f, 0.f}; } };

}  // namespace sn
