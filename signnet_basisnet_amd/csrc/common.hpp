// common.hpp — shared device/host helpers for libsignnet_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/signnet_hip.h"

namespace sn {

// ---------------------------------------------------------------- host side: errors
inline char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}
inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}
#define SN_REQUIRE(cond, ...) \
  do {                        \
    if (!(cond)) return sn::fail(SN_ERR_ARG, __VA_ARGS__); \
  } while (0)
#define SN_CHECK_LAUNCH(name)                                                                   \
  do {                                                                                          \
    hipError_t e__ = hipGetLastError();                                                         \
    if (e__ != hipSuccess) return sn::fail(SN_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e__)); \
  } while (0)

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------- device side
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WAVE = 64;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// One 16x16x4 fp32 MFMA step:  D[i][j] += sum_{g<4} A[i][g] * B[g][j]
//   lane l supplies A[i = l&15][g = l>>4] and B[g = l>>4][j = l&15];
//   lane l holds D[i = 4*(l>>4) + r][j = l&15] in acc[r].
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 acc) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
}

// Row-tile GEMM convention used everywhere in this library (transposed product, so that the
// accumulator layout of one GEMM is the operand layout of the next — no LDS round trip):
//   a wave owns 16 activation rows; lane l = (row = l & 15, g = l >> 4);
//   an activation row tile is held as frag[kk] (f32x4) = X[row][16*kk + 4*g + t], t = 0..3;
//   weights are pre-packed as Wp[ot][kk][lane][t] = W[16*ot + (lane&15)][16*kk + 4*(lane>>4) + t];
//   out[ot][r] = sum_k W[16*ot + 4*g + r][k] * X[row][k]  ==  Y[row][16*ot + 4*g + r]
// i.e. the MFMA "A" operand is the weight fragment (i = output channel), the "B" operand is the
// activation fragment (j = row), and the k index of lane group g at step (kk, t) is 16*kk+4*g+t —
// a permutation of k that both operands share, so the sum is over every k exactly once.
// XCD-aware remap of a 1-D block id: consecutive *logical* blocks land on the same XCD (and its
// private L2) in chunks of `chunk` blocks.  Hardware places physical block b on XCD b % 8
// (observed, used for speed only).  Returns a logical id in [0, ceil(nblk/(8*chunk))*8*chunk);
// callers bounds-check against their real work size.
__device__ __forceinline__ int64_t xcd_remap(int64_t b, int chunk) {
  int64_t xcd = b & 7, s = b >> 3;
  int64_t q = s / chunk, r = s - q * chunk;
  return (q * 8 + xcd) * chunk + r;
}

}  // namespace sn
