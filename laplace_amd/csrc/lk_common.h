// Shared helpers for the gfx950 curvature kernels (internal; the public ABI is include/laplace_hip.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "laplace_hip.h"

namespace lk {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WAVE = 64;

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return LK_ELAUNCH;
  }
  return LK_OK;
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Block-wide sum for blockDim.x == 256 (4 waves); result valid in every thread.
__device__ __forceinline__ float block_sum_256(float v, float* red /* >= 4 floats LDS */) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// Exact unsigned division of n < 2^31 by a run-time constant d >= 1 without a divide:
// q = (n * M) >> (32 + s),  s = ceil(log2 d),  M = ceil(2^(32+s) / d)  (fits 34 bits; product < 2^64).
struct FastDiv {
  uint64_t M;
  int s;
  int d;
};
static inline FastDiv make_fastdiv(int d) {
  FastDiv f;
  f.d = d;
  f.s = 0;
  while ((1ll << f.s) < d) ++f.s;
  const unsigned __int128 one = 1;
  f.M = (uint64_t)(((one << (32 + f.s)) + (unsigned)d - 1) / (unsigned)d);
  return f;
}
__device__ __forceinline__ int fdiv(int n, const FastDiv& f) {
  return (int)(((uint64_t)(uint32_t)n * f.M) >> (32 + f.s));
}

}  // namespace lk

#define LK_REQUIRE(cond, ...)            \
  do {                                   \
    if (!(cond)) {                       \
      lk::set_error(__VA_ARGS__);        \
      return LK_EINVAL;                  \
    }                                    \
  } while (0)
