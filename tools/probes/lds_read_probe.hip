// Probe: what a CU's LDS delivers per clock to fragment reads — the transposing read ds_read_b64_tr_b16 in the address pattern of
// gram16_kernel / pixpair13_kernel (a 16-lane group reads a [4 k][16 channel] block of a [rows][64 channel] image, 64-byte chunks
// XOR-swizzled by the row) against plain ds_read_b64 and ds_read_b128 on lane-linear addresses.  Waves only read (8 independent
// reads in flight each).  Prints bytes per nominal clock (2.4 GHz) and CU, and clocks per wave-instruction.
// Build: hipcc --offload-arch=gfx950 -O3 lds_read_probe.hip -o lds_read_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int KIND>  // 0: ds_read_b64_tr_b16, gram16 pattern; 1: ds_read_b64 lane-linear; 2: ds_read_b128 lane-linear; 3: tr, lane-linear
__global__ __launch_bounds__(1024) void probe(int iters, int* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<int*>(smem)[i] = i;
  __syncthreads();
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  unsigned addr[8];
  for (int j = 0; j < 8; ++j) {
    if (KIND == 0) {
      const int grp = lane >> 4, r16 = lane & 15;
      const int row = (grp >> 1) * 8 + (r16 >> 2) + (j & 1) * 4, col = (j >> 1) * 32 % 64 + (grp & 1) * 16 + (r16 & 3) * 4;
      addr[j] = lds0 + ((wave * 4 + (j >> 2)) % 16) * 4096 + row * 128 + (((col >> 5) ^ ((row >> 1) & 1)) << 6) + (col & 31) * 2;
    } else {
      addr[j] = lds0 + ((wave * 8 + j) * 1024) % 65536 + lane * (KIND == 2 ? 16 : 8);
    }
  }
  int acc = 0;
  for (int it = 0; it < iters; ++it) {
    if (KIND == 2) {
      i32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(v[j]) : "v"(addr[j]));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < 8; ++j) acc ^= v[j][0];
    } else {
      i32x2 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (KIND == 1) asm volatile("ds_read_b64 %0, %1" : "=v"(v[j]) : "v"(addr[j]));
        else asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v[j]) : "v"(addr[j]));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < 8; ++j) acc ^= v[j][0];
    }
  }
  if (acc == 0x12345678) out[threadIdx.x] = acc;
}

template <int KIND>
static void run(int* out, int waves, const char* what) {
  const int iters = 4000;
  hipFuncSetAttribute((const void*)probe<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  probe<KIND><<<256, waves * 64, 65536>>>(50, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<KIND><<<256, waves * 64, 65536>>>(iters, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)iters * 8 * waves, clk = ms * 1e-3 * 2.4e9, bytes = n * 64 * (KIND == 2 ? 16 : 8);
  printf("%-46s waves/CU %2d  %7.3f ms  %5.1f clocks per wave-instruction and CU, %5.1f B per clock and CU\n", what, waves, ms, clk / n, bytes / clk);
}

int main() {
  int* out;
  hipMalloc(&out, 4096 * 4);
  for (int w : {4, 8, 16}) {
    run<0>(out, w, "ds_read_b64_tr_b16, gram16 fragment pattern");
    run<3>(out, w, "ds_read_b64_tr_b16, lane-linear addresses");
    run<1>(out, w, "ds_read_b64, lane-linear");
    run<2>(out, w, "ds_read_b128, lane-linear");
  }
  return 0;
}
