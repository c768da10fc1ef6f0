// Probe: what the matrix pipe of an MI355X sustains with the instruction mix of lk::conv_f16x2_kernel.
//   variant 0: 12 x v_mfma_f32_32x32x16_f16 per iteration from registers only (4 accumulators x 3, the split scheme)
//   variant 1: + the 8 ds_read_b128 fragment reads of one k16 step per 12 MFMAs (two register sets, counted waits)
//   variant 2: + one s_barrier per 24 MFMAs (the stage hand-over)
//   variant 3: + 10 LDS-DMA loads (global_load_lds, 16 B per lane) per 24 MFMAs from a small L2-resident buffer
//   variants 4 / 5 / 6: the same loads in the convolution's address pattern (64-byte pieces of rows 128 / 256 / 64 B apart)
// Launch shape = the conv kernel's: 256-thread workgroups, `wgs_per_cu` x 256 of them.  Prints fp16 TFLOP/s and the
// fraction of 2.5 PFLOP/s.   Build: hipcc --offload-arch=gfx950 -O3 mfma_peak_probe.hip -o mfma_peak_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

template <int V>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void probe(const _Float16* __restrict__ src,
                                                                                        float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 80 KB: two 40 KB stages
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 80 * 1024 / 4; i += 256) ((unsigned*)smem)[i] = 0x3c003a00u + (i & 0xff);
  __syncthreads();
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  f16x8 fr[2][8];
  for (int s = 0; s < 2; ++s)
    for (int j = 0; j < 8; ++j)
      for (int e = 0; e < 8; ++e) fr[s][j][e] = (_Float16)(0.001f * (lane + j + e + s));
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a)
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  auto reads = [&](int set, int buf) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      // conflict-free: 64 lanes x 16 B contiguous
      const unsigned ad = lds0 + buf * 40960 + ((wave * 8 + j) * 1024 + lane * 16) % 40960;
      asm volatile("ds_read_b128 %0, %1" : "=v"(fr[set][j]) : "v"(ad));
    }
  };
  auto wait8 = [&](int set) {
    asm volatile("s_waitcnt lgkmcnt(8)"
                 : "+v"(fr[set][0]), "+v"(fr[set][1]), "+v"(fr[set][2]), "+v"(fr[set][3]), "+v"(fr[set][4]),
                   "+v"(fr[set][5]), "+v"(fr[set][6]), "+v"(fr[set][7]));
  };
  auto mfmas = [&](int set) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        f32x16 c = acc[a * 2 + b];
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[set][2 + a], fr[set][4 + b], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[set][a], fr[set][6 + b], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[set][a], fr[set][4 + b], c, 0, 0, 0);
        acc[a * 2 + b] = c;
      }
  };
  if (V >= 1) { reads(0, 0); reads(1, 0); wait8(0); }
  for (int it = 0; it < iters; ++it) {
    if (V >= 2) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_waitcnt(0x0f70);
      __builtin_amdgcn_s_barrier();
    }
    if (V == 3) {
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        const _Float16* sp = src + ((size_t)((blockIdx.x * 7 + it * 13 + i) & 1023) * 256 + threadIdx.x) * 8;
        __builtin_amdgcn_global_load_lds((gbl_void*)sp, (lds_void*)(smem + (it & 1) * 40960 + (i * 256 + wave * 64) * 16), 16, 0, 0);
      }
    }
    if (V >= 4) {
      // the convolution's pattern: four lanes read one 64-byte piece of a row, rows ROWB bytes apart (V = 4: 128 B =
      // half of every cache line is used; 5: 256 B; 6: 64 B = contiguous), 64 rows per instruction, from a 1 MB window
      constexpr int ROWB = V == 4 ? 128 : (V == 5 ? 256 : 64);
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        const size_t row = ((size_t)(blockIdx.x * 37 + it * 11 + i) * 64 + (threadIdx.x >> 2)) & (size_t)(1048576 / ROWB - 1);
        const char* sp = (const char*)src + row * ROWB + (threadIdx.x & 3) * 16;
        __builtin_amdgcn_global_load_lds((gbl_void*)sp, (lds_void*)(smem + (it & 1) * 40960 + (i * 256 + wave * 64) * 16), 16, 0, 0);
      }
    }
    if (V >= 1) { reads(0, (it + 1) & 1); wait8(1); }
    __builtin_amdgcn_sched_barrier(0);
    mfmas(1);
    __builtin_amdgcn_sched_barrier(0);
    if (V >= 1) { reads(1, (it + 1) & 1); wait8(0); }
    __builtin_amdgcn_sched_barrier(0);
    mfmas(0);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  float s = 0.f;
  for (int a = 0; a < 4; ++a)
    for (int r = 0; r < 16; ++r) s += acc[a][r];
  for (int j = 0; j < 8; ++j) s += (float)fr[1][j][0];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int V>
static void run(const _Float16* src, float* out, int wgs_per_cu, int iters) {
  hipFuncSetAttribute((const void*)probe<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  const int grid = 256 * wgs_per_cu;
  probe<V><<<grid, 256, 80 * 1024>>>(src, out, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<V><<<grid, 256, 80 * 1024>>>(src, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)grid * 4 * iters * 24 * 32768.0;
  printf("variant %d  wgs/cu %d  %.3f ms  %.1f TFLOP/s fp16  frac of 2500 = %.3f   (hipError %d)\n", V, wgs_per_cu, ms,
         flop / ms / 1e9, flop / ms / 1e9 / 2500.0, (int)hipGetLastError());
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 4000;
  _Float16* src; float* out;
  hipMalloc(&src, 1024 * 256 * 16 + 4096);
  hipMemset(src, 0x3c, 1024 * 256 * 16 + 4096);
  hipMalloc(&out, 4096);
  for (int rep = 0; rep < 2; ++rep) {
    run<0>(src, out, 2, iters);
    run<1>(src, out, 2, iters);
    run<2>(src, out, 2, iters);
    run<3>(src, out, 2, iters);
    run<4>(src, out, 2, iters);
    run<5>(src, out, 2, iters);
    run<6>(src, out, 2, iters);
    run<0>(src, out, 1, iters);
    run<2>(src, out, 1, iters);
    run<3>(src, out, 1, iters);
  }
  return 0;
}
