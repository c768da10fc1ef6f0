// Probe: what a workgroup generation costs on an MI355X when the grid is several rounds of LDS-limited workgroups — the
// launch shapes of lk::conv_win_f16x2_kernel (2304 x 512 threads, 100 KB of LDS: one per CU) and of the generic kernel
// (4608 x 256 threads, 80 KB: two per CU).  Variants: empty body; one dependent global load (first-load latency);
// `spin` microseconds of s_sleep work.   Build: hipcc --offload-arch=gfx950 -O3 wg_dispatch_probe.hip -o wg_dispatch_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ void probe(const float* __restrict__ src, float* out, int mode, int spin) {
  extern __shared__ float smem[];
  float v = 0.f;
  if (mode >= 1) {
    v = src[(blockIdx.x * 4096 + threadIdx.x) & 0xfffff];
    smem[threadIdx.x] = v;
    __syncthreads();
    v = smem[(threadIdx.x + 1) & (blockDim.x - 1)];
  }
  for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(32);  // 32 * 64 cycles = ~1 us at 2 GHz
  if (v == 12345.678f) out[threadIdx.x] = v;
}

static float run(const float* src, float* out, int grid, int block, int lds, int mode, int spin) {
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  probe<<<grid, block, lds>>>(src, out, mode, spin);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) probe<<<grid, block, lds>>>(src, out, mode, spin);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 100.f;  // us per launch
}

int main() {
  float *src, *out;
  hipMalloc(&src, 4 << 20), hipMalloc(&out, 4096);
  hipMemset(src, 0, 4 << 20);
  struct { int grid, block, lds; const char* what; } shapes[] = {
      {2304, 512, 100 * 1024, "2304 x 512 thr, 100 KB (1 / CU, 9 rounds)"}, {4608, 256, 80 * 1024, "4608 x 256 thr, 80 KB (2 / CU, 9 rounds)"},
      {256, 512, 100 * 1024, "256 x 512 thr, 100 KB (one round)"},          {2304, 512, 1024, "2304 x 512 thr, 1 KB"},
      {4608, 256, 1024, "4608 x 256 thr, 1 KB"}};
  for (auto& s : shapes)
    for (int mode = 0; mode < 2; ++mode)
      for (int spin : {0, 10})
        printf("%-44s mode %d spin %2d us: %8.1f us per launch\n", s.what, mode, spin, run(src, out, s.grid, s.block, s.lds, mode, spin));
  return 0;
}
