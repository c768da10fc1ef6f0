// Probe: what one CU of an MI355X ingests through `global_load_lds` (16 B per lane, 1 KB per wave-instruction) — the path the
// persistent convolution stages its window and weights through — by source pattern and by the number of issuing waves.
//   pattern 0: every workgroup re-reads ONE 288 KB block (a layer's weight slice: L2-resident, contiguous kilobytes)
//   pattern 1: every workgroup streams its own part of a 2 GB buffer, contiguous kilobytes (HBM)
//   pattern 2: the window's pattern from HBM: 32 pixels x 32 bytes per instruction, pixel rows `rowb` bytes apart
//   pattern 3: pattern 2 over a 4 MB region per XCD-ful of workgroups (L2 / MALL resident)
// Nothing computes; waves keep `depth` instructions in flight.  Prints GB/s per CU, B per nominal clock (2.4 GHz) per CU, TB/s.
// Build: hipcc --offload-arch=gfx950 -O3 ldsdma_ingest_probe.hip -o ldsdma_ingest_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

template <int PAT, int DEPTH>
__global__ __launch_bounds__(512) void probe(const char* __restrict__ src, size_t bytes, int rowb, int iters, float* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // NW x DEPTH x 2 KB
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const size_t gw = (size_t)blockIdx.x * nw + wave, tw = (size_t)gridDim.x * nw;
  char* dst = smem + wave * (2 * DEPTH * 1024);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const size_t k = (size_t)it * DEPTH + d;
      const char* sp;
      if (PAT == 0) sp = src + ((k * nw + wave) * 1024) % (288 * 1024) + lane * 16;
      else if (PAT == 1) sp = src + ((k * tw + gw) * 1024) % bytes + lane * 16;
      else {
        // instruction = 32 consecutive pixels x 32 B (two 16-byte lanes per pixel); a wave walks pixels, then the next 32-byte chunk
        const size_t region = PAT == 2 ? bytes : (size_t)(4 << 20);
        const size_t px = ((k * tw + gw) * 32 + (lane >> 1));
        const size_t off = (px * (size_t)rowb) % region;
        sp = src + off + ((k / 64) % (rowb / 32)) * 32 + (lane & 1) * 16;
      }
      __builtin_amdgcn_global_load_lds((gbl_void*)sp, (lds_void*)(dst + ((it & 1) * DEPTH + d) * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (reinterpret_cast<float*>(smem)[threadIdx.x] == 12345.678f) out[threadIdx.x] = 1.f;
}

template <int PAT, int DEPTH>
static void run(const char* src, size_t bytes, float* out, int nw, int wgs_per_cu, int rowb, int iters) {
  const int lds = nw * 2 * DEPTH * 1024;
  hipFuncSetAttribute((const void*)probe<PAT, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  const int grid = 256 * wgs_per_cu;
  probe<PAT, DEPTH><<<grid, nw * 64, lds>>>(src, bytes, rowb, 50, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<PAT, DEPTH><<<grid, nw * 64, lds>>>(src, bytes, rowb, iters, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double total = (double)grid * nw * iters * DEPTH * 1024.0;
  printf("pattern %d depth %2d  waves/wg %d  wgs/cu %d  rowb %4d  %8.3f ms  %7.1f GB/s per CU  %5.1f B/clk/CU  %6.2f TB/s  (err %d)\n", PAT,
         DEPTH, nw, wgs_per_cu, rowb, ms, total / ms / 1e6 / 256, total / ms / 1e6 / 256 / 2.4, total / ms / 1e9, (int)hipGetLastError());
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  const size_t bytes = (size_t)2 << 30;
  char* src;
  float* out;
  hipMalloc(&src, bytes + 4096);
  hipMemset(src, 0x3c, bytes + 4096);
  hipMalloc(&out, 4096);
  for (int rep = 0; rep < 2; ++rep) {
    for (int nw : {4, 8}) {
      run<0, 4>(src, bytes, out, nw, 1, 0, iters);
      run<0, 8>(src, bytes, out, nw, 1, 0, iters);
      run<0, 16>(src, bytes, out, nw, 1, 0, iters / 2);
      run<1, 4>(src, bytes, out, nw, 1, 0, iters);
      run<1, 8>(src, bytes, out, nw, 1, 0, iters);
      run<1, 16>(src, bytes, out, nw, 1, 0, iters / 2);
      for (int rowb : {128, 256, 1024}) {
        run<2, 8>(src, bytes, out, nw, 1, rowb, iters);
        run<3, 8>(src, bytes, out, nw, 1, rowb, iters);
      }
    }
    run<0, 8>(src, bytes, out, 4, 2, 0, iters);
    run<1, 8>(src, bytes, out, 4, 2, 0, iters);
    run<2, 8>(src, bytes, out, 4, 2, 128, iters);
  }
  return 0;
}
