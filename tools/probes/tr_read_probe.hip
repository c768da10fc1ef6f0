// Probe: lane -> element mapping of ds_read_b64_tr_b16 (gfx950) and fp16 MFMA behaviour on subnormal inputs.
// Build: hipcc --offload-arch=gfx950 -O2 tr_read_probe.hip -o tr_read_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// LDS holds 16-bit words w[i] = i (i < 4096).  Every lane reads at byte address addr[lane]; we dump what it got.
__global__ void tr_probe(const int* addr, unsigned short* out) {
  __shared__ unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int a = addr[threadIdx.x];
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)((char*)lds + a));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}

// MFMA f16 with subnormal inputs: A = 2^-20 (subnormal in fp16), B = 2^10 -> expect 16 * 2^-10 per element if not flushed
__global__ void mfma_sub_probe(float* out) {
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)9.5367431640625e-07f; b[j] = (_Float16)1024.f; }
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  out[threadIdx.x] = c[0];
}

int main() {
  int h_addr[64];
  unsigned short h_out[256];
  int* d_addr; unsigned short* d_out; float* d_f;
  hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out)); hipMalloc(&d_f, 64 * 4);
  // experiment 1: lane l reads at byte address l * 8 (contiguous 4-element groups): a [16 rows? ][...] guess
  for (int variant = 0; variant < 3; ++variant) {
    for (int l = 0; l < 64; ++l) {
      if (variant == 0) h_addr[l] = l * 8;                       // rows of 4 elements, lane-linear
      if (variant == 1) h_addr[l] = (l & 15) * 64 + (l >> 4) * 8;  // 16 rows of pitch 64 B, 4 column groups
      if (variant == 2) h_addr[l] = (l & 3) * 128 + ((l >> 2) & 3) * 8 + (l >> 4) * 32;  // 4 rows x pitch 128B
    }
    hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(tr_probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    printf("variant %d (word index = byte address / 2)\n", variant);
    for (int l = 0; l < 64; ++l)
      printf("  lane %2d addr %4d(word %4d) -> %4d %4d %4d %4d\n", l, h_addr[l], h_addr[l] / 2, h_out[l * 4], h_out[l * 4 + 1],
             h_out[l * 4 + 2], h_out[l * 4 + 3]);
  }
  float h_f[64];
  hipLaunchKernelGGL(mfma_sub_probe, dim3(1), dim3(64), 0, 0, d_f);
  hipMemcpy(h_f, d_f, sizeof(h_f), hipMemcpyDeviceToHost);
  printf("mfma f16 subnormal A: got %g, expected %g if subnormals are honoured (0 if flushed)\n", h_f[0], 16 * 9.5367431640625e-07 * 1024);
  return 0;
}
