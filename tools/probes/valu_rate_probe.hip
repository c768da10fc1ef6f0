// Probe: issue rate of the fp32 vector instructions the predictive's pair sums are made of — v_fma_f32 against v_pk_fma_f32
// (two products per instruction), independent and dependent back to back, at one / two / four waves per SIMD; and the same
// pk stream beside a partner wave's MFMAs.  Prints instructions per SIMD and clock at the nominal 2.4 GHz (1 / 4 = a wave64
// instruction every four clocks) and the ratio that matters: does the packed form halve the time of 2 x the scalar one?
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate_probe.hip -o valu_rate_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// MODE 0: v_fma_f32, 16 independent chains; 1: v_pk_fma_f32, 16 independent chains; 2: v_pk_fma_f32, 2 chains (dependent pairs
// back to back); 3: v_fma_f32, 2 chains; 4: MODE 1 on even waves, v_mfma_f32_16x16x32_f16 on odd waves
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

// matrix instructions alone: KIND 0 = v_mfma_f32_32x32x16_f16, 1 = v_mfma_f32_16x16x32_f16, 2 = v_mfma_f32_16x16x16_f16; four chains
template <int KIND>
__global__ __launch_bounds__(1024) void mprobe(int iters, float seed, float* out) {
  h8 a, b;
  for (int i = 0; i < 8; ++i) a[i] = (_Float16)(seed + i), b[i] = (_Float16)(seed - i);
  h4 a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
  f16v D[4];
  f4 d[4];
  for (int j = 0; j < 4; ++j) {
    for (int i = 0; i < 16; ++i) D[j][i] = 0;
    d[j] = f4{0, 0, 0, 0};
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (KIND == 0) D[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, D[c], 0, 0, 0);
        if (KIND == 1) d[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d[c], 0, 0, 0);
        if (KIND == 2) d[c] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, d[c], 0, 0, 0);
      }
  }
  float s = 0;
  for (int c = 0; c < 4; ++c) s += D[c][3] + d[c][1];
  if (s == 12345.f) out[threadIdx.x] = s;
}

template <int KIND>
static void mrun(float* out, int waves_per_simd, const char* what) {
  const int iters = 5000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  mprobe<KIND><<<256, 256 * waves_per_simd>>>(50, 1.5f, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  mprobe<KIND><<<256, 256 * waves_per_simd>>>(iters, 1.5f, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)iters * 64 * waves_per_simd, clk = ms * 1e-3 * 2.4e9;
  const double flop = (KIND == 2 ? 8192.0 : (KIND == 1 ? 16384.0 : 32768.0)) * n * 1024;
  printf("%-40s waves/SIMD %d  %8.3f ms  %.1f nominal clocks per instruction and SIMD, %.0f TFLOP/s\n", what, waves_per_simd, ms, clk / n,
         flop / (ms * 1e-3) / 1e12);
}

// Co-execution: PARTNER = true: odd waves issue matrix instructions, even waves `nv` vector instructions per matrix instruction
// of the partner; PARTNER = false: every wave issues both, interleaved (nv vector instructions behind each matrix instruction).
// PK: v_pk_fma_f32 instead of v_fma_f32 (not run: hipcc keeps that variant's accumulators in scratch; the packed stream beside a
// partner's MFMAs is what `probe<4>` measures).  Per loop trip: 16 matrix instructions (four chains) and 16 nv vector instructions.
template <bool PARTNER, bool PK, int NV>
__global__ __launch_bounds__(1024) void coprobe(int iters, float seed, float* out, int do_m, int do_v) {
  const int wave = threadIdx.x >> 6;
  h8 a, b;
  for (int i = 0; i < 8; ++i) a[i] = (_Float16)(seed + i), b[i] = (_Float16)(seed - i);
  f4 d[4] = {f4{0, 0, 0, 0}, f4{0, 0, 0, 0}, f4{0, 0, 0, 0}, f4{0, 0, 0, 0}};
  f2 acc[16];
  float acc1[16];
  f2 x = {seed, seed + 1}, y = {seed * 0.5f, seed};
  const float x1 = seed, y1 = seed * 0.5f;
  for (int i = 0; i < 16; ++i) acc[i] = f2{(float)i, 1.f}, acc1[i] = i;
  const bool m = do_m && (!PARTNER || (wave & 1)), v = do_v && (!PARTNER || !(wave & 1));
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (m) d[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d[j & 3], 0, 0, 0);
      if (v) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          if constexpr (PK) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[(j * NV + i) & 15]) : "v"(x), "v"(y));
          else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc1[(j * NV + i) & 15]) : "v"(x1), "v"(y1));
        }
      }
    }
  }
  float s = d[0][0] + d[1][1] + d[2][2] + d[3][3];
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc1[i];
  if (s == 12345.f) out[threadIdx.x] = s;
}

template <bool PARTNER, bool PK, int NV>
static void corun(float* out, int waves_per_simd) {
  float ms[3];
  for (int k = 0; k < 3; ++k) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    coprobe<PARTNER, PK, NV><<<256, 256 * waves_per_simd>>>(50, 1.5f, out, 1, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    coprobe<PARTNER, PK, NV><<<256, 256 * waves_per_simd>>>(5000, 1.5f, out, k != 1, k != 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms[k], e0, e1);
  }
  printf("%s, %d x %s per v_mfma_f32_16x16x32_f16, %d waves/SIMD: matrix alone %.3f ms, vector alone %.3f ms, together %.3f ms (sum %.3f)\n",
         PARTNER ? "partner waves" : "same wave   ", NV, PK ? "v_pk_fma_f32" : "v_fma_f32   ", waves_per_simd, ms[0], ms[1], ms[2], ms[0] + ms[1]);
}

// MODE 4: `vmul` x the vector work on even waves (so that both halves take about as long alone)
template <int MODE>
__global__ __launch_bounds__(1024) void probe(int iters, float seed, float* out, int vmul = 1) {
  const int wave = threadIdx.x >> 6;
  if (MODE == 4 && !(wave & 1)) iters *= vmul;
  if (MODE == 4 && (wave & 1)) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) a[i] = (_Float16)(seed + i), b[i] = (_Float16)(seed - i);
    f4 d0 = {0, 0, 0, 0}, d1 = d0, d2 = d0, d3 = d0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d1, 0, 0, 0);
        d2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d2, 0, 0, 0);
        d3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d3, 0, 0, 0);
      }
    }
    if (d0[0] + d1[1] + d2[2] + d3[3] == 12345.f) out[threadIdx.x] = 1.f;
    return;
  }
  constexpr int NCH = (MODE == 2 || MODE == 3) ? 2 : 16;
  if (MODE == 0 || MODE == 3) {
    float acc[NCH], x = seed, y = seed * 0.5f;
    for (int i = 0; i < NCH; ++i) acc[i] = i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 64 / NCH; ++j)
#pragma unroll
        for (int i = 0; i < NCH; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y));
    }
    float s = 0;
    for (int i = 0; i < NCH; ++i) s += acc[i];
    if (s == 12345.f) out[threadIdx.x] = s;
  } else {
    f2 acc[NCH], x = {seed, seed + 1}, y = {seed * 0.5f, seed};
    for (int i = 0; i < NCH; ++i) acc[i] = f2{(float)i, 1.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 64 / NCH; ++j)
#pragma unroll
        for (int i = 0; i < NCH; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y));
    }
    float s = 0;
    for (int i = 0; i < NCH; ++i) s += acc[i][0] + acc[i][1];
    if (s == 12345.f) out[threadIdx.x] = s;
  }
}

template <int MODE>
static double run(float* out, int waves_per_simd, const char* what) {
  const int threads = 256 * waves_per_simd, iters = 20000;  // one workgroup per CU: 4 SIMDs x waves_per_simd waves
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  probe<MODE><<<256, threads>>>(100, 1.5f, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<MODE><<<256, threads>>>(iters, 1.5f, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const int valu_waves = MODE == 4 ? waves_per_simd / 2 : waves_per_simd;
  const double inst_per_simd = (double)iters * 64 * valu_waves;
  const double clk = ms * 1e-3 * 2.4e9;
  printf("%-64s waves/SIMD %d  %8.3f ms  %.3f vector instructions per SIMD and nominal clock (%.2f clocks each)\n", what, waves_per_simd,
         ms, inst_per_simd / clk, clk / inst_per_simd);
  return ms;
}

int main() {
  float* out;
  hipMalloc(&out, 4096 * 4);
  for (int w : {1, 2, 4}) {
    const double a = run<0>(out, w, "v_fma_f32, 16 independent chains");
    const double b = run<1>(out, w, "v_pk_fma_f32, 16 independent chains");
    const double c = run<2>(out, w, "v_pk_fma_f32, 2 chains (dependent every other instruction)");
    const double d = run<3>(out, w, "v_fma_f32, 2 chains");
    printf("   packed / scalar time at equal instruction count: independent %.2f, dependent %.2f\n", b / a, c / d);
  }
  mrun<0>(out, 1, "v_mfma_f32_32x32x16_f16");
  mrun<0>(out, 2, "v_mfma_f32_32x32x16_f16");
  mrun<1>(out, 1, "v_mfma_f32_16x16x32_f16");
  mrun<1>(out, 2, "v_mfma_f32_16x16x32_f16");
  mrun<2>(out, 1, "v_mfma_f32_16x16x16_f16");
  mrun<2>(out, 2, "v_mfma_f32_16x16x16_f16");
  // do a wave's vector instructions run beside its partner's matrix instructions?  alone / alone / together
  for (int vmul : {1, 4, 8}) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipEventRecord(e0);
    probe<4><<<256, 512>>>(5000, 1.5f, out, vmul);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("one wave per SIMD: 320 000 v_mfma_f32_16x16x32_f16; its partner: %d x 320 000 v_pk_fma_f32: %.3f ms together\n", vmul, ms);
  }
  corun<true, false, 4>(out, 2);
  corun<true, false, 2>(out, 2);
  corun<false, false, 4>(out, 1);
  corun<false, false, 4>(out, 2);
  corun<false, false, 2>(out, 2);
  corun<true, false, 4>(out, 4);
  return 0;
}
