// mfma_f64_probe.hip -- does the f64 matrix pipe of gfx950 beat its f64 vector pipe? (decides whether a DFT-as-GEMM formulation
// of the FFT builtins can pay: a 16-point DFT as a dense 16 x 16 complex product costs 6.4x the flops of its radix-2 form.)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f64_probe tools/probes/mfma_f64_probe.hip && /tmp/mfma_f64_probe
// Both kernels keep every SIMD busy with long chains of independent operations on registers only; flops are counted as issued.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef double double4v __attribute__((ext_vector_type(4)));

template <int CHAINS>
__global__ void __launch_bounds__(256) k_mfma(double* out, int iters, double seed) {
  double4v acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = (double4v){seed + c, seed, seed, seed};
  const double a = seed * 1e-3 + threadIdx.x * 1e-9, b = 1.0 + seed * 1e-6;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
  }
  double s = 0.0;
  for (int c = 0; c < CHAINS; ++c) s += acc[c].x + acc[c].y + acc[c].z + acc[c].w;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CHAINS>
__global__ void __launch_bounds__(256) k_fma(double* out, int iters, double seed) {
  double acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = seed + c;
  const double a = 1.0 - seed * 1e-9, b = seed * 1e-6 + threadIdx.x * 1e-12;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_fma(acc[c], a, b);
  }
  double s = 0.0;
  for (int c = 0; c < CHAINS; ++c) s += acc[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class K>
static double time_ms(K launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch();                                       // warm-up
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 5.0;
}

int main() {
  const int blocks = 256 * 8, threads = 256, iters = 20000;       // 8 blocks of 4 waves per CU: 8 waves per SIMD
  double* out = nullptr;
  hipMalloc(&out, sizeof(double) * blocks * threads);
  const double waves = (double)blocks * threads / 64.0;
  // one v_mfma_f64_16x16x4_f64 = 16 x 16 x 4 multiply-adds = 2048 flop per wave; one v_fma_f64 = 64 x 2 = 128 flop per wave
  const double ms_m = time_ms([&] { hipLaunchKernelGGL(k_mfma<8>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.25); });
  const double ms_f = time_ms([&] { hipLaunchKernelGGL(k_fma<16>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.25); });
  const double tf_m = waves * iters * 8 * 2048.0 / (ms_m * 1e-3) / 1e12;
  const double tf_f = waves * iters * 16 * 128.0 / (ms_f * 1e-3) / 1e12;
  printf("{\"probe\": \"f64 matrix vs vector pipe, gfx950\", \"mfma_f64_16x16x4_tflops\": %.2f, \"mfma_ms\": %.3f, "
         "\"v_fma_f64_tflops\": %.2f, \"fma_ms\": %.3f, \"ratio_matrix_over_vector\": %.3f, "
         "\"dft16_flop_dense\": 2048, \"dft16_flop_radix2\": 320, \"dense_over_radix2\": 6.4}\n",
         tf_m, ms_m, tf_f, ms_f, tf_m / tf_f);
  hipFree(out);
  return 0;
}
