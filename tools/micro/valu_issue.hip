// valu_issue.hip -- what one wavefront alone pays per VALU instruction on gfx950: independent v_mul_f32 streams of width 1..8 (width 1 =
// a dependent chain), with 4 or 64 active lanes, one wave per workgroup, one workgroup. Cycles from s_memtime.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/valu_issue.bin tools/micro/valu_issue.hip ; run it on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int W, bool MIX>
__global__ void k(float* out, uint64_t* cyc, int lanes, float m) {
  float x[W];
  for (int i = 0; i < W; ++i) x[i] = 1.0f + threadIdx.x + i;
  uint64_t best = ~0ull;
  if ((int)threadIdx.x < lanes) {
    for (int rep = 0; rep < 4; ++rep) {
      const uint64_t t0 = __builtin_readcyclecounter();
#pragma unroll
      for (int n = 0; n < 512; ++n) {
#pragma unroll
        for (int i = 0; i < W; ++i) {
          if (MIX && (n & 1)) asm volatile("v_max_f32 %0, %1, %2" : "=v"(x[i]) : "v"(x[i]), "v"(m));
          else asm volatile("v_mul_f32 %0, %1, %2" : "=v"(x[i]) : "v"(x[i]), "v"(m));
        }
      }
      const uint64_t t1 = __builtin_readcyclecounter();
      if (t1 - t0 < best) best = t1 - t0;
    }
  }
  float s = 0;
  for (int i = 0; i < W; ++i) s += x[i];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) *cyc = best;
}

template <int W, bool MIX> void run(float* out, uint64_t* cyc, int lanes) {
  hipLaunchKernelGGL((k<W, MIX>), dim3(1), dim3(64), 0, 0, out, cyc, lanes, 0.999f);
  hipDeviceSynchronize();
  uint64_t c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("streams %d %s lanes %2d: %6.2f cycles per instruction\n", W, MIX ? "mul/max" : "mul    ", lanes, (double)c / (512.0 * W));
}

int main() {
  float* out; uint64_t* cyc;
  hipMalloc(&out, 64 * 4); hipMalloc(&cyc, 8);
  for (int lanes : {4, 64}) {
    run<1, false>(out, cyc, lanes); run<2, false>(out, cyc, lanes); run<3, false>(out, cyc, lanes); run<4, false>(out, cyc, lanes);
    run<6, false>(out, cyc, lanes); run<8, false>(out, cyc, lanes); run<1, true>(out, cyc, lanes); run<4, true>(out, cyc, lanes);
  }
  return 0;
}
