// What v_mfma_f32_32x32x2_f32 sustains on this box: W waves per SIMD, C independent accumulator chains per wave, no memory
// traffic.  hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_f32_rate_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int C>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float a0) {
  f32x16 acc[C];
  for (int c = 0; c < C; c++) for (int e = 0; e < 16; e++) acc[c][e] = 0.f;
  float a = a0 + threadIdx.x, b = a0 * 2.f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 16; u++)
#pragma unroll
      for (int c = 0; c < C; c++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < C; c++) for (int e = 0; e < 16; e++) s += acc[c][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int C>
void run(int wgs_per_cu, const char* what) {
  int dev = 0; hipDeviceProp_t p; hipGetDeviceProperties(&p, dev);
  const int cus = p.multiProcessorCount, grid = cus * wgs_per_cu, iters = 20000;
  float* out; hipMalloc(&out, (size_t)grid * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<C><<<grid, 256>>>(out, 100, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<C><<<grid, 256>>>(out, iters, 1.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)grid * 4 * iters * 16 * C * 4096.0;
  printf("%-44s %3d CUs  %.1f TFLOP/s  (%.1f ms)\n", what, cus, flops / ms / 1e9, ms);
  hipFree(out);
}
int main() {
  run<1>(1, "1 wave/SIMD, 1 chain");
  run<2>(1, "1 wave/SIMD, 2 chains");
  run<4>(1, "1 wave/SIMD, 4 chains");
  run<2>(2, "2 waves/SIMD, 2 chains");
  run<1>(2, "2 waves/SIMD, 1 chain");
  run<2>(4, "4 waves/SIMD, 2 chains");
  return 0;
}
