// Issue rates of the instruction patterns the register-resident LDL^T leans on (one wave per workgroup, 256 workgroups):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/valu_rate_probe.hip -o tools/probes/valu_rate_probe
// prints s_memtime ticks per instruction (and ticks per nanosecond from the kernel's wall time).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); std::exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(64) void rate_kernel(float* out, unsigned long long* ticks, int reps) {
  float acc[32], u[4], l[4];
  for (int i = 0; i < 32; i++) acc[i] = threadIdx.x * 0.001f + i;
  for (int i = 0; i < 4; i++) { u[i] = 1.0f + 1e-3f * (threadIdx.x + i); l[i] = 1e-3f * (i + 1); }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; r++) {
#pragma unroll
    for (int e = 0; e < 4; e++)
#pragma unroll
      for (int i = 0; i < 32; i++) {
        if constexpr (MODE == 0) {          // independent v_fmac_f32
          asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i]) : "v"(u[e]), "v"(l[e]));
        } else if constexpr (MODE == 1) {   // independent v_fmac_f32_dpp row_newbcast
          asm volatile("v_fmac_f32_dpp %0, -%1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(u[e]), "v"(l[e]));
        } else if constexpr (MODE == 2) {   // dependent chain of v_fmac_f32
          asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[0]) : "v"(u[e]), "v"(l[e]));
        } else if constexpr (MODE == 3) {   // v_readlane -> v_fma reading the SGPR at once
          float s;
          asm volatile("v_readlane_b32 %0, %1, 7" : "=s"(s) : "v"(u[e]));
          asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i]) : "s"(s), "v"(l[e]));
        } else if constexpr (MODE == 4) {   // independent v_fmac_f32_dpp quad_perm (the classic DPP)
          asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(u[e]), "v"(l[e]));
        } else if constexpr (MODE == 5) {   // v_pk_fma_f32 on register pairs
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*reinterpret_cast<double*>(&acc[i & ~1])) : "v"(*reinterpret_cast<double*>(&u[e & 2])), "v"(*reinterpret_cast<double*>(&l[e & 2])));
        } else if constexpr (MODE == 6) {   // v_readlane only
          float s;
          asm volatile("v_readlane_b32 %0, %1, 7" : "=s"(s) : "v"(acc[i]));
          asm volatile("" : : "s"(s));
        }
      }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 32; i++) s += acc[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* what, float* dout, unsigned long long* dt, int waves_per_wg) {
  const int reps = 2000, grid = 256;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(rate_kernel<MODE>, dim3(grid), dim3(64), 0, 0, dout, dt, 10);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(rate_kernel<MODE>, dim3(grid * waves_per_wg), dim3(64), 0, 0, dout, dt, reps);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipDeviceSynchronize());
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> t(grid * waves_per_wg);
  CHECK(hipMemcpy(t.data(), dt, t.size() * 8, hipMemcpyDeviceToHost));
  double s = 0;
  for (auto v : t) s += (double)v;
  s /= t.size();
  const double n = (double)reps * 128;
  std::printf("%-52s %d wave(s)/CU: %6.2f ticks per instruction (pair), %.3f ticks/ns, %.2f ns per instruction\n", what, waves_per_wg, s / n, s / (ms * 1e6), ms * 1e6 / n);
}

int main() {
  float* dout;
  unsigned long long* dt;
  CHECK(hipMalloc(&dout, 256 * 16 * 64 * 4));
  CHECK(hipMalloc(&dt, 256 * 16 * 8));
  for (int w : {1, 8}) {
    run<0>("independent v_fmac_f32", dout, dt, w);
    run<1>("independent v_fmac_f32_dpp row_newbcast", dout, dt, w);
    run<4>("independent v_fmac_f32_dpp quad_perm", dout, dt, w);
    run<2>("dependent v_fmac_f32 chain", dout, dt, w);
    run<3>("v_readlane + v_fmac reading the SGPR", dout, dt, w);
    run<6>("v_readlane alone", dout, dt, w);
    run<5>("independent v_pk_fma_f32", dout, dt, w);
  }
  return 0;
}
