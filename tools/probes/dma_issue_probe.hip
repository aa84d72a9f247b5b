// Micro-benchmark 2: what does one LDS-DMA instruction cost a wave?  No compiler-visible VMEM in the loop (vector ids are
// computed, not loaded), counted vmcnt waits.  Variants of the M0 handling:
//   V = 0  save m0 / set / s_nop / dma / restore        (the recipe of wrmf_ne.hip)
//   V = 1  set / s_nop / dma, "m0" clobbered, no save/restore
//   V = 2  plain global_load_dwordx4 into registers (asm, same address stream) as the yardstick
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int V> __device__ __forceinline__ void dma16(const void* g, unsigned lds_base) {
  if constexpr (V == 0) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_base) : "memory");
  } else {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(lds_base) : "memory", "m0");
  }
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int V, int DEPTH, bool RANDOM>
__global__ __launch_bounds__(256, 1) void probe(const float* __restrict__ X, unsigned n_vec, int iters, float* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned gw = blockIdx.x * (blockDim.x >> 6) + wv;
  const unsigned base = (unsigned)(uintptr_t)(smem) + wv * DEPTH * 1024;
  float4 acc = make_float4(0, 0, 0, 0);
  auto src = [&](int i) {
    const unsigned key = (gw * 65536u + (unsigned)i) * 2u + (lane >> 5);
    const unsigned id = RANDOM ? hash32(key) % n_vec : key % n_vec;
    return X + (size_t)id * 128 + (lane & 31) * 4;
  };
  if constexpr (V < 2) {
    for (int i = 0; i < DEPTH - 1; i++) dma16<V>(src(i), base + (i % DEPTH) * 1024);
    for (int i = 0; i < iters; i++) {
      wait_vm<DEPTH - 2>();
      dma16<V>(src(i + DEPTH - 1), base + ((i + DEPTH - 1) % DEPTH) * 1024);
      const float4 v = *reinterpret_cast<const float4*>(smem + wv * DEPTH * 1024 + (i % DEPTH) * 1024 + lane * 16);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    wait_vm<0>();
  } else {
    for (int i = 0; i < iters; i += DEPTH) {
      float4 v[DEPTH];
#pragma unroll
      for (int u = 0; u < DEPTH; u++) v[u] = *reinterpret_cast<const float4*>(src(i + u));
#pragma unroll
      for (int u = 0; u < DEPTH; u++) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}

template <int V, int DEPTH, bool RANDOM>
void run(const float* X, unsigned n_vec, int waves, int iters, float* sink) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto k = probe<V, DEPTH, RANDOM>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (int rep = 0; rep < 2; rep++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(256), dim3(waves * 64), V < 2 ? waves * DEPTH * 1024 : 16, 0, X, n_vec, iters, sink);
    CK(hipGetLastError());
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep == 1) {
      const double bytes = 256.0 * waves * iters * 1024.0;
      printf("%s V%d waves/CU %d depth %2d: %7.3f ms  %5.2f TB/s  %6.1f GB/s/CU  %6.1f ns per instruction per wave\n", RANDOM ? "random " : "sequent", V,
             waves, DEPTH, ms, bytes / ms / 1e9, bytes / ms / 1e6 / 256, ms * 1e6 / iters);
    }
  }
}

int main(int argc, char** argv) {
  const unsigned n_vec = argc > 1 ? atoll(argv[1]) : 10000000;   // 512 B each
  const int iters = 8192;
  float* X; CK(hipMalloc(&X, (size_t)n_vec * 512)); CK(hipMemset(X, 0, (size_t)n_vec * 512));
  float* sink; CK(hipMalloc(&sink, 64));
  run<2, 16, true>(X, n_vec, 4, iters, sink);
  run<2, 32, true>(X, n_vec, 4, iters, sink);
  run<0, 8, true>(X, n_vec, 1, iters, sink);
  run<0, 32, true>(X, n_vec, 1, iters, sink);
  run<1, 8, true>(X, n_vec, 1, iters, sink);
  run<1, 32, true>(X, n_vec, 1, iters, sink);
  run<1, 32, false>(X, n_vec, 1, iters, sink);
  run<0, 32, true>(X, n_vec, 2, iters, sink);
  run<1, 32, true>(X, n_vec, 2, iters, sink);
  run<1, 48, true>(X, n_vec, 2, iters, sink);
  run<1, 16, true>(X, n_vec, 4, iters, sink);
  run<1, 32, true>(X, n_vec, 4, iters, sink);
  run<1, 32, false>(X, n_vec, 4, iters, sink);
  return 0;
}
