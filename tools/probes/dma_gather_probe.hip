// Micro-benchmark: random gather of 512-byte vectors from a multi-GB array, per CU, through
//   mode 0: global_load_dwordx4 into registers      mode 1: global_load_lds_dwordx4 (LDS-DMA, M0 per instruction)
// W waves per workgroup (1 workgroup per CU), each wave keeps `depth` instructions (1 KB each = 2 vectors) in flight.
// Build: hipcc --offload-arch=gfx950 -O3 dma_gather_probe.hip -o dma_gather_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ void dma16(const void* g, unsigned lds_base) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g), "s"(lds_base) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// idx: per (workgroup, wave, iteration, instruction) two vector ids (lanes 0-31 -> first, 32-63 -> second)
template <int MODE, int DEPTH>
__global__ __launch_bounds__(256, 1) void probe(const float* __restrict__ X, const int* __restrict__ idx, int iters,
                                                float* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = blockDim.x >> 6;
  const int* my = idx + ((size_t)blockIdx.x * nw + wv) * (size_t)iters * 2;
  const unsigned base = (unsigned)(uintptr_t)(smem) + wv * DEPTH * 1024;
  float4 acc = make_float4(0, 0, 0, 0);
  if constexpr (MODE == 1) {
    for (int i = 0; i < DEPTH - 1; i++) {
      const int id = my[2 * i + (lane >> 5)];
      dma16(X + (size_t)id * 128 + (lane & 31) * 4, base + (i % DEPTH) * 1024);
    }
    for (int i = 0; i < iters; i++) {
      wait_vm<DEPTH - 2>();
      const int j = i + DEPTH - 1;
      const int id = my[2 * (j < iters ? j : i) + (lane >> 5)];
      dma16(X + (size_t)id * 128 + (lane & 31) * 4, base + (j % DEPTH) * 1024);
      const float4 v = *reinterpret_cast<const float4*>(smem + wv * DEPTH * 1024 + (i % DEPTH) * 1024 + lane * 16);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    wait_vm<0>();
  } else {
    // register gather: batches of DEPTH independent loads, then consume (the pattern of the round-1 kernels)
    for (int i = 0; i < iters; i += DEPTH) {
      float4 v[DEPTH];
#pragma unroll
      for (int u = 0; u < DEPTH; u++) {
        const int id = my[2 * min(i + u, iters - 1) + (lane >> 5)];
        v[u] = *reinterpret_cast<const float4*>(X + (size_t)id * 128 + (lane & 31) * 4);
      }
#pragma unroll
      for (int u = 0; u < DEPTH; u++) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}

template <int MODE, int DEPTH>
void run(const float* X, const int* idx, int waves, int iters, float* sink, size_t n_vec, const char* label) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int lds = waves * DEPTH * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (int rep = 0; rep < 2; rep++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe<MODE, DEPTH>), dim3(256), dim3(waves * 64), MODE == 1 ? lds : 16, 0, X, idx, iters, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep == 1) {
      const double bytes = 256.0 * waves * iters * 1024.0;
      printf("%-8s mode %d waves/CU %d depth %2d: %.3f ms  %.2f TB/s  %.1f GB/s/CU  (Little: %.1f KB/CU in flight -> latency %.2f us)\n", label, MODE,
             waves, DEPTH, ms, bytes / ms / 1e9, bytes / ms / 1e6 / 256, waves * (DEPTH - 1) * 1.0, waves * (DEPTH - 1) * 1024.0 / (bytes / ms / 1e3 / 256) * 1e3 / 1e3);
    }
  }
}

int main(int argc, char** argv) {
  const size_t n_vec = argc > 1 ? atoll(argv[1]) : 10000000;   // 512 B each
  const int iters = 4096, max_waves = 8;
  float* X; CK(hipMalloc(&X, n_vec * 512)); CK(hipMemset(X, 0, n_vec * 512));
  std::vector<int> h((size_t)256 * max_waves * iters * 2);
  unsigned long long s = 88172645463325252ull;
  int* idx; CK(hipMalloc(&idx, h.size() * 4));
  float* sink; CK(hipMalloc(&sink, 64));
  for (int pattern = 0; pattern < 2; pattern++) {
    for (size_t i = 0; i < h.size(); i++) {
      s ^= s << 13; s ^= s >> 7; s ^= s << 17;
      h[i] = pattern == 0 ? (int)(s % n_vec) : (int)(i % n_vec);
    }
    CK(hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const char* label = pattern == 0 ? "random" : "sequent";
    run<0, 8>(X, idx, 4, iters, sink, n_vec, label);
    run<0, 16>(X, idx, 4, iters, sink, n_vec, label);
    run<0, 16>(X, idx, 8, iters, sink, n_vec, label);
    run<0, 32>(X, idx, 8, iters, sink, n_vec, label);
    run<1, 8>(X, idx, 1, iters, sink, n_vec, label);
    run<1, 8>(X, idx, 2, iters, sink, n_vec, label);
    run<1, 16>(X, idx, 2, iters, sink, n_vec, label);
    run<1, 32>(X, idx, 2, iters, sink, n_vec, label);
    run<1, 48>(X, idx, 2, iters, sink, n_vec, label);
    run<1, 16>(X, idx, 4, iters, sink, n_vec, label);
    run<1, 32>(X, idx, 4, iters, sink, n_vec, label);
    run<1, 16>(X, idx, 8, iters, sink, n_vec, label);
  }
  return 0;
}
