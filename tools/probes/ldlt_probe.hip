// Standalone check + timing of the register-resident LDL^T solve (rsparse_amd/csrc/wrmf_ldlt.h):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I rsparse_amd/csrc tools/probes/ldlt_probe.hip -o tools/probes/ldlt_probe
//   tools/probes/ldlt_probe [rank 128|64] [k actual] [matrices per workgroup]
// Random SPD systems (dense in HBM, a few hundred of them: they stay in L2 / MALL), each workgroup copies one into the
// tile layout of wrmf_ne.hip, solves, stores; the host checks every distinct system against a double Cholesky.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define LDLT_PROF
#include "wrmf_ldlt.h"

using namespace rsparse_hip::dev;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); std::exit(1); } } while (0)

template <int KP>
__global__ __launch_bounds__(256, 2) void probe_kernel(const float* A, const float* B, float* Y, int* nbad, int k, int n_distinct, int per_wg,
                                                        unsigned long long* cycles) {
  constexpr int TLD = 33, NB = KP / 32, NT = NB * (NB + 1) / 2;
  using L = Ldlt<KP, TLD>;
  constexpr int A_FLOATS = NT * 32 * TLD;
  constexpr int CH = L::FLOATS > A_FLOATS ? L::FLOATS : A_FLOATS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sCh = reinterpret_cast<float*>(smem);
  float* sA = sCh + (CH - A_FLOATS);
  float* sU = sCh + CH;
  int* sFlag = reinterpret_cast<int*>(sU + KP);
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned long long t_solve = 0;
  unsigned long long prof[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int it = 0; it < per_wg; it++) {
    const int m = (blockIdx.x * per_wg + it) % n_distinct;
    const float* Am = A + (size_t)m * k * k;
    __syncthreads();
    for (int e = tid; e < NT * 1024; e += 256) {
      const int t = e >> 10, i = (e >> 5) & 31, j = e & 31;
      const int R = t >= 6 ? 3 : (t >= 3 ? 2 : (t >= 1 ? 1 : 0)), C = t - R * (R + 1) / 2;
      const int gi = 32 * R + i, gj = 32 * C + j;
      sA[t * 32 * TLD + i * TLD + j] = (gi < k && gj < k) ? Am[(size_t)gi * k + gj] : (gi == gj ? 1.f : 0.f);
    }
    if (tid < KP) sU[tid] = tid < k ? B[(size_t)m * k + tid] : 0.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const bool bad = L::solve(sA, sCh, sU, sFlag, wv, lane, prof);
    t_solve += __builtin_amdgcn_s_memtime() - t0;
    if (bad && tid == 0) atomicAdd(nbad, 1);
    if (tid < k) Y[((size_t)blockIdx.x * per_wg + it) * k + tid] = sU[tid];
  }
  if (tid == 0) cycles[blockIdx.x] = t_solve;
  if (lane == 0)
    for (int j = 0; j < 9; j++) cycles[gridDim.x + ((size_t)blockIdx.x * 4 + wv) * 9 + j] = prof[j];
}

int main(int argc, char** argv) {
  const int KP = argc > 1 ? std::atoi(argv[1]) : 128;
  const int k = argc > 2 ? std::atoi(argv[2]) : KP;
  const int per_wg = argc > 3 ? std::atoi(argv[3]) : 64;
  const int n_distinct = 256, grid = argc > 4 ? std::atoi(argv[4]) : 512;
  std::vector<float> A((size_t)n_distinct * k * k), B((size_t)n_distinct * k);
  std::srand(7);
  auto rnd = []() { return (float)std::rand() / RAND_MAX - 0.5f; };
  std::vector<float> F((size_t)k * 40);
  for (int m = 0; m < n_distinct; m++) {
    const int nf = 8 + m % 32;
    for (auto& v : F) v = rnd();
    for (int i = 0; i < k; i++)
      for (int j = 0; j <= i; j++) {
        double s = i == j ? 0.3 : 0.0;
        for (int f = 0; f < nf; f++) s += (double)F[(size_t)i * 40 + f] * F[(size_t)j * 40 + f] * (1.0 + f);
        A[(size_t)m * k * k + (size_t)i * k + j] = A[(size_t)m * k * k + (size_t)j * k + i] = (float)s;
      }
    for (int i = 0; i < k; i++) B[(size_t)m * k + i] = rnd();
  }
  float *dA, *dB, *dY;
  int* dbad;
  unsigned long long* dcyc;
  const size_t ny = (size_t)grid * per_wg * k;
  CHECK(hipMalloc(&dA, A.size() * 4));
  CHECK(hipMalloc(&dB, B.size() * 4));
  CHECK(hipMalloc(&dY, ny * 4));
  CHECK(hipMalloc(&dbad, 4));
  CHECK(hipMalloc(&dcyc, grid * 8 * 37));
  CHECK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemset(dbad, 0, 4));
  const int KPv = KP;
  auto launch = [&]() {
    if (KPv == 128) {
      using L = Ldlt<128, 33>;
      constexpr int AF = 10 * 32 * 33, CH = L::FLOATS > AF ? L::FLOATS : AF;
      const size_t lds = (size_t)(CH + 128 + 16) * 4;
      CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(probe_kernel<128>, dim3(grid), dim3(256), lds, 0, dA, dB, dY, dbad, k, n_distinct, per_wg, dcyc);
    } else {
      using L = Ldlt<64, 33>;
      constexpr int AF = 3 * 32 * 33, CH = L::FLOATS > AF ? L::FLOATS : AF;
      const size_t lds = (size_t)(CH + 64 + 16) * 4;
      CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(probe_kernel<64>, dim3(grid), dim3(256), lds, 0, dA, dB, dY, dbad, k, n_distinct, per_wg, dcyc);
    }
    CHECK(hipGetLastError());
  };
  launch();
  CHECK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0, 0));
  launch();
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipDeviceSynchronize());
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<float> Y(ny);
  std::vector<unsigned long long> cyc(grid * 37);
  int nbad = 0;
  CHECK(hipMemcpy(Y.data(), dY, ny * 4, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(cyc.data(), dcyc, grid * 8 * 37, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(&nbad, dbad, 4, hipMemcpyDeviceToHost));
  // reference: double Cholesky of every distinct system
  double worst = 0.0;
  std::vector<double> Ld((size_t)k * k), y(k);
  for (int m = 0; m < n_distinct; m++) {
    const float* Am = &A[(size_t)m * k * k];
    for (int i = 0; i < k; i++)
      for (int j = 0; j <= i; j++) {
        double s = Am[(size_t)i * k + j];
        for (int p = 0; p < j; p++) s -= Ld[(size_t)i * k + p] * Ld[(size_t)j * k + p];
        Ld[(size_t)i * k + j] = i == j ? std::sqrt(s) : s / Ld[(size_t)j * k + j];
      }
    for (int i = 0; i < k; i++) {
      double s = B[(size_t)m * k + i];
      for (int p = 0; p < i; p++) s -= Ld[(size_t)i * k + p] * y[p];
      y[i] = s / Ld[(size_t)i * k + i];
    }
    for (int i = k - 1; i >= 0; i--) {
      double s = y[i];
      for (int p = i + 1; p < k; p++) s -= Ld[(size_t)p * k + i] * y[p];
      y[i] = s / Ld[(size_t)i * k + i];
    }
    double ny2 = 0, err = 0;
    // (workgroup 0 .. : the slot that solved system m first)
    for (size_t slot = 0; slot < (size_t)grid * per_wg; slot++)
      if ((int)(slot % n_distinct) == m) {
        for (int i = 0; i < k; i++) {
          const double dlt = Y[slot * k + i] - y[i];
          err += dlt * dlt;
          ny2 += y[i] * y[i];
        }
        break;
      }
    worst = std::fmax(worst, std::sqrt(err / ny2));
  }
  double csum = 0;
  for (int b = 0; b < grid; b++) csum += (double)cyc[b];
  const char* nm[9] = {"load", "bar0", "upd0", "fac0", "upd1", "fac1", "bar", "back", "barb"};
  for (int w = 0; w < 4; w++) {
    std::printf("  wave %d:", w);
    for (int j = 0; j < 9; j++) {
      double s = 0;
      for (int b = 0; b < grid; b++) s += (double)cyc[grid + ((size_t)b * 4 + w) * 9 + j];
      std::printf(" %s %.0f", nm[j], s / grid / per_wg);
    }
    std::printf("\n");
  }
  std::printf("rank %d (padded %d): %d solves in %.3f ms = %.2f us per solve per CU; %.0f s_memtime ticks per solve per workgroup; worst relative error %.2e; non-positive pivots %d\n",
              k, KP, grid * per_wg, ms, ms * 1e3 * 256 / (grid * per_wg), csum / grid / per_wg, worst, nbad);
  return 0;
}
