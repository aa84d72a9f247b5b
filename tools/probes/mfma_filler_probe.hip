// What one vector instruction costs BESIDE the matrix pipe, one wave per SIMD (gfx950): a loop of
//     v_mfma_f32_32x32x16_f16 (accumulators a[16 t ..], four tiles in turn)  +  N fillers of one kind
// timed with s_memtime; cycles per matrix instruction for N = 0 .. 12 and a dozen filler kinds.  wrmf_cg_mf.hip's pipelined
// step measured 2778 cycles for 40 matrix instructions + ~360 others that take 641 cycles alone and 1392 for the matrix
// instructions alone (profiles/r06/r6x_*): this says which of the others are the expensive ones.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_filler_probe tools/probes/mfma_filler_probe.hip && /tmp/mfma_filler_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <utility>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__device__ __forceinline__ void filler(float (&r)[8], unsigned& h, float& d0, float& d1, const float* lds, int i, const char* gsrc = nullptr, unsigned ldsdst = 0) {
  float& x = r[i & 7];
  float& y = r[(i + 1) & 7];
  if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(x) : "v"(y));
  if constexpr (KIND == 1) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x) : "v"(y));
  // (destinations rotate over eight registers: a filler never waits for its predecessor)
  if constexpr (KIND == 2) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "+v"(x) : "v"(y), "v"(d0));
  if constexpr (KIND == 3) asm volatile("v_cvt_f32_f16 %0, %1" : "+v"(x) : "v"(h));
  if constexpr (KIND == 4) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(x) : "v"(h));
  if constexpr (KIND == 5) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(*reinterpret_cast<double*>(&r[(2 * i) & 6])) : "v"(*reinterpret_cast<double*>(&r[(2 * i + 2) & 6])));
  if constexpr (KIND == 6) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(*reinterpret_cast<double*>(&r[(2 * i) & 6])) : "v"(*reinterpret_cast<double*>(&r[(2 * i + 2) & 6])));
  if constexpr (KIND == 7) asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(d0));
  if constexpr (KIND == 8) asm volatile("ds_read_b32 %0, %1" : "+v"(x) : "v"((unsigned)(size_t)lds + 4u * (unsigned)(i & 7)));
  if constexpr (KIND == 9) asm volatile("v_and_b32 %0, %1, %2" : "+v"(x) : "v"(h), "v"(d0));
  if constexpr (KIND == 10) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x) : "v"(y));
  if constexpr (KIND == 11) asm volatile("v_readlane_b32 s20, %0, 3" :: "v"(x) : "s20");
  if constexpr (KIND == 12) asm volatile("s_nop 0");
  if constexpr (KIND == 13) asm volatile("v_accvgpr_read_b32 %0, a100" : "+v"(x));
  if constexpr (KIND == 14) { if ((i & 3) == 0) asm volatile("s_add_u32 s20, s24, 1" ::: "s20", "scc"); if ((i & 3) == 1) asm volatile("s_add_u32 s21, s24, 1" ::: "s21", "scc");
    if ((i & 3) == 2) asm volatile("s_add_u32 s22, s24, 1" ::: "s22", "scc"); if ((i & 3) == 3) asm volatile("s_add_u32 s23, s24, 1" ::: "s23", "scc"); }
  if constexpr (KIND == 16) asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "+v"(x) : "v"(d0), "v"(d1), "v"(h));
  if constexpr (KIND == 17) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(x) : "v"(d0), "v"(d1));
  if constexpr (KIND == 18) asm volatile("ds_read2_b32 %0, %1 offset1:32" : "+v"(*reinterpret_cast<double*>(&r[(2 * i) & 6])) : "v"((unsigned)(size_t)lds + 4u * (unsigned)(i & 7)));
  if constexpr (KIND == 19) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    asm volatile("ds_read_b128 %0, %1" : "+v"(*reinterpret_cast<f4*>(&r[(4 * i) & 4])) : "v"((unsigned)(size_t)lds + 16u * (unsigned)(i & 3)));
  }
  if constexpr (KIND == 15) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(*reinterpret_cast<double*>(&r[(2 * i) & 6])) : "v"(*reinterpret_cast<double*>(&r[(2 * i + 2) & 6])));
}

template <int T>
__device__ __forceinline__ void mma(const f16x8& a, const f16x8& b) {
  asm volatile("v_mfma_f32_32x32x16_f16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(a), "v"(b), "i"(16 * T), "i"(16 * T + 15) : "a63");
}
template <class F, int... I>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) { sfor_impl(f, std::make_integer_sequence<int, N>{}); }

template <int KIND, int N, bool MFMA>
__global__ __launch_bounds__(256, 1) void probe(unsigned long long* out, float* sink, int iters, int slot, const float* big) {
  __shared__ float lds[64 + 64 + 4 * 16 * 256];
  lds[threadIdx.x & 63] = (float)threadIdx.x;
  // (kind 19) an LDS-DMA piece: 1 KB per wave-instruction from a 64 MB buffer, a fresh line each time
  const char* gbase = reinterpret_cast<const char*>(big) + (size_t)((blockIdx.x * 4 + (threadIdx.x >> 6)) & 1023) * 65536 + 16 * (threadIdx.x & 63);
  const unsigned ldsdst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds + 128 + 4096 * (threadIdx.x >> 6)));
  __syncthreads();
  __attribute__((aligned(16))) float r[8];
  for (int i = 0; i < 8; i++) r[i] = sink[(threadIdx.x + i) & 63] * 1e-3f + 1.f;
  unsigned h = 0x3c003c00u;
  float d0 = 0.f, d1 = 0.f;
  f16x8 a, b;
  for (int i = 0; i < 8; i++) { a[i] = (_Float16)r[i]; b[i] = (_Float16)0.f; }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
    sfor<4>([&](auto tt) {
      if constexpr (MFMA) mma<decltype(tt)::value>(a, b);
      sfor<N>([&](auto ft) {
        constexpr int fi = decltype(tt)::value * N + decltype(ft)::value;
        filler<KIND>(r, h, d0, d1, lds, fi, gbase + (size_t)((it * 4 * N + fi) & 63) * 1024, ldsdst + 1024u * (unsigned)(fi & 15));
      });

    });
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  float s = d0 + d1 + __uint_as_float(h);
  for (int i = 0; i < 8; i++) s += r[i];
  if (s == 123.456f) sink[0] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) out[slot] = t1 - t0;
}

static unsigned long long* d_out;
static float* d_sink;
static float* d_big;
static int g_slot = 0;
constexpr int kIters = 2000;
static std::vector<std::pair<int, int>> g_meta;

template <int KIND, int N>
void run_one() {
  hipLaunchKernelGGL((probe<KIND, N, true>), dim3(256), dim3(256), 0, 0, d_out, d_sink, kIters, g_slot++, d_big);
  hipLaunchKernelGGL((probe<KIND, N, false>), dim3(256), dim3(256), 0, 0, d_out, d_sink, kIters, g_slot++, d_big);
  g_meta.push_back({KIND, N});
}
template <int KIND>
void run_kind() {
  run_one<KIND, 0>(); run_one<KIND, 2>(); run_one<KIND, 4>(); run_one<KIND, 5>(); run_one<KIND, 6>(); run_one<KIND, 8>(); run_one<KIND, 10>(); run_one<KIND, 12>();
}

int main() {
  hipMalloc(&d_out, 4096 * 8);
  hipMalloc(&d_sink, 64 * 4);
  hipMemset(d_sink, 0, 64 * 4);
  hipMalloc(&d_big, (size_t)64 << 20);
  hipMemset(d_big, 0, (size_t)64 << 20);
  const char* names[20] = {"v_fma_f32", "v_mul_f32", "v_cvt_pk_f16_f32", "v_cvt_f32_f16", "v_cvt_f32_f16_sdwa", "v_pk_mul_f32", "v_pk_fma_f32", "v_mov_b32",
                           "ds_read_b32", "v_and_b32", "v_sub_f32", "v_readlane_b32", "s_nop 0", "v_accvgpr_read", "s_add_u32", "v_pk_add_f32", "v_fma_mix_f32", "v_fma_mixlo_f16", "ds_read2_b32", "ds_read_b128"};
  for (int rep = 0; rep < 2; rep++) {   // (the first pass warms the clocks)
    g_slot = 0; g_meta.clear();
    run_kind<0>(); run_kind<1>(); run_kind<2>(); run_kind<3>(); run_kind<4>(); run_kind<5>(); run_kind<6>(); run_kind<7>();
    run_kind<8>(); run_kind<9>(); run_kind<10>(); run_kind<11>(); run_kind<12>(); run_kind<13>(); run_kind<14>(); run_kind<15>(); run_kind<16>(); run_kind<17>(); run_kind<18>(); run_kind<19>();
    hipDeviceSynchronize();
  }
  std::vector<unsigned long long> h(g_slot);
  hipMemcpy(h.data(), d_out, g_slot * 8, hipMemcpyDeviceToHost);
  printf("cycles per matrix instruction with N fillers behind it (one wave per SIMD, 256 workgroups); in brackets: the N fillers alone\n");
  printf("%-22s", "filler \\ N");
  for (int n : {0, 2, 4, 5, 6, 8, 10, 12}) printf(" %12d", n);
  printf("\n");
  for (size_t i = 0; i < g_meta.size(); i++) {
    if (g_meta[i].second == 0) printf("%-22s", names[g_meta[i].first]);
    printf(" %5.1f [%4.1f]", (double)h[2 * i] / (4.0 * kIters), (double)h[2 * i + 1] / (4.0 * kIters));
    if (g_meta[i].second == 12) printf("\n");
  }
  return 0;
}
