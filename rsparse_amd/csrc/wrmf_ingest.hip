// On-device ingest: the second orientation of the interaction matrix (gfx950).
//
// The reference holds both orientations for a fit: c_ui = as.csc.matrix(x) and
// c_iu = t_shallow(as.csr.matrix(c_ui)) (R/model_WRMF.R:184-191), built on the host by MatrixExtra.  Here R
// (or any caller) hands over ONE orientation; the other is produced in HBM:
//     CSC (p, i, x) of an n_rows x n_cols matrix  ->  CSC (pt, it, xt) of its n_cols x n_rows transpose,
// row indices ascending inside every output column (what dgCMatrix guarantees and the top-k exclusion search
// relies on).  It is a stable counting sort by row index:
//   1. pack      one wave per input column writes (key = row index, payload = column id | value bits)
//   2. sort      rocprim::radix_sort_pairs on the low ceil(log2 n_rows) key bits -- stable, so entries of one
//                output column keep their input order = ascending input column
//   3. unpack    payload -> it / xt; boundaries of the sorted keys -> pt (every empty output column included)
// HBM-bound integer/byte work: ~3 radix passes over 12 bytes per non-zero plus one pass each for pack / unpack.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "wrmf_internal.h"

namespace rsparse_hip {
namespace {

__global__ __launch_bounds__(256) void ingest_pack_kernel(const int32_t* __restrict__ p, const int32_t* __restrict__ i,
                                                          const float* __restrict__ x, int n_cols,
                                                          uint32_t* __restrict__ keys, uint64_t* __restrict__ payload) {
  const int lane = threadIdx.x & 63;
  const int wave = (int)((blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6);
  const int n_waves = (int)((gridDim.x * (size_t)blockDim.x) >> 6);
  for (int c = wave; c < n_cols; c += n_waves) {
    const int p1 = p[c], p2 = p[c + 1];
    for (int e = p1 + lane; e < p2; e += 64) {
      keys[e] = (uint32_t)i[e];
      payload[e] = ((uint64_t)(uint32_t)c << 32) | (uint64_t)__float_as_uint(x[e]);
    }
  }
}

__global__ __launch_bounds__(256) void ingest_unpack_kernel(const uint32_t* __restrict__ keys,
                                                            const uint64_t* __restrict__ payload, int64_t nnz, int n_rows,
                                                            int32_t* __restrict__ pt, int32_t* __restrict__ it,
                                                            float* __restrict__ xt) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e <= nnz; e += stride) {
    // pt[r] = first sorted position whose key is >= r: position e opens every column in (key[e-1], key[e]]
    const int64_t lo = e == 0 ? -1 : (int64_t)keys[e - 1];
    const int64_t hi = e == nnz ? (int64_t)n_rows : (int64_t)keys[e];
    for (int64_t r = lo + 1; r <= hi; r++) pt[r] = (int32_t)e;
    if (e < nnz) {
      const uint64_t v = payload[e];
      it[e] = (int32_t)(v >> 32);
      xt[e] = __uint_as_float((uint32_t)v);
    }
  }
}

__global__ __launch_bounds__(256) void ingest_range_check_kernel(const int32_t* __restrict__ i, int64_t nnz, int n_rows,
                                                                 int* __restrict__ bad) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int local = 0;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += stride)
    local |= (i[e] < 0 || i[e] >= n_rows) ? 1 : 0;
  if (local) atomicOr(bad, 1);
}

}  // namespace

// *bad_index = 1 when some i[e] is outside [0, n_rows) (synchronises the stream)
hipError_t check_row_indices_device(const int32_t* i, int64_t nnz, int n_rows, hipStream_t s, int* bad_index) {
  *bad_index = 0;
  if (nnz <= 0) return hipSuccess;
  int* d_bad = nullptr;
  hipError_t err = hipMalloc(&d_bad, sizeof(int));
  if (err != hipSuccess) return err;
  int host_bad = 0;
  if ((err = hipMemsetAsync(d_bad, 0, sizeof(int), s)) == hipSuccess) {
    hipLaunchKernelGGL(ingest_range_check_kernel, dim3(256 * 8), dim3(256), 0, s, i, nnz, n_rows, d_bad);
    if ((err = hipGetLastError()) == hipSuccess &&
        (err = hipMemcpyAsync(&host_bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, s)) == hipSuccess)
      err = hipStreamSynchronize(s);
  }
  (void)hipFree(d_bad);
  *bad_index = host_bad;
  return err;
}

// Returns hipSuccess, or hipErrorInvalidValue with *bad_index = 1 when a row index is outside [0, n_rows).
hipError_t transpose_csc_device(int n_rows, int n_cols, int64_t nnz, const int32_t* p, const int32_t* i, const float* x,
                                int32_t* pt, int32_t* it, float* xt, hipStream_t s, int* bad_index) {
  hipError_t err;
  *bad_index = 0;
  uint32_t *keys = nullptr, *keys_out = nullptr;
  uint64_t *pay = nullptr, *pay_out = nullptr;
  void* tmp = nullptr;
  int* d_bad = nullptr;
  auto release = [&]() {
    (void)hipFree(keys); (void)hipFree(keys_out); (void)hipFree(pay); (void)hipFree(pay_out); (void)hipFree(tmp);
    (void)hipFree(d_bad);
  };
  const size_t n = (size_t)(nnz > 0 ? nnz : 1);
#define RSP_TRY(e) do { if ((err = (e)) != hipSuccess) { release(); return err; } } while (0)
  RSP_TRY(hipMalloc(&d_bad, sizeof(int)));
  RSP_TRY(hipMemsetAsync(d_bad, 0, sizeof(int), s));
  RSP_TRY(hipMalloc(&keys, n * 4));
  RSP_TRY(hipMalloc(&keys_out, n * 4));
  RSP_TRY(hipMalloc(&pay, n * 8));
  RSP_TRY(hipMalloc(&pay_out, n * 8));
  const int grid = 256 * 8;
  if (nnz > 0) {
    hipLaunchKernelGGL(ingest_range_check_kernel, dim3(grid), dim3(256), 0, s, i, nnz, n_rows, d_bad);
    RSP_TRY(hipGetLastError());
    int host_bad = 0;
    RSP_TRY(hipMemcpyAsync(&host_bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, s));
    RSP_TRY(hipStreamSynchronize(s));
    if (host_bad) {
      *bad_index = 1;
      release();
      return hipErrorInvalidValue;
    }
    hipLaunchKernelGGL(ingest_pack_kernel, dim3(grid), dim3(256), 0, s, p, i, x, n_cols, keys, pay);
    RSP_TRY(hipGetLastError());
    int bits = 1;
    while (bits < 32 && ((int64_t)1 << bits) < (int64_t)n_rows) bits++;
    size_t tmp_bytes = 0;
    RSP_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys, keys_out, pay, pay_out, (size_t)nnz, 0, bits, s));
    RSP_TRY(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
    RSP_TRY(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, keys_out, pay, pay_out, (size_t)nnz, 0, bits, s));
  }
  hipLaunchKernelGGL(ingest_unpack_kernel, dim3(grid), dim3(256), 0, s, keys_out, pay_out, nnz, n_rows, pt, it, xt);
  RSP_TRY(hipGetLastError());
  RSP_TRY(hipStreamSynchronize(s));  // the scratch below is freed on return
#undef RSP_TRY
  release();
  return hipSuccess;
}

}  // namespace rsparse_hip
