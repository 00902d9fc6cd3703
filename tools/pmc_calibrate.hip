// pmc_calibrate.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for THIS library's access width.
//
// MI355X_MICROARCH.md: FETCH_SIZE reports half the bytes of a 16-B-per-lane streaming read on gfx950, WRITE_SIZE is exact
// for 16-B-per-lane stores, and "other access widths are uncalibrated: calibrate on a known byte count in your own access
// pattern before trusting an absolute".  The step kernels move their state rows as one dword per lane (a 64-lane wave reads
// a 176-byte row), so this program streams a known 1 GiB (past the 256 MiB Infinity Cache) with dword-per-lane loads and
// stores, and with 16-B-per-lane ones as the control:
//     hipcc --offload-arch=gfx950 -O3 tools/pmc_calibrate.hip -o tools/_diag/pmc_calibrate          (here)
//     rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT -- tools/_diag/pmc_calibrate   (GPU box; then WRITE_SIZE)
// tools/pmc_calibrate.sh runs both passes and prints bytes / (counter x 1024) per kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_read_dword(const float* __restrict__ in, float* __restrict__ out, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) acc += in[i];
  if (acc == 12345.678f) out[0] = acc;   // (keeps the loads alive; never true for the zero-filled input)
}
__global__ void k_read_dwordx4(const float4* __restrict__ in, float* __restrict__ out, size_t n4) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) { const float4 v = in[i]; acc += v.x + v.y + v.z + v.w; }
  if (acc == 12345.678f) out[0] = acc;
}
__global__ void k_write_dword(float* __restrict__ out, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = 1.f;
}
__global__ void k_write_dwordx4(float4* __restrict__ out, size_t n4) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) out[i] = make_float4(1.f, 1.f, 1.f, 1.f);
}
// the step kernels' own pattern: one wave = one row of 44 floats, read and written once (rows of different waves are
// contiguous, as qpos[N][44] is)
__global__ void k_rows44(const float* __restrict__ in, float* __restrict__ out, size_t nrows) {
  const size_t row = (size_t)blockIdx.x;
  if (row < nrows && threadIdx.x < 44) out[row * 44 + threadIdx.x] = in[row * 44 + threadIdx.x] + 1.f;
}

int main() {
  const size_t bytes = (size_t)1 << 30, n = bytes / 4;
  float *a = nullptr, *b = nullptr;
  CHK(hipMalloc(&a, bytes)); CHK(hipMalloc(&b, bytes));
  CHK(hipMemset(a, 0, bytes)); CHK(hipMemset(b, 0, bytes));
  CHK(hipDeviceSynchronize());
  const int blocks = 256 * 8, threads = 256;
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(k_read_dword, dim3(blocks), dim3(threads), 0, 0, a, b, n);
    hipLaunchKernelGGL(k_read_dwordx4, dim3(blocks), dim3(threads), 0, 0, (const float4*)a, b, n / 4);
    hipLaunchKernelGGL(k_write_dword, dim3(blocks), dim3(threads), 0, 0, b, n);
    hipLaunchKernelGGL(k_write_dwordx4, dim3(blocks), dim3(threads), 0, 0, (float4*)b, n / 4);
    const size_t nrows = n / 44;
    hipLaunchKernelGGL(k_rows44, dim3((unsigned)nrows), dim3(64), 0, 0, a, b, nrows);
  }
  CHK(hipDeviceSynchronize());
  printf("bytes per kernel: %zu (rows44: %zu read + %zu written)\n", bytes, (n / 44) * 44 * 4, (n / 44) * 44 * 4);
  return 0;
}
