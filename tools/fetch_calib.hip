// fetch_calib.hip -- calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the learner kernels use (MI355X_MICROARCH.md, HBM section:
// "calibrate on a known byte count in your own access pattern"). Streams a buffer far larger than the 256 MiB Infinity Cache once with (a) one dword per lane,
// (b) 16 bytes per lane, and writes one dword per lane of a second large buffer.   hipcc --offload-arch=gfx950 -O2 -o fetch_calib tools/fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_read_b32(const float* __restrict__ x, size_t n, float* __restrict__ out) {
  float s = 0.f; for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += x[i];
  if (s == 12345.678f) out[0] = s;
}
__global__ void k_read_b128(const float4* __restrict__ x, size_t n4, float* __restrict__ out) {
  float s = 0.f; for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) { const float4 v = x[i]; s += v.x + v.y + v.z + v.w; }
  if (s == 12345.678f) out[0] = s;
}
// the learner's access pattern: 16-byte rows gathered through a shuffled order, 4 lanes per row
__global__ void k_read_rows16(const float* __restrict__ x, size_t n_rows, size_t n_reads, float* __restrict__ out) {
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < 4 * n_reads; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = ((i >> 2) * 2654435761ull + 12345ull) % n_rows;      // a fixed pseudo-random row per group of 4 lanes
    s += x[4 * r + (i & 3)]; }
  if (s == 12345.678f) out[0] = s;
}
__global__ void k_write_b32(float* __restrict__ y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = 1.0f;
}
int main() {
  const size_t n = (size_t)1 << 29;   // 2 GiB of floats
  float *x, *y, *o; if (hipMalloc(&x, 4 * n) != hipSuccess || hipMalloc(&y, 4 * n) != hipSuccess || hipMalloc(&o, 256) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(x, 0, 4 * n); hipDeviceSynchronize();
  hipLaunchKernelGGL(k_read_b32, dim3(4096), dim3(256), 0, 0, x, n, o);
  hipLaunchKernelGGL(k_read_b128, dim3(4096), dim3(256), 0, 0, (const float4*)x, n / 4, o);
  hipLaunchKernelGGL(k_read_rows16, dim3(4096), dim3(256), 0, 0, x, n / 4, (size_t)1 << 25, o);      // 32 Mi rows x 16 B = 512 MiB requested, scattered over 2 GiB
  hipLaunchKernelGGL(k_write_b32, dim3(4096), dim3(256), 0, 0, y, n);
  hipDeviceSynchronize(); printf("bytes per kernel: %zu\n", 4 * n); return 0;
}
