// Calibration of the FETCH_SIZE counter on gfx950 for the access widths the ISP kernels use: every kernel reads a
// known number of bytes once, wave-contiguously (lane l reads bytes [W l, W l + W) of each wave chunk).
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d OUT -- ./scratch/fetch_calib
// factor(width) = bytes read / (FETCH_SIZE * 1024); MI355X_MICROARCH.md states 2 for 16-byte-per-lane streams.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u3 __attribute__((ext_vector_type(3)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
template <int WIDTH>
__global__ __launch_bounds__(256) void read_kernel(const uint8_t* src, uint32_t* sink, size_t n_units) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src), 0, 0x7FFFFFFF, 0x00020000);
  uint32_t acc = 0;
  for (size_t u = (size_t)blockIdx.x * 256 + threadIdx.x; u < n_units; u += (size_t)gridDim.x * 256) {
    const uint32_t off = (uint32_t)(u * WIDTH);
    if (WIDTH == 12) { const u3 v = __builtin_amdgcn_raw_buffer_load_b96(r, off, 0, 0); acc += v.x ^ v.y ^ v.z; }
    if (WIDTH == 16) { const u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0); acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (WIDTH == 4) acc += __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0);
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void read12_kernel(const uint8_t*, uint32_t*, size_t) {}
template <int WIDTH> void run(const uint8_t* src, uint32_t* sink, size_t bytes) {
  hipLaunchKernelGGL((read_kernel<WIDTH>), dim3(4096), dim3(256), 0, 0, src, sink, bytes / WIDTH);
  (void)hipDeviceSynchronize();
  printf("width %2d: %zu bytes read\n", WIDTH, bytes / WIDTH * WIDTH);
}
int main() {
  const size_t bytes = (size_t)1 << 30;        // 1 GiB: beyond the 256 MiB Infinity Cache
  uint8_t* src; uint32_t* sink;
  (void)hipMalloc(&src, bytes); (void)hipMalloc(&sink, 64);
  (void)hipMemset(src, 1, bytes);
  (void)hipDeviceSynchronize();
  run<16>(src, sink, bytes); run<12>(src, sink, bytes); run<4>(src, sink, bytes);
  return 0;
}
