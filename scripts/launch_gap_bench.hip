// What does a back-to-back kernel launch cost on this stack, and what does it depend on?  (round 6: the traced bench step shows a gap of 5.8 us
// before kernels with little or no LDS and 10.4 us before kernels with >= 16 KiB of LDS - 13 ms of a 110-ms step are gaps.)
// A chain of 2000 tiny kernels on one stream, every kernel with `lds` bytes of dynamic LDS, `wg` threads, `grid` workgroups: us per launch.
//   hipcc -O3 --offload-arch=gfx950 scripts/launch_gap_bench.hip -o build/launch_gap_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void tiny(float* out, int n) {
  extern __shared__ float sm[];
  if (n < 0) sm[threadIdx.x] = 1.f;      // (never taken: the LDS allocation is what is measured)
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] += 1.f;
}
__global__ void writer(float* out, size_t n) {     // leaves dirty lines in the L2s: n floats written
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = 1.f;
}
static double chain(int lds, int wg, int grid, int reps, float* buf, int alt_lds = -1) {
  hipFuncSetAttribute((const void*)tiny, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(tiny, dim3(grid), dim3(wg), lds, 0, buf, 1);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(tiny, dim3(grid), dim3(wg), (alt_lds >= 0 && (i & 1)) ? alt_lds : lds, 0, buf, 1);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3 / reps;
}
int main() {
  float* buf;
  hipMalloc(&buf, 1 << 28);
  printf("us per launch in a chain of 2000 tiny kernels (grid 256 x 256 threads), by dynamic LDS bytes:\n");
  for (int lds : {0, 2048, 4096, 8192, 12288, 16384, 32768, 65536, 66560, 98304, 131072, 163840})
    printf("  lds %6d: %6.2f us   (alternating with lds 0: %6.2f us)\n", lds, chain(lds, 256, 256, 2000, buf), chain(lds, 256, 256, 2000, buf, 0));
  printf("by workgroup size at lds 0 / 65536: ");
  for (int wg : {64, 256, 512, 1024}) printf(" wg %d: %.2f / %.2f", wg, chain(0, wg, 256, 2000, buf), chain(65536, wg, 256, 2000, buf));
  printf("\nby grid at lds 0 / 163840 (256 threads): ");
  for (int g : {1, 256, 2048, 16384}) printf(" grid %d: %.2f / %.2f", g, chain(0, 256, g, 2000, buf), chain(163840, 256, g, 2000, buf));
  printf("\n");
  // dirty data: writer kernel (64 MB) followed by a tiny kernel, vs two tiny kernels
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (size_t mb : {1, 16, 64, 256}) {
    const size_t n = mb << 18;
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(writer, dim3(2048), dim3(256), 0, 0, buf, n);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(writer, dim3(2048), dim3(256), 0, 0, buf, n);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float a, b;
    hipEventElapsedTime(&a, e0, e1);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 200; ++i) { hipLaunchKernelGGL(writer, dim3(2048), dim3(256), 0, 0, buf, n); hipLaunchKernelGGL(tiny, dim3(256), dim3(256), 0, 0, buf, 1); }
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&b, e0, e1);
    printf("writer %3zu MB: %.2f us per launch alone (%.0f GB/s); writer + tiny pair %.2f us (the tiny kernel behind a writer costs %.2f us)\n", mb, a * 5, mb * 1.048576e-3 / (a * 5e-6),
           b * 5, (b - a) * 5);
  }
  return 0;
}
