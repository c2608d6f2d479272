// What one MI355X sustains on plain streams of the sizes the short-K 1x1 convolutions move (round 3): fill, read, copy, and the store
// pattern of conv_ws_kernel's epilogue (a wave writes 64 B of each of 16 rows at a 2 KiB pitch per instruction).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 scripts/hbm_bw_bench.hip -o build/hbm_bw_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void fill_k(u32x4* __restrict__ d, size_t n) {
  const u32x4 v = {1u, 2u, 3u, 4u};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] = v;
}
__global__ __launch_bounds__(256) void read_k(const u32x4* __restrict__ s, size_t n, u32x4* __restrict__ out) {
  u32x4 a = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a ^= s[i];
  if (a[0] == 0x12345678u) out[0] = a;
}
__global__ __launch_bounds__(256) void copy_k(const u32x4* __restrict__ s, u32x4* __restrict__ d, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] = s[i];
}
// rows of `pitch` bytes; a workgroup (8 waves) owns a 128-row x 512-byte panel tile: wave w writes bytes [64 w, 64 w + 64) of the panel
// for 128 rows, 16 rows per instruction (lane = row16 * 4 + 16-byte piece) - conv_ws_kernel's epilogue
__global__ __launch_bounds__(512) void ws_pattern_k(unsigned char* __restrict__ d, int rows, int pitch) {
  const int panels = pitch / 512, tiles = rows / 128;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const u32x4 v = {1u, 2u, 3u, 4u};
  for (int t = blockIdx.x; t < tiles * panels; t += gridDim.x) {
    const int panel = t % panels, tile = t / panels;
    unsigned char* base = d + (size_t)tile * 128 * pitch + panel * 512 + wave * 64;
#pragma unroll
    for (int i = 0; i < 8; ++i) *(u32x4*)(base + (size_t)(16 * i + (lane & 15)) * pitch + (lane >> 4) * 16) = v;
  }
}
// the same bytes with whole 512-byte panel rows per instruction pair: wave w writes rows 16 w .. 16 w + 15 of the tile, 2 rows (1 KiB) per instruction
__global__ __launch_bounds__(512) void ws_rows_k(unsigned char* __restrict__ d, int rows, int pitch) {
  const int panels = pitch / 512, tiles = rows / 128;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const u32x4 v = {1u, 2u, 3u, 4u};
  for (int t = blockIdx.x; t < tiles * panels; t += gridDim.x) {
    const int panel = t % panels, tile = t / panels;
    unsigned char* base = d + (size_t)tile * 128 * pitch + panel * 512;
#pragma unroll
    for (int i = 0; i < 8; ++i) *(u32x4*)(base + (size_t)(16 * wave + 2 * i + (lane >> 5)) * pitch + (lane & 31) * 16) = v;
  }
}
// conv_ws_kernel's store pattern with its in-order completion rule: before a tile's stores go out, all but the stores of the last DEPTH
// tiles must have completed (the kernel waits for LDS-DMA pieces that are older than those stores: vmcnt retires in order)
template <int DEPTH>
__global__ __launch_bounds__(512) void ws_depth_k(unsigned char* __restrict__ d, int rows, int pitch) {
  const int panels = pitch / 512, tiles = rows / 128;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const u32x4 v = {1u, 2u, 3u, 4u};
  for (int t = blockIdx.x; t < tiles * panels; t += gridDim.x) {
    const int panel = t % panels, tile = t / panels;
    unsigned char* base = d + (size_t)tile * 128 * pitch + panel * 512 + wave * 64;
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * DEPTH > 63 ? 63 : 8 * DEPTH) : "memory");
#pragma unroll
    for (int i = 0; i < 8; ++i) *(u32x4*)(base + (size_t)(16 * i + (lane & 15)) * pitch + (lane >> 4) * 16) = v;
  }
}
// conv_ws_kernel's store pattern AND its workgroup -> (panel, pixel-tile stream) map: the panels of a pixel tile on CUs of ONE XCD
// (blockIdx & 7 = XCD), each workgroup walking its stream of tiles
__global__ __launch_bounds__(512) void ws_map_k(unsigned char* __restrict__ d, int rows, int pitch, int same_xcd) {
  const int np = pitch / 512, tiles = rows / 128;
  const int G = gridDim.x, c8 = G >> 3, xcd = blockIdx.x & 7, idx8 = blockIdx.x >> 3, spx = c8 / np;
  int panel, stream, nstreams;
  if (same_xcd) { panel = idx8 % np; stream = xcd * spx + idx8 / np; nstreams = 8 * spx; }
  else { const int lin = xcd * c8 + idx8; panel = lin % np; stream = lin / np; nstreams = G / np; }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const u32x4 v = {1u, 2u, 3u, 4u};
  for (int tile = stream; tile < tiles; tile += nstreams) {
    unsigned char* base = d + (size_t)tile * 128 * pitch + panel * 512 + wave * 64;
#pragma unroll
    for (int i = 0; i < 8; ++i) *(u32x4*)(base + (size_t)(16 * i + (lane & 15)) * pitch + (lane >> 4) * 16) = v;
  }
}
template <typename F> static double timeit(F f, int reps = 20) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) f();
  std::vector<float> ts;
  for (int r = 0; r < reps; ++r) { hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms); }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2] * 1e-3;
}
int main() {
  const int rows = 135168, pitch = 2048;            // 277 MB: the layer-3 block output of the c2 workload (1024 channels, bf16)
  const size_t bytes = (size_t)rows * pitch, n = bytes / 16;
  unsigned char *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
  for (int blocks : {256, 1024, 2048, 8192}) {
    double t;
    t = timeit([&] { hipLaunchKernelGGL(fill_k, dim3(blocks), dim3(256), 0, 0, (u32x4*)a, n); });
    printf("fill   %5d blocks: %7.1f us  %6.2f TB/s written\n", blocks, t * 1e6, bytes / t / 1e12);
    t = timeit([&] { hipLaunchKernelGGL(read_k, dim3(blocks), dim3(256), 0, 0, (const u32x4*)a, n, (u32x4*)b); });
    printf("read   %5d blocks: %7.1f us  %6.2f TB/s read\n", blocks, t * 1e6, bytes / t / 1e12);
    t = timeit([&] { hipLaunchKernelGGL(copy_k, dim3(blocks), dim3(256), 0, 0, (const u32x4*)a, (u32x4*)b, n); });
    printf("copy   %5d blocks: %7.1f us  %6.2f TB/s read + written\n", blocks, t * 1e6, 2.0 * bytes / t / 1e12);
  }
  for (int blocks : {256, 512, 1024}) {
    double t = timeit([&] { hipLaunchKernelGGL(ws_pattern_k, dim3(blocks), dim3(512), 0, 0, a, rows, pitch); });
    printf("conv_ws store pattern (64 B x 16 rows per instruction), %4d workgroups: %7.1f us  %6.2f TB/s written\n", blocks, t * 1e6, bytes / t / 1e12);
    t = timeit([&] { hipLaunchKernelGGL(ws_rows_k, dim3(blocks), dim3(512), 0, 0, a, rows, pitch); });
    printf("whole 512-byte panel rows (2 rows per instruction),     %4d workgroups: %7.1f us  %6.2f TB/s written\n", blocks, t * 1e6, bytes / t / 1e12);
  }
  {
    double t;
    t = timeit([&] { hipLaunchKernelGGL(ws_depth_k<0>, dim3(256), dim3(512), 0, 0, a, rows, pitch); });
    printf("conv_ws store pattern, at most 0 earlier tiles' stores in flight when a tile's go out: %7.1f us  %6.2f TB/s\n", t * 1e6, bytes / t / 1e12);
    t = timeit([&] { hipLaunchKernelGGL(ws_depth_k<1>, dim3(256), dim3(512), 0, 0, a, rows, pitch); });
    printf("conv_ws store pattern, at most 1 earlier tile's:  %7.1f us  %6.2f TB/s\n", t * 1e6, bytes / t / 1e12);
    t = timeit([&] { hipLaunchKernelGGL(ws_depth_k<2>, dim3(256), dim3(512), 0, 0, a, rows, pitch); });
    printf("conv_ws store pattern, at most 2 earlier tiles': %7.1f us  %6.2f TB/s\n", t * 1e6, bytes / t / 1e12);
    t = timeit([&] { hipLaunchKernelGGL(ws_depth_k<4>, dim3(256), dim3(512), 0, 0, a, rows, pitch); });
    printf("conv_ws store pattern, at most 4 earlier tiles': %7.1f us  %6.2f TB/s\n", t * 1e6, bytes / t / 1e12);
    t = timeit([&] { hipLaunchKernelGGL(ws_depth_k<7>, dim3(256), dim3(512), 0, 0, a, rows, pitch); });
    printf("conv_ws store pattern, at most 7 earlier tiles': %7.1f us  %6.2f TB/s\n", t * 1e6, bytes / t / 1e12);
  }
  for (int pitch2 : {512, 1024, 2048, 4096}) {
    const int rows2 = (int)(bytes / pitch2) / 128 * 128;
    for (int same = 1; same >= 0; --same) {
      double t = timeit([&] { hipLaunchKernelGGL(ws_map_k, dim3(256), dim3(512), 0, 0, a, rows2, pitch2, same); });
      printf("conv_ws store pattern with conv_ws's workgroup map, row pitch %4d B (%d panels), panels of a tile on %s: %7.1f us  %6.2f TB/s\n", pitch2, pitch2 / 512,
             same ? "ONE XCD       " : "different XCDs", t * 1e6, (double)rows2 * pitch2 / t / 1e12);
    }
  }
  { double t = timeit([&] { hipMemsetAsync(a, 0, bytes, 0); }); printf("hipMemsetAsync: %7.1f us  %6.2f TB/s\n", t * 1e6, bytes / t / 1e12); }
  return 0;
}
