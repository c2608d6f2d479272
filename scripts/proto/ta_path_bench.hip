// PROTOTYPE micro-benchmark for the next round (not measured yet: written after this round's GPU budget was spent).
//
// Question it answers: what does ONE CU's vector-memory path carry?  Round 2 found the 256x256 conv kernels and conv_ws_kernel bound
// by it: LDS-DMA fill at 13-18 B/clk per CU inside the conv kernels, fill + stores together 23 GB/s per CU in
// conv_ws_kernel - but never measured the path by itself.  One persistent workgroup of 8 waves per CU, no MFMA, no LDS reads:
//   fill   : `buffer_load_dwordx4 ... lds` pieces of 1 KiB (8 rows x 128 B, the pieces of conv_pp64 / conv_ws) from
//            (a) a 1 MiB region every workgroup shares (L2-resident: weights), (b) a private stream per workgroup (HBM / Infinity Cache),
//            with 1, 2, 4 or 8 waves issuing and 4, 8 or 16 pieces in flight per issuing wave (counted vmcnt);
//   store  : 16-byte stores, one wave instruction = 16 pixels x 64 B (conv_ws_kernel's store) or 8 pixels x 128 B (whole lines), pixel pitch
//            512 B or 2 KiB, streaming into a private region per workgroup;
//   both   : the fill of (a) or (b) next to the stores, in the ratio 1 : 1 of conv_ws_kernel's 256 -> 1024 call.
// Output: GB/s per CU and TB/s over the chip for every combination - the ceiling the next conv design has to be drawn against.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 scripts/proto/ta_path_bench.hip -o build/ta_path_bench && ./build/ta_path_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct Args {
  const void* src; unsigned src_bytes;      // fill source
  void* dst; unsigned dst_bytes;            // store target
  int iters;                                // LDS-DMA pieces per issuing wave
  int siters;                               // store instructions per issuing wave (one pass over the workgroup's share of dst: nothing is rewritten)
  int shared_src;                           // 1: every workgroup reads the same 1 MiB; 0: private stream per workgroup
  int fill_waves;                           // waves that issue LDS-DMA (0: none)
  int store_waves;                          // waves that issue stores (0: none)
  int wide_store;                           // 0: 16 pixels x 64 B per instruction, 1: 8 pixels x 128 B
  int pitch;                                // bytes between consecutive pixels of the output
};

// DEPTH pieces in flight per issuing wave (a ring of DEPTH 1-KiB slots per wave in LDS)
template <int DEPTH>
__global__ __launch_bounds__(512) void ta_kernel(const Args a) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[8 * 16 * 1024];     // 16 KiB per wave
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.src), 0, (int)a.src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(a.dst, 0, (int)a.dst_bytes, 0x00020000);
  const bool do_fill = wave < a.fill_waves, do_store = wave >= 8 - a.store_waves;
  // fill: piece p of this wave = 8 rows of 128 B; source rows 128 B apart inside a 1-KiB block (contiguous KiB: the best case)
  const unsigned region = a.shared_src ? (1u << 20) : a.src_bytes / gridDim.x;
  const unsigned fbase = (a.shared_src ? 0u : blockIdx.x * region) + (unsigned)wave * (region / 8);
  const unsigned fwrap = region / 8;
  // stores: the output is [pixels][pitch bytes]; a workgroup writes a 512-byte stripe (its "panel") of 16 pixels per step - narrow: wave w
  // 64 B per pixel at column 64 w (conv_ws_kernel); wide: wave w 128 B per pixel at column 128 (w & 3) for pixels 8 (w >> 2) .. + 7 - and
  // the pitch / 512 workgroups of a group write the stripes of the same pixels (as the panels of conv_ws_kernel do)
  const int P = a.pitch / 512, group = blockIdx.x / P, panel = blockIdx.x % P;
  const unsigned px_per_group = (a.dst_bytes / (unsigned)a.pitch) / (gridDim.x / P);
  const int px = a.wide_store ? (wave >> 2) * 8 + (lane >> 3) : (lane & 15);
  const unsigned col = (unsigned)panel * 512u + (a.wide_store ? (unsigned)(wave & 3) * 128u + (unsigned)(lane & 7) * 16u : (unsigned)wave * 64u + (unsigned)(lane >> 4) * 16u);
  u32x4 v = {(unsigned)lane, 1u, 2u, 3u};
  unsigned char* const ring = smem + wave * 16 * 1024;
  int slot = 0;
  unsigned foff = 0, step = 0;
  const int n_it = a.iters > a.siters ? a.iters : a.siters;
  for (int it = 0; it < n_it; ++it) {
    if (do_fill && it < a.iters) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(ring + slot * 1024), 16, (int)(fbase + foff + (unsigned)lane * 16u), 0, 0, 0);
      foff += 1024;
      if (foff >= fwrap) foff = 0;
      slot = slot == DEPTH - 1 ? 0 : slot + 1;
      wait_vm<DEPTH - 1>();
    }
    if (do_store && it < a.siters) {
      const unsigned pixel = (unsigned)group * px_per_group + (step * 16u + (unsigned)px) % px_per_group;
      __builtin_amdgcn_raw_buffer_store_b128(v, rd, (int)(pixel * (unsigned)a.pitch + col), 0, 0);
      ++step;
      if (!do_fill) wait_vm<DEPTH - 1>();
    }
  }
  wait_vm<0>();
}

template <int DEPTH>
static float run(const Args& a, int n_cu) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(ta_kernel<DEPTH>, dim3(n_cu), dim3(512), 0, 0, a);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(ta_kernel<DEPTH>, dim3(n_cu), dim3(512), 0, 0, a);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 5 * 1e3f;     // us per launch
}

int main() {
  const int n_cu = 256;
  const size_t SRC = 1ull << 30, DST = 1ull << 30;      // 1 GiB each: private streams of 4 MiB per workgroup (wrapping) - larger than L2
  void *src, *dst;
  if (hipMalloc(&src, SRC) != hipSuccess || hipMalloc(&dst, DST) != hipSuccess) { printf("alloc failed\n"); return 2; }
  hipMemset(src, 1, SRC);
  hipMemset(dst, 0, DST);
  const int iters = 4096;                                // fill: 4 MiB per issuing wave
  const int siters = 512;                                // stores: 16 pixels per step, 8192 pixels per workgroup (group): one pass over dst
  printf("%-58s %10s %12s %12s\n", "configuration", "us", "GB/s per CU", "TB/s chip");
  auto report = [&](const char* name, float us, double bytes_per_cu) {
    printf("%-58s %10.1f %12.1f %12.2f\n", name, us, bytes_per_cu / (us * 1e-6) / 1e9, bytes_per_cu * n_cu / (us * 1e-6) / 1e12);
    fflush(stdout);
  };
  char name[128];
  for (int shared = 1; shared >= 0; --shared)
    for (int fw : {1, 2, 4, 8}) {
      Args a{src, (unsigned)(SRC - 1), dst, (unsigned)(DST - 1), iters, 0, shared, fw, 0, 0, 512};
      for (int depth : {4, 8, 16}) {
        const float us = depth == 4 ? run<4>(a, n_cu) : depth == 8 ? run<8>(a, n_cu) : run<16>(a, n_cu);
        snprintf(name, sizeof name, "fill %s, %d waves x %2d KiB in flight", shared ? "L2-resident 1 MiB" : "private stream (HBM)", fw, depth);
        report(name, us, (double)fw * iters * 1024);
      }
    }
  for (int wide = 0; wide < 2; ++wide)
    for (int pitch : {512, 2048})
      for (int sw : {2, 4, 8}) {
        Args a{src, (unsigned)(SRC - 1), dst, (unsigned)(DST - 1), 0, siters, 1, 0, sw, wide, pitch};
        const float us = run<16>(a, n_cu);
        snprintf(name, sizeof name, "store %s, pitch %4d B, %d waves", wide ? "8 px x 128 B" : "16 px x 64 B", pitch, sw);
        report(name, us, (double)sw * siters * 1024);
      }
  for (int shared = 1; shared >= 0; --shared)
    for (int wide = 0; wide < 2; ++wide) {
      Args a{src, (unsigned)(SRC - 1), dst, (unsigned)(DST - 1), siters, siters, shared, 4, 4, wide, 2048};
      const float us = run<8>(a, n_cu);
      snprintf(name, sizeof name, "both: fill %s (4 waves) + store %s pitch 2048 (4 waves)", shared ? "L2" : "HBM", wide ? "8x128" : "16x64");
      report(name, us, 2.0 * 4 * siters * 1024);
    }
  return 0;
}
