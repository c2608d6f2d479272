// Peer exchange of SyncBN statistics through device memory the ranks map from each other (one process per GPU, xGMI peer access) - the
// replacement for the ~350 host-issued all-reduces of a training step (nn.SyncBatchNorm, /root/reference/mix_label.py:76: every batch norm
// all-gathers its statistics in forward and all-reduces two sums in backward).  DESIGN.md section 6b has the protocol, its ordering
// argument and the failure behaviour; this file is the kernel both directions share.
//
// Exchange buffer of ONE rank (allocated by the caller, mapped by every peer):
//     double  slot[CSS_PEER_DEPTH][slot_doubles]     payload ring: exchange number s uses slot s % DEPTH
//     uint64  flag[CSS_PEER_DEPTH]                   s once slot s % DEPTH holds the payload of exchange s (0 = never)
// One exchange = ONE single-workgroup launch per rank, all ranks with the same sequence number s (they run the same layers in the same
// order):
//   1. publish: copy this rank's n doubles into its own slot with system-scope stores (they pass the caches of this GPU), workgroup
//      barrier, system-scope release, flag[s % DEPTH] = s;
//   2. wait: lane r of the first wave polls peer r's flag (system-scope acquire loads) until it reads >= s, bounded by `timeout` ticks of
//      the 100 MHz wall clock: on expiry status[0] = s is recorded (host reads it at the end of the step and raises) and the kernel goes on
//      with what it has - a dead peer can never hang this GPU;
//   3. sum: out[i] = sum over ranks r = 0 .. W-1 IN RANK ORDER of slot_r[i] (system-scope loads: never a stale cached line) - every rank adds
//      the same numbers in the same order, so all ranks hold bit-identical statistics;
//   4. forward only: the train-mode finalize (mean / invstd / scale / shift / running statistics) from the summed statistics, in the same launch.
// Why DEPTH = 2 would do (4 is used): a rank enters exchange s+1 only after it left exchange s (stream order), and it leaves s only after
// it has seen every peer's flag s.  So when a peer overwrites slot (s+2) % 2 = s % 2 it has passed the wait of s+1, hence this rank has
// PUBLISHED s+1, hence this rank had finished reading exchange s.
#include "common.h"
#include "launchers.h"

namespace {
__device__ __forceinline__ void sys_store(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ double sys_load(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
}

struct PeerFinalize {       // forward: what bn_finalize_kernel takes (bn.hip), per group [G][C] outputs
  const float *gamma, *beta;
  float *running_mean, *running_var, *mean, *invstd, *scale, *shift;
  double* count_out;        // [G] global rows per group (css_bn_bwd_apply's count_dev)
  float momentum, eps;
  int G, C;
};

// phase: 0 = the whole exchange; 1 = publish only; 2 = wait + sum (+ finalize) only.  (1 / 2: single-process tests that play several ranks
// one after the other on one GPU - a rank that waited for a peer's publish in the same stream would wait for ever.)
template <bool FINALIZE>
__global__ __launch_bounds__(1024) void bn_peer_exchange_kernel(const unsigned long long* __restrict__ bases, int world, int rank,
                                                                unsigned long long seq, int slot_doubles, const double* __restrict__ local, int n,
                                                                double* __restrict__ out, PeerFinalize f, int* __restrict__ status,
                                                                long long timeout, int phase) {
  extern __shared__ double tot[];                               // FINALIZE: the n summed doubles
  const int tid = threadIdx.x, slot = (int)(seq % CSS_PEER_DEPTH);
  const size_t flag_off = (size_t)CSS_PEER_DEPTH * slot_doubles;      // (in 8-byte words from a base)
  if (phase != 2) {
    double* mine = reinterpret_cast<double*>(bases[rank]) + (size_t)slot * slot_doubles;
    for (int i = tid; i < n; i += 1024) sys_store(mine + i, local[i]);
    __syncthreads();
    if (tid == 0) {
      __threadfence_system();
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(bases[rank]) + flag_off + slot, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (phase == 1) return;
  }
  if (tid < world) {
    const unsigned long long* fl = reinterpret_cast<const unsigned long long*>(bases[tid]) + flag_off + slot;
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(fl, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
      if (wall_clock64() - t0 > timeout) {
        atomicMax(status, (int)(seq & 0x7FFFFFFF));          // which exchange gave up (the host raises at the end of the step)
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
  }
  __syncthreads();
  for (int i = tid; i < n; i += 1024) {
    double s = 0.0;
    // rank order; this rank's own term comes from `local` (the very numbers it published: no round trip through its slot)
    for (int r = 0; r < world; ++r)
      s += (r == rank && phase != 2) ? local[i] : sys_load(reinterpret_cast<const double*>(bases[r]) + (size_t)slot * slot_doubles + i);
    if (FINALIZE) tot[i] = s;
    else out[i] = s;
  }
  if (!FINALIZE) return;
  __syncthreads();
  const int G = f.G, C = f.C;
  if (tid < G && f.count_out) f.count_out[tid] = tot[(size_t)G * 2 * C + tid];
  for (int c = tid; c < C; c += 1024) {
    float rm = f.running_mean ? f.running_mean[c] : 0.f, rv = f.running_var ? f.running_var[c] : 0.f;
    const float gam = f.gamma[c], bet = f.beta[c];
    for (int g = 0; g < G; ++g) {           // the running statistics take their G momentum updates in order (bn.hip: bn_finalize_one)
      const double count = tot[(size_t)G * 2 * C + g];
      const double mean = tot[(size_t)g * 2 * C + c] / count;
      double var = tot[(size_t)g * 2 * C + C + c] / count - mean * mean;
      if (var < 0) var = 0;
      const float fmean = (float)mean, fvar = (float)var;
      const float invstd = 1.0f / sqrtf(fvar + f.eps);
      f.mean[g * C + c] = fmean;
      f.invstd[g * C + c] = invstd;
      const float sc = gam * invstd;
      f.scale[g * C + c] = sc;
      f.shift[g * C + c] = bet - fmean * sc;
      if (f.running_mean) {
        const double unbiased = count > 1 ? var * count / (count - 1) : var;
        rm = (1.f - f.momentum) * rm + f.momentum * fmean;
        rv = (1.f - f.momentum) * rv + f.momentum * (float)unbiased;
      }
    }
    if (f.running_mean) { f.running_mean[c] = rm; f.running_var[c] = rv; }
  }
}
}  // namespace

size_t css_peer_buffer_bytes_(int slot_doubles) { return ((size_t)CSS_PEER_DEPTH * slot_doubles + CSS_PEER_DEPTH) * 8; }

int css_launch_bn_peer_gather(const unsigned long long* bases, int world, int rank, unsigned long long seq, int slot_doubles, const double* local, int n,
                              double* out, int* status, long long timeout, int phase, hipStream_t st) {
  if (world < 1 || world > 64 || rank < 0 || rank >= world || n < 1 || n > slot_doubles || seq == 0 || !bases || !local || !status) return CSS_ERR_ARG;
  hipLaunchKernelGGL(bn_peer_exchange_kernel<false>, dim3(1), dim3(1024), 0, st, bases, world, rank, seq, slot_doubles, local, n, out, PeerFinalize{},
                     status, timeout, phase);
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}
int css_launch_bn_peer_finalize(const unsigned long long* bases, int world, int rank, unsigned long long seq, int slot_doubles, const double* local, int G,
                                int C, const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                                float* mean, float* invstd, float* scale, float* shift, double* count_out, int* status, long long timeout, int phase,
                                hipStream_t st) {
  const int n = G * 2 * C + G;
  if (world < 1 || world > 64 || rank < 0 || rank >= world || G < 1 || G > 1024 || n > slot_doubles || seq == 0 || !bases || !local || !status)
    return CSS_ERR_ARG;
  if ((size_t)n * 8 > 96 * 1024) return CSS_ERR_ARG;           // (summed statistics in LDS: G * (2 C + 1) doubles; 2 x 2048 channels = 64 KiB)
  PeerFinalize f{gamma, beta, running_mean, running_var, mean, invstd, scale, shift, count_out, momentum, eps, G, C};
  hipLaunchKernelGGL(bn_peer_exchange_kernel<true>, dim3(1), dim3(1024), (size_t)n * 8, st, bases, world, rank, seq, slot_doubles, local, n,
                     (double*)nullptr, f, status, timeout, phase);
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}
