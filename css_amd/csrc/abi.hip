// extern "C" surface of libcss_hip.so (declared in include/css_hip.h).  Thin: argument packing, device
// selection, optional HIP-event bracketing of the hot kernels; never allocates, never synchronises
// (css_prof_read is the one documented exception: it waits for the recorded events).
#include "../../include/css_hip.h"
#include "launchers.h"

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

namespace {
inline hipStream_t S(css_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
inline void set_dev(int device) {
  if (device >= 0) (void)hipSetDevice(device);
}
int g_cu_count[64] = {0};
int cu_count(int device) {
  int d = device;
  if (d < 0) (void)hipGetDevice(&d);
  if (d < 0 || d >= 64) return 256;
  if (!g_cu_count[d]) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || v <= 0) v = 256;
    g_cu_count[d] = v;
  }
  return g_cu_count[d];
}

// ---- profiling ----
struct ProfRec { hipEvent_t a, b; int kind; double work; bool alias; };   // alias: second record over the same event pair (not pooled)
constexpr int PROF_KINDS = 16;
bool g_prof_on = false;
std::mutex g_prof_mu;
std::vector<ProfRec> g_prof;       // recorded this epoch
std::vector<ProfRec> g_prof_pool;  // reusable event pairs
constexpr size_t PROF_MAX = 1u << 18;

struct ProfScope {
  bool on;
  ProfRec r;
  hipStream_t st;
  ProfScope(int kind, double work, hipStream_t s) : on(g_prof_on), st(s) {
    if (!on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (g_prof.size() >= PROF_MAX) { on = false; return; }
    if (!g_prof_pool.empty()) { r = g_prof_pool.back(); g_prof_pool.pop_back(); }
    else if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) { on = false; return; }
    r.kind = kind;
    r.work = work;
    r.alias = false;
    (void)hipEventRecord(r.a, st);
  }
  // also file this launch under further kinds (e.g. the HBM view of a kernel that is listed under an MFMA kind)
  int n_alias = 0, alias_kind[3] = {-1, -1, -1};
  double alias_work[3] = {0, 0, 0};
  void add_alias(int kind, double work) { if (n_alias < 3) { alias_kind[n_alias] = kind; alias_work[n_alias++] = work; } }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(r.b, st);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.push_back(r);
    for (int i = 0; i < n_alias; ++i) {
      ProfRec q = r;
      q.kind = alias_kind[i]; q.work = alias_work[i]; q.alias = true;
      g_prof.push_back(q);
    }
  }
};
// one ProfScope per KERNEL launch of a convolution call: the 256x256 LDS-DMA kernels get their own kinds (5/6/7)
struct ConvProf : LaunchProf {
  int kind_other, kind_big;
  double flops;
  hipStream_t st;
  alignas(ProfScope) unsigned char buf[sizeof(ProfScope)];
  ProfScope* cur = nullptr;
  double hbm_bytes = 0;     // > 0: an HBM-bound shape class (short-K 1x1 forward): its launches are also filed under kind 12 with their bytes
  double ws_bytes = 0;      // algorithmic bytes of the call if it runs on conv_ws_kernel (kinds 13 / 14: FLOPs / bytes of those launches)
  double conv_bytes = 0;    // algorithmic bytes of the call (conv_alg_bytes): the persistent 256x256-tile launches are also filed under kind 15 with them
  ConvProf(int ko, int kb, double f, hipStream_t s) : kind_other(ko), kind_big(kb), flops(f), st(s) {}
  void begin(bool big, double share, bool ws) override {
    cur = new (buf) ProfScope(big ? kind_big : kind_other, flops * share, st);
    if (hbm_bytes > 0) cur->add_alias(12, hbm_bytes * share);
    if (ws) { cur->add_alias(13, flops * share); cur->add_alias(14, ws_bytes * share); }
    else if (big) cur->add_alias(15, conv_bytes * share);
  }
  void end() override { if (cur) { cur->~ProfScope(); cur = nullptr; } }
};
// the write-bound 1x1 class (resnet.py:131-133, conv3 of a Bottleneck: K = planes <= 512 in, 4 x planes out): algorithmic bytes of one call
double conv1x1_bytes(const ConvArgs& a, int dtype) {       // a 1x1 product read and written once (+ the addend)
  const double e = dtype == CSS_BF16 ? 2 : 4;
  return ((double)a.M * a.Ktot + (double)a.M * a.Cd * (a.addend ? 2 : 1) + (double)a.Cd * a.Ktot) * e;
}
// algorithmic HBM bytes of one convolution call: the source tensor, the weights and the output once each (+ the addend and its bit mask)
double conv_alg_bytes(const ConvArgs& a, int dtype) {
  const double e = dtype == CSS_BF16 ? 2 : 4;
  return ((double)a.N * a.Hs * a.Ws * a.Cs + (double)a.Cd * a.Ktot + (double)a.M * a.Cd * (a.addend ? 2 : 1)) * e +
         (a.add_mask ? (double)a.M * a.Cd / (dtype == CSS_BF16 ? 8 : 4) : 0.0);
}
double short_k_bytes(const ConvArgs& a, int dtype) {
  if (a.R != 1 || a.S != 1 || a.Ktot > 512 || a.Cd < 4 * a.Ktot) return 0;
  const double e = dtype == CSS_BF16 ? 2 : 4;
  return ((double)a.M * a.Ktot + (double)a.M * a.Cd + (double)a.Cd * a.Ktot) * e;
}
}  // namespace

extern "C" {

int css_abi_version(void) { return 1; }
int css_device_cu_count(int device) { return cu_count(device); }

int css_prof_enable(int on) { g_prof_on = on != 0; return 0; }
int css_prof_reset(void) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& r : g_prof) if (!r.alias) g_prof_pool.push_back(r);
  g_prof.clear();
  return 0;
}
int css_prof_read(int kind, double* total_ms, double* launches, double* alg_work) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  double ms = 0, n = 0, w = 0;
  for (auto& r : g_prof) {
    if (r.kind != kind) continue;
    if (hipEventSynchronize(r.b) != hipSuccess) return CSS_ERR_LAUNCH;
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) return CSS_ERR_LAUNCH;
    ms += t; n += 1; w += r.work;
  }
  if (total_ms) *total_ms = ms;
  if (launches) *launches = n;
  if (alg_work) *alg_work = w;
  return 0;
}

// ---- convolution ----
int css_conv2d_forward(const void* x, const void* w, const float* bias, void* y, int N, int H, int W, int Cin, int ldx, int Ho, int Wo, int Cout,
                       int ldy, int R, int Sk, int stride, int pad, int dil, double alg_flops, int dtype, int device, css_stream_t stream) {
  set_dev(device);
  ConvArgs a = {};
  a.src = x; a.wt = w; a.dst = y; a.bias = bias;
  a.N = N; a.Hs = H; a.Ws = W; a.Cs = Cin; a.lds = ldx;
  a.Hd = Ho; a.Wd = Wo; a.Cd = Cout; a.ldd = ldy;
  a.R = R; a.S = Sk; a.stride = stride; a.pad = pad; a.dil = dil; a.mode = 0;
  a.M = N * Ho * Wo; a.Ktot = R * Sk * Cin;
  ConvProf cp(0, 5, alg_flops, S(stream));
  cp.hbm_bytes = short_k_bytes(a, dtype);
  cp.ws_bytes = conv1x1_bytes(a, dtype);
  cp.conv_bytes = conv_alg_bytes(a, dtype);
  return css_launch_conv(a, dtype, cu_count(device), S(stream), g_prof_on ? &cp : nullptr);
}
int css_conv2d_forward_bnstats(const void* x, const void* w, void* y, float* stats, int Mg, int N, int H, int W, int Cin, int ldx, int Ho, int Wo,
                               int Cout, int ldy, int R, int Sk, int stride, int pad, int dil, double alg_flops, int dtype, int device,
                               css_stream_t stream) {
  set_dev(device);
  ConvArgs a = {};
  a.src = x; a.wt = w; a.dst = y; a.bias = nullptr;
  a.N = N; a.Hs = H; a.Ws = W; a.Cs = Cin; a.lds = ldx;
  a.Hd = Ho; a.Wd = Wo; a.Cd = Cout; a.ldd = ldy;
  a.R = R; a.S = Sk; a.stride = stride; a.pad = pad; a.dil = dil; a.mode = 0;
  a.M = N * Ho * Wo; a.Ktot = R * Sk * Cin;
  a.stats = stats; a.stat_Mg = Mg;
  if (!stats || Mg < 128 || a.M % Mg) return CSS_ERR_ARG;
  ConvProf cp(0, 5, alg_flops, S(stream));
  cp.hbm_bytes = short_k_bytes(a, dtype);
  cp.ws_bytes = conv1x1_bytes(a, dtype);
  cp.conv_bytes = conv_alg_bytes(a, dtype);
  return css_launch_conv(a, dtype, cu_count(device), S(stream), g_prof_on ? &cp : nullptr);
}
int css_conv2d_forward_bnstats_tile_rows(const void* x, const void* w, void* y, int N, int H, int W, int Cin, int ldx, int Ho, int Wo, int Cout,
                                         int ldy, int R, int Sk, int stride, int pad, int dil, int dtype, int device) {
  ConvArgs a = {};
  a.src = x; a.wt = w; a.dst = y; a.bias = nullptr;
  a.N = N; a.Hs = H; a.Ws = W; a.Cs = Cin; a.lds = ldx;
  a.Hd = Ho; a.Wd = Wo; a.Cd = Cout; a.ldd = ldy;
  a.R = R; a.S = Sk; a.stride = stride; a.pad = pad; a.dil = dil; a.mode = 0;
  a.M = N * Ho * Wo; a.Ktot = R * Sk * Cin;
  alignas(16) static float dummy[4];
  a.stats = dummy; a.stat_Mg = a.M;                  // (only the fact that statistics are requested matters to the plan)
  return css_conv_tile_rows_(a, dtype, cu_count(device));
}
int css_conv2d_dgrad(const void* dy, const void* w_t, void* dx, int N, int H, int W, int Cin, int lddx, int Ho, int Wo, int Cout, int lddy, int R,
                     int Sk, int stride, int pad, int dil, double alg_flops, int dtype, int device, css_stream_t stream) {
  set_dev(device);
  if (stride != 1 && stride != 2) return CSS_ERR_ARG;
  ConvArgs a = {};
  a.src = dy; a.wt = w_t; a.dst = dx; a.bias = nullptr;
  a.N = N; a.Hs = Ho; a.Ws = Wo; a.Cs = Cout; a.lds = lddy;
  a.Hd = H; a.Wd = W; a.Cd = Cin; a.ldd = lddx;
  a.R = R; a.S = Sk; a.stride = stride; a.pad = pad; a.dil = dil; a.mode = 1;
  a.M = N * H * W; a.Ktot = R * Sk * Cout;
  ConvProf cp(1, 6, alg_flops, S(stream));
  cp.ws_bytes = conv1x1_bytes(a, dtype);
  cp.conv_bytes = conv_alg_bytes(a, dtype);
  return css_launch_conv(a, dtype, cu_count(device), S(stream), g_prof_on ? &cp : nullptr);
}
static int dgrad_add_impl(const void* dy, const void* w_t, void* dx, const void* addend, int ld_add, const unsigned char* mask, int N, int H, int W, int Cin,
                          int lddx, int Ho, int Wo, int Cout, int lddy, int R, int Sk, int stride, int pad, int dil, double alg_flops, int dtype,
                          int device, css_stream_t stream) {
  set_dev(device);
  if (stride != 1 && stride != 2) return CSS_ERR_ARG;
  if (mask) {       // one mask byte per 16-byte vector of the addend: whole aligned vectors only
    const int vec = dtype == CSS_BF16 ? 8 : 4;
    if (!addend || (Cin % vec) || (lddx % vec) || (ld_add % vec) || (reinterpret_cast<uintptr_t>(dx) & 15) || (reinterpret_cast<uintptr_t>(addend) & 15))
      return CSS_ERR_ARG;
  }
  ConvArgs a = {};
  a.src = dy; a.wt = w_t; a.dst = dx; a.bias = nullptr;
  a.N = N; a.Hs = Ho; a.Ws = Wo; a.Cs = Cout; a.lds = lddy;
  a.Hd = H; a.Wd = W; a.Cd = Cin; a.ldd = lddx;
  a.R = R; a.S = Sk; a.stride = stride; a.pad = pad; a.dil = dil; a.mode = 1; a.addend = addend; a.ld_add = ld_add; a.add_mask = mask;
  a.M = N * H * W; a.Ktot = R * Sk * Cout;
  ConvProf cp(1, 6, alg_flops, S(stream));
  cp.ws_bytes = conv1x1_bytes(a, dtype);
  cp.conv_bytes = conv_alg_bytes(a, dtype);
  return css_launch_conv(a, dtype, cu_count(device), S(stream), g_prof_on ? &cp : nullptr);
}
int css_conv2d_dgrad_add(const void* dy, const void* w_t, void* dx, const void* addend, int ld_add, int N, int H, int W, int Cin, int lddx, int Ho, int Wo, int Cout, int lddy, int R,
                     int Sk, int stride, int pad, int dil, double alg_flops, int dtype, int device, css_stream_t stream) {
  return dgrad_add_impl(dy, w_t, dx, addend, ld_add, nullptr, N, H, W, Cin, lddx, Ho, Wo, Cout, lddy, R, Sk, stride, pad, dil, alg_flops, dtype, device, stream);
}
int css_conv2d_dgrad_add_masked(const void* dy, const void* w_t, void* dx, const void* addend, int ld_add, const unsigned char* mask, int N, int H, int W, int Cin,
                                int lddx, int Ho, int Wo, int Cout, int lddy, int R, int Sk, int stride, int pad, int dil, double alg_flops, int dtype,
                                int device, css_stream_t stream) {
  if (!mask) return CSS_ERR_ARG;
  return dgrad_add_impl(dy, w_t, dx, addend, ld_add, mask, N, H, W, Cin, lddx, Ho, Wo, Cout, lddy, R, Sk, stride, pad, dil, alg_flops, dtype, device, stream);
}
size_t css_conv2d_wgrad_ws_bytes(int M, int Ktot, int Cout, int dtype, int device) {
  return css_wgrad_ws_bytes_(M, Ktot, Cout, dtype, cu_count(device));
}
int css_conv2d_wgrad(const void* x, const void* dy, float* dw, float* ws, size_t ws_bytes, int N, int H, int W, int Cin, int ldx, int Ho, int Wo,
                     int Cout, int lddy, int R, int Sk, int stride, int pad, int dil, double alg_flops, int dtype, int device,
                     css_stream_t stream) {
  set_dev(device);
  WgradArgs a;
  a.ws = ws; a.ws_bytes = ws ? ws_bytes : 0;
  a.x = x; a.dy = dy; a.dw = dw;
  a.N = N; a.Hs = H; a.Ws = W; a.Cs = Cin; a.ldx = ldx;
  a.Hd = Ho; a.Wd = Wo; a.Cd = Cout; a.ldy = lddy;
  a.R = R; a.S = Sk; a.stride = stride; a.pad = pad; a.dil = dil;
  a.M = N * Ho * Wo; a.Ktot = R * Sk * Cin; a.m_per_split = a.M;
  ConvProf cp(2, 7, alg_flops, S(stream));
  return css_launch_wgrad(a, dtype, cu_count(device), S(stream), g_prof_on ? &cp : nullptr);
}
int css_conv_ws_applies(int M, int K, int ld_src, int N, int ld_dst, int R, int Sk, int stride, int pad, int has_stats, int has_addend, int ld_add,
                        int has_bias, int dtype, int n_cu) {
  static const bool no_dma = getenv("CSS_NO_DMA_CONV") != nullptr, no_256 = getenv("CSS_NO_DMA256_CONV") != nullptr;
  if (dtype != CSS_BF16 || no_dma || no_256) return 0;
  alignas(16) static float dummy[4];
  ConvArgs a = {};
  a.N = 1; a.Hs = a.Hd = 1; a.Ws = a.Wd = M; a.M = M;
  a.Cs = K; a.lds = ld_src; a.Cd = N; a.ldd = ld_dst; a.Ktot = R * Sk * K;
  a.R = R; a.S = Sk; a.stride = stride; a.pad = pad; a.dil = 1;
  a.stats = has_stats ? dummy : nullptr;
  a.addend = has_addend ? dummy : nullptr; a.ld_add = ld_add;
  a.bias = has_bias ? dummy : nullptr;
  return css_conv_ws_supported(a, n_cu > 0 ? n_cu : 256) ? 1 : 0;
}
int css_conv_c64_applies(int N, int H, int W, int Cin, int ld_src, int Cout, int ld_dst, int R, int Sk, int stride, int pad, int dil, int has_addend,
                         int has_bias, int dtype) {
  static const bool no_dma = getenv("CSS_NO_DMA_CONV") != nullptr;
  if (no_dma) return 0;
  alignas(16) static float dummy[4];
  ConvArgs a = {};
  a.N = N; a.Hs = a.Hd = H; a.Ws = a.Wd = W; a.M = N * H * W;
  a.Cs = Cin; a.lds = ld_src; a.Cd = Cout; a.ldd = ld_dst; a.Ktot = R * Sk * Cin;
  a.R = R; a.S = Sk; a.stride = stride; a.pad = pad; a.dil = dil;
  a.src = a.wt = dummy; a.dst = dummy;
  a.addend = has_addend ? dummy : nullptr;
  a.bias = has_bias ? dummy : nullptr;
  return css_conv_c64_supported(a, dtype) ? 1 : 0;
}
int css_conv_c64_set_enabled(int on) { return css_conv_c64_set_enabled_(on); }
int css_wgrad_splits(int M, int Ktot, int Cout, int dtype, int n_cu) {
  int splits = 0, mps = 0;
  css_wgrad_plan_(M, Ktot, Cout, dtype, n_cu > 0 ? n_cu : 256, &splits, &mps);
  return splits;
}
int css_weight_layout(const float* w, void* out, int Cout, int taps, int Cin, int CinPad, int dgrad, int dtype, int device, css_stream_t stream) {
  set_dev(device);
  return css_launch_weight_layout(w, out, Cout, taps, Cin, CinPad, dgrad, dtype, S(stream));
}

int css_weight_dgrad_layout_batched(const float* flat, void* out, const long* desc, int n_layers, long total_tiles, int dtype, int device,
                                    css_stream_t stream) {
  set_dev(device);
  return css_launch_weight_dgrad_layout_batched(flat, out, desc, n_layers, total_tiles, dtype, S(stream));
}

// ---- batch norm ----
int css_bn_nrb(int Mg, int G, int C, int dtype) { return css_bn_nrb_(Mg, G, C, dtype); }
int css_bn_stats(const void* y, int Mg, int G, int C, int ld, double* partial, int dtype, int device, css_stream_t stream) {
  set_dev(device);
  return css_launch_bn_stats(y, Mg, G, C, ld, partial, dtype, S(stream));
}
int css_bn_reduce(const double* partial, int nrb, int C, int G, double* sums, float* dgamma, float* dbeta, int accumulate, double count_local,
                  int device, css_stream_t stream) {
  set_dev(device);
  return css_launch_bn_reduce(partial, nrb, C, G, sums, dgamma, dbeta, accumulate, count_local, S(stream));
}
int css_bn_reduce_finalize(const double* partial, int nrb, int G, double count, const float* gamma, const float* beta, float* running_mean,
                           float* running_var, float momentum, float eps, float* mean, float* invstd, float* scale, float* shift, int C,
                           int device, css_stream_t stream) {
  set_dev(device);
  return css_launch_bn_reduce_finalize(partial, nrb, G, count, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift,
                                       C, S(stream));
}
int css_bn_reduce_finalize_slabs(const float* partial, int M, int Mg, int G, double count, const float* gamma, const float* beta,
                                 float* running_mean, float* running_var, float momentum, float eps, float* mean, float* invstd, float* scale,
                                 float* shift, double* sums_out, int C, const void* y, int ldy, int tile_rows, int device, css_stream_t stream) {
  set_dev(device);
  return css_launch_bn_reduce_slabs(partial, M, Mg, G, count, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift,
                                    sums_out, C, y, ldy, tile_rows, S(stream));
}
int css_bn_finalize(const double* sums, int G, double count, const double* count_dev, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, float momentum, float eps, float* mean, float* invstd, float* scale, float* shift, int C, int device,
                    css_stream_t stream) {
  set_dev(device);
  return css_launch_bn_finalize(sums, G, count, count_dev, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift, C, S(stream));
}
int css_peer_alloc(size_t bytes, int device, void** out, int* mem_kind_out) {
  if (!out || bytes == 0) return CSS_ERR_ARG;
  set_dev(device);
  void* p = nullptr;
  int kind = 1;
  if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) != hipSuccess || !p) {
    (void)hipGetLastError();
    kind = 2;
    p = nullptr;
    if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) != hipSuccess || !p) {
      (void)hipGetLastError();
      return CSS_ERR_WORKSPACE;
    }
  }
  if (hipMemset(p, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    (void)hipFree(p);
    return CSS_ERR_LAUNCH;
  }
  *out = p;
  if (mem_kind_out) *mem_kind_out = kind;
  return CSS_OK;
}
int css_peer_free(void* p, int device) {
  set_dev(device);
  return (!p || hipFree(p) == hipSuccess) ? CSS_OK : CSS_ERR_ARG;
}
int css_peer_ipc_export(void* p, int device, unsigned char* handle64) {
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
  if (!p || !handle64) return CSS_ERR_ARG;
  set_dev(device);
  hipIpcMemHandle_t h;
  if (hipIpcGetMemHandle(&h, p) != hipSuccess) { (void)hipGetLastError(); return CSS_ERR_ARG; }
  memcpy(handle64, &h, 64);
  return CSS_OK;
}
int css_peer_ipc_open(const unsigned char* handle64, int device, void** out) {
  if (!handle64 || !out) return CSS_ERR_ARG;
  set_dev(device);
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, 64);
  void* p = nullptr;
  if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess || !p) { (void)hipGetLastError(); return CSS_ERR_ARG; }
  *out = p;
  return CSS_OK;
}
int css_peer_ipc_close(void* p, int device) {
  set_dev(device);
  return (!p || hipIpcCloseMemHandle(p) == hipSuccess) ? CSS_OK : CSS_ERR_ARG;
}
size_t css_peer_buffer_bytes(int slot_doubles) { return css_peer_buffer_bytes_(slot_doubles); }
int css_bn_peer_finalize(const unsigned long long* bases, int world, int rank, unsigned long long seq, int slot_doubles, const double* local, int G, int C,
                         const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum, float eps, float* mean,
                         float* invstd, float* scale, float* shift, double* count_out, int* status, long timeout_ticks, int phase, int device,
                         css_stream_t stream) {
  set_dev(device);
  return css_launch_bn_peer_finalize(bases, world, rank, seq, slot_doubles, local, G, C, gamma, beta, running_mean, running_var, momentum, eps, mean,
                                     invstd, scale, shift, count_out, status, timeout_ticks, phase, S(stream));
}
int css_bn_peer_gather(const unsigned long long* bases, int world, int rank, unsigned long long seq, int slot_doubles, const double* local, int n,
                       double* out, int* status, long timeout_ticks, int phase, int device, css_stream_t stream) {
  set_dev(device);
  return css_launch_bn_peer_gather(bases, world, rank, seq, slot_doubles, local, n, out, status, timeout_ticks, phase, S(stream));
}
int css_bn_eval_coeff(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps, float* scale,
                      float* shift, int C, int device, css_stream_t stream) {
  set_dev(device);
  return css_launch_bn_eval_coeff(gamma, beta, running_mean, running_var, eps, scale, shift, C, S(stream));
}
int css_bn_apply(const void* y, int ldy, const void* res, int ldr, void* out, int ldo, const float* scale, const float* shift, int M, int C, int relu,
                 int Mg, int dtype, int device, css_stream_t stream) {
  set_dev(device);
  ProfScope ps(8, (double)M * C * (dtype == CSS_BF16 ? 2 : 4) * (2 + (res ? 1 : 0)), S(stream));
  return css_launch_bn_apply(y, ldy, res, ldr, out, ldo, scale, shift, M, C, relu, Mg, nullptr, dtype, S(stream));
}
int css_bn_apply_mask(const void* y, int ldy, const void* res, int ldr, void* out, int ldo, const float* scale, const float* shift, int M, int C,
                      int relu, int Mg, unsigned char* mask, int dtype, int device, css_stream_t stream) {
  set_dev(device);
  ProfScope ps(8, (double)M * C * (dtype == CSS_BF16 ? 2 : 4) * (2 + (res ? 1 : 0)) + (mask ? (double)M * C / 8 : 0), S(stream));
  return css_launch_bn_apply(y, ldy, res, ldr, out, ldo, scale, shift, M, C, relu, Mg, mask, dtype, S(stream));
}
int css_bn_apply_maxpool(const void* y, void* out, uint8_t* argmax, const float* scale, const float* shift, int N, int H, int W, int C, int Ho, int Wo,
                         int G, int relu, int dtype, int device, css_stream_t stream) {
  set_dev(device);
  const double e = dtype == CSS_BF16 ? 2 : 4;
  ProfScope ps(8, (double)N * H * W * C * e + (double)N * Ho * Wo * C * (e + (argmax ? 1 : 0)), S(stream));
  return css_launch_bn_apply_pool(y, out, argmax, scale, shift, N, H, W, C, Ho, Wo, G, relu, dtype, S(stream));
}
int css_bn_bwd_reduce(const void* da, int ldda, const void* a, int lda, const void* y, int ldy, const float* mean, const float* invstd,
                      const float* scale, const float* shift, int Mg, int G, int C, int relu, double* partial, int dtype, int device,
                      css_stream_t stream) {
  set_dev(device);
  ProfScope ps(10, (double)Mg * G * C * (dtype == CSS_BF16 ? 2 : 4) * (2 + (a ? 1 : 0)), S(stream));
  return css_launch_bn_bwd_reduce(da, ldda, a, lda, y, ldy, mean, invstd, scale, shift, Mg, G, C, relu, partial, nullptr, dtype, S(stream));
}
int css_bn_bwd_reduce_mask(const void* da, int ldda, const unsigned char* mask, const void* y, int ldy, const float* mean, const float* invstd,
                           int Mg, int G, int C, double* partial, int dtype, int device, css_stream_t stream) {
  set_dev(device);
  if (!mask) return CSS_ERR_ARG;
  ProfScope ps(10, (double)Mg * G * C * (dtype == CSS_BF16 ? 2 : 4) * 2 + (double)Mg * G * C / 8, S(stream));
  return css_launch_bn_bwd_reduce(da, ldda, nullptr, 0, y, ldy, mean, invstd, nullptr, nullptr, Mg, G, C, 1, partial, mask, dtype, S(stream));
}
int css_bn_bwd_apply(const void* da, int ldda, const void* a, int lda, const void* y, int ldy, void* dy, int lddy, void* dres, int lddr,
                     const float* mean, const float* invstd, const float* gamma, const double* sums, const float* scale, const float* shift,
                     double count, const double* count_dev, int M, int C, int relu, int Mg, int dtype, int device, css_stream_t stream) {
  set_dev(device);
  ProfScope ps(9, (double)M * C * (dtype == CSS_BF16 ? 2 : 4) * (3 + (a ? 1 : 0) + (dres ? 1 : 0)), S(stream));
  return css_launch_bn_bwd_apply(da, ldda, a, lda, y, ldy, dy, lddy, dres, lddr, mean, invstd, gamma, sums, scale, shift, count, count_dev, M, C, relu, Mg,
                                 nullptr, dtype, S(stream));
}
int css_bn_bwd_apply_mask(const void* da, int ldda, const unsigned char* mask, const void* y, int ldy, void* dy, int lddy, void* dres, int lddr,
                          const float* mean, const float* invstd, const float* gamma, const double* sums, double count, const double* count_dev,
                          int M, int C, int Mg, int dtype, int device, css_stream_t stream) {
  set_dev(device);
  if (!mask) return CSS_ERR_ARG;
  ProfScope ps(9, (double)M * C * (dtype == CSS_BF16 ? 2 : 4) * (3 + (dres ? 1 : 0)) + (double)M * C / 8, S(stream));
  return css_launch_bn_bwd_apply(da, ldda, nullptr, 0, y, ldy, dy, lddy, dres, lddr, mean, invstd, gamma, sums, nullptr, nullptr, count, count_dev, M, C, 1,
                                 Mg, mask, dtype, S(stream));
}

// ---- pooling / resize / concat ----
int css_maxpool_fwd(const void* x, void* out, uint8_t* argmax, int N, int H, int W, int C, int Ho, int Wo, int ks, int stride, int pad, int dtype,
                    int device, css_stream_t stream) {
  set_dev(device);
  return css_launch_maxpool_fwd(x, out, argmax, N, H, W, C, Ho, Wo, ks, stride, pad, dtype, S(stream));
}
int css_maxpool_bwd(const void* dout, const uint8_t* argmax, void* dx, int N, int H, int W, int C, int Ho, int Wo, int ks, int stride, int pad,
                    int dtype, int device, css_stream_t stream) {
  set_dev(device);
  return css_launch_maxpool_bwd(dout, argmax, dx, N, H, W, C, Ho, Wo, ks, stride, pad, dtype, S(stream));
}
int css_bilinear(const void* x, int ldx, void* out, int ldo, int N, int Hs, int Ws, int C, int Hd, int Wd, int dtype_in, int dtype_out, int backward,
                 int device, css_stream_t stream) {
  set_dev(device);
  return css_launch_bilinear(x, ldx, out, ldo, N, Hs, Ws, C, Hd, Wd, dtype_in, dtype_out, backward, S(stream));
}
int css_spatial_sum(const void* x, int ldx, void* out, int N, int HW, int C, float scale, int dtype, int device, css_stream_t stream) {
  set_dev(device);
  return css_launch_spatial_sum(x, ldx, out, N, HW, C, scale, dtype, S(stream));
}
int css_spatial_bcast(const void* x, void* out, int ldo, int N, int HW, int C, float scale, int dtype, int device, css_stream_t stream) {
  set_dev(device);
  return css_launch_spatial_bcast(x, out, ldo, N, HW, C, scale, dtype, S(stream));
}
int css_copy_channels(const void* src, int lds, void* dst, int ldd, long M, int C, int dtype_in, int dtype_out, int device, css_stream_t stream) {
  set_dev(device);
  return css_launch_copy_channels(src, lds, dst, ldd, M, C, dtype_in, dtype_out, S(stream));
}
size_t css_colsum_ws_bytes(long M, int C) { return css_colsum_ws_bytes_(M, C); }
int css_colsum(const void* x, int ld, long M, int C, float* out, float* ws, int dtype, int device, css_stream_t stream) {
  set_dev(device);
  return css_launch_colsum(x, ld, M, C, out, ws, dtype, S(stream));
}
int css_stem_s2d_enabled(void) { return css_stem_s2d_enabled_(); }
int css_conv2d_stem_s2d_tile_rows(void) { return css_stem_s2d_tile_rows_(); }
int css_nchw_to_s2d(const float* x, void* out, int N, int C, int H, int W, int device, css_stream_t stream) {
  set_dev(device);
  return css_launch_nchw_to_s2d(x, out, N, C, H, W, S(stream));
}
int css_stem_s2d_weights(const float* w, void* out, int Cout, int R, int device, css_stream_t stream) {
  set_dev(device);
  return css_launch_stem_s2d_weights(w, out, Cout, R, S(stream));
}
int css_stem_s2d_fold_wgrad(const float* dw2, float* dw, int Cout, int R, int device, css_stream_t stream) {
  set_dev(device);
  return css_launch_stem_s2d_fold_wgrad(dw2, dw, Cout, R, S(stream));
}
int css_conv2d_stem_s2d_forward(const void* x_s2d, const void* w2, void* y, float* stats, int Mg, int N, int Hs, int Ws, int Cout, int R,
                                double alg_flops, int device, css_stream_t stream) {
  set_dev(device);
  ProfScope ps(0, alg_flops, S(stream));
  return css_launch_conv_stem_s2d(x_s2d, w2, y, stats, Mg, N, Hs, Ws, Cout, R, cu_count(device), S(stream));
}
int css_nchw_to_nhwc(const float* x, void* out, int N, int C, int HW, int Cpad, int dtype, int device, css_stream_t stream) {
  set_dev(device);
  return css_launch_nchw_to_nhwc(x, out, N, C, HW, Cpad, dtype, S(stream));
}
int css_cast(const void* x, void* out, long n, int dtype_in, int dtype_out, int device, css_stream_t stream) {
  set_dev(device);
  return css_launch_cast(x, out, n, dtype_in, dtype_out, S(stream));
}
int css_sgd_ema(float* p, const float* g, float* buf, float* ema, long n, float lr, float momentum, float wd, int first, float decay,
                float grad_scale, const float* skip_flag, int device, css_stream_t stream) {
  set_dev(device);
  ProfScope ps(11, (double)n * 28.0, S(stream));     // p, g, momentum, ema read + p, momentum, ema written: 7 x 4 bytes per element
  return css_launch_sgd_ema(p, g, buf, ema, n, lr, momentum, wd, first, decay, grad_scale, skip_flag, S(stream));
}
int css_ema(float* ema, const float* p, long n, float decay, int device, css_stream_t stream) {
  set_dev(device);
  return css_launch_ema(ema, p, n, decay, S(stream));
}

// ---- similarity / pseudo labels ----
int css_proto_normalize(const float* proto, void* out, int K, int C, int dtype, int device, css_stream_t stream) {
  set_dev(device);
  return css_launch_proto_normalize(proto, out, K, C, dtype, S(stream));
}
int css_similarity(const void* rep, int ld, const void* proto_n, float* sim, float* prob, const int* cls, uint8_t* hard, int P, int K, int C,
                   float temp, float strong_thr, int dtype, int device, css_stream_t stream) {
  set_dev(device);
  ProfScope ps(4, (double)P * C * (dtype == CSS_BF16 ? 2 : 4), S(stream));
  return css_launch_similarity(rep, ld, proto_n, sim, prob, cls, hard, P, K, C, temp, strong_thr, dtype, cu_count(device), S(stream));
}
int css_softmax_hard_flags(const void* pred, int ld, const int* cls, int P, int K, float strong_thr, uint8_t* hard, int dtype, int device,
                           css_stream_t stream) {
  set_dev(device);
  return css_launch_softmax_hard(pred, ld, cls, P, K, strong_thr, hard, dtype, S(stream));
}
int css_pseudo_label(const float* sim, const void* pred, int ldp, int B, int h, int w, int K, int H, int W, float temp, float* logits_rep,
                     int64_t* labels_rep, float* logits_cls, int64_t* labels_cls, float* pseudo, int dtype, int device, css_stream_t stream) {
  set_dev(device);
  return css_launch_pseudo_label(sim, pred, ldp, B, h, w, K, H, W, temp, logits_rep, labels_rep, logits_cls, labels_cls, pseudo, dtype, S(stream));
}
int css_aug_geom(const float* img, const float* label, const float* logits1, const float* logits2, const int* params, int* table, int maxlen, int B,
                 int H, int W, int Hc, int Wc, uint8_t* img_q, uint8_t* lab_q, uint8_t* l1_q, uint8_t* l2_q, int device, css_stream_t stream) {
  set_dev(device);
  return css_launch_aug_geom(img, label, logits1, logits2, params, table, maxlen, B, H, W, Hc, Wc, img_q, lab_q, l1_q, l2_q, S(stream));
}
int css_aug_color(uint8_t* img_q, uint8_t* tmp, const int* jp, int64_t* sums, int B, int H, int W, int any_jitter, int any_blur, int device,
                  css_stream_t stream) {
  set_dev(device);
  return css_launch_aug_color(img_q, tmp, jp, reinterpret_cast<unsigned long long*>(sums), B, H, W, any_jitter, any_blur, S(stream));
}
int css_mix_boxes(const void* self, const void* partner, void* out, const int* boxes, const int* pj, int B, int P, int H, int W, int elem_bytes, int mode,
                  long fill_bits, int device, css_stream_t stream) {
  set_dev(device);
  return css_launch_mix_boxes(self, partner, out, boxes, pj, B, P, H, W, elem_bytes, mode, (long long)fill_bits, S(stream));
}
int css_aug_finish(const uint8_t* img_q, const uint8_t* lab_q, const uint8_t* l1_q, const uint8_t* l2_q, const int* flags, int B, int Hc, int Wc,
                   float* img, int64_t* label, float* logits1, float* logits2, int device, css_stream_t stream) {
  set_dev(device);
  return css_launch_aug_finish(img_q, lab_q, l1_q, l2_q, flags, B, Hc, Wc, img, label, logits1, logits2, S(stream));
}
int css_eval_confusion(const void* pred, int ldp, const int64_t* label, int B, int h, int w, int K, int H, int W, int64_t* mat, uint8_t* argmax_out,
                       int dtype, int device, css_stream_t stream) {
  set_dev(device);
  return css_launch_eval_confusion(pred, ldp, label, B, h, w, K, H, W, mat, argmax_out, dtype, S(stream));
}
int css_confusion_bincount(const int64_t* pred, const int64_t* label, long n, int K, int64_t* mat, int device, css_stream_t stream) {
  set_dev(device);
  return css_launch_confusion_bincount(pred, label, n, K, mat, S(stream));
}
int css_class_map(const int64_t* l_lab, const int64_t* u_lab, const float* u_logits, float weak_thr, int B, int H, int W, int h, int w, int* cls,
                  int device, css_stream_t stream) {
  set_dev(device);
  return css_launch_class_map(l_lab, u_lab, u_logits, weak_thr, B, H, W, h, w, cls, S(stream));
}

// ---- cross entropy ----
int css_ce_fwd(const float* logits, const int64_t* label, const float* conf, float conf_thr, const float* keep_thr, int K, long P, int HW,
               int64_t* stats, float* gtprob_out, int device, css_stream_t stream) {
  set_dev(device);
  return css_launch_ce_fwd(logits, label, conf, conf_thr, keep_thr, K, P, HW, stats, gtprob_out, S(stream));
}
int css_ce_finalize(const int64_t* stats, int B, int mode, float* loss, float* coef, int device, css_stream_t stream) {
  set_dev(device);
  return css_launch_ce_finalize(stats, B, mode, loss, coef, S(stream));
}
int css_ce_bwd(const float* logits, const int64_t* label, const float* keep_thr, int K, long P, int HW, const float* coef, const float* gscale,
               int pos_only, float* dlogits, int device, css_stream_t stream) {
  set_dev(device);
  return css_launch_ce_bwd(logits, label, keep_thr, K, P, HW, coef, gscale, pos_only, dlogits, S(stream));
}
int css_ce_small_fwd(const void* small, int ld, int B, int h, int w, const int64_t* label, const float* conf, float conf_thr, const float* keep_thr,
                     int K, int H, int W, int64_t* stats, float* gtprob_out, int dtype, int device, css_stream_t stream) {
  set_dev(device);
  return css_launch_ce_small_fwd(small, ld, B, h, w, label, conf, conf_thr, keep_thr, K, H, W, stats, gtprob_out, dtype, S(stream));
}
int css_ce_small_bwd(const void* small, int ld, int B, int h, int w, const int64_t* label, const float* keep_thr, int K, int H, int W,
                     const float* coef, const float* gscale, int pos_only, float* dsmall, int dtype, int device, css_stream_t stream) {
  set_dev(device);
  return css_launch_ce_small_bwd(small, ld, B, h, w, label, keep_thr, K, H, W, coef, gscale, pos_only, dsmall, dtype, S(stream));
}
size_t css_ohem_state_bytes(void) { return css_ohem_state_bytes_(); }
size_t css_ohem_thr_offset(void) { return css_ohem_thr_offset_(); }
int css_ohem_threshold(const float* gtprob, long P, const int64_t* stats, int B, int min_kept, float thresh, void* state, int device,
                       css_stream_t stream) {
  set_dev(device);
  return css_launch_ohem_threshold(gtprob, P, stats, B, min_kept, thresh, state, S(stream));
}

// ---- contrastive loss ----
size_t css_contrast_meta_bytes(void) { return css_contrast_meta_bytes_(); }
int css_contrast_nchunks(int P) { return css_contrast_nchunks_(P); }
int css_contrast_classify(const float* label, const float* mask, const float* prob, long sb, long sk, long sp, long psb, long psk, long psp, int P,
                          int HW, int K, float strong_thr, int* cls, uint8_t* hard, void* meta, int device, css_stream_t stream) {
  set_dev(device);
  return css_launch_contrast_classify(label, mask, prob, sb, sk, sp, psb, psk, psp, P, HW, K, strong_thr, cls, hard, meta, S(stream));
}
size_t css_contrast_class_sums_ws_bytes(int P, int K, int C) { return css_contrast_class_sums_ws_bytes_(P, K, C); }
int css_contrast_class_sums(const void* rep, int ld, const int* cls, int P, int K, int C, double* out, float* ws, int dtype, int device,
                            css_stream_t stream) {
  set_dev(device);
  return css_launch_contrast_class_sums(rep, ld, cls, P, K, C, out, ws, dtype, S(stream));
}
int css_contrast_compact(const int* cls, const uint8_t* hard, int P, int K, int* chunkhist, int* listV, int* listH, void* meta, int device,
                         css_stream_t stream) {
  set_dev(device);
  return css_launch_contrast_compact(cls, hard, P, K, chunkhist, listV, listH, meta, S(stream));
}
int css_contrast_proto_update(float* proto, const double* sums, int K, int C, float alpha, const void* meta, int device, css_stream_t stream) {
  set_dev(device);
  return css_launch_contrast_proto_update(proto, sums, K, C, alpha, meta, S(stream));
}
int css_contrast_sample(const float* proto, int C, const void* meta, float temp, float* cdf, const int* listV, const int* listH, int Q, int N,
                        unsigned long long seed, unsigned long long offset, int* anchor_pix, int* neg_pix, int device, css_stream_t stream) {
  set_dev(device);
  return css_launch_contrast_sample(proto, C, meta, temp, cdf, listV, listH, Q, N, seed, offset, anchor_pix, neg_pix, S(stream));
}
int css_contrast_resolve(const void* meta, const int* listV, const int* listH, int Q, int N, const int* anchor_idx, const int* neg_idx,
                         int* anchor_pix, int* neg_pix, int device, css_stream_t stream) {
  set_dev(device);
  return css_launch_contrast_resolve(meta, listV, listH, Q, N, anchor_idx, neg_idx, anchor_pix, neg_pix, S(stream));
}
int css_contrast_loss(const void* rep, int ld, const float* proto, int K, int C, const void* meta, const int* anchor_pix, const int* neg_pix, int Q,
                      int N, float temp, float* loss_vq, float* gradbuf, float* loss, int dtype, int device, css_stream_t stream) {
  set_dev(device);
  ProfScope ps(3, (double)K * Q * (N + 1) * C * (dtype == CSS_BF16 ? 2 : 4), S(stream));
  return css_launch_contrast_loss(rep, ld, proto, K, C, meta, anchor_pix, neg_pix, Q, N, temp, loss_vq, gradbuf, loss, dtype, S(stream));
}
int css_contrast_scatter_grad(const float* gradbuf, const int* anchor_pix, const void* meta, int K, int Q, const float* gscale, void* drep, int ld,
                              int dtype, int device, css_stream_t stream) {
  set_dev(device);
  return css_launch_contrast_scatter_grad(gradbuf, anchor_pix, meta, K, Q, gscale, drep, ld, dtype, S(stream));
}

}  // extern "C"
