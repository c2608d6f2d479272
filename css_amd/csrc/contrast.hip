// Pixel-wise prototype contrastive loss (Contrast_Loss.forward, generalframeworks/loss/loss.py:75-149,
// negative_index_sampler :410-418) as a chain of device kernels with NO host synchronisation:
//
//   classify      label*mask one-hot + prob -> class id / hard flag per pixel        loss.py:80,94-99
//   class_sums    per-class sum of embeddings + counts (fp64)                          loss.py:102 (mean, after an
//                 all-reduce of [K*(C+1)] numbers replacing the two all_gathers at :77,:81)
//   proto_update  first-seen / EMA prototype update, local-presence rule              loss.py:96,103-109
//   hist/scan/scatter  stable per-class pixel lists (valid and hard)                   loss.py:111-113
//   class_probs   softmax(cos(proto_v, proto_j)/temp) over the other present classes   loss.py:133-135
//   sample        Philox: anchors, negative class, negative pixel                       loss.py:127,136-140,410-418
//   resolve       (parity tests) injected reference-style indices -> pixel ids
//   loss          gather + cosine + online logsumexp + d loss / d anchor in one pass   loss.py:141-147
//   reduce/scatter_grad  mean over queries, sum over classes, / V ; dense grad rows    loss.py:147-149
//
// HBM-bound: the loss kernel gathers Q*N embedding rows per class.
#include "common.h"

constexpr int CT_MAXK = 32;
struct ContrastMeta {
  int V;                 // locally present classes
  int present[CT_MAXK];  // class id of present index v (ascending)
  int cntV[CT_MAXK];     // per class id
  int cntH[CT_MAXK];
  int baseV[CT_MAXK];    // offsets into listV / listH per class id
  int baseH[CT_MAXK];
  int err;               // bit0: a pixel was valid for more than one class (label not one-hot)
};

// ---- 1. classify (reference-signature path; tensors may be NCHW or NHWC via strides) -------------
__global__ __launch_bounds__(256) void contrast_classify_kernel(const float* __restrict__ label, const float* __restrict__ mask,
                                                                const float* __restrict__ prob, long sb, long sk, long sp, long psb, long psk,
                                                                long psp, int P, int HW, int K, float strong_thr, int* __restrict__ cls,
                                                                uint8_t* __restrict__ hard, ContrastMeta* meta) {
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
    const int b = p / HW, s = p - b * HW;
    const float m = mask[p];
    int c = -1, n = 0;
    for (int k = 0; k < K; ++k) {
      const float v = label[b * sb + k * sk + s * sp] * m;
      if (v != 0.f) { if (c < 0) c = k; ++n; }
    }
    if (n > 1) atomicOr(&meta->err, 1);
    cls[p] = c;
    hard[p] = (c >= 0 && prob[b * psb + c * psk + s * psp] < strong_thr) ? 1 : 0;
  }
}

// ---- 2. per-class embedding sums ---------------------------------------------------------------
// out: double [K][C] sums followed by double [K] counts
template <typename T, int C>
__global__ __launch_bounds__(256) void contrast_class_sums_kernel(const T* __restrict__ rep, int ld, const int* __restrict__ cls, int P, int K,
                                                                  int pix_per_block, float* __restrict__ out) {
  // every wave owns a private [K][C] fp32 accumulator in LDS and a lane owns 4 channels of it: plain 16-byte
  // read-modify-write, no atomics (LDS float atomics made this kernel 0.63 ms; a wave's LDS operations execute in order, so
  // consecutive rows of the same class are safe)
  extern __shared__ float cs_lds[];                       // [4 waves][K*C] then [4][CT_MAXK] counts
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* acc = cs_lds + (size_t)wave * K * C;
  float* cnt = cs_lds + (size_t)4 * K * C + wave * CT_MAXK;
  for (int i = lane; i < K * C; i += 64) acc[i] = 0.f;
  if (lane < CT_MAXK) cnt[lane] = 0.f;
  __syncthreads();
  const int p0 = blockIdx.x * pix_per_block, p1 = min(P, p0 + pix_per_block);
  constexpr int EPL = C / 64, UN = 16;
  static_assert(EPL == 4, "a lane owns one float4 of the accumulator row");
  int cur = -1;
  float racc[4] = {0.f, 0.f, 0.f, 0.f}, rcnt = 0.f;
  auto flush = [&]() {
    if (cur >= 0) {
      float4* a4 = reinterpret_cast<float4*>(acc + cur * C + lane * EPL);
      float4 a = *a4;
      a.x += racc[0]; a.y += racc[1]; a.z += racc[2]; a.w += racc[3];
      *a4 = a;
      if (lane == 0) cnt[cur] += rcnt;
    }
    racc[0] = racc[1] = racc[2] = racc[3] = 0.f;
    rcnt = 0.f;
  };
  // A wave takes 64 consecutive pixels at a time: ONE load fetches their class ids (lane i: pixel pb + i; the ids reach the loop as
  // wave-uniform scalars through v_readlane), then the rows go out UN = 16 at a time - 8 KiB in flight per wave, 32 KiB per CU (the first
  // form had 8 rows and 8 same-address class loads in flight per wave and ran at 0.9 TB/s).
  for (int pb = p0 + wave * 64; pb < p1; pb += 4 * 64) {
    const int ci = pb + lane < p1 ? cls[pb + lane] : -1;
#pragma unroll
    for (int u0 = 0; u0 < 64; u0 += UN) {
      int c[UN];
      float v[UN][EPL];
#pragma unroll
      for (int u = 0; u < UN; ++u) c[u] = __builtin_amdgcn_readlane(ci, u0 + u);
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        if (c[u] >= 0) {
          const T* row = rep + (size_t)(pb + u0 + u) * ld + lane * EPL;
          if constexpr (sizeof(T) == 2) {
            union { uint2 u2; T e[EPL]; } pk;
            pk.u2 = *reinterpret_cast<const uint2*>(row);
#pragma unroll
            for (int e = 0; e < EPL; ++e) v[u][e] = (float)pk.e[e];
          } else {
#pragma unroll
            for (int e = 0; e < EPL; ++e) v[u][e] = (float)row[e];
          }
        }
      }
      // Runs of one class (labels are piecewise constant) are summed in registers and flushed into the LDS row when the class changes;
      // the class id is wave-uniform, so the branch is too; the order of the adds is fixed.
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        if (c[u] < 0) continue;
        if (c[u] != cur) {
          flush();
          cur = c[u];
        }
        racc[0] += v[u][0]; racc[1] += v[u][1]; racc[2] += v[u][2]; racc[3] += v[u][3];
        rcnt += 1.f;
      }
    }
  }
  flush();
  __syncthreads();
  // the workgroup's partial sums -> its own row of the workspace (plain stores; contrast_class_sums_reduce_kernel adds the rows up in
  // workgroup order: fp64 atomics would add in arrival order, and the prototypes - hence the sampled negatives - would not be reproducible)
  float* row = out + (size_t)blockIdx.x * (K * C + K);
  for (int i = tid; i < K * C; i += 256)
    row[i] = (cs_lds[i] + cs_lds[(size_t)K * C + i]) + (cs_lds[(size_t)2 * K * C + i] + cs_lds[(size_t)3 * K * C + i]);
  if (tid < K) {
    const float* cb = cs_lds + (size_t)4 * K * C;
    row[(size_t)K * C + tid] = (cb[tid] + cb[CT_MAXK + tid]) + (cb[2 * CT_MAXK + tid] + cb[3 * CT_MAXK + tid]);
  }
}
// out[i] = sum over the workgroups' rows, in row order, in fp64 (four independent chains for latency, combined in a fixed tree)
__global__ __launch_bounds__(256) void contrast_class_sums_reduce_kernel(const float* __restrict__ ws, int nrows, int n, double* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  int r = 0;
  for (; r + 3 < nrows; r += 4) {
    s0 += (double)ws[(size_t)(r + 0) * n + i]; s1 += (double)ws[(size_t)(r + 1) * n + i];
    s2 += (double)ws[(size_t)(r + 2) * n + i]; s3 += (double)ws[(size_t)(r + 3) * n + i];
  }
  for (; r < nrows; ++r) s0 += (double)ws[(size_t)r * n + i];
  out[i] = (s0 + s1) + (s2 + s3);
}

// ---- 3. stable compaction ----------------------------------------------------------------------
// one wave = one chunk of 64*R consecutive pixels; lane t<32 counts valid pixels of class t, lane 32+t hard ones
constexpr int CHUNK_R = 16;
__global__ __launch_bounds__(256) void contrast_hist_kernel(const int* __restrict__ cls, const uint8_t* __restrict__ hard, int P, int K,
                                                            int* __restrict__ chunkhist) {
  const int lane = threadIdx.x & 63;
  const int chunk = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int p0 = chunk * 64 * CHUNK_R;
  if (p0 >= P) return;
  int mine = 0;
  for (int r = 0; r < CHUNK_R; ++r) {
    const int p = p0 + r * 64 + lane;
    const int c = p < P ? cls[p] : -1;
    const int h = (p < P && c >= 0) ? hard[p] : 0;
    for (int k = 0; k < K; ++k) {
      const unsigned long long bv = __ballot(c == k), bh = __ballot(c == k && h);
      if (lane == k) mine += __popcll(bv);
      if (lane == 32 + k) mine += __popcll(bh);
    }
  }
  chunkhist[chunk * 64 + lane] = mine;
}
// exclusive prefix over chunks per slot (in place) + meta
__global__ __launch_bounds__(64) void contrast_scan_kernel(int* __restrict__ chunkhist, int nchunks, int K, ContrastMeta* meta) {
  const int t = threadIdx.x;
  int run = 0;
  // one wave, one slot per lane, chunks in order - but SCAN_U rows requested at once: the loop was a chain of dependent ~0.5 us loads
  // (122 us at c2, r03); the adds stay serial and in chunk order
  constexpr int SCAN_U = 16;
  for (int c0 = 0; c0 < nchunks; c0 += SCAN_U) {
    int v[SCAN_U];
#pragma unroll
    for (int u = 0; u < SCAN_U; ++u) v[u] = c0 + u < nchunks ? chunkhist[(c0 + u) * 64 + t] : 0;
#pragma unroll
    for (int u = 0; u < SCAN_U; ++u) {
      if (c0 + u < nchunks) chunkhist[(c0 + u) * 64 + t] = run;
      run += v[u];
    }
  }
  __shared__ int tot[64];
  tot[t] = run;
  __syncthreads();
  if (t == 0) {
    int bv = 0, bh = 0, V = 0;
    for (int k = 0; k < CT_MAXK; ++k) {
      const int cv = k < K ? tot[k] : 0, ch = k < K ? tot[32 + k] : 0;
      meta->cntV[k] = cv;
      meta->cntH[k] = ch;
      meta->baseV[k] = bv;
      meta->baseH[k] = bh;
      bv += cv;
      bh += ch;
      if (cv > 0) meta->present[V++] = k;
    }
    for (int v = V; v < CT_MAXK; ++v) meta->present[v] = -1;
    meta->V = V;
  }
}
__global__ __launch_bounds__(256) void contrast_scatter_kernel(const int* __restrict__ cls, const uint8_t* __restrict__ hard, int P, int K,
                                                               const int* __restrict__ chunkoff, const ContrastMeta* __restrict__ meta,
                                                               int* __restrict__ listV, int* __restrict__ listH) {
  const int lane = threadIdx.x & 63;
  const int chunk = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int p0 = chunk * 64 * CHUNK_R;
  if (p0 >= P) return;
  int run = chunkoff[chunk * 64 + lane];
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  for (int r = 0; r < CHUNK_R; ++r) {
    const int p = p0 + r * 64 + lane;
    const int c = p < P ? cls[p] : -1;
    const int h = (p < P && c >= 0) ? hard[p] : 0;
    for (int k = 0; k < K; ++k) {
      const unsigned long long bv = __ballot(c == k), bh = __ballot(c == k && h);
      const int rv = __shfl(run, k, 64), rh = __shfl(run, 32 + k, 64);
      if (c == k) {
        listV[meta->baseV[k] + rv + __popcll(bv & lt)] = p;
        if (h) listH[meta->baseH[k] + rh + __popcll(bh & lt)] = p;
      }
      if (lane == k) run += __popcll(bv);
      if (lane == 32 + k) run += __popcll(bh);
    }
  }
}

// ---- 4. prototype update (after the cross-rank all-reduce of sums/counts) ----------------------
// prototypes fp32 [K][C] updated in place like the reference (loss.py:105,108)
template <int C>
__global__ __launch_bounds__(64) void contrast_proto_update_kernel(float* __restrict__ proto, const double* __restrict__ sums, int K, float alpha,
                                                                   const ContrastMeta* __restrict__ meta) {
  const int k = blockIdx.x, lane = threadIdx.x;
  if (k >= K || meta->cntV[k] == 0) return;   // local-presence rule, loss.py:96
  const double n = sums[(size_t)K * C + k];
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += proto[(size_t)k * C + c];
  s = wave_sum(s);
  const bool first = (s == 0.f);              // loss.py:103
  for (int c = lane; c < C; c += 64) {
    const float mean = (float)(sums[(size_t)k * C + c] / n);
    float* pp = proto + (size_t)k * C + c;
    *pp = first ? mean : alpha * (*pp) + (1.f - alpha) * mean;
  }
}

// ---- 5. negative-class distribution (cdf over the other present classes in cyclic order) ---------
template <int C>
__global__ __launch_bounds__(64) void contrast_class_probs_kernel(const float* __restrict__ proto, const ContrastMeta* __restrict__ meta,
                                                                  float inv_temp, float* __restrict__ cdf /* [32][32] */) {
  const int v = blockIdx.x, lane = threadIdx.x, V = meta->V;
  if (v >= V || V < 2) return;
  const float* pv = proto + (size_t)meta->present[v] * C;
  float nv = 0.f;
  for (int c = lane; c < C; c += 64) nv += pv[c] * pv[c];
  nv = fmaxf(sqrtf(wave_sum(nv)), 1e-8f);
  float lg[CT_MAXK];
  float mx = -INFINITY;
  for (int j = 0; j < V - 1; ++j) {
    int o = v + 1 + j;
    if (o >= V) o -= V;
    const float* pj = proto + (size_t)meta->present[o] * C;
    float d = 0.f, nj = 0.f;
    for (int c = lane; c < C; c += 64) { d += pv[c] * pj[c]; nj += pj[c] * pj[c]; }
    d = wave_sum(d);
    nj = fmaxf(sqrtf(wave_sum(nj)), 1e-8f);
    lg[j] = d / (nv * nj) * inv_temp;
    mx = fmaxf(mx, lg[j]);
  }
  if (lane == 0) {
    float se = 0.f;
    for (int j = 0; j < V - 1; ++j) se += expf(lg[j] - mx);
    float run = 0.f;
    for (int j = 0; j < V - 1; ++j) {
      run += expf(lg[j] - mx) / se;
      cdf[v * 32 + j] = run;
    }
  }
}

// ---- 6. sampler ---------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned* out) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// grid: (ceil((N+1)/256), Q, 32): thread n==N draws the anchor of (v, q), threads n<N the negatives
__global__ __launch_bounds__(256) void contrast_sample_kernel(const ContrastMeta* __restrict__ meta, const float* __restrict__ cdf,
                                                              const int* __restrict__ listV, const int* __restrict__ listH, int Q, int N,
                                                              unsigned long long seed, unsigned long long offset, int* __restrict__ anchor_pix,
                                                              int* __restrict__ neg_pix) {
  const int v = blockIdx.z, q = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x, V = meta->V;
  if (v >= V || V < 2 || n > N) return;
  const int cid = meta->present[v];
  if (meta->cntH[cid] == 0) return;
  unsigned r[4];
  philox4x32((unsigned)n, (unsigned)(q | (v << 24)), (unsigned)offset, (unsigned)(offset >> 32), (unsigned)seed, (unsigned)(seed >> 32), r);
  if (n == N) {
    const int i = (int)(((unsigned long long)r[0] * (unsigned)meta->cntH[cid]) >> 32);
    anchor_pix[v * Q + q] = listH[meta->baseH[cid] + i];
    return;
  }
  const float u = (float)(r[0] >> 8) * (1.0f / 16777216.0f);
  int j = 0;
  while (j < V - 2 && !(cdf[v * 32 + j] > u)) ++j;
  int o = v + 1 + j;
  if (o >= V) o -= V;
  const int cj = meta->present[o];
  const int i = (int)(((unsigned long long)r[1] * (unsigned)meta->cntV[cj]) >> 32);
  neg_pix[((size_t)v * Q + q) * N + n] = listV[meta->baseV[cj] + i];
}
// injected reference-style indices (positions in the hard list / in the cyclic concatenation) -> pixel ids
__global__ __launch_bounds__(256) void contrast_resolve_kernel(const ContrastMeta* __restrict__ meta, const int* __restrict__ listV,
                                                               const int* __restrict__ listH, int Q, int N, const int* __restrict__ anchor_idx,
                                                               const int* __restrict__ neg_idx, int* __restrict__ anchor_pix,
                                                               int* __restrict__ neg_pix) {
  const int v = blockIdx.z, q = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x, V = meta->V;
  if (v >= V || V < 2 || n > N) return;
  const int cid = meta->present[v];
  if (meta->cntH[cid] == 0) return;
  if (n == N) {
    // indices are reduced modulo the list length so that out-of-range test input cannot read outside the lists
    anchor_pix[v * Q + q] = listH[meta->baseH[cid] + (unsigned)anchor_idx[v * Q + q] % (unsigned)meta->cntH[cid]];
    return;
  }
  int total = 0;
  for (int j = 0; j < V - 1; ++j) {
    int o = v + 1 + j;
    if (o >= V) o -= V;
    total += meta->cntV[meta->present[o]];
  }
  int idx = (int)((unsigned)neg_idx[((size_t)v * Q + q) * N + n] % (unsigned)total);
  int pix = -1;
  for (int j = 0; j < V - 1; ++j) {
    int o = v + 1 + j;
    if (o >= V) o -= V;
    const int cj = meta->present[o];
    if (idx < meta->cntV[cj]) { pix = listV[meta->baseV[cj] + idx]; break; }
    idx -= meta->cntV[cj];
  }
  neg_pix[((size_t)v * Q + q) * N + n] = pix;
}

// ---- 7. loss + d loss / d anchor ----------------------------------------------------------------
template <typename T> __device__ __forceinline__ void load16(const T* p, float* o);
template <> __device__ __forceinline__ void load16<bf16_t>(const bf16_t* p, float* o) {
  Vec16<bf16_t> a, b;
  a.load(p);
  b.load(p + 8);
#pragma unroll
  for (int e = 0; e < 8; ++e) { o[e] = a.f(e); o[8 + e] = b.f(e); }
}
template <> __device__ __forceinline__ void load16<float>(const float* p, float* o) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float4 v = reinterpret_cast<const float4*>(p)[i];
    o[4 * i] = v.x; o[4 * i + 1] = v.y; o[4 * i + 2] = v.z; o[4 * i + 3] = v.w;
  }
}
__device__ __forceinline__ float sum16(float v) {   // over the 16 lanes that share a row
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 8, 64);
  return v;
}
// one wave per (v, q); 4 gathered rows in flight per wave (16 lanes x 16 channels each); C == 256
template <typename T>
__global__ __launch_bounds__(256) void contrast_loss_kernel(const T* __restrict__ rep, int ld, const float* __restrict__ proto,
                                                            const ContrastMeta* __restrict__ meta, const int* __restrict__ anchor_pix,
                                                            const int* __restrict__ neg_pix, int Q, int N, float inv_temp,
                                                            float* __restrict__ loss_vq, float* __restrict__ gradbuf) {
  constexpr int C = 256;
  const int lane = threadIdx.x & 63, sub = lane >> 4, sl = lane & 15;
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int v = gw / Q, q = gw - v * Q, V = meta->V;
  if (v >= V || V < 2) return;
  const int cid = meta->present[v];
  if (meta->cntH[cid] == 0) {
    if (lane == 0) loss_vq[gw] = 0.f;
    return;
  }
  float a[16], pr[16], G[16];
  load16<T>(rep + (size_t)anchor_pix[gw] * ld + sl * 16, a);
  load16<float>(proto + (size_t)cid * C + sl * 16, pr);
  float na2 = 0.f, np2 = 0.f, d0 = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) { na2 += a[e] * a[e]; np2 += pr[e] * pr[e]; d0 += a[e] * pr[e]; }
  na2 = sum16(na2); np2 = sum16(np2); d0 = sum16(d0);
  const float na = fmaxf(sqrtf(na2), 1e-8f), npn = fmaxf(sqrtf(np2), 1e-8f);   // cosine_similarity eps (loss.py:146)
  const float cos0 = d0 / (na * npn), l0 = cos0 * inv_temp, m = inv_temp;
  const float e0 = __expf(l0 - m);
  float S = sub == 0 ? e0 : 0.f, Cc = sub == 0 ? e0 * cos0 : 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) G[e] = sub == 0 ? e0 * pr[e] / npn : 0.f;
  const int* np_ = neg_pix + (size_t)gw * N;
  for (int n0 = 0; n0 < N; n0 += 8) {
    float r0[16], r1[16];
    const int i0 = n0 + sub, i1 = n0 + 4 + sub;
    const bool ok0 = i0 < N, ok1 = i1 < N;
    if (ok0) load16<T>(rep + (size_t)np_[i0] * ld + sl * 16, r0);
    if (ok1) load16<T>(rep + (size_t)np_[i1] * ld + sl * 16, r1);
    float dt0 = 0.f, rr0 = 0.f, dt1 = 0.f, rr1 = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      if (ok0) { dt0 += a[e] * r0[e]; rr0 += r0[e] * r0[e]; }
      if (ok1) { dt1 += a[e] * r1[e]; rr1 += r1[e] * r1[e]; }
    }
    dt0 = sum16(dt0); rr0 = sum16(rr0); dt1 = sum16(dt1); rr1 = sum16(rr1);
    if (ok0) {
      const float nr = fmaxf(sqrtf(rr0), 1e-8f), cs = dt0 / (na * nr), ex = __expf(cs * inv_temp - m), w = ex / nr;
      S += ex; Cc += ex * cs;
#pragma unroll
      for (int e = 0; e < 16; ++e) G[e] += w * r0[e];
    }
    if (ok1) {
      const float nr = fmaxf(sqrtf(rr1), 1e-8f), cs = dt1 / (na * nr), ex = __expf(cs * inv_temp - m), w = ex / nr;
      S += ex; Cc += ex * cs;
#pragma unroll
      for (int e = 0; e < 16; ++e) G[e] += w * r1[e];
    }
  }
  // combine the 4 row slots (S, Cc were accumulated identically by the 16 lanes of a slot)
  S += __shfl_xor(S, 16, 64); S += __shfl_xor(S, 32, 64);
  Cc += __shfl_xor(Cc, 16, 64); Cc += __shfl_xor(Cc, 32, 64);
#pragma unroll
  for (int e = 0; e < 16; ++e) { G[e] += __shfl_xor(G[e], 16, 64); G[e] += __shfl_xor(G[e], 32, 64); }
  if (lane == 0) loss_vq[gw] = logf(S) + m - l0;
  if (sub == 0) {
    const float sc = inv_temp / na / (float)(Q * V);
    const float t = Cc / S - cos0;
    float* g = gradbuf + (size_t)gw * C + sl * 16;
#pragma unroll
    for (int e = 0; e < 16; ++e) g[e] = sc * (G[e] / S - pr[e] / npn - t * a[e] / na);
  }
}

// loss = (1/V) * sum_v mean_q loss_vq over classes with hard pixels; 0 when V <= 1 (loss.py:116-117,149)
__global__ __launch_bounds__(256) void contrast_reduce_kernel(const float* __restrict__ loss_vq, const ContrastMeta* __restrict__ meta, int Q,
                                                              float* __restrict__ loss) {
  __shared__ float red[256];
  const int V = meta->V;
  float s = 0.f;
  if (V >= 2)
    for (int v = 0; v < V; ++v) {
      if (meta->cntH[meta->present[v]] == 0) continue;
      for (int q = threadIdx.x; q < Q; q += 256) s += loss_vq[v * Q + q];
    }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *loss = V >= 2 ? red[0] / (float)(Q * V) : 0.f;
}

// dense d loss / d rep: rows of anchor pixels only; duplicates (sampling with replacement) are summed by the first occurrence
template <typename T>
__global__ __launch_bounds__(256) void contrast_scatter_grad_kernel(const float* __restrict__ gradbuf, const int* __restrict__ anchor_pix,
                                                                    const ContrastMeta* __restrict__ meta, int Q, const float* __restrict__ gscale,
                                                                    T* __restrict__ drep, int ld) {
  constexpr int C = 256;
  const int lane = threadIdx.x & 63;
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int v = gw / Q, q = gw - v * Q, V = meta->V;
  if (v >= V || V < 2 || meta->cntH[meta->present[v]] == 0) return;
  const int* ap = anchor_pix + v * Q;
  const int pix = ap[q];
  int dup = 0;
  for (int j = lane; j < q; j += 64) dup |= (ap[j] == pix);
  if (__ballot(dup)) return;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int j = q; j < Q; ++j) {
    if (ap[j] != pix) continue;
    const float4 g = *reinterpret_cast<const float4*>(gradbuf + ((size_t)v * Q + j) * C + lane * 4);
    acc[0] += g.x; acc[1] += g.y; acc[2] += g.z; acc[3] += g.w;
  }
  const float gs = *gscale;
  T* o = drep + (size_t)pix * ld + lane * 4;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = (T)(acc[e] * gs);
}

// ---- launchers ------------------------------------------------------------------------------------
size_t css_contrast_meta_bytes_() { return sizeof(ContrastMeta); }
int css_contrast_nchunks_(int P) { return (P + 64 * CHUNK_R - 1) / (64 * CHUNK_R); }

int css_launch_contrast_classify(const float* label, const float* mask, const float* prob, long sb, long sk, long sp, long psb, long psk,
                                 long psp, int P, int HW, int K, float strong_thr, int* cls, uint8_t* hard, void* meta, hipStream_t st) {
  if (K > CT_MAXK) return CSS_ERR_ARG;
  hipLaunchKernelGGL(contrast_classify_kernel, dim3(min(cdiv(P, 256), 4096)), dim3(256), 0, st, label, mask, prob, sb, sk, sp, psb, psk, psp, P,
                     HW, K, strong_thr, cls, hard, (ContrastMeta*)meta);
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}
constexpr int CT_SUMS_PPB = 2048;     // pixels per workgroup of contrast_class_sums_kernel
size_t css_contrast_class_sums_ws_bytes_(int P, int K, int C) { return (size_t)cdiv(P > 0 ? P : 1, CT_SUMS_PPB) * ((size_t)K * C + K) * sizeof(float); }
int css_launch_contrast_class_sums(const void* rep, int ld, const int* cls, int P, int K, int C, double* out, float* ws, int dtype, hipStream_t st) {
  if (K > CT_MAXK || C != 256 || !ws || P <= 0) return CSS_ERR_ARG;
  const int ppb = CT_SUMS_PPB, nb = cdiv(P, ppb);
  const size_t lds = ((size_t)4 * K * C + 4 * CT_MAXK) * sizeof(float);     // 86 KiB at K = 21: one workgroup per CU
  if (dtype == CSS_BF16)
    hipLaunchKernelGGL((contrast_class_sums_kernel<bf16_t, 256>), dim3(nb), dim3(256), lds, st, (const bf16_t*)rep, ld, cls, P, K, ppb, ws);
  else if (dtype == CSS_F32)
    hipLaunchKernelGGL((contrast_class_sums_kernel<float, 256>), dim3(nb), dim3(256), lds, st, (const float*)rep, ld, cls, P, K, ppb, ws);
  else return CSS_ERR_DTYPE;
  hipLaunchKernelGGL(contrast_class_sums_reduce_kernel, dim3(cdiv(K * C + K, 256)), dim3(256), 0, st, ws, nb, K * C + K, out);
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}
// chunkhist: int [nchunks][64]; listV/listH: int [P]
int css_launch_contrast_compact(const int* cls, const uint8_t* hard, int P, int K, int* chunkhist, int* listV, int* listH, void* meta,
                                hipStream_t st) {
  if (K > CT_MAXK) return CSS_ERR_ARG;
  const int nch = css_contrast_nchunks_(P);
  hipLaunchKernelGGL(contrast_hist_kernel, dim3(cdiv(nch, 4)), dim3(256), 0, st, cls, hard, P, K, chunkhist);
  hipLaunchKernelGGL(contrast_scan_kernel, dim3(1), dim3(64), 0, st, chunkhist, nch, K, (ContrastMeta*)meta);
  hipLaunchKernelGGL(contrast_scatter_kernel, dim3(cdiv(nch, 4)), dim3(256), 0, st, cls, hard, P, K, chunkhist, (const ContrastMeta*)meta, listV,
                     listH);
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}
int css_launch_contrast_proto_update(float* proto, const double* sums, int K, int C, float alpha, const void* meta, hipStream_t st) {
  if (K > CT_MAXK || C != 256) return CSS_ERR_ARG;
  hipLaunchKernelGGL(contrast_proto_update_kernel<256>, dim3(K), dim3(64), 0, st, proto, sums, K, alpha, (const ContrastMeta*)meta);
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}
int css_launch_contrast_sample(const float* proto, int C, const void* meta, float temp, float* cdf, const int* listV, const int* listH, int Q,
                               int N, unsigned long long seed, unsigned long long offset, int* anchor_pix, int* neg_pix, hipStream_t st) {
  if (C != 256) return CSS_ERR_ARG;
  hipLaunchKernelGGL(contrast_class_probs_kernel<256>, dim3(CT_MAXK), dim3(64), 0, st, proto, (const ContrastMeta*)meta, 1.f / temp, cdf);
  hipLaunchKernelGGL(contrast_sample_kernel, dim3(cdiv(N + 1, 256), Q, CT_MAXK), dim3(256), 0, st, (const ContrastMeta*)meta, cdf, listV, listH, Q,
                     N, seed, offset, anchor_pix, neg_pix);
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}
int css_launch_contrast_resolve(const void* meta, const int* listV, const int* listH, int Q, int N, const int* anchor_idx, const int* neg_idx,
                                int* anchor_pix, int* neg_pix, hipStream_t st) {
  hipLaunchKernelGGL(contrast_resolve_kernel, dim3(cdiv(N + 1, 256), Q, CT_MAXK), dim3(256), 0, st, (const ContrastMeta*)meta, listV, listH, Q, N,
                     anchor_idx, neg_idx, anchor_pix, neg_pix);
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}
int css_launch_contrast_loss(const void* rep, int ld, const float* proto, int K, int C, const void* meta, const int* anchor_pix,
                             const int* neg_pix, int Q, int N, float temp, float* loss_vq, float* gradbuf, float* loss, int dtype,
                             hipStream_t st) {
  if (C != 256 || K > CT_MAXK) return CSS_ERR_ARG;
  const int nw = K * Q;
  if (dtype == CSS_BF16)
    hipLaunchKernelGGL(contrast_loss_kernel<bf16_t>, dim3(cdiv(nw, 4)), dim3(256), 0, st, (const bf16_t*)rep, ld, proto, (const ContrastMeta*)meta,
                       anchor_pix, neg_pix, Q, N, 1.f / temp, loss_vq, gradbuf);
  else if (dtype == CSS_F32)
    hipLaunchKernelGGL(contrast_loss_kernel<float>, dim3(cdiv(nw, 4)), dim3(256), 0, st, (const float*)rep, ld, proto, (const ContrastMeta*)meta,
                       anchor_pix, neg_pix, Q, N, 1.f / temp, loss_vq, gradbuf);
  else return CSS_ERR_DTYPE;
  hipLaunchKernelGGL(contrast_reduce_kernel, dim3(1), dim3(256), 0, st, loss_vq, (const ContrastMeta*)meta, Q, loss);
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}
int css_launch_contrast_scatter_grad(const float* gradbuf, const int* anchor_pix, const void* meta, int K, int Q, const float* gscale,
                                     void* drep, int ld, int dtype, hipStream_t st) {
  const int nw = K * Q;
  if (dtype == CSS_BF16)
    hipLaunchKernelGGL(contrast_scatter_grad_kernel<bf16_t>, dim3(cdiv(nw, 4)), dim3(256), 0, st, gradbuf, anchor_pix, (const ContrastMeta*)meta, Q,
                       gscale, (bf16_t*)drep, ld);
  else if (dtype == CSS_F32)
    hipLaunchKernelGGL(contrast_scatter_grad_kernel<float>, dim3(cdiv(nw, 4)), dim3(256), 0, st, gradbuf, anchor_pix, (const ContrastMeta*)meta, Q,
                       gscale, (float*)drep, ld);
  else return CSS_ERR_DTYPE;
  CSS_CHECK_LAUNCH();
  return CSS_OK;
}
