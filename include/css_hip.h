/* css_hip.h -- C ABI of libcss_hip.so: the MI355X (gfx950) kernels behind the CSS hot path.
 *
 * Boundary rules: plain pointers and sizes, no torch types; every buffer is caller
 * owned device memory; no hidden allocation and no host synchronisation inside any entry point; every
 * entry point takes the device index and the hipStream_t to launch on (backward runs on autograd-engine
 * threads, so nothing may depend on thread-local device/stream state); returns 0 or a negative CSS_ERR_*.
 *
 * Layout: activations are NHWC ("[N][H][W][ld]", channels innermost, ld >= C elements between pixels),
 * conv weights [Cout][R][S][Cin].  dtype: 0 = fp32 (parity path, exact-fp32 MFMA), 1 = bf16 (fp32 accumulate).
 *
 * Each group cites the reference interface it replaces (paths relative to the reference root).
 */
#ifndef CSS_HIP_H
#define CSS_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CSS_API __attribute__((visibility("default")))
typedef void* css_stream_t; /* hipStream_t */

#define CSS_DTYPE_F32 0
#define CSS_DTYPE_BF16 1

CSS_API int css_abi_version(void);
CSS_API int css_device_cu_count(int device);

/* ---- per-kernel timing with HIP events on the launch stream (bench.py roofline leg) ----------------
 * Every kernel launch of a bracketed call gets its own event pair.  kind: 0 = conv forward, 1 = conv dgrad, 2 = conv wgrad
 * (launches of every kernel but the 256x256 LDS-DMA ones), 3 = contrast loss gather, 4 = similarity,
 * 5 / 6 / 7 = forward / dgrad launches of the 256-channel-panel MFMA kernels (conv_igemm_p8_kernel, conv_igemm_pp_kernel, conv_ws_kernel)
 * and conv_wgrad_p8_kernel launches; 13 / 14 = the conv_ws_kernel launches among 5 / 6
 * again, with their FLOPs (13) and with their algorithmic bytes (14: the kernel is judged against both rooflines); 15 = the OTHER
 * launches among 5 / 6 (the persistent 256x256-tile kernels) with the call's algorithmic bytes - source, weights, output once each,
 * + addend and mask - so that a PMC traffic figure for that kernel can be read against them.
 * alg_work of a convolution launch = the call's algorithmic FLOPs x the share of output rows that launch covers.
 * HBM-bound kernels, alg_work = algorithmic BYTES (every operand read once, every result written once):
 * 8 = bn_apply, 9 = bn_bwd_apply, 10 = bn_bwd_reduce, 11 = sgd_ema, 12 = the write-bound 1x1 forward convolutions
 * (K <= 512 in, >= 4K out; the same launches are also counted under 0 / 5 with their FLOPs). */
CSS_API int css_prof_enable(int on);
CSS_API int css_prof_reset(void);
CSS_API int css_prof_read(int kind, double* total_ms, double* launches, double* alg_work);

/* ---- convolution: nn.Conv2d forward/backward as used by
 *      generalframeworks/networks/resnet.py:24-40,119-139 (Bottleneck), deeplabv3/aspp.py:17-72 (ASPP),
 *      deeplabv3/deeplabv3.py:115-133,151-169 (stem, decoder heads).  alg_flops is only recorded for profiling. */
CSS_API int css_conv2d_forward(const void* x, const void* w, const float* bias, void* y, int N, int H, int W, int Cin, int ldx, int Ho, int Wo,
                               int Cout, int ldy, int R, int S, int stride, int pad, int dil, double alg_flops, int dtype, int device,
                               css_stream_t stream);
/* css_conv2d_forward (bias-free, bf16) that also emits the batch-norm statistics of its output, saving bn_stats' pass over
 * the tensor (every convolution of the reference's backbone/ASPP/decoder is followed by BatchNorm: resnet.py:119-137,
 * aspp.py:21-62, deeplabv3.py:115-133).  The output holds G = M/Mg statistics groups of Mg >= 128 rows.
 * stats: fp32 [2 * ceil(M/256)][2][Cout]: two rows ("slabs") per convolution tile of BM = 256 rows - the tile's first and last 128 rows -
 * each holding the sums of the slab's rows that lie in the statistics group of its first row; BM = css_conv2d_forward_bnstats_tile_rows(same
 * arguments) (256 for every kernel the dispatcher can reach; the query stays so that callers never hard-code it).  Consumed by
 * css_bn_reduce_finalize_slabs, which sums the (< 128) rows past each group boundary from y itself. */
CSS_API int css_conv2d_forward_bnstats(const void* x, const void* w, void* y, float* stats, int Mg, int N, int H, int W, int Cin, int ldx, int Ho,
                                       int Wo, int Cout, int ldy, int R, int S, int stride, int pad, int dil, double alg_flops, int dtype,
                                       int device, css_stream_t stream);
CSS_API int css_conv2d_forward_bnstats_tile_rows(const void* x, const void* w, void* y, int N, int H, int W, int Cin, int ldx, int Ho, int Wo,
                                                 int Cout, int ldy, int R, int S, int stride, int pad, int dil, int dtype, int device);
/* host-side query (no launch): 1 when css_conv2d_forward[_bnstats] / css_conv2d_dgrad[_add] run this product on the weight-stationary
 * short-K kernel (csrc/conv_ws.hip: 1x1, stride 1, no padding, bf16, K = channels of the gathered tensor in {64, 128, 256}, N = output
 * channels a multiple of 256; conv3 of a Bottleneck forward - resnet.py:131-133 - and conv1 of a Bottleneck backward - resnet.py:123-125),
 * 0 when it takes the 256x256-tile kernels.  M = output rows, ld_* = row pitches in elements, n_cu = css_device_cu_count(). */
CSS_API int css_conv_ws_applies(int M, int K, int ld_src, int N, int ld_dst, int R, int S, int stride, int pad, int has_stats, int has_addend,
                                int ld_add, int has_bias, int dtype, int n_cu);
/* host-side query (no launch): 1 when css_conv2d_forward[_bnstats] / css_conv2d_dgrad run this product on the patch-in-LDS kernel of the 3x3
 * stride-1 pad-1 convolutions with 64 input channels (csrc/conv_c64.hip: conv2 of the layer-1 Bottlenecks - resnet.py:126-129 - forward and
 * data gradient, the deep stem's second and third convolution - resnet.py:177-190), 0 when it takes the implicit-GEMM kernels.  Cin / Cout =
 * channels of the gathered tensor / of the result; ld_* = row pitches in elements.  css_conv_c64_set_enabled(0) sends these shapes back to the
 * implicit-GEMM kernels (the A/B reference of tests/test_conv_c64_gpu.py; CSS_NO_C64_CONV=1 is the process-wide form) and returns the old state. */
CSS_API int css_conv_c64_applies(int N, int H, int W, int Cin, int ld_src, int Cout, int ld_dst, int R, int S, int stride, int pad, int dil,
                                 int has_addend, int has_bias, int dtype);
CSS_API int css_conv_c64_set_enabled(int on);
/* w_t: weights re-laid out as [Cin][R][S][Cout] (css_weight_layout dgrad=1); stride 1 or 2 */
CSS_API int css_conv2d_dgrad(const void* dy, const void* w_t, void* dx, int N, int H, int W, int Cin, int lddx, int Ho, int Wo, int Cout, int lddy,
                             int R, int S, int stride, int pad, int dil, double alg_flops, int dtype, int device, css_stream_t stream);
/* dx = dgrad(dy) + addend ([N*H*W][ld_add], same dtype): the sum autograd forms where a tensor feeds a convolution AND a
 * residual connection (Bottleneck: resnet.py:119-139), fused into the store */
CSS_API int css_conv2d_dgrad_add(const void* dy, const void* w_t, void* dx, const void* addend, int ld_add, int N, int H, int W, int Cin, int lddx,
                                 int Ho, int Wo, int Cout, int lddy, int R, int S, int stride, int pad, int dil, double alg_flops, int dtype,
                                 int device, css_stream_t stream);
/* dx = dgrad(dy) + addend (.) mask: as css_conv2d_dgrad_add, with the ReLU backward of the residual sum applied to the addend on the fly.
 * `addend` is the gradient that ARRIVED at relu(bn3(..) + identity) (resnet.py:135-137) and `mask` the bit mask css_bn_apply_mask wrote
 * for that ReLU ([N*H*W][Cin / V] bytes, V = elements per 16 bytes; bit e of byte (m, v) = output element v*V + e was positive): the
 * batch-norm backward then needs no masked copy of that gradient for the residual branch (css_bn_bwd_apply_mask with dres = NULL).
 * Cin, lddx, ld_add multiples of V; dx and addend 16-byte aligned. */
CSS_API int css_conv2d_dgrad_add_masked(const void* dy, const void* w_t, void* dx, const void* addend, int ld_add, const unsigned char* mask, int N,
                                        int H, int W, int Cin, int lddx, int Ho, int Wo, int Cout, int lddy, int R, int S, int stride, int pad,
                                        int dil, double alg_flops, int dtype, int device, css_stream_t stream);
/* dw: fp32 [Cout][R][S][Cin], ACCUMULATED: zero it first unless accumulating on purpose.  The pixels are reduced in slices (one
 * workgroup per weight tile and slice).  ws (optional, caller-owned, ws_bytes >= css_conv2d_wgrad_ws_bytes(N*Ho*Wo, R*S*Cin, Cout, ..)):
 * the slices' partial tiles are written there with plain stores and summed into dw in a fixed order by a second kernel - faster than
 * fp32 atomics (1.3 TB/s chip-wide on MI355X) and bit-reproducible; ws == NULL or too small: fp32 atomic adds into dw. */
CSS_API size_t css_conv2d_wgrad_ws_bytes(int M, int Ktot, int Cout, int dtype, int device);
CSS_API int css_conv2d_wgrad(const void* x, const void* dy, float* dw, float* ws, size_t ws_bytes, int N, int H, int W, int Cin, int ldx, int Ho,
                             int Wo, int Cout, int lddy, int R, int S, int stride, int pad, int dil, double alg_flops, int dtype, int device,
                             css_stream_t stream);
/* host-side query (no launch): the number of pixel slices css_conv2d_wgrad cuts M = N*Ho*Wo rows into on a device with n_cu compute
 * units (one accumulation pass of fp32 atomics per slice); tests use it to check that bench-size layers take the split path */
CSS_API int css_wgrad_splits(int M, int Ktot, int Cout, int dtype, int n_cu);
/* fp32 master [Cout][taps][Cin] -> compute dtype; dgrad=0: [Cout][taps][CinPad], dgrad=1: [Cin][taps][Cout] */
CSS_API int css_weight_layout(const float* w, void* out, int Cout, int taps, int Cin, int CinPad, int dgrad, int dtype, int device,
                              css_stream_t stream);
/* every dgrad layout of a flat fp32 parameter buffer in one launch (the weights all change together, at the optimizer step:
 * mix_label.py:194).  desc: device int64 [n_layers][6] = {src_off, dst_off, Cout, taps, Cin, first_tile} in elements of flat / out;
 * layer l owns tiles [first_tile_l, first_tile_l + ceil(Cin/32)*ceil(Cout/32)*taps);  total_tiles = their sum */
CSS_API int css_weight_dgrad_layout_batched(const float* flat, void* out, const long* desc, int n_layers, long total_tiles, int dtype,
                                            int device, css_stream_t stream);

/* ---- batch norm: nn.BatchNorm2d / nn.SyncBatchNorm (mix_label.py:76) in train and eval mode --------
 * Tensors are [M = G*Mg][C]: G statistics groups of Mg rows each -- G forward passes of the reference batched into one tensor
 * (teacher on labeled+unlabeled, ddp_model.py:102-103; student on labeled+augmented, :140-143), every group normalised with
 * its own batch statistics and the running statistics updated once per group in order, i.e. exactly G separate calls.
 * Two-stage per-channel reduction: css_bn_stats / css_bn_bwd_reduce store one fp64 partial row [2][C] per row-block and group
 * (partial is [G][nrb][2][C], nrb = css_bn_nrb(Mg, G, C, dtype); plain stores, no atomics); css_bn_reduce sums them
 * (sums is [G][2][C]; can also emit the BN parameter gradients, summed over groups); css_bn_reduce_finalize fuses the sum
 * with the train-mode finalize for the single-rank case; css_bn_finalize starts from (all-reduced) sums.
 * mean / invstd / scale / shift are [G][C].
 * SyncBN with per-rank pixel counts (nn.SyncBatchNorm exchanges counts, mix_label.py:76): a sums buffer handed to a collective is
 * [G][2][C] + [G]: css_bn_reduce (count_local > 0) and css_bn_reduce_finalize_slabs (sums_out) put THIS rank's rows per group behind
 * the sums, so one all-reduce yields the global sums and the global counts; css_bn_finalize / css_bn_bwd_apply read the count of group
 * g from count_dev[g] when count_dev != NULL (the host value `count` is then ignored). */
CSS_API int css_bn_nrb(int Mg, int G, int C, int dtype);
CSS_API int css_bn_stats(const void* y, int Mg, int G, int C, int ld, double* partial, int dtype, int device, css_stream_t stream);
CSS_API int css_bn_reduce(const double* partial, int nrb, int C, int G, double* sums, float* dgamma, float* dbeta, int accumulate,
                          double count_local, int device, css_stream_t stream);
CSS_API int css_bn_reduce_finalize(const double* partial, int nrb, int G, double count, const float* gamma, const float* beta,
                                   float* running_mean, float* running_var, float momentum, float eps, float* mean, float* invstd, float* scale,
                                   float* shift, int C, int device, css_stream_t stream);
/* stage 2 for css_conv2d_forward_bnstats: partial fp32 [2 * ceil(M/tile_rows)][2][C] (tile_rows = 256, see there) + the bf16
 * tensor y [M][ldy] the statistics are of -> per-group sums (fp64).  sums_out == NULL:
 * train-mode finalize like css_bn_reduce_finalize; else only write sums_out [G][2][C] + [G] local counts (SyncBN all-reduces them, then
 * css_bn_finalize) */
CSS_API int css_bn_reduce_finalize_slabs(const float* partial, int M, int Mg, int G, double count, const float* gamma, const float* beta,
                                         float* running_mean, float* running_var, float momentum, float eps, float* mean, float* invstd,
                                         float* scale, float* shift, double* sums_out, int C, const void* y, int ldy, int tile_rows,
                                         int device, css_stream_t stream);
CSS_API int css_bn_finalize(const double* sums, int G, double count, const double* count_dev, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, float momentum, float eps, float* mean, float* invstd, float* scale,
                            float* shift, int C, int device, css_stream_t stream);
/* ---- SyncBN statistics through peer-mapped device memory instead of ~350 host-issued all-reduces per step (mix_label.py:76; DESIGN.md 6b,
 * css_amd/csrc/peer.hip).  Every rank owns one exchange buffer of css_peer_buffer_bytes(slot_doubles) bytes (zeroed once) that all ranks of
 * the node map; bases = device array of the `world` buffer addresses AS THIS RANK SEES THEM, in rank order.  One exchange = one call per rank
 * with the same seq (1, 2, 3, ... - the ranks run the same layers in the same order): the n local doubles are published, every peer's are
 * awaited (bounded by timeout_ticks of the 100 MHz wall clock: on expiry status[0] = seq and the call completes with what it has - never a
 * hang) and summed in rank order (bit-identical on all ranks).  css_bn_peer_finalize (forward): local = [G][2][C] sums + [G] local row
 * counts as css_bn_reduce / css_bn_reduce_finalize_slabs emit them; does css_bn_finalize's work in the same launch and writes the global
 * counts to count_out[G].  css_bn_peer_gather (backward): out[n] = the summed doubles (out may alias local).  phase 0 = whole exchange;
 * 1 = publish only, 2 = wait + sum only (tests that play several ranks in one process). */
/* Exchange-buffer memory (peers poll flags and read payloads INSIDE a running kernel; HIP promises cross-device visibility of
 * ordinary coarse-grained hipMalloc memory at kernel boundaries only).  css_peer_alloc: bytes of FINE-GRAINED device memory
 * (hipExtMallocWithFlags(hipDeviceMallocFinegrained), hipDeviceMallocUncached if that is refused), zero-filled, the device synchronised - the
 * one documented allocation of the library (a caching allocator cannot hand out this memory type).  css_peer_ipc_export: the 64-byte
 * hipIpcMemHandle_t of such a buffer; css_peer_ipc_open: a peer's buffer mapped into this process (node-local: hipIpc does not cross hosts);
 * css_peer_ipc_close / css_peer_free undo them.  *mem_kind_out: 1 = fine-grained, 2 = uncached. */
CSS_API int css_peer_alloc(size_t bytes, int device, void** out, int* mem_kind_out);
CSS_API int css_peer_free(void* p, int device);
CSS_API int css_peer_ipc_export(void* p, int device, unsigned char* handle64);
CSS_API int css_peer_ipc_open(const unsigned char* handle64, int device, void** out);
CSS_API int css_peer_ipc_close(void* p, int device);
CSS_API size_t css_peer_buffer_bytes(int slot_doubles);
CSS_API int css_bn_peer_finalize(const unsigned long long* bases, int world, int rank, unsigned long long seq, int slot_doubles, const double* local,
                                 int G, int C, const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                                 float eps, float* mean, float* invstd, float* scale, float* shift, double* count_out, int* status,
                                 long timeout_ticks, int phase, int device, css_stream_t stream);
CSS_API int css_bn_peer_gather(const unsigned long long* bases, int world, int rank, unsigned long long seq, int slot_doubles, const double* local,
                               int n, double* out, int* status, long timeout_ticks, int phase, int device, css_stream_t stream);
CSS_API int css_bn_eval_coeff(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps, float* scale,
                              float* shift, int C, int device, css_stream_t stream);
CSS_API int css_bn_apply(const void* y, int ldy, const void* res, int ldr, void* out, int ldo, const float* scale, const float* shift, int M, int C,
                         int relu, int Mg, int dtype, int device, css_stream_t stream);
/* css_bn_apply + ReLU + the 3x3 stride-2 pad-1 max pool behind the stem's batch norm (/root/reference/generalframeworks/networks/resnet.py:186-190,
 * torchvision's bn1 / relu / maxpool) in one pass: y [N][H][W][C] -> out [N][Ho][Wo][C] + argmax bytes (tap index r * 3 + s, NULL: not wanted); scale /
 * shift [G][C] (group of image n = n / (N / G)).  Bit-identical to css_bn_apply followed by css_maxpool_fwd; backward = css_maxpool_bwd then the
 * batch-norm backward entry points (the ReLU mask is recomputed from y). */
CSS_API int css_bn_apply_maxpool(const void* y, void* out, uint8_t* argmax, const float* scale, const float* shift, int N, int H, int W, int C, int Ho,
                                 int Wo, int G, int relu, int dtype, int device, css_stream_t stream);
/* backward: `a` (the saved activation, for the ReLU mask) may be NULL for layers without a residual: the mask is then
 * recomputed as y*scale+shift > 0 from the forward's scale/shift ([G][C]), saving one full read of the layer tensor */
CSS_API int css_bn_bwd_reduce(const void* da, int ldda, const void* a, int lda, const void* y, int ldy, const float* mean, const float* invstd,
                              const float* scale, const float* shift, int Mg, int G, int C, int relu, double* partial, int dtype, int device,
                              css_stream_t stream);
CSS_API int css_bn_bwd_apply(const void* da, int ldda, const void* a, int lda, const void* y, int ldy, void* dy, int lddy, void* dres, int lddr,
                             const float* mean, const float* invstd, const float* gamma, const double* sums, const float* scale,
                             const float* shift, double count, const double* count_dev, int M, int C, int relu, int Mg, int dtype, int device,
                             css_stream_t stream);
/* Residual layers (bn3 + identity + ReLU of a Bottleneck, resnet.py:133-137): the ReLU mask travels as ONE BYTE per 16-byte vector of
 * the output (bit e = element e of the vector is > 0 after the ReLU; mask is [M][C / (16 / sizeof(element))] bytes, contiguous) instead of
 * the backward passes re-reading the activation tensor: css_bn_apply_mask = css_bn_apply that also writes the mask (mask may be NULL);
 * css_bn_bwd_reduce_mask / css_bn_bwd_apply_mask = the ReLU forms of css_bn_bwd_reduce / css_bn_bwd_apply reading it. */
CSS_API int css_bn_apply_mask(const void* y, int ldy, const void* res, int ldr, void* out, int ldo, const float* scale, const float* shift, int M,
                              int C, int relu, int Mg, unsigned char* mask, int dtype, int device, css_stream_t stream);
CSS_API int css_bn_bwd_reduce_mask(const void* da, int ldda, const unsigned char* mask, const void* y, int ldy, const float* mean,
                                   const float* invstd, int Mg, int G, int C, double* partial, int dtype, int device, css_stream_t stream);
CSS_API int css_bn_bwd_apply_mask(const void* da, int ldda, const unsigned char* mask, const void* y, int ldy, void* dy, int lddy, void* dres,
                                  int lddr, const float* mean, const float* invstd, const float* gamma, const double* sums, double count,
                                  const double* count_dev, int M, int C, int Mg, int dtype, int device, css_stream_t stream);

/* ---- pooling / resize / concat: deeplabv3.py:153,164-166; aspp.py:27-38,67-72; ddp_model.py:141,144 */
CSS_API int css_maxpool_fwd(const void* x, void* out, uint8_t* argmax, int N, int H, int W, int C, int Ho, int Wo, int ks, int stride, int pad,
                            int dtype, int device, css_stream_t stream);
CSS_API int css_maxpool_bwd(const void* dout, const uint8_t* argmax, void* dx, int N, int H, int W, int C, int Ho, int Wo, int ks, int stride,
                            int pad, int dtype, int device, css_stream_t stream);
/* bilinear, align_corners=True.  backward=0: x [N,Hs,Ws,ldx] -> out [N,Hd,Wd,ldo]; backward=1: x = d(out) [N,Hd,Wd,ldx] -> out = d(x) [N,Hs,Ws,ldo] */
CSS_API int css_bilinear(const void* x, int ldx, void* out, int ldo, int N, int Hs, int Ws, int C, int Hd, int Wd, int dtype_in, int dtype_out,
                         int backward, int device, css_stream_t stream);
CSS_API int css_spatial_sum(const void* x, int ldx, void* out, int N, int HW, int C, float scale, int dtype, int device, css_stream_t stream);
CSS_API int css_spatial_bcast(const void* x, void* out, int ldo, int N, int HW, int C, float scale, int dtype, int device, css_stream_t stream);
CSS_API int css_copy_channels(const void* src, int lds, void* dst, int ldd, long M, int C, int dtype_in, int dtype_out, int device,
                              css_stream_t stream);
/* out[c] += sum_m x[m][c]  (bias gradient of the 1x1 heads, deeplabv3.py:125,132); out is fp32 and accumulated.  Two stages through
 * ws (fp32, css_colsum_ws_bytes(M, C) bytes, caller-owned): one partial row per row block, then an ordered sum - no float atomics. */
CSS_API size_t css_colsum_ws_bytes(long M, int C);
CSS_API int css_colsum(const void* x, int ld, long M, int C, float* out, float* ws, int dtype, int device, css_stream_t stream);
/* ---- the stride-2 stem convolution on the SPACE-TO-DEPTH image (round 5; css_amd/csrc/conv_stem.hip).  Replaces, for bf16, the first convolution
 * of the backbone: torchvision's conv1 = Conv2d(3, 64, 7, stride 2, padding 3) behind models.resnet101() (/root/reference/mix_label.py:68) and
 * ResNet_Stem.conv1[0] = conv3x3(3, 64, stride 2) (/root/reference/generalframeworks/networks/resnet.py:177-190), R = 7 or 3.
 * css_nchw_to_s2d: fp32 NCHW image [N][C <= 3][H][W] -> bf16 [N][Hs][Ws][16], Hs = ceil(H / 2), Ws = ceil(W / 2); channel (2 py + px) 3 + c of
 * s2d pixel (ys, xs) = image(c, 2 ys + py, 2 xs + px), zero outside the image and in channels 12..15.
 * css_stem_s2d_weights: fp32 master [64][R][R][3] -> bf16 [64][TA][TA][16], TA = (R + 1) / 2: w2[a][b][(2 py + px) 3 + c] = w[2 a + py - 1][2 b + px - 1][c].
 * css_conv2d_stem_s2d_forward: y [N][Hs][Ws][64] bf16 = the R x R stride-2 pad R/2 convolution; stats (optional) = the batch-norm statistics slabs
 * of css_conv2d_forward_bnstats (fp32 [2 ceil(M / BT)][2][64], BT = css_conv2d_stem_s2d_tile_rows() - the value stage 2 must be given), Mg rows
 * per statistics group.
 * css_stem_s2d_fold_wgrad: the weight gradient computed in s2d space (css_conv2d_wgrad on the s2d image: TA x TA taps, stride 1, pad TA / 2,
 * 16 channels) dw2 fp32 [64][TA][TA][16] ADDED into dw fp32 [64][R][R][3].  css_stem_s2d_enabled: 0 under CSS_NO_STEM_S2D=1 (the gather kernels). */
CSS_API int css_stem_s2d_enabled(void);
CSS_API int css_conv2d_stem_s2d_tile_rows(void);
CSS_API int css_nchw_to_s2d(const float* x, void* out, int N, int C, int H, int W, int device, css_stream_t stream);
CSS_API int css_stem_s2d_weights(const float* w, void* out, int Cout, int R, int device, css_stream_t stream);
CSS_API int css_stem_s2d_fold_wgrad(const float* dw2, float* dw, int Cout, int R, int device, css_stream_t stream);
CSS_API int css_conv2d_stem_s2d_forward(const void* x_s2d, const void* w2, void* y, float* stats, int Mg, int N, int Hs, int Ws, int Cout, int R,
                                        double alg_flops, int device, css_stream_t stream);
CSS_API int css_nchw_to_nhwc(const float* x, void* out, int N, int C, int HW, int Cpad, int dtype, int device, css_stream_t stream);
CSS_API int css_cast(const void* x, void* out, long n, int dtype_in, int dtype_out, int device, css_stream_t stream);

/* ---- optimiser + EMA teacher: torch.optim.SGD(nesterov) mix_label.py:96-97,194; Model_mix.ema_update ddp_model.py:93-97 */
CSS_API int css_sgd_ema(float* p, const float* g, float* buf, float* ema, long n, float lr, float momentum, float wd, int first, float decay,
                        float grad_scale, const float* skip_flag, int device, css_stream_t stream);
CSS_API int css_ema(float* ema, const float* p, long n, float decay, int device, css_stream_t stream);

/* ---- similarity / pseudo labels: ddp_model.py:104-118,147-154; mix_label.py:175-183 ------------------ */
CSS_API int css_proto_normalize(const float* proto, void* out, int K, int C, int dtype, int device, css_stream_t stream);
CSS_API int css_similarity(const void* rep, int ld, const void* proto_n, float* sim, float* prob, const int* cls, uint8_t* hard, int P, int K, int C,
                           float temp, float strong_thr, int dtype, int device, css_stream_t stream);
/* ori_pseudo.py:178-180 + loss.py:90-91: hard[p] = cls[p] >= 0 && softmax(pred[p][0..K))[cls[p]] < strong_thr, pred = the student's own
 * low-resolution class logits [P][ld] (no prototype similarity in that script) */
CSS_API int css_softmax_hard_flags(const void* pred, int ld, const int* cls, int P, int K, float strong_thr, uint8_t* hard, int dtype, int device,
                                   css_stream_t stream);
CSS_API int css_pseudo_label(const float* sim, const void* pred, int ldp, int B, int h, int w, int K, int H, int W, float temp, float* logits_rep,
                             int64_t* labels_rep, float* logits_cls, int64_t* labels_cls, float* pseudo, int dtype, int device,
                             css_stream_t stream);
CSS_API int css_class_map(const int64_t* l_lab, const int64_t* u_lab, const float* u_logits, float weak_thr, int B, int H, int W, int h, int w,
                          int* cls, int device, css_stream_t stream);

/* ---- in-step augmentation of the unlabeled batch (dataset_helpers/VOC.py:126-196,284-291,339-352) on 8-bit planes.
 * css_aug_geom: img fp32 [B][3][H][W] (ImageNet-normalised), label fp32 [B][H][W] (class id, 255 or -1), two confidence maps
 * fp32 [B][H][W] in [0,1] -> the PIL images after tensor_to_pil_2 + rescale + pad + crop, as uint8 planes [B][3|1][Hc][Wc].
 * params int32 [B][4] = {resized height, resized width, crop row, crop column} (host draws); table: int32 workspace
 * [2B][maxlen], maxlen >= max(resized sizes).  Bit-exact to PIL (BILINEAR two-pass fixed point / NEAREST) for scales >= 0.5.
 * css_aug_finish: flags int32 [B] (bit 0 = horizontal flip) -> img fp32 normalised, label int64 (255 -> -1), maps fp32 (q/255). */
CSS_API int css_aug_geom(const float* img, const float* label, const float* logits1, const float* logits2, const int* params, int* table,
                         int maxlen, int B, int H, int W, int Hc, int Wc, uint8_t* img_q, uint8_t* lab_q, uint8_t* l1_q, uint8_t* l2_q, int device,
                         css_stream_t stream);
/* colour jitter (torchvision ColorJitter = PIL ImageEnhance blends + 8-bit HSV hue shift, random order) and
 * ImageFilter.GaussianBlur (three 3-tap box passes per axis) in place on the uint8 planes [B][3][H][W]; tmp: same size; sums: int64 [B]
 * workspace.  jp int32 [B][16]: 0 jitter on, 1..4 op order (0 brightness 1 contrast 2 saturation 3 hue), 5..7 the three factors
 * (float bits), 8 hue shift (uint8), 9 blur on, 10 / 11 box-blur centre / neighbour weight (24-bit fixed point). */
CSS_API int css_aug_color(uint8_t* img_q, uint8_t* tmp, const int* jp, int64_t* sums, int B, int H, int W, int any_jitter, int any_blur,
                          int device, css_stream_t stream);
/* cutmix / cutout boxes of a whole batch in one launch per tensor (generate_cut_gather*, VOC.py:354-477): out[b][p][y][x] = inside box b ?
 * (mode 0: partner[pj[b]][p][y][x] | mode 1: fill_bits) : self[b][p][y][x]; boxes int32 [B][4] = {y0, y1, x0, x1} half-open; contiguous
 * [B][P][H][W] tensors of elem_bytes = 4 or 8 (fp32 images / confidence maps, int64 label maps).  A pure copy: bit-exact. */
CSS_API int css_mix_boxes(const void* self, const void* partner, void* out, const int* boxes, const int* pj, int B, int P, int H, int W, int elem_bytes,
                          int mode, long fill_bits, int device, css_stream_t stream);
CSS_API int css_aug_finish(const uint8_t* img_q, const uint8_t* lab_q, const uint8_t* l1_q, const uint8_t* l2_q, const int* flags, int B, int Hc,
                           int Wc, float* img, int64_t* label, float* logits1, float* logits2, int device, css_stream_t stream);

/* ---- evaluation: test() of mix_label.py:199-225.  css_eval_confusion fuses F.interpolate(bilinear,
 * align_corners=True) of the NHWC logits [B][h][w][ldp] to the label size, argmax over K and ConfMatrix.update
 * (util/meter.py:39-48): mat (int64 [K][K], row = target, column = prediction) is ACCUMULATED; labels outside [0,K) are
 * ignored; argmax_out (uint8 [B][H][W]) is optional.  css_confusion_bincount is ConfMatrix.update on class indices. K <= 32. */
CSS_API int css_eval_confusion(const void* pred, int ldp, const int64_t* label, int B, int h, int w, int K, int H, int W, int64_t* mat,
                               uint8_t* argmax_out, int dtype, int device, css_stream_t stream);
CSS_API int css_confusion_bincount(const int64_t* pred, const int64_t* label, long n, int K, int64_t* mat, int device, css_stream_t stream);

/* ---- cross-entropy family: mix_label.py:81,169; loss/loss.py:19-46 (OHEM), :53-64 (Attention_Threshold_Loss).
 * stats: int64 [B][4] per-image accumulators, zeroed by the caller: {sum of losses in units of 2^-32, #(loss > 0), #(counted pixels),
 * #(conf >= conf_thr)} - integers, so the workgroups' atomic adds give the same bits in any arrival order (reproducible losses and,
 * through coef, reproducible gradients). */
CSS_API int css_ce_fwd(const float* logits, const int64_t* label, const float* conf, float conf_thr, const float* keep_thr, int K, long P, int HW,
                       int64_t* stats, float* gtprob_out, int device, css_stream_t stream);
CSS_API int css_ce_finalize(const int64_t* stats, int B, int mode, float* loss, float* coef, int device, css_stream_t stream);
CSS_API int css_ce_bwd(const float* logits, const int64_t* label, const float* keep_thr, int K, long P, int HW, const float* coef,
                       const float* gscale, int pos_only, float* dlogits, int device, css_stream_t stream);
/* the same losses computed straight from the LOW-resolution logits small [B][h][w][ld] (bf16 / fp32): the bilinear
 * (align_corners=True) up-sampling to the label size [B][H][W] of ddp_model.py:141,144 is applied on the fly, forward and
 * adjoint, so the [B,K,H,W] logits and their gradient are never materialised.  css_ce_small_bwd ACCUMULATES into dsmall
 * (fp32 [B][h][w][K], zero it first) and needs an up-sampling factor >= 2.  Same stats / coef / keep_thr protocol as above.
 * The backward runs as 2 x 2 (factor <= 16) launches over tiles with pairwise disjoint footprints, each adding with plain
 * read-modify-writes in stream order: no float atomics, bit-reproducible (factor <= 4; above that the adjoint inside a workgroup
 * still uses LDS float atomics). */
CSS_API int css_ce_small_fwd(const void* small, int ld, int B, int h, int w, const int64_t* label, const float* conf, float conf_thr,
                             const float* keep_thr, int K, int H, int W, int64_t* stats, float* gtprob_out, int dtype, int device,
                             css_stream_t stream);
CSS_API int css_ce_small_bwd(const void* small, int ld, int B, int h, int w, const int64_t* label, const float* keep_thr, int K, int H, int W,
                             const float* coef, const float* gscale, int pos_only, float* dsmall, int dtype, int device, css_stream_t stream);
CSS_API size_t css_ohem_state_bytes(void);
CSS_API size_t css_ohem_thr_offset(void);
CSS_API int css_ohem_threshold(const float* gtprob, long P, const int64_t* stats, int B, int min_kept, float thresh, void* state, int device,
                               css_stream_t stream);

/* ---- contrastive loss: loss/loss.py:75-149, 410-418 ---------------------------------------------------- */
CSS_API size_t css_contrast_meta_bytes(void);
CSS_API int css_contrast_nchunks(int P);
CSS_API int css_contrast_classify(const float* label, const float* mask, const float* prob, long sb, long sk, long sp, long psb, long psk, long psp,
                                  int P, int HW, int K, float strong_thr, int* cls, uint8_t* hard, void* meta, int device, css_stream_t stream);
/* out fp64 [K][C] class sums + [K] class counts (loss.py:77-101 as sums).  ws: fp32 workspace of css_contrast_class_sums_ws_bytes(P, K, C)
 * bytes (one partial [K][C] + [K] per workgroup; stage 2 adds them in workgroup order: no float atomics, bit-reproducible). */
CSS_API size_t css_contrast_class_sums_ws_bytes(int P, int K, int C);
CSS_API int css_contrast_class_sums(const void* rep, int ld, const int* cls, int P, int K, int C, double* out, float* ws, int dtype, int device,
                                    css_stream_t stream);
CSS_API int css_contrast_compact(const int* cls, const uint8_t* hard, int P, int K, int* chunkhist, int* listV, int* listH, void* meta, int device,
                                 css_stream_t stream);
CSS_API int css_contrast_proto_update(float* proto, const double* sums, int K, int C, float alpha, const void* meta, int device,
                                      css_stream_t stream);
CSS_API int css_contrast_sample(const float* proto, int C, const void* meta, float temp, float* cdf, const int* listV, const int* listH, int Q, int N,
                                unsigned long long seed, unsigned long long offset, int* anchor_pix, int* neg_pix, int device,
                                css_stream_t stream);
CSS_API int css_contrast_resolve(const void* meta, const int* listV, const int* listH, int Q, int N, const int* anchor_idx, const int* neg_idx,
                                 int* anchor_pix, int* neg_pix, int device, css_stream_t stream);
CSS_API int css_contrast_loss(const void* rep, int ld, const float* proto, int K, int C, const void* meta, const int* anchor_pix, const int* neg_pix,
                              int Q, int N, float temp, float* loss_vq, float* gradbuf, float* loss, int dtype, int device, css_stream_t stream);
CSS_API int css_contrast_scatter_grad(const float* gradbuf, const int* anchor_pix, const void* meta, int K, int Q, const float* gscale, void* drep,
                                      int ld, int dtype, int device, css_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CSS_HIP_H */
