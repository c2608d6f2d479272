// Internal: argument structs and the C++ launch entry points each .hip file exports to abi.hip.
#pragma once
#include "common.h"

struct ConvArgs {
  const void* src;    // gathered tensor   [N][Hs][Ws][lds]
  const void* wt;     // [Cd][Ktot]  (Ktot = R*S*Cs, Cs innermost)
  void* dst;          // [N][Hd][Wd][ldd]
  const float* bias;  // [Cd] or null
  int N, Hs, Ws, Cs, lds;
  int Hd, Wd, Cd, ldd;
  int R, S, stride, pad, dil;
  int mode;  // 0: forward gather  hs = hd*stride - pad + r*dil
             // 1: dgrad gather    hs = (hd + pad - r*dil) / stride   (must divide)
  int M, Ktot;
  unsigned src_bytes, wt_bytes;   // filled by the launcher (buffer descriptors)
  int m_begin;                    // first output row handled by this launch (rows m_begin .. M-1)
  // fused epilogue extras (all optional):
  float* stats;                   // batch-norm partial statistics [stat_nslab + G][2][Cd] (see conv.hip: store_wave_tile), or null
  int stat_Mg, stat_nslab;        // rows per statistics group; ceil(total M / 128), filled by the launcher
  int stat_G;                     // number of statistics groups (total M / stat_Mg), filled by the launcher
  const void* addend;             // [M][ld_add] tensor added to the result before it is stored (residual gradient), or null
  int ld_add;
  const unsigned char* add_mask;  // optional [M][Cd / VEC] bytes (VEC = elements per 16 bytes): bit e of byte (m, v) set = element v*VEC + e of the
                                  // addend is kept, clear = it counts as zero (the ReLU mask css_bn_apply_mask wrote: the addend is then the
                                  // gradient BEFORE that ReLU's backward, which this epilogue applies on the fly - css_conv2d_dgrad_add_masked)
  // conv_igemm_pp_kernel (conv_pp.hip), filled by the launcher:
  unsigned dst_bytes;             // bytes of dst the launch may write (buffer descriptor: rows >= M are dropped by the range check)
  FastDiv fd_hw, fd_w;            // division by Hd*Wd and Wd
  unsigned stat_bytes, add_bytes; // sizes of the stats / addend buffers (buffer descriptors)
  unsigned mask_bytes;            // size of add_mask
  int korder;                     // order of the K steps (conv_pp.hip: issue())
  int tab_da[63], tab_kb[63], tab_tap[63];   // conv_p8.hip: tap lists per set of valid kernel rows (7 x 9)
  int st_nt;                      // non-zero: the output tile leaves with non-temporal stores (filled by the launcher: CSS_CONV_NT; conv_p8.hip, conv_ws.hip)
};

struct WgradArgs {
  const void* x;   // [N][Hs][Ws][ldx]
  const void* dy;  // [M][ldy]
  float* dw;       // [Cd][Ktot] fp32, accumulated with atomics
  int N, Hs, Ws, Cs, ldx;
  int Hd, Wd, Cd, ldy;
  int R, S, stride, pad, dil;
  int M, Ktot, m_per_split;
  FastDiv fd_hw, fd_w;
  unsigned x_bytes, dy_bytes;   // filled by the launcher (buffer descriptors)
  int splits, tiles_k, tiles_n;
  float* ws;         // optional workspace for per-slice partial tiles (plain stores + ordered reduction instead of fp32 atomics), or null
  size_t ws_bytes;
  // Live-row compaction (conv_wgrad_p8_kernel, filled by the launcher; compact = 0: off).  For a dilated R x S convolution the output rows
  // whose source row lies in the padding for kernel row r contribute nothing to that row's weights: the pixel loop of a k-column tile of
  // kernel row r runs over the LIVE output rows [row_lo[r], row_lo[r] + row_n[r]) of every image only (row_mps[r] compacted pixels per slice).
  int nt;            // bit 0: non-temporal slab stores (filled by the launcher: CSS_WGRAD_NT)
  int compact;
  int row_lo[3], row_n[3], row_mps[3];
  // compact = 2: the kernel rows fall into a LONG class (all output rows live: the centre row) and a SHORT class; every XCD is dealt its share
  // of the long work items first, then its share of the short ones (longest-first: the short items fill the round's tail).  cls_rows[c] lists
  // the kernel rows of class c (cls_nrows[c] of them).
  int cls_nrows[2], cls_rows[2][3];
  FastDiv fd_L[3];   // division by row_n[r] * Wd
};


// Optional per-KERNEL timing hook (abi.hip brackets each kernel launch with HIP events when profiling is on):
// begin(big, share): big = the 256x256 LDS-DMA kernel, share = fraction of the call's output rows this launch covers.
struct LaunchProf {
  virtual void begin(bool big, double share, bool ws) = 0;     // ws: the launch is conv_ws_kernel (conv_ws.hip)
  virtual void end() = 0;
};
int css_launch_conv(const ConvArgs& a, int dtype, int n_cu, hipStream_t st, LaunchProf* prof = nullptr);
int css_launch_wgrad(WgradArgs a, int dtype, int n_cu, hipStream_t st, LaunchProf* prof = nullptr);
bool css_conv_pp_supported(const ConvArgs& a);
bool css_conv_c64_supported(const ConvArgs& a, int dtype);      // conv_c64.hip: 3x3 s1 p1, 64 input channels, 64 / 128 output channels
void css_launch_conv_c64(ConvArgs a, int n_cu, hipStream_t st);
int css_conv_c64_set_enabled_(int on);                           // returns the previous state
int css_conv_pp_plan(const ConvArgs& a, int n_cu);
void css_launch_conv_pp(ConvArgs a, int grid, hipStream_t st);
int css_conv_tile_rows_(const ConvArgs& a, int dtype, int n_cu);
bool css_conv_p8_supported(const ConvArgs& a);
void css_launch_conv_p8(ConvArgs a, int grid, hipStream_t st);
bool css_conv_ws_supported(const ConvArgs& a, int n_cu);
void css_launch_conv_ws(ConvArgs a, int n_cu, hipStream_t st);
void css_conv_ws_set_enabled(int on);
void css_wgrad_plan_(int M, int Ktot, int Cd, int dtype, int n_cu, int* splits_out, int* mps_out);
size_t css_wgrad_ws_bytes_(int M, int Ktot, int Cd, int dtype, int n_cu);

// conv_stem.hip: the stride-2 stem convolutions on the space-to-depth image (round 5)
int css_stem_s2d_enabled_();
int css_stem_s2d_tile_rows_();
int css_launch_nchw_to_s2d(const float* x, void* out, int N, int C, int H, int W, hipStream_t st);
int css_launch_stem_s2d_weights(const float* w, void* out, int Cout, int R, hipStream_t st);
int css_launch_stem_s2d_fold_wgrad(const float* dw2, float* dw, int Cout, int R, hipStream_t st);
int css_launch_conv_stem_s2d(const void* x, const void* w2, void* y, float* stats, int Mg, int N, int Hs, int Ws, int Cout, int R, int n_cu,
                             hipStream_t st);

int css_bn_nrb_(int Mg, int G, int C, int dtype);
int css_launch_bn_stats(const void* y, int Mg, int G, int C, int ld, double* partial, int dtype, hipStream_t st);
int css_launch_bn_reduce(const double* partial, int nrb, int C, int G, double* sums, float* g1, float* g0, int accumulate, double count_local,
                         hipStream_t st);
int css_launch_bn_reduce_finalize(const double* partial, int nrb, int G, double count, const float* gamma, const float* beta,
                                  float* running_mean, float* running_var, float momentum, float eps, float* mean, float* invstd, float* scale,
                                  float* shift, int C, hipStream_t st);
int css_launch_bn_reduce_slabs(const float* partial, int M, int Mg, int G, double count, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, float momentum, float eps, float* mean, float* invstd, float* scale,
                               float* shift, double* sums_out, int C, const void* y, int ldy, int tile_rows, hipStream_t st);
int css_launch_bn_finalize(const double* sums, int G, double count, const double* count_dev, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, float momentum, float eps, float* mean, float* invstd, float* scale,
                           float* shift, int C, hipStream_t st);
int css_launch_bn_eval_coeff(const float* gamma, const float* beta, const float* rm, const float* rv, float eps, float* scale, float* shift, int C,
                             hipStream_t st);
int css_launch_bn_apply(const void* y, int ldy, const void* res, int ldr, void* out, int ldo, const float* scale, const float* shift, int M, int C,
                        int relu, int Mg, unsigned char* mask, int dtype, hipStream_t st);
int css_launch_bn_apply_pool(const void* y, void* out, uint8_t* arg, const float* scale, const float* shift, int N, int H, int W, int C, int Ho, int Wo,
                             int G, int relu, int dtype, hipStream_t st);
int css_launch_bn_bwd_reduce(const void* da, int ldda, const void* a, int lda, const void* y, int ldy, const float* mean, const float* invstd,
                             const float* scale, const float* shift, int Mg, int G, int C, int relu, double* partial, const unsigned char* mask,
                             int dtype, hipStream_t st);
int css_launch_bn_bwd_apply(const void* da, int ldda, const void* a, int lda, const void* y, int ldy, void* dy, int lddy, void* dres, int lddr,
                            const float* mean, const float* invstd, const float* gamma, const double* sums, const float* scale, const float* shift,
                            double count, const double* count_dev, int M, int C, int relu, int Mg, const unsigned char* mask, int dtype,
                            hipStream_t st);

int css_launch_maxpool_fwd(const void* x, void* out, uint8_t* arg, int N, int H, int W, int C, int Ho, int Wo, int ks, int stride, int pad,
                           int dtype, hipStream_t st);
int css_launch_maxpool_bwd(const void* dout, const uint8_t* arg, void* dx, int N, int H, int W, int C, int Ho, int Wo, int ks, int stride, int pad,
                           int dtype, hipStream_t st);
int css_launch_bilinear(const void* x, int ldx, void* out, int ldo, int N, int Hs, int Ws, int C, int Hd, int Wd, int dtype_in, int dtype_out,
                        int backward, hipStream_t st);
int css_launch_spatial_sum(const void* x, int ldx, void* out, int N, int HW, int C, float scale, int dtype, hipStream_t st);
int css_launch_spatial_bcast(const void* x, void* out, int ldo, int N, int HW, int C, float scale, int dtype, hipStream_t st);
int css_launch_copy_channels(const void* src, int lds, void* dst, int ldd, long M, int C, int dtype_in, int dtype_out, hipStream_t st);
size_t css_colsum_ws_bytes_(long M, int C);
int css_launch_colsum(const void* x, int ld, long M, int C, float* out, float* ws, int dtype, hipStream_t st);
int css_launch_nchw_to_nhwc(const float* x, void* out, int N, int C, int HW, int Cpad, int dtype, hipStream_t st);
int css_launch_weight_layout(const float* w, void* out, int Cout, int taps, int Cin, int CinPad, int dgrad, int dtype, hipStream_t st);
int css_launch_weight_dgrad_layout_batched(const float* flat, void* out, const long* desc, int n_layers, long total_tiles, int dtype,
                                           hipStream_t st);
int css_launch_cast(const void* x, void* out, long n, int dtype_in, int dtype_out, hipStream_t st);
int css_launch_sgd_ema(float* p, const float* g, float* buf, float* ema, long n, float lr, float momentum, float wd, int first, float decay,
                       float grad_scale, const float* skip_flag, hipStream_t st);
int css_launch_ema(float* ema, const float* p, long n, float decay, hipStream_t st);

int css_launch_ce_small_fwd(const void* small, int ld, int B, int h, int w, const int64_t* label, const float* conf, float conf_thr,
                            const float* keep_thr, int K, int H, int W, int64_t* stats, float* gtprob_out, int dtype, hipStream_t st);
int css_launch_ce_small_bwd(const void* small, int ld, int B, int h, int w, const int64_t* label, const float* keep_thr, int K, int H, int W,
                            const float* coef, const float* gscale, int pos_only, float* dsmall, int dtype, hipStream_t st);
int css_launch_aug_geom(const float* img, const float* label, const float* l1, const float* l2, const int* params, int* table, int maxlen, int B,
                        int H, int W, int Hc, int Wc, unsigned char* img_q, unsigned char* lab_q, unsigned char* l1_q, unsigned char* l2_q,
                        hipStream_t st);
int css_launch_aug_color(unsigned char* img_q, unsigned char* tmp, const int* jp, unsigned long long* sums, int B, int H, int W, int any_jitter,
                         int any_blur, hipStream_t st);
int css_launch_aug_finish(const unsigned char* img_q, const unsigned char* lab_q, const unsigned char* l1_q, const unsigned char* l2_q,
                          const int* flags, int B, int Hc, int Wc, float* img, int64_t* label, float* l1, float* l2, hipStream_t st);
int css_launch_eval_confusion(const void* pred, int ldp, const int64_t* label, int B, int h, int w, int K, int H, int W, int64_t* mat,
                              uint8_t* argmax_out, int dtype, hipStream_t st);
int css_launch_confusion_bincount(const int64_t* pred, const int64_t* label, long n, int K, int64_t* mat, hipStream_t st);
int css_launch_proto_normalize(const float* proto, void* out, int K, int C, int dtype, hipStream_t st);
int css_launch_similarity(const void* rep, int ld, const void* pn, float* sim, float* prob, const int* cls, uint8_t* hard, int P, int K, int C,
                          float temp, float strong_thr, int dtype, int n_cu, hipStream_t st);
int css_launch_softmax_hard(const void* pred, int ld, const int* cls, int P, int K, float strong_thr, uint8_t* hard, int dtype, hipStream_t st);
int css_launch_pseudo_label(const float* sim, const void* pred, int ldp, int B, int h, int w, int K, int H, int W, float temp, float* logits_rep,
                            int64_t* labels_rep, float* logits_cls, int64_t* labels_cls, float* pseudo, int dtype, hipStream_t st);
int css_launch_class_map(const int64_t* l_lab, const int64_t* u_lab, const float* u_logits, float weak_thr, int B, int H, int W, int h, int w,
                         int* cls, hipStream_t st);

int css_launch_ce_fwd(const float* logits, const int64_t* label, const float* conf, float conf_thr, const float* keep_thr, int K, long P, int HW,
                      int64_t* stats, float* gtprob_out, hipStream_t st);
int css_launch_ce_finalize(const int64_t* stats, int B, int mode, float* loss, float* coef, hipStream_t st);
int css_launch_ce_bwd(const float* logits, const int64_t* label, const float* keep_thr, int K, long P, int HW, const float* coef,
                      const float* gscale, int pos_only, float* dlogits, hipStream_t st);
size_t css_ohem_state_bytes_();
size_t css_ohem_thr_offset_();
int css_launch_ohem_threshold(const float* gtprob, long P, const int64_t* stats, int B, int min_kept, float thresh, void* state, hipStream_t st);

size_t css_contrast_meta_bytes_();
int css_contrast_nchunks_(int P);
int css_launch_contrast_classify(const float* label, const float* mask, const float* prob, long sb, long sk, long sp, long psb, long psk, long psp,
                                 int P, int HW, int K, float strong_thr, int* cls, uint8_t* hard, void* meta, hipStream_t st);
size_t css_contrast_class_sums_ws_bytes_(int P, int K, int C);
int css_launch_contrast_class_sums(const void* rep, int ld, const int* cls, int P, int K, int C, double* out, float* ws, int dtype, hipStream_t st);
int css_launch_contrast_compact(const int* cls, const uint8_t* hard, int P, int K, int* chunkhist, int* listV, int* listH, void* meta,
                                hipStream_t st);
int css_launch_contrast_proto_update(float* proto, const double* sums, int K, int C, float alpha, const void* meta, hipStream_t st);
int css_launch_contrast_sample(const float* proto, int C, const void* meta, float temp, float* cdf, const int* listV, const int* listH, int Q, int N,
                               unsigned long long seed, unsigned long long offset, int* anchor_pix, int* neg_pix, hipStream_t st);
int css_launch_contrast_resolve(const void* meta, const int* listV, const int* listH, int Q, int N, const int* anchor_idx, const int* neg_idx,
                                int* anchor_pix, int* neg_pix, hipStream_t st);
int css_launch_contrast_loss(const void* rep, int ld, const float* proto, int K, int C, const void* meta, const int* anchor_pix, const int* neg_pix,
                             int Q, int N, float temp, float* loss_vq, float* gradbuf, float* loss, int dtype, hipStream_t st);
int css_launch_contrast_scatter_grad(const float* gradbuf, const int* anchor_pix, const void* meta, int K, int Q, const float* gscale, void* drep,
                                     int ld, int dtype, hipStream_t st);

// peer.hip: SyncBN statistics exchanged through peer-mapped device memory (no RCCL call)
constexpr int CSS_PEER_DEPTH = 4;
size_t css_peer_buffer_bytes_(int slot_doubles);
int css_launch_bn_peer_gather(const unsigned long long* bases, int world, int rank, unsigned long long seq, int slot_doubles, const double* local, int n,
                              double* out, int* status, long long timeout, int phase, hipStream_t st);
int css_launch_bn_peer_finalize(const unsigned long long* bases, int world, int rank, unsigned long long seq, int slot_doubles, const double* local, int G,
                                int C, const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                                float* mean, float* invstd, float* scale, float* shift, double* count_out, int* status, long long timeout, int phase,
                                hipStream_t st);
int css_launch_mix_boxes(const void* self, const void* partner, void* out, const int* boxes, const int* pj, int B, int P, int H, int W, int elem_bytes,
                         int mode, long long fill_bits, hipStream_t st);
