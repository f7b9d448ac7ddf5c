/* avdf.h -- C ABI of libavdf_hip.so: the MI355X (gfx950) hot path of the
 * audio-visual deepfake temporal-localisation system.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous fp32 unless stated otherwise;
 *   - activations are TIME-MAJOR [B, T, C] (the reference is channel-major [B, C, T]);
 *   - masks are fp32 [B, T] holding 1.0 (valid) / 0.0 (padding) at the tensor's own T;
 *   - weights keep the REFERENCE layouts: Conv1d [Cout, Cin/groups, K], Linear [out, in],
 *     LayerNorm / AffineDropPath [1, C, 1] (read as C floats), so state-dicts are interchangeable;
 *   - `stream` is a hipStream_t; kernels are asynchronous on it, never allocate, never sync;
 *   - every entry point returns 0 on success or a negative AVDF_E* code; the message is
 *     available from avdf_last_error() (thread local).
 *
 * Each entry point cites the reference code it replaces (paths relative to the reference repo).
 */
#ifndef AVDF_H
#define AVDF_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define AVDF_OK 0
#define AVDF_EINVAL (-22)
#define AVDF_ELAUNCH (-5)

#define AVDF_ACT_NONE 0
#define AVDF_ACT_RELU 1
#define AVDF_ACT_GELU 2   /* exact erf GELU (nn.GELU default) */
#define AVDF_ACT_LEAKY 3  /* LeakyReLU(0.01) (nn.LeakyReLU default; LAV-DF visual encoder) */

const char* avdf_last_error(void);
int avdf_version(void);

/* ---- dense conv1d as implicit GEMM on exact-fp32 MFMA ------------------------------------
 * Replaces MaskedConv1D.forward (libs/modeling/blocks.py:41-63) for groups==1, every 1x1
 * nn.Conv1d / nn.Linear in the blocks (blocks.py:957-966,1291-1297,1604-1605) and the DownBlock
 * conv (blocks.py:1498-1505).
 *   y[b,t,n] = act( sum_{k<K} sum_{c<Cin} x[b, t*stride + k - K/2, c] * w[n,c,k] + bias[n] ) * mask[b,t]
 * K in {1,3}, stride in {1,2}; bias/mask/ypre may be NULL; ypre (optional) receives the
 * pre-activation (needed by the backward of act).  Cin % 4 == 0 is required.               */
int avdf_conv1d_fwd(const float* x, const float* w, const float* bias, const float* mask,
                    float* y, float* ypre, int B, int Tin, int Cin, int Tout, int Cout,
                    int K, int stride, int act, void* stream);

/* Up to 3 same-shape problems in one launch (q / k / v projections: blocks.py:1176-1178, 734-736).  Tables of n
 * device pointers (host arrays); bias / mask / ypre tables may be NULL or hold NULL entries.                  */
int avdf_conv1d_fwd_grouped(int n, const float* const* x, const float* const* w, const float* const* bias,
                            const float* const* mask, float* const* y, float* const* ypre, int B, int Tin,
                            int Cin, int Tout, int Cout, int K, int stride, int act, void* stream);
int avdf_conv1d_bwd_data_grouped(int n, const float* const* dy, const float* const* w, const float* const* mask,
                                 const float* const* ypre, float* const* dx, int B, int Tin, int Cin, int Tout,
                                 int Cout, int K, int stride, int act, void* stream);
/* accumulate != 0: one launch, float atomics into dw[i] / dbias[i]; accumulate == 0: deterministic per-problem
 * path, ws sized n * round4(avdf_conv1d_bwd_weight_ws(...)).                                                   */
int avdf_conv1d_bwd_weight_grouped(int n, const float* const* dy, const float* const* x, const float* const* mask,
                                   const float* const* ypre, float* const* dw, float* const* dbias, float* ws,
                                   int B, int Tin, int Cin, int Tout, int Cout, int K, int stride, int act,
                                   int accumulate, void* stream);

/* avdf_conv1d_fwd with the block's second residual fused into the epilogue: h = (conv(x) + bias) * mask is stored to y
 * (y may be NULL: inference), out = res + rscale[c] * rkeep[b] * h (rkeep may be NULL).  Replaces MaskedConv1D +
 * AffineDropPath + the residual add of blocks.py:1313 / 872 (`out + self.drop_path_mlp(self.mlp(self.ln2(out)) * mask)`). */
int avdf_conv1d_fwd_residual(const float* x, const float* w, const float* bias, const float* mask, const float* res,
                             const float* rscale, const float* rkeep, float* y, float* out, int B, int Tin, int Cin,
                             int Tout, int Cout, int K, int stride, void* stream);

/* K=1 / stride-1 layers: data gradient (dx[i] overwritten) AND weight / bias gradient (ACCUMULATED into dw[i] /
 * dbias[i] with float atomics, as avdf_conv1d_bwd_weight_grouped with accumulate != 0) of n <= 3 same-shape problems
 * in one launch.  Replaces the two autograd nodes of nn.Conv1d(k=1) / nn.Linear backward in the reference's blocks
 * (libs/modeling/blocks.py:53-94 MaskedConv1D).  avdf_bwd_pair_mode(0) falls back to two launches.               */
int avdf_conv1d_bwd_pair_grouped(int n, const float* const* dy, const float* const* w, const float* const* x,
                                 const float* const* mask, const float* const* ypre, float* const* dx,
                                 float* const* dw, float* const* dbias, int B, int T, int Cin, int Cout, int act,
                                 void* stream);
int avdf_bwd_pair_mode(int mode);

/* FPN1D tail of ALL pyramid levels (libs/modeling/necks.py:62-93 after the lateral convs): top-down nearest x2 add,
 * masked depthwise k3 conv and LayerNorm of every level, written into the gapped level sequence [B, sum(T_l + 1), C] the
 * shared heads run on (a zero row after each level).  lens[l] = T_l with T_l = 2 T_{l+1}; lat / mask / dw / gamma / beta:
 * host tables of n device pointers ([B,T_l,C], [B,T_l], [C,1,3], [C], [C]).  td [B,Tg,C]: workspace kept for the backward;
 * y [B,Tg,C], gmask [B,Tg]; u [B,Tg,C], mean, rstd [B,Tg] saved for the backward (all three NULL at inference).
 * Backward: dlat[l] [B,T_l,C] overwritten; ddw / dgamma / dbeta tables are ACCUMULATED into (float atomics; NULL = skip);
 * dum, dtd [B,Tg,C]: workspaces.                                                                                    */
int avdf_fpn_tail_fwd(int n, const int* lens, const float* const* lat, const float* const* mask, const float* const* dw,
                      const float* const* gamma, const float* const* beta, float* td, float* y, float* gmask, float* u,
                      float* mean, float* rstd, int B, int C, float eps, void* stream);
int avdf_fpn_tail_bwd(int n, const int* lens, const float* dy, const float* u, const float* mean, const float* rstd,
                      const float* td, const float* const* mask, const float* const* dw, const float* const* gamma,
                      float* dum, float* dtd, float* const* dlat, float* const* ddw, float* const* dgamma,
                      float* const* dbeta, int B, int C, void* stream);
/* avdf_conv1d_bwd_pair_grouped with the PRODUCER layer's activation derivative folded into the data gradient:
 * dx[i] = (g w) * xact'(xpre[i]), xpre[i] [B, T, Cin] = the pre-activation whose activation produced x[i].  For the MLP
 * of a block (fc1 -> GELU -> fc2, libs/modeling/blocks.py:1291-1297) the GELU derivative is then evaluated once per
 * element here, and fc1's own backward pair runs with act = AVDF_ACT_NONE on dx as it is.                           */
int avdf_conv1d_bwd_pair_dact(int n, const float* const* dy, const float* const* w, const float* const* x,
                              const float* const* mask, const float* const* ypre, float* const* dx,
                              float* const* dw, float* const* dbias, const float* const* xpre, int xact, int B, int T,
                              int Cin, int Cout, int act, void* stream);

/* Tuner hook: force the plan of every following launch of a kind (0 fwd, 1 dgrad: (tile 0..4, K-groups + 8*double
 * buffer); 2 wgrad: (token splits, K-groups); 4 backward pair: (dgrad tile, K-groups | wgrad splits << 3)); a = -1
 * restores table + heuristic, a = -2 heuristic only.  tools/autotune.py only.                                           */
int avdf_gemm_override(int kind, int a, int b);
/* Tuner hook (tools/plan_search.py): replace the process's AVDF_TUNED_EXTRA plan rows ("kind,K,stride,M,N,Kred,a,b;...";
 * kinds 3 / 5 / 6 = forward / backward pair / dgrad in split-bf16 mode); NULL or "" clears them.                    */
int avdf_tuned_extra(const char* rows);

/* avdf_conv1d_fwd_ex with explicit batch strides (floats): batch b reads x + b*xbs, multiplies by the weight matrix
 * w + b*wbs and writes y + b*ybs -- every head of an attention layer (Q_h.K_h^T, P_h.V_h^T) in one launch
 * (AltAttention.forward, audio_feature/emotion_audio/upstream/models/modules.py:303-316).                        */
int avdf_conv1d_fwd_strided(const float* x, const float* w, const float* bias, float* y, int B, int Tin, int Cin,
                            int Tout, int Cout, int K, int stride, int pad, int ldx, int ldw, int ldy,
                            int64_t xbs, int64_t wbs, int64_t ybs, int act, void* stream);

/* Arithmetic mode of the conv GEMMs, a bit mask: 0 = exact fp32 MFMA (default; what the parity fixtures are held
 * against); bit 0 = split-bf16 operands ("bf16x3": hi*hi + hi*lo + lo*hi on the bf16 matrix cores, fp32 accumulate,
 * ~2^-17 relative error per product) in the forward kernels; bit 1 = the same in the backward kernels (data gradient
 * of every conv, weight gradient of the K = 1 layers incl. the fused pair launch).  The switch is THREAD-LOCAL host
 * state read at launch time: it applies to the launches the calling thread issues afterwards and never to another
 * thread's.                                                                                                      */
int avdf_gemm_mode(int mode);

/* General forward form for the frozen extractors: explicit zero padding `pad` (0 = the "valid" convs of the
 * wav2vec2 feature extractor, fairseq ConvFeatureExtractionModel named at audio_feature/emotion_audio/upstream/
 * models/audio.py:12-17,66-71) and explicit row strides (floats), so x / w / y may be column slices of wider
 * matrices -- attention heads Q_h.K_h^T and P.V_h^T (modules.py:303-316) without copies.
 * (K, stride) in {(1,1),(2,2),(3,1),(3,2)}; Cin, ldx, ldw multiples of 4.                                  */
int avdf_conv1d_fwd_ex(const float* x, const float* w, const float* bias, float* y, int B, int Tin, int Cin,
                       int Tout, int Cout, int K, int stride, int pad, int ldx, int ldw, int ldy, int act,
                       void* stream);

/* dx[b,t',c] = sum_k sum_n g[b,t,n] * w[n,c,k],  t*stride + k - K/2 == t',
 * with g = dy * mask[b,t] * act'(ypre) (mask / ypre optional).  Cout % 4 == 0 required.      */
int avdf_conv1d_bwd_data(const float* dy, const float* w, const float* mask, const float* ypre,
                         float* dx, int B, int Tin, int Cin, int Tout, int Cout,
                         int K, int stride, int act, void* stream);

/* dw[n,c,k] (+)= sum_{b,t} g[b,t,n] * x[b, t*stride + k - K/2, c];  dbias[n] (+)= sum g[b,t,n].
 * `ws` is a caller-owned workspace of avdf_conv1d_bwd_weight_ws() floats; accumulate!=0 adds
 * into dw/dbias instead of overwriting.  dbias may be NULL.                                  */
int64_t avdf_conv1d_bwd_weight_ws(int B, int Tout, int Cin, int Cout, int K);
int avdf_conv1d_bwd_weight(const float* dy, const float* x, const float* mask, const float* ypre,
                           float* dw, float* dbias, float* ws, int B, int Tin, int Cin, int Tout,
                           int Cout, int K, int stride, int act, int accumulate, void* stream);

/* ---- channel LayerNorm (libs/modeling/blocks.py:97-112), rows = tokens -------------------
 * y = (x-mean)/sqrt(var+eps)*gamma+beta over C, optional fused ReLU (backbones.py:439,
 * av_fd_no_recon.py:84) and optional "+ pe[t,c]*mask[b,t]" (backbones.py:451,463; pe is the
 * [T,C] time-major sinusoid table).  mean/rstd [M] are saved for the backward (may be NULL). */
int avdf_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y,
                       float* mean, float* rstd, int M, int C, float eps, int relu,
                       const float* pe, const float* mask, int T, void* stream);
int64_t avdf_layernorm_bwd_ws(int M, int C);
int avdf_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* beta,
                       const float* mean, const float* rstd, float* dx, float* dgamma, float* dbeta,
                       float* ws, int M, int C, int relu, int accumulate, void* stream);
/* LayerNorm with a row scale folded in: y = act(LN(x)) * rowscale[row] (forward) and dy * rowscale[row] on load
 * (backward).  The shared classification / regression heads run on the gapped concatenation of the pyramid levels and
 * zero the gap rows after every LayerNorm (av_fd_no_recon.py:78-85,145-152 applied level by level in the reference). */
int avdf_layernorm_fwd_rs(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                          int M, int C, float eps, int relu, const float* rowscale, void* stream);
int avdf_layernorm_bwd_rs(const float* dy, const float* x, const float* gamma, const float* beta, const float* mean,
                          const float* rstd, float* dx, float* dgamma, float* dbeta, float* ws, int M, int C, int relu,
                          int accumulate, const float* rowscale, void* stream);

/* ---- depthwise conv k3 (+mask) -- query/key/value_conv (blocks.py:934-952) and fpn_convs
 * (necks.py:48-51): y[b,t,c] = (sum_k x[b,t*stride+k-1,c]*w[c,0,k]) * mask[b,t]            */
int avdf_dwconv_fwd(const float* x, const float* w, const float* mask, float* y,
                    int B, int Tin, int Tout, int C, int stride, void* stream);
int64_t avdf_dwconv_bwd_ws(int B, int Tout, int C);
int avdf_dwconv_bwd(const float* dy, const float* x, const float* w, const float* mask,
                    float* dx, float* dw, float* ws, int B, int Tin, int Tout, int C, int stride,
                    int accumulate, void* stream);

/* ---- banded (+-window/2) local attention == LocalMaskedMHCA steps 3-4
 * (blocks.py:1181-1221): scores = (q*scale).k over the band, -inf outside the sequence
 * (blocks.py:1006-1018), -1e4 on masked keys (blocks.py:1191-1203), softmax over the band,
 * rows of masked queries zeroed (blocks.py:1207-1209), out = P.V.
 * q,k,v,out: [B,T,H*hs] (head h = channels h*hs..), kvmask [B,T], probs [B,T,H,window].     */
int avdf_local_attn_fwd(const float* q, const float* k, const float* v, const float* kvmask,
                        float* out, float* probs, int B, int T, int H, int hs, int window,
                        float scale, void* stream);
int64_t avdf_local_attn_bwd_ws(int B, int T, int H, int window);
int avdf_local_attn_bwd(const float* dout, const float* q, const float* k, const float* v,
                        const float* probs, float* dq, float* dk, float* dv, float* ws,
                        int B, int T, int H, int hs, int window, float scale, void* stream);
/* The same with the optional per-head tables of LocalMaskedMHCA's use_rel_pe / use_time_weight (blocks.py:968-975,
 * 1188-1189, 1211-1212): rel [H, window] is added to the scores before the softmax, tw [H, window] multiplies the
 * probabilities after it.  Backward: ws receives dS [B,T,H,window] (column sums over (b,t) = d rel), dT (optional, same
 * shape) receives p * (dout . v) (column sums = d tw).                                                            */
int avdf_local_attn_fwd_ex(const float* q, const float* k, const float* v, const float* kvmask, const float* rel,
                           const float* tw, float* out, float* probs, int B, int T, int H, int hs, int window,
                           float scale, void* stream);
int avdf_local_attn_bwd_ex(const float* dout, const float* q, const float* k, const float* v, const float* probs,
                           const float* tw, float* dq, float* dk, float* dv, float* ws, float* dT, int B, int T,
                           int H, int hs, int window, float scale, void* stream);

/* ---- global masked attention == MaskedMHCA core (blocks.py:299-309), Tq,Tk <= 256 ------- */
int avdf_global_attn_fwd(const float* q, const float* k, const float* v, const float* kvmask,
                         float* out, float* probs, int B, int Tq, int Tk, int H, int hs,
                         float scale, void* stream);
int64_t avdf_global_attn_bwd_ws(int B, int Tq, int Tk, int H);
int avdf_global_attn_bwd(const float* dout, const float* q, const float* k, const float* v,
                         const float* kvmask, const float* probs, float* dq, float* dk, float* dv,
                         float* ws, int B, int Tq, int Tk, int H, int hs, float scale, void* stream);

/* ---- block glue ---------------------------------------------------------------------------
 * maxpool: nn.MaxPool1d(3, stride 2, padding 1) skip path (blocks.py:1277-1281,1311)         */
int avdf_maxpool3s2_fwd(const float* x, float* y, int B, int Tin, int C, void* stream);
int avdf_maxpool3s2_bwd(const float* dy, const float* x, float* dx, int B, int Tin, int C, void* stream);
/* out = a*ma + scale[c]*keep[b]*(r*mr): residual + AffineDropPath (blocks.py:1311-1313,
 * 1397-1439); ma/mr/keep optional.                                                           */
int avdf_affine_residual_fwd(const float* a, const float* ma, const float* r, const float* mr,
                             const float* scale, const float* keep, float* out,
                             int B, int T, int C, void* stream);
int64_t avdf_affine_residual_bwd_ws(int B, int T, int C);
/* accumulate == 2 (avdf_affine_residual_bwd, avdf_residual_ln_bwd, avdf_layernorm_bwd): the kernel leaves one row of
 * parameter partials per workgroup in ws instead of adding them by float atomics; the caller ADDS the rows
 * (avdf_reduce_segments_batched).  Rows: this function (0 = mode not available for this width) / ws floats / 3C / 2C.  */
int avdf_affine_residual_bwd_rows(int B, int T, int C);
int avdf_affine_residual_bwd(const float* dout, const float* ma, const float* r, const float* mr,
                             const float* scale, const float* keep, float* da, float* dr,
                             float* dscale, float* ws, int B, int T, int C, int accumulate, void* stream);
/* nearest resampling of k/v sources (backbones.py:487,490) and FPN top-down (necks.py:78-80) */
int avdf_upsample_fwd(const float* x, float* y, int B, int Tin, int C, int r, void* stream);
int avdf_upsample_bwd(const float* dy, float* dx, int B, int Tin, int C, int r, void* stream);
int avdf_downsample_fwd(const float* x, float* y, int B, int Tin, int C, int r, void* stream);
int avdf_downsample_bwd(const float* dy, float* dx, int B, int Tin, int C, int r, void* stream);
int avdf_add_upsampled(const float* a, const float* small, float* out, int B, int T, int C, int r, void* stream);
/* y = (x + bias[c]) * mask[row]: MaskedConvTranspose1D epilogue (blocks.py:1477-1489); its bias gradient */
int avdf_bias_mask_fwd(const float* x, const float* bias, const float* mask, float* y, int64_t M, int C, void* stream);
int64_t avdf_masked_colsum_ws(int C);
int avdf_masked_colsum(const float* dy, const float* mask, float* out, float* ws, int64_t M, int C,
                       int accumulate, void* stream);
/* y = x * s (per-row scalar s[M], e.g. dropout keep mask) */
int avdf_scale_rows(const float* x, const float* s, float* y, int M, int C, void* stream);
/* layout change at the plugin boundary: [B,C,T] <-> [B,T,C] */
int avdf_transpose(const float* x, float* y, int B, int R, int Cc, void* stream);

/* ---- InstanceNorm1d(eps, no affine, stats over all T) + LeakyReLU -- DownBlock
 * (blocks.py:1508-1516) and DeepInterpolator.conv0 (blocks.py:1600-1603)                    */
int avdf_instnorm_lrelu_fwd(const float* x, float* y, float* mean, float* rstd,
                            int B, int T, int C, float eps, float slope, void* stream);
int avdf_instnorm_lrelu_bwd(const float* dy, const float* x, const float* mean, const float* rstd,
                            float* dx, int B, int T, int C, float slope, void* stream);
/* max + avg pooling over T -> [B, 2C] (blocks.py:1610-1614) */
int avdf_maxavgpool_fwd(const float* x, float* y, int B, int T, int C, void* stream);
int avdf_maxavgpool_bwd(const float* dy, const float* x, float* dx, int B, int T, int C, void* stream);

/* ---- narrow-output conv (Cout <= 4): cls_head / offset_head (av_fd_no_recon.py:56-59,
 * 139-142,85,155-156) and Linear(H->num_classes) (blocks.py:1605):
 *   y[b,t,n] = f( (sum x*w + bias[n]) * mask[b,t] ),  f = relu(scale * .) if scale!=NULL     */
int avdf_smallconv_fwd(const float* x, const float* w, const float* bias, const float* mask,
                       const float* scale, float* y, int B, int T, int Cin, int Cout, int K, void* stream);
int64_t avdf_smallconv_bwd_ws(int B, int T, int Cin, int Cout, int K);
int avdf_smallconv_bwd(const float* dy, const float* x, const float* w, const float* y /* fwd output */,
                       const float* mask, const float* scale, float* dx, float* dw, float* dbias,
                       float* dscale, float* ws, int B, int T, int Cin, int Cout, int K,
                       int accumulate, void* stream);

/* ---- feature-length linear interpolation + channel concat (K1) --------------------------
 * DeepFakeVideoAudioDataset.__getitem__ interp block (libs/datasets/deepfake_video_audio.py:
 * 1722-1756): F.interpolate(linear, align_corners=False) of one [Tsrc, Csrc] time-major stream
 * to Tout rows, written into the time-major [Tout, Ctot] model input at channel offset `coff`
 * (call once per stream = the torch.cat of :1756).  Tsrc == Tout copies (the reference skips
 * the video stream when it already has max_seq_len rows, :1725).                            */
int avdf_interp_linear(const float* src, float* dst, int Tsrc, int Csrc, int Tout, int Ctot,
                       int coff, void* stream);
/* nb clips per launch: clip b reads src + b*src_bstride and writes dst + b*dst_bstride (strides in floats) */
int avdf_interp_linear_batched(const float* src, float* dst, int nb, int64_t src_bstride, int64_t dst_bstride,
                               int Tsrc, int Csrc, int Tout, int Ctot, int coff, void* stream);

/* ---- training targets + losses (fused forward+gradient) ---------------------------------
 * avdf_loc_assign: label_points_single_video (av_fd_no_recon.py:502-600) for every clip.
 *   points [P,4]=(t,lo,hi,stride); pmask [B,P]; gt [B,G,2] (feature-grid units); ngt [B] int32
 *   (0 => real clip: excluded from cls/reg like `vaild_idx`, :388-395,613).  Writes the targets
 *   into ws (avdf_loc_losses_ws floats) and the number of positives into num_pos[0] (device).
 *   Data-parallel callers all-reduce num_pos between the two calls (:645-648 uses the global count).
 * avdf_loc_losses: sigmoid focal loss (losses.py:5-53) with label smoothing (:654-655), DIoU
 *   (losses.py:57-117), normaliser EMA (:645-648), video-level focal loss (:688-692), weighting
 *   (:702).  state[0] = loss_normalizer (in/out).  out: [0]=cls [1]=reg [2]=video [3]=final
 *   [4]=num_pos.  dlogits [B,P] / doffsets [B,P,2] / dvideo [B] receive d(final)/d(.).        */
int64_t avdf_loc_losses_ws(int B, int P);
int avdf_loc_assign(const float* points, const float* pmask, const float* gt, const int32_t* ngt,
                    float* ws, float* num_pos, int B, int P, int G, float radius, int center_sample,
                    void* stream);
int avdf_loc_losses(const float* logits, const float* offsets, const float* pmask, const int32_t* ngt,
                    const float* video_logits, const float* ws_targets, const float* num_pos,
                    float* state, float* out, float* dlogits, float* doffsets, float* dvideo,
                    float* ws_partial, int B, int P, float label_smoothing, float loss_weight,
                    float video_weight, float momentum, void* stream);

/* L1 reconstruction loss of exp5 over the real clips (av_fd_meta_arch.py:683-686): out[0] = inv_denom *
 * sum_b wb[b] * sum|a-b|; da (optional) = wb[b]*sign(a-b)*inv_denom.  ws: avdf_l1_mean_ws() floats.     */
int64_t avdf_l1_mean_ws(void);
int avdf_l1_mean(const float* a, const float* b, const float* wb, float* out, float* da, float* ws,
                 int B, int64_t per_clip, float inv_denom, void* stream);
/* The same loss inside a captured training step (engine.TrainStep): real clips = {b : ngt[b] == 0};
 * n_real = DEVICE float[1], the number of real clips of the GLOBAL batch (avdf_count_real, then all-reduced across
 * data-parallel ranks); out[0] = this rank's share of the mean (nan when n_real == 0, like torch.mean of nothing),
 * added to final_loss[0] when given; da (optional) = d out / d a.                                        */
int avdf_count_real(const int32_t* ngt, float* out /*[1]*/, int B, void* stream);
int avdf_l1_mean_dev(const float* a, const float* b, const int32_t* ngt, const float* n_real, float* out,
                     float* final_loss, float* da, float* ws, int B, int64_t per_clip, void* stream);
/* sigmoid_focal_loss(x, t, reduction='sum') (losses.py:5-53, alpha .25, gamma 2) over n elements; dx optional.
 * Used for the frame-level loss of exp13 (av_fd_no_recon2.py:718-725).                               */
int avdf_focal_loss_sum(const float* x, const float* t, float* out, float* dx, int64_t n, void* stream);

/* ---- decode + 1-D NMS + voting, batched over clips (device) ------------------------------
 * inference_single_video (av_fd_no_recon.py:760-825) + batched_nms/NMSop/seg_voting
 * (libs/utils/nms.py:8-35,67-190) + nms_1d_cpu (libs/utils/csrc/nms_cpu.cpp:19-58).
 * Ties are ordered (score desc, candidate index asc).  Outputs per clip: up to max_seg
 * (seg[2], score, point index) + count; segments stay in feature-grid units.               */
int avdf_decode_nms(const float* points, const float* logits, const float* offsets, const float* pmask,
                    float* out_segs, float* out_scores, int32_t* out_pidx, int32_t* out_count,
                    int B, int P, float pre_nms_thresh, float duration_thresh,
                    float min_score, float iou_thr, float voting_thr, int max_seg, void* stream);

/* host-side hard NMS with the semantics of nms_1d_cpu (libs/utils/csrc/nms_cpu.cpp:19-58);
 * HOST pointers; returns kept indices (score-descending, stable).                           */
int avdf_nms1d_host(const float* segs, const float* scores, int64_t n, float iou_thr,
                    int64_t* keep, int64_t* n_keep);

/* ---- audio front-end + BYOL-A encoder glue (frozen extractor, forward only) -------------------------
 * Reference: audio_feature/content_audio/extract_audio_feature_batch.py:42-60 (torchaudio MelSpectrogram
 * 16 kHz / n_fft 1024 / hop 160 / 64 mel / 60-7800 Hz, log, PrecomputedNorm) and
 * audio_feature/content_audio/byol_a/models.py:53-85 (AudioNTT2020Task6).  The STFT (frames x DFT matrix),
 * the mel projection, conv2/conv3 (as k=3 conv1d over time on mel-im2col'd data) and the two Linear layers
 * run on avdf_conv1d_fwd; these are the HBM-bound pieces around them.  CNN activations: [M(mel), T, C].   */
int avdf_frame_window(const float* pcm, const float* window, float* frames, int64_t n_samples,
                      int n_frames, int n_fft, int hop, void* stream);   /* centre + reflect pad, [F, n_fft] */
int avdf_power_spectrum(const float* spec /*[F,2*kp]=re|im*/, float* power /*[F,kp]*/, int64_t n_frames, int kp, void* stream);
int avdf_logmel_norm(const float* mel /*[F,n_mels]*/, float* lms /*[n_mels,F]*/, int64_t n_frames, int n_mels,
                     float eps, float mean, float stdv, void* stream);
int avdf_conv2d_c1_bn_relu_pool(const float* x /*[M,T]*/, const float* w /*[C,9]*/, const float* scale,
                                const float* shift, float* out /*[M/2,T/2,C]*/, int M, int T, int C, void* stream);
int avdf_mel_im2col(const float* x /*[M,T,C]*/, float* xcol /*[M,T,3C]*/, int M, int T, int C, void* stream);
int avdf_maxpool2x2(const float* x /*[M,T,C]*/, float* out, int M, int T, int C, int t_major, void* stream);

/* Soft-NMS of the reference's native module (`nms_1d_cpu.softnms`, libs/utils/csrc/nms_cpu.cpp:67-158; method 0 hard,
 * 1 linear, 2 gaussian).  dets [n,3] = (x1, x2, decayed score) in emission order, inds = source indices.  HOST pointers. */
int avdf_softnms1d_host(const float* segs, const float* scores, int64_t n, float iou_thr, float sigma,
                        float min_score, int method, float* dets, int64_t* inds, int64_t* n_out);
/* host mirror of batched_nms (class-agnostic hard branch) + seg_voting (libs/utils/nms.py:67-190); HOST pointers */
int avdf_batched_nms_host(const float* segs, const float* scores, int64_t n, float iou_thr, float min_score,
                          int64_t max_seg, float voting_thr, float* out_segs, float* out_scores,
                          int64_t* out_idx, int64_t* n_out);

/* ---- Emotion2Vec (data2vec 2.0 audio) encoder glue, forward only -----------------------------------
 * first feature-extractor layer (Cin=1, k=10, s=5: audio.py:25-28), grouped positional conv (768 ch, 16 groups,
 * k=19, same padding: audio.py:79-102) and the in-place softmax with the symmetric ALiBi bias
 * softmax_j(S*qk_scale - slope*|i-j|) (base.py:538-642; modules.py:303-316).  scores: [T, ld] per head,
 * columns >= T are zeroed.                                                                             */
int avdf_wave_conv_c1(const float* pcm, const float* w, float* y, int64_t n_samples, int64_t Tout, int C,
                      int K, int stride, void* stream);
int avdf_grouped_conv_pos(const float* x, const float* w, const float* bias, float* y, int T, int C,
                          int groups, int K, void* stream);
/* avdf_grouped_conv_pos on the MFMA GEMM (K = 19): every channel group is one batch entry of the strided forward kernel. */
int avdf_grouped_conv1d_fwd(const float* x, const float* w, const float* bias, float* y, int T, int C, int groups, int K,
                            void* stream);
int avdf_softmax_alibi(float* scores, int T, int ld, float qk_scale, float slope, void* stream);
/* all H heads of a layer in one launch: scores [H][T][ld] with `head_stride` floats between heads, slopes [H] (device) */
int avdf_softmax_alibi_heads(float* scores, int H, int T, int ld, int64_t head_stride, float qk_scale,
                             const float* slopes, void* stream);

/* Fused attention of one Emotion2Vec layer (AltAttention.forward, upstream/models/modules.py:280-322; ALiBi of
 * base.py:538-604): out = softmax(q k^T * scale - slopes[h] * |i - j|) v for all H heads in one launch; the
 * [H, T, T] scores stay on chip (exact-fp32 MFMA, online softmax).  q / k / v / out: [B][rows][ld] fp32, head h =
 * columns [h*hd, (h+1)*hd), hd = 64; rows >= T are ignored (out rows >= T are not written).  slopes: device [H] or NULL. */
int avdf_flash_attn_alibi(const float* q, const float* k, const float* v, float* out, const float* slopes, int B,
                          int T, int H, int hd, int ld, int64_t clip_stride, float scale, void* stream);

/* ---- fused AffineDropPath residual + LayerNorm (TransformerBlock: blocks.py:1311-1313, 870-872) ------------
 *   out = a*ma[b,t] + scale[c]*keep[b]*(r*mr[b,t]);  h = LN(out)*gamma + beta  (ma / mr / keep may be NULL)
 * backward: dout = gradient reaching `out` from its other consumer (NULL = none), dh = gradient of h.        */
int avdf_residual_ln_fwd(const float* a, const float* ma, const float* r, const float* mr, const float* scale,
                         const float* keep, const float* gamma, const float* beta, float* out, float* h,
                         float* mean, float* rstd, int B, int T, int C, float eps, void* stream);
int64_t avdf_residual_ln_bwd_ws(int B, int T, int C);
int avdf_residual_ln_bwd(const float* dout, const float* dh, const float* out, const float* mean,
                         const float* rstd, const float* gamma, const float* ma, const float* r,
                         const float* mr, const float* scale, const float* keep, float* da, float* dr,
                         float* dscale, float* dgamma, float* dbeta, float* ws, int B, int T, int C,
                         int accumulate, void* stream);

/* ---- fused q/k/v input chains of the conv-attention blocks ---------------------------------------------
 * y_i = LN_b,i( dwconv_k3,stride s( LN_a,i(x_i) ) * mask ), i = 0..2 (query / key / value), one launch:
 * ln1 | lnq,lnk,lnv -> query_conv/key_conv/value_conv -> query_norm/key_norm/value_norm
 * (libs/modeling/blocks.py:1307-1309, 866-868, 239-262, 931-955, 1166-1173).  Every table has 3 entries
 * (inputs / parameters may alias).  x_i [B,Tin,C]; mask [B,Tout] or NULL; u_i (conv output, kept for the
 * backward) and y_i [B,Tout,C]; stats_a_i [2][B*Tin], stats_b_i [2][B*Tout] = (mean, rstd).  C % 4 == 0, C <= 256. */
int avdf_chain_fwd(const float* const* x, const float* const* gamma_a, const float* const* beta_a,
                   const float* const* w, const float* const* gamma_b, const float* const* beta_b,
                   const float* mask, float* const* u, float* const* y, float* const* stats_a,
                   float* const* stats_b, int B, int Tin, int Tout, int C, int stride, float eps, void* stream);
/* Backward.  dx_i is written when bit i of store_mask is set and then holds the SUM over the branches since the
 * previous set bit (branches that share an input: 0b100 = one shared input, 0b101 = x0 | x1==x2, 0b111 = all
 * distinct).  Parameter partials: ws[avdf_chain_bwd_rows][3][7*C] = per branch dgamma_a, dbeta_a, dw ([C][3]),
 * dgamma_b, dbeta_b -- fold them with avdf_reduce_segments.                                                */
int64_t avdf_chain_bwd_ws(int B, int Tin, int C);
int avdf_chain_bwd_rows(int B, int Tin);
int avdf_chain_bwd(const float* const* dy, const float* const* u, const float* const* stats_b,
                   const float* const* x, const float* const* stats_a, const float* const* gamma_a,
                   const float* const* beta_a, const float* const* w, const float* const* gamma_b,
                   const float* mask, float* const* dx, int store_mask, float* ws, int B, int Tin, int Tout,
                   int C, int stride, void* stream);
/* avdf_chain_bwd with dres[i] (table or entries may be NULL): a [B,Tin,C] gradient reaching the input closed by branch i
 * through the block's residual connection (blocks.py:1311 `out = pool_skip(x) * mask + drop_path(out)`); dx[i] gets the
 * sum, which replaces autograd's separate accumulation kernel.                                                   */
int avdf_chain_bwd_res(const float* const* dy, const float* const* u, const float* const* stats_b,
                       const float* const* x, const float* const* stats_a, const float* const* gamma_a,
                       const float* const* beta_a, const float* const* w, const float* const* gamma_b,
                       const float* mask, float* const* dx, const float* const* dres, int store_mask, float* ws,
                       int B, int Tin, int Tout, int C, int stride, void* stream);
/* avdf_chain_fwd / avdf_chain_bwd_res with a nearest resampling of each branch's INPUT folded into the row address (the
 * up/down-sampling in front of the k / v chains of the cross-level blocks, backbones.py:487-490): xshift[i] > 0: x[i] is
 * [B, Tin >> s, C] and row t reads source row t >> s; xshift[i] < 0: x[i] is [B, Tin << -s, C] and row t reads source row
 * t << -s; 0: as is.  The backward returns dx[i] at the chain's resolution [B, Tin, C]; fold it back with
 * avdf_upsample_bwd / avdf_downsample_bwd.                                                                        */
int avdf_chain_fwd_map(const float* const* x, const float* const* gamma_a, const float* const* beta_a,
                       const float* const* w, const float* const* gamma_b, const float* const* beta_b,
                       const float* mask, float* const* u, float* const* y, float* const* stats_a,
                       float* const* stats_b, const int* xshift, int B, int Tin, int Tout, int C, int stride, float eps,
                       void* stream);
/* 1 if avdf_chain_bwd_map hands back dx of a branch with input shift `xshift` at the SOURCE resolution [B, Tin >> xshift, C]
 * (groups of 2^xshift rows summed inside the kernel), 0 if at the chain's resolution [B, Tin, C].              */
int avdf_chain_bwd_folds(int xshift, int stride);
int avdf_chain_bwd_map(const float* const* dy, const float* const* u, const float* const* stats_b,
                       const float* const* x, const float* const* stats_a, const float* const* gamma_a,
                       const float* const* beta_a, const float* const* w, const float* const* gamma_b,
                       const float* mask, float* const* dx, const float* const* dres, int store_mask, float* ws,
                       const int* xshift, int B, int Tin, int Tout, int C, int stride, void* stream);
/* dst[s][i] (+)= sum over rows r < nrow and sources j < nsrc[s] of ws[r*row_stride + src_off[s*3+j] + i],
 * i < len[s]; up to 16 segments per launch (host arrays).                                                   */
int avdf_reduce_segments(const float* ws, int nrow, int row_stride, int nseg, float* const* dst, const int* len,
                         const int* nsrc, const int* src_off, int accumulate, void* stream);
/* njobs ACCUMULATING avdf_reduce_segments jobs in ceil(njobs / 10) launches (the engine flushes the parameter partials of
 * every LN -> dw-conv -> LN chain backward of a backward phase together).  ws / nrow / stride / nseg: one entry per job;
 * dst / len / nsrc (one entry per segment) and src_off (three) are the jobs' segment tables concatenated.            */
int avdf_reduce_segments_batched(int njobs, const float* const* ws, const int* nrow, const int* stride, const int* nseg,
                                 float* const* dst, const int* len, const int* nsrc, const int* src_off, void* stream);

/* ---- LAV-DF visual encoder (C3D; frozen extractor, forward only) -------------------------------------
 * Reference: video_feature/aligned_video/model/video_encoder.py:27-62 (7x Conv3d 3x3x3 pad 1 + LeakyReLU,
 * MaxPool3d((1,p,p))) and extract_video_features.py:19-37 (/255, tail zero pad, nearest resize to 96x96).
 * Activations are voxel-major [B, T, H, W, C]; weights tap-major wp[27][Cout][Cin] (tap = (dt*3+dh)*3+dw,
 * packed once from the reference [Cout, Cin, 3, 3, 3]).                                                  */
int avdf_conv3d_k3(const float* x, const float* wp, const float* bias, float* y, int B, int T, int H, int W,
                   int Cin, int Cout, int act, void* stream);
/* 0 = exact fp32 MFMA (default), 1 = split-bf16 operands for the layers with Cin >= 16 (~5e-6 relative error).
 * Per calling thread (thread-local host switch read at launch time), like avdf_gemm_mode.                */
int avdf_conv3d_mode(int mode);
/* uint8 frames [T, H0, W0, 3] -> fp32 [Tp, S, S, 4] (/255, nearest resize, 4th channel and frames >= T zero) */
int avdf_video_prep(const uint8_t* frames, float* out, int T, int H0, int W0, int Tp, int S, void* stream);
/* [N, H, W, C] -> [N, H/p, W/p, C] max over p x p windows (floor mode) */
int avdf_maxpool_hw(const float* x, float* y, int64_t N, int H, int W, int C, int p, void* stream);

/* ---- pyramid glue (replaces torch.cat / index_select / relu / mul and their autograd backward on the heads' path;
 * av_fd_no_recon.py:75-89,144-159; Scale: blocks.py:1374-1392; mask down-sampling: blocks.py:51-55) ----------------
 * Gapped sequence of n_levels (<= 8) pyramid levels: level l (lens[l] rows) is followed by ONE all-zero row, the zero
 * padding of that level's k3 convs, so the shared heads run once over [B, sum(lens) + n_levels, C].
 * x / masks / dx / scale / dscale are HOST arrays of n_levels device pointers.                                  */
int avdf_gap_concat_fwd(const float* const* x /*[B,lens[l],C]*/, const float* const* masks /*[B,lens[l]] or NULL*/,
                        const int* lens, int n_levels, float* out /*[B,Tg,C]*/, float* mask_out /*[B,Tg] or NULL*/,
                        int B, int C, void* stream);
int avdf_gap_concat_bwd(const float* dout, float* const* dx /*NULL entries are skipped*/, const int* lens, int n_levels,
                        int B, int C, void* stream);
/* Compaction of a head output on the gapped sequence to point order [B, P, NC] (P = sum lens), with the regression
 * head's per-level Scale and ReLU: out = relu?(y * scale_l[0]).  scale == NULL: plain compaction (cls head).
 * bwd: dy (gap rows 0) and, when given, dscale_l[0] += sum dout * [out > 0] * y (atomics: dscale accumulates).     */
int avdf_head_out_fwd(const float* y /*[B,Tg,NC]*/, const float* const* scale, const int* lens, int n_levels,
                      float* out, int B, int NC, int relu, void* stream);
int avdf_head_out_bwd(const float* dout, const float* y, const float* out, const float* const* scale,
                      float* const* dscale, const int* lens, int n_levels, float* dy, int B, int NC, int relu,
                      void* stream);
/* out[l][b, t] = mask[b, t * stride^(l+1)], l = 0 .. n_levels-1: every pyramid mask of a forward in one launch */
int avdf_mask_pyramid(const float* mask /*[B,T]*/, float* const* out, int n_levels, int B, int T, int stride,
                      void* stream);

/* Keep factors of stochastic depth / dropout from uniform draws u in [0,1) (the draws themselves stay torch's Philox
 * stream, which is hipGraph-safe): mode 0 = AffineDropPath (blocks.py:1403-1409), out[r,b] = floor(keep_prob[r] +
 * u[r,b]) / keep_prob[r] over n = rows x cols elements; mode 1 = nn.Dropout(p), out[i] = u[i] >= p ? 1/(1-p) : 0.  */
int avdf_keep_from_uniform(const float* u, const float* keep_prob, float* out, int64_t n, int cols, int mode, float p,
                           void* stream);

/* ---- optimiser step on flat buffers (next-row: train_utils.py:65-144,247-269,306-316) ---- */
int64_t avdf_sumsq_ws(void);
int avdf_sumsq(const float* x, float* out /*[1], accumulates*/, float* ws, int64_t n, void* stream);
int avdf_sumsq_ex(const float* x, float* out /*[1]*/, float* ws, int64_t n, int accumulate /*0: out is overwritten*/,
                  void* stream);
/* hp: DEVICE float[3] = {lr, 1-beta1^t, 1-beta2^t}, rewritten by the host each step (graph-safe schedule);
 * gnorm_sq: device float[1] = sum of squared gradients over ALL parameters (NULL or clip_norm<=0: no clipping) */
int avdf_adamw_ema(float* p, const float* g, float* m, float* v, float* ema, const float* gnorm_sq,
                   const float* hp, int64_t n, float beta1, float beta2, float eps, float wd,
                   float clip_norm, float ema_decay, void* stream);
/* One launch over nseg (<= 8) consecutive ranges of the flat buffers, range i = [seg_end[i-1], seg_end[i]) with weight
 * decay seg_wd[i] (HOST arrays; seg_end[nseg-1] == n): the decay / no-decay groups of make_optimizer
 * (train_utils.py:125-144) of every gradient bucket in a single pass.  zero_grad != 0: g is cleared after use (the
 * model.zero_grad() of the next iteration, train_utils.py:302, without a separate pass).                 */
int avdf_adamw_ema_seg(float* p, const float* g, float* m, float* v, float* ema, const float* gnorm_sq,
                       const float* hp, int64_t n, int nseg, const int64_t* seg_end, const float* seg_wd,
                       float beta1, float beta2, float eps, float clip_norm, float ema_decay, int zero_grad,
                       void* stream);

#ifdef __cplusplus
}
#endif
#endif
