/* libtatt_hip.so — C ABI of the MI355X-native TATT super-resolution hot path.
 *
 * Drop-in boundary: the reference is pure Python on stock torch ops; it has no FFI of its
 * own.  Each entry point below replaces the stock op(s) the reference reaches through torch.nn at the cited
 * reference file:line (paths relative to the reference tree).  Conventions:
 *   - every pointer is a DEVICE pointer to fp32 unless noted; the caller owns all memory, incl. workspaces;
 *   - the library never allocates, frees or synchronises; all work is enqueued on `stream`;
 *   - return 0 on success, non-zero otherwise (message via tatt_last_error(), thread-local);
 *   - activations are NHWC ("rows x channels": row = (batch, y, x)); `ld*` = floats between rows;
 *   - dropout sites take (seed_site, thresh, scale): keep iff (hash(seed_site, index) >> 8) >= thresh,
 *     kept values are multiplied by scale = 1/(1-p); thresh == 0 disables dropout; seed_dev (optional device
 *     pointer) adds seed_dev[0] to the seed at run time so a captured hipGraph draws new masks per replay.
 * tatt_amd/lib.py parses this header to build the ctypes signatures: keep one declaration per statement and
 * use only the scalar types  int, long, float, uint32_t, size_t, hipStream_t; every pointer argument (whatever it points to) is
 * passed as void*.  Arguments documented as HOST arrays (the grouped entry points) are read during the call, not later.
 */
#ifndef TATT_HIP_H
#define TATT_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifndef __HIP_PLATFORM_AMD__
typedef void* hipStream_t;
#endif

#define TATT_RELU 1
#define TATT_ACCUMULATE 1
#define TATT_BF16 2
#define TATT_F32X3 4
#define TATT_X_BF16 8
#define TATT_DY_BF16 16

#ifdef __cplusplus
extern "C" {
#endif

const char* tatt_last_error(void);
int tatt_version(void);

/* ---- fp32 MFMA GEMM: every nn.Linear / 1x1 conv / attention projection / GRU input projection ----------
 * C[m*ldc+n] = sum_k A[m*sam+k*sak] * B[n*sbn+k*sbk] (+bias[n]) (+C) (relu).  `flags` (here and in the convolution entry
 * points) is a bit set, per call -- the library keeps no precision state:
 *     TATT_RELU (1)        tatt_gemm / tatt_conv_fwd*: ReLU epilogue
 *     TATT_ACCUMULATE (1)  tatt_conv_wgrad / tatt_conv_skinny_fwd: add to the destination
 *     TATT_BF16 (2)        operands rounded to bf16 (nearest-even) on their way into v_mfma_f32_32x32x16_bf16, fp32 accumulation
 *                          (BASELINE.json config 3; the reference itself has no autocast).  Default = exact fp32 operands on
 *                          v_mfma_f32_32x32x2_f32, the parity path.  Launches whose shapes take the generic (non-FAST) loaders
 *                          stay fp32 either way; tensors in memory are fp32 unless TATT_X_BF16 / TATT_DY_BF16 say otherwise.
 *     TATT_F32X3 (4)       opt-in: fp32 products on the bf16 matrix core.  Every operand element is split exactly into three
 *                          bf16 pieces (3 x 8 significant bits) on its way into LDS and a*b is accumulated in fp32 as six
 *                          v_mfma_f32_32x32x16_bf16 terms (a0b0 + a0b1 + a1b0 + a0b2 + a1b1 + a2b0; what is dropped is below the
 *                          rounding of one fp32 product).  Error against fp64 <= that of the fp32 FMA chain (tests/test_gpu_f32x3.py);
 *                          3x3 64->64 convolution 34 us instead of 49.  Only launches with k-contiguous operands (NT GEMMs, implicit-GEMM
 *                          convolution forward / dgrad) have such a kernel; any other launch carrying the flag runs on fp32 MFMA.
 *                          TATT_BF16 wins if both bits are set.
 *     TATT_X_BF16 (8)      tatt_conv_fwd* / tatt_conv_fwd_bn / tatt_conv_wgrad, together with TATT_BF16: X points at bf16 elements
 *     TATT_DY_BF16 (16)    tatt_conv_wgrad, together with TATT_BF16: dY points at bf16 elements.
 *                          bf16 STORAGE (BASELINE.json config 3 "bf16"): in the bf16 step every tensor whose only
 *                          readers are bf16 matrix-core launches -- mish(BN(conv1)) of each SRB, block7's output + skip, and the
 *                          gradients arriving at the 3x3 convolutions' outputs -- is written once as bf16 by its producer
 *                          (tatt_bn_apply_bf16, tatt_bn_bwd_from_partial_bf16, tatt_pixshuf_mish_bwd_bf16) and read as bf16: the same
 *                          numbers the launch would have rounded to, half the bytes.  ldx / lddy stay in elements; supported: the
 *                          convolution input (Cin % 32 == 0, packed weights), weight gradients of more than 16 output channels with
 *                          dY, or X and dY, in bf16 (refused with TATT_ERR_ARG otherwise).
 * Replaces F.linear at
 * model/tsrn.py:213, :1083 (1x1 conv), model/transformer_v2.py:476,:480,:819,:829 (MHA in/out proj, FFN),
 * nn.GRU input projections (model/tsrn.py:1088, model/transformer_v2.py:218), stn_head.py:99-100. */
size_t tatt_gemm_workspace(int M, int N, int splitk);
int tatt_gemm(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, long sam, long sak, long sbn, long sbk, long ldc, int batch, long sAb, long sBb, long sCb, int accumulate, int flags, int splitk, float* workspace, hipStream_t stream);

/* tatt_gemm over a k-concatenated A operand and/or into two destinations, without materialising either concatenation:
 *   A(m,k) = k < ksplit ? A[m*sam + k] : A2[m*sam + k - ksplit]      (A2 = NULL: single source; ksplit multiple of 32)
 *   column n < nsplit -> C[m*ldc + n] (accumulate), else C2[m*ldc2 + n - nsplit] (accumulate2)   (C2 = NULL: single destination; nsplit
 *   multiple of 64).  k-contiguous A rows; B(n,k) = B[n*sbn + k*sbk]; operands 16-byte aligned, K and strides multiples of 4.
 * RecurrentResidualBlockTL feeds its first GruBlock cat(conv features, text-prior map) (model/tsrn.py:893-895, torch.cat +
 * the 1x1 conv at :1083) and its backward pass splits the gradient again: both directions run as one launch each. */
int tatt_gemm_cat(const float* A, const float* A2, int ksplit, const float* B, float* C, float* C2, int nsplit, const float* bias, int M, int N, int K, long sam, long sbn, long sbk, long ldc, long ldc2, int accumulate, int accumulate2, int flags, hipStream_t stream);

/* GruBlock (model/tsrn.py:1075-1092) = Conv2d 1x1 (:1083) directly followed by nn.GRU (:1088): the GRU's input projection of both
 * directions composes with the conv, gi = (W_ih W_c) x + (W_ih b_c + b_ih).  tatt_gru_fold_w builds Wf[192][Cin] (followed in the
 * same buffer by WcT[Cin][64] = W_c transposed, which tatt_gru_unfold_grad reads: Wf buffers hold 256*Cin floats) / bf[192] for
 * n <= 16 blocks in one launch (HOST arrays of length n; Wih / bih hold 2n pointers: forward, reverse per block);
 * tatt_gru_unfold_grad adds the gradient of (Wf, bf) onto the reference's parameters: dW_ih += dWf W_c^T + dbf b_c^T,
 * db_ih += dbf, dW_c += W_ih^T dWf, db_c += W_ih^T dbf; tatt_gru_unfold_grad_group: the same for n <= 16 blocks in one launch (HOST arrays
 * of length n; Wih / dWih / dbih hold 2n pointers). */
int tatt_gru_fold_w(int n, const float* const* Wc, const float* const* bc, const float* const* Wih, const float* const* bih, float* const* Wf, float* const* bf, const int* Cin, hipStream_t stream);
int tatt_gru_unfold_grad_group(int n, const float* const* dWf, const float* const* dbf, const float* const* WcT, const float* const* bc, const float* const* Wih, float* const* dWih, float* const* dbih, float* const* dWc, float* const* dbc, const int* Cin, hipStream_t stream);
int tatt_gru_unfold_grad(const float* dWf, const float* dbf, const float* WcT, const float* bc, const float* Wih_f, const float* Wih_b, float* dWih_f, float* dWih_b, float* dbih_f, float* dbih_b, float* dWc, float* dbc, int Cin, hipStream_t stream);

/* Weight + bias gradient of a Linear / 1x1 conv with a small output (M <= 96, N <= 64): dW[M,N] += dY[R,M]^T X[R,N],
 * db[M] += colsum(dY) in ONE streaming pass over the R rows (autograd of F.linear at the sites above). */
size_t tatt_tn_small_workspace(int M, int N);
int tatt_tn_small(const float* dY, long lddy, const float* X, long ldx, float* dW, long lddw, float* db, int M, int N, int R, float* workspace, hipStream_t stream);
/* Grouped form: nprob <= 8 independent (dY, X, dW, db, M, N, R) problems in ONE launch pair, e.g. the weight gradients of a BiGRU
 * block (nn.GRU W_ih / W_hh of both directions + the 1x1 conv in front, model/tsrn.py GruBlock).  All array arguments are HOST
 * arrays of length nprob (db[p] may be NULL); workspace: tatt_tn_small_workspace(96, 64) floats.  dY2 / lddy2 / msplit (all NULL, or
 * arrays of length nprob with dY2[p] NULL where unused): columns m >= msplit[p] (a multiple of 32) of problem p's dY are read from
 * dY2[p] at column m - msplit[p] -- the W_hh gradient takes its r / z gate columns from dgi and the n gate from tatt_gru32_bwd's dghn. */
int tatt_tn_small_group(int nprob, const float* const* dY, const long* lddy, const float* const* X, const long* ldx, float* const* dW, const long* lddw, float* const* db, const int* M, const int* N, const int* R, const float* const* dY2, const long* lddy2, const int* msplit, float* workspace, hipStream_t stream);

/* ---- NHWC implicit-GEMM convolution (stride 1, "same" zero padding), fwd / dgrad / wgrad -------------------
 * Replaces nn.Conv2d at model/tsrn.py:598 (9x9 4->64), :868,:876 (SRB 3x3), :614, :1034 (64->256), :625
 * (9x9 64->4) and model/stn_head.py:15.  Weights stay in the reference's OIHW layout; (w_sn,w_sci,w_base,
 * w_stap) describe how (cout, cin, tap) is addressed: forward (Cin*T, T, 0, 1); dgrad (T, Cin_f*T, T-1, -1). */
int tatt_conv_fwd(const float* X, const float* Wt, const float* bias, float* Y, int B, int H, int W, int Cin, int Cout, int KH, int KW, long ldx, long ldy, long w_sn, long w_sci, long w_base, long w_stap, int accumulate, int flags, hipStream_t stream);
/* tatt_conv_fwd with the KH*KW*Cin reduction split `splitk` ways (workspace: splitk*B*H*W*Cout floats): the small feature
 * maps of the STN head (model/stn_head.py:15, 4x8 .. 2x2 pixels, K up to 2304) have too few output tiles to fill the chip. */
int tatt_conv_fwd_splitk(const float* X, const float* Wt, const float* bias, float* Y, int B, int H, int W, int Cin, int Cout, int KH, int KW, long ldx, long ldy, long w_sn, long w_sci, long w_base, long w_stap, int accumulate, int flags, int splitk, float* workspace, hipStream_t stream);
/* Per-step weight repack for tatt_conv_fwd (same nn.Conv2d weights as above): Wf[co][tap][ci] for the forward pass,
 * strides (T*Cin, 1, 0, Cin), and Wd[ci][tap][co] with the taps already flipped for dgrad, strides (T*Cout, 1, 0, Cout):
 * the kernel's weight-tile loads become 16-byte contiguous instead of stride-T gathers. */
int tatt_conv_pack_w(const float* Wt, float* Wf, float* Wd, int Cout, int Cin, int KH, int KW, hipStream_t stream);
/* The same for n <= 32 layers in one launch (host arrays of length n; KHW[i] = KH*KW of layer i). */
int tatt_conv_pack_w_group(int n, const float* const* Wt, float* const* Wf, float* const* Wd, const int* Cout, const int* Cin, const int* KHW, hipStream_t stream);
/* Cout in {4,8,16}, instead of padding Cout to the 64-wide MFMA tile.  The 9x9 convolutions with FOUR output channels (the 64->4
 * output conv model/tsrn.py:625 and the dgrad of the 9x9 4->64 conv :598; Cin a multiple of 8) run as a direct VALU convolution: an
 * 8x64-pixel tile + halo of eight input channels at a time in LDS, no intermediate; workspace = the packed weights only, and the call
 * needs ldx % 4 == 0 and a 16-byte aligned X (refused with TATT_ERR_ARG otherwise).  Every other shape: dense planar GEMM
 * Z[(tap,co)][pixel] into the workspace followed by a shifted gather-sum. */
size_t tatt_conv_skinny_workspace(int B, int H, int W, int Cin, int Cout, int KH, int KW);
int tatt_conv_skinny_fwd(const float* X, const float* Wt, const float* bias, float* Y, int B, int H, int W, int Cin, int Cout, int KH, int KW, long ldx, long ldy, long w_sn, long w_sci, long w_base, long w_stap, int flags, float* workspace, hipStream_t stream);
size_t tatt_conv_wgrad_workspace(int Cin, int Cout, int KH, int KW, int splitk);
int tatt_conv_wgrad(const float* X, const float* dY, float* dW, int B, int H, int W, int Cin, int Cout, int KH, int KW, long ldx, long lddy, int splitk, float* workspace, int flags, hipStream_t stream);

/* ---- BatchNorm2d/1d (model/tsrn.py:869,:877,:615; model/stn_head.py:19,:51) ------------------------------- */
size_t tatt_bn_workspace(int C);
int tatt_bn_stats(const float* x, long ldx, int R, int C, float eps, float momentum, float* stat, float* run_mean, float* run_var, float* workspace, hipStream_t stream);
int tatt_bn_stat_from_running(const float* rm, const float* rv, int C, float eps, float* stat, hipStream_t stream);
int tatt_bn_apply(const float* x, long ldx, const float* stat, const float* gamma, const float* beta, const float* res, long ldr, float* y, long ldy, int R, int C, int act, hipStream_t stream);
/* tatt_bn_apply with y stored as bf16 (round to nearest even; ldy in elements; y 8-byte aligned) -- see TATT_X_BF16 */
int tatt_bn_apply_bf16(const float* x, long ldx, const float* stat, const float* gamma, const float* beta, const float* res, long ldr, uint16_t* y, long ldy, int R, int C, int act, hipStream_t stream);
int tatt_bn_bwd(const float* x, long ldx, const float* dy, long lddy, const float* stat, const float* gamma, const float* beta, float* dx, long lddx, float* dgamma, float* dbeta, int R, int C, int act, int train, int accumulate_params, int accumulate_dx, float* workspace, hipStream_t stream);

/* BatchNorm fused into the producing GEMM / convolution (model/tsrn.py:868-877 conv-BN pairs and their autograd): the first half of
 * each BatchNorm pass -- the row reduction -- runs in the epilogue of the launch that produces the tensor, per 128-row block:
 *   tatt_conv_fwd_bn(epi = 1): Y = conv(X) and partial[blk][2][Cout] = (mean, M2) of Y's rows -> tatt_bn_stats_from_partial
 *       (rows_per_blk = 128) = the statistics half of tatt_bn_stats;
 *   tatt_conv_fwd_bn(epi = 2) / tatt_gemm_cat_bnbwd: the launch computes dy, the gradient wrt y = act(BatchNorm(bn_x)); it stores
 *       g = dy * act'(bn(bn_x)) instead and partial = (sum g, sum g xhat) -> tatt_bn_bwd_from_partial = the rest of tatt_bn_bwd.
 * partial: cdiv(rows, 128) * 2 * C floats; weights packed (tatt_conv_pack_w); operand layouts as the FAST loaders need them
 * (16-byte granular), else the call is refused.  tatt_gemm_cat_bnbwd: the BatchNorm (bn_C channels) sits on the first destination. */
int tatt_conv_fwd_bn(const float* X, const float* Wt, const float* bias, float* Y, int B, int H, int W, int Cin, int Cout, int KH, int KW, long ldx, long ldy, long w_sn, long w_sci, long w_base, long w_stap, int flags, int epi, const float* bn_x, long bn_ldx, const float* bn_stat, const float* bn_gamma, const float* bn_beta, int bn_act, float* partial, hipStream_t stream);
int tatt_gemm_cat_bnbwd(const float* A, const float* A2, int ksplit, const float* B, float* C, float* C2, int nsplit, const float* bias, int M, int N, int K, long sam, long sbn, long sbk, long ldc, long ldc2, int accumulate, int accumulate2, int flags, const float* bn_x, long bn_ldx, const float* bn_stat, const float* bn_gamma, const float* bn_beta, int bn_act, int bn_C, float* partial, hipStream_t stream);
int tatt_bn_stats_from_partial(const float* partial, int nblk, int rows_per_blk, int R, int C, float eps, float momentum, float* stat, float* run_mean, float* run_var, hipStream_t stream);
/* The batched TSC step pushes both SR passes through one convolution launch but keeps BatchNorm per pass: the per-pass
 * finals and applies as ONE launch each -- ng groups of R rows, partial [ng][nblk][2][C], stat [ng][2][C], running statistics updated
 * group after group; apply: row r uses stat[r / Rg] (Rg a multiple of the apply kernel's rows per block: 8192 / C for C <= 1024). */
int tatt_bn_stats_from_partial_groups(const float* partial, int nblk, int rows_per_blk, int R, int C, int ng, float eps, float momentum, float* stat, float* run_mean, float* run_var, hipStream_t stream);
int tatt_bn_apply_groups(const float* x, long ldx, const float* stat, const float* gamma, const float* beta, const float* res, long ldr, float* y, long ldy, int R, int C, int act, int Rg, hipStream_t stream);
int tatt_bn_apply_bf16_groups(const float* x, long ldx, const float* stat, const float* gamma, const float* beta, const float* res, long ldr, uint16_t* y, long ldy, int R, int C, int act, int Rg, hipStream_t stream);
int tatt_bn_bwd_from_partial(const float* x, long ldx, const float* g, long ldg, const float* stat, const float* gamma, const float* beta, const float* partial, int nblk, float* dx, long lddx, float* dgamma, float* dbeta, int R, int C, int train, int accumulate_params, int accumulate_dx, float* sums, hipStream_t stream);
/* the same with dx stored as bf16 (no accumulation onto dx): the gradient that only the convolution's dgrad / wgrad launches read */
int tatt_bn_bwd_from_partial_bf16(const float* x, long ldx, const float* g, long ldg, const float* stat, const float* gamma, const float* beta, const float* partial, int nblk, uint16_t* dx, long lddx, float* dgamma, float* dbeta, int R, int C, int train, int accumulate_params, float* sums, hipStream_t stream);

/* ---- LayerNorm(64) with fused residual + dropout (model/transformer_v2.py:478-482, :827-831, :380) -------- */
int tatt_ln_fwd(const float* x, const float* res, uint32_t seed_site, uint32_t thresh, float scale, const uint32_t* seed_dev, const float* gamma, const float* beta, float* vout, float* y, int rows, float eps, hipStream_t stream);
size_t tatt_ln_workspace(int rows);
int tatt_ln_bwd(const float* v, const float* dy, float dyscale, const float* gamma, float* dv, int dv_accumulate, float* dres, uint32_t seed_site, uint32_t thresh, float scale, const uint32_t* seed_dev, float* dgamma, float* dbeta, int accumulate_params, int rows, float eps, float* workspace, hipStream_t stream);

/* ---- BiGRU(64 -> 2x32) recurrence of GruBlock (model/tsrn.py:1080,:1088), persistent half-wave per sequence - */
int tatt_gru32_fwd(const float* gi, const float* whh_f, const float* whh_b, const float* bhh_f, const float* bhh_b, float* out, float* save, int nseq, int T, int inner, long outer_stride, long inner_stride, long tstride, hipStream_t stream);
/* the same recurrence, and out2 [P,64] = res [P,64] + out: the sum of SRB input and first GruBlock that the second GruBlock reads
 * (model/tsrn.py:901 "x + residual") written by the recurrence itself instead of a 16.8 MB element-wise launch */
int tatt_gru32_fwd_res(const float* gi, const float* whh_f, const float* whh_b, const float* bhh_f, const float* bhh_b, float* out, float* save, const float* res, float* out2, int nseq, int T, int inner, long outer_stride, long inner_stride, long tstride, hipStream_t stream);
/* dgi [P,192] = gradient of the input projections (dr, dz, dn per direction); dghn [P,64] (dir*32 + j) = dn * r, the one gate where
 * the gradient of W_hh h + b_hh differs from dgi; hprev [P,64] = the state each step started from. */
int tatt_gru32_bwd(const float* dout, const float* out, const float* save, const float* whh_f, const float* whh_b, float* dgi, float* dghn, float* hprev, int nseq, int T, int inner, long outer_stride, long inner_stride, long tstride, hipStream_t stream);

/* ---- recurrent positional encoding GRU(1024 -> 2x512) gates (model/transformer_v2.py:177,:215-221) -------- */
int tatt_rpe_gate_fwd(const float* gi, const float* gh, const float* bhh_f, const float* bhh_b, const float* hprev, float* hout, float* qp, int s, int B, long dir_stride_gh, long dir_stride_h, hipStream_t stream);
int tatt_rpe_gate_bwd(const float* gi, const float* gh, const float* bhh_f, const float* bhh_b, const float* hprev, const float* dqp, float* carry, float* dgh, float* dgi_sum, int s, int B, long dir_stride_gh, long dir_stride_h, hipStream_t stream);
/* One launch per recurrence step of the same GRU (nn.GRU at model/transformer_v2.py:177 run over the batch axis, :215-221):
 * h_{s+1} = GRU(x, h_s) with gh = h_s W_hh^T computed in the kernel (fwd), and the BPTT step that first adds
 * dgh_{s+1} W_hh to the carried d*z and then differentiates the gates of step s (bwd; whhT from tatt_rpe_whh_transpose,
 * dgh_next = NULL on the first call, s = B-1). */
int tatt_rpe_step_fwd(const float* gi, const float* whh_f, const float* whh_b, const float* bhh_f, const float* bhh_b, const float* hprev, float* hout, float* gh_out, float* qp, int s, int B, long dir_stride_gh, long dir_stride_h, hipStream_t stream);
/* The B forward steps as ONE launch of persistent workgroups: the 8 independent chains (4 blocks of 16 rows x 2 directions) run one
 * per XCD (chain = XCC id + n_xcd * group; n_xcd from hipDeviceAttributeNumberOfXccs), 32 workgroups per chain keep their W_hh slice in
 * registers and hand h_s over through flag lines in that XCD's L2; every wait is bounded.  No counterpart call in the reference (it is
 * nn.GRU's time loop, model/transformer_v2.py:177,:215-221).  hstep [2][B+1][64][512] (h_0 set by the caller), gh_all [2][B][64][1536];
 * sync: device buffer of tatt_rpe_chain_sync_words(B) uint32 that the caller zeroes ONCE; every call clears its working part on `stream`
 * but not the first 8 words: sync[1] collects error flags across calls (sticky) -- != 0 = some call's wait gave up (its results are invalid).  Needs the GPU's CUs free enough to hold 32 workgroups of 512 threads per XCD at once; two calls that
 * share a sync buffer must not overlap.  Returns TATT_ERR_ARG when tatt_rpe_chain_supported() is 0. */
size_t tatt_rpe_chain_supported(void);   /* 1: the current device can hold the chains (8 XCDs x >= 32 CUs, or fewer XCDs with room for their share); 0: use the step kernels */
size_t tatt_rpe_chain_sync_words(int B);
int tatt_rpe_chain_fwd(const float* gi, const float* whh_f, const float* whh_b, const float* bhh_f, const float* bhh_b, float* hstep, float* gh_all, float* qp, int B, long dir_stride_gh, long dir_stride_h, uint32_t* sync, hipStream_t stream);
/* The B BPTT steps (s = B-1 .. 0) as ONE launch, same scheme; carry in/out and dgi_sum accumulated as by tatt_rpe_step_bwd; a sync buffer of its own. */
int tatt_rpe_chain_bwd(const float* gi, const float* gh_all, const float* bhh_f, const float* bhh_b, const float* hstep, const float* dqp, float* carry, float* dgh_all, float* dgi_sum, const float* whhT, int B, long dir_stride_gh, long dir_stride_h, uint32_t* sync, hipStream_t stream);
int tatt_rpe_whh_transpose(const float* whh_f, const float* whh_b, float* whhT, hipStream_t stream);
int tatt_rpe_step_bwd(const float* gi, const float* gh, const float* bhh_f, const float* bhh_b, const float* hprev, const float* dqp, float* carry, const float* dgh_next, float* dgh_out, float* dgi_sum, const float* whhT, int s, int B, long dir_stride_gh, long dir_stride_h, hipStream_t stream);
int tatt_rpe_gather_x(const float* init_factor, float* x, hipStream_t stream);
int tatt_rpe_scatter_dx(const float* dx, float* dinit, int accumulate, hipStream_t stream);

/* ---- attention core, 4 heads x 16, 26 keys (nn.MultiheadAttention at model/transformer_v2.py:476,:819) ---- */
int tatt_attn_fwd(const float* Q, const float* Kp, const float* Vp, float* ctx, float* pw, int B, int Lq, int nkeys, uint32_t seed_site, uint32_t thresh, float scale, const uint32_t* seed_dev, hipStream_t stream);
size_t tatt_attn_bwd_workspace(int B, int Lq);
int tatt_attn_bwd(const float* Q, const float* Kp, const float* Vp, const float* dctx, float* dQ, float* dK, float* dV, int B, int Lq, int nkeys, uint32_t seed_site, uint32_t thresh, float scale, const uint32_t* seed_dev, int accumulate_kv, float* workspace, hipStream_t stream);
int tatt_attn_avg(const float* pw, float* out, int B, int Lq, hipStream_t stream);

/* dK, dV (+)= sum over chunks of the partials [nchunk][2][B*26][64] (which 0 = dV, 1 = dK) that tatt_dec_bwd writes: the reduction
 * over the queries of torch's MultiheadAttention backward (model/transformer_v2.py:819). */
int tatt_attn_kv_final(const float* part, int nchunk, int B, float* dK, float* dV, int accumulate_kv, hipStream_t stream);

/* ---- the TP interpreter's decoder as fused launches (model/transformer_v2.py:805-832 TransformerDecoderLayer_TP.forward_post for
 *      both layers, :368-389 TransformerDecoder incl. `self.norm` on every intermediate, model/tsrn.py:220-224 mean over the layers) ----
 * Row-local chain per image token: q-projection of (tgt + query_pos), 26-key attention of 4 heads (softmax and dropout in
 * registers), out-projection, residual + dropout + LayerNorm2, Linear-ReLU-dropout-Linear, residual + dropout + LayerNorm3, the
 * decoder's final LayerNorm, mean over the two layers -> tp_map.  Keys / values are the projected text tokens Kp, Vp [B*26][64]
 * (tatt_gemm with multihead_attn.in_proj_weight[64:128] / [128:192]) of each layer.
 *   tatt_dec_pack_w: W = host array of 8 device pointers (layer 0 then layer 1: multihead_attn.in_proj_weight[0:64], out_proj.weight,
 *       linear1.weight, linear2.weight; row-major [64][64]) -> packed (tatt_dec_packed_w_floats() floats): each matrix and its
 *       transpose in MFMA-fragment order.
 *   tatt_dec_pack_kv: one layer's (Kp, Vp) -> table (tatt_dec_kv_table_floats(B) floats): the attention's A operands.
 *   tatt_dec_fwd: kv = host array [2] of tables; params = host array [2][8] (in_proj_bias[0:64], out_proj.bias, linear1.bias,
 *       linear2.bias, norm2.weight, norm2.bias, norm3.weight, norm3.bias); saved = host array [2][7] of [B*Lq][64] outputs the
 *       backward pass reads (qin, Q, context, LayerNorm2 input, LayerNorm2 output, dropped FFN hidden, LayerNorm3 input);
 *       seed_sites = host array [2][4] of tatt_seed_site words (attention, dropout2, FFN dropout, dropout3); thresh / scale /
 *       seed_dev as for tatt_ln_fwd.  Lq % 128 == 0.  npass = 2: the B batch elements are two passes of B/2 over the same weights (the
 *       TSC step's second SR forward, interfaces/super_resolution_eng.py:254-261, batched with the first): query_pos has B/2 * Lq rows
 *       and serves both, dropout indices count within a pass, seed_sites = [2 passes][2 layers][4] (tatt_dec_bwd: [2 passes][4]).
 *   tatt_dec_bwd: one layer (1 first, then 0).  dt3_in: gradient wrt the layer's output from the layer above (NULL for layer 1);
 *       dtp: gradient of tp_map; dtgt / dqp (+)= the gradients wrt the layer's input and query_pos; dY = host array [4] of [B*Lq][64]
 *       outputs (dQ, d attn-out, d hidden, d ffn-out), the left operands of the four weight gradients dY^T X with X = saved qin,
 *       context, LayerNorm2 output, dropped hidden (tatt_tn_small_group); kvpart (tatt_dec_bwd_kvpart_floats): per-128-token partials
 *       of dKp / dVp for tatt_attn_kv_final(nchunk = Lq/128); lnpart (tatt_dec_bwd_lnpart_floats): per-workgroup partials of the
 *       LayerNorm weight / bias gradients for tatt_dec_ln_final (grads = host array [6]: norm2.weight, norm2.bias, norm3.weight,
 *       norm3.bias, decoder.norm.weight, decoder.norm.bias; each +=). */
size_t tatt_dec_packed_w_floats(void);
size_t tatt_dec_kv_table_floats(int B);
size_t tatt_dec_bwd_kvpart_floats(int B, int Lq);
size_t tatt_dec_bwd_lnpart_floats(int B, int Lq);
int tatt_dec_pack_w(const float* const* W, float* packed, hipStream_t stream);
int tatt_dec_pack_kv(const float* Kp, const float* Vp, float* table, int B, hipStream_t stream);
int tatt_dec_fwd(const float* tgt, const float* qp, const float* packed_w, const float* const* kv, const float* const* params, const float* norm_w, const float* norm_b, float* const* saved, float* tp_map, int B, int Lq, int npass, const uint32_t* seed_sites, uint32_t thresh, float scale, const uint32_t* seed_dev, hipStream_t stream);
int tatt_dec_bwd(int layer, const float* dt3_in, const float* dtp, const float* packed_w, const float* kv, const float* const* params, const float* norm_w, float* const* saved, float* dtgt, int dtgt_accumulate, float* dqp, int dqp_accumulate, float* const* dY, float* kvpart, float* lnpart, int B, int Lq, int npass, const uint32_t* seed_sites, uint32_t thresh, float scale, const uint32_t* seed_dev, hipStream_t stream);
int tatt_dec_ln_final(const float* lnpart, int B, int Lq, float* const* grads, hipStream_t stream);

/* ---- element-wise / layout (model/tsrn.py:599 PReLU, :1042-1043 PixelShuffle+mish, :683 sigmoid, :208, :212;
 *      model/stn_head.py:36-44 MaxPool2d; model/transformer_v2.py:39-42 PE+dropout, :480 FFN dropout) -------- */
int tatt_nchw_to_nhwc(const float* x, float* y, int B, int C, int HW, hipStream_t stream);
int tatt_nhwc_to_nchw(const float* x, float* y, int B, int C, int HW, hipStream_t stream);
int tatt_axpby(const float* a, float alpha, const float* b, float beta, float* out, long n, int accumulate, hipStream_t stream);
/* n <= 64 contiguous float segments src[i][0..len[i]) copied to dst + dst_off[i] by one launch (host arrays).  Used to stack the
 * forward / reverse nn.GRU input weights of every BiGRU block ([96,64] + [96,64] -> [192,64], model/tsrn.py GruBlock) so that
 * both directions' input projections are one GEMM. */
int tatt_multi_copy(int n, const float* const* src, const long* dst_off, const long* len, float* dst, hipStream_t stream);
int tatt_prelu_fwd(const float* x, const float* slope, float* y, long n, hipStream_t stream);
int tatt_prelu_bwd(const float* x, const float* slope, const float* dy, float* dx, float* dslope, int accumulate, long n, float* workspace, hipStream_t stream);
size_t tatt_colsum_workspace(int C);
int tatt_colsum(const float* x, long ld, int R, int C, float* out, int accumulate, float* workspace, hipStream_t stream);
int tatt_pixshuf_mish_fwd(const float* x, float* y, int B, int H, int W, int C, hipStream_t stream);
int tatt_pixshuf_mish_bwd(const float* x, const float* dy, float* dx, int B, int H, int W, int C, hipStream_t stream);
/* dx stored as bf16: the gradient of the up-sampling convolution's output, read by that convolution's dgrad / wgrad launches only (C = 64).
 * colpart (may be NULL): [tatt_pixshuf_colpart_rows(B, H, W)][4C] fp32 column sums of the UNROUNDED values, one row per 64 pixels --
 * tatt_colsum over it is the convolution's bias gradient, as exact as from an fp32-stored dx. */
size_t tatt_pixshuf_colpart_rows(int B, int H, int W);
int tatt_pixshuf_mish_bwd_bf16(const float* x, const float* dy, uint16_t* dx, float* colpart, int B, int H, int W, int C, hipStream_t stream);
int tatt_sigmoid_to_nchw(const float* pre, float* sr, int B, int C, int HW, hipStream_t stream);
int tatt_sigmoid_bwd_to_nhwc(const float* sr, const float* dsr, float* dpre, int B, int C, int HW, hipStream_t stream);
int tatt_maxpool_fwd(const float* x, float* y, int B, int H, int W, int C, int kh, int kw, hipStream_t stream);
/* conv3x3_block's BatchNorm2d + ReLU and the MaxPool2d behind it (model/stn_head.py:19-20, :38-44), and stn_fc1's BatchNorm1d + ReLU
 * (:51-52), on the small maps of the STN head as ONE launch per direction (training mode: batch statistics, running statistics
 * updated): at most 1024 pooling windows (tatt_bn_pool_small_supported), window kh x kw in {1x1 (no pool), 2x2, 1x2}, C % 4 == 0.
 * act_out = relu(bn(x)) is kept for the backward call, which routes dpooled to the first maximum of each window
 * (tatt_maxpool_bwd's rule) and ACCUMULATES dgamma / dbeta. */
size_t tatt_bn_pool_small_supported(int B, int H, int W, int C, int kh, int kw);   /* 1: the pair below takes this shape */
int tatt_bn_pool_small_fwd(const float* x, float* stat, const float* gamma, const float* beta, float* run_mean, float* run_var, float* act_out, float* pooled, int B, int H, int W, int C, int kh, int kw, float eps, float momentum, hipStream_t stream);
int tatt_bn_pool_small_bwd(const float* x, const float* act, const float* dpooled, const float* stat, const float* gamma, float* dx, float* dgamma, float* dbeta, int B, int H, int W, int C, int kh, int kw, hipStream_t stream);
int tatt_maxpool_bwd(const float* x, const float* dy, float* dx, int B, int H, int W, int C, int kh, int kw, hipStream_t stream);
int tatt_tp_rows(const float* tp, long sB, long sK, long sL, float* out, int B, int K, int L, hipStream_t stream);
int tatt_pe_dropout(const float* pe, float* pos, int B, int L, uint32_t seed_site, uint32_t thresh, float scale, const uint32_t* seed_dev, hipStream_t stream);
int tatt_dropout_mul(const float* x, float* y, long n, uint32_t seed_site, uint32_t thresh, float scale, const uint32_t* seed_dev, hipStream_t stream);
int tatt_seed_bump(uint32_t* seed, hipStream_t stream);
int tatt_relu_drop_bwd(const float* h, const float* dh, float* out, long n, uint32_t seed_site, uint32_t thresh, float scale, const uint32_t* seed_dev, hipStream_t stream);

/* ---- TPS rectification = grid build + bilinear gather (model/tps_spatial_transformer.py:97-112, :10-11) and
 *      the TSC rotation (interfaces/super_resolution_eng.py:50-82) ----------------------------------------- */
int tatt_tps_sample_fwd(const float* x, const float* ctrl, const float* T, float* y, float* src_out, int B, int C, int H, int W, hipStream_t stream);
int tatt_tps_sample_bwd(const float* x, const float* ctrl, const float* T, const float* dy, float* dctrl, int B, int C, int H, int W, int accumulate, hipStream_t stream);
int tatt_rotate_fwd(const float* x, const float* arc, float* y, int B, int C, int H, int W, hipStream_t stream);
/* dx += rotate_fwd^T dy (the caller zeroes dx or holds a gradient to add to); a gather in a fixed summation order (no atomics), C <= 8 */
int tatt_rotate_bwd(const float* arc, const float* dy, float* dx, int B, int C, int H, int W, hipStream_t stream);

/* ---- TRI_SSIM, the TSC loss term (utils/ssim_psnr.py:99-128, :229-254; called at interfaces/super_resolution_eng.py:261),
 * and the evaluation metrics SSIM (utils/ssim_psnr.py:76-96, :200-225, channels 0..Cuse-1) and calculate_psnr (:9-15).
 * NCHW fp32, H*W <= 4096; out[0] is a device scalar; workspace: tatt_ssim_workspace(B, C) floats.  The backward adds
 * dmean * d(mean)/d(x1), d(x2) (x3 = the HR image takes no gradient). */
size_t tatt_ssim_workspace(int B, int C);
int tatt_tri_ssim_fwd(const float* x1, const float* x2, const float* x3, float* out, int B, int C, int H, int W, float* workspace, hipStream_t stream);
/* No counterpart in the reference: tatt_tri_ssim_fwd for callers that cannot allocate or pass a workspace -- code under hipGraph
 * capture, tight evaluation / serving loops, harnesses that forbid hipMalloc.  Same result, bit for bit; the per-plane partial
 * sums live in a 64 KB static device array of the library (one copy per GPU, held until the process ends), so B*C <= 16384 planes
 * or the call is refused with an error.  It calls no hipMalloc / hipFree.  Calls that overlap in time on one GPU race on that
 * array; the first call on each device resolves the array's address and must be made outside stream capture. */
int tatt_tri_ssim_fwd_static(const float* x1, const float* x2, const float* x3, float* out, int B, int C, int H, int W, hipStream_t stream);
int tatt_tri_ssim_bwd(const float* x1, const float* x2, const float* x3, float dmean, float* dx1, float* dx2, int accumulate, int B, int C, int H, int W, hipStream_t stream);
int tatt_ssim_fwd(const float* x1, const float* x2, float* out, int B, int Ctot, int Cuse, int H, int W, float* workspace, hipStream_t stream);
int tatt_psnr(const float* x1, const float* x2, float* out, int B, int Ctot, int Cuse, int H, int W, float* workspace, hipStream_t stream);

/* ---- CRNN text-prior generator, forward (model/crnn/crnn.py:45-119) -----------------------------
 * tatt_crnn_prep = parse_crnn_data (interfaces/base.py:441-459): bicubic resize (torch defaults) of channels 0..2 + gray,
 * NCHW in -> NHWC out with C padded to 4 (channel 0 = gray).  tatt_maxpool_gen_fwd = nn.MaxPool2d(kernel, stride, padding)
 * (crnn.py:77-85), NHWC.  tatt_linear_f64 = the `embedding` Linear evaluated in float64 (crnn.py:31-33).
 * tatt_lstm256_step_fwd = one time step of nn.LSTM(nIn, 256, bidirectional=True) (crnn.py:10,:19): gi [T*B][2*1024] is
 * W_ih x + b_ih of both directions (time-major rows), h/c state [2][B][256] (hprev and hout must not alias),
 * out [T*B][512]; step s handles time s for direction 0 and T-1-s for direction 1. */
int tatt_crnn_prep(const float* x, float* y, int B, int Ctot, int H, int W, int OH, int OW, hipStream_t stream);
int tatt_maxpool_gen_fwd(const float* x, float* y, int B, int H, int W, int C, int KH, int KW, int SH, int SW, int PH, int PW, hipStream_t stream);
int tatt_linear_f64(const float* x, long ldx, const float* Wt, const float* bias, float* out, long ldo, int R, int N, int K, hipStream_t stream);
int tatt_lstm256_step_fwd(const float* gi, const float* whh_f, const float* whh_b, const float* bhh_f, const float* bhh_b, const float* hprev, const float* cprev, float* hout, float* cout, float* out, float* save, int s, int T, int B, hipStream_t stream);
/* The T steps of one nn.LSTM(., 256, bidirectional) (model/crnn/crnn.py:10,:19-21) as ONE launch of persistent workgroups, forward
 * and BPTT: a chain = (16 batch rows, direction) runs inside one XCD's L2 (32 workgroups x 8 hidden units, W_hh slice in registers,
 * flag hand-over as tatt_rpe_chain_fwd); B <= 128, device as tatt_rpe_chain_supported().  gi / out / save / dgates / whhT exactly as the
 * step entry points below (h_0 = c_0 = 0; the carried dc is internal).  sync: tatt_lstm256_chain_sync_words() uint32, zeroed once by
 * the caller, one buffer per concurrently running call; sync[1] != 0 = a bounded wait gave up (sticky; outputs NaN from there on). */
size_t tatt_lstm256_chain_sync_words(void);
int tatt_lstm256_chain_fwd(const float* gi, const float* whh_f, const float* whh_b, const float* bhh_f, const float* bhh_b, float* out, float* save, int T, int B, uint32_t* sync, hipStream_t stream);
int tatt_lstm256_chain_bwd(const float* save, const float* whhT, const float* dout, float* dgates, int T, int B, uint32_t* sync, hipStream_t stream);
/* Backward pieces of the CRNN student (autograd of the modules above).  tatt_lstm256_step_bwd: BPTT step s (call with
 * s = T-1 .. 0); save [T*B][2][5][256] is what tatt_lstm256_step_fwd stored (i, f, g, o, c), whhT [2][256][1024] = W_hh^T
 * per direction (tatt_transpose), dout = d(out), dgates [T*B][2048] receives the pre-activation gate gradients (read back by
 * the next call and by the caller's weight-gradient GEMMs), dc_carry [2][B][256] is scratch carried between calls.
 * tatt_maxpool_gen_bwd: gather form of MaxPool2d's backward (first maximum wins, deterministic). */
int tatt_lstm256_step_bwd(const float* save, const float* whhT, const float* dout, float* dgates, float* dc_carry, int s, int T, int B, hipStream_t stream);
int tatt_transpose(const float* in, float* out, int rows, int cols, hipStream_t stream);
int tatt_maxpool_gen_bwd(const float* x, const float* dy, float* dx, int B, int H, int W, int C, int KH, int KW, int SH, int SW, int PH, int PW, hipStream_t stream);
/* SemanticLoss (loss/semantic_loss.py:21-38: mean |gt - pred| + KLDivLoss()(log(pred + 1e-20), gt + 1e-20)) fused with the two
 * softmaxes of the trainer (interfaces/super_resolution_eng.py:222-228,:240): zs / zt = student / teacher logits [R = T*B, K];
 * out[0] = loss; if dzs != NULL, dzs = dloss * d loss / d zs.  workspace: tatt_semantic_loss_workspace(R) floats.
 * tatt_semantic_loss_probs is the module form on probabilities (pred, gt of n elements; dpred optional), and
 * tatt_softmax_rows the prior itself, label_vecs = softmax(logits, -1) (super_resolution_eng.py:227). */
size_t tatt_semantic_loss_workspace(int R);
int tatt_semantic_loss(const float* zs, const float* zt, float* out, float* dzs, float dloss, int R, int K, float* workspace, hipStream_t stream);
int tatt_semantic_loss_probs(const float* pred, const float* gt, float* out, float* dpred, float dloss, long n, float* workspace, hipStream_t stream);
int tatt_softmax_rows(const float* z, float* out, int R, int K, hipStream_t stream);
/* Greedy CTC decode of get_string_crnn (utils/metrics.py:128-146) on logits [T][B][K]: arg [B][T] = per-step argmax (first
 * index on ties, like torch.max), codes [B][T] / lens [B] = the class indices left after merging runs and dropping blank 0. */
int tatt_ctc_greedy(const float* logits, int* arg, int* codes, int* lens, int T, int B, int K, hipStream_t stream);
/* F.interpolate(x, (OH, OW), mode="bicubic") on `planes` = B*C contiguous [H][W] planes: the LR image at SR size for the
 * psnr_lr / ssim_lr metrics of the eval loop (interfaces/super_resolution_eng.py:486). */
int tatt_bicubic_resize(const float* x, float* y, int planes, int H, int W, int OH, int OW, hipStream_t stream);

/* ---- input transform: resizeNormalize.__call__ (dataset/dataset.py:282-325) per sample of
 * alignCollate.__call__ (:677-693) = PIL resize(BICUBIC) -> ToTensor -> mask channel (1 where L <= mean L), bit-exact with
 * Pillow's integer resampler.  data = the batch's uint8 RGB images (HWC) packed in one device buffer, image b at byte
 * offsets[b] with heights[b] x widths[b] pixels (device arrays); out [B][mask ? 4 : 3][OH][OW] float.  max_h / max_w bound the
 * sizes in the batch (they size the LDS; a larger image yields NaNs in its slot).  tatt_resize_normalize_lds = the bytes of
 * LDS one image then needs (<= 160 KB - 64 or the call is refused). */
size_t tatt_resize_normalize_lds(int OH, int OW, int max_h, int max_w);
int tatt_resize_normalize(const unsigned char* data, const long* offsets, const int* heights, const int* widths, float* out, int B, int OH, int OW, int mask, int max_h, int max_w, hipStream_t stream);

/* ---- ImageLoss = MSE + gradient-prior L1 (loss/image_loss.py:19-32, :40-56), NCHW ------------------------- */
int tatt_image_loss_fwd(const float* sr, const float* hr, float* per, int B, int C, int H, int W, float w0, float w1, hipStream_t stream);
int tatt_image_loss_bwd(const float* sr, const float* hr, const float* dper, float uniform, float* dsr, int B, int C, int H, int W, float w0, float w1, hipStream_t stream);
/* both in one launch (per [B] and dsr = uniform * d per / d sr): the step's form when no per-sample mask weights the gradient */
int tatt_image_loss_fwd_bwd(const float* sr, const float* hr, float* per, float uniform, float* dsr, int B, int C, int H, int W, float w0, float w1, hipStream_t stream);

/* ---- clip_grad_norm_ + Adam over a flat parameter buffer (interfaces/super_resolution_eng.py:277-279) ------ */
int tatt_clip_adam(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, float max_norm, float* state, float* workspace, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif
