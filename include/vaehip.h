/* libvaehip.so -- C ABI of the MI355X (gfx950) SDXL-VAE train-step hot path.
 *
 * The reference (olegroshka/vae-channel-dynamics) has no native code; the
 * boundary it sits behind is the Python object protocol of
 *   src/models/sdxl_vae_wrapper.py:42-77   (SDXLVAEWrapper.forward)
 *   src/train.py:283-306                   (step body: loss, backward, clip, AdamW)
 *   src/tracking/monitor.py:56-80          (per-channel mean|A| tracker)
 * and every numerically heavy call goes into torch/diffusers library kernels.
 * Each entry point below names the library op(s) at the reference call site it
 * replaces.
 *
 * Conventions
 *   - every function returns 0 on success, a negative VAE_E* code otherwise;
 *     vae_last_error() returns a thread-local message for the last failure.
 *   - all pointers are DEVICE pointers owned by the caller (PyTorch allocator);
 *     nothing is allocated, freed or synchronised inside; every call only
 *     enqueues work on `stream` (a hipStream_t passed as void*).
 *   - activations are NHWC fp32: [B][H][W][C], C contiguous.
 *   - conv / linear weights are OHWI fp32: [Cout][KH][KW][Cin] (this is the
 *     memory of an nn.Conv2d weight in torch.channels_last layout, and of an
 *     nn.Linear weight as is), read live every call -- no packed copies.
 */
#ifndef VAEHIP_H
#define VAEHIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define VAE_OK 0
#define VAE_EINVAL (-1)   /* shape / pointer / alignment check failed */
#define VAE_ELAUNCH (-2)  /* hip launch error */

const char* vae_last_error(void);
int vae_abi_version(void);
/* sizeof(vae_conv_geom / vae_igemm_args / vae_wgrad_args) for which = 0 / 1 / 2: lets a binding verify its struct mirrors */
int vae_sizeof_args(int32_t which);
/* Process-wide kernel-selection switches (no launch path reads the environment): "flat_conv" (flat implicit-GEMM kernels
 * everywhere: the second algorithm of the two-algorithm tests), "no_wino" (fp32: the direct halo-tile kernels instead of the
 * Winograd ones: the parity reference), "no_wino4" (fp32: F(2x2,3x3) also on the layers F(4x4,3x3) would serve; a Wu image
 * must be built and consumed under the same value: vae_wino_weight_floats / vae_wino_weights / vae_igemm_rows all follow it),
 * "no_thin_mfma" (bf16: the <= 4-channel-side layers stay on the VALU kernels), "no_wide" (bf16: the 128-pixel halo-tile kernel instead of the wide-tile one),
 * "no_wgrad_dma" (bf16: the 3x3 weight gradient stages its bf16 images through registers instead of by LDS-DMA; same result bit for bit).
 * One option is a count, not a switch: "wide_reserved_cus" (0..128, default 0): the persistent bf16 wide-tile kernel launches
 * 256 - n workgroups instead of one per CU, leaving n CUs to RCCL's workgroups while gradient buckets are in flight
 * (data-parallel runs: reference src/train.py:204-211); any grid covers all tiles, so results do not depend on it.
 * Initial values come from VAEHIP_FLAT_CONV / VAEHIP_NO_WINO / VAEHIP_NO_WINO4 / VAEHIP_NO_WIDE / VAEHIP_NO_THIN_MFMA / VAEHIP_NO_WGRAD_DMA /
 * VAEHIP_WIDE_RESERVED_CUS, read ONCE when the
 * library is loaded.  vae_get_option returns the value, or -1 for an unknown name. */
int vae_set_option(const char* name, int32_t value);
int vae_get_option(const char* name);

/* input transform applied to the A operand while it is staged into LDS */
#define VAE_XF_NONE 0
#define VAE_XF_AFFINE 1      /* x*scale[b][c]+shift[b][c]            (GroupNorm, no activation) */
#define VAE_XF_AFFINE_SILU 2 /* silu(x*scale[b][c]+shift[b][c])      (GroupNorm + SiLU)         */

/* arithmetic of the contraction.  BF16 multiplies bf16 operands on v_mfma_f32_32x32x16_bf16 with fp32 accumulation
 * (`training.mixed_precision: bf16`); parameters, GroupNorm statistics, loss and optimizer stay fp32 in both modes.
 * STORAGE of the activation / gradient tensors is a separate, per-tensor property in bf16 mode: every operand is fp32
 * unless the argument block says otherwise (A16 / a_bf16 / out_bf16 / res_bf16, X16 / dY16 / x_bf16 / y_bf16, x_bf16 of the
 * GroupNorm entry points) -- the host keeps conv outputs, the residual stream and their gradients as bf16, which is what
 * autocast does for the reference (src/train.py:147-154).  Shapes without a bf16 kernel run the fp32 one (fp32 storage). */
#define VAE_PREC_F32 0
#define VAE_PREC_BF16 1

/* how a row (b,y,x) of the GEMM maps onto the source tensor */
#define VAE_MODE_FWD 0   /* sy = y*stride + kh - pad_t                          */
#define VAE_MODE_UP2X 1  /* source is virtually nearest-upsampled 2x, 3x3 pad 1 */
#define VAE_MODE_DGRAD 2 /* sy = (y + pad_t - kh)/stride when divisible         */
#define VAE_MODE_DGRAD_S2 3 /* dgrad of a 3x3 stride-2 pad-0 conv with the rows (= conv input pixels, Ho x Wo even)
                             * enumerated PARITY-CLASS-major: m = ((cls*B + b)*Ho/2 + i)*Wo/2 + j, pixel (2i+cls/2, 2j+cls%2).
                             * A class only meets the taps of its parity (4, 2, 2 or 1 of the 9), so no masked work;
                             * needs B*Ho*Wo/4 % 128 == 0 (class-uniform tiles).  Output rows are written at the pixel. */
#define VAE_MODE_UP2X_DGRAD 4 /* gradient wrt the LOW-resolution input of conv3x3(nearest_upsample_2x(x)): source = the
                             * HIGH-resolution dY (Hs = 2 Ho, Ws = 2 Wo), row grid = the low-resolution pixels; the 3x3 dgrad and the
                             * 2x2 sum-pool in one pass.  Only the fp32 Winograd-type kernel implements it (vae_wino_ok, Wu). */

typedef struct vae_conv_geom {
  int32_t B, Hs, Ws, Cs; /* source tensor [B][Hs][Ws][Cs]                         */
  int32_t Ho, Wo;        /* row grid: GEMM row m = (b, y<Ho, x<Wo)                */
  int32_t taps;          /* 1 (1x1 / linear / plain GEMM) or 9 (3x3)              */
  int32_t stride;        /* 1 or 2                                                */
  int32_t pad_t, pad_l;  /* top/left padding (bottom/right is implied by bounds)  */
  int32_t mode;          /* VAE_MODE_*                                            */
} vae_conv_geom;

/* C[z][m][n] = sum_{tap,k} XF(A[z][row(m,tap)][k]) * W[z][n*sn + k*sk + tap*st]
 *              (+ bias[n]) (+ res[z][m][n])
 * replaces: conv2d fwd (K1,K5), conv2d dgrad (K7), linear fwd/dgrad and the
 * attention bmm's Q.K^T, P.V, dO.V^T, dS.K (K3,K7).
 * sk==1 selects the "k-contiguous" weight tile (forward); sn==1 the
 * "n-contiguous" one (dgrad / P.V).  Exactly one of them must be 1.
 * track (optional): partial sums of |C| per column, [ceil(M/128)][N] per z==0
 * (conv only; used for the fused ActivityMonitor metric, monitor.py:66).       */
typedef struct vae_igemm_args {
  const float* A; const float* W; float* C;
  const float* bias; const float* res;
  const float* scale; const float* shift; /* [B][Cs] when xf != NONE */
  float* track;
  vae_conv_geom g;
  int32_t M, N, K;       /* K = channels per tap actually contracted (<= Cs)      */
  int32_t ldc;           /* row stride of C and res                               */
  int64_t sn, sk, st;    /* weight element strides                                */
  int32_t batch;         /* grid.z; 1 for conv                                    */
  int64_t sAb, sWb, sCb; /* per-batch element strides (attention)                 */
  int32_t xf;            /* VAE_XF_*                                              */
  float alpha;           /* C = alpha * acc (+bias+res); 1.0 for conv             */
  int32_t prec;          /* VAE_PREC_*: arithmetic of the products                */
  const void* Wh;        /* optional (prec == BF16): bf16 image of W, same element layout (vae_pack_bf16); NULL = round W on the fly */
  const void* A16;       /* optional (prec == BF16, xf == NONE, vae_bf16_act_image_ok): bf16 image of the ALREADY TRANSFORMED
                          * operand, same NHWC layout (vae_gn_apply_bf16); the kernel then reads it instead of A */
  /* Sub-sampled views and tap subsets (vae_conv_phase_ok(a) != 0; zero-initialised = plain): the conv runs on the row grid
   * of `g`, but pixel (y,x) of A lives at (y*a_step + a_oy, x*a_step + a_ox) of a tensor a_step times larger in H and W,
   * likewise for C / res with c_step, c_oy, c_ox; only the taps set in tapmask (bit kh*3+kw; 0 = all 9) are computed.
   * Used for conv-over-nearest-2x-upsample as 4 phase convolutions with 2x2 effective kernels (vae_upconv_phase_weights):
   * 16 instead of 36 tap-MACs per low-resolution pixel, forward and dgrad.                                             */
  int32_t tapmask;
  int32_t a_step, a_oy, a_ox;
  int32_t c_step, c_oy, c_ox;
  float* gstat;          /* optional (vae_conv_gstat_chunks(a) > 0): GroupNorm statistics of the OUTPUT from the epilogue:  */
  int32_t gstat_groups;  /* ws[b][chunk][gstat_groups][2] = (mean, M2) per output tile -- the layout
                          * vae_gn_stats_partial writes, so vae_gn_stats_final finishes it; saves re-reading the output */
  const float* Wu;       /* optional (prec == F32, vae_wino_ok(a)): the Winograd-transformed weights vae_wino_weights(a, Wu) built from
                          * W; the 3x3 stride-1 layer then runs as F(2x2,3x3): 16 instead of 36 multiplications per 2x2 outputs */
  /* storage of the operands in bf16 mode (all zero = fp32 everywhere); vae_conv_io16_ok(a) tells whether the kernel that
   * serves `a` honours the combination that is set:                                                                      */
  int32_t out_bf16;      /* C is a bf16 tensor (2 B per element, same [m][ldc] layout); bias / res / gstat / c_step apply as ever
                          * (the statistics epilogue then describes the ROUNDED values, i.e. the tensor as stored); no track   */
  int32_t a_bf16;        /* A itself is a bf16 tensor (same layout) and xf still applies: the flat and the <= 4-channel kernels.
                          * (A16 is the other case: an already transformed image next to / instead of A, xf == NONE.)          */
  int32_t res_bf16;      /* res is a bf16 tensor                                                                               */
  /* GroupNorm-backward partial sums from a DGRAD epilogue (all zero = off; vae_conv_gnb_chunks(a) > 0): the launch computes
   * dA = dL/d silu(gn(x)) (gnb_silu) or dL/d gn(x); with the GroupNorm input x (same [B,Ho,Wo,N] shape and ldc as C, stored
   * fp32 or bf16), its statistics and affine parameters the epilogue also leaves what vae_gn_bwd_partial would compute from
   * (x, dA) in a separate pass over both tensors: gnb_ws[b][chunk][N][2] = (sum dz, sum dz * xhat) per output tile, dz = dA *
   * silu'(gamma * xhat + beta) -- vae_gn_bwd_final(nchunk = chunks) finishes it.  (reference: the autograd backward of
   * diffusers' ResnetBlock2D norm -> nonlinearity -> conv, call sites src/models/sdxl_vae_wrapper.py:60,71)               */
  const void* gnb_x;
  const float* gnb_mean; const float* gnb_rstd;   /* [B][gnb_groups] */
  const float* gnb_gamma; const float* gnb_beta;  /* [N]             */
  float* gnb_ws;
  int32_t gnb_groups, gnb_silu, gnb_x_bf16;
} vae_igemm_args;
int vae_igemm_rows(const vae_igemm_args* a, void* stream);
/* chunks per image the launch for `a` (gnb_x .. gnb_groups set) would write into a->gnb_ws, or 0 when the kernel serving it
 * has no GroupNorm-backward epilogue (the caller then runs vae_gn_bwd_partial)                                       */
int vae_conv_gnb_chunks(const vae_igemm_args* a);
/* number of chunks per image the launch for `a` (with a->gstat_groups set) would write into a->gstat, or 0 when the
 * kernel serving it has no statistics epilogue (the caller then runs vae_gn_stats_partial on the output)           */
int vae_conv_gstat_chunks(const vae_igemm_args* a);
/* Winograd minimal filtering, fp32 forward / dgrad of plain 3x3 stride-1 layers: F(4x4,3x3) (csrc/conv3_wino4.hip: 36 positions)
 * where the maps are whole 16 x 32 tiles and N whole 64-channel blocks (and option "no_wino4" is off), else F(2x2,3x3)
 * (csrc/conv3_wino.hip: 16 positions); 9 positions for the upsampler geometries (csrc/conv3_upwino.hip).  vae_wino_ok: 1 when the
 * layer `a` describes (a->Wu ignored) is served; vae_wino_weight_floats: size of the transformed weights for THAT kernel
 * (36, 16 or 9 x N * K floats: always ask, never assume); vae_wino_weights: U = G g G^T of a->W (the dgrad geometry rotates and
 * transposes), layout [K/8][positions][N][8], into Wu.  The weights are transformed per launch because they are live
 * (optimizer step, in-place nudges).  Accuracy against a float64 convolution, worst element relative to the tensor's max:
 * F(2x2) 1e-6, F(4x4) 1-2e-5 (tools/wino4_accuracy.py). */
int vae_wino_ok(const vae_igemm_args* a);
int64_t vae_wino_weight_floats(const vae_igemm_args* a);
int vae_wino_weights(const vae_igemm_args* a, float* Wu, void* stream);
/* 1 when the kernel that would serve `a` honours a->out_bf16 / a->a_bf16 / a->res_bf16 exactly as they are set (with all
 * three zero: always 1).  The bf16 flat kernels take any combination; the halo-tile kernels bf16 outputs with a residual of
 * the same storage; the <= 4-channel kernels the flag of their wide side; fp32-arithmetic kernels none.                  */
int vae_conv_io16_ok(const vae_igemm_args* a);
/* 1 when the kernel that would serve `a` honours tapmask / a_step.. / c_step.. (the halo-tile kernels), else 0     */
int vae_conv_phase_ok(const vae_igemm_args* a);
/* W [Co][3][3][Ci] (OHWI) -> Weff [4 phases (a*2+b)][Co][3][3][Ci]: the 3x3 kernel each output parity (a,b) of
 * conv3x3(nearest_upsample_2x(x)) applies to the LOW-resolution x (zeros outside its 2x2 support; sums of 1, 2 or 4
 * original taps inside).  tapmask of phase (a,b): rows {0,1} (a=0) or {1,2} (a=1), columns likewise.               */
int vae_upconv_phase_weights(const float* W, int32_t Co, int32_t Ci, float* Weff, void* stream);
/* 1 when xf != NONE can be fused for this geometry (the GroupNorm scale/shift rows a tile needs are staged in
 * LDS once per workgroup); 0 => the caller materialises XF(x) with vae_gn_apply and passes xf = NONE.
 * Only tiny spatial sizes (H*W < 128 with several batch items per tile) are not fusable.                  */
int vae_xf_fusable_rows(const vae_conv_geom* g, int32_t M, int32_t K);

/* dW[z][split][m][tap][n] = sum_{pix in split} dY[z][pix][m] * XF(X[z][row(pix,tap)][n])
 * replaces: conv2d wgrad, linear wgrad, attention P^T.dO and dS^T.Q (K7).
 * When nsplit==1 the result goes straight to `out` (ld = taps*N); otherwise to
 * `partial` ([nsplit][M][taps][N]) and the caller runs vae_reduce_splits.      */
typedef struct vae_wgrad_args {
  const float* dY; const float* X; float* out; float* partial;
  float* bias_partial;   /* optional [nsplit][M]: per-split column sums of dY (bias gradient); reduce like `partial` */
  const float* scale; const float* shift;
  vae_conv_geom g;       /* row grid = output pixels of the conv; source = X      */
  int32_t M, N;          /* M = Cout (cols of dY used), N = Cin (<= g.Cs)         */
  int32_t ldy;           /* row stride of dY                                      */
  int32_t npix;          /* contraction length = B*Ho*Wo                          */
  int32_t nsplit;
  int32_t batch; int64_t sYb, sXb, sOb;
  int32_t xf;
  float alpha;
  int32_t prec;          /* VAE_PREC_* */
  const void* X16;       /* optional: as vae_igemm_args.A16, for X (xf must be NONE) */
  const void* dY16;      /* optional (vae_bf16_grad_image_ok): bf16 image of dY, same [pix][ldy] layout; the kernel then reads it
                          * instead of dY (which may be NULL): half the bytes, and no rounding pass in the kernel            */
  /* phase convolutions of an upsampler (vae_wgrad_phase_ok(a) != 0; zero-initialised = plain): pixel (y,x) of dY lives at
   * (y*y_step + y_oy, x*y_step + y_ox) of a tensor y_step times larger in H and W; only the taps in tapmask are computed
   * (the others are written as zeros).  vae_upconv_fold_wgrad turns the four phase results into the 3x3 gradient.     */
  int32_t tapmask;
  int32_t y_step, y_oy, y_ox;
  int32_t x_bf16, y_bf16; /* X / dY themselves are bf16 tensors (xf still applies to X): the bf16 flat kernel and the wide side of
                           * the <= 4-channel kernels (vae_wgrad_io16_ok); the halo-tile kernel takes images through X16 / dY16 */
} vae_wgrad_args;
int vae_wgrad(const vae_wgrad_args* a, void* stream);
/* Winograd F(3x3,2x2) weight gradient of plain 3x3 stride-1 layers (and, with 9 positions, of upsampler convolutions) in fp32 (csrc/wgrad3_wino.hip; 2.25x fewer multiplications
 * than vae_wgrad).  vae_wgrad_wino_plan: *nsplit = 0 when the layer `a` describes is not served, else the split count to use;
 * vae_wgrad_wino: a->partial receives the transform-domain slab [nsplit][16][Cin][Cout] (a->out unused), a->bias_partial
 * (optional) [nsplit][Cout] as in vae_wgrad; vae_wgrad_wino_reduce: fixed-order sum over the splits (into scratch
 * [16][Cin][Cout], needed when nsplit > 1) + output transform into dW [Cout][3][3][Cin] (OHWI) and, when bias_partial != NULL,
 * db [Cout].  Replaces the same reference call as vae_wgrad.                                                              */
int vae_wgrad_wino_plan(const vae_wgrad_args* a, int32_t* nsplit);
int vae_wgrad_wino(const vae_wgrad_args* a, void* stream);
int vae_wgrad_wino_reduce(const float* slab, int32_t nsplit, int32_t npos, int32_t Cin, int32_t Cout, float* scratch, float* dW,
                          const float* bias_partial, float* db, void* stream);
/* positions of the transform domain the kernel serving `a` accumulates: 16 (plain 3x3 stride-1 layer), or 9 for the convolution
 * of an Upsample2D block (a->g.mode == VAE_MODE_UP2X: rows and columns of the upsampled patch repeat in pairs, csrc/wgrad3_upwino.hip);
 * the slab is [nsplit][npos][Cin][Cout] and npos goes to vae_wgrad_wino_reduce                                                   */
int vae_wgrad_wino_positions(const vae_wgrad_args* a);
/* 1 when the kernel serving `a` honours a->x_bf16 / a->y_bf16 as they are set */
int vae_wgrad_io16_ok(const vae_wgrad_args* a);
/* 1 when the kernel serving `a` honours tapmask / y_step.. (the fp32 halo-tile wgrad kernel)                       */
int vae_wgrad_phase_ok(const vae_wgrad_args* a);
/* transpose of vae_upconv_phase_weights: dWeff [4][Co][3][3][Ci] (+ dbeff [4][Co] or NULL) -> dW [Co][3][3][Ci] (+ db) */
int vae_upconv_fold_wgrad(const float* dWeff, const float* dbeff, int32_t Co, int32_t Ci, float* dW, float* db, void* stream);
/* split-K plan for `a` (a->nsplit ignored): the nsplit to launch with, and whether a->xf can be fused
 * (0 => materialise XF(X) with vae_gn_apply and pass xf = NONE; only tiny spatial sizes).                */
int vae_wgrad_plan(const vae_wgrad_args* a, int32_t* nsplit, int32_t* xf_fusable);
/* name of the kernel instantiation the two entry points dispatch to for these arguments (profiling labels
 * that match the rocprofv3 kernel names); nothing is launched                                             */
int vae_igemm_kernel_name(const vae_igemm_args* a, char* buf, int32_t n);
int vae_wgrad_kernel_name(const vae_wgrad_args* a, char* buf, int32_t n);
/* out[i] = sum_s partial[s][i], fixed order (deterministic)                     */
int vae_reduce_splits(const float* partial, int32_t nsplit, int64_t n, float* out, void* stream);
/* the same for two slabs with one nsplit in ONE launch: the weight-gradient partials and the (small) bias-gradient partials */
int vae_reduce_splits2(const float* partial, int32_t nsplit, int64_t n, float* out, const float* partial2, int32_t n2, float* out2,
                       void* stream);
/* ---- GroupNorm (32 groups, eps 1e-6) replaces group_norm fwd/bwd (K2,K7) ----
 * x_bf16 != 0: the activation tensor x (and, in the backward, the residual-path gradient `add`) is stored as bf16 -- bf16 mode
 * keeps conv outputs and the residual stream as bf16, as autocast does for the reference (src/train.py:147-154); statistics,
 * partial sums and all arithmetic are fp32 either way. */
/* stage 1: per (b, chunk, group) CENTRED moments ws [B][nchunk][G][2] = (mean, M2 = sum of squared deviations) of chunk
 * `chunk` = pixels [chunk*per, min(HW, (chunk+1)*per)), per = ceil(HW/nchunk): shifted sums, no E[x^2]-mean^2 cancellation
 * (torch's CPU group_norm, which the reference runs on, is a Welford pass; activations with |mean| >> std need it)     */
int vae_gn_stats_partial(const void* x, int32_t x_bf16, int32_t B, int32_t HW, int32_t C, int32_t G,
                         int32_t nchunk, float* ws, void* stream);
/* stage 2: Chan's merge of the chunk moments in fp64 -> mean/rstd [B][G] and the fused affine scale/shift [B][C]:
 *   scale = rstd*gamma, shift = beta - mean*rstd*gamma                            */
int vae_gn_stats_final(const float* ws, int32_t B, int32_t HW, int32_t C, int32_t G, int32_t nchunk,
                       const float* gamma, const float* beta, float eps,
                       float* mean, float* rstd, float* scale, float* shift, void* stream);
/* y = XF(x) materialised (only for layers with foreign hooks / full maps)        */
int vae_gn_apply(const void* x, int32_t x_bf16, const float* scale, const float* shift, int32_t B, int32_t HW,
                 int32_t C, int32_t xf, float* y, void* stream);
/* y16 = bf16(XF(x)): the activation image of a GroupNorm(+SiLU)'d conv input for bf16 mode (C % 8 == 0); the layer's
 * forward and wgrad then read 2 B per element and transform nothing (vae_igemm_args.A16 / vae_wgrad_args.X16)      */
int vae_gn_apply_bf16(const void* x, int32_t x_bf16, const float* scale, const float* shift, int32_t B, int32_t HW,
                      int32_t C, int32_t xf, void* y16, void* stream);
/* 1 when BOTH the forward (vae_igemm_rows) and the weight gradient (vae_wgrad) of the 3x3 stride-1 layer with this
 * forward geometry, Cout and Cin accept a bf16 activation image in bf16 mode (the halo-tile kernels serve them)   */
int vae_bf16_act_image_ok(const vae_conv_geom* fwd_geom, int32_t Cout, int32_t Cin);
/* 1 when BOTH the dgrad (vae_igemm_rows, mode DGRAD, A16 = the bf16 gradient) and the weight gradient (vae_wgrad, dY16) of
 * the 3x3 stride-1 layer with this forward geometry accept the output gradient as a bf16 image in bf16 mode            */
int vae_bf16_grad_image_ok(const vae_conv_geom* fwd_geom, int32_t Cout, int32_t Cin);
/* tracker (monitor.py:66): partial sums of |x*scale+shift| per (b,chunk,c);
 * ws [B][nchunk][C]; then vae_track_final                                         */
int vae_gn_track_partial(const void* x, int32_t x_bf16, const float* scale, const float* shift, int32_t B, int32_t HW,
                         int32_t C, int32_t nchunk, float* ws, void* stream);
/* out[c] = (sum over rows of ws[r][c]) * inv_count ; fixed order                   */
int vae_track_final(const float* ws, int32_t rows, int32_t C, float inv_count, float* out, void* stream);
/* GroupNorm(+SiLU) backward.  g = dL/d(XF(gn(x))): fp32, or bf16 when g_bf16 != 0 (bf16 mode stores the dgrad outputs of
 * the halo-tile kernels as bf16, vae_igemm_args.out_bf16).
 * stage 1: ws [B][nchunk][C][2] partial sums of du and du*xhat                     */
int vae_gn_bwd_partial(const void* x, int32_t x_bf16, const void* g, const float* mean, const float* rstd,
                       const float* gamma, const float* beta, int32_t B, int32_t HW, int32_t C, int32_t G,
                       int32_t nchunk, int32_t silu, int32_t g_bf16, float* ws, void* stream);
/* stage 2: dgamma/dbeta [C] (written, not accumulated) and coefficients
 * coef [B][G][2] = {rstd*s2/N, rstd*s1/N}; one launch (ws is only read).
 * Shapes: C <= 1024, channels per group C / G in {1, 2, 4, 8, 16} and G % 4 == 0 (one workgroup finishes 4 groups; SDXL-VAE:
 * G = 32, C / G in {4, 8, 16}); anything else returns VAE_EINVAL with the reason in vae_last_error()                        */
int vae_gn_bwd_final(const float* ws, const float* rstd, const float* gamma, int32_t B, int32_t HW,
                     int32_t C, int32_t G, int32_t nchunk, float* dgamma, float* dbeta, float* coef,
                     void* stream);
/* stage 3: dx = du*rstd*gamma - xhat*coef0 - coef1 (+ add), written as fp32 (dx) and / or as bf16 (dx16); at least one.
 * A gradient that only feeds convolutions (dL/dh of a resnet) needs the bf16 copy alone -- the bf16 kernels would round it
 * at the same point; one that also carries the residual stream gets both.                                              */
int vae_gn_bwd_apply(const void* x, int32_t x_bf16, const void* g, const float* mean, const float* rstd,
                     const float* gamma, const float* beta, const float* coef, const void* add,
                     int32_t B, int32_t HW, int32_t C, int32_t G, int32_t silu, int32_t g_bf16, float* dx, void* dx16, void* stream);

/* ---- attention softmax (K3) ---- */
int vae_softmax_rows(float* S, int64_t rows, int32_t cols, void* stream);           /* in place */
/* dS = P*(dP - rowsum(dP*P)) in place on dP                                        */
int vae_softmax_bwd_rows(const float* P, float* dP, int64_t rows, int32_t cols, void* stream);

/* ---- blockwise (flash-style) attention, single head of width C = 512 (K3; no T x T score matrix) ----
 * replaces: scaled_dot_product_attention forward / backward inside diffusers' Attention (call sites
 * sdxl_vae_wrapper.py:60,71) for sequences where the materialised scores would not be reasonable (T = 16384 at 1024x1024).
 * q, k, v (and dout): [B][T][C]; fp32 when prec == VAE_PREC_F32 (exact fp32 MFMA), bf16 images (vae_pack_bf16) when
 * prec == VAE_PREC_BF16.  o, dq, dk, dv: fp32 [B][T][C]; lse [B][T] = row log-sum-exp of the scaled scores (saved by the
 * forward, P is recomputed from it in the backward); dsum [B][T]: workspace (rowsum(dO*O), written here).           */
int vae_attn_supported(int32_t T, int32_t C);  /* 1 when the kernels serve this shape (C == 512, T % 64 == 0) */
int vae_attn_fwd(const void* q, const void* k, const void* v, int32_t B, int32_t T, int32_t C, float scale, int32_t prec,
                 float* o, float* lse, void* stream);
/* o32 / do32: the fp32 output of the forward and its gradient (for dsum); dout: do32 in the operand precision */
int vae_attn_bwd(const void* q, const void* k, const void* v, const void* dout, const float* o32, const float* do32,
                 const float* lse, int32_t B, int32_t T, int32_t C, float scale, int32_t prec,
                 float* dq, float* dk, float* dv, float* dsum, void* stream);

/* ---- posterior sample + KL + MSE (K4,K6; train.py:289-291) ---- */
/* moments [B][hw][2*L] -> z [B][hw][L] = mu + exp(.5*clamp(lv,-30,20))*eps (eps may be null => mode())
 * kl_partial [B][nblk] ; nblk = ceil(hw*L/256)                                       */
int vae_sample_kl(const float* moments, const float* eps, int32_t B, int32_t hw, int32_t L,
                  float* z, float* kl_partial, void* stream);
/* scalars[0]=mse(mean) [1]=kl(mean over b) [2]=total ; deterministic finalisation   */
int vae_mse_partial(const float* recon, const float* target, int64_t n, float* ws, int32_t nblk, void* stream);
int vae_loss_final(const float* mse_ws, int32_t mse_nblk, int64_t mse_n, const float* kl_partial,
                   int32_t B, int32_t kl_nblk, float kl_weight, float* scalars, void* stream);
/* scale * d(total)/d(recon) = scale*2*(recon-target)/n  (scale = 1/gradient_accumulation_steps, train.py:286,299) */
int vae_mse_bwd(const float* recon, const float* target, int64_t n, float scale, float* drecon, void* stream);
/* d(total)/d(moments) from dz and the KL term                                        */
int vae_sample_kl_bwd(const float* moments, const float* eps, const float* dz, int32_t B, int32_t hw,
                      int32_t L, float kl_weight, float* dmoments, void* stream);

/* ---- layout / misc ---- */
int vae_nchw_to_nhwc(const float* src, int32_t B, int32_t C, int32_t HW, int32_t Cpad, float* dst, void* stream);
int vae_nhwc_to_nchw(const float* src, int32_t B, int32_t C, int32_t HW, float* dst, void* stream);
/* dst[b][y][x][c] = sum of the 2x2 block of src[b][2y..][2x..][c] (dgrad of nearest-up2x) */
int vae_sumpool2x2(const float* src, int32_t B, int32_t H, int32_t W, int32_t C, float* dst, void* stream);
int vae_add(const float* a, const float* b, int64_t n, float* out, void* stream);
/* the same on bf16 tensors (fp32 sum, one rounding); n % 4 == 0 */
int vae_add_bf16(const void* a, const void* b, int64_t n, void* out, void* stream);
/* dst[i] = float(src16[i]): a bf16-stored activation for a consumer that needs fp32 (foreign hooks, fp32-only kernels) */
int vae_unpack_bf16(const void* src16, int64_t n, float* dst, void* stream);
/* dst[i] = bf16(src[i]) (round to nearest even): the bf16 weight image the bf16 kernels read through vae_igemm_args.Wh;
 * run on the whole parameter arena once per step (replaces the per-step autocast weight casts of the reference) */
int vae_pack_bf16(const float* src, int64_t n, void* dst, void* stream);

/* ---- input transform (replaces the per-item CPU chain of src/data_utils.py:24-30) ----
 * Resize(shorter side -> R, bilinear exactly as Pillow resamples it) -> CenterCrop(R) -> RGB -> ToTensor -> Normalize(.5,.5)
 * for `n` uint8 images of one size, src [n][H][W][C] (C = 3 RGB, or 1 = grey replicated), out fp32 [n][3][R][R] in [-1,1].
 * bounds_x [R][2] = {first source column, count}, kk_x [R][ksx]: Pillow's 22-bit fixed-point coefficients of the R output
 * COLUMNS inside the centre crop; bounds_y / kk_y likewise for the R output rows (built by vaehip/preprocess.py from the
 * sizes).  Only source rows [row0, row0 + nrows) feed the crop; tmp: workspace u8 [n][nrows][R][3] (the 8-bit
 * intermediate between the two passes).  Integer arithmetic: bit-identical to the CPU path.                          */
int vae_preprocess_u8(const uint8_t* src, int32_t n, int32_t H, int32_t W, int32_t C, int32_t R,
                      const int32_t* bounds_x, const int32_t* kk_x, int32_t ksx,
                      const int32_t* bounds_y, const int32_t* kk_y, int32_t ksy,
                      int32_t row0, int32_t nrows, uint8_t* tmp, float* out, void* stream);

/* ---- optimizer (K8,K9; train.py:184-187,301-302) ---- */
/* sum of squares of g[0..n): stage 1 -> ws[nblk], stage 2 -> out[0]                   */
int vae_sqnorm(const float* g, int64_t n, float* ws, int32_t nblk, float* out, void* stream);
/* clip coefficient min(1, max_norm/(sqrt(*sqnorm)+1e-6)) is read from device memory
 * (max_norm <= 0 disables clipping) then torch.optim.AdamW math, one pass over p,g,m,v */
int vae_adamw(float* p, const float* g, float* m, float* v, int64_t n, const float* sqnorm,
              float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay,
              int32_t step, void* stream);
/* dead-weight scan (deadneuron.py:78-115): per segment, the count of |w| < thr and the sum of |w|.
 * seg_off [nseg][2] = {begin,end} element offsets into w (segments need not be adjacent).  The segments are scanned in
 * chunks of vae_dead_scan_chunk() elements, one workgroup per chunk: seg_chunk0 [nseg+1] = prefix sum of the segments'
 * chunk counts (ceil(len / chunk)), nchunk = seg_chunk0[nseg]; part_counts / part_abssum [nchunk]: workspace.  A second
 * kernel adds each segment's partials in chunk order: out_counts [nseg] (u64), out_abssum [nseg] (double), reproducible. */
int vae_dead_scan_chunk(void);
int vae_dead_scan(const float* w, const int64_t* seg_off, const int32_t* seg_chunk0, int32_t nseg, int32_t nchunk, float thr,
                  unsigned long long* part_counts, double* part_abssum, unsigned long long* out_counts, double* out_abssum,
                  void* stream);
/* counts of |w| < adaptive_thr[seg] (and < thr when use_fixed): the "percent_of_mean" / "both" modes */
int vae_dead_scan_adaptive(const float* w, const int64_t* seg_off, const int32_t* seg_chunk0, int32_t nseg, int32_t nchunk, float thr,
                           int32_t use_fixed, const float* adaptive_thr, unsigned long long* part_counts,
                           unsigned long long* out_counts, void* stream);

#ifdef __cplusplus
}
#endif
#endif
