/* libpeppa_hip.so -- C ABI of the MI355X (gfx950) hot path of the PeppaPig training step.
 *
 * The reference (gchrupala/peppa) has no FFI: its boundary is the Python API of
 * pig/models.py, pig/loss.py, pig/optimization.py.  Each entry point below
 * names the reference call it replaces (file:line under the reference tree).  The library
 *   - borrows raw device pointers for the duration of the call (PyTorch owns all memory),
 *   - only enqueues work on the given hipStream_t, never synchronises, never allocates,
 *   - returns 0 or a negative pp_status; pp_last_error() gives the message (thread-local).
 * Activations are bf16 (raw uint16 bits), channels-last, channel stride padded to a multiple
 * of 16; statistics, norms, loss and optimizer state are fp32.
 */
#ifndef PEPPA_HIP_H
#define PEPPA_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* pp_stream_t; /* hipStream_t */
enum pp_status { PP_OK = 0, PP_ERR_INVALID = -1, PP_ERR_UNSUPPORTED = -2, PP_ERR_HIP = -3 };
enum pp_act { PP_ACT_NONE = 0, PP_ACT_GELU = 1, PP_ACT_RELU = 2 };
enum pp_gather_mode { PP_DENSE = 0, PP_CONV_FWD = 1, PP_CONV_DGRAD = 2 };

int pp_version(void);
/* The 16-bit operand / activation type of THIS library: libpeppa_hip.so = bf16 (BASELINE configs[1]),
 * libpeppa_hip_f16.so = the same sources built with -DPP_F16, IEEE half as in the reference's `precision: 16` AMP runs
 * (hparams_base.yaml:45, BASELINE configs[4]).  Wherever this header says "bf16" for an operand it means that type. */
enum pp_dtype_t { PP_DTYPE_BF16 = 0, PP_DTYPE_F16 = 1 };
int pp_dtype(void);
/* non-zero when the library was compiled with one of the experiment macros of tools/probe/ (PP_WIN_ABLATE bit 0,
 * PP_TW_ABLATE bit 1, PP_LN_VARIANT bit 2): ablation builds whose RESULTS ARE WRONG by design, or kernel variants under
 * test.  The Python binding refuses such a library unless PEPPA_ALLOW_EXPERIMENTAL=1; the shipped build returns 0. */
int pp_experimental_build(void);
/* tuning switches (process-wide; also PEPPA_HIP_OPTIONS="name=value,..." through the Python loader):
 *   "xcd_remap_igemm", "xcd_remap_wgrad"  0/1   XCD-contiguous tile order
 *   "persistent_igemm"                    0/1   persistent workgroups with cross-tile prefetch (plain epilogues)
 *   "ring_igemm"   n   LDS-DMA ring GEMM once there are >= n 256-row tiles (0 = never; default 128)
 *   "ring_wn"      w   dense ring tile width in 16-column units: 6, 8 or 9 (0 = chosen per problem, the default)
 *   "win_igemm"    n   window kernel for (1,3,3) stride-1 convs, forward / data gradient, once M >= n (default 1024)
 *   "win_temporal" 0/1 the same kernel in its temporal form for (3,1,1) stride-1 convs with 4 / 8 / 16 frames (default 1)
 *   "sw_wgrad"     n   sliding-window weight gradients ((1,3,3) and (3,1,1) stride-1 convs) once M >= n (default 4096;
 *                      1 also takes shapes the (3,1,1) kernel would decline as not worth it)
 *   "ring_wgrad"   n   LDS-DMA ring variant of the generic weight gradient once M >= n (default 0 = never)
 *   "win_tall"     0/1/2  512-row window tiles for narrow outputs (0: never, the default since round 4 -- the 256-row tile
 *                      reads its operand once; 1: with enough rows; 2: always)
 *   "win_s2d"      0/1 stride-(1,2,2) spatial data gradients as ONE window-kernel launch that reads dy once (default 1)
 *   "win_partial"  0/1 window kernels also for channel counts that are no multiple of 48 / 64 (464, 928): 64-channel chunks,
 *                      the last one partial (default 1)
 *   "win_kpb"      1/2 K-steps per workgroup barrier in the 64-column window tiles (layer-1 data gradient, S2D): 2 (default)
 *                      = a ring slot holds two K-steps, one wait + barrier + DMA issue per pair
 *   "win_ragged"   0/1 temporal window form for frame counts / sizes that do not tile evenly (default 1)
 *   "wgrad_group_ring" 0/1 grouped weight gradients on the ring kernel's 128 x 256 tiles (default 0: measured slower)
 *   "deterministic" 0/1  bitwise-reproducible results: every sum that crosses workgroups is taken in a fixed order instead
 *                      of by fp32 atomics -- weight gradients through per-split slabs (pp_wgrad_desc.ws) and an ordered
 *                      pass, the wav2vec2 conv0 statistics / weight-norm sums / column sums by one workgroup per sum,
 *                      BertAdam's per-tensor norms and the loss over per-block partials, the fp32 head GEMMs without split-K.
 *                      Default 0: the atomics are faster (cost: DESIGN.md section 7); LayerNorm and BatchNorm reductions
 *                      are ordered in either mode
 *   "win_stagger"  0/1 window kernels with three weight slots: waves 4-7 request their fragments ahead of the K-step's
 *                      barrier, so that the two halves' LDS reads and MFMAs alternate instead of coinciding
 *   "win_producers" 0..3  window kernels with four extra producer waves that issue every LDS-DMA (the eight multiplying
 *                      waves issue none): 1 = spatial form, tiles up to 128 columns; 2 = every spatial tile; 3 = 1 + the
 *                      temporal form; 4 = 2 + the temporal form (default); 0 = the lockstep
 *                      kernels
 *   "ring_producers" 0/1 LDS-DMA ring GEMM / gather kernels with four producer waves, tiles up to 128 columns (default 1)
 *   "tw_producers" 0/1 temporal sliding-window weight gradient with three producer waves (default 1)
 *   "tw_narrow"    0/1 its 48-channel form (deep look-ahead) for convolutions with at most 48 input channels (default 1)
 *   "ln_bwd_alone" 0/1  pp_layernorm_bwd reserves LDS it never touches so that no LDS-using kernel shares its CUs
 *                      (default 0; an A/B switch from the investigation in DESIGN.md section 7)
 *   "win_out_nt"   0/1 non-temporal stores of the window kernels' output tiles (default 1)
 *   "bn_nt" b, "bn_grid" n   BatchNorm streaming passes: non-temporal loads (bit 0) / stores (bit 1), workgroups per launch
 *   "persist_cus"  n   workgroups of the persistent ring / window kernels (8..256, default 256 = one per CU)
 *   "wgrad_flat"   0/1 grouped weight gradients (pp_wgrad_desc.ptr_table) walk (problem, tile) as one flat grid in
 *                      slab-sharing order (default 1; 0 = one problem per grid.z slice, round 3's order)
 *   "wgrad_big"    m   pp_wgrad on 256 x 256 tiles for problems that fill the chip with them and whose reduce dimension has
 *                      at least m rows (default 32768; grouped launches and problems of >= 32 tiles from 4096; 0 = never);
 *                      never in deterministic mode
 *   "igemm_big"    m   pp_igemm on 256 x 256 tiles from M = |m| for problems of >= 192 tiles: m > 0 plain epilogues only,
 *                      m < 0 fused ones too; bit-identical to the default kernels (default 0 = never: DESIGN.md section 8) */
int pp_set_option(const char* name, int value);
const char* pp_last_error(void);

/* Row gather shared by pp_igemm / pp_wgrad: how logical row m and reduce index k of the
 * left operand map to memory.
 *   PP_DENSE      : A[m*lda + k].
 *   PP_CONV_FWD   : m -> (n, rt, rh, rw) over (Rt,Rh,Rw); k -> (tap, c), tap -> (dt,dh,dw);
 *                   source position g = r*stride - pad + d inside (Gt,Gh,Gw), else zero.
 *   PP_CONV_DGRAD : source position g = (r + pad - d)/stride when divisible and in range.
 * The source tensor is channels-last [n][Gt][Gh][Gw][cstride] (bf16); `cg` channels are
 * read per tap (multiple of 8).  1-D convolutions use Gh=Gw=Rh=Rw=1.
 * Replaces torch.nn.Conv3d / Conv1d / Linear inside torchvision r2plus1d_18 and torchaudio
 * wav2vec2_base as called from pig/models.py:141-150 and pig/models.py:101-105. */
typedef struct pp_gather {
  int mode;
  int lda;                /* PP_DENSE: row stride in elements */
  int Rt, Rh, Rw;         /* row decomposition extents */
  int Gt, Gh, Gw;         /* source tensor extents */
  int kt, kh, kw;         /* taps */
  int st, sh, sw;         /* strides (dgrad: 1 or 2) */
  int pt, ph, pw;         /* paddings */
  int cg;                 /* channels per tap */
  int cstride;            /* channel stride of the source tensor */
} pp_gather;

/* C[M,N] = act(A_gather[M,K] * Bt[N,K]^T + bias) (+ residual).  bf16 in, fp32 accumulate.
 * grid.z runs `nbatch` independent problems; batch z uses offsets
 * (z / inner) * stride0 + (z % inner) * stride1 on A, Bt, C, bias (elements). */
typedef struct pp_igemm_desc {
  int M, N, K;
  pp_gather g;
  const void* A;
  const void* Bt; int ldb; int b_rows;   /* Bt is [b_rows][ldb], rows >= b_rows read as 0 */
  void* C; int ldc; int c_fp32;          /* bf16 (default) or fp32 output */
  void* Cpre;                            /* optional: pre-activation copy (bf16, ldc) */
  const float* bias;                     /* optional [N] */
  int act;
  const void* residual; int ldr;         /* optional bf16 [M][ldr], added after act */
  float* colstats; int ldstat;           /* optional [ceil(M/128)][2][ldstat] partial col sums */
  int nbatch, inner;
  long long a_s0, a_s1, b_s0, b_s1, c_s0, c_s1, bias_s0, bias_s1;
  /* optional output row map (conv modes): compact row (n, rt, rh, rw) is stored at row
   * ((n*Ot + rt*os_t + oo_t)*Oh + rh*os_h + oo_h)*Ow + rw*os_w + oo_w of C (and read there from `residual`).
   * Used by the parity-class decomposition of stride-2 data gradients. */
  int omap, Ot, Oh, Ow, os_t, os_h, os_w, oo_t, oo_h, oo_w;
  /* optional dropout in the epilogue (bf16 outputs with bias / activation / residual): C = dropout(act(AB + bias)) +
   * residual, with the mask pp_dropout_bf16 would apply to the flat [M][ldc] tensor for the same (p, seed) */
  float drop_p; unsigned drop_seed;
  /* optional (data gradients): the BatchNorm-backward sums of the layer that CONSUMES this output as its dz, taken in the
   * epilogue while the tile is still on chip instead of by a separate pass over dz (pp_bn_bwd_reduce):
   *   g = C * mask,  mask = [ (bnr_z ? bnr_z : bnr_y * scale + shift) > 0 ] if bnr_relu else 1,  xhat = (bnr_y - mean) * rstd
   *   bnr_partials[t][0][n] = sum_rows g,  bnr_partials[t][1][n] = sum_rows g * xhat   over rows [256 t, 256 t + 256)
   * bnr_y / bnr_z: bf16 [M][ldc] like C; mean / rstd / scale / shift: fp32 [ldc]; bnr_partials: fp32 [ceil(M/256)][2][ldc].
   * Only the window kernels implement it: pp_igemm returns PP_BNR_SKIPPED (the product IS computed) when the kernel it
   * dispatched to does not, and the caller runs pp_bn_bwd_reduce as before. */
  const void* bnr_y; const void* bnr_z;
  const float* bnr_mean; const float* bnr_rstd; const float* bnr_scale; const float* bnr_shift;
  int bnr_relu; float* bnr_partials;
  /* optional (conv forward): A is the RAW output y of a BatchNorm unit and the kernel reads z = relu?(y * a_bn_scale +
   * a_bn_shift) instead -- applied to each activation window once, in LDS, so the unit's activated tensor is never
   * written or read (the convolution's zero padding applies to z).  fp32 [cg] each.  Only the temporal window kernel
   * implements it: ask pp_igemm_abn_supported(d) first; pp_igemm fails with PP_ERR_INVALID for any other problem. */
  const float* a_bn_scale; const float* a_bn_shift; int a_bn_relu;
} pp_igemm_desc;
#define PP_BNR_SKIPPED 2
int pp_igemm(const pp_igemm_desc* d, pp_stream_t s);
/* 1 if pp_igemm would run `d` (sizes, gather, epilogue; pointers are not looked at) with a_bn_scale / a_bn_shift, else 0 */
int pp_igemm_abn_supported(const pp_igemm_desc* d);
/* rows of `colstats` ([rows][2][ldstat] fp32 partial column sums) pp_igemm writes for `d` -- what the caller allocates and
 * pp_bn_finalize then reduces: one per 128 output rows, except where the temporal window kernel takes the problem with a frame
 * count / frame size that does not tile evenly (two per TILE of 256 / T positions x all frames: a few more).  Set every
 * field pp_igemm will see (colstats may be any non-null value). */
long long pp_igemm_stat_rows(const pp_igemm_desc* d);

/* dW[Ni,Kj] += sum_m dY[m,Ni]^T * X_gather[m,Kj]  (weight gradients; fp32 atomics).
 * Replaces autograd's conv/linear weight-gradient for the same modules. dW must be zeroed. */
typedef struct pp_wgrad_desc {
  int M, Ni, Kj;
  pp_gather g;                    /* gather of X (same geometry as the forward) */
  const void* X;
  const void* dY; int ldy;
  float* dW; int ldw;
  int msplit;                     /* 0 = choose automatically */
  int nbatch;                     /* grouped conv: groups */
  long long x_s, dy_s, dw_s;      /* per-batch offsets (elements) */
  float* dbias;                   /* optional: dbias[Ni] += column sums of dY (zeroed by the caller) */
  long long dbias_s;
  /* optional: X is the RAW output y of a BatchNorm unit; the kernel uses relu?(y * x_bn_scale + x_bn_shift) (see
   * pp_igemm_desc.a_bn_scale).  Only the temporal sliding-window kernel implements it: pp_wgrad_xbn_supported(d). */
  const float* x_bn_scale; const float* x_bn_shift; int x_bn_relu;
  /* optional GROUPED launch: `nbatch` independent problems of the same (M, Ni, Kj, gather) whose operands live anywhere:
   * a device table of nbatch x 4 pointers {X, dY, dW, dbias (or 0)}; X / dY / dW / dbias and the per-batch offsets above
   * are then ignored (dbias != NULL still says "take the bias gradients").  One launch covers e.g. a weight gradient of all
   * twelve transformer layers: every (i, j) tile reduces its whole M (msplit = 1), so no sum crosses workgroups and
   * the result is bitwise reproducible.  Dense gathers with the generic kernel only. */
  const unsigned long long* ptr_table;
  /* deterministic mode (pp_set_option("deterministic", 1)): where M is split over workgroups, every split stores its
   * partial tile to a slab of `ws` (fp32, pp_wgrad_ws_floats(d) elements, contents irrelevant) and an ordered pass adds the
   * slabs into dW / dbias; without the mode, or where nothing is split, ws is not touched and may be NULL. */
  float* ws; long long ws_floats;
} pp_wgrad_desc;
int pp_wgrad(const pp_wgrad_desc* d, pp_stream_t s);
/* workspace (floats) pp_wgrad needs for `d` in the deterministic mode; 0 when it needs none (mode off, or no split) */
long long pp_wgrad_ws_floats(const pp_wgrad_desc* d);
int pp_wgrad_xbn_supported(const pp_wgrad_desc* d);

/* ---- weight preparation (fp32 master -> bf16 operand layouts) and gradient un-preparation */
/* w [Co][Ci][taps] fp32 -> out [rows_out][taps][cg] bf16 (zero padded), optional tap flip.
 * transpose_io=1 builds the dgrad operand [Ci][taps(flipped)][cog]. */
int pp_prep_conv_weight(const float* w, int Co, int Ci, int taps, void* out, int rows_out, int cg,
                        int transpose_io, int flip, float scale, pp_stream_t s);
/* The paired-pixel stem convolution (Conv3d(3, Co <= 48, (1,7,7), stride (1,2,2), padding (0,3,3)) over the input
 * pp_video_normalize_ndhwc4 writes and the weights pp_prep_conv_weight_pairs lays out) as a window kernel: x [images][Hi][Wp
 * pairs][8] bf16, wf [Co][7 x 4 taps][8], y [images * Ho * Wp][ldc] bf16 with Ho = (Hi - 1) / 2 + 1; Wp <= 64.  colstats (or
 * NULL): [pp_stem_pairs_stat_rows(images, Hi)][2][ldstat] fp32 partial column sums / sums of squares, one row per (image, four
 * output rows).  Same bits in y as pp_igemm on the same operands (same K order).  torchvision stem[0], pig/models.py:141-150 */
long long pp_stem_pairs_stat_rows(int images, int Hi);
int pp_stem_pairs_fwd(const void* x, const void* wf, void* y, float* colstats, int images, int Hi, int Wp, int Co, int ldc,
                      int ldstat, pp_stream_t s);
/* its weight gradient on the same window: dw [Co][7 x 4 taps][8] fp32 (pp_unprep_conv_grad_pairs' input) += sum over the
 * output positions; the caller zeroes dw; dy [images * Ho * Wp][ldy] bf16.  fp32 atomics across workgroups */
int pp_stem_pairs_wgrad(const void* x, const void* dy, float* dw, int images, int Hi, int Wp, int Co, int ldy, pp_stream_t s);
/* n contiguous fp32 copies in one launch (the gradient buckets' packing).  Item = 4 x int64 {src, dst, n floats, blk0}: one
 * block copies 4096 consecutive floats, item i owns blocks [blk0_i, blk0_(i+1)), total_blocks = their sum */
int pp_copy_f32_multi(const void* items, int n, long long total_blocks, pp_stream_t s);
/* pp_prep_conv_weight for a DEVICE table of n weights in one launch (a tower's convolutions, both layouts).  Item = 88 bytes:
 * 10 x int64 {w (const float*), out, Co, Ci, taps, rows_out, cg, transpose_io, flip, blk0}, float scale, 4 bytes of padding.
 * One block converts 2048 consecutive elements of one `out`: item i owns blocks [blk0_i, blk0_(i+1)), blk0 ascending from 0,
 * total_blocks = their sum */
int pp_prep_conv_weight_multi(const void* items, int n, long long total_blocks, pp_stream_t s);
/* out[r][i][cg] = w[r][sel[i]][cg] (bf16): tap subset of a conv operand, nsel <= 32 (sel is a HOST array) */
int pp_select_taps(const void* w, int rows, int taps, int cg, const int* sel, int nsel, void* out, pp_stream_t s);
/* g [Co][taps][cg] fp32 (pp_wgrad layout) -> dw [Co][Ci][taps] fp32, dw = g (beta=0) or += */
int pp_unprep_conv_grad(const float* g, int Co, int Ci, int taps, int cg, float* dw, pp_stream_t s);
/* generic 2-D fp32 -> bf16 copy with padding / transpose: out[r*ld_out + c] = in[r][c] (or in[c][r]) for
 * r < rows_out, c < cols_out; zero outside rows x cols */
int pp_cast_pad_2d(const float* in, int rows, int cols, int ld_in, void* out, int rows_out, int cols_out,
                   int ld_out, int transpose, pp_stream_t s);
/* the same for a DEVICE table of n jobs in one launch.  Job = 10 x int64: {in (const float*), out, rows, cols, ld_in,
 * rows_out, cols_out, ld_out, transpose, out_f32 (1: fp32 output, a strided copy; 0: bf16)} */
int pp_cast_pad_2d_multi(const void* items, int n, int blocks_per_item, pp_stream_t s);
int pp_cast_f32_to_bf16(const float* in, void* out, long long n, pp_stream_t s);
int pp_cast_bf16_to_f32(const void* in, float* out, long long n, pp_stream_t s);
/* strided fp32 2-D copy: out[r*ld_out + c] = in[r*ld_in + c] */
int pp_copy_2d_f32(const float* in, int ld_in, float* out, int ld_out, int rows, int cols, pp_stream_t s);
/* batched bf16 transpose: in [nb][R][ld_in] (cols C) -> out [nb][C][ld_out]; columns R..r_pad-1 of each
 * output row are zero-filled (r_pad = 0 means ld_out) */
int pp_transpose_bf16(const void* in, long long in_bs, int ld_in, void* out, long long out_bs, int ld_out,
                      int nb, int R, int C, int inner, long long in_s1, long long out_s1, int r_pad,
                      pp_stream_t s);
int pp_fill_f32(float* p, float v, long long n, pp_stream_t s);

/* ---- video input transform: pig/models.py:327-342 build_transform + layout change ------
 * x fp32 [B][3][T][H][W] in [0,1] -> out bf16 [B][T][H][W][8], (x-mean)/std, channels 3..7 = 0 */
int pp_video_normalize_ndhwc(const float* x, void* out, int B, int T, int H, int W,
                             const float* mean3 /* HOST */, const float* std3 /* HOST */, pp_stream_t s);

/* ---- on-device collate: pig/data.py:60-78 featurize + collate, pig/util.py:19-33 pad_*_batch -------
 * `items` is a DEVICE table of n clips, 2 x int64 each: {pointer, length}.  Byte/integer work, bit-exact.
 * video: clip i = uint8 [T_i][H][W][3] frames as the decoder delivers them (length = T_i) ->
 *        out fp32 [n][3][Tmax][H][W] = frame / 255 (float64 division rounded to fp32, as numpy + .float() do),
 *        zero frames after T_i (pad_video_batch).  4-byte-aligned clips with H*W % 4 == 0 take the vector path. */
int pp_collate_video_u8(const void* items, int n, int Tmax, int H, int W, float* out, pp_stream_t s);
/* rows: clip i = `length` bytes -> out [n][row_bytes], zero after `length` (pad_audio_batch on fp32 samples, or the
 *       uint8 batch [n][Tmax][H][W][3] that pp_video_normalize_u8_ndhwc reads) */
int pp_collate_rows(const void* items, int n, long long row_bytes, void* out, pp_stream_t s);
/* x uint8 [B][T][H][W][3] -> out bf16 [B][T][H][W][8]: /255, (x-mean)/std, channels 3..7 = 0; bit-identical to
 * pp_collate_video_u8 followed by pp_video_normalize_ndhwc without the fp32 batch in between */
int pp_video_normalize_u8_ndhwc(const void* x, void* out, int B, int T, int H, int W,
                                const float* mean3 /* HOST */, const float* std3 /* HOST */, pp_stream_t s);
/* The same two with FOUR channels per pixel (3 real + 0): out [B*T*H*W][4].  Two pixels then share a 16-byte chunk, and a
 * stride-2 first convolution over [W][4] is run as a stride-1 convolution over the pixel PAIRS [W/2][8] of the same
 * memory with weights laid out by pp_prep_conv_weight_pairs: kernel width kw -> kwp = floor((kw-1-pw)/2) - floor(-pw/2) + 1
 * pair taps (7 -> 4 at pw = 3), left padding pj = -floor(-pw/2) (2), output width forced to the original Wo (the
 * gather's bounds check supplies the right-hand padding).  K of the (1,7,7) stem: 49 x 8 -> 28 x 8. */
int pp_video_normalize_ndhwc4(const float* x, void* out, int B, int T, int H, int W,
                              const float* mean3, const float* std3, pp_stream_t s);
int pp_video_normalize_u8_ndhwc4(const void* x, void* out, int B, int T, int H, int W,
                                 const float* mean3, const float* std3, pp_stream_t s);
/* w fp32 [Co][Ci <= 4][kth][kw] -> out 16-bit [Co][kth * kwp][8]: element q * 4 + c of pair tap dj = w[co][c][a][2 (dj - pj) + q + pw]
 * (0 outside the kernel row / for c >= Ci); and back for the gradient: g fp32 [Co][kth * kwp][8] -> dw [Co][Ci][kth][kw] */
int pp_prep_conv_weight_pairs(const float* w, int Co, int Ci, int kth, int kw, int pw, void* out, pp_stream_t s);
int pp_unprep_conv_grad_pairs(const float* g, int Co, int Ci, int kth, int kw, int pw, float* dw, pp_stream_t s);

/* MaxPool2d(3, 2, 1) of torchvision resnet18 (static ImageEncoder, pig/models.py:181-186) on channels-last
 * bf16 [N][H][W][Cp]; the backward routes each window's gradient to its first maximum (PyTorch's rule) */
int pp_maxpool3x3s2_fwd(const void* x, void* y, int N, int H, int W, int Cp, pp_stream_t s);
int pp_maxpool3x3s2_bwd(const void* x, const void* dy, void* dx, int N, int H, int W, int Cp, pp_stream_t s);

/* ---- BatchNorm3d (train mode) : torchvision BN layers inside R3DEncoder ---------------- */
/* out[2][ld] = sum over the nblk rows of a partials table [nblk][2][ld] (ws: [64][2][ld] scratch).  With SyncBN the
 * caller all-reduces `out` across ranks (RCCL, SUM) and finalizes it as a one-row table with the global count. */
int pp_partials_sum(const float* partials, int nblk, int ld, float* ws, float* out, pp_stream_t s);
/* reduce igemm colstats partials -> mean, rstd, scale=gamma*rstd, shift=beta-mean*scale;
 * updates running stats (momentum, unbiased var). C real channels, Cp padded (scale/shift=0). */
int pp_bn_finalize(const float* partials, int nblk, int ldstat, long long count, int C, int Cp,
                   const float* gamma, const float* beta, float eps, float momentum,
                   float* running_mean, float* running_var, float* mean, float* rstd, float* scale,
                   float* shift, float* ws /* optional 64*2*ldstat floats: parallel first-level reduce */,
                   pp_stream_t s);
/* eval mode: scale/shift from the running statistics */
int pp_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean,
                      const float* running_var, float eps, int C, int Cp, float* scale, float* shift,
                      pp_stream_t s);
/* stats straight from a bf16 tensor [M][Cp] (used when no igemm epilogue produced them) */
int pp_colstats_bf16(const void* y, long long M, int Cp, float* partials, int nblk, pp_stream_t s);
/* z = relu?(y*scale+shift (+res)) */
int pp_bn_apply(const void* y, const float* scale, const float* shift, const void* res, int relu, void* z,
                long long M, int Cp, pp_stream_t s);
/* backward: pass 1 partial sums of dzm and dzm*xhat (dzm = dz masked by z>0 when relu) */
/* z may be NULL when there was no residual: the ReLU mask is then recomputed as y*scale+shift > 0 */
int pp_bn_bwd_reduce(const void* dz, const void* y, const void* z, const float* mean, const float* rstd,
                     const float* scale, const float* shift, int relu, float* partials, int nblk, long long M,
                     int Cp, pp_stream_t s);
/* pass 2: sums -> dgamma,dbeta (C real channels), coefficient vectors for the apply pass.  ws: 64 * 2 * Cp floats or NULL;
 * with it and nblk > 256 the partial rows are summed in two levels (64 slices in parallel, fixed order), like pp_bn_finalize */
int pp_bn_bwd_finalize(const float* partials, int nblk, long long count, int C, int Cp, const float* gamma,
                       const float* rstd, float* dgamma, float* dbeta, float* coef, float* ws, pp_stream_t s);
/* pass 3: dy = gamma*rstd*(dzm - mean(dzm) - xhat*mean(dzm*xhat)); optional dres = dzm */
int pp_bn_bwd_apply(const void* dz, const void* y, const void* z, const float* mean, const float* rstd,
                    const float* coef, const float* scale, const float* shift, int relu, void* dy, void* dres,
                    long long M, int Cp, pp_stream_t s);

/* ---- elementwise on bf16 [n] ------------------------------------------------------------ */
int pp_gelu_fwd(const void* x, void* y, long long n, pp_stream_t s);
int pp_gelu_bwd(const void* dy, const void* x, void* dx, long long n, pp_stream_t s);
/* dx = dropout_bwd(dy; p, seed) * gelu'(x): the mask of pp_dropout_bf16 with the same (p, seed), one pass */
int pp_gelu_bwd_dropout(const void* dy, const void* x, void* dx, long long n, float p, unsigned seed, pp_stream_t s);
int pp_add_bf16(const void* a, const void* b, void* out, long long n, pp_stream_t s);
/* dropout with a counter-based mask (seed, element index): y = keep ? x/(1-p) : 0 (+ res).  Calling it again with
 * the same seed on the gradient is the backward pass.  n multiple of 8; x and y may alias. */
int pp_dropout_bf16(const void* x, const void* res, void* y, long long n, float p, unsigned seed, pp_stream_t s);
int pp_dropout_f32(const float* x, float* y, long long n, float p, unsigned seed, pp_stream_t s);
/* per-column sums of a bf16 matrix [M][ld] -> fp32 out[N] (bias gradients); out is overwritten */
int pp_colsum_bf16(const void* x, long long M, int N, int ld, float* out, pp_stream_t s);

/* ---- LayerNorm over the last dim (torchaudio wav2vec2 components) ---------------------- */
int pp_layernorm_fwd(const void* x, const float* gamma, const float* beta, float eps, void* y, float* mean,
                     float* rstd, int rows, int D, pp_stream_t s);
/* dgamma/dbeta are accumulated (+=); zero them first.  ws = optional scratch [ws_blocks][2][D] fp32: per-workgroup
 * partials + a second pass (deterministic, no same-address atomics); NULL -> atomics */
int pp_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                     void* dx, float* dgamma, float* dbeta, int rows, int D, float* ws, int ws_blocks, pp_stream_t s);

/* ---- softmax over attention scores (SelfAttention in wav2vec2) ------------------------- */
/* Fused attention core of torchaudio's SelfAttention (T <= 320 frames, 64-wide heads): qkv bf16 [B*T][3*heads*64]
 * (q | k | v), ctx / dctx bf16 [B*T][heads*64], dqkv like qkv.  ctx = dropout(softmax(scale * q k^T)) v per (clip, head);
 * the backward recomputes the probabilities and regenerates the dropout mask from (seed, element index of the
 * [B*heads][T][Tp] probability tensor, Tp = T rounded up to 16) -- the same stream pp_dropout_bf16 would use. */
int pp_attention_fwd(const void* qkv, int B, int T, int heads, float scale, float drop_p, unsigned seed, void* ctx,
                     float* lse /* out: [B*heads][T] log-sum-exp of the scaled scores, kept for the backward pass */,
                     pp_stream_t s);
/* ctx = the forward's output, lse = its log-sum-exp rows; T <= 320 (T > 128: 2 x 2 or 3 x 3 blocks of 128 in one workgroup) */
int pp_attention_bwd(const void* qkv, const void* ctx, const float* lse, const void* dctx, int B, int T, int heads,
                     float scale, float drop_p, unsigned seed, void* dqkv, pp_stream_t s);
/* S fp32 [nb][T][lds] -> P bf16 [nb][T][ldp], P = softmax(scale*S) over T cols, pad cols = 0; T <= ldp <= 1024 (the unfused
 * attention path of clips beyond the fused kernels' 320 frames) */
int pp_softmax_fwd(const float* S, int lds, void* P, int ldp, int nb, int T, float scale, pp_stream_t s);
/* dS bf16 [nb][T][ldp] = scale * P * (dP - sum_j P*dP); dP fp32 [nb][T][lds] */
int pp_softmax_bwd(const float* dP, int lds, const void* P, int ldp, void* dS, int nb, int T, float scale,
                   pp_stream_t s);

/* ---- wav2vec2 feature extractor layer 0: Conv1d(1,512,10,5) + GroupNorm(512,512) + GELU  */
/* pass 1: per-(b,c) sum / sumsq of the conv output over time -> stats [B][512][2] (zeroed) */
int pp_conv0_stats(const float* wave, int B, int L, int T0, const float* w, float* stats, pp_stream_t s);
/* pass 2: recompute conv, normalise, affine, GELU -> out bf16 [B][T0][512] */
int pp_conv0_apply(const float* wave, int B, int L, int T0, const float* w, const float* stats,
                   const float* gamma, const float* beta, float eps, void* out, pp_stream_t s);
/* backward: dout bf16 [B][T0][512] -> dw [512][10], dgamma, dbeta (all accumulated, zero first).
 * red [B][512][2] scratch (zeroed). */
int pp_conv0_bwd_reduce(const float* wave, int B, int L, int T0, const float* w, const float* stats,
                        const float* gamma, const float* beta, float eps, const void* dout, float* red,
                        pp_stream_t s);
/* ws: fp32 [B][512 * 10], needed only in the deterministic mode (per-clip partial rows of dw, summed in clip order) */
int pp_conv0_bwd_apply(const float* wave, int B, int L, int T0, const float* w, const float* stats,
                       const float* gamma, const float* beta, float eps, const void* dout, const float* red,
                       float* dw, float* dgamma, float* dbeta, float* ws, pp_stream_t s);

/* ---- positional conv weight-norm (dim=2): w = g * v / ||v||_{(0,1)} --------------------- */
/* v [768][48][128], g [128] -> norm [128], out bf16 operand [768][128][48] (pp_igemm layout) */
int pp_weightnorm_fwd(const float* v, const float* g, int Co, int Ci, int Kk, float* norm, void* out,
                      pp_stream_t s);
/* dwt fp32 [Co][Kk][Ci] (pp_wgrad layout) -> dv [Co][Ci][Kk], dg [Kk] */
int pp_weightnorm_bwd(const float* dwt, const float* v, const float* g, const float* norm, int Co, int Ci,
                      int Kk, float* dv, float* dg, float* dot_ws /* [Kk] scratch */, pp_stream_t s);

/* ---- pooling heads: pig/models.py:30-43 Attention, :213-221 VideoAttention -------------- */
/* spatial mean: x bf16 [B][T][HW][Cp] -> out fp32 [B][T][C] */
int pp_spatial_mean_fwd(const void* x, float* out, int B, int T, int HW, int C, int Cp, pp_stream_t s);
int pp_spatial_mean_bwd(const float* dout, void* dx, int B, int T, int HW, int C, int Cp, pp_stream_t s);
/* pig/models.py:45-51 AveragePool = nn.AdaptiveAvgPool2d((S, 1)) on the 3-D (B, T, F) tensor (read by torch as (C, H, W)):
 * out[b][i] = mean over t in [floor(i T / S), ceil((i + 1) T / S)) and over ALL f of x[b][t][f]; fp32. */
int pp_avgpool_tf_fwd(const float* x, int B, int T, int F, int S, float* out, pp_stream_t s);
int pp_avgpool_tf_bwd(const float* dout, int B, int T, int F, int S, float* dx, pp_stream_t s);

/* attention pooling over time + Linear projection + L2 normalise (F.normalize eps 1e-12):
 * x fp32 [B][T][F]; W1 [Hd][F], b1 [Hd], W2 [F][Hd], b2 [F], Wp [E][F], bp [E] (Wp may be NULL)
 * saves alpha [B][T][F], hid [B][T][Hd], pooled [B][F], pre [B][E], out [B][E] */
int pp_attnpool_fwd(const float* x, int B, int T, int F, int Hd, int E, const float* W1, const float* b1,
                    const float* W2, const float* b2, const float* Wp, const float* bp, int normalize,
                    float* hid, float* alpha, float* pooled, float* pre, float* out, pp_stream_t s);
/* dW1..dbp and dx [B][T][F] are overwritten */
int pp_attnpool_bwd(const float* dout, const float* x, int B, int T, int F, int Hd, int E, const float* W1,
                    const float* W2, const float* Wp, int normalize, const float* hid, const float* alpha,
                    const float* pooled, const float* pre, const float* out, float* dx, float* dW1,
                    float* db1, float* dW2, float* db2, float* dWp, float* dbp,
                    float* ws /* pp_attnpool_ws_floats() floats */, pp_stream_t s);
size_t pp_attnpool_ws_floats(int B, int T, int F, int Hd, int E);

/* ---- pig/loss.py:33-55 TripletLoss = contrastive(cosine_matrix(V, A), margin) ---------- */
size_t pp_triplet_workspace_bytes(int N, int D);
/* V, A fp32 [N][D] (rows >= n_valid must not exist); loss[0] overwritten; ws keeps the state
 * pp_triplet_loss_bwd needs. */
int pp_triplet_loss_fwd(const float* V, const float* A, int N, int D, float margin, float* loss, void* ws,
                        size_t ws_bytes, pp_stream_t s);
/* dV, dA fp32 [N][D] = dloss * dL/dV, dL/dA (dloss read from device memory) */
int pp_triplet_loss_bwd(const float* V, const float* A, int N, int D, const float* dloss, const void* ws,
                        float* dV, float* dA, pp_stream_t s);
/* OPT-IN extension, not in the reference (its loss sums all negatives: pig/loss.py:41-48): in-batch hardest-negative
 * mining -- loss = (1/N) sum_i [relu(m + max_{j != i} S_ij - S_ii) + relu(m + max_{j != i} S_ji - S_ii)], the maxima by a
 * wavefront-64 arg-max.  Same workspace as pp_triplet_loss_fwd; the backward pass is pp_triplet_loss_bwd. */
int pp_triplet_loss_hardest_fwd(const float* V, const float* A, int N, int D, float margin, float* loss, void* ws,
                                size_t ws_bytes, pp_stream_t s);

/* pig/loss.py:51-55 cosine_matrix(U, V): out [Nu][Nv]; ws (Nu+Nv)*D floats */
int pp_cosine_matrix(const float* U, const float* V, int Nu, int Nv, int D, float* out, float* ws, pp_stream_t s);
/* pig/loss.py:41-48 contrastive(M, margin) forward on a given similarity matrix; ws N*N+3N floats */
int pp_contrastive_fwd(const float* S, int N, float margin, float* loss, float* ws, pp_stream_t s);
/* Backward passes of the two (the reference's are plain differentiable torch expressions, pig/loss.py:41-55):
 * dU [Nu][D], dV [Nv][D] from dS [Nu][Nv]; ws 2*(Nu+Nv)*D + Nu + Nv floats */
int pp_cosine_matrix_bwd(const float* U, const float* V, int Nu, int Nv, int D, const float* dS, float* dU, float* dV,
                         float* ws, pp_stream_t s);
/* dS [N][N] = dloss[0] * d contrastive / dS (dloss read from device memory); ws 3*N + 1 floats */
int pp_contrastive_bwd(const float* S, int N, float margin, const float* dloss, float* dS, float* ws, pp_stream_t s);
/* pig/metrics.py:45-52 triplet_accuracy: a, p, n fp32 [M][D] -> out [M] ((sign(diff)+1)/2 or diff) */
int pp_triplet_accuracy(const float* a, const float* p, const float* n, int M, int D, int discrete, float* out,
                        pp_stream_t s);
/* Rank-based recall of pig/metrics.py:7-42 (recall_at_n, recall_at_1_to_n) and :54-81 (resampled_*), batched on device.
 * S [Nr][ld] fp32 similarities (rows = references / queries, columns = candidates); positions follow the ascending order
 * of 1 - S, ties by index.  idx = NULL: one problem over the whole matrix; else nsets index sets [nsets][size] (int32),
 * set t ranks row idx[t][j] against columns idx[t][*].  correct = NULL: the target of row j is column j (torch.eye);
 * else [Nr][Nc] bytes, non-zero = target.  out [nsets][Nmax][rows] = recall@1..Nmax per row. */
int pp_recall_at_n(const float* S, int Nr, int Nc, int ld, const int* idx, int nsets, int size,
                   const unsigned char* correct, int Nmax, float* out, pp_stream_t s);

/* ---- pig/optimization.py:101-179 BertAdam.step (multi-tensor) --------------------------- */
typedef struct pp_tensor_list {
  int n_tensors;
  float* const* p;
  const float* const* g;
  float* const* m;
  float* const* v;
  const long long* numel;
} pp_tensor_list; /* the pointer arrays live in DEVICE memory */
/* chunk table (device): chunk c covers tensor chunk_tensor[c], elements [chunk_off[c], +chunk).
 * lr_per_tensor (device, [n_tensors], optional): scheduled learning rate of each tensor -- the reference keeps a step
 * count per tensor (pig/optimization.py:120-128), and tensors that skipped steps (LayerDrop) lag behind in the schedule;
 * with it every tensor of a parameter group goes into one launch.  NULL: lr_scheduled for all. */
int pp_bertadam_step(const pp_tensor_list* tl, const int* chunk_tensor, const long long* chunk_off,
                     int n_chunks, int chunk, float* norms /* [n_tensors + n_chunks] scratch (the chunk part: deterministic mode) */, float lr_scheduled,
                     float b1, float b2, float eps, float weight_decay, float max_grad_norm,
                     const float* lr_per_tensor,
                     const float* skip_flag /* optional (device): != 0 -> the whole step is a no-op (fp16 overflow) */,
                     pp_stream_t s);

/* ---- dynamic loss scaling for the fp16 build (what Lightning's native AMP does around `BertAdam.step` under
 *      `precision: 16`, /root/reference/hparams_base.yaml:45; torch.cuda.amp.GradScaler semantics) -------------------- */
/* g /= scale[0] (scale read from device memory) for every tensor of the list, in place; found_inf[0] = 1.0 if any element is inf / nan (else untouched:
 * zero it first).  = torch._amp_foreach_non_finite_check_and_unscale_.  Only tl->g and tl->numel are read. */
int pp_grad_unscale_check(const pp_tensor_list* tl, const int* chunk_tensor, const long long* chunk_off, int n_chunks,
                          int chunk, const float* scale, float* found_inf, pp_stream_t s);
/* scale / growth_tracker update after a step (= torch._amp_update_scale_): found_inf -> scale *= backoff, tracker = 0;
 * else tracker += 1 and, once it reaches growth_interval, scale *= growth (kept if the product overflows), tracker = 0. */
int pp_amp_update_scale(float* scale, int* growth_tracker, const float* found_inf, float growth_factor,
                        float backoff_factor, int growth_interval, pp_stream_t s);

#ifdef __cplusplus
}
#endif
#endif
