/* ldn_hip.h -- C ABI of libldn_hip.so, the MI355X (gfx950) implementation of LAUDNet's
 * dynamic-inference hot path (masker-driven masked bottleneck).
 *
 * The reference (LeapLabTHU/LAUDNet) has no FFI: its "operator API" is the Python nn.Module
 * surface.  Each entry point below therefore cites the reference Python it
 * replaces (paths relative to imagenet_classification/); the laudnet_amd python modules bind them with
 * ctypes (INTEGRATION.md shows the stub a maintainer of the reference would add).
 *
 * Conventions
 *   - every function returns 0 on success, a negative LDN_E* code otherwise;
 *     ldn_last_error() returns a thread-local message for the last failure on this thread;
 *   - all data pointers are DEVICE pointers owned by the caller; nothing is allocated or freed;
 *   - `stream` is a hipStream_t (pass NULL for the default stream); no entry point synchronises;
 *   - activations are NHWC fp32 ("channels-last": one pixel = one contiguous row of C floats);
 *   - index/count tensors are int32; masks are fp32 {0,1} like the reference's;
 *   - data-dependent sizes (active-row counts, per-image channel counts) stay on the device:
 *     grids are sized for the worst case and surplus workgroups exit on the device-side count.
 */
#ifndef LDN_HIP_H
#define LDN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LDN_OK 0
#define LDN_EINVAL (-1)   /* bad argument (shape/alignment/unsupported combination) */
#define LDN_EHIP (-2)     /* HIP runtime error at launch */

/* PER-THREAD STATE.  The entry points take everything they compute from their arguments; what the library keeps between calls is exactly this,
 * all of it thread-local (two threads never see each other's):
 *   - the text behind ldn_last_error();
 *   - ldn_hint_rows(n): an ADVISORY row count consumed by the thread's next ldn_conv_rows_split / ldn_conv_rows_pool call (tile shape only;
 *     results are bit-identical with any hint or none) -- ldn_conv3x3_rows_ps takes the same hint as an argument;
 *   - ldn_plan_work_zeroed(1): the caller's PROMISE that the `work` buffer of the thread's next list-build call is zero, consumed by that call
 *     (which hands the buffer back zeroed whatever path it takes).
 * Process-wide and read-only after the first use: the environment switches named in this header and the default arithmetic mode.  Device-side
 * counters (ldn_debug_violations, ldn_plan_timeouts) and the fault word (ldn_fault_flag) are diagnostics, not inputs of any computation. */
const char* ldn_last_error(void);
int ldn_version(void);
/* Index-bounds audit ("race detection / sanitizers").  In the LDN_DEBUG build of the library (python -m
 * laudnet_amd.build --debug -> libldn_hip_debug.so) every kernel checks the index lists it consumes or produces on the device
 * (entries inside their tensor, counts inside the list capacity, channel pairs aligned and ascending, every in-bounds 3x3 tap
 * of an active pixel present in the dilated list); violations are counted, never trapped.  *count = violations since the last
 * reset (synchronises the device), *first_code = code of the first one (1xx conv, 2xx index build, 3xx fused tail, 4xx RegNet, 5xx packed-row 1x1);
 * the release build reports *count = -1 (checks compiled away).  Debug-only tooling: not on the hot path. */
int ldn_debug_violations(int* count, int* first_code, int reset);
/* Robustness counter of the one-launch list build (ldn_mask_plan, and ldn_mask_to_index on maps it runs as one launch): its
 * workgroups wait, bounded in time (2 s; LDN_PLAN_TIMEOUT_MS), for the counts of the images in front of them.  A wait that runs into
 * the bound leaves EMPTY lists -- counts, every image prefix and the statistics are zero, whatever the interleaving of the
 * workgroups -- never uninitialised rows, and is counted here: *count = such events since the last reset (0 on a healthy device).
 * Synchronises the device: call it at a point where the caller synchronises anyway (end of a batch, health check). */
int ldn_plan_timeouts(int* count, int reset);
/* The LOUD side of the same events (and of the hand-off waits of the chained stage kernel, which are bounded the same way and counted in
 * ldn_plan_timeouts as well): one int of pinned host memory per process that a kernel sets to 1 the moment one of its bounded waits fails.  The
 * caller reads it WITHOUT synchronising -- laudnet_amd._lib.check() does after every library call and raises LdnError -- so a forward that ran
 * on empty lists cannot hand its logits on silently.  Sticky until ldn_plan_timeouts(.., reset = 1).  NULL when no pinned memory could be had. */
const int* ldn_fault_flag(void);
/* number of compute units of the current device (used by callers to size persistent grids) */
int ldn_device_cus(int* cus);
/* Arithmetic of the MFMA convolutions -- the `math_mode` ARGUMENT of ldn_conv_image / ldn_conv_packed / ldn_conv_rows /
 * ldn_bottleneck_tail (the library keeps no mutable state: two threads or streams may use different modes concurrently):
 *   LDN_MATH_FP32   (0) = fp32 operands on v_mfma_f32_32x32x2_f32 (fp32 multiply, fp32 accumulate);
 *   LDN_MATH_BF16X3 (1) = "bf16x3" split precision: each fp32 operand x is split into bf16 hi + bf16 lo
 *       (round-to-nearest-even both), the product is hi*hi + lo*hi + hi*lo on v_mfma_f32_32x32x16_bf16 with
 *       fp32 accumulation; per-product relative error <= ~2^-16 (measured: 5e-6 relative on a K=2304 conv),
 *       inside the north star's 1e-3 fp32 parity tolerance.  Storage stays fp32 everywhere.
 *   LDN_MATH_DEFAULT (-1) = the read-only process default, taken once from the environment variable LDN_MATH_MODE
 *       (0 if unset); ldn_default_math_mode() returns it.
 * The reference has no counterpart (cuDNN chooses its own algorithms, TF32 included, behind
 * torch.backends.cudnn.allow_tf32). */
#define LDN_MATH_DEFAULT (-1)
#define LDN_MATH_FP32 0
#define LDN_MATH_BF16X3 1
int ldn_default_math_mode(void);

/* ---- a1: Masker_spatial.forward, eval branch (models/utils.py:47-65) ------------------------
 * x [B,Hi,Wi,C] NHWC -> adaptive average pool to SxS (only if S < Hi, utils.py:48; bins
 * floor(i*H/S)..ceil((i+1)*H/S)) -> 1x1 conv C->2g (+bias) -> mask[b,j,y,x] = (l[j] >= l[g+j]).
 * logits [B,2g,S,S] may be NULL.  mask [B,g,S,S] fp32 {0,1}.  work: float scratch of
 * B * ldn_channel_masker_splits(Hi*Wi) * C entries, only needed for S == 1 (layer skip: whole-image window); after the call it
 * holds the images' partial channel sums.  carry_prefix (optional, S == 1 only) [B+1]: the kept-row prefix (img_prefix3 of
 * ldn_mask_to_index) of the PREVIOUS layer-skip block on the same residual stream, with `work` the buffer that block's call
 * filled: an image that block skipped (empty range) is unchanged, so its sums are reused instead of re-read.
 * Patch masks (1 < S < Hi): `work` (optional, B*S*S*C floats) receives every patch's pooled channel means; carry_mask (optional)
 * [B][S][S] {0,1} = the patch mask the PREVIOUS block on the same residual stream executed (the union of its mask groups), with
 * `work` the buffer the previous call filled: a patch that block did not touch keeps its means, its window is not re-read. */
int ldn_spatial_masker(const float* x, int B, int Hi, int Wi, int C, const float* w /*[2g,C]*/,
                       const float* bias /*[2g]*/, int g, int S, float* mask, float* logits, float* work,
                       const int32_t* carry_prefix, const float* carry_mask, void* stream);
/* bytes of `work` the call above needs for this shape (0 = none); every *_workspace_bytes twin below follows the same rule */
size_t ldn_spatial_masker_workspace_bytes(int B, int Hi, int Wi, int C, int S);

/* ---- a4/a11: F.interpolate(nearest) + ExpandMask x2 -> packed index lists -------------------
 * (laud_resnet.py:106-110, models/utils.py:74-89).  patch_mask [B,Sy,Sx] fp32 {0,1} (one mask
 * group) is up-sampled (nearest: src = min(floor(i*float(Sy)/float(Ho)), Sy-1), same for x) to the output
 * resolution Ho x Wo (= mask3 = mask2) and dilated to the block's input resolution
 * Hi x Wi = Ho*stride x Wo*stride with a 3x3 box after zero-insertion (= mask1).
 * Outputs (all row-major over b,y,x; "row" = flat pixel number b*H*W + y*W + x):
 *   idx3 [B*Ho*Wo]   rows of set mask3 pixels (first cnt[0] entries valid)  == torch.nonzero
 *   pos3 [B*Ho*Wo]   rank of the pixel in idx3, -1 if unset
 *   idx1 [B*Hi*Wi]   rows of set mask1 pixels (first cnt[1] entries valid)
 *   pos1 [B*Hi*Wi]   rank in idx1 or -1
 *   nbr  [B*Ho*Wo*9] for packed output row r, tap t=ky*3+kx: rank in idx1 of input pixel
 *                    (oy*stride-1+ky, ox*stride-1+kx), -1 if out of bounds
 *   cnt  [2]         {#mask3 rows, #mask1 rows}
 *   img_prefix3/1 [B+1]  exclusive per-image prefix of the two lists
 *   stats [3]        {mean(patch_mask), mean(mask2 pixels), mean(mask1 pixels)}  (the three
 *                    spatial sparsities of laud_resnet.py:98,108,110)
 * work: int32 scratch of ldn_mask_to_index_workspace_bytes (3 entries per image; per (image, band of rows) for maps whose
 * per-image tables do not fit one workgroup's LDS -- those are built by bands of output rows, same lists). */
int ldn_mask_to_index(const float* patch_mask, int B, int Sy, int Sx, int Ho, int Wo, int stride, int32_t* idx3,
                      int32_t* pos3, int32_t* idx1, int32_t* pos1, int32_t* nbr, int32_t* cnt,
                      int32_t* img_prefix3, int32_t* img_prefix1, float* stats, int32_t* work, void* stream);
size_t ldn_mask_to_index_workspace_bytes(int B, int Ho, int Wo, int stride);
/* The one-launch build (ldn_mask_plan, and ldn_mask_to_index where ldn_mask_plan_fits) keeps its flag words in `work`: they must be zero when the
 * launch starts, so a zeroing launch runs in front of it -- unless the caller vouches for the buffer: ldn_plan_work_zeroed(1) says that the `work`
 * of the NEXT such call of this thread is all zero.  Every one-launch build LEAVES its flag words zero (the last workgroup out clears them), so a
 * buffer zeroed once and only ever handed to one-launch builds on ONE stream stays valid: 1 launch per list build instead of 2.  The word is
 * consumed by the next ldn_mask_plan / ldn_mask_to_index call whatever path it takes; never give it for a buffer the banded / two-launch build
 * (maps beyond ldn_mask_plan_fits, LDN_INDEX_PLAN=0) has used. */
int ldn_plan_work_zeroed(int yes);
/* Cell means on a 2S x 2S grid (pool [B][2S][2S][C], as ldn_conv_rows_pool / ldn_spatial_masker leave them) -> the means of the S x S grid of
 * 2 x 2 groups of cells, coarse [B][S][S][C] = 0.25 * ((f00 + f01) + (f10 + f11)): for the head of a stage whose masker pools coarser cells
 * than its predecessor wrote (models/utils.py:47-52 with a halved mask_size on the same map), in front of ldn_mask_plan's decide mode. */
int ldn_coarsen_cell_means(const float* fine, int B, int S, int C, float* coarse, void* stream);

/* ---- a1 + a4 in ONE launch (round 4): the fused spatial masker of a run of identity blocks (models/utils.py:47-65 +
 * laud_resnet.py:96-110).  The lists, counts, prefixes and statistics of ldn_mask_to_index, from
 *   patch_mask [B][S][Sx] {0,1}                                 (pool / w / bias / mask_out / logits unused), or
 *   patch_mask == NULL: the pooled channel means pool [B][S*Sx][C] of the block's input (the `work` buffer of
 *     ldn_spatial_masker, whose touched patches the previous block's ldn_conv_rows_pool refreshed) and the masker's 1x1 conv
 *     w [2][C], bias [2] (one mask group): mask_out [B][1][S][Sx] = (l0 >= l1) exactly as ldn_spatial_masker decides from the
 *     same means (same arithmetic order), logits [B][2][S][Sx] optional -- x is not read.
 * patch_major = 1 (even grids: Ho % S == 0, Wo % Sx == 0): idx3 lists the kept pixels PATCH BY PATCH (row-major inside a patch,
 * patches in row-major order) instead of row-major over the image -- the same set of rows, counts and prefixes; pos3 / nbr follow
 * the order -- so that Ho/S * Wo/Sx consecutive packed rows are one patch (what ldn_conv_rows_pool needs).
 * Images are processed one workgroup each; the prefix over the images is taken inside the launch (every workgroup publishes its
 * counts and waits, bounded, for those of the workgroups dispatched in front of it);
 * ldn_mask_plan_fits says whether the per-image tables fit one workgroup's LDS (else: ldn_mask_to_index, which builds by bands).
 * work: int32 scratch of ldn_mask_to_index_workspace_bytes(B, Ho, Wo, stride). */
/* Layer skip (mask_size 1, laud_resnet.py:91-94 / dyn_mode 'layer') on the same fusion: ldn_layer_index = the lists of
 * ldn_mask_to_index for one decision per image (image_mask [B]), the kept images' pixels TILE BY TILE (tile_gy x tile_gx pixels
 * each; 0, 0 = row-major) so that ldn_conv_rows_pool leaves every kept image's tile means; ldn_layer_head = the decision from those
 * means (pool [B][nparts][C], nparts equal tiles per image: the global average pool is the mean of the tile means) -- x is not read. */
int ldn_layer_index(const float* image_mask, int B, int Ho, int Wo, int stride, int tile_gy, int tile_gx, int32_t* idx3,
                    int32_t* pos3, int32_t* idx1, int32_t* pos1, int32_t* nbr, int32_t* cnt, int32_t* img_prefix3,
                    int32_t* img_prefix1, float* stats, void* stream);
int ldn_layer_head(const float* pool, int B, int nparts, int C, const float* w /*[2g,C]*/, const float* bias /*[2g]*/, int g,
                   float* mask /*[B,g]*/, float* logits /*[B,2g] or NULL*/, void* stream);
int ldn_mask_plan_fits(int S, int Sx, int Ho, int Wo, int stride);
int ldn_mask_plan(const float* patch_mask, const float* pool, int C, const float* w, const float* bias, float* mask_out,
                  float* logits, int B, int S, int Sx, int Ho, int Wo, int stride, int patch_major, int32_t* idx3,
                  int32_t* pos3, int32_t* idx1, int32_t* pos1, int32_t* nbr, int32_t* cnt, int32_t* img_prefix3,
                  int32_t* img_prefix1, float* stats, int32_t* work, void* stream);

/* ---- K2/K5: stand-alone row gather / masked scatter-add (DyNetSimulator simulate_gather,
 * simulate_scatter_add; laud_resnet.py:133,143-144) ----------------------------------------- */
/* packed[r,:] = src[rows[r],:] for r < *count (count==NULL -> cap) ; C % 4 == 0 */
int ldn_gather_rows(const float* src, int ld_src, const int32_t* rows, const int32_t* count, int cap, int C,
                    float* packed, int ld_packed, void* stream);
/* out[rows[r],:] = relu(identity[rows[r],:] + packed[r,:]) ; out may alias identity */
int ldn_scatter_add_relu(const float* packed, int ld_packed, const int32_t* rows, const int32_t* count, int cap,
                         int C, const float* identity, int ld_id, float* out, int ld_out, void* stream);

/* ---- a7 (spatial / layer mode): convolution over PACKED ROWS with shared weights ------------
 * (laud_resnet.py:115-144 restricted to the active pixels; conv+BN(+ReLU) fused, fp32 MFMA)
 *   acc[m,o] = sum_{t<taps} sum_{c<cin} A[a_rows[m*taps+t], c] * w[o, t, c]     (a_rows<0 -> 0)
 *   v        = scale[o]*acc + shift[o]
 *   dst row  = out_rows ? out_rows[m] : m
 *   if residual: v += residual[dst row, o]
 *   relu     : relu==1 always; relu==2 only where relu_if_neg[m] < 0; relu==0 never
 *   out[dst row, o] = v                                   for m < *m_count (NULL -> m_cap)
 * a_rows==NULL means A row m is row m (taps must be 1).  cin % 4 == 0, lda/ldo/ldr % 4 == 0. */
int ldn_conv_rows(const float* a, int lda, const int32_t* a_rows, int taps, const int32_t* m_count, int m_cap,
                  const float* w, int cin, int cout, const float* scale, const float* shift, int relu,
                  const int32_t* relu_if_neg, const int32_t* out_rows, const float* residual, int ldr,
                  float* out, int ldo, int math_mode, void* stream);

/* The same contract on the round-2 kernel k_dense (bf16x3 arithmetic): the weights come PRE-SPLIT, n-major --
 * w_split [cout][taps*cin/8][32 B], row n, octet o = bf16 {8 hi | 8 lo} of w[n, 8o .. 8o+7] over the K axis (tap, channel)
 * (one-time module preparation) -- so that a weight fragment needs no conversion in the K loop; cin % 8 == 0, cout % 4 == 0
 * (widths that are not multiples of 32 -- LAD-RegNet's 144 / 784 -- run with a zero-filled K tail and a ragged last column tile);
 * taps == 9: a_rows is the [rows][9] neighbour table of ldn_mask_to_index.  shift_classes 16 + pix_map + geometry: the
 * border-class shift table of the channel algebra, as in ldn_conv_packed.  Two optional epilogue terms of the dense
 * execution of channel mode: post_sub [cout] is subtracted after the ReLU, chan_mask [B][cout] {0,1} multiplies
 * row r by the mask of image floor(dst(r) / rows_per_image) (apply_channel_mask, models/utils.py:18-25, fused).
 * relu == 3 (this entry point only): exact GELU 0.5 v (1 + erf(v / sqrt 2)) instead of the ReLU -- the fc1 -> GELU of a
 * token-skipping transformer block (DyNetSimulator/adavit/simulate_adavit.py:136-150) without a pass over the hidden rows.
 * ln_stats / ln_c1 (optional, taps == 1): the LayerNorm in front of a token-skip linear (simulate_adavit.py:90-93,136-140) as an
 * epilogue term -- LN(x) . w + b = rstd (x . w' - mean c1) + (w . beta + b) with w' = gamma * w (what w_split then holds),
 * c1[n] = sum_k w'[n][k] (ln_c1 [cout]), {mean, rstd} of every SOURCE row in ln_stats [rows of a][2] (ldn_row_stats), and the
 * constant in `shift`: the normalised rows never exist in memory. */
int ldn_conv_rows_split(const float* a, int lda, const int32_t* a_rows, int taps, const int32_t* m_count, int m_cap,
                        const void* w_split, int cin, int cout, const float* scale, const float* shift, int relu,
                        const int32_t* relu_if_neg, const int32_t* out_rows, const float* residual, int ldr, float* out,
                        int ldo, const float* post_sub, const float* chan_mask, int rows_per_image, int shift_classes,
                        const int32_t* pix_map, int Hi, int Wi, int Ho, int Wo, int stride, const float* ln_stats,
                        const float* ln_c1, void* stream);
/* The same kernel in TRUE fp32 arithmetic (v_mfma_f32_32x32x2_f32: fp32 multiply and accumulate, the `fp32` math mode): w is the plain
 * row-major fp32 weight matrix [cout][taps * cin] (tap-major rows for taps == 9) -- no pre-split copy; every other argument as in
 * ldn_conv_rows_split.  Five times the matrix time of the bf16x3 form per product, the same staging pipeline and epilogue. */
int ldn_conv_rows_f32(const float* a, int lda, const int32_t* a_rows, int taps, const int32_t* m_count, int m_cap,
                        const float* w, int cin, int cout, const float* scale, const float* shift, int relu,
                        const int32_t* relu_if_neg, const int32_t* out_rows, const float* residual, int ldr, float* out,
                        int ldo, const float* post_sub, const float* chan_mask, int rows_per_image, int shift_classes,
                        const int32_t* pix_map, int Hi, int Wi, int Ho, int Wo, int stride, const float* ln_stats,
                        const float* ln_c1, void* stream);

/* Advisory, per thread, consumed by the next ldn_conv_rows_split / ldn_conv_rows_pool call: about how many rows that call will find in
 * its device-side count *m_count (the host cannot read it without a synchronisation; a caller passes e.g. the previous forward's
 * count, copied to pinned memory asynchronously).  It only selects the tile width (a launch takes rounds-of-workgroups x tile time,
 * DESIGN.md 4n / 4t); results are bit-identical with any hint or none.  Calls without a device-side count know their rows exactly. */
int ldn_hint_rows(int rows);   /* returns 0 */

/* ldn_conv_rows_split / ldn_conv_rows_f32 with taps == 1, plus a by-product: pool [B][S*Sx][cout] receives, for every patch
 * this launch writes, the MEAN of the final output (after residual and ReLU) over the patch's Ho/S x Wo/Sx pixels (4 or 16) --
 * the pooled means the next block's spatial masker needs (adaptive_avg_pool2d of models/utils.py:48-52 on an even grid), so
 * that masker never reads x (ldn_mask_plan, patch_mask == NULL).  The packed rows must list whole patches, Ho/S * Wo/Sx
 * consecutive rows each (ldn_mask_plan with patch_major = 1; the sums are a fixed-order register tree: deterministic);
 * out_rows = flat pixel numbers b*Ho*Wo + y*Wo + x.  cout % 128 == 0.  math_mode LDN_MATH_FP32: w = plain [cout][cin] floats;
 * LDN_MATH_BF16X3: w = the pre-split rows of ldn_conv_rows_split. */
int ldn_conv_rows_pool(const float* a, int lda, const int32_t* a_rows, const int32_t* m_count, int m_cap, const void* w,
                       int cin, int cout, const float* scale, const float* shift, int relu, const int32_t* relu_if_neg,
                       const int32_t* out_rows, const float* residual, int ldr, float* out, int ldo, float* pool, int S,
                       int Sx, int Ho, int Wo, int math_mode, void* stream);

/* Round 5: the packed spatial / layer path keeps its intermediates PRE-SPLIT between its three launches, so that no operand is split
 * into bf16 hi / lo inside a K loop (models/laud_resnet.py:115-144 on the active pixels; the operator list the reference's simulator
 * times: conv1 on the dilated list, gather_conv2, conv3 + scatter_add, DyNetSimulator/eval_example.py:39-48).  A pre-split row holds, per
 * octet of channels, 8 bf16 hi then 8 bf16 lo (x = hi + lo + O(2^-17 |x|), both round-to-nearest-even) -- the layout of the pre-split
 * weights (ldn_conv_rows_split: `w_split`); an element still takes 4 bytes, strides (`lda`, `ldo`) still count 4-byte elements.
 *
 * ldn_conv_rows_ps: ldn_conv_rows_split with taps = 1 where `a` (a_presplit != 0) and / or `out` (out_presplit != 0) are pre-split rows.
 *   A pre-split output takes no residual, scatter (out_rows) or pooled means.  pool != NULL: as ldn_conv_rows_pool (S, Sx, Ho, Wo).
 *   bf16x3 arithmetic; cin % 32 == 0, cout % 64 == 0; results are bit-identical to ldn_conv_rows_split on the same values.
 * ldn_conv3x3_rows_ps: out[m, :] = act(scale * sum_{tap, k} a[nbr[m][tap], k] * w[:, tap, k] + shift) over the first min(*m_count, m_cap)
 *   packed rows (m_count NULL: m_cap): `a_presplit` pre-split rows, nbr [m_cap][9] (row of `a` per tap, -1 = zero), w_split the pre-split
 *   [cout][9][cin] weights, relu 0 | 1, out fp32 or pre-split rows (out_presplit).  rows_hint >= 0: about how many rows *m_count will hold
 *   (tile shapes only; -1 = unknown).  cin % 64 == 0, cin <= 2048 (the zero row a missing neighbour is read from), cout % 64 == 0.  Bit-identical to
 *   ldn_conv_rows_split(taps = 9) on the same values. */
int ldn_conv_rows_ps(const float* a, int lda, int a_presplit, const int32_t* a_rows, const int32_t* m_count, int m_cap,
                     const void* w_split, int cin, int cout, const float* scale, const float* shift, int relu,
                     const int32_t* relu_if_neg, const int32_t* out_rows, const float* residual, int ldr, float* out, int ldo,
                     int out_presplit, float* pool, int S, int Sx, int Ho, int Wo, void* stream);
int ldn_conv3x3_rows_ps(const void* a_presplit, int lda, const int32_t* nbr, const int32_t* m_count, int m_cap, const void* w_split,
                        int cin, int cout, const float* scale, const float* shift, int relu, void* out, int ldo, int out_presplit,
                        int rows_hint, void* stream);
/* stats[r] = {mean, 1 / sqrt(biased variance + eps)} of row r of x [rows][ld >= C] (nn.LayerNorm's statistics), C % 4 == 0, C <= 2048 */
int ldn_row_stats(const float* x, int ld, int rows, int C, float eps, float* stats, void* stream);
/* the same for the rows list[0 .. *count) only (stats of the other rows are left untouched): the LayerNorm statistics of the tokens a
 * token-skip block works on -- the rows of the others are never read */
int ldn_row_stats_list(const float* x, int ld, int rows, int C, float eps, const int32_t* list, const int32_t* count, float* stats,
                       void* stream);

/* ---- a2: Masker_channel_MLP.forward, eval branch (models/utils.py:113-131) + index build ----
 * x [B,HW,C] NHWC -> global average pool -> Linear(C,hidden)+ReLU+Linear(hidden,2G) (hidden>0)
 * or Linear(C,2G) (hidden==0: w1=[2G,C], b1=[2G], w2/b2 unused) -> mask[b,j] = l[j] >= l[G+j].
 * Also emits the per-image ACTIVE CHANNEL LIST over `width` = G*gran channels (group j owns
 * channels [j*gran,(j+1)*gran), models/utils.py:18-25): ch_idx [B,width] left-packed ascending,
 * ch_cnt [B].  logits [B,2G] may be NULL.  If mask_in != NULL the masker arithmetic is skipped
 * and the lists are built from mask_in ("identical masks" parity runs).
 * work: float scratch of B*splits*C entries, splits = ldn_channel_masker_splits(HW).
 * Fused GAP: if gap_partial != NULL it holds [B][gap_splits][C] partial channel sums of x already produced by
 * ldn_conv_image(..., colsum) of the previous block; x and work are then unused (no extra pass over x). */
int ldn_channel_masker_splits(int HW);
int ldn_channel_masker(const float* x, int B, int HW, int C, const float* w1, const float* b1, const float* w2,
                       const float* b2, int hidden, int G, int gran, const float* mask_in, float* mask,
                       float* logits, int32_t* ch_idx, int32_t* ch_cnt, float* work, const float* gap_partial,
                       int gap_splits, void* stream);
size_t ldn_channel_masker_workspace_bytes(int B, int HW, int C);

/* ---- a7/a8: the FLOPs bookkeeping of a whole forward in one launch (laud_resnet.py:112-147 per block -- sparse_flops, dense_flops,
 * flops_perc -- and :321-356 for the static stem / pool / classifier terms).  For block j (n_blocks of them):
 *   (s3, s2, s1, cs) = st_in[j][0..st_cols-1] if st_in != NULL (st_cols 3: cs = 1), else (1, 1, 1, 1);
 *   cs = sum_b cnt[j][b] / denom[j] where denom[j] > 0   (channel-mode blocks: cnt = ch_cnt [n_blocks][B], denom = B * width);
 *   sparse = t0 + t1 cs s1 + t2 cs^2 s2 + t3 cs s3 + t4 with terms[j] = (masker, conv1, conv2, conv3, downsample) FLOPs (fp64);
 *   perc[j] = sparse / (t0 + .. + t4);  flops[0] = sum_j sparse + static_flops;  st_out[j] = (s3, s2, s1, cs). */
int ldn_forward_stats(const int32_t* cnt, int B, const float* denom, const float* st_in, int st_cols, const double* terms,
                      double static_flops, int n_blocks, float* st_out, float* perc, float* flops, void* stream);

/* ---- a7 (channel mode): per-image channel-subset convolution --------------------------------
 * (laud_resnet.py:115-144 with apply_channel_mask, models/utils.py:18-25; also the plain dense
 * NHWC 1x1/3x3 conv when k_idx==n_idx==NULL, used for the downsample branch :138-141)
 * For image b, output pixel p=(oy,ox), packed output column j (channel o = n_idx? n_idx[b,j] : j):
 *   acc = sum_{tap} sum_{i<Kb} A[b, pix(p,tap), i] * w[o, tap, (k_idx? k_idx[b,i] : i)]
 *         (Kb = k_cnt[b] or cin; 3x3 taps use pad 1, out-of-bounds taps contribute 0)
 *   v   = scale[o]*acc + shift[cls(p)*cout + o]      cls = 0 if shift_classes==1, else the 4x4
 *         border class (top|bottom<<1)*4 + (left|right<<1) of the taps that fall outside
 *   if residual: v += residual[b,p,j] ; if relu: v = max(v,0) ; if post_sub: v -= post_sub[o]
 *   out[b,p,j] = v for j < Nb (= n_cnt[b] or cout); columns Nb..roundup4(Nb)-1 are written 0.
 *   scale == NULL (here, in ldn_conv_packed and in ldn_conv_rows) means scale[o] == 1: the caller has multiplied the
 *   BN scale into w.  With a residual the bf16x3 kernel then starts its accumulators from the residual tile, so its
 *   epilogue only stores (same result: v = acc + shift + residual).
 *   colsum (optional, dense output only) [B][ceil(Ho*Wo/32)][cout]: sum of out over each run of 32 pixels --
 *   the global-average-pool partials the next block's channel masker needs (a2), fused into this epilogue.
 * A columns are "left-packed": column i of image b is channel k_idx[b,i].
 * Weight layout: without k_idx, w is n-major  [cout][ksize*ksize][cin]  (rows gathered through n_idx);
 *                with k_idx,    w is k-major  [ksize*ksize][cin][cout]  (rows gathered through k_idx, columns
 *                through n_idx) -- every fetch then stays inside one row of the weight matrix.
 * kgran: the channel granularity of the index lists (channel_dyn_granularity): every run of `kgran` consecutive
 * list entries is a run of consecutive channels; must divide every count.
 * out_format: 0 = fp32 rows.  1 = rows PRE-SPLIT for ldn_bottleneck_tail (bf16x3 arithmetic only, no residual / colsum,
 *   ldo % 32 == 0): every octet of 8 packed output columns is stored as [8 hi bf16 | 8 lo bf16] (hi = bf16(v), lo =
 *   bf16(v - hi), the same 4 bytes per element), columns Nb .. roundup32(Nb)-1 zero-filled. */
int ldn_conv_image(const float* a, int lda, int B, int Hi, int Wi, int ksize, int stride, int Ho, int Wo,
                   const float* w, int cin, int cout, const int32_t* k_idx, const int32_t* k_cnt, int kgran,
                   const int32_t* n_idx, const int32_t* n_cnt, const float* scale, const float* shift,
                   int shift_classes, const float* post_sub, int relu, const float* residual, int ldr,
                   float* out, int ldo, float* colsum, int out_format, int math_mode, void* stream);

/* ---- a7 (channel mode, bf16x3): FUSED TAIL of the bottleneck -- conv2 (3x3) -> bn2 + ReLU -> conv3 (1x1) -> bn3 + residual
 * + ReLU in one launch (laud_resnet.py:123-144 on the active channels; stride 1 or 2 (the 3x3 of a stage's first block, :123 with
 * stride 2: H x Wd is conv2's INPUT map, the output map is Ho x Wo = ((H-1)/stride+1) x ((Wd-1)/stride+1), the residual then is the
 * projection shortcut's output); channel_dyn_granularity % 2 == 0; width in {64, 128, 256}; output map width <= 256).  h2 never
 * exists in memory and the 3x3 reads each K slice of h1 once for all nine taps.  Per image b with channel list
 * ch_idx[b, 0 .. Kb-1] (ascending, aligned pairs), Kb = ch_cnt[b]:
 *   h1_split [B*H*Wd][ldh]   conv1's output in ldn_conv_image's out_format 1 (left-packed columns = list positions)
 *   w2_pairs [9][width/2][width/2][16 B]  piece (tap, kp, np) = for n in {2np, 2np+1}: bf16 {hi(w[n][2kp]), hi(w[n][2kp+1]),
 *            lo(w[n][2kp]), lo(w[n][2kp+1])} with w[n][k] = conv2.weight[n, k, tap]            (k = input, n = output channel)
 *   w3_pairs [width/2][cout][8 B]  entry (kp, c) = bf16 {hi(w3[c][2kp]), hi(w3[c][2kp+1]), lo(..), lo(..)}, w3 = bn3.scale * conv3.weight
 *   u   = relu(scale2[n] * conv2 + shift2_tab[class(pixel)][n]) - post_sub2[n]      (16 border classes as in ldn_conv_image)
 *   out = relu(w3 . u + shift3 + residual)          out/residual [B*Ho*Wo][ldo/ldr] fp32, may alias (in-place residual stream)
 *   colsum (optional) [B][ldn_bottleneck_tail_splits(H, Wd, width, stride)][cout]: partial sums of out over disjoint pixel sets covering
 *   the image (the next block's channel masker takes them as gap_partial). */
/* conv1 of the same block in the same style (k_head): h1 = relu(scale1 * conv1x1(x)[active channels] + shift1) - post_sub1, written
 * in out_format 1 for ldn_bottleneck_tail (laud_resnet.py:115-118).  x [B*HW][ldx] fp32, cin % 32 == 0;
 * w1_split [width][cin/8][32 B]: row n, octet o = bf16 {8 hi | 8 lo} of conv1.weight[n, 8o .. 8o+7] (n-major: the per-image
 * OUTPUT-channel gather is a row gather). */
int ldn_bottleneck_head(const float* x, int ldx, int B, int HW, int cin, const void* w1_split, int width,
                        const int32_t* ch_idx, const int32_t* ch_cnt, const float* scale1, const float* shift1,
                        const float* post_sub1, void* h1_split, int ldh, void* stream);
/* The same three kernels in TRUE fp32 arithmetic (v_mfma_f32_32x32x2_f32: fp32 multiply and accumulate, the library's `fp32` math mode).
 * An element takes 4 bytes as a bf16 hi + lo pair and as a float, so every pre-split layout above read as plain floats IS its fp32 twin:
 *   w1 [width][cin] fp32 row-major;  h1 [B*HW][ldh] plain fp32 rows (left-packed columns, zero-filled to a multiple of 32);
 *   w2_pairs_f32 [9][width/2][width/2][2 n][w(n, 2kp), w(n, 2kp + 1)] fp32;  w3_pairs_f32 [width/2][cout][w3(c, 2kp), w3(c, 2kp + 1)] fp32.
 * Same staging pipelines, eight 32x32x2 products per K16 step instead of three 32x32x16 ones; every other argument as in the bf16x3 forms
 * (ldn_bottleneck_chain_f32: the blocks' w1s / w2p / w3p fields point at the fp32 twins). */
int ldn_bottleneck_head_f32(const float* x, int ldx, int B, int HW, int cin, const float* w1, int width,
                            const int32_t* ch_idx, const int32_t* ch_cnt, const float* scale1, const float* shift1,
                            const float* post_sub1, float* h1, int ldh, void* stream);
int ldn_bottleneck_tail_f32(const void* h1, int ldh, int B, int H, int Wd, int stride, int width, const void* w2_pairs_f32,
                            const void* w3_pairs_f32, int cout, const int32_t* ch_idx, const int32_t* ch_cnt, const float* scale2,
                            const float* shift2_tab, const float* post_sub2, const float* shift3, const float* residual,
                            int ldr, float* out, int ldo, float* colsum, void* stream);
int ldn_bottleneck_tail_splits(int H, int Wd, int width, int stride);   /* H x Wd = conv2's INPUT map; 0 = this map / width does not fit the fused tail */
int ldn_bottleneck_tail(const void* h1_split, int ldh, int B, int H, int Wd, int stride, int width, const void* w2_pairs,
                        const void* w3_pairs, int cout, const int32_t* ch_idx, const int32_t* ch_cnt, const float* scale2,
                        const float* shift2_tab, const float* post_sub2, const float* shift3, const float* residual,
                        int ldr, float* out, int ldo, float* colsum, void* stream);
/* The same tail with the block's PROJECTION SHORTCUT folded into conv3 (laud_resnet.py:138-141 with a stride-1 1x1 downsample: the first
 * block of stage 1, whose input is the stem's 64 channels): out = relu(w3 . u + wd . x + shift3d), shift3d = shift3 + bn_d's shift --
 * 64 more K values of the same GEMM instead of a projection launch that writes the identity tensor and a tail that reads it back.
 *   x_split  the block INPUT pre-split to bf16 hi / lo, written by ldn_bottleneck_head_split beside h1 (conv1 splits x anyway), in tiles
 *            of 32 consecutive pixels of the flat [B*HW] batch: [tile][cin/16 K16 steps][2 octets of the step][hi | lo][32 pixels][8 bf16]
 *            (a consumer wave's fragment load -- lane = pixel -- is then two contiguous 512-byte runs); ldn_x_split_bytes(B*HW, cin) bytes;
 *   wd_pairs [cin/2][cout][8 B] = w3_pairs' layout of bn_d.scale * downsample.weight;
 *   stride 1, width 64, cin 64 (ldn_bottleneck_tail_proj_fits says whether a map qualifies). */
size_t ldn_x_split_bytes(size_t pixels, int cin);
int ldn_bottleneck_head_split(const float* x, int ldx, int B, int HW, int cin, const void* w1_split, int width,
                              const int32_t* ch_idx, const int32_t* ch_cnt, const float* scale1, const float* shift1,
                              const float* post_sub1, void* h1_split, int ldh, void* x_split, void* stream);
int ldn_bottleneck_tail_proj_fits(int H, int Wd, int width, int cin);
int ldn_bottleneck_tail_proj(const void* h1_split, int ldh, int B, int H, int Wd, int width, const void* w2_pairs,
                             const void* w3_pairs, int cout, const int32_t* ch_idx, const int32_t* ch_cnt, const float* scale2,
                             const float* shift2_tab, const float* post_sub2, const float* shift3d, const void* x_split,
                             int cin, const void* wd_pairs, float* out, int ldo, float* colsum, void* stream);

/* ---- a7 (channel mode, bf16x3): a WHOLE stride-1 identity-shortcut bottleneck on a SMALL map (H * Wd <= 64 pixels: the 7x7 maps of
 * stage 4, laud_resnet.py:115-144 on the image's active channels) as ONE launch, one workgroup per image: conv1 -> bn1 + ReLU ->
 * conv2 3x3 -> bn2 + ReLU -> conv3 -> bn3 + residual + ReLU.  h1 and h2 stay in the workgroup's LDS; the eight waves tile the image as
 * 2 pixel tiles x 4 channel quarters, so widths up to 512 fit the accumulator registers (csrc/ldn_small.hip).  Operands, weight
 * layouts and formulas are those of ldn_bottleneck_head + ldn_bottleneck_tail (stride 1); width % 64 == 0 and <= 512, cin % 32 == 0,
 * cout % 128 == 0; x / residual / out [B*H*Wd][ldx / ldr / ldo] fp32, out may alias residual and x (in-place residual stream).
 *   colsum (optional) [B][2][cout]: partial sums of out over the two pixel tiles (the next block's channel masker: gap_partial, 2 splits).
 * ldn_bottleneck_smallmap_fits: 1 when this map / these widths fit the 160 KiB of LDS for EVERY channel count (0: use the other paths). */
int ldn_bottleneck_smallmap_fits(int H, int Wd, int cin, int width, int cout);
int ldn_bottleneck_smallmap(const float* x, int ldx, int B, int H, int Wd, int cin, int width, const void* w1_split,
                            const void* w2_pairs, const void* w3_pairs, int cout, const int32_t* ch_idx, const int32_t* ch_cnt,
                            const float* scale1, const float* shift1, const float* post_sub1, const float* scale2,
                            const float* shift2_tab, const float* post_sub2, const float* shift3, const float* residual, int ldr,
                            float* out, int ldo, float* colsum, void* stream);

/* ---- a2 + a7 (channel mode, bf16x3): a RUN of consecutive stride-1 channel-mode bottlenecks on maps of at most 256 pixels
 * (stage 3 of the ResNets: 14x14) as ONE launch -- per block: ldn_channel_masker on the GAP of the block's input, then
 * ldn_bottleneck_head, then ldn_bottleneck_tail (laud_resnet.py:104-147, block after block; models/utils.py:92-131).
 * Workgroup b walks image b through the whole run; images never synchronise with each other, so a launch no longer lasts as
 * long as its heaviest image per phase.  Results are bit-identical to the three stand-alone entry points.
 *   blocks [nblocks]  DEVICE array of per-block constants (layouts as in ldn_bottleneck_head / _tail / ldn_channel_masker;
 *                     mw2 / mb2 NULL and hidden == 0 for a one-layer masker); every block has the same shapes
 *   x_in   [B*H*Wd][ldx]  input of the first block (C channels; also its residual); x_work: the residual stream the run
 *                     updates (may alias x_in: in place); holds the run's output afterwards
 *   gap_in [B][gap_splits][C]  partial channel sums of x_in (ldn_bottleneck_tail colsum / ldn_conv_image colsum / any producer)
 *   colsum [B][8][C]      GAP partials of the run's output (and scratch between blocks)
 *   masks [nblocks][B][G], ch_idx [nblocks][B][width], ch_cnt [nblocks][B]: the maskers' decisions and channel lists, per block
 *   h1_split [B*H*Wd][ldh] scratch (conv1's pre-split output). */
typedef struct ldn_chain_block {
    const void* w1_split; const float* scale1; const float* shift1; const float* post_sub1;
    const void* w2_pairs; const void* w3_pairs; const float* scale2; const float* shift2_tab; const float* post_sub2;
    const float* shift3;
    const float* mw1; const float* mb1; const float* mw2; const float* mb2;
} ldn_chain_block;
/* 1 when a run on an H x Wd map (C channels, this width / masker shape) fits the workgroup's 160 KiB of LDS in every phase,
 * 0 otherwise (the caller then launches the blocks one by one: ldn_bottleneck_head / ldn_bottleneck_tail). */
int ldn_bottleneck_chain_fits(int H, int Wd, int C, int width, int hidden, int G);
int ldn_bottleneck_chain(const float* x_in, float* x_work, int ldx, int B, int H, int Wd, int C, int width,
                         const ldn_chain_block* blocks, int nblocks, int hidden, int G, int gran, const float* gap_in,
                         int gap_splits, float* colsum, float* masks, int32_t* ch_idx, int32_t* ch_cnt, void* h1_split,
                         int ldh, void* stream);
int ldn_bottleneck_chain_f32(const float* x_in, float* x_work, int ldx, int B, int H, int Wd, int C, int width,
                         const ldn_chain_block* blocks, int nblocks, int hidden, int G, int gran, const float* gap_in,
                         int gap_splits, float* colsum, float* masks, int32_t* ch_idx, int32_t* ch_cnt, float* h1,
                         int ldh, void* stream);

/* ---- a8: the static stem of ResNet.forward in eval mode (laud_resnet.py:316-326: conv1 7x7 stride 2 pad 3 -> bn1 -> ReLU ->
 * max-pool 3x3 stride 2 pad 1) as ONE launch, bf16x3 arithmetic.  The full-resolution conv output never exists in memory.
 *   x [B,H,W,3] NHWC fp32;  out [B,Hp,Wp,cout] NHWC fp32 with Hc = (H-1)/2+1, Hp = (Hc-1)/2+1 (same for W);  cout = 32 or 64
 *   out = relu(maxpool(conv(x, bn1.scale * conv1.weight)) + shift)     shift[c] = bn1.bias - bn1.running_mean * bn1.scale
 *         (== maxpool(relu(bn1(conv1(x)))): adding a per-channel constant and the ReLU commute with the max)
 *   w_frag: ldn_stem_weight_bytes(cout) bytes, MFMA fragment order [cout/32][11 k-steps][64 lanes][8 hi | 8 lo] bf16:
 *           lane (n = lane & 31, h = lane >> 5) of step s holds k-group q = 2s + h: kernel row ky = q / 3 and elements
 *           i = 8 (q % 3) + e of that row's 24 slots, slot i = (kx = i / 3, c = i % 3) for i < 21, zero for the 3 pad slots
 *           and for ky == 7. */
size_t ldn_stem_weight_bytes(int cout);
int ldn_stem_conv_pool(const float* x, int B, int H, int W, const void* w_frag, const float* shift, int cout, float* out,
                       int Hp, int Wp, void* stream);
/* The same launch also leaving the channel sums of its output for the first block's channel masker (a2): gap [B][ldn_stem_gap_splits(H, W)][cout]
 * = per tile of 8 x 7 pooled pixels, the sum of out over the tile's pixels (ldn_channel_masker's gap_partial). */
int ldn_stem_gap_splits(int H, int W);
int ldn_stem_conv_pool_gap(const float* x, int B, int H, int W, const void* w_frag, const float* shift, int cout,
                           float* out, int Hp, int Wp, float* gap, void* stream);

/* ---- a10: the static stem of LAD_RegNet.forward in eval mode (laud_regnet.py:59-71 SimpleStemIN: conv 3x3 stride 2 pad 1 -> BN ->
 * ReLU) as ONE launch, bf16x3 arithmetic: the image is read once, the output written once.
 *   x [B,H,W,3] NHWC fp32;  out [B,Ho,Wo,cout] NHWC fp32 with Ho = (H-1)/2+1, Wo = (W-1)/2+1 (Wo <= 256);  cout = 32 or 64
 *   out = act(conv(x, bn.scale * conv.weight) + shift)     shift[c] = bn.bias - bn.running_mean * bn.scale;  relu != 0: ReLU
 *   w_frag: ldn_stem3_weight_bytes(cout) bytes, MFMA fragment order [cout/32][3 k-steps][64 lanes][8 hi | 8 lo] bf16:
 *           lane (n = lane & 31, h = lane >> 5) of step s holds kernel row ky = s, slots i = 8 h + e of that row's 16 slots,
 *           slot i = (kx = i / 3, c = i % 3) for i < 9, zero for the 7 pad slots. */
size_t ldn_stem3_weight_bytes(int cout);
int ldn_stem3_conv(const float* x, int B, int H, int W, const void* w_frag, const float* shift, int cout, int relu, float* out,
                   int Ho, int Wo, void* stream);

/* ---- a7 (spatial / layer / both): the same kernel over PACKED PIXEL LISTS ---------------------
 * Image b owns the packed rows [row_prefix[b], row_prefix[b+1]) (B == 1, row_prefix == NULL: rows [0, *m_count),
 * m_count == NULL -> m_cap).  For packed row R:
 *   A row of tap t  = a_map ? a_map[R*taps + t] : R        (-1 = zero row; a_map required for taps == 9)
 *   destination row = out_map ? out_map[R] : R             (scatter into the NHWC residual stream)
 *   border class    = from pix_map[R] (flat output pixel b*Ho*Wo + oy*Wo + ox) when shift_classes == 16
 * Channel subsets (k_idx/n_idx, per image) and all epilogue terms are as in ldn_conv_image; relu == 2 applies
 * ReLU only where relu_if_neg[R] < 0.  ldn_conv_rows is the B == 1, no-channel-list special case.
 * dyn_mode 'both' (laud_resnet.py:101-103) = packed rows from ldn_mask_to_index + lists from ldn_channel_masker. */
int ldn_conv_packed(const float* a, int lda, int B, const int32_t* row_prefix, const int32_t* m_count, int m_cap,
                    const int32_t* a_map, int taps, const int32_t* out_map, const int32_t* pix_map, int Hi, int Wi,
                    int Ho, int Wo, int stride, const float* w, int cin, int cout, const int32_t* k_idx,
                    const int32_t* k_cnt, int kgran, const int32_t* n_idx, const int32_t* n_cnt, const float* scale,
                    const float* shift, int shift_classes, const float* post_sub, int relu,
                    const int32_t* relu_if_neg, const float* residual, int ldr, float* out, int ldo, int math_mode,
                    void* stream);

/* ---- a9: LAD-RegNet BottleneckTransform (laud_regnet.py:157-217), layer-skip execution ------------------------
 * b: grouped 3x3 conv (+BN+ReLU) over packed rows: out[r,c] = act(scale[c]*sum_{t<9} sum_{i<gw}
 *    a[nbr[r*9+t], (c/gw)*gw + i] * w[c,t,i] + shift[c]); nbr as produced by ldn_mask_to_index; w is [C][9][gw]. */
int ldn_grouped_conv3x3_rows(const float* a, int lda, const int32_t* nbr, const int32_t* m_count, int m_cap,
                             const float* w, int C, int group_width, const float* scale, const float* shift, int relu,
                             float* out, int ldo, void* stream);
/* b on the matrix cores (group width 16, bf16x3 arithmetic): same contract as ldn_grouped_conv3x3_rows with the weights pre-split
 *    into MFMA fragment order: w_frag = ldn_grouped16_weight_bytes(C) bytes, [C/16][5 steps][64 lanes][8 hi | 8 lo] bf16; lane
 *    (i = lane & 15, kg = lane >> 4) of step s holds w[16 g + i][tap 2 s + (kg >> 1)][8 (kg & 1) + e], zero for tap 9. */
size_t ldn_grouped16_weight_bytes(int C);
int ldn_grouped16_conv3x3_rows(const float* a, int lda, const int32_t* nbr, const int32_t* m_count, int m_cap,
                               const void* w_frag, int C, const float* scale, const float* shift, int relu, float* out,
                               int ldo, void* stream);

/* The same convolution when the packed rows are WHOLE IMAGES (layer skip, one decision per image): kept image k owns the input rows
 * [k Hi Wi, (k+1) Hi Wi) of a and the output rows [k Ho Wo, (k+1) Ho Wo) of out; *m_count = kept images x Ho Wo (device side),
 * images_cap = the batch.  The image's channels are staged in LDS once instead of nine reads per row through the neighbour table.
 * ldn_grouped16_images_fit(Hi, Wi, C) > 0 (groups per workgroup) iff the input map fits the LDS. */
int ldn_grouped16_images_fit(int Hi, int Wi, int C);
int ldn_grouped16_conv3x3_images(const float* a, int lda, const int32_t* m_count, int images_cap, int Hi, int Wi, int Ho, int Wo,
                                 int stride, const void* w_frag, int C, const float* scale, const float* shift, int relu, float* out,
                                 int ldo, void* stream);
/* b in CHANNEL mode (laud_regnet.py:160-189: the mask is applied AFTER conv+BN+ReLU, so masked channels are exact zeros and the
 *    subset execution is exact): dense image whose columns are left-packed per image (column j of image b = channel
 *    ch_idx[b,j], ascending, j < ch_cnt[b]); output column j sums over the ACTIVE input channels of its group only.
 *    a [B,Hi,Wi,lda], out [B,Ho,Wo,ldo] (columns >= ch_cnt[b] written as zeros), w [C][9][gw], pad 1. */
int ldn_grouped_conv3x3_image(const float* a, int lda, int B, int Hi, int Wi, int stride, int Ho, int Wo, const float* w,
                              int C, int group_width, const int32_t* ch_idx, const int32_t* ch_cnt, const float* scale,
                              const float* shift, int relu, float* out, int ldo, void* stream);
/* se: torchvision SqueezeExcitation (laud_regnet.py:119-123,194) applied IN PLACE to the packed rows of every kept
 *    image: gate[b,:] = sigmoid(W2 relu(W1 mean_rows(a_b) + b1) + b2), a[r,:] *= gate[image(r),:].
 *    w1 [S][C], w2 [C][S].  ch_idx / ch_cnt (optional, [B][C] / [B]): the columns of image b are its active channels
 *    (channel mode); the weights are gathered through the list.  work: ldn_se_packed_workspace_bytes. */
int ldn_se_packed(float* a, int lda, const int32_t* row_prefix, int B, int C, int S, const float* w1, const float* b1,
                  const float* w2, const float* b2, const int32_t* ch_idx, const int32_t* ch_cnt, int max_rows_per_image,
                  float* work, void* stream);
size_t ldn_se_packed_workspace_bytes(int B, int C, int max_rows_per_image);

/* The SE block folded into its neighbours (layer skip on whole images; laud_regnet.py:194-197 `x = self.b(x); x = self.se(x); x = self.c(x)`):
 *   1. ldn_grouped16_conv3x3_images_gap = ldn_grouped16_conv3x3_images that also leaves the channel SUMS of its output per kept image:
 *      gap_partial [images_cap][bands][C], bands = ldn_grouped16_images_bands(Hi, Wi, Ho, stride, C) (1 when a workgroup holds a whole image);
 *      a fixed order of additions (a tree over each tile of 16 pixels, tiles in order): deterministic;
 *   2. ldn_se_gate_slots: gate[k][:] = sigmoid(W2 relu(W1 (sum over bands / rows_per_image) + b1) + b2) for every kept image k
 *      (k * rows_per_image < *m_count), gate [images_cap][C]; w1 [S][C], w2 [C][S];
 *   3. ldn_conv_rows_gated: conv c reads h_b and multiplies row r by gate[r / gate_rows][:] in flight (bf16x3; products formed from
 *      fp32(h_b * gate): bit-identical to scaling h_b first); any cin % 8 == 0 (<= 2048), cout % 4 == 0; (255 / gate_rows + 2) *
 *      roundup32(cin) * 4 <= 44 KB (the gate vectors of the images a 256-row tile touches sit in LDS); relu 0 / 1 / 2; residual /
 *      out_rows as ldn_conv_rows_split.
 * Three launches (conv b, the SE head, conv c) instead of six; h_b is read once by conv c and never rewritten. */
int ldn_grouped16_images_bands(int Hi, int Wi, int Ho, int stride, int C);
int ldn_grouped16_conv3x3_images_gap(const float* a, int lda, const int32_t* m_count, int images_cap, int Hi, int Wi, int Ho, int Wo,
                                     int stride, const void* w_frag, int C, const float* scale, const float* shift, int relu, float* out,
                                     int ldo, float* gap_partial, void* stream);
int ldn_se_gate_slots(const float* gap_partial, int splits, const int32_t* m_count, int images_cap, int rows_per_image, int C, int S,
                      const float* w1, const float* b1, const float* w2, const float* b2, float* gate, void* stream);
int ldn_conv_rows_gated(const float* a, int lda, const int32_t* m_count, int m_cap, const void* w_split, int cin, int cout,
                        const float* scale, const float* shift, int relu, const int32_t* relu_if_neg, const int32_t* out_rows,
                        const float* residual, int ldr, float* out, int ldo, const float* gate, int gate_rows, void* stream);

/* ---- a14: token skipping (BASELINE config 5, AdaViT / DeiT-S shaped blocks).  The reference holds no model code for it, only the
 * latency model of the operator (DyNetSimulator/adavit/simulate_adavit.py:77-131: q / k / v for every token, attention
 * [B, heads, L_select, d] over the selected tokens): parity of this entry point is therefore UNPINNED -- it is tested against a
 * dense masked attention written for the purpose (oracle/adavit_ref.py).
 * qkv [rows][ld_qkv] fp32: per token row q | k | v, each [heads][64].  tok_rows [N]: flat row of every kept token, image-major and
 * ascending (ldn_mask_to_index of the [B, L, 1] keep mask gives it as idx3); img_prefix [B + 1]: exclusive prefix of kept tokens per
 * image (at most max_tokens <= 256 each).  out [N][ldo]: row n = softmax(scale * q_n K_b^T) V_b over the kept tokens of n's image,
 * heads concatenated.  bf16x3 products, fp32 softmax. */
int ldn_packed_mha(const float* qkv, int ld_qkv, const int32_t* tok_rows, const int32_t* img_prefix, int B, int heads,
                   int head_dim, int max_tokens, float scale, float* out, int ldo, void* stream);
/* ... with HEAD skipping (simulate_adavit.py:81-88: attention over the selected heads of every image): head_keep [B][heads] {0,1}; the
 * workgroup of a dropped (image, head) writes zeros to its 64 output columns and computes nothing. */
int ldn_packed_mha_heads(const float* qkv, int ld_qkv, const int32_t* tok_rows, const int32_t* img_prefix, int B, int heads,
                         int head_dim, int max_tokens, float scale, const float* head_keep, float* out, int ldo, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LDN_HIP_H */
