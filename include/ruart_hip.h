/* ruart_hip.h - C ABI of libruart_hip.so (gfx950 / MI355X).
 *
 * The reference (xiaojino/RUArt) has no FFI: its hot path is stock PyTorch ops called from Python
 * (SURVEY.md section 8b).  This header is the boundary one level below the Python drop-in classes
 * (ruart_amd.SDNet / SDNetTrainer / VQA_collate): one extern "C" launcher per kernel family, each
 * citing the reference op site it replaces.  INTEGRATION.md shows the ctypes stub a maintainer of the
 * reference would add to call them from Models/Bert/Bert.py and Models/Layers.py.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the comment says "host";
 *   - no allocation, no synchronisation, no host<->device copy inside any entry point: the caller owns
 *     all buffers and workspaces and passes the hipStream_t (as void*) to launch on; launches are
 *     graph-capturable;
 *   - return value: 0 on success, otherwise the hipError_t of the failed check/launch
 *     (hipErrorInvalidValue = 1 for shape/alignment contract violations);
 *   - dtype codes: RUART_DT_F32 = 0 (exact fp32 validation path), RUART_DT_BF16 = 1 / RUART_DT_F16 = 2 (production
 *     paths: 16-bit storage and MFMA operands - both forms run at the same MFMA rate - with fp32 accumulation and
 *     fp32 softmax / layer-norm / GELU internals);
 *   - matrices are row-major with an explicit leading dimension in ELEMENTS.
 */
#ifndef RUART_HIP_H
#define RUART_HIP_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RUART_DT_F32 0
#define RUART_DT_BF16 1
#define RUART_DT_F16 2
#define RUART_ACT_NONE 0
#define RUART_ACT_GELU 1
#define RUART_ACT_RELU 2

/* library / build info: returns a static string such as "ruart_hip 0.1 gfx950" */
const char* ruart_version(void);

/* ---- dense projections (reference: nn.Linear sites, Models/Bert/modeling.py:225-227, 261, 287-288, 300;
 *      Models/Layers.py:226-227 for the fp32 form) --------------------------------------------------------
 * C[M,N] = act(A[M,K] . W[N,K]^T + bias[N]) + residual[M,N]
 * 16-bit form (in_dtype = BF16 or F16): M % 128 == 0, N % 128 == 0, K % 64 == 0 (the caller pads M; BERT's N, K always
 *            qualify); residual_dtype / out_dtype are in_dtype or F32; GELU and residual are exclusive. */
int ruart_gemm_16_nt(const void* A, int lda, const void* W, int ldw, const float* bias, const void* residual, int ldr,
                     int residual_dtype, void* C, int ldc, int out_dtype, int M, int N, int K, int act, int in_dtype,
                     void* stream);
/* ruart_gemm_16_nt with the TAIL SPLIT of ruart_gemm_16c_nt_ws (below): tail_ws = ruart_gemm_16_tail_ws_bytes(M, N, K, cus) bytes of scratch,
 * cus = the CU count the plan is made for; NULL / too small / cus <= 0: the single launch.  Takes effect on the 256 x 256 four-phase kernel. */
size_t ruart_gemm_16_tail_ws_bytes(int M, int N, int K, int cus);
int ruart_gemm_16_nt_ws(const void* A, int lda, const void* W, int ldw, const float* bias, const void* residual, int ldr, int residual_dtype,
                        void* C, int ldc, int out_dtype, int M, int N, int K, int act, int in_dtype, void* tail_ws, size_t tail_ws_bytes, int cus,
                        void* stream);
/* The same projections in the "f16 + fp8 correction" precision mode (csrc/gemm_corr.hip):
 *   C = act(A16 . W16^T + 2^-18 * A8 . W8^T + bias) [+ residual]
 * A16 (M, K) f16 and A8 (M, 2K) e4m3 bytes with the SAME row pitch in bytes (2 * lda): A8 row = [fp8((a - f16(a)) * 2^11), K bytes |
 * fp8(a), K bytes]; W16 (N, K) f16 and W8 (N, 2K) likewise with [fp8(f16(w) * 2^7) | fp8((w - f16(w)) * 2^18)].  The fp8 part
 * runs on v_mfma_scale_f32_16x16x128_f8f6f4 (twice the f16 MFMA rate) into the same fp32 accumulators, so the product carries
 * ~2^-16 relative operand error instead of the 2^-12 of a plain f16 product, at 2x (not 3x) the matrix time.
 *   act NONE: C (M, N) fp32, optional fp32 residual (row stride ldr), C8 must be NULL;
 *   act GELU: C (M, N) f16 and C8 (M, 2N) bytes (row pitch 2 * ldc) in the A8 layout - the operand of the next projection.
 * M % 256 == 0, N % 256 == 0, K % 128 == 0. */
int ruart_gemm_16c_nt(const void* A16, const void* A8, int lda, const void* W16, const void* W8, int ldw, const float* bias,
                      const float* residual, int ldr, void* C, int ldc, void* C8, int M, int N, int K, int act, void* stream);
/* out4 = {SA_LO, SA_HI, SW_HI, SW_LO}: the power-of-two exponents of the e4m3 companions described above, as compiled into this library
 * (activation residual, activation value, weight value, weight residual); 2^-(SA_LO + SW_HI) is the scale of the fp8 product. */
int ruart_f16c_shifts(int* out4);
/* Tile form of the fp16c products without a residual (the QKV and intermediate dense sites, Models/Bert/modeling.py:225-227, 261;
 * ruart_gemm_16c_nt / _ws / _fold kinds 0 and 2): 0 = 256 x 256 tiles, one workgroup of eight waves per CU; 1 = 256 x 128 tiles, TWO
 * workgroups of four waves per CU, so that one multiplies while the other stores its tile (gemm_corr.hip, gemm_16c_nt_256x128d).  Same
 * products in the same order per output element: results are bit-identical.  Returns the previous setting. */
int ruart_gemm_16c_set_dual(int on);
/* The same product with a chosen subset of the correction terms (same sites; the ablation of tools/corr_ablation.py): corr 3 = both
 * (== ruart_gemm_16c_nt), 1 = only a_lo . w_hi (the activation's rounding residual), 2 = only a_hi . w_lo (the weight's), 0 = none
 * (a plain f16 product through this kernel).  corr 1 / 2 need K % 256 == 0. */
int ruart_gemm_16c_nt_sel(const void* A16, const void* A8, int lda, const void* W16, const void* W8, int ldw, const float* bias,
                          const float* residual, int ldr, void* C, int ldc, void* C8, int M, int N, int K, int act, int corr, void* stream);
/* The same product with a TAIL SPLIT: a 256 x 256 tile per CU runs a product in whole rounds of `cus` tiles, and the tiles of the last,
 * partial round (e.g. 21 of 501 on a 240-CU stream: 2.09 -> 3 rounds) idle most of the chip for a full tile time.  With a workspace
 * of ruart_gemm_16c_tail_ws_bytes(M, N, K, cus) bytes those tiles are cut along K into 2 .. 8 slices (short workgroups dispatched behind
 * the full tiles) and a second small launch adds a tile's slices in slice order and runs its epilogue: deterministic, and a function of
 * (M, N, K, cus) only.  tail_ws NULL / too small, cus <= 0, or corr != 3: exactly ruart_gemm_16c_nt_sel. */
size_t ruart_gemm_16c_tail_ws_bytes(int M, int N, int K, int cus);
int ruart_gemm_16c_nt_ws(const void* A16, const void* A8, int lda, const void* W16, const void* W8, int ldw, const float* bias,
                         const float* residual, int ldr, void* C, int ldc, void* C8, int M, int N, int K, int act, int corr, void* tail_ws,
                         size_t tail_ws_bytes, int cus, void* stream);
/* Training forward of the intermediate dense (Models/Bert/modeling.py:287-288): G16 = gelu(A . W^T + bias) and the pre-activation H16, both
 * (M x N) in the operands' 16-bit type, row stride ldc.  M, N % 256 == 0, K % 128 == 0. */
int ruart_gemm_16_nt_gelu2(const void* A, int lda, const void* W, int ldw, const float* bias, void* H16, void* G16, int ldc, int M, int N, int K,
                           int in_dtype, void* stream);
/* Backward from the output dense to the intermediate dense output in one kernel: acc = dY (M x K bf16) . Wt (N x K bf16 = W2^T)^T, then with
 * the saved f16 pre-activation H (M x N, row stride ldh): dH = acc * gelu'(H) and G = gelu(H), both bf16 (row stride ldc); colpart (optional,
 * ruart_gemm_16_nt_gelu_bwd_ws_floats(M, N) floats) = column sums of the unrounded dH per 128-row strip, strip-major - sum the M / 128 strips
 * in order for the intermediate bias gradient (ruart_colsum_f32_rows).  M, N % 256 == 0, K % 128 == 0. */
size_t ruart_gemm_16_nt_gelu_bwd_ws_floats(int M, int N);
int ruart_gemm_16_nt_gelu_bwd(const void* dY_bf16, int lda, const void* Wt_bf16, int ldw, const void* H16, int ldh, void* dH_bf16, void* G_bf16,
                              int ldc, float* colpart, int M, int N, int K, void* stream);
/* Split-K form for weight gradients (dW = dY^T . X with both operands given K-contiguous, i.e. transposed: ruart_transpose16):
 * part[z] (M x N fp32, row stride ldc; slabs M * ldc floats apart) = A[:, z kchunk ...] . W[:, z kchunk ...]^T, z < ceil(K / kchunk);
 * sum the slabs with ruart_splitk_reduce.  M, N % 256 == 0, K % 128 == 0, kchunk % 128 == 0. */
int ruart_gemm_16_nt_splitk(const void* A, int lda, const void* W, int ldw, float* part, int ldc, int M, int N, int K, int kchunk,
                            int in_dtype, void* stream);
/* The same product straight from the layouts the backward pass holds - no transposes: part[z] (M x N fp32) = P[zT..][:, :M]^T . Q[zT..][:, :N]
 * over token rows z * tchunk .. of P (T x M, row stride ldp) and Q (T x N, row stride ldq), both bf16 or both f16
 * (autograd's grad_output.t().mm(input) of the nn.Linear sites, Models/Bert/modeling.py:225-227, :261, :287, :300).
 * M, N % 256 == 0, T % 128 == 0, tchunk % 128 == 0; rows past the real tokens must be zero in at least one operand and finite in both. */
int ruart_gemm_16_tn_splitk(const void* P, int ldp, const void* Q, int ldq, float* part, int ldc, int M, int N, int T, int tchunk,
                            int in_dtype, void* stream);
/* Tuning knob: pins GROUP_M of the L2-friendly tile walk of ruart_gemm_16_nt / ruart_gemm_16c_nt (0 = plain row-major; 1 .. 64); -1 returns
 * to the default, a per-problem rule in the tile counts and K (csrc/gemm_shared.h, ruart_tile_group_m). */
int ruart_gemm_set_tile_order(int group_m);
/* Tile variant of ruart_gemm_16_nt: 5 (default) = 256x256 tile, four phases per K-tile with the prefetch in flight across
 * barriers (needs M, N % 256 == 0 and K % 128 == 0); 3 = 256x256 tile, plain two-stage loop (M, N % 256 == 0); 0 = 128x128
 * tile; 7 = round 4's experiment, one wave per SIMD: 4 waves x 128x128 per 256x256 tile, accumulators in AGPRs, one barrier per K-tile
 * (M, N % 256 == 0, K % 64 == 0, K >= 128; 7-12 % slower than 5 on the encoder's shapes).  A shape a variant cannot take
 * falls back to the next one.  All variants sum in the same order and are bitwise identical (tools/gemm_check.py,
 * test_gemm_16_one_wave_per_simd_variant).  Other values are rejected. */
int ruart_gemm_set_variant(int v);
/* fp32 form: any M, N, K; act in {NONE, GELU, RELU}; bias / residual may be NULL. */
int ruart_gemm_f32_nt(const float* A, int lda, const float* W, int ldw, const float* bias, const float* residual, int ldr,
                      float* C, int ldc, int M, int N, int K, int act, void* stream);

/* Live timing of the dominant kernel for bench.py's roofline: while enabled, every ruart_gemm_16_nt launch is
 * bracketed by a hipEvent pair on its stream (pool of 8192 launches).  ruart_prof_read synchronises on those events,
 * returns the summed kernel time, launch count and ALGORITHMIC flops (2 * real_rows * N * K), and resets the pool.
 * The only entry points that synchronise or allocate; never call them inside a timed or captured region.
 * on = 2: only the markers of ruart_prof_mark are recorded, the GEMM launches are left alone (six events per step instead of ~100). */
int ruart_prof_enable(int on);
int ruart_prof_read(double* total_ms, long long* launches, double* flops);
/* Diagnostics on the same record pool: ruart_prof_mark puts a marker (flops = -tag) on any stream; ruart_prof_timeline returns every
 * record's begin / end time in ms relative to the first record (it synchronises on the events; the pool is left as it is). */
int ruart_prof_mark(int tag, void* stream);
int ruart_prof_timeline(float* begin_ms, float* end_ms, double* flops, int max_records, int* n_records);

/* ---- BERT row kernels ------------------------------------------------------------------------------------ */
/* Models/Bert/modeling.py:185-199: out[r] = LN(word[ids[r]] + pos[pos_ids[r]] + type[0]). */
int ruart_bert_embed_ln(const int* ids, const int* pos_ids, const float* word_emb, const float* pos_emb, const float* type_emb,
                        const float* gamma, const float* beta, float eps, void* out, int ldo, int out_dtype, int rows, int H,
                        void* stream);
/* Models/Bert/modeling.py:164-168: TF-style layer norm (eps inside the sqrt) of fp32 rows. H % 4 == 0, H <= 1024. */
int ruart_rows_layernorm(const float* x, int ldx, const float* gamma, const float* beta, float eps, void* out, int ldo,
                         int out_dtype, int rows, int H, void* stream);
/* The two row kernels above writing the "f16 + fp8 correction" triple (RUART_DT_F16C mode of the encoder): out32 (rows, H) fp32 - the
 * residual stream / layer output -, out16 (rows, H) f16 and out8 (rows, 2H) e4m3 bytes [lo | hi] (ruart_gemm_16c_nt's A16 / A8);
 * all three with row stride ldo elements (out8: 2 * ldo bytes). */
int ruart_rows_layernorm_split(const float* x, int ldx, const float* gamma, const float* beta, float eps, float* out32, void* out16,
                               void* out8, int ldo, int rows, int H, void* stream);
int ruart_bert_embed_ln_split(const int* ids, const int* pos_ids, const float* word_emb, const float* pos_emb, const float* type_emb,
                              const float* gamma, const float* beta, float eps, float* out32, void* out16, void* out8, int ldo, int rows,
                              int H, void* stream);
/* Models/Bert/modeling.py:234-250 on a packed token stream.  qkv rows are [Q | K | V] (3H wide), Q already
 * scaled by 1/sqrt(64).  Two kinds of query blocks:
 *   short windows (n_blocks): block b covers tokens [blk_q0[b], blk_q1[b]) (<= 64, whole short sequences) and stages keys
 *     [blk_k0[b], blk_k1[b]); token t attends to keys [tok_lo[t], tok_hi[t]) (its own sequence) - VALU kernel, any dtype;
 *   long blocks (n_long_blocks): up to 128 consecutive queries of ONE sequence, keys [lblk_k0, lblk_k1) = that whole sequence -
 *     MFMA flash-attention kernel, 16-bit dtypes only (in fp32 mode the host plans long sequences as short-window blocks).
 * key_bias (may be NULL) is added to every score of key j (the reference's -10000 for kept-but-masked positions). */
int ruart_bert_attention(const void* qkv, int ld, void* ctx, int ldc, int dtype, int H, int n_heads, int n_blocks,
                         const int* blk_q0, const int* blk_q1, const int* blk_k0, const int* blk_k1, const int* tok_lo,
                         const int* tok_hi, const float* key_bias, int n_long_blocks, const int* lblk_q0, const int* lblk_q1,
                         const int* lblk_k0, const int* lblk_k1, void* stream);
/* ruart_bert_attention for the RUART_DT_F16C mode: fp32 [Q | K | V] rows in (the scores, the softmax and P.V stay fp32 - they are
 * 0.1 % of the encoder's flops at item lengths of 3-8 pieces), context rows out as ctx16 (f16) + ctx8 (2H e4m3 bytes per row, [lo | hi]),
 * both with row stride ldc elements.  Short-window blocks only (the host plans a long sequence as 64-query blocks over all its keys). */
int ruart_bert_attention_split(const float* qkv, int ld, void* ctx16, void* ctx8, int ldc, int H, int n_heads, int n_blocks,
                               const int* blk_q0, const int* blk_q1, const int* blk_k0, const int* blk_k1, const int* tok_lo,
                               const int* tok_hi, const float* key_bias, void* stream);
/* heads one workgroup of ruart_bert_attention_split walks (the next head's loads in flight under this head's products; rounded down
 * to a divisor of n_heads): 0 or 1 = the one-head kernel of rounds 2-4.  Results are bit-identical for every setting. */
int ruart_bert_attention_split_set_heads(int heads_per_workgroup);
/* Models/Bert/Bert.py:149-165 + Models/SDNet.py:573-581: out[dst_row[w]] = sum_l layer_w[l] *
 * mean(layer_l[span_start[w] .. +span_len[w])).  layers = n_layers matrices [rows, H], layer_stride elements apart.
 * span_start_last (optional): the spans' first rows inside the LAST layer's matrix when ruart_bert_forward left it compacted
 * (ruart_bert_batch.last_rows); NULL = the same rows as in every other layer.
 * Rows of `out` that no word maps to are left untouched (the caller zero-fills: masked words are zeros). */
/* forward kernel form: 1 (default) = workgroup split along the columns (no cross-wave sum) where H % 256 == 0; 0 = split along layers */
int ruart_bert_pool_set_variant(int cols);
int ruart_bert_pool_mix(const void* layers, long long layer_stride, int ldl, int dtype, int n_layers, const int* span_start,
                        const int* span_start_last, const int* span_len, const int* dst_row, const float* layer_w, float* out, int ldo,
                        int n_words, int H, void* stream);
/* d(loss)/d(layer_w[l]) for the op above; partial_ws holds n_words * n_layers floats. Deterministic.  (Both: H % 4 == 0, H <= 1024.) */
int ruart_bert_pool_mix_bwd(const void* layers, long long layer_stride, int ldl, int dtype, int n_layers, const int* span_start,
                            const int* span_start_last, const int* span_len, const int* dst_row, const float* grad_out, int ldg,
                            float* partial_ws, float* grad_layer_w, int n_words, int H, void* stream);
/* dst_k[i][0 .. bytes_k) = src_k[rows[i]][0 .. bytes_k) for up to three row-major byte matrices (k = 0..2; a NULL src ends the list):
 * the compaction of the last encoder layer's inputs to the rows of ruart_bert_batch.last_rows.  bytes_k % 16 == 0, 16-byte aligned. */
int ruart_rows_gather(const int* rows, int n, const void* src0, long long sp0, void* dst0, long long dp0, int bytes0, const void* src1,
                      long long sp1, void* dst1, long long dp1, int bytes1, const void* src2, long long sp2, void* dst2, long long dp2,
                      int bytes2, void* stream);
int ruart_cast_f32_to_16(const float* in, void* out, int out_dtype, long long n, float scale, void* stream);

/* ---- kernels of the TRAINABLE encoder's 16-bit path (conf without LOCK_BERT, opt['bert_train_gemm'] = '16'; csrc/bert_train_*.hip).
 * Activations f16, GEMM-bound gradients bf16, residual-stream gradient fp32.  Dropout masks are regenerated from (seed, element index)
 * by a counter-based hash: the same seed in the forward and the backward call gives the same mask. ------------------------------- */
/* Models/Bert/modeling.py:260-264 / 299-303: y = LN(dropout(x) + res) with x fp32 (dense output incl. bias) and res f16 (post = 0), or
 * :196-199: y = dropout(LN(x)) (post = 1, res ignored).  Saves pre16 = f16(LayerNorm input) and stats[row] = (mean, rstd). */
int ruart_ln_train_fwd(const float* x, int ldx, const void* res16, int ldr, const float* gamma, const float* beta, float eps, float p,
                       unsigned seed, int post, void* y16, void* pre16, float* stats, int ld16, int rows, int H, void* stream);
/* backward: dy fp32 (+ add_scale[0] * add when add != NULL) -> d_res fp32 (gradient at the LayerNorm input: the residual path) and, post = 0,
 * d_gemm bf16 (the same times the dropout multiplier: gradient at the dense output); d_gamma / d_beta (H) and, when not NULL and post = 0,
 * d_bias (H: column sums of the unrounded d_gemm = bias gradient of the dense layer in front) written or accumulated.
 * ws: ruart_ln_train_bwd_ws_floats(H) floats. */
size_t ruart_ln_train_bwd_ws_floats(int H);
int ruart_ln_train_bwd(const float* dy, int ldy, const float* add, const float* add_scale, const void* pre16, int ld16, const float* stats,
                       const float* gamma, float p, unsigned seed, int post, float* d_res, int ldd, void* d_gemm_bf16, int ldg,
                       float* d_gamma, float* d_beta, float* d_bias, int accumulate, float* ws, int rows, int H, void* stream);
/* One pass over an fp32 master weight W (rows x cols, row stride ldw): out16 (rows x cols f16, row stride ld16) = scale * W, the forward's
 * GEMM operand, and outT_bf16 (cols x rows bf16, row stride ldT) = (scale * W)^T, the operand of dX = dY . W as an NT product.  Either output
 * may be NULL; both may point into wider matrices (the fused QKV weight: three items with row / column offsets).  The batch form takes up to
 * eight weights in one launch (the six of a BERT layer). */
typedef struct {
  const float* w;
  void* out16;
  void* outT_bf16;
  int ldw, ld16, ldT, rows, cols;
  float scale;
} ruart_wprep_item;
int ruart_weight_prep_batch(const ruart_wprep_item* items, int n, void* stream);
int ruart_weight_prep(const float* w, int ldw, float scale, void* out16, int ld16, void* outT_bf16, int ldT, int rows, int cols, void* stream);
/* bias gradients of an fp32 matrix of any width and row stride (the trunk's projections, Models/Layers.py:155, 226 under autograd: widths 250,
 * 300, 1000 ...): out[j] (+)= sum_r x[r][j], two ordered stages; ws: ruart_colsum_f32_ws_floats(rows, cols) floats */
size_t ruart_colsum_f32_ws_floats(int rows, int cols);
int ruart_colsum_f32(const float* x, int ld, int rows, int cols, float* out, int accumulate, float* ws, void* stream);
/* out[j] (+)= sum over `rows` rows of part[r * ld + j], rows in a fixed order (the reduction behind per-strip partial sums such as
 * ruart_gemm_16_nt_gelu_bwd's colpart) */
int ruart_colsum_f32_rows(const float* part, int rows, int ld, int cols, float* out, int accumulate, void* stream);
/* Backward through the GELU of the intermediate dense (Models/Bert/modeling.py:287-288 under autograd) as an elementwise pass, in place:
 * d (M x N bf16, row stride ld) holds dY . W2 and becomes that times gelu'(h); g = gelu(h) (bf16, same stride; h: the saved f16
 * pre-activation); colpart[(M / 128) x N] = column sums of the unrounded d per 128-row strip.  M % 128 == 0, N % 256 == 0.  Same results
 * contract as ruart_gemm_16_nt_gelu_bwd except that the product was rounded to bf16 once before the multiplication. */
int ruart_gelu_bwd_rows(void* d_bf16, const void* h16, int ld, void* g_bf16, float* colpart, int M, int N, void* stream);
/* f16 -> bf16 copy (n elements, n % 4 == 0): saved activations as the bf16 operand of a weight-gradient product */
int ruart_f16_to_bf16(const void* in16, void* out_bf16, long long n, void* stream);
/* bias gradient: out[j] (+)= sum_r x[r][j] of a bf16 matrix; ws: ceil(rows / 256) * cols floats */
int ruart_colsum_bf16(const void* x_bf16, int ld, int rows, int cols, float* out, int accumulate, float* ws, void* stream);
/* out (cols x rows, row stride ldo) = in^T for a 16-bit matrix (rows x cols, row stride ldi); f16_to_bf16 != 0 converts f16 elements
 * to bf16 on the way (activations as the bf16 operand of a weight-gradient product) */
int ruart_transpose16(const void* in, int ldi, void* out, int ldo, int rows, int cols, int f16_to_bf16, void* stream);
/* C (n floats) = scale * sum_z part[z] (+ C when accumulate), slabs slab_floats apart, z in order (deterministic) */
int ruart_splitk_reduce(const float* part, long long slab_floats, int nz, float* C, long long n, float scale, int accumulate, void* stream);
/* Models/SDNet.py:573-581 on the token stream: out[r] = sum_l w[l] * layers[l][r] (f16 layers, fp32 out) and d w[l] = sum_r <g[r],
 * layers[l][r]> (ws: 512 * n_layers floats) */
int ruart_mix_rows(const void* layers16, long long layer_stride, int ld, int n_layers, const float* w, float* out, int ldo, int rows, int H,
                   void* stream);
int ruart_mix_rows_bwd(const void* layers16, long long layer_stride, int ld, int n_layers, const float* g, int ldg, float* d_w, float* ws,
                       int rows, int H, void* stream);
/* Models/Bert/modeling.py:224-250 for training: windows of whole sequences (<= 64 word pieces per block [blk_q0, blk_q1), keys = the same
 * tokens), f16 [Q | K | V] rows in (Q pre-scaled), f16 context rows out, attention-probability dropout p_drop from `seed`; the backward
 * takes the context gradient (bf16) and writes [dQ | dK | dV] rows (bf16), recomputing the probabilities; bias_part (optional,
 * n_blocks x 2H floats) receives every window's column sums of the unrounded dQ and dV rows ([dQ | dV]: the query / value bias gradients
 * after ruart_colsum_f32_rows over the windows; the key bias gradient is identically zero - the softmax ignores a shift of a row). */
int ruart_attn_train_fwd(const void* qkv16, int ld, void* ctx16, int ldc, int H, int n_heads, int n_blocks, const int* blk_q0,
                         const int* blk_q1, const int* tok_lo, float p_drop, unsigned seed, void* stream);
int ruart_attn_train_bwd(const void* qkv16, int ld, const void* dctx_bf16, int ldc, void* dqkv_bf16, int ldd, int H, int n_heads, int n_blocks,
                         const int* blk_q0, const int* blk_q1, const int* tok_lo, float p_drop, unsigned seed, float* bias_part, void* stream);
/* The same attention for sequences LONGER than one window (65 .. 512 word pieces; Models/Bert/Bert.py:96-99 windows longer inputs): the
 * host cuts such a sequence into chunks of <= 64 consecutive tokens [chunk_q0, chunk_q1), all starting at the sequence's first token in
 * steps of 64 and stored consecutively; chunk_k0 / chunk_k1 = the chunk's whole sequence, chunk_first = the index of the sequence's
 * first chunk.  One workgroup per (chunk, head).  Forward: the chunk's queries against all key tiles of the sequence (online softmax)
 * + lse2 (n_tokens x n_heads floats: log2 of each row's sum of exp(score), read by the backward).  Backward: [dQ | dK | dV] rows of the
 * chunks' tokens in two launches (the chunk's queries over the key tiles; the chunk's keys over the query tiles), no atomics;
 * needs lse2 and two workspaces: delta_ws (n_tokens x n_heads floats) and scale_ws
 * (n_chunks x n_heads floats); bias_part (optional, n_chunks x 2H) as in ruart_attn_train_bwd.  Same dropout stream as the window
 * kernels (indexed by query token and key offset inside the sequence). */
int ruart_attn_train_fwd_long(const void* qkv16, int ld, void* ctx16, int ldc, int H, int n_heads, int n_chunks, const int* chunk_q0,
                              const int* chunk_q1, const int* chunk_k0, const int* chunk_k1, float p_drop, unsigned seed, float* lse2,
                              void* stream);
int ruart_attn_train_bwd_long(const void* qkv16, int ld, const void* dctx_bf16, int ldc, void* dqkv_bf16, int ldd, int H, int n_heads,
                              int n_chunks, const int* chunk_q0, const int* chunk_q1, const int* chunk_k0, const int* chunk_k1, const int* chunk_first, float p_drop, unsigned seed, const float* lse2, float* delta_ws,
                              float* scale_ws, float* bias_part, void* stream);

/* ---- whole BERT encoder (Models/Bert/modeling.py:585-614, all layer outputs kept as Bert.py:137 needs) ---- */
typedef struct {
  int hidden, n_heads, n_layers, intermediate, dtype;
  float ln_eps;
  const float *word_emb, *pos_emb, *type_emb, *emb_ln_g, *emb_ln_b;
  /* host arrays of n_layers device pointers; GEMM weights are [out, in] in `dtype`, biases / LN params fp32 */
  const void* const* w_qkv;   /* [3H, H], rows = Q (pre-scaled by 1/8) | K | V */
  const float* const* b_qkv;  /* [3H] (Q part pre-scaled) */
  const void* const* w_ao;    const float* const* b_ao;
  const float* const* ln1_g;  const float* const* ln1_b;
  const void* const* w_ff1;   const float* const* b_ff1;
  const void* const* w_ff2;   const float* const* b_ff2;
  const float* const* ln2_g;  const float* const* ln2_b;
  int f32_gemm;   /* dtype == F32 only: 0 = exact fp32 MFMA (v_mfma_f32_16x16x4_f32), 1 = fp32 operands split into bf16 hi + lo and
                     multiplied as three 16-bit MFMA products (ruart_gemm_x3: ~2^-16 per product, ~2.5x faster) */
  int corr8;      /* dtype == F16 only: != 0 selects the "f16 + fp8 correction" mode (ruart_gemm_16c_nt): w_* are the f16 weights,
                     w8_* the (out, 2 in) e4m3 companions; layers_out, the residual stream and the QKV rows are fp32 */
  const void* const* w8_qkv;
  const void* const* w8_ao;
  const void* const* w8_ff1;
  const void* const* w8_ff2;
  int tail_cus;   /* CU count the encoder GEMMs plan their TAIL SPLIT for (ruart_gemm_16c_nt_ws / ruart_gemm_16_nt_ws): the CUs of the stream's
                     mask, or of the device; 0 = single-launch products.  The plan depends on (rows, N, K, tail_cus) only, so passes on
                     different streams of one model agree bit for bit.  Enlarges ruart_bert_workspace_bytes by tail_cus x 256 KB. */
  int ln_fold;    /* corr8, or (round 6) plain F16 / BF16: != 0 = the weights are prepared for ruart_bert_forward_folded (the only forward such a
                     model may be given to): for every layer l >= 1, w_qkv[l] / w8_qkv[l] hold W' = Wqkv diag(ln2_g[l-1]) 2^-s and b_qkv[l] holds
                     d = b + Wqkv ln2_b[l-1]; for every layer, w_ff1 / w8_ff1 / b_ff1 the same with ln1_g[l] / ln1_b[l] */
  const float* const* fold_c_qkv;   /* [3H] per layer: c_j = sum_i W'_ji (entry 0 unused: layer 0 reads the materialised embedding rows) */
  const float* const* fold_c_ff1;   /* [intermediate] per layer */
  const float* fold_s_qkv;          /* HOST arrays of n_layers floats: 2^s of the layer's folded QKV / intermediate weights */
  const float* fold_s_ff1;
} ruart_bert_model;

typedef struct {
  int n_tokens;        /* real packed tokens T */
  int n_rows;          /* padded row count Tp >= T, multiple of 128; ids/pos have Tp entries (pad with 0) */
  const int* ids;
  const int* pos_ids;
  int n_blocks;
  const int *blk_q0, *blk_q1, *blk_k0, *blk_k1, *tok_lo, *tok_hi;
  const float* key_bias; /* NULL when every kept token is attendable */
  int n_long_blocks;     /* <=128-query blocks of sequences longer than 64 tokens (MFMA kernel); 0 in fp32 mode */
  const int *lblk_q0, *lblk_q1, *lblk_k0, *lblk_k1;
  /* Optional: the packed rows some word span reads from the LAST layer (Models/Bert/Bert.py:153-165 pools word pieces only; the
   * [CLS] / [SEP] rows - 41 % of the stream at ST-VQA item lengths - feed the next layer's attention and nothing else, and the last
   * layer has no next layer).  Ascending, n_last_rows > 0: the last layer's output projection, FFN and layer norms run on these rows
   * only and layers_out[n_layers - 1] holds them COMPACTED (row i = packed row last_rows[i]); 0 / NULL: every row, in place. */
  int n_last_rows;
  const int* last_rows;
} ruart_bert_batch;

size_t ruart_bert_workspace_bytes(const ruart_bert_model* m, int n_rows);
/* fp16c mode of ruart_bert_forward (Models/Bert/modeling.py:225-227, 261, 287-288, 300): which correction products (the `corr` of
 * ruart_gemm_16c_nt_sel) the QKV / attention-output / intermediate / output projections carry, in the layers whose bit is set in
 * layer_mask.  Default (3, 3, 3, 3, all layers) - the only setting the parity tests hold to 1e-3.  Process-wide. */
int ruart_bert_set_correction(int qkv, int ao, int ff1, int ff2, unsigned long long layer_mask);
/* layers_out: [n_layers][n_rows][hidden] in m->dtype (fp32 when m->corr8).  `m` and `b` are HOST structs. */
int ruart_bert_forward(const ruart_bert_model* m, const ruart_bert_batch* b, void* layers_out, void* workspace,
                       size_t workspace_bytes, void* stream);

/* The fp16c encoder with every LayerNorm (Models/Bert/modeling.py:164-168) folded into the projections around it (csrc/gemm_corr.hip,
 * CorrFold): five launches per layer, no normalised row is ever written.  m->ln_fold != 0 (weights prepared accordingly), H % 256 == 0.
 * layers_pre: [n_layers][n_rows][hidden] fp32 PRE-LayerNorm rows y; ln_stats: [n_layers][n_rows][2] floats (mu, rstd); the layer output
 * the reference keeps is (y - mu) rstd ln2_g[l] + ln2_b[l] - formed on the fly by ruart_bert_pool_mix_ln.  Last layer compacted as in
 * ruart_bert_forward when b->last_rows is set (its stats rows are compacted the same way).
 * Round 6: a plain 16-bit model (corr8 == 0, dtype F16 / BF16, ln_fold != 0) runs the same five-launch layer on ruart_gemm_16_nt_fold:
 * layers_pre is then 16-bit, b->last_rows is refused (hipErrorNotSupported) - whole-sequence encoding, bench.py --mode bert512. */
size_t ruart_bert_workspace_bytes_folded(const ruart_bert_model* m, int n_rows);
int ruart_bert_forward_folded(const ruart_bert_model* m, const ruart_bert_batch* b, void* layers_pre, float* ln_stats, void* workspace,
                              size_t workspace_bytes, void* stream);
/* One projection of that pass (see CorrFold in csrc/gemm_corr.hip): kind 0 fp32 out, kind 2 GELU + split out - with in_part != NULL the
 * A rows are pre-LayerNorm rows and the epilogue finishes rstd 2^s (A W'^T - mu c) + d (`bias` = d, wscale = 2^s); kind 3: y = A W^T +
 * bias + residual (the residual rows normalised with res_part / res_gamma / res_beta when res_part != NULL), written fp32 (C), split
 * (C16, C8) and as row partials out_part.  Partials: [M][4][2] floats, slot t = (sum, sum of squares) of the row over columns
 * 256 t .. 256 t + 255 (N <= 1024); in_np / res_np = slots in use; stat_len = the row length they cover. */
int ruart_gemm_16c_nt_fold(const void* A16, const void* A8, int lda, const void* W16, const void* W8, int ldw, const float* bias, int kind,
                           const float* in_part, int in_np, const float* colc, float wscale, const float* residual, int ldr,
                           const float* res_part, int res_np, const float* res_gamma, const float* res_beta, void* C, int ldc, void* C16,
                           void* C8, float* out_part, int M, int N, int K, int stat_len, float eps, void* stream);
/* The same projections for the plain 16-bit pass (dtype = F16 or BF16; round 6; Models/Bert/modeling.py:164-168 folded into :225-227,
 * 261, 287-288, 300): kind 0 / 2: C (16-bit) = [gelu] (rstd 2^s (A W'^T - mu c) + d), in_part required; kind 3: y = A W^T + bias + residual
 * (16-bit rows, normalised on the way in when res_part != NULL), C = y in 16 bits, out_part = the partials of the unrounded y. */
int ruart_gemm_16_nt_fold(const void* A, int lda, const void* W, int ldw, const float* bias, int kind, const float* in_part, int in_np,
                          const float* colc, float wscale, const void* residual, int ldr, const float* res_part, int res_np,
                          const float* res_gamma, const float* res_beta, void* C, int ldc, float* out_part, int M, int N, int K, int stat_len,
                          float eps, int dtype, void* stream);
/* (mu, rstd) [rows][2] from such partials */
int ruart_rows_stats_finish(const float* part, int np, int rows, float inv_h, float eps, float* stats, void* stream);
/* ruart_bert_pool_mix / _bwd (Models/Bert/Bert.py:149-165 + Models/SDNet.py:573-581) over the pre-LayerNorm rows of the folded pass:
 * ln_stats [n_layers][stats_stride][2], ln_gamma / ln_beta [n_layers][H] (the layers' output LayerNorm parameters); fp32, H % 256 == 0. */
int ruart_bert_pool_mix_ln(const float* layers_pre, long long layer_stride, int ldl, int n_layers, const float* ln_stats,
                           long long stats_stride, const float* ln_gamma, const float* ln_beta, const int* span_start,
                           const int* span_start_last, const int* span_len, const int* dst_row, const float* layer_w, float* out, int ldo,
                           int n_words, int H, void* stream);
/* Tuning knob: 1 (default) = twelve-layer encoders take the form of the FORWARD pooling kernel that keeps gamma / beta in registers
 * over four words per workgroup; 0 = one word per workgroup, tables reloaded; 2 = the backward in its register-table form too
 * (measured slower).  Results agree to the rounding of the fp32 sums. */
int ruart_bert_pool_ln_set_variant(int reg_tables);
int ruart_bert_pool_mix_ln_bwd(const float* layers_pre, long long layer_stride, int ldl, int n_layers, const float* ln_stats,
                               long long stats_stride, const float* ln_gamma, const float* ln_beta, const int* span_start,
                               const int* span_start_last, const int* span_len, const int* dst_row, const float* grad_out, int ldg,
                               float* partial_ws, float* grad_layer_w, int n_words, int H, void* stream);

/* ---- SDNet kernels (Models/Layers.py) ---------------------------------------------------------------------- */
/* Layers.py:228-231 + :244 + :275-288: for each batch b, with pa = x1 W^T (B,L1,h) and pk = x2 W^T (B,L2,h) from the caller,
 *   a = act(pa) * diag,  k = act(pk)      act = ReLU when relu != 0; diag_len 0: no diag, 1: one scalar, h: per column
 *   S = a . k^T;  S[:, j] = -inf where mask[b][j] == 0;  P = softmax_j(S);  out = P . v[b]  (v: (B,L2,D3))
 * P (B, L1, L2) is saved for backward when probs != NULL.  fp32.  L2 <= 384 (the key/value panel lives in LDS). */
int ruart_attn_fwd(const float* pa, const float* pk, const float* v, const unsigned char* mask, const float* diag, int diag_len,
                   int relu, float* out, float* probs, int B, int L1, int L2, int h, int D3, void* stream);
/* gradients of the op above, given grad_out (B,L1,D3) and saved P: grad_pa (B,L1,h), grad_pk (B,L2,h), grad_v (B,L2,D3);
 * grad_diag_partial (B * ceil(L1/16) rows of h floats, one per workgroup; the caller sums the rows - deterministic, no
 * atomics) only for a vector diag, else pass NULL. */
int ruart_attn_bwd(const float* pa, const float* pk, const float* v, const float* probs, const float* grad_out, const float* diag,
                   int diag_len, int relu, float* grad_pa, float* grad_pk, float* grad_v, float* grad_diag_partial,
                   float* ds_ws /* (B,L1,L2) scratch */, int B, int L1, int L2, int h, int D3, void* stream);
/* ruart_attn_fwd / _bwd kernel form: 1 (default) = operand chunks register-prefetched under the previous chunk's MFMAs (key panels of up
 * to 128 rows), 0 = the forms of rounds 1-4.  Bit-identical results. */
int ruart_attn_set_prefetch(int on);
/* The same pair with a multiplier on the probabilities, for attention-probability dropout (Models/Bert/modeling.py:244-246):
 * out = (P * prob_scale) . v with prob_scale (B, L1, L2) fp32 holding 0 or 1/(1-p) (NULL: plain call); the saved `probs` are the
 * pre-dropout P, which is what the softmax backward needs; the backward takes the same prob_scale. */
int ruart_attn_fwd_pscale(const float* pa, const float* pk, const float* v, const unsigned char* mask, const float* diag, int diag_len,
                          int relu, const float* prob_scale, float* out, float* probs, int B, int L1, int L2, int h, int D3,
                          void* stream);
int ruart_attn_bwd_pscale(const float* pa, const float* pk, const float* v, const float* probs, const float* grad_out, const float* diag,
                          int diag_len, int relu, const float* prob_scale, float* grad_pa, float* grad_pk, float* grad_v,
                          float* grad_diag_partial, float* ds_ws, int B, int L1, int L2, int h, int D3, void* stream);

/* Layers.py:167-168: F.layer_norm over the WHOLE tensor of n elements, no affine.  stats[0] = mean, stats[1] = rstd.
 * ws: 2 * 1024 floats of scratch. */
int ruart_whole_ln_fwd(const float* x, float* y, float* stats, float* ws, long long n, float eps, void* stream);
int ruart_whole_ln_bwd(const float* y, const float* grad_y, const float* stats, float* grad_x, float* ws, long long n,
                       void* stream);

/* Fused answer scorer: Models/Layers.py:352-432 (GetFinalScores with useES and no_answer, the two BilinearSeqAttn :435-468 and
 * get_single_score :421-432) for x (B, L, D) fp32 and per-sample vectors u1 (scores the OCR slots i >= ES), u2 (the first ES slots),
 * uh (the no-answer attention) (B, D) - the caller's three small projections of h0, variational-dropout masks of x folded in -,
 * w (D) / bna (1) the no-answer read-out, mask (B, L) uint8 (0 = masked slot; the candidate scores are masked only with mask_flag,
 * the no-answer attention always):  probs (B, L + 1) = softmax([x_i . u(i), w . (softmax(x . uh) . x) + bna]).  a_out (B, L): the
 * no-answer attention, kept for the backward.  One workgroup per sample, fixed summation orders.  D % 4 == 0, L <= 1024. */
int ruart_scorer_fwd(const float* x, const float* u1, const float* u2, const float* uh, const float* w, const float* bna,
                     const unsigned char* mask, float* probs, float* a_out, int B, int L, int D, int ES, int mask_flag, void* stream);
/* gradients of the above for gprobs (B, L + 1): gx (B, L, D), gu1 / gu2 / guh (B, D), and per-sample partials gw_part (B, D), gb_part (B)
 * whose sums over the batch are d w, d bna. */
int ruart_scorer_bwd(const float* x, const float* u1, const float* u2, const float* uh, const float* w, const float* probs,
                     const float* a_in, const float* gprobs, float* gx, float* gu1, float* gu2, float* guh, float* gw_part,
                     float* gb_part, int B, int L, int D, int ES, void* stream);

/* LSTM recurrence (what Layers.py:166 delegates to nn.LSTM), BOTH directions of one layer per call
 * (grid = B x ndir), gate order i,f,g,o, zero initial state, padding not masked.
 *   xproj : (B, T, ndir*4h)  x W_ih^T + b_ih + b_hh, direction d at column offset d*4h (one GEMM by the caller)
 *   w_hh  : (ndir, 4h, h)
 *   y     : (B, T, ndir*h)   direction d at column offset d*h; direction 1 runs t = T-1 .. 0
 *   gates : (B, T, ndir*4h)  post-activation i,f,g,o   } saved for backward when non-NULL
 *   cells : (B, T, ndir*h)   c_t                       }
 *   hprev : (B, T, ndir*h)   h of the previous step of the same direction (0 at its first step): the right-hand operand of
 *                            grad_W_hh = grad_xproj^T . hprev, so the caller needs no shifted copy of y
 * h <= 128, ndir in {1, 2}. */
/* Kernel form of ruart_lstm_fwd / ruart_lstm_bwd: 1 (default) = 16 batch rows per workgroup, the recurrent product on the matrix cores
 * with split-bf16 operands (hi.hi + hi.lo + lo.hi, fp32 accumulation - the arithmetic of ruart_gemm_x3), B * ndir / 16 workgroups;
 * 0 = one workgroup per (row, direction) with exact fp32 FMAs (rounds 1-2).  Process-wide. */
int ruart_lstm_set_variant(int variant);
int ruart_lstm_fwd(const float* xproj, const float* w_hh, float* y, float* gates, float* cells, float* hprev, int B, int T, int h,
                   int ndir, void* stream);
/* BPTT for the op above: grad_y (B,T,ndir*h) -> grad_xproj (B,T,ndir*4h) (gradient w.r.t. the gate pre-activations).
 * grad_W_hh = grad_xproj^T . h_prev, grad_W_ih = grad_xproj^T . x, grad_x = grad_xproj . W_ih are plain GEMMs done
 * by the caller. */
int ruart_lstm_bwd(const float* grad_y, const float* w_hh, const float* gates, const float* cells, float* grad_xproj, int B,
                   int T, int h, int ndir, void* stream);

/* The operands of one BIDIRECTIONAL nn.LSTM layer (Models/Layers.py:124-180 via nn.LSTM; torch's parameter layout, G = 4h gate rows) in
 * one launch: w (2G, K) = [w_ih ; w_ih_reverse] for the input projection of both directions, b (2G) = [b_ih + b_hh ; the reverse pair],
 * whh (2, G, h) = the recurrent matrices as ruart_lstm_fwd takes them.  All fp32, contiguous. */
int ruart_lstm_pack_params(const float* w_ih, const float* w_ih_r, const float* b_ih, const float* b_hh, const float* b_ih_r,
                           const float* b_hh_r, const float* w_hh, const float* w_hh_r, float* w, float* b, float* whh, int G, int K, int h,
                           void* stream);

/* out[w][c] = x[w][c] * mask[row_of[w]][c] (fp32; row strides in elements; D % 4 == 0, 16-byte aligned rows; row_of int64): the
 * variational dropout of a packed (words, D) matrix with one mask row per item (layers.row_dropout; Models/Layers.py:23-30). */
int ruart_rows_scale(const float* x, int ldx, const float* mask, int ldm, const long long* row_of, float* out, int ldo, int rows, int D,
                     void* stream);
/* Pointwise part of ONE step of a wide LSTM over a ragged, length-sorted batch (the `multi2one` LSTM, Models/SDNet.py:137,
 * 269-271: hidden 300, 1-3 real words per item).  pre (n_active, 4h) = x W_ih^T + b + h_prev W_hh^T from the caller's GEMMs;
 * rows < n_active are advanced (acts (n_active,4h) = post-activation i,f,g,o saved for backward), rows >= n_active of
 * h_out / c_out (n_rows, h) copy h_prev / c_prev.  Backward: grad_h / grad_c (may be NULL = zero) -> grad_pre (n_active,4h),
 * grad_c_prev (n_rows,h) and the pass-through part of grad_h_prev (n_rows,h; rows < n_active are zero - their gradient
 * arrives through the recurrent GEMM). */
int ruart_lstm_cell_fwd(const float* pre, const float* h_prev, const float* c_prev, float* h_out, float* c_out, float* acts,
                        int n_active, int n_rows, int h, void* stream);
int ruart_lstm_cell_bwd(const float* grad_h, const float* grad_c, const float* acts, const float* c_prev, const float* c_out,
                        float* grad_pre, float* grad_h_prev, float* grad_c_prev, int n_active, int n_rows, int h, void* stream);

/* Weight gradient of an embedding lookup (Models/SDNet.py:439-493, nn.Embedding tables) from a sort of the ids prepared on the
 * host: order[n] = lookup positions grouped by table row, seg_start[n_seg + 1] = slices of order[], seg_row[n_seg] = the table
 * row of each slice.  grad_out (n, D) fp32; grad_weight (rows, D) must be zero-filled by the caller; rows are summed in the
 * order given (deterministic).  Replaces the device-side sort torch runs in every backward. */
int ruart_embedding_bwd_sorted(const float* grad_out, const int* order, const int* seg_start, const int* seg_row, int n_seg, int D,
                               float* grad_weight, void* stream);
/* Two-level form for tables where one row collects thousands of occurrences per batch (batch._sort_ids cuts such rows into sub-segments of
 * <= 64): ws (n_sub x D floats) receives the sub-segment sums, grad_weight[row_id[r]] = sum of ws rows row_first[r] .. row_first[r+1]. */
int ruart_embedding_bwd_split(const float* grad_out, const int* order, const int* sub_start, int n_sub, const int* row_first, const int* row_id,
                              int n_rows, int D, float* ws, float* grad_weight, void* stream);

/* PHOC table (Utils/cphoc.c:12-113 `build_phoc`, applied per vocabulary word by Utils/CoQAUtils.py:75-87): row w of `out`
 * (n_words x 604 fp32, row stride ldo >= 604, ldo % 4 == 0, 16-byte aligned) = the pyramidal histogram of characters of the word
 * chars[offsets[w] .. offsets[w+1]) - 36 unigrams x 14 regions of levels 2..5, then 50 bigrams x 2 regions of level 2; entries are
 * 0.0 / 1.0.  Words must already be lower-cased and reduced to [a-z0-9] (Utils/phoc.py:8-10); on any other byte the reference
 * raises: here `status` (device int, zero it first; may be NULL) receives 1 + the index of one offending word and that character
 * is skipped.  An empty word gives a zero row. */
int ruart_phoc_table(const unsigned char* chars, const int* offsets, int n_words, float* out, int ldo, int* status, void* stream);

/* Optimizer step (Models/SDNetTrainer.py:366-367: clip_grad_norm_(params, grad_clipping) then Adamax.step()) over all trainable
 * tensors in three launches.  `grads` / `params` / `exp_avg` / `exp_inf` are DEVICE arrays of device pointers (one per tensor);
 * the work list is cut into chunks of <= 8192 elements: chunk c covers elements [c_start[c], c_start[c] + c_count[c]) of tensor
 * c_tensor[c] (c_start multiples of 4).
 *   ruart_grad_norm_clip: partial (n_chunks floats, scratch); norm_coef[0] = total 2-norm, norm_coef[1] = min(1, max_norm / (norm + 1e-6));
 *                         extra_sq (device float, may be NULL) is added to the sum of squares before the root - the squared norm of
 *                         gradients the chunk list leaves out (data parallelism: the re-pinned embedding rows, which are not exchanged).
 *   ruart_adamax_step:    g' = g * norm_coef[1] (norm_coef NULL: no clipping);  m += (1 - beta1)(g' - m);  u = max(beta2 u, |g'| + eps);
 *                         p -= clr[t] * m / u  with clr[t] = lr / (1 - beta1^step_t), one DEVICE float per tensor (torch.optim.Adamax
 *                         counts the steps of every parameter separately).
 * The update's chunk list may leave out elements (embedding rows the trainer re-pins every step): they are not touched. */
int ruart_grad_norm_clip(const float* const* grads, const int* c_tensor, const int* c_start, const int* c_count, int n_chunks,
                         float max_norm, float* partial, float* norm_coef, const float* extra_sq, void* stream);
int ruart_adamax_step(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_inf, const int* c_tensor,
                      const int* c_start, const int* c_count, int n_chunks, const float* norm_coef, const float* clr, float beta1,
                      float beta2, float eps, void* stream);

/* NaN contract of the reference (assert torch.sum(torch.isnan(x)) == 0, Layers.py:169,290,430,462,467): after this
 * call every SDNet kernel ORs 1 into *flag (a device int) when it writes a NaN; the Python layer checks and clears it
 * once per step instead of one device->host sync per op.  Pass NULL to disable. */
int ruart_set_nan_flag(int* flag);

/* fp32 GEMM of the SDNet trunk on the 16-bit matrix cores (csrc/sdnet_gemm.hip): C[M,N] = A(M,K) . B(K,N) (+ bias[N]), every
 * fp32 element split into two bf16 (hi + lo) and the product taken as hi.hi + hi.lo + lo.hi with fp32 accumulation (relative
 * error ~2^-16 per product).  Replaces the library GEMMs behind torch.mm / addmm at Models/Layers.py:155, 166, 226-227 and in
 * their backward.  A element (m,k) at A[m*sam + k*sak], B element (k,n) at B[k*sbk + n*sbn]; each operand needs ONE unit
 * stride, so x.W^T, dY.W and dY^T.X all fit.  Products with a small output and a long reduction are split along K: ask
 * ruart_gemm_x3_plan (same M, N, K and operand layouts) for the workspace size first (0 when not split) and pass a workspace of
 * at least that many bytes; with a smaller (or no) workspace the product is computed unsplit - same result up to fp32
 * summation order, fewer workgroups. */
int ruart_gemm_x3_plan(int M, int N, int K, int a_k_contiguous /* sak == 1 */, int b_k_contiguous /* sbk == 1 */, int* splitk,
                       size_t* ws_bytes);
/* Optional fused variational dropout (Models/Layers.py:23-30: one mask row per batch row, shared by `rows_per_scale_row`
 * consecutive time steps; a mask element is 0 or keep_scale = 1/(1-p)); NULL = none:
 *   a_keep  (M / rows_per_scale_row, K) bytes: A(m,k) *= a_keep[m / rpm][k] ? keep_scale : 0   (needs sak == 1) - forward (x*mask) W^T
 *   b_keep  (K / rows_per_scale_row, N) bytes: B(k,n) *= b_keep[k / rpm][n] ? keep_scale : 0   (needs sbn == 1) - dW = dY^T (x*mask)
 *   c_scale (M / rows_per_scale_row, N) fp32 : C(m,n) *= c_scale[m / rpm][n]                                    - dX = (dY W) * mask
 * The operand masks travel as one byte per element and are read inside the operand loads (at most one of a_keep / b_keep per call;
 * M, K < 2^20); c_scale is the mask itself and is applied in the epilogue. */
/* Epilogue: C = act(A.B + bias) [* c_scale] + residual;  act = RUART_ACT_NONE | RUART_ACT_GELU (exact erf form), residual (M, N)
 * fp32 with row stride ldr or NULL - the encoder's dense+bias(+GELU)(+residual) steps in its split-operand precision mode. */
int ruart_gemm_x3(const float* A, long long sam, long long sak, const float* B, long long sbk, long long sbn, const float* bias,
                  const float* residual, int ldr, int act, float* C, int ldc, int M, int N, int K, float* ws, size_t ws_bytes,
                  const unsigned char* a_keep, const unsigned char* b_keep, float keep_scale, const float* c_scale, int rows_per_scale_row,
                  void* stream);
/* All weight gradients of a training step in ONE launch (+ one for the split-K sums): problem i is dW_i (M x N, row stride ldc) (+)=
 * dY_i^T . X_i with dY_i (K rows, M columns, row stride lda) and X_i (K rows, N columns, row stride ldb) as the backward pass holds
 * them - autograd's grad_weight = grad_output.t().mm(input) of the nn.Linear sites at Models/Layers.py:155, 166, 226-227.  Every
 * problem is tiled, split along K and summed exactly as ruart_gemm_x3 would do it alone (bitwise the same result); `accumulate`:
 * add to the existing dW (a module used twice).  Problems that write the SAME dW must not be in one call.  `probs` is a HOST array;
 * ws: ruart_gemm_x3_tn_grouped_ws(probs, n) bytes. */
typedef struct {
  const float* A;
  const float* B;
  float* C;
  int lda, ldb, ldc, M, N, K, accumulate;
} ruart_x3_tn_problem;
size_t ruart_gemm_x3_tn_grouped_ws(const ruart_x3_tn_problem* probs, int n);
int ruart_gemm_x3_tn_grouped(const ruart_x3_tn_problem* probs, int n, float* ws, size_t ws_bytes, void* stream);
/* ruart_gemm_x3 with ONE bf16 product (hi.hi) instead of three: the operands are rounded to bf16, the accumulation stays fp32.
 * Same arguments, layouts, epilogue and workspace rule.  For products whose result is a gradient (dX, dW) beside a 16-bit encoder. */
int ruart_gemm_x1(const float* A, long long sam, long long sak, const float* B, long long sbk, long long sbn, const float* bias,
                  const float* residual, int ldr, int act, float* C, int ldc, int M, int N, int K, float* ws, size_t ws_bytes,
                  const unsigned char* a_keep, const unsigned char* b_keep, float keep_scale, const float* c_scale, int rows_per_scale_row,
                  void* stream);

/* Weight-gradient product in plain bf16 (one MFMA product instead of three): C (M, N) fp32 = A^T . B, where A is stored (K, M)
 * with row stride `sak_rows` and B is stored (K, N) with row stride `sbk_rows` (both row-contiguous, 16-byte aligned, strides
 * multiples of 4) - i.e. dW = dY^T . X straight from dY (rows, N_out) and X (rows, K_in), no transposed copies.  fp32 operands are
 * rounded to bf16 on the way to LDS, accumulation is fp32, long reductions are split along K and summed in slice order
 * (workspace size from ruart_gemm_x3_plan(M, N, K, 0, 0, ...)). */
int ruart_gemm_bf16_tn(const float* A, long long sak_rows, const float* B, long long sbk_rows, float* C, int ldc, int M, int N, int K,
                       float* ws, size_t ws_bytes, void* stream);

/* A HIP stream restricted to ``n_cus`` compute units (the mask enables the first n_cus bits).  Optional knob for the encoder
 * pass that runs one step ahead beside the SDNet trunk (opt["bert_prefetch_cus"]): the CUs left out of the mask stay free for
 * the trunk's short kernels (240 of 256 in the fp16c schedule).  n_cus = 0 or |n_cus| >= the device's CU count creates an
 * ordinary stream; n_cus < 0 enables the LAST |n_cus| bits instead (experiments).  ruart_stream_destroy drops it early; never call that from an atexit hook (teardown order
 * of the runtime / profiler is not under the caller's control) - process exit releases the stream. */
int ruart_stream_create_cu_masked(int n_cus, void** stream_out);
/* A non-blocking HIP stream of the given priority (HIP's scale: -1 high, 0 normal, 1 low; clamped to the device's range - torch's own
 * stream pool offers high and normal only).  Returns the granted level counted from the highest (>= 0) or a negative error. */
int ruart_stream_create_priority(int priority, void** stream_out);
int ruart_stream_destroy(void* stream);

#ifdef __cplusplus
}
#endif
#endif
