/* mmx_hip.h — C ABI of libmmx_hip.so: the MI355X (gfx950) kernels behind the TTS inference hot path
 * of ishine/minimax-speech (Qwen2 AR speech-token LM -> CosyVoice2 flow-matching decoder -> DAC-VAE
 * decoder).  SURVEY.md §8(b) "C-ABI underneath".
 *
 * The reference has no FFI for this path (it is torch.nn all the way down); each entry point below
 * replaces the torch operator sequence of the cited reference lines, and is what a ctypes binding in
 * the reference would call (INTEGRATION.md shows that stub).
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is DEVICE memory unless named h_*;
 *  - the caller owns all buffers; the library keeps no pointer after a call returns and has no
 *    global mutable state -> re-entrant, one stream per caller thread
 *    (reference threading: speech/cosyvoice/cli/model.py:332-335, one llm_job thread per request);
 *  - every launch goes to the `hipStream_t` argument (NULL = legacy default stream), is asynchronous and
 *    capture-safe (no allocation, no synchronisation) so callers can record hipGraphs;
 *  - return 0 on success, MMX_EARG (-1) on an argument/shape error (nothing launched),
 *    <= -1000 for -(hipError_t) - 1000.  The Python host turns non-zero into RuntimeError, matching the
 *    reference's exception/assert convention (flow.py:453, llm.py:273);
 *  - `dtype`: MMX_F32 (parity build: fp32 storage, exact fp32 MFMA) or MMX_BF16 (bf16 storage of weights
 *    and GEMM-input activations, fp32 accumulation and fp32 residual streams).  "T" below means that type.
 *    MMX_X2 / MMX_X3 (the split build): weights are stored and streamed as bf16 exactly as in MMX_BF16, every
 *    activation is stored as fp32 ("T" = float for activations), and each product against an activation runs on the
 *    bf16 MFMA with the activation split into 2 (hi + lo: 16 significant bits) or 3 (24 bits) bf16 terms accumulated
 *    in fp32.  With bf16-representable weights the result is fp32-grade (X3) at the bf16 build's weight bytes.
 *    Entry points without a matrix product treat MMX_X2 / MMX_X3 as MMX_F32.
 *    MMX_X2W / MMX_X3W (the split build on an fp32 CHECKPOINT: mmx_gemm_win, mmx_skinny2): the weights too are carried as 2 / 3
 *    bf16 planes hi + [mid +] lo = w, and a product keeps every term (activation plane s) x (weight plane p) with
 *    s + p < 2 / 3 - three / six bf16 MFMAs per fragment pair, 16 / 24 significant bits of both operands.  Weight layout:
 *    mmx_gemm_win: the planes side by side in every row, W[n][p * (ldw / planes) + k]; mmx_skinny2: the planes' packs one
 *    after the other.  Every other argument as for MMX_X2 / MMX_X3.
 *  - activations are TIME-MAJOR: a [B,C,T] tensor of the reference is stored as rows of C channels.
 */
#ifndef MMX_HIP_H
#define MMX_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP__
typedef struct ihipStream_t* hipStream_t;
#endif

#define MMX_F32 0
#define MMX_BF16 1
#define MMX_X2 2
#define MMX_X3 3
#define MMX_X2W 0x12
#define MMX_X3W 0x13
/* fp16 planes of the LM decode step (mmx_skinny2, mmx_decode_prep, mmx_decode_attn's split output only): activations as TWO fp16
 * planes hi + lo (22 significant bits), weights fp16 and stored * 2^8 - one plane for a bf16-representable checkpoint (MMX_H2),
 * two planes hi + lo for an fp32 checkpoint (MMX_H2W); v_mfma_f32_16x16x32_f16.  |activation| < 65504, |weight| < 255. */
#define MMX_H2 4
#define MMX_H2W 0x14

/* activation codes */
#define MMX_ACT_NONE 0
#define MMX_ACT_LRELU 1
#define MMX_ACT_GELU 2   /* exact erf gelu (diffusers 0.29 GELU) */
#define MMX_ACT_SILU 3
#define MMX_ACT_MISH 4
#define MMX_ACT_TANH 5

int mmx_abi_version(void);

/* ---------------------------------------------------------------------------------------------
 * Windowed GEMM:  C[b][m][n] = sum_tap sum_c A[b][m*row_stride + tap*dil + row_off][c] * W[b][n][tap*cin + c]
 * Replaces torch Linear / Conv1d / ConvTranspose1d (+ the elementwise ops fused in its epilogue):
 *   dac-vae/model.py:107-143,237-323,342-371 (WNConv1d+LeakyReLU, Snake, WNConvTranspose1d, residual add)
 *   speech/cosyvoice/flow/decoder.py:36-85 (CausalConv1d), speech/matcha/models/components/transformer.py:243-316
 *   (to_q/k/v, to_out, GELU proj, ff out), speech/cosyvoice/transformer Linear layers.
 * Epilogue, per element v:  v += bias;  v = act(v);  v += residual;  v *= rowmask[m];
 *   out_f32[lin] = v;   out_act[lin] = T( alpha ? snake(act2(v), alpha) : act2(v) )
 *   with lin = m*ldo + n + out_off, written only when 0 <= lin < out_len (ConvTranspose1d edge clip).
 */
typedef struct MmxGemmParams {
    const void* A;          /* T, rows of lda elements */
    const void* W;          /* T, [N][ldw], K contiguous, zero padded to a multiple of 32 */
    const float* bias;      /* [bias_mod] (or [M] when bias_per_row) or NULL */
    const float* residual;  /* fp32 [M][ldr] or NULL */
    const float* rowmask;   /* fp32 [M] or NULL */
    const float* alpha;     /* Snake alpha [alpha_mod] for the out_act copy, or NULL */
    float* out_f32;         /* fp32 output or NULL */
    void* out_act;          /* T output or NULL */
    int64_t lda, ldw, ldr, ldo_f, ldo_a;
    int64_t a_bstride, w_bstride, r_bstride, rm_bstride, of_bstride, oa_bstride;   /* per batch, in elements */
    int64_t row_off, row_lo, row_hi;   /* A row = m*row_stride + tap*dil + row_off, rows outside [row_lo,row_hi) read 0 */
    int64_t out_off, out_len;
    int32_t M, N, batch;
    int32_t ntaps, cin, dil;
    int32_t bias_mod, alpha_mod, bias_per_row;
    int32_t act, act2;
    float slope;
    int32_t row_stride;                /* 1 for Linear / stride-1 convs; s for a stride-s Conv1d (DAC-VAE encoder) */
} MmxGemmParams;
int mmx_gemm_win(const MmxGemmParams* p, int dtype, hipStream_t stream);
/* The same launch with the block tile forced (0 = the library's heuristic): a tuning entry for tools/microbench.py. */
#define MMX_TILE_128x128 1
#define MMX_TILE_128x64 2
#define MMX_TILE_64x64 3
#define MMX_TILE_32x64 4
int mmx_gemm_win_tile(const MmxGemmParams* p, int dtype, int tile, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Row-wise normalisation (LayerNorm / RMSNorm) with fused tail:
 *   y = norm(x[b][t][:]) * gamma (+ beta);  y = act(y);  y = (y*rowmask + addvec[b][:]) * rowmask
 * written to out_f32 and/or out_act (T).  rms != 0 selects RMSNorm (no mean, beta ignored).
 * Replaces nn.LayerNorm (+Mish, + time-embedding add) of decoder.py:65-85 / matcha decoder.py:56-61,
 * transformer.py norm1/norm3, encoder_layer.py norm_mha/norm_ff, Qwen2 RMSNorm (prefill).
 */
int mmx_rownorm(const float* x, int64_t ldx, int64_t x_bstride, int rows, int C, int batch,
                const float* gamma, const float* beta, float eps, int rms, int act,
                const float* rowmask, int64_t rm_bstride, const float* addvec, int64_t av_bstride,
                float* out_f32, int64_t ldo_f, int64_t of_bstride,
                void* out_act, int64_t ldo_a, int64_t oa_bstride, int dtype, hipStream_t stream);

/* GroupNorm over a time-major fp32 tensor [B][T][C] (statistics per (batch, group) over C/groups channels x T),
 * then act (MMX_ACT_NONE / MMX_ACT_MISH) and * rowmask[b][t] (optional), output in T.  Replaces arch_util.py:21-38
 * GroupNorm32 inside LearnableSpeakerEncoder (llm.py:34-96) and GroupNorm + Mish of matcha Block1D (decoder.py:32-43). */
int mmx_groupnorm(const float* x, int B, int T, int C, int groups, const float* gamma, const float* beta, float eps,
                  int act, const float* rowmask, void* out, int dtype, hipStream_t stream);

/* out = act(x) * rowmask[row] on a fp32 [rows][C] tensor (fp32 and / or T output): the stand-alone Mish / SiLU / mask
 * multiplications of the matcha block classes (decoder.py:41-43,49,59; transformer.py) when they are called one by one. */
int mmx_act_rows(const float* x, int64_t rows, int C, int act, const float* rowmask, float* out_f32, void* out_act, int dtype,
                 hipStream_t stream);

/* out[i][:] = table[ids[i]][:] * scale * (rowmask ? rowmask[i] : 1)   (ids < 0 are clamped to 0,
 * flow.py:477).  Replaces nn.Embedding lookups (flow.py:477, llm.py:694-700). */
int mmx_gather_rows(const int64_t* ids, int n, const float* table, int C, float scale, const float* rowmask,
                    float* out_f32, int64_t ldo_f, void* out_act, int64_t ldo_a, int dtype, hipStream_t stream);

/* Strided copy/cast: out[b][r][c] = T(in[b*ibs + r*irs + c*ics]) for r<rows, c<cols; used for
 * [B,C,T] <-> time-major conversions at the API boundary and nearest-neighbour upsampling
 * (row r reads input row r / rep; upsample_encoder.py:60). in_dtype/out_dtype may differ. */
int mmx_copy2d(const void* in, int in_dtype, int64_t ibs, int64_t irs, int64_t ics, int rep,
               void* out, int out_dtype, int64_t obs, int64_t ors, int64_t ocs,
               int rows, int cols, int batch, hipStream_t stream);

/* Estimator input pack (decoder.py:424-432): h[b][t] = [x | mu | spks | cond] as T, + sinusoidal
 * timestep embedding (matcha decoder.py:14-29, scale 1000) emb[b][:] as T.
 * x/mu/cond are fp32 time-major [B][T][80]; batch b reads x[(b % x_mod) * x_bs ...] so the conditional and the
 * unconditional half of a CFG batch share one state tensor (x_mod = number of utterances);
 * spks [B][80]; NULL mu/spks/cond read as zero. */
int mmx_est_pack(const float* x, int64_t x_bs, int x_mod, const float* mu, const float* spks, const float* cond, int B, int T, int C,
                 void* h, int64_t ldh, int dtype, hipStream_t stream);
int mmx_sinusoidal_emb(const float* t, int B, int dim, float scale, void* out, int dtype, hipStream_t stream);

/* CFG + Euler update (flow_matching.py:118-120): x += dt * ((1+cfg)*d[0] - cfg*d[1]); n elements. */
int mmx_cfg_euler(float* x, const float* d_cond, const float* d_uncond, float cfg, float dt, int64_t n,
                  hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Attention.
 * mmx_attn_dense: softmax(q k^T * scale [+ rel-pos term] masked) v for [B][T][H*D] T-typed q/k/v
 *   (row strides ldq/ldk/ldv, head h at column h*D).  Key j is visible to query i iff
 *   j < Tk  and (keymask == NULL or keymask[b][j] != 0) and (chunk == 0 or j < (i/chunk + 1)*chunk);
 *   rows with no visible key output 0 (they are padding rows, masked by every consumer).
 *   rel-pos (pos != NULL): score = ((q+u).k + (q+v).pos[T-1-i+j]) * scale — attention.py:225-330 with
 *   rel_shift folded into the index; pos is [2T-1][H*D] T-typed, u/v fp32 [H][D].
 *   head_stride: column distance between heads inside q/k/v (0 = D; 3*D for the head-interleaved qkv of
 *   arch_util.py:58-77 QKVAttentionLegacy, whose q*s . k*s with s = D^-1/4 equals scale = D^-1/2).
 *   Replaces diffusers Attention (SDPA) of transformer.py:196-204 with the additive bias of
 *   decoder.py:441-445 / common.py:160-168, and RelPositionMultiHeadedAttention.
 * q_begin (multiple of 16; 0 = all): only queries q_begin .. T-1 are computed (keys still 0 .. T-1) — the streaming hop with
 *   cached K / V.
 * mmx_attn_flash_bf16: the MFMA flash-attention kernel for the same contract without rel-pos, bf16,
 *   D = 64, V given TRANSPOSED as vt[b][h*D + d][t] (ldvt elements per row, zero padded).
 *   klen (NULL = none): int32 [B], the number of valid keys of each batch row of a zero-padded batch whose key masks are
 *   prefixes (decoder.py:433-445 with a padding mask): keys j >= klen[b] are invisible, key tiles beyond klen[b] are not
 *   visited, query rows >= klen[b] (padding) that fill a whole workgroup are written as zeros.  Cheaper than the same
 *   mask given as keymask (which makes every tile a masked tile).
 */
int mmx_attn_dense(const void* q, int64_t ldq, int64_t q_bs, const void* k, int64_t ldk, int64_t k_bs,
                   const void* v, int64_t ldv, int64_t v_bs, void* out, int64_t ldo, int64_t o_bs,
                   int B, int H, int D, int Tq, int Tk, float scale, const float* keymask, int64_t km_bs, int chunk,
                   const void* pos, int64_t ldp, const float* pos_u, const float* pos_v,
                   int head_stride, int q_begin, int dtype, hipStream_t stream);
int mmx_attn_flash_bf16(const void* q, int64_t ldq, int64_t q_bs, const void* k, int64_t ldk, int64_t k_bs,
                        const void* vt, int64_t ldvt, int64_t vt_bs, void* out, int64_t ldo, int64_t o_bs,
                        int B, int H, int T, float scale, const float* keymask, int64_t km_bs, int chunk, int q_begin,
                        const int32_t* klen, hipStream_t stream);
/* Conformer rel-pos attention of the split build (speech/cosyvoice/transformer/attention.py:215-330 incl. rel_shift): fp32
 * q / k / v rows, pos fp32 [2T - 1][ldp] = linear_pos(ESPnet table), pos_u / pos_v fp32 [H * 64]; every product as bf16
 * hi*hi + lo*hi + hi*lo on the MFMA (as mmx_attn_flash_x); klen int32 [B] valid rows per batch member or NULL; out fp32. */
int mmx_attn_relpos_x(const float* q, int64_t ldq, int64_t q_bs, const float* k, int64_t ldk, int64_t k_bs, const float* v,
                      int64_t ldv, int64_t v_bs, const float* pos, int64_t ldp, const float* pos_u, const float* pos_v,
                      float* out, int64_t ldo, int64_t o_bs, int B, int H, int T, float scale, int chunk, const int32_t* klen,
                      hipStream_t stream);
/* mmx_attn_flash_xs: the split build's attention on operands the PRODUCER has already split (MmxEstNext, MMX_X2 with
 * vt_out): qk bf16 [B][T][ldqk >= 2048] = [hi Q | hi K | lo Q | lo K], vt bf16 [B][2][512][ldvt], out fp32 [B][T][ldo].
 * form: 0 = workgroup shape chosen per launch (fastest alone); 1 = the shape that leaves LDS and registers for a co-resident
 * workgroup of another stream (launches beside the LM decode loop); 2 / 3 = 256-query / 4-wave 64-query workgroups (measurements).
 * The forms agree to fp32 rounding (different query-tile heights accumulate the online softmax over the same key tiles).
 * mmx_attn_flash_x: the same contract on fp32 operands:
 * q / k / v / out fp32, V ROW-major (v[b][t][h*D + d], ldv elements per
 * row — the QKV projection's own output, no transposed copy), both operands of Q K^T and P V split into bf16 hi + lo
 * (3 MFMAs per product).  Replaces the same reference lines as mmx_attn_flash_bf16. */
int mmx_attn_flash_x(const float* q, int64_t ldq, int64_t q_bs, const float* k, int64_t ldk, int64_t k_bs,
                     const float* v, int64_t ldv, int64_t v_bs, float* out, int64_t ldo, int64_t o_bs,
                     int B, int H, int T, float scale, const float* keymask, int64_t km_bs, int chunk,
                     int q_begin, const int32_t* klen, hipStream_t stream);
int mmx_attn_flash_xs(const void* qk, int64_t ldqk, int64_t qk_bs, const void* vt, int64_t ldvt, int64_t vt_bs,
                      float* out, int64_t ldo, int64_t o_bs, int B, int H, int T, float scale, const float* keymask,
                      int64_t km_bs, int chunk, int q_begin, const int32_t* klen, int form, hipStream_t stream);
/* Conformer rel-pos attention on the MFMA (speech/cosyvoice/transformer/attention.py:215-330), bf16 build:
 *   score(i, j) = ((q_i + pos_u) . k_j + (q_i + pos_v) . pos[T - 1 - i + j]) * scale      (rel_shift folded into the index)
 * q, k: bf16 rows (head h at column h * 64), vt: V TRANSPOSED [B][H * 64][ldvt] (zero padded to ldvt >= round_up(T, 8)
 * columns), pos: bf16 [2T - 1][ldp] = linear_pos(pos_emb) (head h at column h * 64), pos_u / pos_v fp32 [H * 64];
 * chunk > 0: query i sees keys j < (i / chunk + 1) * chunk (streaming); klen (optional, int32 [B]): valid rows per member of
 * a zero-padded batch (keys beyond are masked; `pos` is the table of the padded length T).  Replaces mmx_attn_dense's VALU path for the
 * encoder's 10 layers in the bf16 build; the fp32 / split builds keep mmx_attn_dense. */
int mmx_attn_relpos_bf16(const void* q, int64_t ldq, int64_t q_bs, const void* k, int64_t ldk, int64_t k_bs,
                         const void* vt, int64_t ldvt, int64_t vt_bs, const void* pos, int64_t ldp,
                         const float* pos_u, const float* pos_v, void* out, int64_t ldo, int64_t o_bs,
                         int B, int H, int T, float scale, int chunk, const int32_t* klen, hipStream_t stream);

/* The same contract (bf16 tensors in HBM) with Q, K, V^T and P quantised to OCP fp8 e4m3 inside the kernel and both
 * products on the fp8 MFMA (BASELINE config 5).  Accuracy: the bound stated in tests/test_gpu_kernels.py (<= 7 % of the output RMS). */
int mmx_attn_flash_fp8(const void* q, int64_t ldq, int64_t q_bs, const void* k, int64_t ldk, int64_t k_bs,
                       const void* vt, int64_t ldvt, int64_t vt_bs, void* out, int64_t ldo, int64_t o_bs,
                       int B, int H, int T, float scale, const float* keymask, int64_t km_bs, int chunk, int q_begin,
                       const int32_t* klen, hipStream_t stream);

/* DAC-VAE encoder head: Conv1d(1 -> C, k) + LeakyReLU (dac-vae/model.py:208 with :509-514) on a mono waveform
 * x fp32 [B][T]; w fp32 [C][k]; out_f32 [B][T][C] (optional) and out_act T [B][T][C] = snake(v, alpha) (alpha optional). */
int mmx_conv_cin1(const float* x, int64_t x_bs, int T, int C, int k, const float* w, const float* bias, float slope,
                  const float* alpha, float* out_f32, void* out_act, int batch, int dtype, hipStream_t stream);
/* VAE head (dac-vae/model.py:476-481): ml fp32 [rows][2D] = (m | logs); logs clamped to [-14, 14];
 * z = m + noise * exp(logs).  All outputs fp32 [rows][D]. */
int mmx_vae_sample(const float* ml, const float* noise, int64_t rows, int D, float* z, float* m, float* logs,
                   hipStream_t stream);
/* x[row][:] = 0 where rowmask[row] == 0, in place; x is T [rows][C] (C a multiple of 16 bytes).  The rows beyond a member's length
 * of a zero-padded batch behind a ConvTranspose1d (dac-vae/model.py:252-284), whose GEMM rows straddle that boundary. */
int mmx_mask_rows(void* x, int64_t rows, int C, const float* rowmask, int dtype, hipStream_t stream);
/* Cache prefetch: reads the byte ranges [p_i, p_i + n_i) (16-byte aligned, n_i multiples of 16; NULL = none) with default-policy
 * loads and keeps nothing, so the lines are left in L2 / the Infinity Cache for a later launch (the next LM layer's weights while
 * the current layer computes: a side stream of the captured decode step).  sink: 4 writable bytes (never written in practice). */
int mmx_prefetch4(const void* p0, int64_t n0, const void* p1, int64_t n1, const void* p2, int64_t n2, const void* p3, int64_t n3,
                  void* sink, int workgroups, hipStream_t stream);
/* Speed change (speech/cosyvoice/cli/model.py:312-314): x fp32 [rows][T] -> out fp32 [rows][T2], linear interpolation in time
 * with torch's F.interpolate(mode="linear", align_corners=False) sample positions. */
int mmx_resample_linear(const float* x, int64_t rows, int T, int T2, float* out, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * DAC tail: Conv1d(C -> 1, k) + LeakyReLU(0.1) + tanh (dac-vae/model.py:364-370,509-514).
 * act: T [B][T][C] (Snake already applied by the producer); w fp32 [k][C]; out fp32 [B][T]. */
int mmx_conv_cout1_tanh(const void* act, int64_t a_bs, int T, int C, int k, const float* w, const float* bias,
                        float slope, int use_tanh, float* out, int64_t o_bs, int batch, int dtype, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Autoregressive LM decode step (speech/cosyvoice/llm/llm.py:745-760 + HF Qwen2 layer).
 *
 * mmx_skinny_gemm: out[b][n] = (rs ? rstd[b] : 1) * sum_k x[b][k] * Wp[n][k] (+bias) for B <= 64 rows:
 *   weights are streamed once, straight to registers, in MFMA-fragment order (pack with mmx_pack_skinny).
 *   x is fp32 [B][ldx] (x_dtype MMX_F32) or T.  rs != 0: rstd[b] = rsqrt(mean_k x^2 + eps) is computed
 *   in-kernel (RMSNorm with its weight pre-folded into Wp).  epi: 0 store fp32 (+bias);
 *   1 SwiGLU: Wp rows are [gate tile | up tile] pairs, out_act[b][n] = T(silu(g)*u);
 *   2 residual: out_f32[b][n] += acc (in place).
 *   flags: MMX_X_PACKED — x (type T) is in MFMA A-fragment order xp[m][kb][lane][E] (lane = g*16 + l16 holds
 *   x[m*16 + l16][kb*KB + g*E + j]; E = 8 bf16 / 4 fp32, KB = 4E; ceil(B/16)*16 rows allocated, ldx ignored);
 *   MMX_OUT_PACKED — out_act is written in that order for a consumer whose K is this N (N % 32 == 0, ldo_a ignored).
 *   Pays at batch > 8, where every workgroup re-reads the whole activation matrix from L2.
 *   dtype MMX_X2 / MMX_X3 (split build): x fp32 row-major (flags 0), Wp the bf16 pack made WITHOUT kscale, kgamma [K] fp32
 *   (or NULL) = the RMSNorm gain applied to x before the split, results to out_f32 only (epi 1 writes silu(g)*u there).
 *   MMX_BF16 with fp32 x (flags 0) also takes kgamma (epi 0 / 1): the gain is applied to x before it is rounded to bf16, Wp
 *   packed without kscale (the form the bf16 build's prompt chunks use, its decode step running on mmx_skinny2).
 */
#define MMX_X_PACKED 1
#define MMX_OUT_PACKED 2
int mmx_pack_skinny(const void* w, int64_t ldw, int N, int K, const float* kscale, int interleave_half,
                    void* wp, int dtype, hipStream_t stream);
int mmx_skinny_gemm(const void* x, int x_dtype, int64_t ldx, int B, int K, int N, const void* wp,
                    const float* bias, int rs, float eps, int epi, float* out_f32, int64_t ldo_f,
                    void* out_act, int64_t ldo_a, int dtype, int flags, const float* kgamma, hipStream_t stream);

/* Decode-step projections, B <= 32 sequences, on SPLIT-PLANE activations (dtype MMX_X3: 3 planes as below; dtype MMX_BF16:
 * ONE plane = bf16(x), which is the packed A-fragment order of MMX_X_PACKED):
 *   an activation x [B][K] (already multiplied by the RMSNorm gain of its consumer) is stored as 3 bf16 planes
 *   hi + mid + lo = x, each in MFMA A-fragment order:
 *     xs[plane s][m][kb][lane = g*16 + l16][j] = term s of x[m*16 + l16][kb*32 + g*8 + j]      (ceil(B/16) row tiles m)
 *   The producer of an activation splits it once; consumers load fragments straight into MFMA operands.
 *   Sum-of-squares tables `ssq` [32 rows][64 tile slots] fp32 carry the RMSNorm statistic the same way: a producer of the
 *   residual stream writes, per 16-column output tile, each row's sum of squares over that tile's columns; the consumer
 *   adds a row's slots (unused slots must be zero: allocate zeroed).
 *
 * mmx_decode_prep: x fp32 [B][K] (K <= 1024) -> h (copy, optional), xs = planes of x * gamma, ssq of x.  The head of a
 *   decode step: the sampler's next input embedding becomes the residual stream and the first projection's operand.
 * mmx_skinny2:  acc[b][n] = rstd[b] * sum_k xs[b][k] * Wp[n][k],  rstd from ssq_in (NULL: 1), Wp the bf16 pack of
 *   mmx_pack_skinny (no kscale);
 *   epi 0: out[b][n] = acc + bias;
 *   epi 1: SwiGLU of [gate tile | up tile] pairs, written as planes xs_out of width N (the down projection's operand);
 *   epi 2: out[b][n] += acc (+ bias) in place (the residual stream); if xs_out: planes of out * gamma_next; if ssq_out:
 *          the per-tile sums of squares of out.
 *   tiles_per_wg (1 | 2): 16-column tiles a workgroup produces from ONE pass over xs (8 waves = 8 k slices).
 *   ksplit (1, or > 1 with epi 2): K is also cut into `ksplit` slices across workgroups; partial tiles go through
 *   `part` (>= ksplit * ceil(N/16) * ceil(B/16)*4 * 64 floats) and are summed in slice order by the workgroup that
 *   takes the last ticket of its tile (`tickets`: ceil(N/16) int32, zero before the first launch; left zero).  Launches
 *   that share part / tickets must be ordered on one stream.
 *   Replace the q/k/v, o, gate/up (+SiLU*up), down projections and the llm_decoder head of one decode step
 *   (speech/cosyvoice/llm/llm.py:359-371,749; HF Qwen2 MLP / attention projections / RMSNorm). */
int mmx_decode_prep(const float* x, int64_t ldx, int B, int K, const float* gamma, float* h, int64_t ldh, void* xs,
                    float* ssq, int dtype, hipStream_t stream);
int mmx_skinny2(const void* xs, int B, int K, int N, const void* wp, const float* bias, const float* ssq_in, float eps,
                int epi, float* out, int64_t ldo, void* xs_out, const float* gamma_next, float* ssq_out,
                int tiles_per_wg, int ksplit, float* part, int64_t part_floats, int32_t* tickets, int dtype,
                hipStream_t stream);

/* RoPE (HF rotate_half; inv_freq[D/2] fp32 = 1/theta^(2i/D) as HF computes it) on q/k of
 * qkv[b][t][: (Hq+2Hkv)*D] at position pos[b] + t, K/V appended to the paged cache, q written as T.  Cache layout:
 * kc/vc [n_pages][Hkv][page][D] T, block_table [B][max_pages] int32.  rows = tokens per sequence. */
int mmx_rope_kv_store(const float* qkv, int64_t ldqkv, int64_t qkv_bs, int B, int rows, int Hq, int Hkv, int D,
                      const float* inv_freq, const int32_t* pos, void* q_out, int64_t ldq, int64_t q_bs,
                      void* kc, void* vc, const int32_t* block_table, int max_pages, int page,
                      int dtype, hipStream_t stream);
/* Causal GQA attention over the paged cache: query row t of sequence b sees keys [0, pos[b] + t]. */
int mmx_paged_attn(const void* q, int64_t ldq, int64_t q_bs, int B, int rows, int Hq, int Hkv, int D, float scale,
                   const int32_t* pos, const void* kc, const void* vc, const int32_t* block_table, int max_pages,
                   int page, void* out, int64_t ldo, int64_t o_bs, int dtype, hipStream_t stream);
/* Single-token decode: RoPE(q), RoPE(k_new), KV append and causal GQA attention in ONE launch per layer
 * (same arithmetic as mmx_rope_kv_store + mmx_paged_attn with rows = 1).  qkv fp32 [B][ldqkv], out T [B][ldo].
 * rope_tab (optional, fp32 [max_pos][D] = cos | sin per position, as HF's rotary embedding computes them) replaces
 * the in-kernel cosf/sinf of pos * inv_freq.  out_packed: bit 0 = write out in the MMX_OUT_PACKED order (K = Hq*D);
 * bit 1 = use the one-workgroup-per-query-head kernel even where the GQA-shared one applies (bf16, page = 16, Hq = 7 Hkv:
 * one workgroup per kv head serves its 7 query heads, Q K^T on the MFMA with the queries split into bf16 hi + lo). */
/* out_packed bits: 1 = output in the packed A-fragment order of T; 2 = force the per-head kernel; 4 = output as SPLIT PLANES
 * (see mmx_skinny2; dtype MMX_F32 / the split builds only); 8 = output as TWO fp16 planes (MMX_H2, same conditions);
 * 16 = the per-head kernel with ONE query head per workgroup at every batch size (default: from B * Hq > 256 on, two heads share a
 * workgroup's K / V reads; same results bit for bit). */
int mmx_decode_attn(const float* qkv, int64_t ldqkv, int B, int Hq, int Hkv, int D, const float* inv_freq,
                    const float* rope_tab, const int32_t* pos, void* kc, void* vc, const int32_t* block_table, int max_pages, int page,
                    float scale, void* out, int64_t ldo, int dtype, int out_packed, hipStream_t stream);
/* SwiGLU for prefill: out = T(silu(gu[:, :I]) * gu[:, I:2I]) */
int mmx_swiglu(const float* gu, int64_t ldgu, int rows, int I, void* out, int64_t ldo, int dtype, hipStream_t stream);

/* Sampler = log_softmax + ras_sampling + sampling_ids + the loop bookkeeping of inference_wrapper
 * (llm.py:259-274,751-760; utils/common.py:111-139), one workgroup per sequence, all state on device:
 *   state[field * B + b], fields {0 pos, 1 step, 2 n_out, 3 finished, 4 min_len, 5 max_len, 6 seq_id, 7 error}
 *   (int32, field-major so that &state[0] is the pos[] array the attention kernels read)
 * Draws follow oracle/philox.py (Philox4x32-10 keyed by seed; exponential race == torch.multinomial).
 * On an accepted token: appended to out_tokens[b][n_out++], next_x[b][:] = speech_emb[token][:].
 * EOS (== eos_id) sets finished; ids > eos_id leave next_x unchanged (llm.py:755-756).
 * Every call advances state.step and state.pos (+1) of unfinished sequences.
 * forced != NULL: teacher forcing, forced[b*max_out + step] replaces the accepted token (the sampled id is
 * still recorded in sampled[b][step]).  logp_out (optional) receives log_softmax(logits). */
int mmx_sample_step(const float* logits, int64_t ldl, int V, int B, int eos_id, int top_k, float top_p,
                    int win_size, float tau_r, uint64_t seed, int32_t* state, int32_t* out_tokens, int max_out,
                    int32_t* sampled, const int32_t* forced, const float* speech_emb, int E, float* next_x,
                    int64_t ldx, float* logp_out, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Row-tile fused kernels of the CFM estimator (csrc/fused.hip).  One workgroup takes a tile of `bm` frames of one
 * sequence through every row-local op between two attentions, so a transformer block is 2 launches (this + attention).
 * Dimensions are those of speech/config.yaml:104-116: C = 256 channels, 8 heads x 64, FF 1024.
 * Weights are in MFMA fragment order (mmx_pack_skinny of the [N][K] matrix; Conv1d k3: K = tap*Cin + c).
 *
 * MmxEstNext (optional, wqkv != NULL): LayerNorm (n1g, n1b) of the result followed by the Q/K/V projection of the NEXT
 *   transformer block (matcha transformer.py:256-285: norm1 -> attn1.to_q/to_k/to_v, no bias).
 *   MMX_BF16: q_out = [B][T][ldq >= 1024] (Q | K) and vt_out = V transposed [B][512][ldvt] (frames >= T written as 0);
 *   MMX_F32 : q_out = [B][T][ldq >= 1536] (Q | K | V), vt_out unused.
 *   MMX_X2  : vt_out == NULL: fp32 q_out = [B][T][ldq >= 1536] (Q | K | V);  vt_out != NULL: the operands of
 *             mmx_attn_flash_xs, split here once: q_out bf16 [B][T][ldq >= 2048] = [hi Q | hi K | lo Q | lo K] and
 *             vt_out bf16 [B][2 planes][512][ldvt] (V transposed, hi then lo; vt_bs >= 2 * 512 * ldvt).
 * mmx_est_tail: x += attn1.to_out(ao) ; x += ff(norm3(x))  (transformer.py:286-313), x fp32 [B][T][256] in place;
 *   rowmask (optional) multiplies the result; act_out (optional) receives T(x) with row stride act_ld.
 * mmx_est_resnet: CausalResnetBlock1D (flow/decoder.py:65-85 + matcha decoder.py:56-61):
 *   h = mish(LN(conv3(a_in) + b1)) * m;  h = (h + tv[b]) * m;  h = mish(LN(conv3(h) + b2)) * m;
 *   x = h + conv1(a_in) + br.  a_in T [B][T][lda] (cin columns, already masked); tv = mlp(mish(t_emb)) fp32 [B][256].
 * bm: 16 / 32 / 64 (bf16), 16 / 32 (fp32 tail), 16 (fp32 resnet).
 * t_begin > 0 (streaming with cached state, config 5): only frames t_begin .. T-1 are computed; every buffer is indexed by
 * absolute frame with the batch strides given, so the rows of earlier hops (the conv halo, the K / V rows) are found
 * where those hops left them.
 */
typedef struct MmxEstNext {
    const void* wqkv;
    const float* n1g;
    const float* n1b;
    void* q_out;
    void* vt_out;
    int64_t q_bs, vt_bs;
    int32_t ldq, ldvt;
} MmxEstNext;
typedef struct MmxEstTailParams {
    const void* ao;         /* T [B][T][ldao], 512 columns: attention output */
    float* x;               /* fp32 [B][T][256] residual stream, in place */
    const void* wo;         /* packed [256][512] */
    const void* w1;         /* packed [1024][256] */
    const void* w2;         /* packed [256][1024] */
    const float* bo;
    const float* b1;
    const float* b2;
    const float* n3g;
    const float* n3b;
    const float* rowmask;   /* fp32 [B][T] or NULL */
    void* act_out;          /* T or NULL */
    int64_t ao_bs, x_bs, rm_bs, act_bs;
    int32_t ldao, act_ld, B, T;
    int32_t t_begin;        /* first frame to process (multiple of 16; 0 = all): frames before it are only read */
    float eps;
    MmxEstNext next;
} MmxEstTailParams;
typedef struct MmxEstResnetParams {
    const void* a_in;
    float* x;               /* fp32 [B][T][256] out */
    const void* w1;         /* packed [256][3*cin] */
    const void* w2;         /* packed [256][768] */
    const void* wr;         /* packed [256][cin] */
    const float* b1;
    const float* g1;
    const float* be1;
    const float* b2;
    const float* g2;
    const float* be2;
    const float* br;
    const float* tv;        /* fp32 [B][...], row stride tv_bs, this block's 256 values */
    const float* rowmask;
    int64_t a_bs, x_bs, tv_bs, rm_bs;
    int32_t lda, cin, B, T;
    int32_t t_begin;        /* as in MmxEstTailParams; the causal halo rows t_begin-18 .. are read from a_in */
    float eps;
    MmxEstNext next;
} MmxEstResnetParams;
/* cfg = pf + 16 * waves + 512 * narrow (0 = the library's defaults; mmx_est_tail only: narrow = 8 waves with 32-column passes,
 * the bf16 default for 64 / 32 rows and the form of the split build's 64-row tile): pf = k-steps of weight fragments a wave keeps in flight (2 / 4 / 8);
 * bm = rows per workgroup: 64 / 32 / 16 (MMX_BF16, MMX_X2, MMX_X2W), 32 / 16 (MMX_F32).  The split build's 64-row tile takes the
 * attention rows in two K halves and the FF intermediate in 256-wide chunks (two bf16 planes of 64 rows fit LDS that way) and is
 * bit-identical to its 32-row tile;
 * waves = 4 (one wave per SIMD, 64-column slices) or 8 (two per SIMD, 32-column slices: one wave's epilogue runs under the
 * other's MFMA stage; bf16 only). */
int mmx_est_tail(const MmxEstTailParams* p, int dtype, int bm, int cfg, hipStream_t stream);
int mmx_est_resnet(const MmxEstResnetParams* p, int dtype, int bm, int cfg, hipStream_t stream);
/* ---------------------------------------------------------------------------------------------
 * DAC-VAE ResidualUnit as one kernel (dac-vae/model.py:107-143; :509-514: LeakyReLU(slope) after every Conv1d):
 *     x_out = x + lrelu(conv1(snake_a2(lrelu(conv7, dilation dil (snake_a0(x))))))      [act_out = snake_alpha_next(x_out)]
 * for the narrow decoder stages, C = 48 / 96 / 192; dtype MMX_BF16 or MMX_X2 (the fp32 build runs the unit as two
 * mmx_gemm_win launches).  x / x_out: fp32 [B][T][C] residual stream (batch stride x_bs elements; x_out must not alias x:
 * neighbouring tiles read each other's halo rows); act_out (optional): the next layer's input activation, bf16 (MMX_BF16) or
 * fp32 (MMX_X2), same strides.  w7 / w1: mmx_pack_skinny packs of the weight matrices [C][7 * CP] (tap-major, each tap's
 * C input channels zero-padded to CP = 32 * ceil(C / 32)) and [C][CP].  lens (optional, int32 [B]): rows >= lens[b] of batch
 * member b are conv padding - read as zero, written as zero (utterances of different lengths decoded in one batch).
 * bm: rows per workgroup, 0 = the library's default for (C, dtype). */
typedef struct {
    const float* x;
    float* x_out;
    void* act_out;
    const void* w7;
    const void* w1;
    const float* b7;
    const float* b1;
    const float* a0;
    const float* a2;
    const float* alpha_next;
    const int32_t* lens;
    int64_t x_bs;
    int32_t B, T, C, dil;
    float slope;
} MmxDacRuParams;
int mmx_dac_ru(const MmxDacRuParams* p, int dtype, int bm, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif
