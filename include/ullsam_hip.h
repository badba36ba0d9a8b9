/* libullsam_hip.so -- C ABI of the MI355X (gfx950) uLLSAM hot path.
 *
 * The reference (ieellee/uLLSAM) has NO native/FFI layer: its hot path is PyTorch ATen calls inside nn.Module.forward
 * (SURVEY.md section 8(b)).  The drop-in boundary is therefore the Python module surface (ullsam_amd/modeling/*, same
 * names / signatures / state_dict keys as the reference); this header is the ABI those modules bind with ctypes --
 * each entry point names the reference call site (file:line under /root/reference) whose ATen ops it replaces.
 *
 * Conventions
 *   - plain pointers are DEVICE pointers (torch tensor.data_ptr()); no torch types cross this boundary
 *   - dtype: 0 = float32 (parity mode, exact-fp32 MFMA), 1 = bfloat16 (throughput mode, fp32 accumulate)
 *   - `stream` is a hipStream_t (torch.cuda.current_stream().cuda_stream); kernels are stream-ordered, never
 *     allocate, never synchronise, never throw
 *   - return 0 on success, < 0 on error; ullsam_last_error_string() describes the last error of the calling thread
 */
#ifndef ULLSAM_HIP_H
#define ULLSAM_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

#define ULLSAM_DT_F32 0
#define ULLSAM_DT_BF16 1
#define ULLSAM_ACT_NONE 0
#define ULLSAM_ACT_GELU 1   /* exact erf GELU (nn.GELU default) */
#define ULLSAM_ACT_RELU 2
#define ULLSAM_ACT_SWIGLU 3 /* paired 64-column gate/up blocks -> silu(gate)*up */

/* Bumped whenever an entry point is added or a signature changes.  The Python binding refuses a library that reports another
   version (a stale libullsam_hip.so would otherwise receive shifted arguments, e.g. a row count where the stream is expected). */
#define ULLSAM_ABI_VERSION 11

const char* ullsam_last_error_string(void);
int ullsam_abi_version(void); /* == ULLSAM_ABI_VERSION of the header the library was built from */
int ullsam_device_count(void);
/* GEMM kernel selection for A/B measurements and the kernel tests: 0 = auto (by shape), 1 = 128x128 tile, 3 = 256x256 two-buffer kernel,
   6 / 8 / 9 = ring kernel with 256x256 / 256x320 / 272x256 tiles; +64 = no split-K tail; +32768 = stamped diagnostic launch of the ring kernel (+65536 as well: of its persistent form). */
int ullsam_set_gemm_variant(int variant);
/* measurement knob, not part of the reference's interface: key 0 = tile rows per raster group of the 256-row-tile GEMM kernels (default 4);
   key 1 = ring tile shapes the automatic dispatch may use (bit 0 256x256, bit 1 256x320, bit 2 272x256; default 7);
   key 2 = how ring launches of MORE than one round of tiles run (csrc/gemm_ring8p.h; every value gives bit-equal outputs except the ablations): 2 (default) = persistent grid,
   identical trips, both wave groups' epilogues together; 0 = one tile per workgroup (the kernel of rounds 2 - 5); 1 / 4 = the first persistent schedule / one barrier per stage;
   5 - 7 = timing-only ablations of the 272x256 loop (no LDS-DMA requests / no fragment reads / neither: WRONG results);
   key 3 = split-K on the ring kernel for launches of <= 128 tiles whose K cuts into ranges of >= 1280 that fill >= 224 workgroups: 1 (default) = where the caller allows it
   (ullsam_gemm's act | 256), 0 = never, 2 = wherever it fits */
int ullsam_set_gemm_tuning(int key, int value);
/* Attention kernel selection for A/B measurements and the kernel tests: 0 = production; 1 / 2 = windowed attention on the tiled kernel (7-wave /
   4-wave workgroups) instead of the whole-window kernel; 3..8 = start stagger of the whole-window kernel's second resident workgroup; 9 = global
   attention as two 4-wave workgroups; 11 = causal prefill on the tiled kernel instead of the LDS-DMA kernel (the bit-equality test's reference). */
int ullsam_set_attn_variant(int variant);
/* diagnostic hook, not part of the reference's interface: a device buffer (>= 8 x 8 bytes per wave of the launch) that the stamped build of the
   causal prefill attention kernel fills with s_memtime sums (request issue / compute / wait + barrier per wave); NULL (default) = the product kernel */
int ullsam_set_attn_debug(void* stamps);

/* C[M,N] = act(A[M,K] . W[N,K]^T + bias) + residual.  Replaces every nn.Linear / 1x1 conv / stride==kernel conv:
 * image_encoder.py:227,238,387-395,88-104; common.py:21-26; modeling_internvl_sam.py:88-100;
 * modeling_internlm2.py:261-264,359,421,1081; transformer.py:220-227 (image side); mask_decoder.py:53-59.
 * A, W in `dtype`; C float when out_f32 else `dtype`; bias/residual fp32 (nullable); residual row = m %% res_row_mod
 * when res_row_mod > 0 (pos_embed broadcast, image_encoder.py:107-109).  K %% (128/elem_size) == 0.
 * workspace (optional, caller-owned device scratch, >= 32 MiB useful, 128 MiB for every split-K form): lets launches whose last wave of tiles is mostly empty
 * split those tiles along K (fp32 partials in the workspace + a reduce kernel), and launches of few tiles (e.g. 1081 x 4096 outputs) run as up to 8 K ranges
 * side by side (S fp32 planes of the output in the workspace, added in order); null disables both.
 * act: 0 none, 1 GELU (erf), 2 ReLU, 3 SwiGLU pair; | 256 = ULLSAM_ACT_SPLITK_OK: the caller accepts the K-ranges form just described.  Every one-launch kernel adds the
 * 32-deep k steps of an output in sequence, so an image gets the same bits alone and inside a batch; K ranges summed apart do not -- the training step's frozen
 * linears set the flag (one image per step, train_joint_v2.py), the inference modules never do. */
#define ULLSAM_ACT_SPLITK_OK 256
int ullsam_gemm(int dtype, const void* A, long lda, const void* W, long ldw, void* C, long ldc, int out_f32,
                const float* bias, const float* residual, long ldr, int res_row_mod, int act, int M, int N, int K,
                void* workspace, long ws_bytes, void* stream);

/* InternLM2Attention's wqkv projection with `rearrange 'b q (h gs d)'`, apply_rotary_pos_emb and the KV-cache append in the GEMM
 * epilogue (modeling_internlm2.py:359-388, 233-247): q_out T [B*S, KVH*G*128] rotated, k_cache (rotated) / v_cache T [B, KVH, cap, 128]
 * rows cache_pos0..cache_pos0+S-1; pos int32 [B*S] rows of the fp32 cos / sin tables [tab_rows, 128] (clamped).  head_dim = 128. */
int ullsam_gemm_qkv_rope(int dtype, const void* A, long lda, const void* W, long ldw, const float* bias, int B, int S, int K, int KVH,
                         int G, const int* pos, const float* cos_tab, const float* sin_tab, int tab_rows, void* q_out, void* k_cache,
                         void* v_cache, int cap, int cache_pos0, void* workspace, long ws_bytes, void* stream);

/* Decode step (InternVLSAMModel.generate's token loop, modeling_internlm2.py:1112-1149; M <= 4 rows, bf16 weights): the RMSNorm that
 * precedes a layer's wqkv / w13 (modeling_internlm2.py:75-89, 598-618) folded into the GEMM that consumes it:
 * C = act(bf16(x * rsqrt(mean(x^2) + eps) * norm_w) @ W^T + bias) (+ residual), x fp32 [M, K <= 4096], K % 2048 == 0; act as ullsam_gemm. */
int ullsam_gemm_rmsnorm(const float* x, long ldx, const float* norm_w, float eps, const void* W, long ldw, void* C, long ldc, int out_f32,
                        const float* bias, const float* residual, long ldr, int act, int M, int N, int K, void* stream);
/* ... and the decode step's wqkv: ullsam_gemm_qkv_rope with S = 1 and B <= 4, the activations either bf16 `a` [B, K] (norm_w NULL) or the
 * RMSNorm of fp32 x [B, K] as above. */
int ullsam_decode_qkv_rope(const void* a, const float* x, long ldx, const float* norm_w, float eps, const void* W, long ldw, const float* bias,
                           int B, int K, int KVH, int G, const int* pos, const float* cos_tab, const float* sin_tab, int tab_rows, void* q_out,
                           void* k_cache, void* v_cache, int cap, int cache_pos0, void* stream);

/* ---- Training slice: backward kernels of the segmentation branch of train_joint_v2.py:1026-1100, everything
 * downstream of the LLM's last hidden state (mlp2, prompt encoder, mask decoder, upsample, BCE + Dice).  fp32; torch.autograd.Function
 * wrappers in ullsam_amd/training.py.  Buffers that receive atomics (d* of colsum / ln_bwd / attn_bwd / resize_bwd / index_add_rows,
 * `sums`) are zeroed by the caller. */
/* C[b](m,n) = (accumulate ? C : 0) + sum_k A[b](m,k) B[b](k,n), explicit element strides (nn.Linear backward: dX = dY W, dW = dY^T X;
 * the hypernetwork product of mask_decoder.py:146-147 and its gradients) */
int ullsam_train_matmul(const float* A, const float* B, float* C, int M, int N, int K, int batch, long a_b, long a_m, long a_k, long b_b,
                        long b_k, long b_n, long c_b, long c_m, long c_n, int accumulate, void* stream);
/* the product with both fp32 operands rounded to bf16 on load and summed on the bf16 MFMA (fp32 accumulation and result): torch.autocast(bfloat16)'s matmul,
 * i.e. the attention products of the reference's trainer on its bf16 model (train_joint_v2.py:1665) */
int ullsam_train_matmul_bf16(const float* A, const float* B, float* C, int M, int N, int K, int batch, long a_b, long a_m, long a_k, long b_b,
                             long b_k, long b_n, long c_b, long c_m, long c_n, int accumulate, void* stream);
/* the product over (outer, head) pairs whose operands sit inside [rows, heads x hd] activations: entry (o, h) of operand X starts at o x_o + (h / x_hdiv) x_h elements
 * (x_hdiv = query heads per KV head for k / v, modeling_internlm2.py:250-259 repeat_kv); bf16 != 0: operands rounded to bf16 as ullsam_train_matmul_bf16.  The attention
 * products of a training step read q / k / v / dO and write out / dq / dk / dv in place instead of through head-major copies.  tri: the causal structure of a square
 * attention (0 none): 1 = C[query][key], 128 x 128 tiles wholly behind the diagonal are not formed (ullsam_train_attn_rows does not read masked entries and writes zeros there);
 * 2 = the sum runs over queries, m is the key: it starts at the tile's first key; 3 = the sum runs over keys, m is the query: it stops after the tile's last query */
int ullsam_train_matmul_heads(const float* A, const float* B, float* C, int M, int N, int K, int outer, int heads, long a_o, long a_h, int a_hdiv, long a_m, long a_k,
                              long b_o, long b_h, int b_hdiv, long b_k, long b_n, long c_o, long c_h, long c_m, long c_n, int accumulate, int bf16, int tri, void* stream);
/* the same product with its k range cut into `ksplit` pieces run by separate workgroups and added IN ORDER by a second kernel (few output tiles,
 * long sums: rel-pos table gradients, the hypernetwork gradient over 65536 pixels); partial: ksplit * batch * M * N floats of scratch */
int ullsam_train_matmul_splitk(const float* A, const float* B, float* C, int M, int N, int K, int batch, long a_b, long a_m, long a_k, long b_b,
                               long b_k, long b_n, long c_b, long c_m, long c_n, int accumulate, int ksplit, float* partial, void* stream);
/* the product above runs on the matrix pipe (exact fp32 MFMA, 128 x 128 tiles) from M >= 64, N >= 48, K >= 16; 0 keeps every launch on the
 * one-output-per-thread kernel (tests compare the two); returns the previous setting */
int ullsam_train_set_matmul_mfma(int on);
/* the bf16 product (ullsam_train_matmul_bf16 / _heads) fetches its k-fastest operands 16 bytes per lane where their strides and alignment allow (1, default) or element by
 * element as in round 4 (0); the LDS image and the MFMA order are the same: equal bits (tests compare); returns the previous setting */
int ullsam_train_set_matmul_vec(int on);
/* out[c] += sum_r x[r*ld + c] (bias gradients; gradients of parameters broadcast over the batch) */
int ullsam_train_colsum(const float* x, float* out, long rows, int cols, long ld, float* partial, void* stream);
/* (row blocks write partial sums that are added in order -- no atomics; partial: min(64, ceil(rows / 64)) * cols floats, may be NULL for rows <= 64) */
/* nn.LayerNorm / LayerNorm2d backward on rows of D (w NULL: no affine, prompt_encoder.py:141-144); dw / db may be NULL */
int ullsam_train_ln_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, long rows, int D, float eps, float* ws, void* stream);
/* (ws: 2 * rows + 2 * min(64, ceil(rows / 64)) * D floats when dw or db is given: row statistics and ordered partial sums) */
/* kind 1 exact GELU, 2 ReLU: dy NULL -> out = act(x), else out = dy * act'(x) */
int ullsam_train_act(const float* x, const float* dy, float* out, long n, int kind, void* stream);
/* prompt_encoder.py:148 y = x * llm_scale_factor + llm_bias (dy NULL), else out = dy * s, ds += sum dy x, dt += sum dy */
int ullsam_train_scale_shift(const float* x, const float* s, const float* t, const float* dy, float* out, float* ds, float* dt, long n, float* partial, void* stream);
/* (backward: partial = 2 * ceil(n / 1024) floats; ds / dt += the ordered sum of the per-block partials) */
/* softmax attention for the training path, forward (dout NULL: writes out) and backward (dout given: writes dq, adds dk / dv):
 * transformer.py:220-242 (groups 1, causal -1, no mask) and modeling_internlm2.py:383-419 with the additive finfo.min masks of :834-870
 * (groups = H / KV heads, causal = Sk - Sq, key_mask int32 [B, Sk]).  q / dq / out [B,Sq,H,hd], k, v / dk, dv [B,Sk,H/groups,hd] by
 * (batch, token, head) strides; hd <= 128 */
int ullsam_train_attention(const float* q, const float* k, const float* v, const float* dout, float* out, float* dq, float* dk, float* dv,
                           int B, int H, int groups, int hd, int Sq, int Sk, int causal, const int* key_mask, long q_bs, long q_ts, long q_hs,
                           long k_bs, long k_ts, long k_hs, long v_bs, long v_ts, long v_hs, long o_bs, long o_ts, long o_hs, float scale,
                           const float* bias_h, const float* bias_w, float* dbias_h, float* dbias_w, int kw, void* stream);
/* (bias_h [B,H,Sq,Sk/kw], bias_w [B,H,Sq,kw]: the ViT's decomposed relative-position terms, image_encoder.py:325-361, added to the logits
 * as bias_h[q][key / kw] + bias_w[q][key % kw]; NULL elsewhere.  The backward writes dbias_h / dbias_w.) */
/* adjoint of ullsam_im2col3x3 (neck conv3x3 as im2col + Linear): dcols [B*H*W, 9*C] -> dx [B,H,W,C] */
int ullsam_train_col2im3x3(const float* dcols, float* dx, int B, int H, int W, int C, void* stream);
/* Row pass of the training attention in matrix form (large problems: the atomics of the kernel above serialise).  S [B*H, Sq, Sk] holds
 * (q*scale) k^T from ullsam_train_matmul (have_p 0): logits = S + bias + masks, P = softmax in place.  dP NULL: forward, stop there (out = P v by
 * ullsam_train_matmul).  dP = dO v^T given: dS = P (dP - sum_j P_j dP_j) overwrites dP and the decomposed-bias gradient rows are written; have_p 1:
 * S already holds the P the forward kept. */
int ullsam_train_attn_rows(float* S, float* dP, const float* bias_h, const float* bias_w, float* dbias_h, float* dbias_w, const int* key_mask,
                           int B, int H, int Sq, int Sk, int kw, int causal, int have_p, void* stream);
/* 0: keep the row pass on its three-pass form (tests compare it with the register-resident form used for Sk <= 4096); returns the previous setting */
int ullsam_train_set_rows_reg(int on);
/* InternLM2RMSNorm backward (modeling_internlm2.py:75-89); dw may be NULL (frozen LLM) */
int ullsam_train_rmsnorm_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, long rows, int D, float eps, void* stream);
/* apply_rotary_pos_emb (modeling_internlm2.py:233-247) on rows [tokens, heads, hd]; adjoint != 0: its transpose (the backward) */
int ullsam_train_rope(const float* x, const int* pos, const float* cos_tab, const float* sin_tab, float* out, long tokens, int heads, int hd,
                      int tab_rows, int adjoint, void* stream);
/* InternLM2MLP's silu(w1 x) * (w3 x) (modeling_internlm2.py:598-618): dy NULL -> out, else dg / du */
int ullsam_train_swiglu(const float* g, const float* u, const float* dy, float* out, float* dg, float* du, long n, void* stream);
/* adjoint of the bilinear upsample F.interpolate(align_corners=False) of train_joint_v2.py:1073-1078 (planes of oh x ow -> ih x iw) */
int ullsam_train_resize_bwd(const float* dout, float* din, long planes, int ih, int iw, int oh, int ow, void* stream);
/* calc_instance_loss (train_joint_v2.py:774-812) with BCELoss (:638-661) + DiceLoss (:605-636): x logits / t targets [P, npix];
 * sums [P][4], losses [3] = (total, bce, dice); the backward scales by gscale[0] (the incoming gradient of the total) */
int ullsam_train_seg_loss(const float* x, const float* t, float* sums, float* losses, int P, long npix, float smooth, float* partial, void* stream);
/* (partial: P * 4 * ceil(npix / 1024) floats) */
int ullsam_train_seg_loss_bwd(const float* x, const float* t, const float* sums, const float* gscale, float* dx, int P, long npix, float smooth, void* stream);
/* The language-model loss of InternLM2ForCausalLM.forward (modeling_internlm2.py:1084-1096: CrossEntropyLoss() over the shifted logits, mean over labels != -100):
 * fp32 logits [rows, V] with row stride ld, labels int64 [rows]; lse [rows], loss_rows [rows], out2 = {mean loss, 1 / #labelled rows};
 * the backward writes dlogits = (softmax - onehot) * gscale[0] * out2[1] (zero rows where the label is ignored).  Ordered sums: reproducible bits.
 * No labelled row at all: out2[0] is NaN, as torch's mean over zero rows (the reference's `0 * loss + seg` is NaN there too), out2[1] = 0.
 * Deviation: a label outside [0, V) other than -100 is treated as ignored here, where torch raises (device-side assert); the host wrapper (training.lm_loss)
 * rejects such labels when they arrive on the CPU. */
int ullsam_train_cross_entropy(const float* logits, long ld, const long long* labels, float* lse, float* loss_rows, float* out2, long rows, int V, void* stream);
int ullsam_train_cross_entropy_bwd(const float* logits, long ld, const long long* labels, const float* lse, const float* out2, const float* gscale, float* dlogits,
                                   long ldx, long rows, int V, void* stream);   /* dlogits rows of ldx >= V floats: columns V .. ldx - 1 are zeroed (GEMM padding) */
/* dst[idx[r]] += src[r]: gradient of the point-label embedding table (prompt_encoder.py:76-96) */
int ullsam_train_index_add_rows(const float* src, const int* idx, float* dst, long rows, int C, int nrows_dst, void* stream);

/* Row LayerNorm / RMSNorm, fp32 statistics.  image_encoder.py:151,161; common.py:38-43 (LayerNorm2d on NHWC rows);
 * modeling_internlm2.py:138-143 (rms=1); prompt_encoder.py:142-149 (no affine + post scale/shift); transformer.py norms. */
int ullsam_norm(const void* in, int in_dtype, long in_stride, void* out, int out_dtype, long out_stride, const float* w,
                const float* b, long rows, int D, float eps, int rms, int act, const float* post_scale,
                const float* post_shift, void* stream);
int ullsam_set_norm_variant(int v);   /* developer switch: non-zero = wave-per-row kernel for every shape (A/B) */

/* LayerNorm whose result feeds three consumers at once: fp32 stream, compute-dtype copy, compute-dtype (y + pe[row % pe_rows]).
   transformer.py:182 (norm4) followed by :160-165 / :176-178 of the next block. */
int ullsam_norm_fanout(const float* in, long rows, int D, const float* w, const float* b, float eps, float* out_f32, void* out_c,
                       void* out_c_pe, int c_dtype, const float* pe, long pe_rows, void* stream);

/* SAM ViT attention on packed qkv [B, gh*gw, 3*heads*hd]; window > 0 fuses window_partition/unpartition and the pad-token
 * semantics; decomposed rel-pos computed in-kernel.  image_encoder.py:170-177,224-240,243-289,292-361. */
int ullsam_vit_attention(int dtype, const void* qkv, void* out, const void* rel_h, const void* rel_w, const void* qkv_bias,
                         int B, int heads, int hd, int grid_h, int grid_w, int window, void* stream);

/* InternLM2 causal GQA attention (prefill), additive masks as modeling_internlm2.py:96-125,830-851; :383-419. */
int ullsam_causal_attention(int dtype, const void* q, const void* k, const void* v, void* out, const int* key_mask, int B,
                            int H, int KVH, int hd, int Sq, int Sk, int k_cap, int q_pos0, void* stream);

/* Small-shape attention with explicit strides (decoder token attention transformer.py:220-242; q_len==1 decode). */
int ullsam_naive_attention(int dtype, const void* q, const void* k, const void* v, void* out, const int* key_mask, int B,
                           int H, int KVH, int hd, int Sq, int Sk, long q_bs, long q_ts, long q_hs, long k_bs, long k_ts,
                           long k_hs, long v_bs, long v_ts, long v_hs, long o_bs, long o_ts, long o_hs, float scale,
                           void* stream);

/* Decode step (q_len == 1): one bf16 query per head against the bf16 KV cache [B, KVH, cap, 128], the heads of a GQA group
   together, keys split over nsplit workgroups.  modeling_internlm2.py:383-419, mask :834.  workspace f32 [B*KVH*nsplit*(H/KVH)*130]. */
int ullsam_decode_attention(const void* q, const void* kc, const void* vc, const int* key_mask, void* out, int B, int H, int KVH,
                            int hd, int Sk, int cap, float scale, float* workspace, int nsplit, void* stream);

/* Image -> token cross attention (many queries, few keys), fp32.  transformer.py:178-181.  q_batch_stride (elements) = 0 when
   all B batches share one query set (layer 0 of the decoder when B prompts look at one image). */
int ullsam_fewkeys_attention(const float* q, const float* k, const float* v, float* out, int B, int H, int hd, int Sq,
                             int Sk, float scale, long q_batch_stride, void* stream);

/* Token -> image cross attention (T queries, N image keys, 8 heads x 16), K/V streamed once.  transformer.py:160-166,
   100-106.  k/v in kv_dtype with element batch strides (0 = shared image): fp32 K/V T <= 8 (VALU kernel); bf16 K/V T <= 16 (matrix-pipe kernel, queries and
   probabilities as two bf16 terms: fp32-level accuracy on the rounded K/V).  workspace f32 [P*nsplit*T*144]; the caller chooses nsplit (ops.py: from N alone,
   so that the softmax merge order does not depend on the prompt count). */
int ullsam_tok2img_attention(int kv_dtype, const float* q, const void* k, const void* v, float* out, int P, int H, int hd, int T,
                             int N, long k_batch_stride, long v_batch_stride, float scale, float* workspace, int nsplit,
                             void* stream);

/* ViT / projector data movement */
int ullsam_patch_im2col(int dtype, const float* pixels, void* out, int B, int C, int Hs, int Ws, int S, int patch,
                        const float* mean, const float* stdv, void* stream);            /* image_encoder.py:391-395, sam.py:164-174 */
int ullsam_im2col3x3(int dtype, const void* in, void* out, int B, int H, int W, int C, void* stream); /* image_encoder.py:96-102 */
int ullsam_add_cast(const void* a, int a_dtype, long a_rows, const float* b, long b_rows, void* out, int out_dtype,
                    long rows, int cols, void* stream);                                  /* transformer.py:165,181; mask_decoder.py:127 */
int ullsam_transpose_f32(const float* in, float* out, int B, int R, int C, void* stream);  /* image_encoder.py:114 permute */
/* in [R, C] (in_dtype: f32 or bf16) -> out bf16 [C, Rp], transposed, columns R..Rp-1 zero (training: the dW / dX GEMM operands, cast + transposed in one pass) */
int ullsam_transpose_to_bf16(int in_dtype, const void* in, void* out, int R, int C, int Rp, void* stream);
/* fp32 [R, C] -> bf16 [R, C] (out_rm, may be NULL) and bf16 [C, Rp] (out_t: transposed, zero columns behind R) and the column sums sum_r in[r][c] (colsum_out fp32 [C], may be NULL;
 * colsum_ws: ceil(Rp / 64) * C floats of scratch, 64-row blocks added in order) in ONE pass: what a bf16 Linear of the training step needs of an activation and of a gradient.
 * C and Rp multiples of 4. */
int ullsam_cast_transpose_bf16(const float* in, void* out_rm, void* out_t, float* colsum_out, float* colsum_ws, int R, int C, int Rp, void* stream);
int ullsam_pixel_shuffle_ln(int dtype, const float* in_nhwc, void* out, const float* w, const float* b, int B, int H, int W,
                            int C, float eps, void* stream);                             /* modeling_internvl_sam.py:226-251,89 */
int ullsam_pixel_unshuffle(const float* in, float* out_nhwc, int B, int H, int W, int C, void* stream); /* :256-268 */

/* LLM data movement */
int ullsam_scan_image_tokens(const long long* ids, int* rank, int* range, int B, int S, long long img_id, void* stream); /* :135-139,194-199 */
int ullsam_embed_tokens(int dtype, const void* table, const long long* ids, const int* rank, const float* vit_embeds,
                        float* out, int B, int S, int D, int n_img, long vocab, void* stream);                          /* :124-158 */
int ullsam_gather_rows(const void* in, void* out, const int* range, int B, int S, int n, int row_bytes, void* stream);  /* :198-200 */
int ullsam_rope_split(int dtype, const void* qkv, void* q_out, void* k_cache, void* v_cache, const int* pos,
                      const float* cos_tab, const float* sin_tab, int B, int S, int KVH, int G, int hd, int cap,
                      int cache_pos0, int tab_rows, void* stream);                       /* modeling_internlm2.py:361-388,233-247;
                                                                                            pos is clamped to [0, tab_rows) */
int ullsam_argmax(const float* logits, long long* out, int rows, long V, long ld, void* stream);

/* Prompt encoder / mask decoder */
int ullsam_small_linear(const float* x, long ldx, const float* W, const float* b, const float* res, long ldr, float* y,
                        long ldy, int M, int N, int K, int act, void* stream);           /* transformer.py:220-227; mask_decoder.py:171-176 */
/* The image -> token half of a two-way block (transformer.py:176-182) for many prompts in ONE pass over the image-side stream (bf16, embedding 256,
 * internal 128, 8 heads): q = xin Wq^T + bq (xin = keys + pe in bf16), a = softmax_heads(q k_tok^T scale) v_tok, upd = res + a Wo^T + bo (res = keys, fp32),
 * y = LayerNorm(upd) -> out_f32 / out_c (bf16) / out_c_pe = bf16(y + key_pe[row % pe_rows]), each optional.  xin / res have P*N rows, or in_mod / res_mod
 * rows shared by every prompt (layer 0: in_mod / res_mod / pe_rows are 0 or N); ktok / vtok fp32 [P, T, 128], T <= 16. */
int ullsam_i2t_block(const void* xin, long in_mod, const float* res, long res_mod, const void* Wq, const float* bq, const float* ktok,
                     const float* vtok, const void* Wo, const float* bo, const float* lnw, const float* lnb, float eps, const float* key_pe,
                     long pe_rows, float* out_f32, void* out_c, void* out_c_pe, int P, int T, int N, float scale, void* stream);

/* The token -> image attention's k / v projections of the image side in one pass (transformer.py:220-222 on the image tokens; bf16, embedding 256 -> internal 128):
 * K = xk Wk^T + bk, V = xv Wv^T + bv; xk = (keys + pe), xv = keys, bf16 [rows, 256]; Wk / Wv bf16 [128, 256]; bk / bv fp32 [128] or NULL; K / V bf16 [rows, 128]. */
int ullsam_kv_proj(const void* xk, const void* xv, const void* Wk, const void* Wv, const float* bk, const float* bv, void* K, void* V, long rows, void* stream);
/* The token side of a two-way block (transformer.py:153-184; bf16 weights [out, in], embedding 256, 8 heads, T <= 16 tokens per prompt, fp32 token rows) as two launches,
 * one workgroup per prompt (csrc/dectok.hip):
 *   dec_tok_attn: q_in = queries (+ qpe unless skip_pe); self attention (q, k of q_in, v of queries; scale 1 / sqrt(32) after the product, :233-235); out projection
 *     (+ queries unless skip_pe: layer 0 has no residual, :157-158); LayerNorm (norm1) -> queries_out; then q_t2i = (queries_out + qpe) Wq2^T + bq2, the q of the
 *     token -> image attention.  mode 1: q_t2i = (queries + qpe) Wq2^T + bq2 only (the final attention, transformer.py:99-104).
 *   dec_tok_mlp: y = LayerNorm(queries + attn Wo^T + bo) (norm2; with do_mlp 0 that is all: norm_final_attn); y = LayerNorm(y + W2 relu(W1 y + b1) + b2) (norm3) -> queries_out;
 *     k_out = (y + qpe) Wk^T + bk, v_out = y Wv^T + bv: the k / v of the image -> token attention (:176-178).
 * dec_heads: the four hypernetwork MLPs on tokens 1 .. 4 and the IoU head on token 0 (mask_decoder.py:141-149,154-176; three linears, ReLU between): w / b = HOST arrays
 *   of 15 device pointers (chain-major; a chain's last weight zero-padded to a multiple of 16 rows) -> hyper [P, nm, 32], iou [P, n_iou]. */
int ullsam_dec_tok_attn(const float* queries, const float* qpe, float* queries_out, float* q_t2i, const void* Wq, const float* bq, const void* Wk, const float* bk,
                        const void* Wv, const float* bv, const void* Wo, const float* bo, const float* ln_w, const float* ln_b, float eps, const void* Wq2,
                        const float* bq2, int P, int T, int skip_pe, int mode, void* stream);
int ullsam_dec_tok_mlp(const float* queries, const float* attn, const float* qpe, float* queries_out, float* k_out, float* v_out, const void* Wo, const float* bo,
                       const float* ln2_w, const float* ln2_b, float eps2, const void* W1, const float* b1, const void* W2, const float* b2, const float* ln3_w,
                       const float* ln3_b, float eps3, const void* Wk, const float* bk, const void* Wv, const float* bv, int P, int T, int do_mlp, void* stream);
int ullsam_dec_heads(const float* hs, const void* const* w, const float* const* b, float* hyper, float* iou, int P, int T, int n_iou, int m0, int nm, void* stream);
/* (hypernetwork chains m0 .. m0 + nm - 1 only -> hyper [P, nm, 32]: multimask output asks for masks 1 .. 3, mask_decoder.py:100-105) */
/* out[p][t] = t < n0 ? prefix[t] : rows[p][t - n0], fp32 rows of C: the decoder's token matrix (mask_decoder.py:119-123) */
int ullsam_concat_token_rows(const float* prefix, int n0, const float* rows, int n1, float* out, int P, int C, void* stream);
/* Second transposed convolution + GELU + hypernetwork product in one pass (mask_decoder.py:136-147, bf16): u1 bf16 [NB*H*W*4, 64] (first transposed convolution
 * after LayerNorm2d + GELU), w1 bf16 [128 = (ky2, kx2, c), 64], b1 fp32 [128] | NULL, hyper fp32 [NB, NM <= 8, 32] -> out fp32 [NB, NM, 4H, 4W]; the upscaled
 * embedding is never written. */
int ullsam_up2_hyper_masks(const void* u1, const void* w1, const float* b1, const float* hyper, float* out, int NB, int NM, int H, int W, void* stream);
/* First transposed convolution + LayerNorm2d + GELU in one pass (mask_decoder.py:131-138, bf16): src bf16 [rows, 256], w0 bf16 [256 = (ky, kx, c), 256],
 * b0 fp32 [256] | NULL, lnw / lnb fp32 [64] | NULL -> out bf16 [rows * 4, 64]; the fp32 result of the convolution is never written. */
int ullsam_up1_ln_gelu(const void* src, const void* w0, const float* b0, const float* lnw, const float* lnb, float eps, void* out, long rows, void* stream);
int ullsam_skinny_linear(const float* x, long ldx, const float* WT, const float* b, const float* res, long ldr, float* y,
                         long ldy, int M, int N, int K, int act, void* stream);
/* 0: keep ullsam_skinny_linear on its FMA kernel (tests compare it with the exact-fp32 MFMA kernel used for N % 32 == 0, K in {128 .. 2048}); returns the previous setting */
int ullsam_set_skinny_linear_mfma(int on); /* same layers at many prompts; WT = weight^T [K,N] */
int ullsam_sparse_embed(const float* coords, const int* labels, const float* boxes, const float* G, const float* emb,
                        float* out, int P, int Np, int pad, int C, float img_w, float img_h, void* stream); /* prompt_encoder.py:76-103 */
int ullsam_dense_pe(const float* G, float* out_nhwc, int H, int W, int C, void* stream);  /* prompt_encoder.py:230-241 */
int ullsam_mask_downscale(const float* masks, float* out_nhwc, int P, int H, int W, int C, int c1, int c2, const float* w0,
                          const float* b0, const float* g1, const float* be1, const float* w3, const float* b3,
                          const float* g4, const float* be4, const float* w6, const float* b6, void* stream); /* prompt_encoder.py:54-62 */
int ullsam_hyper_masks(int dtype, const void* up2, const float* hyper, float* out, int NB, int NM, int H, int W, int CU, void* stream); /* mask_decoder.py:143-144 */
int ullsam_resize_bilinear(const float* in, long in_plane_stride, int in_ld, int IH, int IW, float* out, unsigned char* mask,
                           int N, int OH, int OW, float thr, void* stream);              /* app.py:635-645; sam.py:154-162,123 */
int ullsam_mask_iou_counts(const unsigned char* a, const unsigned char* b, unsigned long long* counts, int N, long per,
                           void* stream);                                                /* train_joint_v2.py:683-694 */

/* Automatic-mask-generation helpers, bit-exact with utils/amg.py (integer / byte work) */
int ullsam_stability_score(const float* masks, long N, long per, float mask_threshold, float threshold_offset,
                           unsigned int* counts, float* score, void* stream);                 /* amg.py:156-176 */
int ullsam_mask_to_box(const unsigned char* masks, long N, int H, int W, int* boxes, void* stream);   /* amg.py:303-346 */
int ullsam_rle_pack(const unsigned char* masks, long N, int H, int W, unsigned long long* words, int* counts,
                    unsigned char* first, void* stream);                                      /* amg.py:107-135 (transpose + diff) */
int ullsam_rle_emit(const unsigned long long* words, const int* select, long N, int H, int W, const long* offsets, int* out,
                    void* stream);                                                            /* amg.py:119-133 (nonzero -> run edges) */
/* generator fast path: postprocess_masks + stability counts + mask->box + RLE change words in one pass, logits never stored */
int ullsam_amg_postprocess(const float* low, const int* index, long M, int LH, int LW, int S1, int nh, int nw, int CH, int CW,
                           int FH, int FW, int cx0, int cy0, float mask_threshold, float threshold_offset,
                           unsigned long long* words, int* rle_counts, unsigned char* first, int* boxes, unsigned int* stab,
                           void* stream);                      /* sam.py:154-162 + amg.py:156-176, 303-346, 107-135, 251-264 */
int ullsam_nms_mask(const float* boxes, int N, float iou_threshold, unsigned long long* mask, void* stream); /* torchvision.ops.nms (absent dependency) */
int ullsam_threshold_u8(const float* in, unsigned char* out, long n, float thr, void* stream); /* masks > mask_threshold */

/* fp8 (OCP e4m3) ViT path -- BASELINE.json configs[4]; the reference's bf16 encoder linears image_encoder.py:227,171-181 with
   8-bit operands: rows quantised with a per-row scale (optionally behind the block's LayerNorm :166,180), GEMM on the
   block-scaled fp8 MFMA, scales applied in the epilogue */
int ullsam_rows_fp8(const void* in, int in_dtype, long in_stride, void* out_e4m3, long out_stride, float* row_scale, const float* ln_w,
                    const float* ln_b, long rows, int D, float eps, void* stream);
int ullsam_gemm_fp8(const void* A8, long lda, const float* a_scale, const void* W8, long ldw, const float* w_scale, void* C, long ldc,
                    int out_f32, const float* bias, const float* residual, long ldr, int act, int M, int N, int K, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ULLSAM_HIP_H */
