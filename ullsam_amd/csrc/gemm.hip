// C[M,N] = epilogue(A[M,K] . W[N,K]^T)   -- every nn.Linear / 1x1 conv / im2col conv on the hot path.
//
// Reference sites replaced: K1 patch-embed GEMM (image_encoder.py:387-395), K3 qkv
// (:227), K6 proj (:238), K7 MLP (common.py:21-26), K8 neck convs (image_encoder.py:88-104), K9 mlp1
// (modeling_internvl_sam.py:88-93), K12 wqkv / K14 wo / K15 SwiGLU (modeling_internlm2.py:359,421,261-264),
// K16 lm_head (:1081), K17 mlp2 (modeling_internvl_sam.py:95-100), decoder image-side projections
// (transformer.py:220-227) and the two ConvTranspose2d of mask_decoder.py:53-59 (stride == kernel => GEMM).
//
// Layout: A row-major [M, lda], W row-major [N, ldw] (nn.Linear's native [out, in]) so both operands are K-contiguous
// ("B^T input").  K-tile = 128 bytes per row (64 bf16 / 32 f32).  Global->LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per
// wave instruction = 8 rows).  LDS image: 128-byte rows, 16-byte chunk index XOR (row & 7): the DMA writes lane-linear, so the
// swizzle is applied to the per-lane SOURCE address and again on the ds_read_b128 (cdna guide 5.4 rule 21) -> conflict-free reads.
// Kernels in this file (dispatch at the bottom, ullsam_gemm):
//   gemm_ring8_kernel   THE production kernel of bf16 launches with >= 1024 rows (85 % of the bench step): 8 waves in two staggered groups on a
//                       ring of four half-K LDS stages, tile shape as template parameters -- 256x256, 256x320 (ViT-H widths) and 272x256 (the
//                       4324 prompt rows) so that the tile count is a whole number of rounds of the 256 CUs; requests and fragment reads
//                       alternated in the load slot, the slot's control flow resolved at compile time per wave class; epilogues straight
//                       from the accumulators (swapped operands, permuted weight rows, lane-pair swap).
//   gemm_ring8p_kernel  (gemm_ring8p.h, round 6) the ring kernel as a persistent grid -- workgroups walk tiles with the LDS ring kept full across tile borders: launches of more
//                       than one round of tiles with a direct epilogue (llm.w13, llm.wqkv + RoPE, vit.qkv, vit.lin1 of the bench step); outputs bit-equal to gemm_ring8_kernel.
//                       Split-K form (EMODE 2, launch_gemm_ring_splitk + splitk_finish_kernel): launches of <= 128 tiles under a long K (1081 x 4096 outputs: a batch-1
//                       prefill, the frozen LLM of a training step) run up to 8 K ranges side by side into fp32 planes of the workspace, added in order.
//   gemm256_kernel      256x256 tile, two 64 KiB stages, staggered two-group schedule (L0|C0|L1|C1), split-K tail + gemm256_tail_reduce_kernel,
//                       LDS-staged epilogue.  fp32 (parity mode), the wqkv GEMM with its RoPE epilogue, and bf16 launches whose 256x256
//                       tile count leaves a sliver that a split-K tail absorbs.
//   gemm128_kernel      128x128 tile, 4 waves, 2 workgroups per CU, one barrier per K-tile: small M / narrow N / short K and launches whose
//                       256-row tile count would leave the last round mostly empty.
//   gemm256f8_kernel    the 256x256 two-buffer loop on v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 ViT linears, BASELINE configs[4]).
//   gemm_skinny_*       M <= 8 (decode step): weight-streaming, no tiles.
// Variants that were built, measured and removed (a 256x128 three-stage ring, a persistent two-buffer kernel, a four-wave 512-register
// kernel, a stream-K tail, the RoPE epilogue on the ring): DESIGN.md section 7 keeps their numbers.
// Epilogues: bias / GELU(erf) / ReLU / SwiGLU pair / fp32 residual (optionally row-broadcast, for pos_embed), fused.
#include "common.h"
#include <type_traits>

struct GemmArgs {
    const void* A;
    const void* W;
    void* C;
    const float* bias;
    const float* residual;
    int M, N, K;
    long lda, ldw, ldc, ldr;
    int res_row_mod;
    int act;      // 0 none, 1 gelu(erf), 2 relu, 3 swiglu pair (tile = [64 gate | 64 up])
    int out_f32;  // C element type: 1 -> float, 0 -> T
    int vec_ok;   // stores / residual loads may be 16-byte vectors
    int tiles_m, tiles_n;
    // split-K tail (v1 only): tiles [0, full_tiles) run whole; each remaining tile is cut into ksplit K-ranges whose fp32 partial
    // tiles go to `ws` ([tail][ksplit][128*128]) and are summed + finished by gemm_tail_reduce_kernel
    int full_tiles, ksplit;
    float* ws;
    size_t ws_bytes;
    // act == 4: wqkv epilogue = de-interleave + RoPE + KV-cache append (modeling_internlm2.py:361-388, 233-247) instead of a plain store
    const int* rope_pos; const float* rope_cos; const float* rope_sin; void* rope_q; void* rope_k; void* rope_v;
    int rope_S, rope_KVH, rope_G, rope_cap, rope_pos0, rope_rows;
    const float* row_scale;  // fp8 path: per-row scale of A (activation quantisation), per-column scale of W; null elsewhere
    const float* col_scale;
    unsigned long long* dbg;  // diagnostic launches only (ullsam_set_gemm_variant bit 15): s_memtime stamps of the ring kernel (tools/probes/ring8_stamps.py)
    // decode-step prologue (skinny kernels only): A = bf16(RMSNorm(norm_x) * norm_w), normalised while the workgroup stages its rows in LDS
    const float* norm_x; const float* norm_w; float norm_eps; long ldx;
    int group_m;     // tile rows per raster group of the 256-row-tile kernels (tiles of a group run column-major: group_m x tiles_n); 4 by default
};

#ifndef ULLSAM_RING_ASM_DMA
#define ULLSAM_RING_ASM_DMA 1
#endif
static constexpr bool g_asm_dma = ULLSAM_RING_ASM_DMA != 0;   // (side builds with -DULLSAM_RING_ASM_DMA=0 keep the builtin for A/B)
static int g_split_tail = 1;   // ullsam_set_gemm_variant(v | 64) disables the split-K tails (A/B)
static int g_gemm_variant = 0; // bits 0-3 force a kernel: 0 auto, 1 128x128, 3 256x256 two-buffer, 6 256x256 ring, 8 256x320 ring, 9 272x256 ring, 10 208x256 ring (RoPE GEMM only)
static int g_dbg = 0;          // ullsam_set_gemm_variant bit 15: stamp the ring kernel (tools/probes/ring8_stamps.py reads the stamps from the workspace)
static int g_auto_mask = 7;    // ullsam_set_gemm_tuning(1, mask): ring tile shapes the auto dispatch may pick: bit 0 256x256, bit 1 256x320, bit 2 272x256
static int g_group_m = 4;      // ullsam_set_gemm_tuning(0, gm): raster group height (measured: 4 -> 83.26 ms per step, 8 -> 83.73, 2 -> 84.35)

template <typename T>
__device__ __forceinline__ Frag<T> lds_frag(const char* tile, int row, int ks, int g);
template <>
__device__ __forceinline__ Frag<bf16> lds_frag<bf16>(const char* tile, int row, int ks, int g) {
    const int c = (ks * 4 + g) ^ (row & 7);
    return load_frag(reinterpret_cast<const bf16*>(tile + row * 128 + (c << 4)));
}
template <>
__device__ __forceinline__ Frag<float> lds_frag<float>(const char* tile, int row, int ks, int g) {
    (void)ks;
    const int c0 = (2 * g) ^ (row & 7), c1 = (2 * g + 1) ^ (row & 7);
    const float4 a = *reinterpret_cast<const float4*>(tile + row * 128 + (c0 << 4));
    const float4 b = *reinterpret_cast<const float4*>(tile + row * 128 + (c1 << 4));
    Frag<float> f;
    f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w;
    f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
    return f;
}

template <typename OutT>
__device__ __forceinline__ void store_row8(OutT* dst, const float* v, int n_valid, bool vec);
template <>
__device__ __forceinline__ void store_row8<float>(float* dst, const float* v, int n_valid, bool vec) {
    if (vec && n_valid == 8) {
        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
        for (int e = 0; e < n_valid; ++e) dst[e] = v[e];
    }
}
template <>
__device__ __forceinline__ void store_row8<bf16>(bf16* dst, const float* v, int n_valid, bool vec) {
    if (vec && n_valid == 8) {
        bf16x8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
        *reinterpret_cast<bf16x8_t*>(dst) = o;
    } else {
        for (int e = 0; e < n_valid; ++e) dst[e] = (bf16)v[e];
    }
}


// EMODE (compile time, so the common epilogue carries none of the others' registers): 0 standard, 1 wqkv + RoPE (act 4), 2 fp8 scales
template <typename T, typename OutT, int BM, int NTHREADS, int BN = 128, int EMODE = 0>
__device__ __forceinline__ void epilogue_rows(const GemmArgs& p, const float* Cs, int m0, int n0, int tn, int tid) {
    constexpr int TPR = BN / 8;             // threads per row (each owns 8 accumulator columns)
    constexpr int RPP = NTHREADS / TPR;     // rows per pass
    constexpr int PASSES = (BM + RPP - 1) / RPP;   // BM = rows staged in Cs (row stride BN floats)
    constexpr bool RAGGED = BM % RPP != 0;  // 320-wide tiles: 480 of the 512 threads work, 12 rows per pass, the last pass is partial
    static_assert(NTHREADS % TPR == 0, "threads per pass must cover whole rows");
    if (RAGGED && tid >= NTHREADS) return;
    OutT* C = reinterpret_cast<OutT*>(p.C);
    if (p.act == 3) {
        // SwiGLU pair: tile columns [0,64) = gate rows of w1, [64,128) = up rows of w3 (host prepack);
        // out[:, tn*64 + j] = silu(gate_j) * up_j      (modeling_internlm2.py:261-264)
#pragma unroll
        for (int pass = 0; pass < PASSES; ++pass) {
            const int row = pass * RPP + tid / TPR;
            const int t = tid % TPR;
            const int grp = t >> 4, j0 = (t & 15) * 4;  // every 128 accumulator columns = [64 gate | 64 up]
            const int gm = m0 + row;
            if (gm >= p.M) continue;
            const float4 g = *reinterpret_cast<const float4*>(Cs + row * BN + grp * 128 + j0);
            const float4 u = *reinterpret_cast<const float4*>(Cs + row * BN + grp * 128 + 64 + j0);
            float4 o = make_float4(silu_f(g.x) * u.x, silu_f(g.y) * u.y, silu_f(g.z) * u.z, silu_f(g.w) * u.w);
            OutT* dst = C + (size_t)gm * p.ldc + (size_t)tn * (BN / 2) + grp * 64 + j0;
            if (p.vec_ok) {
                store4(dst, o);
            } else {
                dst[0] = from_f32<OutT>(o.x); dst[1] = from_f32<OutT>(o.y);
                dst[2] = from_f32<OutT>(o.z); dst[3] = from_f32<OutT>(o.w);
            }
        }
        return;
    }
    if constexpr (EMODE == 1) {
        // wqkv: columns are (kv head, [G query heads | k | v], 128).  This thread's 8 columns sit in one 128-wide slot of the tile; the
        // rotate_half partner of column d is d +- 64 in the same slot = the same Cs row.  q -> q_out [tok, H*128], k / v -> caches
        // [B, KVH, cap, 128] at cache_pos0 + s; cos / sin rows by position id (clamped to the table), fp32 arithmetic as rope_split_kernel.
        const int c0 = (tid % TPR) * 8;
        const int gn = n0 + c0;
        if (gn >= p.N) return;   // N = KVH*(G+2)*128 is a multiple of 128 only: the upper half of a 256-wide tile may lie past it
        const int slot = gn >> 7, d = gn & 127;
        const int gs = p.rope_G + 2, kv = slot / gs, g = slot - kv * gs;
        const int pc = d < 64 ? c0 + 64 : c0 - 64;   // partner columns in the tile
        float bx[8], bp[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { bx[e] = p.bias ? p.bias[gn + e] : 0.f; bp[e] = p.bias ? p.bias[gn + (pc - c0) + e] : 0.f; }
        T* Q = reinterpret_cast<T*>(p.rope_q);
        T* Kc = reinterpret_cast<T*>(p.rope_k);
        T* Vc = reinterpret_cast<T*>(p.rope_v);
        const bool rotate = g != gs - 1;
        // position ids of all passes first (independent loads), then the cos / sin rows one pass ahead of their use
        int ps[PASSES];
#pragma unroll
        for (int pass = 0; pass < PASSES; ++pass) {
            const int gm = m0 + pass * RPP + tid / TPR;
            ps[pass] = (rotate && gm < p.M) ? min(max(p.rope_pos[gm], 0), p.rope_rows - 1) : 0;
        }
        float4 cn[2], sn[2];
        auto load_cs = [&](int pass, float4 (&c)[2], float4 (&s_)[2]) {
            const float* cp = p.rope_cos + (size_t)ps[pass] * 128 + d;
            const float* sp = p.rope_sin + (size_t)ps[pass] * 128 + d;
            c[0] = *reinterpret_cast<const float4*>(cp); c[1] = *reinterpret_cast<const float4*>(cp + 4);
            s_[0] = *reinterpret_cast<const float4*>(sp); s_[1] = *reinterpret_cast<const float4*>(sp + 4);
        };
        if (rotate) load_cs(0, cn, sn);
#pragma unroll
        for (int pass = 0; pass < PASSES; ++pass) {
            const int row = pass * RPP + tid / TPR;
            const int gm = m0 + row;
            float4 cc[2] = {cn[0], cn[1]}, ss[2] = {sn[0], sn[1]};
            if (rotate && pass + 1 < PASSES) load_cs(pass + 1, cn, sn);
            if (gm >= p.M) continue;
            float x[8], y[8], o[8];
            const float4 xa = *reinterpret_cast<const float4*>(Cs + row * BN + c0), xb = *reinterpret_cast<const float4*>(Cs + row * BN + c0 + 4);
            const float4 ya = *reinterpret_cast<const float4*>(Cs + row * BN + pc), yb = *reinterpret_cast<const float4*>(Cs + row * BN + pc + 4);
            x[0] = xa.x; x[1] = xa.y; x[2] = xa.z; x[3] = xa.w; x[4] = xb.x; x[5] = xb.y; x[6] = xb.z; x[7] = xb.w;
            y[0] = ya.x; y[1] = ya.y; y[2] = ya.z; y[3] = ya.w; y[4] = yb.x; y[5] = yb.y; y[6] = yb.z; y[7] = yb.w;
#pragma unroll
            for (int e = 0; e < 8; ++e) { x[e] += bx[e]; y[e] += bp[e]; }
            const int b = gm / p.rope_S, sq = gm - b * p.rope_S;
            if (!rotate) {
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = x[e];
            } else {
                const float cv[8] = {cc[0].x, cc[0].y, cc[0].z, cc[0].w, cc[1].x, cc[1].y, cc[1].z, cc[1].w};
                const float sv[8] = {ss[0].x, ss[0].y, ss[0].z, ss[0].w, ss[1].x, ss[1].y, ss[1].z, ss[1].w};
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = d < 64 ? x[e] * cv[e] - y[e] * sv[e] : x[e] * cv[e] + y[e] * sv[e];
            }
            T* dst;
            if (g < p.rope_G) dst = Q + (size_t)gm * ((size_t)p.rope_KVH * p.rope_G * 128) + (size_t)(kv * p.rope_G + g) * 128 + d;
            else dst = (g == gs - 2 ? Kc : Vc) + (((size_t)b * p.rope_KVH + kv) * p.rope_cap + p.rope_pos0 + sq) * 128 + d;
            store_row8<T>(dst, o, 8, true);
        }
        return;
    }
    // this thread's 8 columns are the same in every pass: fetch their bias once (it was 64 dependent scalar loads per thread)
    const int c0 = (tid % TPR) * 8;
    const int gn = n0 + c0;
    if (gn >= p.N) return;
    const int n_valid = min(8, p.N - gn);
    float bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = 0.f;
    if (p.bias) {
        if (n_valid == 8 && ((reinterpret_cast<uintptr_t>(p.bias + gn) & 15) == 0)) {
            const float4 b0 = *reinterpret_cast<const float4*>(p.bias + gn);
            const float4 b1 = *reinterpret_cast<const float4*>(p.bias + gn + 4);
            bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
        } else {
            for (int e = 0; e < n_valid; ++e) bv[e] = p.bias[gn + e];
        }
    }
    float cs[8];   // fp8 path: this thread's 8 column scales, fetched once like the bias
    if constexpr (EMODE == 2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) cs[e] = e < n_valid ? p.col_scale[gn + e] : 0.f;
    }
#pragma unroll
    for (int pass = 0; pass < PASSES; ++pass) {
        const int row = pass * RPP + tid / TPR;
        const int gm = m0 + row;
        if (gm >= p.M || (RAGGED && row >= BM)) continue;
        float v[8];
        const float4 a = *reinterpret_cast<const float4*>(Cs + row * BN + c0);
        const float4 b = *reinterpret_cast<const float4*>(Cs + row * BN + c0 + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        if constexpr (EMODE == 2) {  // fp8 operands: acc * scale_A[row] * scale_W[col]
            const float rs = p.row_scale[gm];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= rs * cs[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bv[e];
        if (p.act == 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = gelu_erf(v[e]);
        } else if (p.act == 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (p.residual) {
            const int rr = p.res_row_mod > 0 ? (gm % p.res_row_mod) : gm;
            const float* rp = p.residual + (size_t)rr * p.ldr + gn;
            if (p.vec_ok && n_valid == 8) {
                const float4 r0 = *reinterpret_cast<const float4*>(rp);
                const float4 r1 = *reinterpret_cast<const float4*>(rp + 4);
                v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w;
                v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
            } else {
                for (int e = 0; e < n_valid; ++e) v[e] += rp[e];
            }
        }
        store_row8<OutT>(C + (size_t)gm * p.ldc + gn, v, n_valid, p.vec_ok != 0);
    }
}

template <typename T, int EMODE = 0>
__global__ __launch_bounds__(256) void gemm128_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int EPC = 16 / (int)sizeof(T);
    constexpr int BK = 8 * EPC;
    constexpr int KSTEPS = BK / 32;
    char* As = smem;
    char* Bs = smem + 32768;

    // XCD-aware bijective remap + grouped (8 row-tiles) ordering so neighbouring tiles share an L2
    const int nblk = p.full_tiles;  // == tiles_m * tiles_n when the launch has no split-K tail
    const int bid = blockIdx.x;
    int swz, part = -1, tail_idx = 0;
    if (bid < nblk) {
        const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
        swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    } else {  // tail tile, one K-range of it
        const int t = bid - nblk;
        tail_idx = t / p.ksplit;
        part = t - tail_idx * p.ksplit;
        swz = nblk + tail_idx;
    }
    const int GM = 8;
    const int width = GM * p.tiles_n;
    const int group = swz / width;
    const int first_m = group * GM;
    const int gsize = min(p.tiles_m - first_m, GM);
    const int tm = first_m + (swz % width) % gsize;
    const int tn = (swz % width) / gsize;
    const int m0 = tm * 128, n0 = tn * 128;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const char* a_src[4];
    const char* b_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ (row & 7);
        const int gm = min(m0 + row, p.M - 1);
        const int gn = min(n0 + row, p.N - 1);
        a_src[i] = reinterpret_cast<const char*>(p.A) + (size_t)gm * p.lda * sizeof(T) + (c << 4);
        b_src[i] = reinterpret_cast<const char*>(p.W) + (size_t)gn * p.ldw * sizeof(T) + (c << 4);
    }
    const int nk = p.K / BK;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto stage = [&](int buf, int kt) {
        const size_t koff = (size_t)kt * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds(GLB_PTR(a_src[i] + koff), LDS_PTR(As + buf * 16384 + (wave * 4 + i) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GLB_PTR(b_src[i] + koff), LDS_PTR(Bs + buf * 16384 + (wave * 4 + i) * 1024), 16, 0, 0);
        }
    };

    const int kt0 = part < 0 ? 0 : (int)((long)part * nk / p.ksplit);
    const int kt1 = part < 0 ? nk : (int)((long)(part + 1) * nk / p.ksplit);
    stage(kt0 & 1, kt0);
    for (int kt = kt0; kt < kt1; ++kt) {
        __syncthreads();  // tile kt landed (vmcnt(0) + barrier); everyone is done reading buffer (kt+1)&1
        if (kt + 1 < kt1) stage((kt + 1) & 1, kt + 1);
        const char* Ab = As + (kt & 1) * 16384;
        const char* Bb = Bs + (kt & 1) * 16384;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            Frag<T> a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = lds_frag<T>(Ab, wm * 64 + i * 16 + (lane & 15), ks, lane >> 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = lds_frag<T>(Bb, wn * 64 + j * 16 + (lane & 15), ks, lane >> 4);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) mma16(a[i], b[j], acc[i][j]);
        }
    }

    __syncthreads();
    float* Cs = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wm * 64 + i * 16 + 4 * (lane >> 4) + r;
                const int col = wn * 64 + j * 16 + (lane & 15);
                Cs[row * 128 + col] = acc[i][j][r];
            }
    __syncthreads();
    if (part >= 0) {  // raw fp32 partial tile -> workspace (finished by gemm_tail_reduce_kernel)
        float4* dst = reinterpret_cast<float4*>(p.ws + ((size_t)tail_idx * p.ksplit + part) * 16384);
        const float4* src = reinterpret_cast<const float4*>(Cs);
#pragma unroll
        for (int i = 0; i < 16; ++i) dst[i * 256 + tid] = src[i * 256 + tid];
        return;
    }
    if (p.out_f32)
        epilogue_rows<T, float, 128, 256, 128, EMODE>(p, Cs, m0, n0, tn, tid);
    else
        epilogue_rows<T, T, 128, 256, 128, EMODE>(p, Cs, m0, n0, tn, tid);
}

// Sum the ksplit fp32 partials of a 32-row slab of one tail tile (same grouped tile order) and run the normal epilogue on it.
template <typename T, int EMODE = 0>
__global__ __launch_bounds__(256) void gemm_tail_reduce_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) float Cs[32 * 128];
    const int tid = threadIdx.x;
    const int tail_idx = blockIdx.x >> 2, slab = blockIdx.x & 3;
    const int swz = p.full_tiles + tail_idx;
    const int GM = 8;
    const int width = GM * p.tiles_n;
    const int group = swz / width;
    const int first_m = group * GM;
    const int gsize = min(p.tiles_m - first_m, GM);
    const int tm = first_m + (swz % width) % gsize;
    const int tn = (swz % width) / gsize;
    const float4* src = reinterpret_cast<const float4*>(p.ws + (size_t)tail_idx * p.ksplit * 16384 + slab * 4096);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float4 a = src[i * 256 + tid];
        for (int s = 1; s < p.ksplit; ++s) {
            const float4 b = src[(size_t)s * 4096 + i * 256 + tid];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        reinterpret_cast<float4*>(Cs)[i * 256 + tid] = a;
    }
    __syncthreads();
    if (p.out_f32)
        epilogue_rows<T, float, 32, 256, 128, EMODE>(p, Cs, tm * 128 + slab * 32, tn * 128, tn, tid);
    else
        epilogue_rows<T, T, 32, 256, 128, EMODE>(p, Cs, tm * 128 + slab * 32, tn * 128, tn, tid);
}

// ---------------------------------------------------------------------------------------------------------------
// v3: 256x256 tile, 8 waves (2 x 4), 128x64 per wave (8x4 MFMA 16x16 tiles, 128 accumulator registers), two 64 KiB
// LDS stages, LDS-DMA double buffer with one barrier per K-tile (the v1 loop).  Versus the 64x64 per-wave tile this
// issues 12 instead of 16 ds_read_b128 per 32 MFMAs and half the LDS-DMA instructions per MFMA, and halves the
// L2->LDS bytes per FLOP (ablation + PMC in profiles/).  Used for GEMMs whose 256x256 tile count fills the chip's
// 256 CUs for several rounds; the epilogue is staged through LDS in two 128-row halves.
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int EMODE = 0>  // EMODE: epilogue flavour
__global__ __launch_bounds__(512) void gemm256_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int EPC = 16 / (int)sizeof(T);
    constexpr int BK = 8 * EPC;
    constexpr int KSTEPS = BK / 32;
    constexpr int STAGE = 65536;  // 32 KiB A + 32 KiB B

    const int nblk = p.full_tiles;  // == tiles_m * tiles_n when the launch has no split-K tail
    const int bid = blockIdx.x;
    int swz, part = -1, tail_idx = 0;
    if (bid < nblk) {
        const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
        swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    } else {  // tail tile, one K-range of it (fp32 partial -> workspace, finished by gemm256_tail_reduce_kernel)
        const int t = bid - nblk;
        tail_idx = t / p.ksplit;
        part = t - tail_idx * p.ksplit;
        swz = nblk + tail_idx;
    }
    const int GM = p.group_m;
    const int width = GM * p.tiles_n;
    const int group = swz / width;
    const int first_m = group * GM;
    const int gsize = min(p.tiles_m - first_m, GM);
    const int tm = first_m + (swz % width) % gsize;
    const int tn = (swz % width) / gsize;
    const int m0 = tm * 256, n0 = tn * 256;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    // DMA sources: uniform 64-bit tile base (SGPRs) + one 32-bit byte offset per 1 KiB chunk and lane
    const char* a_base = reinterpret_cast<const char*>(p.A) + (size_t)m0 * p.lda * sizeof(T);
    const char* b_base = reinterpret_cast<const char*>(p.W) + (size_t)n0 * p.ldw * sizeof(T);
    unsigned int a_off[4], b_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ (row & 7);
        const int gm = min(m0 + row, p.M - 1) - m0;
        const int gn = min(n0 + row, p.N - 1) - n0;
        a_off[i] = (unsigned int)((size_t)gm * p.lda * sizeof(T)) + (c << 4);
        b_off[i] = (unsigned int)((size_t)gn * p.ldw * sizeof(T)) + (c << 4);
    }
    const int nk_all = p.K / BK;
    const int kt0 = part < 0 ? 0 : (int)((long)part * nk_all / p.ksplit);
    const int nk = part < 0 ? nk_all : (int)((long)(part + 1) * nk_all / p.ksplit);  // this block's K-tile range is [kt0, nk)

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto stage = [&](int buf, int kt) {
        const size_t koff = (size_t)kt * 128;
        char* base = smem + buf * STAGE;
        const char* ak = a_base + koff;
        const char* bk = b_base + koff;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds(GLB_PTR(ak + a_off[i]), LDS_PTR(base + (wave * 4 + i) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GLB_PTR(bk + b_off[i]), LDS_PTR(base + 32768 + (wave * 4 + i) * 1024), 16, 0, 0);
        }
    };
    if constexpr (KSTEPS == 2) {   // bf16
        // Staggered two-group schedule.  Per K-tile every wave runs four segments  L0 | C0 | L1 | C1  separated by s_barrier:
        //   L0: issue the LDS-DMA of tile kt+1 (8 x 1 KiB) + ds_read the k-step-0 fragments     C0: 32 MFMAs
        //   L1: ds_read the k-step-1 fragments, then wait for this wave's DMA (vmcnt(0))         C1: 32 MFMAs
        // Waves 4-7 run one segment behind waves 0-3 (one extra barrier up front, balanced at the end), so on every SIMD one wave is
        // in a load segment while its partner is in a matrix segment.
        //   RAW: a wave's DMA of tile kt+1 is issued in its L0 and waited for in its L1 (slots 4kt+2 / 4kt+3); the first read of
        //        tile kt+1 is at slot 4kt+4, behind the barrier that ends slot 4kt+3.
        //   WAR: buffer (kt+1)&1 was last read (tile kt-1, k-step 1) at slots 4kt-2 / 4kt-1; the DMA into it starts at slot 4kt.
        const int grp = wave >> 2;
        // (Measured and rejected on this loop in round 2: other placements of the DMA issue / wait -- DESIGN.md section 7.)
        stage(kt0 & 1, kt0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (grp == 1) __builtin_amdgcn_s_barrier();
        for (int kt = kt0; kt < nk; ++kt) {
            const char* Ab = smem + (kt & 1) * STAGE;
            const char* Bb = Ab + 32768;
            Frag<T> a8[8], b[4];
            // ---- L0
            if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = lds_frag<T>(Bb, wn * 64 + j * 16 + (lane & 15), 0, lane >> 4);
#pragma unroll
            for (int i = 0; i < 8; ++i) a8[i] = lds_frag<T>(Ab, wm * 128 + i * 16 + (lane & 15), 0, lane >> 4);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // ---- C0
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) mma16(a8[i], b[j], acc[i][j]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // ---- L1
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = lds_frag<T>(Bb, wn * 64 + j * 16 + (lane & 15), 1, lane >> 4);
#pragma unroll
            for (int i = 0; i < 8; ++i) a8[i] = lds_frag<T>(Ab, wm * 128 + i * 16 + (lane & 15), 1, lane >> 4);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // ---- C1
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) mma16(a8[i], b[j], acc[i][j]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();
    } else {   // fp32: one k-step per K-tile, one barrier per K-tile
        stage(kt0 & 1, kt0);
        for (int kt = kt0; kt < nk; ++kt) {
            __syncthreads();  // tile kt landed (vmcnt(0) + barrier); every wave is done reading buffer (kt+1)&1
            if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
            const char* Ab = smem + (kt & 1) * STAGE;
            const char* Bb = Ab + 32768;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                Frag<T> b[4], a8[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j] = lds_frag<T>(Bb, wn * 64 + j * 16 + (lane & 15), ks, lane >> 4);
#pragma unroll
                for (int i = 0; i < 8; ++i) a8[i] = lds_frag<T>(Ab, wm * 128 + i * 16 + (lane & 15), ks, lane >> 4);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) mma16(a8[i], b[j], acc[i][j]);
                __builtin_amdgcn_s_setprio(0);
            }
        }
    }

    float* Cs = reinterpret_cast<float*>(smem);  // [128][256] fp32 = 128 KiB, one 128-row half at a time
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        __syncthreads();
        if (wm == half) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = i * 16 + 4 * (lane >> 4) + r;
                        const int col = wn * 64 + j * 16 + (lane & 15);
                        Cs[row * 256 + col] = acc[i][j][r];
                    }
        }
        __syncthreads();
        if (part >= 0) {  // raw fp32 partial half-tile -> workspace
            float4* dst = reinterpret_cast<float4*>(p.ws + ((size_t)tail_idx * p.ksplit + part) * 65536 + half * 32768);
            const float4* src = reinterpret_cast<const float4*>(Cs);
#pragma unroll
            for (int i = 0; i < 16; ++i) dst[i * 512 + tid] = src[i * 512 + tid];
        } else if (p.out_f32) {
            epilogue_rows<T, float, 128, 512, 256, EMODE>(p, Cs, m0 + half * 128, n0, tn, tid);
        } else {
            epilogue_rows<T, T, 128, 512, 256, EMODE>(p, Cs, m0 + half * 128, n0, tn, tid);
        }
    }
}

// Sum the ksplit fp32 partials of a 16-row slab of one 256x256 tail tile and run the normal epilogue on it.
// ROWS x 256 slab per workgroup of NT threads.  <16, 256>: few, fat workgroups -- right when the partials are tens of MB and the
// pass is bandwidth-bound (ViT lin2: 64 tail tiles x 4 splits).  <4, 128>: 64 workgroups per tail tile for the LLM's 16-tile tails
// (wo / w2, 8 splits), where the launch is latency-bound: 12.8 -> 8.9 us.  The ksplit loads of a thread are independent and
// unrolled, so they are in flight together.
template <typename T, int ROWS, int NT, int EMODE = 0>
__global__ __launch_bounds__(NT) void gemm256_tail_reduce_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) float Cs[ROWS * 256];
    constexpr int SPT = 256 / ROWS;              // slabs per tile
    constexpr int VPT = ROWS * 64 / NT;          // float4 per thread and partial
    const int tid = threadIdx.x;
    const int tail_idx = blockIdx.x / SPT, slab = blockIdx.x % SPT;
    const int swz = p.full_tiles + tail_idx;
    const int GM = p.group_m;
    const int width = GM * p.tiles_n;
    const int group = swz / width;
    const int first_m = group * GM;
    const int gsize = min(p.tiles_m - first_m, GM);
    const int tm = first_m + (swz % width) % gsize;
    const int tn = (swz % width) / gsize;
    const float4* src = reinterpret_cast<const float4*>(p.ws + (size_t)tail_idx * p.ksplit * 65536 + slab * (ROWS * 256));
    float4 acc[VPT];
#pragma unroll
    for (int i = 0; i < VPT; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
    for (int s = 0; s < p.ksplit; ++s) {
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const float4 b = src[(size_t)s * 16384 + i * NT + tid];
            acc[i].x += b.x; acc[i].y += b.y; acc[i].z += b.z; acc[i].w += b.w;
        }
    }
#pragma unroll
    for (int i = 0; i < VPT; ++i) reinterpret_cast<float4*>(Cs)[i * NT + tid] = acc[i];
    __syncthreads();
    const int mrow = tm * 256 + slab * ROWS;
    if (p.out_f32)
        epilogue_rows<T, float, ROWS, NT, 256, EMODE>(p, Cs, mrow, tn * 256, tn, tid);
    else
        epilogue_rows<T, T, ROWS, NT, 256, EMODE>(p, Cs, mrow, tn * 256, tn, tid);
}

// Helpers of the direct (accumulator -> memory) epilogues of the ring kernel.
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ void gstore16(void* ptr, u32x4 v) {   // exactly one global_store_dwordx4 (the counted vmcnt waits rely on it)
    *reinterpret_cast<u32x4*>(ptr) = v;
}
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    bf16x2_t v;
    v[0] = (bf16)lo; v[1] = (bf16)hi;
    return __builtin_bit_cast(unsigned int, v);
}

__device__ __forceinline__ unsigned int lane_xor1(unsigned int v) {   // value of lane ^ 1 (DPP quad_perm [1,0,3,2], one VALU op)
    return (unsigned int)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);
}
// Full-line stores.  After the MFMA a row's 128 B line (bf16: 64 columns, fp32: 32 columns) sits in FOUR lanes (g4 = 0..3) as two 16 B
// pieces each, so a store instruction could only write 64 B pieces -- measured at 28 GB/s per CU against 50 GB/s for whole lines
// (tools/store_probe.hip).  Neighbouring lanes (rows m, m+1) therefore trade one piece: the even lane ends up with the LOW 64 B halves of
// both rows, the odd lane with the HIGH halves, and one instruction writes 8 rows x 128 B.   lo / hi: this lane's two pieces.
// Returns the pieces to store at (row = m & ~1, then row + 1), column piece (odd ? 4 : 0) + g4.
__device__ __forceinline__ void pair_swap(const u32x4& lo, const u32x4& hi, bool odd, u32x4& first, u32x4& second) {
    u32x4 send, recv;
#pragma unroll
    for (int e = 0; e < 4; ++e) send[e] = odd ? lo[e] : hi[e];
#pragma unroll
    for (int e = 0; e < 4; ++e) recv[e] = lane_xor1(send[e]);
#pragma unroll
    for (int e = 0; e < 4; ++e) { first[e] = odd ? recv[e] : lo[e]; second[e] = odd ? hi[e] : recv[e]; }
}

template <typename T, int EMODE = 0>
static int launch_gemm_v3_impl(GemmArgs a, hipStream_t stream) {
    static PerDeviceOnce attr_set;
    if (attr_set.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_kernel<T, EMODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    }
    a.tiles_m = (a.M + 255) / 256;
    a.tiles_n = (a.N + 255) / 256;
    // split-K tail (one 256x256 tile per CU => 256 tiles per wave): a sliver of <= 64 tiles behind >= 1 full wave is cut along K
    const int T_ = a.tiles_m * a.tiles_n;
    const int nk = a.K / (128 / (int)sizeof(T));
    const int tail = T_ % 256;
    a.full_tiles = T_;
    a.ksplit = 1;
    if (g_split_tail && a.ws && T_ > 256 && tail > 0 && tail <= 64 && nk >= 64) {
        int S = 256 / tail;
        if (S > 8) S = 8;
        if (S > nk / 8) S = nk / 8;
        if (S >= 2 && (size_t)tail * S * 262144 <= a.ws_bytes) {
            a.full_tiles = T_ - tail;
            a.ksplit = S;
        }
    }
    gemm256_kernel<T, EMODE><<<dim3(a.full_tiles + (T_ - a.full_tiles) * a.ksplit), dim3(512), 131072, stream>>>(a);
    ULLSAM_LAUNCH_CHECK();
    if (a.ksplit > 1) {
        const int tail_tiles = T_ - a.full_tiles;
        if (tail_tiles * a.ksplit <= 128) gemm256_tail_reduce_kernel<T, 4, 128, EMODE><<<dim3(tail_tiles * 64), dim3(128), 0, stream>>>(a);
        else gemm256_tail_reduce_kernel<T, 16, 256, EMODE><<<dim3(tail_tiles * 16), dim3(256), 0, stream>>>(a);
        ULLSAM_LAUNCH_CHECK();
    }
    return 0;
}
template <typename T>
static int launch_gemm_v3(GemmArgs a, hipStream_t stream) {
    if (a.act == 4) return launch_gemm_v3_impl<T, 1>(a, stream);   // wqkv + RoPE epilogue
    return launch_gemm_v3_impl<T, 0>(a, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// The ring kernel (bf16): BM x BN tile, 8 waves (2 x 4; a wave owns MI0 or MI1 sub-tile rows x NTW sub-tile columns of 16x16), two wave groups
// (waves 0-3 / 4-7 = upper / lower wave row) staggered by one slot on a ring of FOUR half-K LDS stages.  A stage is ONE MFMA k-step (32 deep):
// A BM rows x 64 B + B BN rows x 64 B; per stage and wave: 4-5 LDS-DMA pieces (1 KiB = 16 rows x 64 B; piece q of an operand belongs to wave
// q % 8), 12-13 fragment reads, 32-40 MFMAs; the request runs two stages ahead:
//   slot 2s   : group 0  L(s) = request stage s+2 + read the fragments of stage s, wait until stage s+1 has landed (counted vmcnt)
//               group 1  C(s-1)
//   slot 2s+1 : group 0  C(s) = the MFMAs          group 1  L(s)
// RAW: a wave's pieces of stage s+2 are issued in its L(s) and waited for in its L(s+1) (the counted wait leaves only the youngest stage in
//      flight); group 1's wait is in slot 2(s+2)-1, the barrier ending that slot precedes the first read in slot 2(s+2).
// WAR: buffer (s+2) & 3 held stage s-2, last read in slots 2(s-2) / +1: four slots before the DMA is issued.
// LDS image: 64-byte rows, 16-byte chunk c of row r at c ^ ((r >> 2) & 2): a ds_read_b128 lane group {rows rho, chunk g} covers all 16
// sixteen-byte bank groups of the four 64 B rows sharing a 256 B bank line exactly once (conflict-free); the DMA applies the same
// permutation to its source chunks (the LDS-DMA writes lane-linear).
// Tile shapes (template arguments): 256x256 <8,8,4>; 256x320 <8,8,5> -- the ViT-H widths are multiples of 320 (1280 = 4 x 320, 3840 = 12 x 320,
// 5120 = 16 x 320), and with 64 tile rows (M = 16384) the tile count becomes a multiple of the 256 CUs: vit.proj / lin2 are ONE round instead of
// 1.25, vit.qkv 3 rounds of 1.25 x the work instead of 4; 272x256 <9,8,4> -- the bench's 4 x 1081 = 4324 prompt rows are 16 x 272: llm.wo / w2
// one round of 256 tiles, llm.w13 7 whole rounds instead of 7.44.  The LDS-staged epilogue (odd shapes) uses up to 128 rows x 320 fp32 = 160 KiB.
// ---------------------------------------------------------------------------------------------------------------
// One LDS-DMA request in its SGPR-base form: wave-uniform 64-bit address + this lane's 32-bit byte offset -> LDS at `lds_addr` + 16 lane.  As asm,
// because through the builtin hipcc adds the two with a 64-bit VALU instruction (v_lshl_add_u64) per request inside the ring loop's load slot, the
// critical path of a stage.  M0 (the LDS base) is saved and restored inside the statement (cdna guide 5.7); no register destination; completion is
// counted by the loop's own s_waitcnt vmcnt.
__device__ __forceinline__ void glds16_sbase(const char* base, unsigned int off, unsigned int lds_addr) {
    unsigned int keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off), "s"(base), "s"(lds_addr) : "memory");
}

template <int MI0, int MI1, int NTW, bool STAMP = false, int EMODE = 0>   // sub-tile rows of the upper / lower wave row, sub-tile columns per wave; EMODE 1: wqkv + RoPE epilogue (act 4)
__global__ __launch_bounds__(512) void gemm_ring8_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned long long t_entry = STAMP ? __builtin_amdgcn_s_memtime() : 0ull;   // (diagnostic build) the workgroup's first instruction
    typedef bf16 T;
    constexpr int BM = 16 * (MI0 + MI1), BN = 64 * NTW, MI = MI0 > MI1 ? MI0 : MI1;
    constexpr int PA = BM / 16, PB = BN / 16;          // DMA pieces (16 rows x 64 B) per stage
    constexpr int ASZ = BM * 64, STG = ASZ + BN * 64;
    static_assert(PA >= 8 && PA <= 24 && PB >= 8 && PB <= 24, "piece assignment below: one to three pieces per wave and operand");
    if constexpr (EMODE == 2) {   // split-K (launch_gemm_ring_splitk): part blockIdx.y sums its K / gridDim.y range into plane `part` of the fp32 workspace that p.C points at
        p.K /= (int)gridDim.y;
        p.A = reinterpret_cast<const char*>(p.A) + (size_t)blockIdx.y * p.K * 2;
        p.W = reinterpret_cast<const char*>(p.W) + (size_t)blockIdx.y * p.K * 2;
        p.C = reinterpret_cast<float*>(p.C) + (size_t)blockIdx.y * p.M * p.ldc;
    }

    const int nblk = p.full_tiles;
    const int bid = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
    const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int GM = p.group_m;
    const int width = GM * p.tiles_n;
    const int group = swz / width;
    const int first_m = group * GM;
    const int gsize = min(p.tiles_m - first_m, GM);
    const int tm = first_m + (swz % width) % gsize;
    const int tn = (swz % width) / gsize;
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3, grp = wave >> 2;
    const int g4 = lane >> 4, mm = lane & 15;

    // DMA pieces (16 rows x 64 B): piece q of an operand belongs to wave q % 8 (slots q / 8 = 0, 1, 2); a wave's piece count per stage
    // (na + nb, wave-uniform) is what its counted wait leaves in flight
    const char* a_base = reinterpret_cast<const char*>(p.A) + (size_t)m0 * p.lda * 2;
    const char* b_base = reinterpret_cast<const char*>(p.W) + (size_t)n0 * p.ldw * 2;
    const int na = (PA - wave + 7) / 8, nb = (PB - wave + 7) / 8;
    // The MFMA operands are SWAPPED (weight fragment first), so a lane's four accumulator registers of a sub-tile are four consecutive COLUMNS
    // of one output row and the epilogue can store straight from the accumulators; for 2-byte outputs the weight rows of a wave are permuted on
    // their way into the LDS so that two sub-tiles give a lane eight consecutive columns (the persistent kernel's scheme).  With WW = 16 NTW
    // columns per wave, LDS row R (wave column wb = R / WW, sub-tile j = (R % WW) >> 4, r = R & 15) holds weight row
    //   bf16 out:  WW wb + 32 (j >> 1) + 8 (r >> 2) + 4 (j & 1) + (r & 3)  for j < 4,  WW wb + 64 + r  for the fifth sub-tile of a 320-wide tile
    //   SwiGLU (NTW 4):  128 (wb >> 1) + 64 (j >> 1) + 32 (wb & 1) + 8 (r >> 2) + 4 (j & 1) + (r & 3)          fp32 out:  R
    static_assert(NTW == 4 || NTW == 5, "epilogue layouts below: four sub-tiles in two pairs, optionally a fifth on its own");
    constexpr int WW = 16 * NTW;
    const int perm = EMODE == 1 ? 2 : p.out_f32 ? 0 : (p.act == 3 ? 2 : 1);   // (RoPE pairs column d with d + 64 of a head slot like SwiGLU pairs gate with up)
    auto w_row = [&](int R) {
        const int wb = R / WW, q = R - wb * WW, j = q >> 4, r = q & 15;
        if (perm == 1) return j < 4 ? WW * wb + 32 * (j >> 1) + 8 * (r >> 2) + 4 * (j & 1) + (r & 3) : WW * wb + 64 + r;
        if (perm == 2) return 128 * (wb >> 1) + 64 * (j >> 1) + 32 * (wb & 1) + 8 * (r >> 2) + 4 * (j & 1) + (r & 3);
        return R;
    };
    unsigned int a_off[3], b_off[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int row = (wave + 8 * i) * 16 + (lane >> 2);
        const int c = (lane & 3) ^ ((row >> 2) & 2);
        a_off[i] = (unsigned int)((size_t)(min(m0 + row, p.M - 1) - m0) * p.lda * 2) + (c << 4);
        b_off[i] = (unsigned int)((size_t)(min(n0 + w_row(row), p.N - 1) - n0) * p.ldw * 2) + (c << 4);
    }
    const int st1 = p.K >> 5;  // stages (32-deep k-steps)

    const int mi = wm == 0 ? MI0 : MI1;            // this wave's sub-tile rows (wave-uniform)
    const int row_w = wm == 0 ? 0 : MI0 * 16;      // first tile row of this wave
    f32x4 acc[MI][NTW];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto stage = [&](int s) {   // request stage s into ring slot s & 3
        char* base = smem + (s & 3) * STG;
        const char* ak = a_base + (size_t)s * 64;
        const char* bk = b_base + (size_t)s * 64;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (i < na) __builtin_amdgcn_global_load_lds(GLB_PTR(ak + a_off[i]), LDS_PTR(base + (wave + 8 * i) * 1024), 16, 0, 0);
            if (i < nb) __builtin_amdgcn_global_load_lds(GLB_PTR(bk + b_off[i]), LDS_PTR(base + ASZ + (wave + 8 * i) * 1024), 16, 0, 0);
        }
    };
    auto frag = [&](const char* tile, int row) -> Frag<T> {
        return load_frag(reinterpret_cast<const T*>(tile + row * 64 + ((g4 ^ ((row >> 2) & 2)) << 4)));
    };
    // wait until only the youngest stage's pieces (this wave's na + nb) are in flight
    const int npc = na + nb;
    auto wait_one_left = [&]() {
        if (npc == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (npc == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (npc == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else if (npc == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    };

    // Bias: the tile's BN bias values travel by LDS-DMA into the ring slot that a request for "stage st1" would have used, issued between the
    // main loop and the peeled tail (whose waits are vmcnt(0)), and are read back after the loop.  (Rounds 1-3 held a lane's 4 NTW values in
    // registers across the K loop: 16-20 VGPRs that made the 256x320 instantiation spill.)  Columns past N read the last aligned group.
    const bool has_bias = p.bias && p.act != 3 && EMODE == 0;
    char* const bias_lds = smem + (st1 & 3) * STG;
    auto bias_dma = [&]() __attribute__((always_inline)) {
        if (has_bias && wave < (BN * 4 + 1023) / 1024) {
            const int c = min(n0 + wave * 256 + lane * 4, p.N - 4);
            __builtin_amdgcn_global_load_lds(GLB_PTR(p.bias + c), LDS_PTR(bias_lds + wave * 1024), 16, 0, 0);
        }
    };
    // prologue: stages 0 and 1 requested, stage 0 landed
    stage(0);
    if (1 < st1) { stage(1); wait_one_left(); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();
    // (diagnostic build only) shader-clock stamps per workgroup and wave group: loop entry, loop exit, end
    auto stamp = [&](int kk) {
        if (STAMP && p.dbg && (tid & 255) == 0) p.dbg[((size_t)bid * 2 + grp) * 4 + kk] = __builtin_amdgcn_s_memtime();
    };
    auto stamp16 = [&](int kk) {   // inside stage 10: [workgroup][group][8]
        if (STAMP && p.dbg && (tid & 255) == 0) p.dbg[(1 << 19) + ((size_t)bid * 2 + grp) * 8 + kk] = __builtin_amdgcn_s_memtime();
    };
    stamp(0);
    {
        // Round 3: the same schedule with everything a stage decides at run time decided ONCE.  The load slot is the critical path of a stage
        // (it is longer than the partner group's matrix slot), and the wave's instruction stream is in order: every scalar compare / branch in
        // it -- "is there a stage s + 2?", "does this wave own a third piece?", the five-way choice of the counted wait -- sat between the
        // requests and reads and was paid twice per stage.  Here the piece counts (NA, NB) are template arguments of the loop body (a wave class
        // is picked once, before the loop: at most four copies of the body), the last two stages (nothing left to request) run in a peeled tail,
        // and the wait is one literal s_waitcnt.
        auto body = [&](int s, auto NA_c, auto NB_c, auto MORE_c) __attribute__((always_inline)) {
            constexpr int NA = decltype(NA_c)::value, NB = decltype(NB_c)::value;
            constexpr bool MORE = decltype(MORE_c)::value;
            const char* Ab = smem + (s & 3) * STG;
            const char* Bb = Ab + ASZ;
            Frag<T> a8[MI], b[NTW];
            {
                char* base = smem + ((s + 2) & 3) * STG;
                const char* ak = a_base + (size_t)(s + 2) * 64;
                const char* bk = b_base + (size_t)(s + 2) * 64;
                auto request = [&](int qi) __attribute__((always_inline)) {   // request qi of this wave: A0 B0 A1 B1 A2 B2 (qi is a constant after unrolling)
                    const int i = qi >> 1;
                    if (g_asm_dma) {
                        if (MORE && !(qi & 1) && i < NA) glds16_sbase(ak, a_off[i], (unsigned int)(uintptr_t)LDS_PTR(base + (wave + 8 * i) * 1024));
                        if (MORE && (qi & 1) && i < NB) glds16_sbase(bk, b_off[i], (unsigned int)(uintptr_t)LDS_PTR(base + ASZ + (wave + 8 * i) * 1024));
                    } else {
                        if (MORE && !(qi & 1) && i < NA) __builtin_amdgcn_global_load_lds(GLB_PTR(ak + a_off[i]), LDS_PTR(base + (wave + 8 * i) * 1024), 16, 0, 0);
                        if (MORE && (qi & 1) && i < NB) __builtin_amdgcn_global_load_lds(GLB_PTR(bk + b_off[i]), LDS_PTR(base + ASZ + (wave + 8 * i) * 1024), 16, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                };
                // requests and reads alternated: one request, then three reads, ... (the order round 2 measured best)
                int qi = 0;
                request(qi++);
#pragma unroll
                for (int r = 0; r < NTW + MI; ++r) {
                    if (r < NTW) b[r] = frag(Bb, wn * (16 * NTW) + r * 16 + mm);
                    else if ((r - NTW) < MI1 || (r - NTW) < mi) a8[r - NTW] = frag(Ab, row_w + (r - NTW) * 16 + mm);
                    __builtin_amdgcn_sched_barrier(0);
                    if ((r % 3) == 2 && qi < 6) request(qi++);
                }
#pragma unroll
                for (; qi < 6; ++qi) request(qi);
            }
            if constexpr (!MORE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if constexpr (NA + NB == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if constexpr (NA + NB == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else if constexpr (NA + NB == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if constexpr (NA + NB == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < MI; ++i)
                if (i < MI1 || i < mi) {
#pragma unroll
                    for (int j = 0; j < NTW; ++j) mma16(b[j], a8[i], acc[i][j]);
                }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        };
        auto run = [&](auto NA_c, auto NB_c) __attribute__((always_inline)) {
            int s = 0;
            for (; s + 2 < st1; ++s) body(s, NA_c, NB_c, std::true_type{});
            bias_dma();
            for (; s < st1; ++s) body(s, NA_c, NB_c, std::false_type{});
        };
        constexpr int NA_LO = PA / 8, NA_HI = (PA + 7) / 8, NB_LO = PB / 8, NB_HI = (PB + 7) / 8;
        if (NA_HI != NA_LO && na == NA_HI) {
            if (NB_HI != NB_LO && nb == NB_HI) run(std::integral_constant<int, NA_HI>{}, std::integral_constant<int, NB_HI>{});
            else run(std::integral_constant<int, NA_HI>{}, std::integral_constant<int, NB_LO>{});
        } else {
            if (NB_HI != NB_LO && nb == NB_HI) run(std::integral_constant<int, NA_LO>{}, std::integral_constant<int, NB_HI>{});
            else run(std::integral_constant<int, NA_LO>{}, std::integral_constant<int, NB_LO>{});
        }
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
    stamp(1);
    // this lane's 4 NTW bias values (sub-tile t: fp32 4 at 16 t + 4 g4; bf16 4 at 32 (t >> 1) + 8 g4 + 4 (t & 1) for t < 4 and at 64 + 4 g4 for the fifth)
    float bv[4 * NTW];
#pragma unroll
    for (int e = 0; e < 4 * NTW; ++e) bv[e] = 0.f;
    if (has_bias) {
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const int col = wn * WW + ((p.out_f32 || t == 4) ? 16 * t + 4 * g4 : 32 * (t >> 1) + 8 * g4 + 4 * (t & 1));
            const float4 x = *reinterpret_cast<const float4*>(bias_lds + col * 4);
            bv[4 * t] = x.x; bv[4 * t + 1] = x.y; bv[4 * t + 2] = x.z; bv[4 * t + 3] = x.w;
        }
    }

    // ---- epilogue.  acc[i][j][r] = C[m0 + row_w + 16 i + mm][n0 + ncol(j) + r]: straight from the accumulators where the layout allows
    {
        const bool odd = mm & 1;
        const int row0 = m0 + row_w + mm;
        const int rowp = row0 & ~1;   // after pair_swap the lane pair (mm, mm ^ 1) owns rows rowp + 16 i and rowp + 16 i + 1
        const bool direct = p.vec_ok && (p.N & 7) == 0 && (p.M & 1) == 0;
        auto gst = [&](void* ptr, u32x4 v) __attribute__((always_inline)) { *reinterpret_cast<u32x4*>(ptr) = v; };
        if constexpr (EMODE == 1) {
            // wqkv + RoPE + KV-cache append (modeling_internlm2.py:359-388, 233-247), straight from the accumulators.  The tile's 256 columns are two
            // 128-wide head slots; with the SwiGLU-style row permutation this lane holds, of output row row0 + 16 i, columns d .. d + 7 of a slot
            // (sub-tiles 0 / 1) and their rotate_half partners d + 64 .. d + 71 (sub-tiles 2 / 3), d = 32 (wn & 1) + 8 g4 < 64.  cos / sin rows are
            // cat(freqs, freqs) (:160-166): the values at d + 64 are those at d, so one 8-wide cos and sin segment per row serves both halves.
            // The position ids of all rows are fetched first, the cos / sin segments one row ahead of their use.
            const int slot = (n0 >> 7) + (wn >> 1), d = 32 * (wn & 1) + 8 * g4;
            const int gs = p.rope_G + 2, kv = slot / gs, g = slot - kv * gs;
            const bool live = slot < p.rope_KVH * gs;          // (N is a multiple of 128 only: the upper slot of the last tile column may lie past it)
            const bool rotate = g != gs - 1;                   // v is not rotated
            float bl[8], bh[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { bl[e] = (p.bias && live) ? p.bias[slot * 128 + d + e] : 0.f; bh[e] = (p.bias && live) ? p.bias[slot * 128 + d + 64 + e] : 0.f; }
            T* const Q = reinterpret_cast<T*>(p.rope_q);
            T* const KVc = reinterpret_cast<T*>(g == gs - 2 ? p.rope_k : p.rope_v);
            int ps[MI];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int gm = row0 + 16 * i;
                ps[i] = (live && rotate && gm < p.M && (i < MI1 || i < mi)) ? min(max(p.rope_pos[gm], 0), p.rope_rows - 1) : 0;
            }
            float4 cn[2], sn[2];
            auto load_cs = [&](int i, float4 (&c)[2], float4 (&s_)[2]) __attribute__((always_inline)) {
                const float* cp = p.rope_cos + (size_t)ps[i] * 128 + d;
                const float* sp = p.rope_sin + (size_t)ps[i] * 128 + d;
                c[0] = *reinterpret_cast<const float4*>(cp); c[1] = *reinterpret_cast<const float4*>(cp + 4);
                s_[0] = *reinterpret_cast<const float4*>(sp); s_[1] = *reinterpret_cast<const float4*>(sp + 4);
            };
            if (rotate) load_cs(0, cn, sn);
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const float4 c0 = cn[0], c1 = cn[1], s0 = sn[0], s1 = sn[1];
                if (rotate && i + 1 < MI) load_cs(i + 1, cn, sn);
                const int gm = row0 + 16 * i;
                if (!(i < MI1 || i < mi) || gm >= p.M || !live) continue;
                float x[8], y[8], lo[8], hi[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { x[e] = acc[i][e >> 2][e & 3] + bl[e]; y[e] = acc[i][2 + (e >> 2)][e & 3] + bh[e]; }
                if (rotate) {   // q_embed = q cos + rotate_half(q) sin, rotate_half = cat(-x2, x1)  (same fp32 products and sums as rope_split_kernel / epilogue_rows)
                    const float cv[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w}, sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) { lo[e] = fmaf(x[e], cv[e], -(y[e] * sv[e])); hi[e] = fmaf(y[e], cv[e], x[e] * sv[e]); }   // one rounded product + one fma, spelled out: left to -ffp-contract the one-tile and the persistent instantiations contracted differently (1 bf16 ulp in 3 of 10^6 outputs)
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { lo[e] = x[e]; hi[e] = y[e]; }
                }
                T* dst;
                if (g < p.rope_G) dst = Q + (size_t)gm * ((size_t)p.rope_KVH * p.rope_G * 128) + (size_t)(kv * p.rope_G + g) * 128 + d;
                else {
                    const int bi = gm / p.rope_S, sq = gm - bi * p.rope_S;
                    dst = KVc + (((size_t)bi * p.rope_KVH + kv) * p.rope_cap + p.rope_pos0 + sq) * 128 + d;
                }
                gst(dst, (u32x4){pack_bf16x2(lo[0], lo[1]), pack_bf16x2(lo[2], lo[3]), pack_bf16x2(lo[4], lo[5]), pack_bf16x2(lo[6], lo[7])});
                gst(dst + 64, (u32x4){pack_bf16x2(hi[0], hi[1]), pack_bf16x2(hi[2], hi[3]), pack_bf16x2(hi[4], hi[5]), pack_bf16x2(hi[6], hi[7])});
            }
        } else if (direct && !p.out_f32 && p.act != 3) {
            // bf16 (+bias, +GELU / ReLU): sub-tiles (2 h, 2 h + 1) are this lane's 8 columns of the 32-column group h; the two groups are 128 contiguous
            // bytes of a row -> pair_swap; the fifth sub-tile of a 320-wide tile is 4 columns (8 bytes) of the lane's own row
            const int colb = n0 + wn * WW + (odd ? 32 : 0) + 8 * g4;
            T* cp = reinterpret_cast<T*>(p.C) + (size_t)rowp * p.ldc + colb;
            const int col5 = n0 + wn * WW + 64 + 4 * g4;
            T* cp5 = reinterpret_cast<T*>(p.C) + (size_t)row0 * p.ldc + col5;
            auto drain = [&](auto ACT) __attribute__((always_inline)) {
                uint2 w5 = make_uint2(0u, 0u);   // (NTW == 5) the even sub-tile row's fifth sub-tile, waiting for its odd partner
                auto actf = [&](float v) __attribute__((always_inline)) {
                    if constexpr (decltype(ACT)::value == 1) return gelu_erfc5(v);   // results are rounded to bf16: the one-transcendental form (common.h), gated by tests/test_kernels_gpu.py
                    else if constexpr (decltype(ACT)::value == 2) return fmaxf(v, 0.f);
                    else return v;
                };
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    if (!(i < MI1 || i < mi)) continue;
                    unsigned int o[8];
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            o[4 * h + q] = pack_bf16x2(actf(acc[i][2 * h + (q >> 1)][2 * (q & 1)] + bv[8 * h + 2 * q]),
                                                       actf(acc[i][2 * h + (q >> 1)][2 * (q & 1) + 1] + bv[8 * h + 2 * q + 1]));
                    u32x4 s0, s1;
                    pair_swap((u32x4){o[0], o[1], o[2], o[3]}, (u32x4){o[4], o[5], o[6], o[7]}, odd, s0, s1);
                    if (rowp + 16 * i < p.M && colb < p.N) {   // M is even: both rows of the pair or neither
                        gst(cp + (size_t)(16 * i) * p.ldc, s0);
                        gst(cp + (size_t)(16 * i + 1) * p.ldc, s1);
                    }
                    if constexpr (NTW == 5) {
                        // the fifth sub-tile: 4 columns (8 bytes) of the lane's own row.  Sub-tile rows (i, i + 1) trade across the 16-lane rows
                        // of the wave (v_permlane16_swap: odd rows of its first operand <-> even rows of its second): lane (mm, g4) ends up with
                        // columns 8 (g4 >> 1) .. +7 of row 16 (i + (g4 & 1)) + mm -> ONE 16-byte store per lane and pair instead of two 8-byte ones
                        // (the tile's store tail is issue-bound: profiles/r03_ring_stamps.txt, 8.0-8.3 k cycles against 4.3-4.9 k at 272x256)
                        uint2 w;
                        w.x = pack_bf16x2(actf(acc[i][4][0] + bv[16]), actf(acc[i][4][1] + bv[17]));
                        w.y = pack_bf16x2(actf(acc[i][4][2] + bv[18]), actf(acc[i][4][3] + bv[19]));
                        if constexpr (MI0 == MI1 && MI % 2 == 0) {
                            if (i & 1) {
                                const auto rx = __builtin_amdgcn_permlane16_swap(w5.x, w.x, false, false);
                                const auto ry = __builtin_amdgcn_permlane16_swap(w5.y, w.y, false, false);
                                const int r5 = row0 + 16 * (i - 1 + (g4 & 1));
                                if (r5 < p.M && col5 < p.N)
                                    gst(reinterpret_cast<T*>(p.C) + (size_t)r5 * p.ldc + (n0 + wn * WW + 64 + 8 * (g4 >> 1)), (u32x4){rx[0], ry[0], rx[1], ry[1]});
                            } else {
                                w5 = w;
                            }
                        } else {
                            if (row0 + 16 * i < p.M && col5 < p.N) *reinterpret_cast<uint2*>(cp5 + (size_t)(16 * i) * p.ldc) = w;
                        }
                    }
                }
            };
            if (p.act == 1) drain(std::integral_constant<int, 1>{});
            else if (p.act == 2) drain(std::integral_constant<int, 2>{});
            else drain(std::integral_constant<int, 0>{});
        } else if (NTW == 4 && direct && !p.out_f32 && p.act == 3) {
            // SwiGLU: out[:, n0/2 + 64 (wn >> 1) + 32 (wn & 1) + 8 g4 + e] = silu(gate_e) * up_e, gate = sub-tiles 0 / 1, up = 2 / 3 (modeling_internlm2.py:261-264)
            T* cp = reinterpret_cast<T*>(p.C) + (size_t)row0 * p.ldc + (n0 >> 1) + (wn >> 1) * 64 + (wn & 1) * 32 + 8 * g4;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                if (!(i < MI1 || i < mi)) continue;
                unsigned int o[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float g0 = acc[i][q >> 1][2 * (q & 1)], g1 = acc[i][q >> 1][2 * (q & 1) + 1];
                    const float u0 = acc[i][2 + (q >> 1)][2 * (q & 1)], u1 = acc[i][2 + (q >> 1)][2 * (q & 1) + 1];
                    // silu = g * rcp(1 + exp(-g)) with the hardware reciprocal (1 ulp) instead of the correctly rounded division (ten instructions): the
                    // result is rounded to bf16 two lines below (gated to 1e-5 of the exact form by tests/test_kernels_gpu.py)
                    o[q] = pack_bf16x2(g0 * __builtin_amdgcn_rcpf(1.0f + __expf(-g0)) * u0, g1 * __builtin_amdgcn_rcpf(1.0f + __expf(-g1)) * u1);
                }
                if (row0 + 16 * i < p.M) gst(cp + (size_t)(16 * i) * p.ldc, (u32x4){o[0], o[1], o[2], o[3]});
            }
        } else if (direct && p.out_f32 && p.act == 0) {
            // fp32 residual stream: C = acc + bias + residual[row (mod res_row_mod)]; sub-tiles (2 jp, 2 jp + 1) are 128 contiguous bytes of a row
            // -> pair_swap; the fifth sub-tile of a 320-wide tile is 16 bytes of the lane's own row.  Two register sets alternate so that the
            // residual rows of sub-tile i + 1 are requested before sub-tile i is stored
            const int colp = n0 + wn * WW + (odd ? 16 : 0) + 4 * g4;
            float* cp = reinterpret_cast<float*>(p.C) + (size_t)rowp * p.ldc + colp;
            const int col5 = n0 + wn * WW + 64 + 4 * g4;
            float* cp5 = reinterpret_cast<float*>(p.C) + (size_t)row0 * p.ldc + col5;
            auto load_res = [&](int i, float4 (&r)[5]) __attribute__((always_inline)) {   // r[2 jp + rsel]: 128 B segment jp of row rowp + 16 i + rsel; r[4]: the fifth sub-tile
#pragma unroll
                for (int rsel = 0; rsel < 2; ++rsel) {
                    const int gm = min(rowp + 16 * i + rsel, p.M - 1);
                    const int rr = p.res_row_mod > 0 ? gm % p.res_row_mod : gm;
                    const float* rp = p.residual + (size_t)rr * p.ldr + colp;
#pragma unroll
                    for (int jp = 0; jp < 2; ++jp) r[2 * jp + rsel] = colp + 32 * jp < p.N ? *reinterpret_cast<const float4*>(rp + 32 * jp) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                if constexpr (NTW == 5) {
                    const int gm = min(row0 + 16 * i, p.M - 1);
                    const int rr = p.res_row_mod > 0 ? gm % p.res_row_mod : gm;
                    r[4] = col5 < p.N ? *reinterpret_cast<const float4*>(p.residual + (size_t)rr * p.ldr + col5) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            };
            auto put = [&](int i, const float4 (&r)[5]) __attribute__((always_inline)) {
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    u32x4 lo, hi, s0, s1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        lo[e] = __float_as_uint(acc[i][2 * jp][e] + bv[8 * jp + e]);
                        hi[e] = __float_as_uint(acc[i][2 * jp + 1][e] + bv[8 * jp + 4 + e]);
                    }
                    pair_swap(lo, hi, odd, s0, s1);
                    const float4 r0 = r[2 * jp], r1 = r[2 * jp + 1];
                    s0 = (u32x4){__float_as_uint(__uint_as_float(s0[0]) + r0.x), __float_as_uint(__uint_as_float(s0[1]) + r0.y),
                                 __float_as_uint(__uint_as_float(s0[2]) + r0.z), __float_as_uint(__uint_as_float(s0[3]) + r0.w)};
                    s1 = (u32x4){__float_as_uint(__uint_as_float(s1[0]) + r1.x), __float_as_uint(__uint_as_float(s1[1]) + r1.y),
                                 __float_as_uint(__uint_as_float(s1[2]) + r1.z), __float_as_uint(__uint_as_float(s1[3]) + r1.w)};
                    if (rowp + 16 * i < p.M && colp + 32 * jp < p.N) {
                        *reinterpret_cast<u32x4*>(cp + (size_t)(16 * i) * p.ldc + 32 * jp) = s0;
                        *reinterpret_cast<u32x4*>(cp + (size_t)(16 * i + 1) * p.ldc + 32 * jp) = s1;
                    }
                }
                if constexpr (NTW == 5) {
                    if (row0 + 16 * i < p.M && col5 < p.N)
                        *reinterpret_cast<float4*>(cp5 + (size_t)(16 * i) * p.ldc) = make_float4(acc[i][4][0] + bv[16] + r[4].x, acc[i][4][1] + bv[17] + r[4].y,
                                                                                                 acc[i][4][2] + bv[18] + r[4].z, acc[i][4][3] + bv[19] + r[4].w);
                }
            };
            auto drain = [&](auto RES) __attribute__((always_inline)) {
                float4 ra[5], rb[5];
#pragma unroll
                for (int j = 0; j < 5; ++j) ra[j] = rb[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (decltype(RES)::value) load_res(0, ra);
#pragma unroll
                for (int i = 0; i < MI; i += 2) {
                    if constexpr (decltype(RES)::value) { if (i + 1 < MI) load_res(i + 1, rb); __builtin_amdgcn_sched_barrier(0); }
                    if (i < MI1 || i < mi) put(i, ra);
                    if constexpr (decltype(RES)::value) { if (i + 2 < MI) load_res(i + 2, ra); __builtin_amdgcn_sched_barrier(0); }
                    if (i + 1 < MI && (i + 1 < MI1 || i + 1 < mi)) put(i + 1, rb);
                }
            };
            if (p.residual) drain(std::true_type{}); else drain(std::false_type{});
        } else {
            // every other epilogue (RoPE, activations on fp32 outputs, unaligned / odd shapes): through the LDS, one wave row at a time
            float* Cs = reinterpret_cast<float*>(smem);
            auto ncol = [&](int j) __attribute__((always_inline)) {   // column of sub-tile j's register 0 inside the TILE (the inverse of w_row)
                if (perm == 1) return j < 4 ? WW * wn + 32 * (j >> 1) + 8 * g4 + 4 * (j & 1) : WW * wn + 64 + 4 * g4;
                if (perm == 2) return 128 * (wn >> 1) + 64 * (j >> 1) + 32 * (wn & 1) + 8 * g4 + 4 * (j & 1);
                return WW * wn + 16 * j + 4 * g4;
            };
            auto staged = [&](auto half_c) __attribute__((always_inline)) {
                constexpr int half = decltype(half_c)::value;
                constexpr int ROWS = 16 * (half == 0 ? MI0 : MI1);
                constexpr int NT_E = (512 / (BN / 8)) * (BN / 8);   // whole rows per pass: 320 columns -> 40 threads per row, 480 threads, 12 rows
                __syncthreads();
                if (wm == half) {
#pragma unroll
                    for (int i = 0; i < (half == 0 ? MI0 : MI1); ++i)
#pragma unroll
                        for (int j = 0; j < NTW; ++j)
                            *reinterpret_cast<float4*>(Cs + (i * 16 + mm) * BN + ncol(j)) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
                }
                __syncthreads();
                if (p.out_f32) epilogue_rows<T, float, ROWS, NT_E, BN>(p, Cs, m0 + half * 16 * MI0, n0, tn, tid);
                else epilogue_rows<T, T, ROWS, NT_E, BN>(p, Cs, m0 + half * 16 * MI0, n0, tn, tid);
            };
            staged(std::integral_constant<int, 0>{});
            staged(std::integral_constant<int, 1>{});
        }
    }
    stamp(2);
    if (STAMP && p.dbg && (tid & 255) == 0) p.dbg[((size_t)bid * 2 + grp) * 4 + 3] = t_entry;
}

template <int MI0, int MI1, int NTW, int EMODE = 0>
static int launch_gemm_ring8(GemmArgs a, hipStream_t stream) {
    constexpr int BM = 16 * (MI0 + MI1), BN = 64 * NTW;
    constexpr int LDS = (4 * (BM + BN) * 64 > 16 * MI0 * BN * 4) ? 4 * (BM + BN) * 64 : 16 * MI0 * BN * 4;   // the ring, or the epilogue's staging rows
    static_assert(LDS <= 163840, "160 KiB of LDS per CU");
    static PerDeviceOnce attr_set;
    if (attr_set.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ring8_kernel<MI0, MI1, NTW, false, EMODE>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if constexpr (EMODE == 0) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ring8_kernel<MI0, MI1, NTW, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    }
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.N + BN - 1) / BN;
    a.full_tiles = a.tiles_m * a.tiles_n;
    a.ksplit = 1;
    if constexpr (EMODE == 0) {
        if (a.dbg) {   // stamped diagnostic build (tools/probes/ring8_stamps.py)
            gemm_ring8_kernel<MI0, MI1, NTW, true><<<dim3(a.full_tiles), dim3(512), LDS, stream>>>(a);
            ULLSAM_LAUNCH_CHECK();
            return 0;
        }
    }
    gemm_ring8_kernel<MI0, MI1, NTW, false, EMODE><<<dim3(a.full_tiles), dim3(512), LDS, stream>>>(a);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}
// Split-K on the ring kernel for launches of FEW tiles under a long sum (M = 1081 rows x N = 4096: 4 x 16 tiles of 272 x 256 -- a batch-1 prefill's wo / w2, the
// frozen LLM's products of a training step; the AMG encoder's lin2 at one image): S = 256 / tiles K-ranges run as S x tiles workgroups of ONE launch, each writing its
// fp32 sums to a plane of the caller's workspace; splitk_finish_kernel adds the planes in order (deterministic) with bias / residual and writes C.
template <typename TO>
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ ws, TO* __restrict__ C, long ldc, const float* __restrict__ bias,
                                                            const float* __restrict__ residual, long ldr, int res_row_mod, int M, int N, int S) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x, nq = N >> 2;
    if (q >= (long)M * nq) return;
    const int m = (int)(q / nq), c = (int)(q - (long)m * nq) * 4;
    const size_t plane = (size_t)M * N;
    const float* src = ws + (size_t)m * N + c;
    float4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) if (k < S) v[k] = *reinterpret_cast<const float4*>(src + k * plane);   // S <= 8 independent loads, then the sum in plane order
    float4 s = v[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) if (k < S) { s.x += v[k].x; s.y += v[k].y; s.z += v[k].z; s.w += v[k].w; }
    if (bias) { const float4 b = *reinterpret_cast<const float4*>(bias + c); s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w; }
    if (residual) {
        const int rr = res_row_mod > 0 ? m % res_row_mod : m;
        const float4 r = *reinterpret_cast<const float4*>(residual + (size_t)rr * ldr + c);
        s.x += r.x; s.y += r.y; s.z += r.z; s.w += r.w;
    }
    if constexpr (sizeof(TO) == 4) *reinterpret_cast<float4*>(C + (size_t)m * ldc + c) = s;
    else *reinterpret_cast<uint2*>(C + (size_t)m * ldc + c) = make_uint2(pack_bf16x2(s.x, s.y), pack_bf16x2(s.z, s.w));
}
template <int MI0, int MI1, int NTW>
static int launch_gemm_ring_splitk(const GemmArgs& a, int S, hipStream_t stream) {
    constexpr int BM = 16 * (MI0 + MI1), BN = 64 * NTW;
    constexpr int LDS = (4 * (BM + BN) * 64 > 16 * MI0 * BN * 4) ? 4 * (BM + BN) * 64 : 16 * MI0 * BN * 4;
    static PerDeviceOnce attr_set;
    if (attr_set.first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ring8_kernel<MI0, MI1, NTW, false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    GemmArgs b = a;
    b.C = a.ws; b.ldc = a.N; b.out_f32 = 1; b.bias = nullptr; b.residual = nullptr; b.act = 0; b.vec_ok = 1; b.dbg = nullptr;
    b.tiles_m = (a.M + BM - 1) / BM;
    b.tiles_n = (a.N + BN - 1) / BN;
    b.full_tiles = b.tiles_m * b.tiles_n;
    b.ksplit = 1;
    gemm_ring8_kernel<MI0, MI1, NTW, false, 2><<<dim3(b.full_tiles, S), dim3(512), LDS, stream>>>(b);
    ULLSAM_LAUNCH_CHECK();
    const long quads = (long)a.M * (a.N >> 2);
    if (a.out_f32) splitk_finish_kernel<float><<<dim3((unsigned)((quads + 255) / 256)), 256, 0, stream>>>(a.ws, static_cast<float*>(a.C), a.ldc, a.bias, a.residual, a.ldr, a.res_row_mod, a.M, a.N, S);
    else splitk_finish_kernel<bf16><<<dim3((unsigned)((quads + 255) / 256)), 256, 0, stream>>>(a.ws, static_cast<bf16*>(a.C), a.ldc, a.bias, a.residual, a.ldr, a.res_row_mod, a.M, a.N, S);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}
static int g_ring_splitk = 1;  // ullsam_set_gemm_tuning(3, v): 1 (default) = where the caller allows it (act | 256), 0 = never, 2 = wherever the plan below fits (tests, A/B).
//                                A launch cut into K ranges sums in another order than the one-launch kernels, which all add the 32-deep k steps in sequence: the inference
//                                modules never ask for it, so a batch of images gives each image the bits it gets alone (tests/test_model_gpu.py: batched == per image)
// -> the number of K ranges (0: not a split-K launch) and the tile shape (6 / 8 / 9 = 256x256 / 256x320 / 272x256) for a launch the ring would leave mostly idle
static int ring_splitk_plan(const GemmArgs& a, int* shape) {
    if (a.act != 0 || !a.vec_ok || !a.ws || (a.N & 7) != 0 || a.M < 1024 || a.N < 256 || a.dbg) return 0;
    const long t272 = (long)((a.M + 271) / 272) * ((a.N + 255) / 256), t256 = (long)((a.M + 255) / 256) * ((a.N + 255) / 256);
    const long t320 = a.N % 320 == 0 ? (long)((a.M + 255) / 256) * (a.N / 320) : 1l << 40;
    // the shape that covers the problem with the least padded area (272-row tiles for 1081 = 4 x 272 - 7 rows; 256x320 for the ViT's 1280-wide outputs)
    const double a272 = 1.0625 * t272, a256 = 1.0 * t256, a320 = 1.25 * t320;
    long t; if (a320 <= a272 && a320 <= a256) { t = t320; *shape = 8; } else if (a272 < a256) { t = t272; *shape = 9; } else { t = t256; *shape = 6; }
    if (t > 128) return 0;
    int S = (int)(256 / t); if (S > 8) S = 8;
    while (S >= 2 && (a.K % (64 * S) != 0 || a.K / S < 1280 || (size_t)S * a.M * a.N * 4 > a.ws_bytes)) --S;
    // measured (tools/probes/splitk_ab.py, against the 128x128 kernel): 1081 x 4096 x K 14336 -35 %, K 6144 -17 %, 4096 x 1280 x 5120 -7 %; K 4096 in four ranges of 1024 equal, and
    // 96 tiles x 2 ranges (1081 x 6144 x 4096: 192 workgroups) +15 % -- hence ranges of >= 1280 and >= 224 workgroups
    return (S >= 2 && S * t >= 224) ? S : 0;
}
#include "gemm_ring8p.h"
// Persistent form (gemm_ring8p.h) for launches of MORE than one round of tiles whose shape and epilogue it takes; everything else -- one-round
// launches, ragged N, K not a multiple of 128, the LDS-staged epilogues -- stays on the one-tile-per-workgroup kernel.
static int g_persist = 2;      // ullsam_set_gemm_tuning(2, v): 0 one tile per workgroup; 2 (default) persistent ring, uniform trips, both groups' epilogues together (-0.5 % of the bench step, outputs bit-equal); 1 / 4 / 5 - 7: the earlier persistent forms and the ablations
static int cu_count() {
    static int n[32] = {};
    int d = 0;
    (void)hipGetDevice(&d);
    d &= 31;
    if (!n[d]) { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || v <= 0) v = 256; n[d] = v; }
    return n[d];
}
template <int BM, int BN, int NTW, int EMODE>
static bool ring_persist_ok(const GemmArgs& a) {
    if (!g_persist || a.N % BN != 0 || a.K % 128 != 0 || a.K < 384 || (a.dbg && !(g_dbg & 2))) return false;
    const long tiles = (long)((a.M + BM - 1) / BM) * (a.N / BN);
    if (tiles <= cu_count()) return false;
    if (((size_t)(a.M + BM) * a.lda + a.K) * 2 >= (1ull << 32) || ((size_t)a.N * a.ldw + a.K) * 2 >= (1ull << 32)) return false;   // 32-bit descriptor offsets
    if (EMODE == 1) return true;                                      // (the RoPE launcher has checked the alignment of q / k / v and the bias)
    if (!(a.vec_ok && (a.N & 7) == 0 && (a.M & 1) == 0)) return false;   // the direct epilogues' condition in gemm_ring8_kernel
    if (!a.out_f32) return a.act != 3 || NTW == 4;
    return a.act == 0;
}
template <int MI0, int MI1, int NTW, int EMODE = 0>
static int launch_ring(const GemmArgs& a, hipStream_t stream) {
    if (ring_persist_ok<16 * (MI0 + MI1), 64 * NTW, NTW, EMODE>(a)) {
        if (g_persist == 2) return launch_gemm_ring8p<MI0, MI1, NTW, EMODE, 2, 2>(a, stream, cu_count());   // uniform trips, both groups' epilogues at the same time
        if constexpr (EMODE == 0) { if (g_persist == 4) return launch_gemm_ring8p<MI0, MI1, NTW, EMODE, 2, 1>(a, stream, cu_count()); }   // one barrier per stage
        if constexpr (EMODE == 0 && MI0 == 9) {   // ablations of the 272x256 loop (wrong results: timing only)
            if (g_persist == 5) return launch_gemm_ring8p<MI0, MI1, NTW, EMODE, 2, 0, 1>(a, stream, cu_count());
            if (g_persist == 6) return launch_gemm_ring8p<MI0, MI1, NTW, EMODE, 2, 0, 2>(a, stream, cu_count());
            if (g_persist == 7) return launch_gemm_ring8p<MI0, MI1, NTW, EMODE, 2, 0, 3>(a, stream, cu_count());
        }
        return launch_gemm_ring8p<MI0, MI1, NTW, EMODE>(a, stream, cu_count());
    }
    return launch_gemm_ring8<MI0, MI1, NTW, EMODE>(a, stream);
}
static int launch_gemm_v6(const GemmArgs& a, hipStream_t stream) { return launch_ring<8, 8, 4>(a, stream); }   // 256 x 256
static int launch_gemm_v8(const GemmArgs& a, hipStream_t stream) { return launch_ring<8, 8, 5>(a, stream); }   // 256 x 320
static int launch_gemm_v9(const GemmArgs& a, hipStream_t stream) { return launch_ring<9, 8, 4>(a, stream); }   // 272 x 256
// The wqkv GEMM with the RoPE epilogue (act 4) on the ring kernel: the tile height among 208 / 256 / 272 rows that covers the problem in the
// fewest tile-rounds of the 256 CUs (cost of a tile = its area).  The bench's 4324 x 6144 launch: 256x256 = 408 tiles = 1.59 rounds run as 2
// (cost 2.0, 215 us on the two-buffer kernel in round 3), 272x256 = 384 tiles (2.125), 208x256 = 504 tiles = 1.97 rounds (1.625).
static int launch_gemm_ring_rope(const GemmArgs& a, hipStream_t stream, int force) {
    auto cost = [&](int bm) { const long t = (long)((a.M + bm - 1) / bm) * ((a.N + 255) / 256); return (double)((t + 255) / 256) * bm / 256.0; };
    const double c208 = cost(208), c256 = cost(256), c272 = cost(272);
    int pick = (c208 < c256 && c208 < c272) ? 208 : (c272 < c256 ? 272 : 256);
    if (force) pick = force;
    if (pick == 208) return launch_ring<7, 6, 4, 1>(a, stream);
    if (pick == 272) return launch_ring<9, 8, 4, 1>(a, stream);
    return launch_ring<8, 8, 4, 1>(a, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// fp8 (OCP e4m3) GEMM for the ViT's LayerNorm-fed linears (BASELINE configs[4], "fp8 MFMA ViT path"): the 256x256 staggered
// kernel with v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales; 2x the bf16 FLOP per clock).  A = activations quantised per
// ROW by the preceding LayerNorm kernel (norm.hip: ullsam_norm_fp8), W = weights quantised per OUTPUT CHANNEL once; the epilogue
// multiplies acc by scale_A[row] * scale_W[col] and continues as usual (bias, GELU, ...).  A K-tile is still 128 bytes per row
// = 128 fp8 elements = ONE MFMA k-step; its 32 MFMAs per wave are issued as two segments of 16 (sub-tile rows 0-3 / 4-7).
// Fragment: lane l holds row l & 15, k = 32 (l >> 4) .. +31 = two adjacent 16-byte chunks (checked with exact integer data:
// tools/probes/fp8_mfma_layout.hip).  LDS swizzle for that access: chunk c of row r sits at c ^ f(r), f(r) = (r & 6) | ((r >> 3) & 1)
// (the bf16 kernels' r & 7 would collide: lanes of one ds_read_b128 group differ by 2 in c here, not by 1).
// ---------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) int i32x8;
__device__ __forceinline__ int f8_swz(int row) { return (row & 6) | ((row >> 3) & 1); }
__device__ __forceinline__ i32x8 lds_frag8(const char* tile, int row, int g) {
    const int f = f8_swz(row);
    const int4 lo = *reinterpret_cast<const int4*>(tile + row * 128 + (((2 * g) ^ f) << 4));
    const int4 hi = *reinterpret_cast<const int4*>(tile + row * 128 + (((2 * g + 1) ^ f) << 4));
    return (i32x8){lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
}
__global__ __launch_bounds__(512) void gemm256f8_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE = 65536;
    const int nblk = p.tiles_m * p.tiles_n;
    const int bid = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
    const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int GM = 4;
    const int width = GM * p.tiles_n;
    const int group = swz / width;
    const int first_m = group * GM;
    const int gsize = min(p.tiles_m - first_m, GM);
    const int tm = first_m + (swz % width) % gsize;
    const int tn = (swz % width) / gsize;
    const int m0 = tm * 256, n0 = tn * 256;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3, grp = wave >> 2;
    const int g4 = lane >> 4, mm = lane & 15;

    const char* a_base = reinterpret_cast<const char*>(p.A) + (size_t)m0 * p.lda;   // 1 byte per element
    const char* b_base = reinterpret_cast<const char*>(p.W) + (size_t)n0 * p.ldw;
    unsigned int a_off[4], b_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ f8_swz(row);
        a_off[i] = (unsigned int)((size_t)(min(m0 + row, p.M - 1) - m0) * p.lda) + (c << 4);
        b_off[i] = (unsigned int)((size_t)(min(n0 + row, p.N - 1) - n0) * p.ldw) + (c << 4);
    }
    const int nk = p.K >> 7;
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * STAGE;
        const char* ak = a_base + (size_t)kt * 128;
        const char* bk = b_base + (size_t)kt * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds(GLB_PTR(ak + a_off[i]), LDS_PTR(base + (wave * 4 + i) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GLB_PTR(bk + b_off[i]), LDS_PTR(base + 32768 + (wave * 4 + i) * 1024), 16, 0, 0);
        }
    };
    const int one = 0x7f7f7f7f;  // E8M0 block scales of 2^0
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < nk; ++kt) {   // slots L0 | C0 | L1 | C1 as in gemm256_kernel (same RAW / WAR argument)
        const char* Ab = smem + (kt & 1) * STAGE;
        const char* Bb = Ab + 32768;
        i32x8 a[4], b[4];
        if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = lds_frag8(Bb, wn * 64 + j * 16 + mm, g4);
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = lds_frag8(Ab, wm * 128 + i * 16 + mm, g4);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[i], b[j], acc[i][j], 0, 0, 0, one, 0, one);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = lds_frag8(Ab, wm * 128 + (4 + i) * 16 + mm, g4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[4 + i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[i], b[j], acc[4 + i][j], 0, 0, 0, one, 0, one);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();

    float* Cs = reinterpret_cast<float*>(smem);  // [128][256] fp32, one 128-row half at a time
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        __syncthreads();
        if (wm == half) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) Cs[(i * 16 + 4 * g4 + r) * 256 + wn * 64 + j * 16 + mm] = acc[i][j][r];
        }
        __syncthreads();
        if (p.out_f32) epilogue_rows<bf16, float, 128, 512, 256, 2>(p, Cs, m0 + half * 128, n0, tn, tid);
        else epilogue_rows<bf16, bf16, 128, 512, 256, 2>(p, Cs, m0 + half * 128, n0, tn, tid);
    }
}

// A8 [M, lda] / W8 [N, ldw] e4m3 bytes, a_scale fp32 [M], w_scale fp32 [N]; C bf16 (or fp32) [M, ldc] = act(acc * a_scale * w_scale + bias) (+ residual)
extern "C" int ullsam_gemm_fp8(const void* A8, long lda, const float* a_scale, const void* W8, long ldw, const float* w_scale, void* C,
                               long ldc, int out_f32, const float* bias, const float* residual, long ldr, int act, int M, int N, int K,
                               void* stream) {
    ULLSAM_CHECK(M > 0 && N > 0 && K > 0 && K % 128 == 0, "ullsam_gemm_fp8: K=%d must be a positive multiple of 128", K);
    ULLSAM_CHECK(((uintptr_t)A8 & 15) == 0 && ((uintptr_t)W8 & 15) == 0 && lda % 16 == 0 && ldw % 16 == 0, "ullsam_gemm_fp8: 16-byte aligned rows needed");
    ULLSAM_CHECK(a_scale && w_scale, "ullsam_gemm_fp8: scales are required");
    ULLSAM_CHECK(act >= 0 && act <= 2, "ullsam_gemm_fp8: bad act %d", act);
    GemmArgs a;
    a.A = A8; a.W = W8; a.C = C; a.bias = bias; a.residual = residual;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.ldr = ldr;
    a.res_row_mod = 0; a.act = act; a.out_f32 = out_f32;
    const int osz = out_f32 ? 4 : 2;
    bool vec = ((uintptr_t)C & 15) == 0 && (ldc * osz) % 16 == 0 && (N % 8 == 0);
    if (residual) vec = vec && ((uintptr_t)residual & 15) == 0 && (ldr % 4 == 0);
    a.group_m = 4;   // gemm256f8_kernel rasters with a fixed group height
    a.vec_ok = vec ? 1 : 0;
    a.ws = nullptr; a.ws_bytes = 0; a.ksplit = 1;
    a.row_scale = a_scale; a.col_scale = w_scale; a.dbg = nullptr;
    a.tiles_m = (M + 255) / 256; a.tiles_n = (N + 255) / 256; a.full_tiles = a.tiles_m * a.tiles_n;
    static PerDeviceOnce attr_set;
    if (attr_set.first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256f8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    gemm256f8_kernel<<<dim3(a.full_tiles), dim3(512), 131072, reinterpret_cast<hipStream_t>(stream)>>>(a);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// measurement knobs (not part of the reference's interface)
extern "C" int ullsam_set_gemm_tuning(int key, int value) {
    if (key == 0 && value >= 1 && value <= 1024) { g_group_m = value; return 0; }
    if (key == 1 && value >= 0 && value <= 7) { g_auto_mask = value; return 0; }
    if (key == 2 && value >= 0 && value <= 7 && value != 3) { g_persist = value; return 0; }
    if (key == 3 && value >= 0 && value <= 2) { g_ring_splitk = value; return 0; }
    ullsam_set_error("ullsam_set_gemm_tuning: unknown key %d / bad value %d", key, value);
    return -1;
}
extern "C" int ullsam_set_gemm_variant(int v) {
    g_gemm_variant = v & 15; g_split_tail = (v & 64) ? 0 : 1; g_dbg = (v >> 15) & 3;   // bit 15: stamp the one-tile ring kernel, bits 15 + 16: the persistent one
    return 0;
}

template <typename T, int EMODE = 0>
static int launch_gemm(GemmArgs a, hipStream_t stream) {
    static PerDeviceOnce attr_set;
    if (attr_set.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm128_kernel<T, EMODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    }
    // Split-K tail: with 2 workgroups per CU the chip runs 512 tiles per wave; when the last wave is at most half full, cut each
    // of its tiles into S K-ranges (fp32 partials + a reduce kernel).  Measured (tools/gemm_bench.py, variant 65 vs 1): pays only
    // when the K loop is long (w2, K=14336: -9 %); at K <= 5120 the reduce pass costs more than the partly empty wave it removes,
    // because real launches do not run in lock-step waves.
    const int T_ = a.tiles_m * a.tiles_n;
    const int C_ = 512;
    a.full_tiles = T_;
    a.ksplit = 1;
    const int nk = a.K / (128 / (int)sizeof(T));
    const int tail = T_ % C_;
    if (g_split_tail && a.ws && T_ > C_ && tail > 0 && tail <= C_ / 2 && nk >= 128) {
        int S = C_ / tail;
        if (S > 16) S = 16;
        if (S > nk / 4) S = nk / 4;
        if (S >= 2 && (size_t)tail * S * 65536 <= a.ws_bytes) {
            a.full_tiles = T_ - tail;
            a.ksplit = S;
        }
    }
    const int grid = a.full_tiles + (T_ - a.full_tiles) * a.ksplit;
    gemm128_kernel<T, EMODE><<<dim3(grid), dim3(256), 65536, stream>>>(a);
    ULLSAM_LAUNCH_CHECK();
    if (a.ksplit > 1) {
        gemm_tail_reduce_kernel<T, EMODE><<<dim3((T_ - a.full_tiles) * 4), dim3(256), 0, stream>>>(a);
        ULLSAM_LAUNCH_CHECK();
    }
    return 0;
}

// A weight row piece of the decode-step kernels: every byte of W is read once per token by ONE wave, so the load carries the non-temporal hint (nt: the line is not kept
// in the L2 / Infinity Cache at the expense of the activations, the KV cache and the next kernel's weights).  -DULLSAM_SKINNY_NT=0 builds the plain-load form for A/B.
#ifndef ULLSAM_SKINNY_NT
#define ULLSAM_SKINNY_NT 1
#endif
__device__ __forceinline__ uint4 load_w16(const bf16* ptr) {
#if ULLSAM_SKINNY_NT
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(ptr));
    return make_uint4(v[0], v[1], v[2], v[3]);
#else
    return *reinterpret_cast<const uint4*>(ptr);
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// Skinny GEMM for the decode step (M <= 8 rows, bf16): every weight byte is needed once and nothing is reused across rows of W, so
// this is a weight stream, not a tile problem -- the 128x128 kernel launches N/128 workgroups (32 for wo) and reads W at 0.27 TB/s.
// Here a wave owns four rows of W and walks K in 1 KiB steps per row (16 bytes per lane, four rows = four independent loads in
// flight per step, two steps unrolled); the M activation rows sit in LDS as bf16 and are read back 16 bytes per lane.  fp32
// accumulation per lane, one wave reduction per (row of W, row of A) at the end, then the usual epilogue (bias, GELU / ReLU,
// SwiGLU pair, fp32 residual) on lane 0.  For SwiGLU a wave takes two gate rows and their two up rows of the packed w13.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void bf16x8_to_f32(const uint4 u, float (&o)[8]) {
    const unsigned int d[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) { o[2 * j] = __uint_as_float(d[j] << 16); o[2 * j + 1] = __uint_as_float(d[j] & 0xffff0000u); }
}
// acc += a.lo * b.lo + a.hi * b.hi on packed bf16 pairs (fp32 accumulate; the products of two bf16 are exact in fp32).  gfx950 has no
// compiler builtin for this opcode, hence the asm; the compiler cannot see that it is a DOT instruction, so the wait states it would
// insert between a DOT result and a different VALU reader (3 on gfx90a and later) are supplied by dot_fence() after the last one.
__device__ __forceinline__ void dot2c(float& acc, const unsigned int a, const unsigned int b) {
    asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void dot_fence() { asm volatile("s_nop 4" ::: "memory"); }
// Sum each of the NV values of a lane over the 64 lanes; returns, in lane l, the total of value (l >> (6 - log2 NV)) ... see below.
// Halving butterfly: at the step with partner distance o a lane keeps half of its values (which half: bit o of the lane) and adds the
// partner's copy of the same half, so 16 values cost 8 + 4 + 2 + 1 exchanges and two more on the single value left, not 16 x 6.
// After it lane l holds the total of value index  bit5(l) * 8 + bit4(l) * 4 + bit3(l) * 2 + bit2(l)  (NV = 16), complete in every lane.
template <int NV>
__device__ __forceinline__ float wave_sum_many(float (&v)[NV]) {
    static_assert(NV == 16 || NV == 32, "");
    const int lane = threadIdx.x & 63;
    int o = 32;
#pragma unroll
    for (int n = NV; n > 1; n >>= 1, o >>= 1) {
        const bool up = (lane & o) != 0;
#pragma unroll
        for (int i = 0; i < n / 2; ++i) {
            const float lo = v[i], hi = v[i + n / 2];    // both read first: a select between two array elements must not become an indexed (scratch) access
            v[i] = (up ? hi : lo) + __shfl_xor(up ? lo : hi, o, 64);
        }
    }
    float t = v[0];
#pragma unroll
    for (; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    return t;
}

// Stage A = bf16(x * rsqrt(mean(x^2) + eps) * w) for MM rows of K <= 4096 fp32 elements into LDS ([MM][K] bf16): the RMSNorm that precedes
// wqkv and w13 in a decode layer (modeling_internlm2.py:75-89), folded into the GEMM's staging.  Same element-to-thread assignment and the
// same order of sums as norm_block_kernel<.., 4, 4> (float4 i*256 + t of a row, wave butterfly, waves in order), so the normalised row
// is the one the separate kernel writes.  `red` is 16 floats of LDS; ends with a barrier.
template <int MM>
struct NormRows { float4 v[MM][4]; float4 w[4]; };
// The loads of stage_rmsnorm, issued BEFORE the kernel's first weight loads: a wave's vector-memory results return in order, so rows requested
// behind the weight stream would not be usable until that whole stream had landed.
template <int MM>
__device__ __forceinline__ void stage_rmsnorm_load(NormRows<MM>& n, const GemmArgs& p, const bool active = true) {
    const int tid = threadIdx.x & 255, M = active ? p.M : 0, nq = p.K >> 10;   // K % 1024 == 0: float4 i*256 + tid exists iff i < K / 1024 (uniform)
    const long ldx = p.ldx;
    const float* xb = p.norm_x + tid * 4;
    const float* wb = p.norm_w + tid * 4;
#pragma unroll
    for (int m = 0; m < MM; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            n.v[m][i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < nq && m < M) n.v[m][i] = *reinterpret_cast<const float4*>(xb + m * ldx + i * 1024);
        }
#pragma unroll
    for (int i = 0; i < 4; ++i) n.w[i] = i < nq ? *reinterpret_cast<const float4*>(wb + i * 1024) : make_float4(0.f, 0.f, 0.f, 0.f);
}
template <int MM>
__device__ __forceinline__ void stage_rmsnorm_finish(const NormRows<MM>& n, char* smem, float* red, const GemmArgs& p, const bool active = true) {
    const int tid = threadIdx.x & 255, lane = tid & 63, wv = tid >> 6;
    const int K = p.K, nq = K >> 10;
    float ss[MM];
#pragma unroll
    for (int m = 0; m < MM; ++m) {
        ss[m] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) ss[m] += (n.v[m][i].x * n.v[m][i].x + n.v[m][i].y * n.v[m][i].y) + (n.v[m][i].z * n.v[m][i].z + n.v[m][i].w * n.v[m][i].w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {     // wave_sum of the MM rows side by side (MM exchanges in flight per step; per row the order of wave_sum)
        float t[MM];
#pragma unroll
        for (int m = 0; m < MM; ++m) t[m] = __shfl_xor(ss[m], o, 64);
#pragma unroll
        for (int m = 0; m < MM; ++m) ss[m] += t[m];
    }
    if (lane == 0 && active) {
#pragma unroll
        for (int m = 0; m < MM; ++m) red[wv * MM + m] = ss[m];
    }
    __syncthreads();
    bf16* dst = reinterpret_cast<bf16*>(smem) + tid * 4;
#pragma unroll
    for (int m = 0; m < MM; ++m) {
        const float tot = ((red[m] + red[MM + m]) + red[2 * MM + m]) + red[3 * MM + m];
        const float rstd = rsqrtf(tot / (float)K + p.norm_eps);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i >= nq || !active) continue;
            const float4 o = make_float4(n.v[m][i].x * rstd * n.w[i].x, n.v[m][i].y * rstd * n.w[i].y, n.v[m][i].z * rstd * n.w[i].z, n.v[m][i].w * rstd * n.w[i].w);
            store4(dst + (size_t)m * K + i * 1024, o);
        }
    }
    __syncthreads();
}

template <int MM, bool NORM = false>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // A as bf16 [MM][K]
    const int tid = threadIdx.x, lane = tid & 63;
    const int K = p.K, KC = K >> 3;                 // 16-byte chunks per row
    const bf16* A = reinterpret_cast<const bf16*>(p.A);
    const bf16* W = reinterpret_cast<const bf16*>(p.W);
    const long wid = (long)blockIdx.x * 4 + (tid >> 6);
    long rows[4];
    if (p.act == 3) {  // packed w13: 128-row blocks [64 gate | 64 up]; this wave: gate rows g, g+1 and their up rows
        const long blk = wid >> 5, wb = wid & 31;
        rows[0] = blk * 128 + 2 * wb; rows[1] = rows[0] + 1; rows[2] = rows[0] + 64; rows[3] = rows[0] + 65;
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) rows[r] = wid * 4 + r;
    }
    const bf16* wr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) wr[r] = W + (size_t)min(rows[r], (long)p.N - 1) * p.ldw + lane * 8;
    // the first two steps of the weight stream are requested before the activations are staged: the stream does not depend on them
    uint4 b0[4], b1[4], b2[4];
    auto fill = [&](uint4 (&b)[4], const int k) {
#pragma unroll
        for (int r = 0; r < 4; ++r) b[r] = load_w16(wr[r] + k);
    };
    if constexpr (NORM) {
        // Order pinned by the sched_barriers: the rows' loads, one step of weight loads, only then the arithmetic on the rows -- whose wait
        // (vmcnt counts in order) then leaves the weight loads in flight.  No branch between the loads and that wait: the compiler would
        // merge the branches' pending-load counts pessimistically and wait for the weights too.
        static_assert(MM == 4, "");
        NormRows<4> nrows;
        __shared__ float nred[16];
        stage_rmsnorm_load<4>(nrows, p);
        __builtin_amdgcn_sched_barrier(0);
        fill(b0, 0);
        __builtin_amdgcn_sched_barrier(0);
        stage_rmsnorm_finish<4>(nrows, smem, nred, p);
        fill(b1, 512 < K ? 512 : 0);
    } else {
        fill(b0, 0);
        fill(b1, 512 < K ? 512 : 0);
        for (int idx = tid; idx < MM * KC; idx += 256) {
            const int m = idx / KC, c = idx - m * KC;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (m < p.M) v = *reinterpret_cast<const uint4*>(A + (size_t)m * p.lda + c * 8);
            *reinterpret_cast<uint4*>(smem + ((size_t)m * KC + c) * 16) = v;
        }
        __syncthreads();
    }
    if (rows[0] >= p.N) return;
    float acc[4 * MM];
#pragma unroll
    for (int i = 0; i < 4 * MM; ++i) acc[i] = 0.f;
    const char* xa = smem + lane * 16;
    // Three register buffers of four 1 KiB row segments rotate: two steps of loads are in flight while the third is multiplied.  The dot
    // products are opaque asm to the scheduler, which would otherwise serialise load -> wait -> 16 dots with one buffer; the
    // sched_barriers pin "issue the loads, then compute".
    auto step = [&](const uint4 (&b)[4], const int k) {
        uint4 av[MM];
#pragma unroll
        for (int m = 0; m < MM; ++m) av[m] = *reinterpret_cast<const uint4*>(xa + ((size_t)m * KC + (k >> 3)) * 16);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int m = 0; m < MM; ++m) {
                dot2c(acc[r * MM + m], b[r].x, av[m].x); dot2c(acc[r * MM + m], b[r].y, av[m].y);
                dot2c(acc[r * MM + m], b[r].z, av[m].z); dot2c(acc[r * MM + m], b[r].w, av[m].w);
            }
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int k0 = 0; k0 < K; k0 += 1536) {
        if (k0 + 1024 < K) fill(b2, k0 + 1024);
        step(b0, k0);
        if (k0 + 512 >= K) break;
        if (k0 + 1536 < K) fill(b0, k0 + 1536);
        step(b1, k0 + 512);
        if (k0 + 1024 >= K) break;
        if (k0 + 2048 < K) fill(b1, k0 + 2048);
        step(b2, k0 + 1024);
    }
    dot_fence();
    const float tot = wave_sum_many<4 * MM>(acc);       // lane l: the total of (r, m) = divmod(l >> SH, MM), complete in every lane
    constexpr int SH = MM == 4 ? 2 : 1;
    const int idx = lane >> SH, r = idx / MM, m = idx - r * MM;
    if (p.act == 3) {
        const float upv = __shfl_xor(tot, 32, 64);       // rows 2, 3 (the up rows) live 32 lanes above rows 0, 1
        if ((lane & ((1 << SH) - 1)) || lane >= 32 || m >= p.M) return;
        const long oc = (rows[0] >> 7) * 64 + (rows[0] & 63) + r;  // output column of gate row rows[r]
        const float o = silu_f(tot) * upv;
        if (p.out_f32) reinterpret_cast<float*>(p.C)[(size_t)m * p.ldc + oc] = o;
        else reinterpret_cast<bf16*>(p.C)[(size_t)m * p.ldc + oc] = (bf16)o;
        return;
    }
    const long n = rows[0] + r;
    if ((lane & ((1 << SH) - 1)) || n >= p.N || m >= p.M) return;
    float v = tot + (p.bias ? p.bias[n] : 0.f);
    if (p.act == 1) v = gelu_erf(v);
    else if (p.act == 2) v = fmaxf(v, 0.f);
    if (p.residual) v += p.residual[(size_t)(p.res_row_mod > 0 ? m % p.res_row_mod : m) * p.ldr + n];
    if (p.out_f32) reinterpret_cast<float*>(p.C)[(size_t)m * p.ldc + n] = v;
    else reinterpret_cast<bf16*>(p.C)[(size_t)m * p.ldc + n] = (bf16)v;
}

// Wide weight matrices at M <= 4 (w13: 28672 rows, the LM head: 92553) as ONE resident workgroup per CU that walks its share of the
// 4-row units: the activations are staged (and normalised) once per CU instead of once per 16 rows, nothing is left for a second,
// partly filled round of workgroups, and a wave's stream of 1 KiB steps runs through the unit boundaries (the next unit's first steps are
// already in flight while the finished unit is reduced and stored).  W waves per workgroup, chosen by the launcher so that
// units = CUs x W x trips comes out even (w13: 7168 units = 256 x 7 x 4).
template <bool NORM>
__global__ __launch_bounds__(512) void gemm_skinny_persist_kernel(GemmArgs p, int units) {
    constexpr int MM = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // A as bf16 [4][K]
    __shared__ float nred[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, W = blockDim.x >> 6;
    const int K = p.K, KC = K >> 3, KS = K >> 9;
    const bf16* Wl = reinterpret_cast<const bf16*>(p.W) + lane * 8;
    const int stride = gridDim.x * W, u0 = blockIdx.x * W + wv;
    const int n_mine = u0 < units ? (units - 1 - u0) / stride + 1 : 0;
    const int total = n_mine * KS;
    const bool swiglu = p.act == 3;
    auto row0_of = [&](const int u) -> long { return swiglu ? (long)(u >> 5) * 128 + 2 * (u & 31) : (long)u * 4; };
    // fill cursor: (unit, k) of the next step to request.  Past the end the loads go to the matrix' first row (cache hits): no branch, so
    // the compiler's load counting stays exact and a wait for one buffer never includes the next
    int fu = u0, fk = 0, fleft = total;
    auto fill_next = [&](uint4 (&b)[4]) {
        const bool on = fleft > 0;
        const long r0 = on ? row0_of(fu) : 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long row = swiglu ? r0 + (r & 1) + (r >> 1) * 64 : r0 + r;
            b[r] = load_w16(Wl + (size_t)(on ? min(row, (long)p.N - 1) : 0) * p.ldw + (on ? fk : 0));
        }
        fk += 512;
        if (fk >= K) { fk = 0; fu += stride; }
        --fleft;
    };
    uint4 b0[4], b1[4], b2[4];
    if constexpr (NORM) {
        NormRows<4> nrows;
        stage_rmsnorm_load<4>(nrows, p, tid < 256);
        __builtin_amdgcn_sched_barrier(0);   // order pinned: see gemm_skinny_kernel
        fill_next(b0);
        __builtin_amdgcn_sched_barrier(0);
        stage_rmsnorm_finish<4>(nrows, smem, nred, p, tid < 256);
        fill_next(b1);
    } else {
        fill_next(b0);
        fill_next(b1);
        const bf16* A = reinterpret_cast<const bf16*>(p.A);
        for (int idx = tid; idx < MM * KC; idx += blockDim.x) {
            const int m = idx / KC, c = idx - m * KC;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (m < p.M) v = *reinterpret_cast<const uint4*>(A + (size_t)m * p.lda + c * 8);
            *reinterpret_cast<uint4*>(smem + ((size_t)m * KC + c) * 16) = v;
        }
        __syncthreads();
    }
    float acc[4 * MM];
#pragma unroll
    for (int i = 0; i < 4 * MM; ++i) acc[i] = 0.f;
    const char* xa = smem + lane * 16;
    int cu = u0, ck = 0;
    auto finish = [&]() {   // the unit's 16 sums: butterfly, epilogue on 16 lanes, accumulators back to zero
        dot_fence();
        const float tot = wave_sum_many<4 * MM>(acc);
        const int idx = lane >> 2, r = idx >> 2, m = idx & 3;
        const long r0 = row0_of(cu);
        if (swiglu) {
            const float upv = __shfl_xor(tot, 32, 64);
            if (!(lane & 3) && lane < 32 && m < p.M) {
                const long oc = (r0 >> 7) * 64 + (r0 & 63) + r;
                const float o = silu_f(tot) * upv;
                if (p.out_f32) reinterpret_cast<float*>(p.C)[(size_t)m * p.ldc + oc] = o;
                else reinterpret_cast<bf16*>(p.C)[(size_t)m * p.ldc + oc] = (bf16)o;
            }
        } else {
            const long n = r0 + r;
            if (!(lane & 3) && n < p.N && m < p.M) {
                float v = tot + (p.bias ? p.bias[n] : 0.f);
                if (p.act == 1) v = gelu_erf(v);
                else if (p.act == 2) v = fmaxf(v, 0.f);
                if (p.residual) v += p.residual[(size_t)(p.res_row_mod > 0 ? m % p.res_row_mod : m) * p.ldr + n];
                if (p.out_f32) reinterpret_cast<float*>(p.C)[(size_t)m * p.ldc + n] = v;
                else reinterpret_cast<bf16*>(p.C)[(size_t)m * p.ldc + n] = (bf16)v;
            }
        }
#pragma unroll
        for (int i = 0; i < 4 * MM; ++i) acc[i] = 0.f;
    };
    auto step = [&](const uint4 (&b)[4]) {
        uint4 av[MM];
#pragma unroll
        for (int m = 0; m < MM; ++m) av[m] = *reinterpret_cast<const uint4*>(xa + ((size_t)m * KC + (ck >> 3)) * 16);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int m = 0; m < MM; ++m) {
                dot2c(acc[r * MM + m], b[r].x, av[m].x); dot2c(acc[r * MM + m], b[r].y, av[m].y);
                dot2c(acc[r * MM + m], b[r].z, av[m].z); dot2c(acc[r * MM + m], b[r].w, av[m].w);
            }
        __builtin_amdgcn_sched_barrier(0);
        ck += 512;
        if (ck >= K) { finish(); ck = 0; cu += stride; }
    };
    for (int g = 0; g < total; g += 3) {
        fill_next(b2);
        step(b0);
        if (g + 1 >= total) break;
        fill_next(b0);
        step(b1);
        if (g + 2 >= total) break;
        fill_next(b1);
        step(b2);
    }
}

// Narrow weight matrices (wo, w2, wqkv at decode: N <= 8192) give the kernel above only ~4 waves per CU.  Here a workgroup owns
// four rows of W and its four waves split K; the activation rows are read straight from global memory (32..115 KB in total,
// cache-resident) instead of being staged per workgroup, so nothing but 256 bytes of LDS is needed and a CU holds many workgroups.
// NORM: the activations are RMSNorm(norm_x) * norm_w, staged as bf16 in LDS by stage_rmsnorm (K <= 4096, M <= 4) instead of read from
// global memory.  act == 4 (R == 8 only): the wqkv epilogue of a decode step -- the workgroup owns rows d .. d+3 and d+64 .. d+67 of one
// 128-row head slot, i.e. four rotate_half pairs, and writes q / the KV-cache rows directly (same arithmetic as the prefill epilogue).
template <int MM, int R, bool NORM>
__global__ __launch_bounds__(256) void gemm_skinny_ksplit_kernel(GemmArgs p) {
    static_assert(R * MM == 16 || R * MM == 32, "the butterfly reduces 16 or 32 values per lane");
    extern __shared__ __attribute__((aligned(16))) char smem[];   // NORM: A as bf16 [MM][K]
    __shared__ float red[4][R * MM];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int KQ = p.K >> 2;                          // this wave's K range (K % 2048 == 0)
    const bf16* A = reinterpret_cast<const bf16*>(p.A) + (size_t)wv * KQ + lane * 8;
    const bf16* W = reinterpret_cast<const bf16*>(p.W) + (size_t)wv * KQ + lane * 8;
    const bool rope = R == 8 && p.act == 4;
    const long n0 = rope ? (long)(blockIdx.x >> 4) * 128 + (blockIdx.x & 15) * 4 : (long)blockIdx.x * R;
    auto row_of = [&](const int r) -> long { return rope ? n0 + (r & 3) + (r >> 2) * 64 : n0 + r; };
    const bf16* wr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) wr[r] = W + (size_t)min(row_of(r), (long)p.N - 1) * p.ldw;
    const bf16* ar[MM];
#pragma unroll
    for (int m = 0; m < MM; ++m) ar[m] = A + (size_t)min(m, p.M - 1) * p.lda;
    float acc[R * MM];
#pragma unroll
    for (int i = 0; i < R * MM; ++i) acc[i] = 0.f;
    struct Buf { uint4 w[R], a[NORM ? 1 : MM]; };
    auto fill = [&](Buf& b, const int k) {
#pragma unroll
        for (int r = 0; r < R; ++r) b.w[r] = load_w16(wr[r] + k);
        if constexpr (!NORM) {
#pragma unroll
            for (int m = 0; m < MM; ++m) b.a[m] = *reinterpret_cast<const uint4*>(ar[m] + k);
        }
    };
    const char* xa = smem + ((size_t)wv * KQ + lane * 8) * 2;
    auto step = [&](const Buf& b, const int k) {
        uint4 av[MM];
#pragma unroll
        for (int m = 0; m < MM; ++m) {
            if constexpr (NORM) av[m] = *reinterpret_cast<const uint4*>(xa + ((size_t)m * p.K + k) * 2);
            else av[m] = b.a[m];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int m = 0; m < MM; ++m) {
                dot2c(acc[r * MM + m], b.w[r].x, av[m].x); dot2c(acc[r * MM + m], b.w[r].y, av[m].y);
                dot2c(acc[r * MM + m], b.w[r].z, av[m].z); dot2c(acc[r * MM + m], b.w[r].w, av[m].w);
            }
        __builtin_amdgcn_sched_barrier(0);
    };
    // Two register buffers rotate: one step of loads is in flight behind the one being multiplied (see gemm_skinny_kernel); the first loads are
    // requested before the activations are staged
    Buf b0, b1;
    if constexpr (NORM) {   // one step of weights in flight under the staging (the rows' registers + two steps would cost the third workgroup per CU)
        NormRows<4> nrows;
        __shared__ float nred[16];
        stage_rmsnorm_load<4>(nrows, p);
        __builtin_amdgcn_sched_barrier(0);   // order pinned: see gemm_skinny_kernel
        fill(b0, 0);
        __builtin_amdgcn_sched_barrier(0);
        stage_rmsnorm_finish<4>(nrows, smem, nred, p);
        fill(b1, 512 < KQ ? 512 : 0);
    } else {
        fill(b0, 0);
        if (512 < KQ) fill(b1, 512);
    }
    for (int k0 = 0; k0 < KQ; k0 += 1024) {
        step(b0, k0);
        if (k0 + 512 >= KQ) break;
        if (k0 + 1024 < KQ) fill(b0, k0 + 1024);
        step(b1, k0 + 512);
        if (k0 + 1536 < KQ) fill(b1, k0 + 1536);
    }
    dot_fence();
    {
        const float tot = wave_sum_many<R * MM>(acc);
        constexpr int SH = R * MM == 16 ? 2 : 1;
        if ((lane & ((1 << SH) - 1)) == 0) red[wv][lane >> SH] = tot;
    }
    __syncthreads();
    if (tid >= R * MM) return;
    const int r = tid / MM, m = tid - r * MM;
    if (m >= p.M) return;
    auto total = [&](const int i) { return (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]); };
    if (rope) {
        if (r >= 4) return;                            // thread (r, m): the pair (row n0 + r, row n0 + 64 + r) of token m
        const long n1 = n0 + r, n2 = n1 + 64;
        const float x1 = total(r * MM + m) + (p.bias ? p.bias[n1] : 0.f), x2 = total((r + 4) * MM + m) + (p.bias ? p.bias[n2] : 0.f);
        const int slot = (int)(n1 >> 7), d = (int)(n1 & 127);
        const int gs = p.rope_G + 2, kv = slot / gs, g = slot - kv * gs;
        const int b = m / p.rope_S, sq = m - b * p.rope_S;
        bf16* dst;
        float o1 = x1, o2 = x2;
        if (g == gs - 1) {
            dst = reinterpret_cast<bf16*>(p.rope_v) + (((long)b * p.rope_KVH + kv) * p.rope_cap + p.rope_pos0 + sq) * 128;
        } else {
            const int ps = min(max(p.rope_pos[m], 0), p.rope_rows - 1);
            const float* cp = p.rope_cos + (size_t)ps * 128 + d;
            const float* sp = p.rope_sin + (size_t)ps * 128 + d;
            o1 = x1 * cp[0] - x2 * sp[0];              // q_embed = q*cos + rotate_half(q)*sin, rotate_half = cat(-x2, x1)
            o2 = x2 * cp[64] + x1 * sp[64];
            dst = g == gs - 2 ? reinterpret_cast<bf16*>(p.rope_k) + (((long)b * p.rope_KVH + kv) * p.rope_cap + p.rope_pos0 + sq) * 128
                              : reinterpret_cast<bf16*>(p.rope_q) + (long)m * ((long)p.rope_KVH * p.rope_G * 128) + ((long)kv * p.rope_G + g) * 128;
        }
        dst[d] = (bf16)o1;
        dst[d + 64] = (bf16)o2;
        return;
    }
    const long n = n0 + r;
    if (n >= p.N) return;
    float v = total(tid);
    if (p.bias) v += p.bias[n];
    if (p.act == 1) v = gelu_erf(v);
    else if (p.act == 2) v = fmaxf(v, 0.f);
    if (p.residual) v += p.residual[(size_t)(p.res_row_mod > 0 ? m % p.res_row_mod : m) * p.ldr + n];
    if (p.out_f32) reinterpret_cast<float*>(p.C)[(size_t)m * p.ldc + n] = v;
    else reinterpret_cast<bf16*>(p.C)[(size_t)m * p.ldc + n] = (bf16)v;
}

static int launch_gemm_skinny(const GemmArgs& a, hipStream_t stream) {
    const bool normed = a.norm_x != nullptr;
    if (normed && (a.M > 4 || a.K > 4096 || a.K % 2048 != 0 || a.ldx % 4 != 0)) { ullsam_set_error("skinny GEMM: the fused RMSNorm prologue needs M <= 4, K <= 4096, K %% 2048 == 0 (M=%d K=%d)", a.M, a.K); return -1; }
    if (a.act == 4 && (a.M > 4 || a.K % 2048 != 0 || a.N % 128 != 0)) { ullsam_set_error("skinny GEMM: the RoPE epilogue needs M <= 4, K %% 2048 == 0 (M=%d K=%d)", a.M, a.K); return -1; }
    if (a.act == 4 || (a.act != 3 && a.N <= 8192 && a.K % 2048 == 0)) {  // narrow: K split over the waves of a workgroup
        const dim3 g4((unsigned)((a.N + 3) / 4)), g8((unsigned)((a.N + 7) / 8));
        const size_t lds = normed ? (size_t)4 * a.K * 2 : 0;
        if (normed) gemm_skinny_ksplit_kernel<4, 8, true><<<g8, 256, lds, stream>>>(a);
        else if (a.act == 4) gemm_skinny_ksplit_kernel<4, 8, false><<<g8, 256, 0, stream>>>(a);
        else if (a.M > 4) gemm_skinny_ksplit_kernel<8, 4, false><<<g4, 256, 0, stream>>>(a);
        else gemm_skinny_ksplit_kernel<4, 8, false><<<g8, 256, 0, stream>>>(a);   // (4 rows x 3 buffers, 4 x 2, 8 x 3 were measured within 1 % of this and removed)
        ULLSAM_LAUNCH_CHECK();
        return 0;
    }
    const int MM = a.M <= 4 ? 4 : 8;
    const size_t lds = (size_t)MM * a.K * 2;
    const long waves = a.act == 3 ? (long)a.N / 4 : ((long)a.N + 3) / 4;
    // measured (tools/probes/skinny_probe.py, us per launch): w13 47.4 -> 45.3 with two resident workgroups per CU (45.8 with one, 46.9 with
    // three); the LM head (92553 rows) 129 -> 137: its 5785 workgroups already amortise the ramp, so it keeps the kernel above
    if (MM == 4 && (normed || a.act == 3) && waves >= 4096 && lds + 256 <= 64 * 1024) {   // (+ the kernel's static nred[16]: K = 8192 would tip over the 64 KiB default)
        // one workgroup per CU; W waves each so that the units divide evenly (fewest idle wave-trips, ties to the larger W)
        static PerDeviceOnce cu_once; static int n_cu = 256;
        if (cu_once.first()) { int dev = 0; hipDeviceProp_t pr; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) n_cu = pr.multiProcessorCount; }
        const int n_wg = n_cu * 2;   // two resident workgroups per CU (one: 45.8 us, three: 46.9, two: 45.3 for w13)
        int W = 8; double best = 1e30;
        for (int w = 8; w >= 4; --w) {
            const long per = (long)n_wg * w, trips = (waves + per - 1) / per;
            const double waste = (double)(trips * per) / (double)waves;
            if (waste < best - 1e-9) { best = waste; W = w; }
        }
        if (normed) gemm_skinny_persist_kernel<true><<<n_wg, W * 64, lds, stream>>>(a, (int)waves);
        else gemm_skinny_persist_kernel<false><<<n_wg, W * 64, lds, stream>>>(a, (int)waves);
        ULLSAM_LAUNCH_CHECK();
        return 0;
    }
    const dim3 grid((unsigned)((waves + 3) / 4));
    static PerDeviceOnce attr4, attr8;
    if (normed) {
        gemm_skinny_kernel<4, true><<<grid, 256, lds, stream>>>(a);   // <= 32 KiB of LDS: no attribute needed
    } else if (MM == 4) {
        if (attr4.first()) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_skinny_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024); }
        gemm_skinny_kernel<4><<<grid, 256, lds, stream>>>(a);
    } else {
        if (attr8.first()) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_skinny_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024); }
        gemm_skinny_kernel<8><<<grid, 256, lds, stream>>>(a);
    }
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

struct RopeEpilogue { const int* pos; const float* cos; const float* sin; void* q; void* k; void* v; int S, KVH, G, cap, pos0, rows; };
struct NormPrologue { const float* x; long ldx; const float* w; float eps; };

static int gemm_impl(int dtype, const void* A, long lda, const void* W, long ldw, void* C, long ldc, int out_f32,
                     const float* bias, const float* residual, long ldr, int res_row_mod, int act, int M, int N,
                     int K, void* workspace, long ws_bytes, void* stream, const RopeEpilogue* rope, const NormPrologue* norm = nullptr) {
    ULLSAM_CHECK(dtype == ULLSAM_DT_F32 || dtype == ULLSAM_DT_BF16, "ullsam_gemm: bad dtype %d", dtype);
    ULLSAM_CHECK(M > 0 && N > 0 && K > 0, "ullsam_gemm: empty problem M=%d N=%d K=%d", M, N, K);
    const int esz = dtype == ULLSAM_DT_F32 ? 4 : 2;
    const int bk = 128 / esz;
    ULLSAM_CHECK(K % bk == 0, "ullsam_gemm: K=%d must be a multiple of %d", K, bk);
    ULLSAM_CHECK(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0, "ullsam_gemm: A/W must be 16-byte aligned");
    ULLSAM_CHECK((lda * esz) % 16 == 0 && (ldw * esz) % 16 == 0, "ullsam_gemm: lda/ldw rows must be 16-byte multiples");
    const bool split_ok = (act & 256) != 0;   // the caller accepts K ranges summed apart (ullsam_hip.h): set by the training step's frozen linears, never by the inference modules
    act &= 255;
    ULLSAM_CHECK(act >= 0 && act <= 4 && (act == 4) == (rope != nullptr), "ullsam_gemm: bad act %d", act);
    if (act == 3) ULLSAM_CHECK(N % 128 == 0 && !bias && !residual, "ullsam_gemm: swiglu needs N%%128==0, no bias/residual");
    GemmArgs a;
    a.rope_pos = nullptr; a.rope_cos = a.rope_sin = nullptr; a.rope_q = a.rope_k = a.rope_v = nullptr;
    a.rope_S = a.rope_KVH = a.rope_G = a.rope_cap = a.rope_pos0 = a.rope_rows = 0;
    if (rope) {
        a.rope_pos = rope->pos; a.rope_cos = rope->cos; a.rope_sin = rope->sin; a.rope_q = rope->q; a.rope_k = rope->k; a.rope_v = rope->v;
        a.rope_S = rope->S; a.rope_KVH = rope->KVH; a.rope_G = rope->G; a.rope_cap = rope->cap; a.rope_pos0 = rope->pos0; a.rope_rows = rope->rows;
    }
    a.norm_x = nullptr; a.norm_w = nullptr; a.norm_eps = 0.f; a.ldx = 0;
    if (norm) {
        ULLSAM_CHECK(norm->x && norm->w && ((uintptr_t)norm->x & 15) == 0 && ((uintptr_t)norm->w & 15) == 0 && norm->ldx % 4 == 0,
                     "ullsam_gemm: the RMSNorm prologue needs 16-byte aligned fp32 rows and weight");
        a.norm_x = norm->x; a.norm_w = norm->w; a.norm_eps = norm->eps; a.ldx = norm->ldx;
    }
    a.A = A; a.W = W; a.C = C; a.bias = bias; a.residual = residual;
    a.M = M; a.N = N; a.K = K;
    a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.ldr = ldr;
    a.res_row_mod = res_row_mod; a.act = act; a.out_f32 = out_f32;
    const int osz = out_f32 ? 4 : esz;
    const int n_out = act == 3 ? N / 2 : N;
    bool vec = ((uintptr_t)C & 15) == 0 && (ldc * osz) % 16 == 0 && (n_out % 8 == 0);
    if (residual) vec = vec && ((uintptr_t)residual & 15) == 0 && (ldr % 4 == 0);
    a.vec_ok = vec ? 1 : 0;
    a.ws = reinterpret_cast<float*>(workspace);
    a.ws_bytes = workspace ? (size_t)ws_bytes : 0;
    a.group_m = g_group_m;
    a.row_scale = nullptr;
    a.col_scale = nullptr;
    a.dbg = (g_dbg && workspace && ws_bytes >= (56l << 20)) ? reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(workspace) + (48l << 20)) : nullptr;
    a.ksplit = 1;
    a.tiles_m = (M + 127) / 128;
    a.tiles_n = (N + 127) / 128;
    a.full_tiles = a.tiles_m * a.tiles_n;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int variant = g_gemm_variant;
    // decode step: a handful of rows against the whole weight matrix
    // (the forced variants select TILE kernels; a launch with the RMSNorm prologue exists in the skinny kernels only and goes there whatever is forced)
    if ((variant == 0 || norm) && (act != 4 || (M <= 4 && K % 2048 == 0)) && dtype == ULLSAM_DT_BF16 && M <= 8 && K % 512 == 0 && lda % 8 == 0 && ldw % 8 == 0 &&
        (size_t)(M <= 4 ? 4 : 8) * K * 2 <= 144 * 1024 && (act != 3 || N % 128 == 0))
        return launch_gemm_skinny(a, s);
    ULLSAM_CHECK(!norm, "ullsam_gemm: the RMSNorm prologue exists in the decode-step kernels only (bf16, M <= 4, K <= 4096; got M=%d K=%d)", M, K);
    const long t256 = (long)((M + 255) / 256) * ((N + 255) / 256);
    // 256-row tiles run one per CU: use them when the last round of tiles is >= 74 % full (measured crossover, tools/gemm_bench.py:
    // 408 / 960 / 1280 / 1904 tiles win, 272 / 320 lose to the 128x128 kernel's finer granularity) or a split-K tail absorbs the sliver
    const double fill = (double)t256 / (double)(((t256 + 255) / 256) * 256);
    const long tail256 = t256 % 256;
    const bool v3_split = g_split_tail && workspace && t256 > 256 && tail256 > 0 && tail256 <= 64 && K >= 64 * bk;  // see launch_gemm_v3_impl
    const bool big = M >= 1024 && N >= 256 && K >= 8 * bk && (act != 3 || N % 256 == 0);  // short K / narrow N: a 256^2 tile's fixed cost or its empty half dominates
    // the ring kernels keep a lane's bias columns in registers, fetched as float4 at 4-column offsets: whole, aligned groups only
    const bool bias_v4 = !bias || (N % 4 == 0 && ((uintptr_t)bias & 15) == 0);
    const bool ring_ok = dtype == ULLSAM_DT_BF16 && act <= 3 && K % 64 == 0 && K >= 128 && bias_v4 && (act != 3 || N % 256 == 0);
    // wqkv + RoPE (act 4), bf16, >= 1024 rows: the ring kernel with the epilogue straight from the accumulators (variant 3 keeps the two-buffer
    // kernel with the LDS-staged epilogue for A/B and the bit-level comparison; variants 6 / 9 / 10 force the 256- / 272- / 208-row tile)
    if (act == 4 && dtype == ULLSAM_DT_BF16 && M >= 1024 && K % 64 == 0 && K >= 256 && (variant == 0 || variant == 6 || variant == 9 || variant == 10) &&
        (((uintptr_t)rope->q | (uintptr_t)rope->k | (uintptr_t)rope->v) & 15) == 0 && (!bias || ((uintptr_t)bias & 15) == 0))
        return launch_gemm_ring_rope(a, s, variant == 6 ? 256 : variant == 9 ? 272 : variant == 10 ? 208 : 0);
    if (variant == 6 || variant == 8 || variant == 9) {
        if (!ring_ok || (variant == 8 && act == 3)) { ullsam_set_error("ullsam_gemm: the ring kernels need bf16, K %% 64 == 0, K >= 128, no RoPE epilogue (256x320: no SwiGLU)"); return -1; }
        return variant == 6 ? launch_gemm_v6(a, s) : variant == 8 ? launch_gemm_v8(a, s) : launch_gemm_v9(a, s);
    }
    if (variant == 1) return a.act == 4 ? (dtype == ULLSAM_DT_F32 ? launch_gemm<float, 1>(a, s) : launch_gemm<bf16, 1>(a, s))
                                        : (dtype == ULLSAM_DT_F32 ? launch_gemm<float>(a, s) : launch_gemm<bf16>(a, s));
    if (variant == 3) {
        if (act == 3 && N % 256 != 0) { ullsam_set_error("ullsam_gemm: the 256x256 kernel's SwiGLU epilogue needs N %% 256 == 0"); return -1; }
        return dtype == ULLSAM_DT_F32 ? launch_gemm_v3<float>(a, s) : launch_gemm_v3<bf16>(a, s);
    }
    // Auto, bf16, >= 1024 rows: the ring tile shape that covers the problem in the fewest tile-rounds of the 256 CUs (a tile's cost = its
    // area in units of 256x256; a split-K tail of the two-buffer kernel counts as half a round).  256x320 needs N % 320 == 0 (ViT-H: 1280,
    // 3840, 5120 -- vit.proj / lin2 become ONE round instead of 1.25, vit.qkv 3 x 1.25 instead of 4); 272x256 makes the bench's 4 x 1081 =
    // 4324 prompt rows 16 tile rows (llm.wo / w2: one round instead of 256 tiles + a tail; llm.w13: 7 whole rounds instead of 7.44).
    if (ring_ok && variant == 0 && (g_ring_splitk == 2 || (g_ring_splitk == 1 && split_ok))) {   // few tiles under a long sum: K ranges of the ring kernel side by side (launch_gemm_ring_splitk)
        int shape = 0;
        const int S = ring_splitk_plan(a, &shape);
        if (S) return shape == 8 ? launch_gemm_ring_splitk<8, 8, 5>(a, S, s) : shape == 9 ? launch_gemm_ring_splitk<9, 8, 4>(a, S, s) : launch_gemm_ring_splitk<8, 8, 4>(a, S, s);
    }
    if (ring_ok && M >= 1024 && N >= 256 && K >= 256) {
        const double c256 = v3_split ? (double)(t256 / 256) + 0.5 : (double)((t256 + 255) / 256);   // 256x256 tiles (with the two-buffer kernel's split-K tail)
        const long t320 = (long)((M + 255) / 256) * (N / 320), t272 = (long)((M + 271) / 272) * ((N + 255) / 256);
        const double c320 = (N % 320 == 0 && act != 3 && (g_auto_mask & 2)) ? 1.25 * (double)((t320 + 255) / 256) : 1e9;
        const double c272 = (g_auto_mask & 4) ? 1.0625 * (double)((t272 + 255) / 256) : 1e9;
        if (c320 <= 0.96 * c256 || (act == 1 && c320 <= c256)) return launch_gemm_v8(a, s);   // (GELU: a tie in tile-rounds goes to 256x320, measured in round 2)
        if (c272 <= 0.96 * c256) return launch_gemm_v9(a, s);
        if ((g_auto_mask & 1) && big && fill >= 0.74 && !v3_split) return launch_gemm_v6(a, s);
    }
    const bool v3 = big && (fill >= 0.74 || v3_split);
    if (v3) return dtype == ULLSAM_DT_F32 ? launch_gemm_v3<float>(a, s) : launch_gemm_v3<bf16>(a, s);   // fp32, the RoPE epilogue, bf16 shapes no ring takes
    if (a.act == 4) return dtype == ULLSAM_DT_F32 ? launch_gemm<float, 1>(a, s) : launch_gemm<bf16, 1>(a, s);
    return dtype == ULLSAM_DT_F32 ? launch_gemm<float>(a, s) : launch_gemm<bf16>(a, s);
}

extern "C" int ullsam_gemm(int dtype, const void* A, long lda, const void* W, long ldw, void* C, long ldc, int out_f32,
                           const float* bias, const float* residual, long ldr, int res_row_mod, int act, int M, int N,
                           int K, void* workspace, long ws_bytes, void* stream) {
    ULLSAM_CHECK(act != 4, "ullsam_gemm: act 4 is ullsam_gemm_qkv_rope");
    return gemm_impl(dtype, A, lda, W, ldw, C, ldc, out_f32, bias, residual, ldr, res_row_mod, act, M, N, K, workspace, ws_bytes, stream, nullptr);
}

// Decode step: out = act(bf16(RMSNorm(x) * norm_w) @ W^T + bias) (+ residual) for M <= 4 rows of K <= 4096 fp32 elements -- the norm that precedes
// w13 (and, through ullsam_decode_qkv_rope, wqkv) in a layer, folded into the weight stream's staging instead of a launch of its own.
extern "C" int ullsam_gemm_rmsnorm(const float* x, long ldx, const float* norm_w, float eps, const void* W, long ldw, void* C, long ldc,
                                   int out_f32, const float* bias, const float* residual, long ldr, int act, int M, int N, int K, void* stream) {
    ULLSAM_CHECK(act != 4, "ullsam_gemm_rmsnorm: act 4 is ullsam_decode_qkv_rope");
    NormPrologue n{x, ldx, norm_w, eps};
    return gemm_impl(ULLSAM_DT_BF16, nullptr, K, W, ldw, C, ldc, out_f32, bias, residual, ldr, 0, act, M, N, K, nullptr, 0, stream, nullptr, &n);
}

// Decode step (one new token per sequence, B <= 4): wqkv with the optional RMSNorm prologue (norm_w NULL: `a` holds bf16 activations, x is
// ignored) and the head split + RoPE + KV-cache append in the epilogue; arguments as ullsam_gemm_qkv_rope with S = 1.
extern "C" int ullsam_decode_qkv_rope(const void* a, const float* x, long ldx, const float* norm_w, float eps, const void* W, long ldw,
                                      const float* bias, int B, int K, int KVH, int G, const int* pos, const float* cos_tab,
                                      const float* sin_tab, int tab_rows, void* q_out, void* k_cache, void* v_cache, int cap, int cache_pos0,
                                      void* stream) {
    ULLSAM_CHECK(B > 0 && B <= 4 && KVH > 0 && G > 0 && tab_rows > 0 && K % 2048 == 0, "ullsam_decode_qkv_rope: B=%d (1..4) K=%d (%% 2048)", B, K);
    ULLSAM_CHECK(cache_pos0 + 1 <= cap, "ullsam_decode_qkv_rope: cache overflow (%d + 1 > %d)", cache_pos0, cap);
    RopeEpilogue r{pos, cos_tab, sin_tab, q_out, k_cache, v_cache, 1, KVH, G, cap, cache_pos0, tab_rows};
    NormPrologue n{x, ldx, norm_w, eps};
    const int N = KVH * (G + 2) * 128;
    return gemm_impl(ULLSAM_DT_BF16, norm_w ? nullptr : a, K, W, ldw, q_out, (long)KVH * G * 128, 0, bias, nullptr, 0, 0, 4, B, N, K, nullptr, 0,
                     stream, &r, norm_w ? &n : nullptr);
}

// wqkv GEMM of InternLM2Attention with the head split, RoPE and the KV-cache append in its epilogue (modeling_internlm2.py:359-388):
// x [B*S, K] . wqkv [(KVH*(G+2))*128, K]^T (+bias)  ->  q_out T [B*S, KVH*G*128] (rotated), k_cache / v_cache T [B, KVH, cap, 128]
// rows cache_pos0 .. cache_pos0+S-1 (k rotated).  head_dim is 128 (the tile holds whole heads); the qkv tensor never exists.
extern "C" int ullsam_gemm_qkv_rope(int dtype, const void* A, long lda, const void* W, long ldw, const float* bias, int B, int S, int K,
                                    int KVH, int G, const int* pos, const float* cos_tab, const float* sin_tab, int tab_rows, void* q_out,
                                    void* k_cache, void* v_cache, int cap, int cache_pos0, void* workspace, long ws_bytes, void* stream) {
    ULLSAM_CHECK(B > 0 && S > 0 && KVH > 0 && G > 0 && tab_rows > 0, "ullsam_gemm_qkv_rope: bad dims");
    ULLSAM_CHECK(cache_pos0 + S <= cap, "ullsam_gemm_qkv_rope: cache overflow (%d + %d > %d)", cache_pos0, S, cap);
    ULLSAM_CHECK(((uintptr_t)q_out & 15) == 0 && ((uintptr_t)k_cache & 15) == 0 && ((uintptr_t)v_cache & 15) == 0, "ullsam_gemm_qkv_rope: 16-byte aligned outputs needed");
    RopeEpilogue r{pos, cos_tab, sin_tab, q_out, k_cache, v_cache, S, KVH, G, cap, cache_pos0, tab_rows};
    const int N = KVH * (G + 2) * 128;
    return gemm_impl(dtype, A, lda, W, ldw, q_out, (long)KVH * G * 128, 0, bias, nullptr, 0, 0, 4, B * S, N, K, workspace, ws_bytes, stream, &r);
}
