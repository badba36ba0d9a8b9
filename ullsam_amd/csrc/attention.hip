// Fused attention kernels.
//
//  flash_attn_kernel<T, HD, MODE, NWAVES>   (MFMA 32x32, online softmax, never materialises S x S)
//    MODE_VIT_WINDOW : SAM ViT windowed attention, image_encoder.py:170-177,224-240,243-289,325-361.
//        window_partition / window_unpartition are fused into the K/V/Q gather and the output scatter:
//        the qkv GEMM runs on the UNPADDED 64x64 tokens; window pad tokens (zero rows after norm1, i.e.
//        qkv == qkv.bias) are synthesised from the bias vector, so they stay live keys exactly as in the
//        reference; their query rows are never written back (the reference crops them).
//    MODE_VIT_GLOBAL : global attention over the 64x64 grid (window_size == 0 blocks).
//        Both ViT modes add the decomposed rel-pos bias  rel_h[q,kh] + rel_w[q,kw]  computed from the
//        UNSCALED q (image_encoder.py:231-234,354-355): T = RelTable . q^T is produced with MFMA at kernel
//        start (table rows staged through the K/V LDS buffers), scattered into per-wave LDS tables
//        relh[kh][q], relw[kw][q] and read back in the softmax.
//    MODE_CAUSAL     : InternLM2 GQA prefill, modeling_internlm2.py:383-419 with the additive masks of
//        :96-125,830-851 reproduced literally (finfo.min for causal, finfo.min for key padding, summed in fp32).
//    MODE_PLAIN      : no bias, no mask (unused by the hot path; kept for tests).
//  Orientation: S^T = K.Q^T (keys on accumulator rows, queries on lanes) so row statistics are lane-local,
//  then O^T = V^T.P^T with the S^T accumulator re-used directly as the B operand (cdna guide section 3,
//  "An accumulator tile as the next MFMA's operand"); V is read column-wise with ds_read_b64_tr_b16 (bf16)
//  or 8 ds_read_b32 (f32).
//
//  naive_attn_kernel / fewkeys_attn_kernel: small-shape attention for the two-way mask decoder
//  (transformer.py:220-242) and the q_len==1 decode step.
#include "common.h"

enum { MODE_PLAIN = 0, MODE_CAUSAL = 1, MODE_VIT_GLOBAL = 2, MODE_VIT_WINDOW = 3 };

struct AttnArgs {
    const void* q; const void* k; const void* v; void* out;
    long q_bs, q_ts, q_hs;  // element strides: batch, token, head
    long k_bs, k_ts, k_hs;
    long v_bs, v_ts, v_hs;
    long o_bs, o_ts, o_hs;
    int B, H, groups;       // groups = q heads per kv head
    int Sq, Sk;
    float scale;
    const int* key_mask;    // [B, Sk] 1 = attend, 0 = padding; may be null
    int q_pos0;             // causal: absolute position of query 0 (number of cached keys before it)
    const void* rel_h; const void* rel_w;  // T [(2G-1), HD]
    int grid_h, grid_w;     // token grid (64 x 64)
    int win, nwin_w, nwin;  // window size, windows per row, windows per image
    const void* bias_q; const void* bias_k; const void* bias_v;  // T [H*HD] slices of qkv.bias for pad tokens
};

template <typename T, int HD, int TR = 64> struct KVTile {
    static constexpr int RS = HD * (int)sizeof(T) + 16;  // padded row stride in bytes (conflict-free b128 reads)
    static constexpr int ROWS = TR;                      // keys per staged tile: 64, or 128 for the windowed mode (196 keys = 2 tiles)
    static constexpr int BYTES = RS * ROWS;
    static constexpr int CPR = HD * (int)sizeof(T) / 16;  // 16-byte chunks per row
};

// ---- V^T fragment: lane (d = d0 + (lane&31), h) element j = V[kv0 + 8*(j>>2) + 4h + (j&3)][d]
template <int RS>
__device__ __forceinline__ Frag<bf16> load_vt_frag(const char* vt, int kv0, int d0, int lane, const bf16*) {
    const int grp = lane >> 4, i = lane & 15;
    const int h = grp >> 1;
    const int row = kv0 + 4 * h + (i >> 2);
    const int col = d0 + 16 * (grp & 1) + 4 * (i & 3);
    const char* p = vt + row * RS + col * 2;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 8 * RS));
    Frag<bf16> f;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 t;
    t[0] = lo[0]; t[1] = lo[1]; t[2] = lo[2]; t[3] = lo[3];
    t[4] = hi[0]; t[5] = hi[1]; t[6] = hi[2]; t[7] = hi[3];
    f.v = __builtin_bit_cast(bf16x8_t, t);
    return f;
}
template <int RS>
__device__ __forceinline__ Frag<float> load_vt_frag(const char* vt, int kv0, int d0, int lane, const float*) {
    const int h = lane >> 5, d = d0 + (lane & 31);
    Frag<float> f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row = kv0 + 8 * (j >> 2) + 4 * h + (j & 3);
        f.v[j] = *reinterpret_cast<const float*>(vt + row * RS + d * 4);
    }
    return f;
}

__device__ __forceinline__ Frag<bf16> pack_p(const f32x16& s, int half, const bf16*) {
    Frag<bf16> f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f.v[j] = (bf16)s[8 * half + j];
    return f;
}
__device__ __forceinline__ Frag<float> pack_p(const f32x16& s, int half, const float*) {
    Frag<float> f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f.v[j] = s[8 * half + j];
    return f;
}

// FAST64: MODE_VIT_GLOBAL on the 64x64 token grid SAM always uses at 1024^2.  A 32-key block is then half a grid row: its
// key row is uniform (ONE rel_h LDS read per block) and its key columns are one of two fixed sets, so rel_w lives in 32
// registers per lane; only rel_h stays in LDS (8 KiB per wave -> two workgroups per CU).
template <typename T, int HD, int MODE, int NWAVES, bool FAST64>
__global__ __launch_bounds__(NWAVES * 64, 2) void flash_attn_kernel(AttnArgs p) {  // >= 2 waves per SIMD: at most 256 VGPR+AGPR
    constexpr int TR = (MODE == MODE_VIT_WINDOW && NWAVES == 7) ? 128 : 64;  // keys per staged K/V tile
    using KT = KVTile<T, HD, TR>;
    constexpr int RS = KT::RS;
    constexpr int NT = NWAVES * 64;
    constexpr int CPR = KT::CPR;
    constexpr int NCH = (TR * CPR + NT - 1) / NT;  // 16-byte chunks per thread per K (or V) tile
    constexpr int KSTEPS = HD / 16;
    constexpr int DT = (HD + 31) / 32;
    constexpr bool REL = (MODE == MODE_VIT_GLOBAL || MODE == MODE_VIT_WINDOW);
    constexpr float LOG2E = 1.4426950408889634f;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = smem + KT::BYTES;
    // V tile gets 16 extra rows of slack: for HD % 32 != 0 the last d-tile's transposed reads run past HD
    int* kms = reinterpret_cast<int*>(smem + 2 * KT::BYTES + 16 * RS);  // key-padding mask of the staged tile (64 ints)
    float* rel_base = reinterpret_cast<float*>(smem + 2 * KT::BYTES + 16 * RS + 256);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, ql = lane & 31;
    const int head = blockIdx.y;
    const int kvh = head / p.groups;
    int b = blockIdx.z, wy = 0, wx = 0;
    if (MODE == MODE_VIT_WINDOW) {
        const int w = blockIdx.z % p.nwin;
        b = blockIdx.z / p.nwin;
        wy = w / p.nwin_w;
        wx = w % p.nwin_w;
    }
    const int G = (MODE == MODE_VIT_WINDOW) ? p.win : p.grid_w;   // key-grid width (and rel table half size)
    const int GHk = (MODE == MODE_VIT_WINDOW) ? p.win : p.grid_h; // key-grid height
    const float invG = 1.0f / (float)G;

    // ---- this lane's query
    const int qi = (blockIdx.x * NWAVES + wave) * 32 + ql;  // index within sequence / window
    bool q_valid = qi < p.Sq;
    long q_tok = qi;       // token index in the [B, tokens] tensors (for load and store)
    bool q_store = q_valid;
    int qh = 0, qw = 0;    // query grid coordinates (rel-pos)
    if (MODE == MODE_VIT_WINDOW) {
        qh = (int)(((float)qi + 0.5f) * invG);
        qw = qi - qh * G;
        const int gy = wy * p.win + qh, gx = wx * p.win + qw;
        q_store = q_valid && gy < p.grid_h && gx < p.grid_w;
        q_tok = (long)gy * p.grid_w + gx;
    } else if (MODE == MODE_VIT_GLOBAL) {
        qh = (int)(((float)qi + 0.5f) * invG);
        qw = qi - qh * G;
    }
    const T* Q = reinterpret_cast<const T*>(p.q);
    const T* K = reinterpret_cast<const T*>(p.k);
    const T* V = reinterpret_cast<const T*>(p.v);

    Frag<T> qf[KSTEPS];
    {
        const bool from_mem = (MODE == MODE_VIT_WINDOW) ? q_store : q_valid;
        const T* qp = Q + (long)b * p.q_bs + q_tok * p.q_ts + (long)head * p.q_hs;
        const T* bq = (MODE == MODE_VIT_WINDOW) ? reinterpret_cast<const T*>(p.bias_q) + (long)head * HD : nullptr;
#pragma unroll
        for (int t = 0; t < KSTEPS; ++t) {
            if (from_mem) qf[t] = load_frag(qp + 16 * t + 8 * h);
            else if (MODE == MODE_VIT_WINDOW && q_valid) qf[t] = load_frag(bq + 16 * t + 8 * h);
            else qf[t] = zero_frag<T>();
        }
    }

    // ---- generic K/V tile staging (global -> registers -> LDS), with mode specific row sources
    const int Sk = p.Sk;
    auto kv_row_src = [&](int kt, const T*& kp, const T*& vp) -> bool {  // returns false -> zero row
        if (kt >= Sk) return false;
        if (MODE == MODE_VIT_WINDOW) {
            const int ky = (int)(((float)kt + 0.5f) * invG);
            const int kx = kt - ky * G;
            const int gy = wy * p.win + ky, gx = wx * p.win + kx;
            if (gy < p.grid_h && gx < p.grid_w) {
                const long tok = (long)gy * p.grid_w + gx;
                kp = K + (long)b * p.k_bs + tok * p.k_ts + (long)kvh * p.k_hs;
                vp = V + (long)b * p.v_bs + tok * p.v_ts + (long)kvh * p.v_hs;
            } else {  // window pad token: LN output padded with zeros => k = bias_k, v = bias_v (live key)
                kp = reinterpret_cast<const T*>(p.bias_k) + (long)kvh * HD;
                vp = reinterpret_cast<const T*>(p.bias_v) + (long)kvh * HD;
            }
            return true;
        }
        kp = K + (long)b * p.k_bs + (long)kt * p.k_ts + (long)kvh * p.k_hs;
        vp = V + (long)b * p.v_bs + (long)kt * p.v_ts + (long)kvh * p.v_hs;
        return true;
    };
    uint4 kreg[NCH], vreg[NCH];
    int km_reg = 1;
    const int* kmask_g = (MODE == MODE_CAUSAL && p.key_mask) ? p.key_mask + (long)b * Sk : nullptr;
    auto load_tile = [&](int tile) {
        if (MODE == MODE_CAUSAL && tid < 64) {
            const int kt = tile * 64 + tid;
            km_reg = (kmask_g && kt < Sk) ? kmask_g[kt] : 1;
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int idx = c * NT + tid;
            const int row = idx / CPR, ch = idx - row * CPR;
            kreg[c] = make_uint4(0, 0, 0, 0);
            vreg[c] = make_uint4(0, 0, 0, 0);
            if (row < TR) {
                const T* kp; const T* vp;
                if (kv_row_src(tile * TR + row, kp, vp)) {
                    kreg[c] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(kp) + ch * 16);
                    vreg[c] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(vp) + ch * 16);
                }
            }
        }
    };
    auto store_tile = [&]() {
        if (MODE == MODE_CAUSAL && tid < 64) kms[tid] = km_reg;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int idx = c * NT + tid;
            const int row = idx / CPR, ch = idx - row * CPR;
            if (row < TR) {
                *reinterpret_cast<uint4*>(Ks + row * RS + ch * 16) = kreg[c];
                *reinterpret_cast<uint4*>(Vs + row * RS + ch * 16) = vreg[c];
            }
        }
    };

    // ---- rel-pos tables: T^T[e][q] = RelTable[e][:] . q[:]  (unscaled q), scattered to relh[kh][q] / relw[kw][q]
    constexpr int RELROWS = (MODE == MODE_VIT_WINDOW) ? 16 : 64;  // key-grid side: window <= 14, global <= 64
    constexpr int RELTABS = FAST64 ? 1 : 2;                       // FAST64 keeps rel_w in registers
    float* relh = rel_base + wave * (RELTABS * RELROWS * 32);     // [RELROWS][32] floats each
    float* relw = FAST64 ? relh : relh + RELROWS * 32;            // FAST64: the same slab is first used as rel_w scratch
    float rw[2][16];                                              // FAST64: rel_w * log2(e) for key columns 32*par + crow32(r, h)
#pragma unroll
    for (int i = 0; i < 32; ++i) rw[i >> 4][i & 15] = 0.f;
    load_tile(0);  // K/V tile 0 -> registers now: its global latency hides behind the rel-pos table phase
    if (MODE == MODE_VIT_WINDOW) {
        // both tables (2*(2G-1) <= 54 rows) in ONE staging pass: rows [0,NE) = rel_h, [NE,2NE) = rel_w, rest zero
        const int NE = 2 * G - 1;
        const T* th = reinterpret_cast<const T*>(p.rel_h);
        const T* tw = reinterpret_cast<const T*>(p.rel_w);
        for (int idx = tid; idx < 64 * CPR; idx += NT) {
            const int row = idx / CPR, ch = idx - row * CPR;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (row < NE) v = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(th + (long)row * HD) + ch * 16);
            else if (row < 2 * NE) v = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(tw + (long)(row - NE) * HD) + ch * 16);
            *reinterpret_cast<uint4*>(Ks + row * RS + ch * 16) = v;
        }
        __syncthreads();
#pragma unroll 1
        for (int t = 0; t < 2; ++t) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                const Frag<T> a = load_frag(reinterpret_cast<const T*>(Ks + (32 * t + ql) * RS) + 16 * ks + 8 * h);
                mma32(a, qf[ks], acc);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int e = 32 * t + crow32(r, h);
                if (e < NE) {
                    const int kk = qh + (G - 1) - e;  // rel index e = q - k + (G-1)  (get_rel_pos, image_encoder.py:318-322)
                    if (kk >= 0 && kk < GHk) relh[kk * 32 + ql] = acc[r];
                } else if (e < 2 * NE) {
                    const int kk = qw + (G - 1) - (e - NE);
                    if (kk >= 0 && kk < G) relw[kk * 32 + ql] = acc[r];
                }
            }
        }
        __syncthreads();
    } else     if (REL) {
        const int NE = 2 * G - 1;  // rows per table (<= 127)
        const T* tabs[2] = {reinterpret_cast<const T*>(p.rel_h), reinterpret_cast<const T*>(p.rel_w)};
#pragma unroll 1
        for (int it = 0; it < 2; ++it) {
            const int tb = FAST64 ? 1 - it : it;  // FAST64: width table first (its slab is then recycled for rel_h)
            __syncthreads();
            // stage table rows 0..127 into the contiguous K|V region (128 rows of RS bytes)
            for (int idx = tid; idx < 128 * CPR; idx += NT) {
                const int row = idx / CPR, ch = idx - row * CPR;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (row < NE) v = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(tabs[tb] + (long)row * HD) + ch * 16);
                *reinterpret_cast<uint4*>(Ks + row * RS + ch * 16) = v;
            }
            __syncthreads();
            const int qc = tb == 0 ? qh : qw;
            const int GK = tb == 0 ? GHk : G;
            float* dst = tb == 0 ? relh : relw;
            const int ntile = (NE + 31) / 32;
#pragma unroll 1
            for (int t = 0; t < ntile; ++t) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < KSTEPS; ++ks) {
                    const Frag<T> a = load_frag(reinterpret_cast<const T*>(Ks + (32 * t + ql) * RS) + 16 * ks + 8 * h);
                    mma32(a, qf[ks], acc);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int e = 32 * t + crow32(r, h);
                    const int kk = qc + (G - 1) - e;  // rel index e = q - k + (G-1)  (get_rel_pos, image_encoder.py:318-322)
                    if (e < NE && kk >= 0 && kk < GK) dst[kk * 32 + ql] = acc[r];
                }
            }
            if (FAST64 && tb == 1) {
                __syncthreads();  // both lane halves of every wave have scattered their rel_w entries
#pragma unroll
                for (int i = 0; i < 32; ++i) rw[i >> 4][i & 15] = relw[(32 * (i >> 4) + crow32(i & 15, h)) * 32 + ql] * LOG2E;
            }
        }
        __syncthreads();
    }

    // ---- main loop
    f32x16 o[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float scale2 = p.scale * LOG2E;
    const int q_pos = p.q_pos0 + qi;

    int ntiles = (Sk + TR - 1) / TR;
    if (MODE == MODE_CAUSAL) {
        const int last_q = p.q_pos0 + min(p.Sq, (int)(blockIdx.x + 1) * NWAVES * 32) - 1;
        ntiles = min(ntiles, last_q / 64 + 1);
    }
    const float FMIN = -3.4028234663852886e38f;  // torch.finfo(float32).min

    store_tile();  // tile 0 was fetched before the table phase
    __syncthreads();
    // SAM's windows are always 14x14 (build_sam.py:78): 196 keys = a compile-time tile count, and with the tile loop unrolled
    // the key -> (row, col) split of every accumulator register is a compile-time constant (two candidates, by lane half).
    const bool win14 = (MODE == MODE_VIT_WINDOW) && p.win == 14;
    constexpr int NT14 = (196 + TR - 1) / TR;
    auto do_tile = [&](const int tile, const bool W14) __attribute__((always_inline)) {
        if (tile + 1 < ntiles) load_tile(tile + 1);
        const bool tile_pad = (MODE == MODE_CAUSAL && kmask_g) ? (__any(kms[lane] == 0) != 0) : false;
        const int wave_first_q = p.q_pos0 + ((int)blockIdx.x * NWAVES + wave) * 32;
#pragma unroll
        for (int sub = 0; sub < TR / 32; ++sub) {
            const int kbase = tile * TR + sub * 32;
            if (kbase >= Sk) continue;
            if (wave_first_q - p.q_pos0 >= p.Sq) continue;  // wave-uniform: no valid query in this wave
            if (MODE == MODE_CAUSAL) {
                const int wave_last_q = p.q_pos0 + min(p.Sq, ((int)blockIdx.x * NWAVES + wave + 1) * 32) - 1;
                if (kbase > wave_last_q) continue;  // wave-uniform: the whole 32-key block is in the future
            }
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                const Frag<T> a = load_frag(reinterpret_cast<const T*>(Ks + (sub * 32 + ql) * RS) + 16 * ks + 8 * h);
                mma32(a, qf[ks], s);
            }
            float mx = -INFINITY;
            const float rh64 = FAST64 ? relh[(kbase >> 6) * 32 + ql] * LOG2E : 0.f;  // key row of this 32-key block
            // causal: blocks entirely in the past of every query of this wave and free of padding need no mask arithmetic
            const bool interior = (MODE == MODE_CAUSAL) && !tile_pad && (kbase + 31 <= wave_first_q) && (kbase + 31 < Sk);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kt = kbase + crow32(r, h);
                float v = s[r] * scale2;
                if (FAST64) {
                    v += rh64 + rw[sub & 1][r];  // key column = 32*(sub&1) + crow32(r, h): compile-time register index
                } else if (MODE == MODE_VIT_WINDOW && W14) {
                    const int k0 = kbase + crow32(r, 0), k1 = k0 + 4;  // the two lane halves' keys; constants after unrolling
                    const int offh = h ? (k1 / 14) * 32 : (k0 / 14) * 32;
                    const int offw = h ? (k1 % 14) * 32 : (k0 % 14) * 32;
                    const bool valid = h ? (k1 < 196) : (k0 < 196);
                    v = valid ? v + (relh[(offh < 14 * 32 ? offh : 0) + ql] + relw[offw + ql]) * LOG2E : -INFINITY;
                } else if (REL) {
                    const int ky = (int)(((float)kt + 0.5f) * invG);
                    const int kx = kt - ky * G;
                    const float bias = (kt < Sk) ? (relh[ky * 32 + ql] + relw[kx * 32 + ql]) : 0.f;
                    v += bias * LOG2E;
                }
                if (MODE == MODE_CAUSAL && !interior) {
                    // additive masks exactly as the reference builds them (fp32): causal min + padding min
                    float add = 0.f;
                    if (kt > q_pos) add += FMIN;
                    if (kms[sub * 32 + crow32(r, h)] == 0) add += FMIN;
                    // = finfo.min (score absorbed) or -inf; deliberately NOT rescaled by log2(e): a row whose
                    // keys are all single-masked must stay uniform over them, as in the reference's eager softmax
                    if (add != 0.f) v = s[r] * p.scale + add;
                }
                if (!FAST64 && !interior && !(MODE == MODE_VIT_WINDOW && W14) && kt >= Sk) v = -INFINITY;
                s[r] = v;
                mx = fmaxf(mx, v);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = exp2f(m_run - m_use);  // m_run = -inf -> 0
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = exp2f(s[r] - m_use);
                s[r] = pv;
                psum += pv;
            }
            l_run = l_run * alpha + psum;
            m_run = m_new;
            if (__any(alpha != 1.0f)) {  // wave-uniform: skip the O rescale when no row maximum moved in this block
#pragma unroll
                for (int d = 0; d < DT; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
            }
            const Frag<T> p0 = pack_p(s, 0, (const T*)nullptr);
            const Frag<T> p1 = pack_p(s, 1, (const T*)nullptr);
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                const Frag<T> v0 = load_vt_frag<RS>(Vs, sub * 32, d * 32, lane, (const T*)nullptr);
                mma32(v0, p0, o[d]);
                const Frag<T> v1 = load_vt_frag<RS>(Vs, sub * 32 + 16, d * 32, lane, (const T*)nullptr);
                mma32(v1, p1, o[d]);
            }
        }
        __syncthreads();
        if (tile + 1 < ntiles) store_tile();
        __syncthreads();
    };
    if (MODE == MODE_VIT_WINDOW && win14) {
#pragma unroll
        for (int tile = 0; tile < NT14; ++tile) do_tile(tile, true);
    } else {
#pragma unroll 1
        for (int tile = 0; tile < ntiles; ++tile) do_tile(tile, false);
    }

    // ---- normalise and store: lane q holds O^T[d][q], d = 32*dt + crow32(r, h)
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv_l = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    if (q_store) {
        T* op = reinterpret_cast<T*>(p.out) + (long)b * p.o_bs + q_tok * p.o_ts + (long)head * p.o_hs;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int dd = 32 * d + 8 * rq + 4 * h;
                if (dd < HD) {
                    const float4 v = make_float4(o[d][4 * rq] * inv_l, o[d][4 * rq + 1] * inv_l, o[d][4 * rq + 2] * inv_l,
                                                 o[d][4 * rq + 3] * inv_l);
                    store4(op + dd, v);
                }
            }
    }
}

template <typename T, int HD, int MODE, int NWAVES, bool FAST64>
static int launch_flash_impl(const AttnArgs& a, hipStream_t s) {
    using KT = KVTile<T, HD, (MODE == MODE_VIT_WINDOW && NWAVES == 7) ? 128 : 64>;
    constexpr bool REL = (MODE == MODE_VIT_GLOBAL || MODE == MODE_VIT_WINDOW);
    constexpr int RELROWS = (MODE == MODE_VIT_WINDOW) ? 16 : 64;
    const size_t lds = 2 * KT::BYTES + 16 * KT::RS + 256 + (REL ? (size_t)NWAVES * (FAST64 ? 1 : 2) * RELROWS * 32 * 4 : 0);
    ULLSAM_CHECK(lds <= 160 * 1024, "flash_attn: LDS %zu exceeds 160 KiB", lds);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(flash_attn_kernel<T, HD, MODE, NWAVES, FAST64>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    const int qtiles = (a.Sq + NWAVES * 32 - 1) / (NWAVES * 32);
    const int nz = MODE == MODE_VIT_WINDOW ? a.B * a.nwin : a.B;
    flash_attn_kernel<T, HD, MODE, NWAVES, FAST64><<<dim3(qtiles, a.H, nz), dim3(NWAVES * 64), lds, s>>>(a);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

template <typename T, int HD, int MODE, int NWAVES>
static int launch_flash(const AttnArgs& a, hipStream_t s) {
    if (MODE == MODE_VIT_GLOBAL && a.grid_h == 64 && a.grid_w == 64) return launch_flash_impl<T, HD, MODE, NWAVES, MODE == MODE_VIT_GLOBAL>(a, s);
    return launch_flash_impl<T, HD, MODE, NWAVES, false>(a, s);
}

template <typename T, int MODE, int NWAVES>
static int dispatch_hd(const AttnArgs& a, int hd, hipStream_t s) {
    switch (hd) {
        case 64: return launch_flash<T, 64, MODE, NWAVES>(a, s);
        case 80: if (MODE != MODE_CAUSAL) return launch_flash<T, (MODE != MODE_CAUSAL ? 80 : 64), MODE, NWAVES>(a, s); break;
        case 128: if (MODE == MODE_CAUSAL) return launch_flash<T, (MODE == MODE_CAUSAL ? 128 : 64), MODE, NWAVES>(a, s); break;
        default: break;
    }
    ULLSAM_CHECK(false, "flash_attn: unsupported head_dim %d for mode %d (ViT: 64/80, causal: 64/128)", hd, MODE);
}

static int g_attn_variant = 0;
extern "C" int ullsam_set_attn_variant(int v) { g_attn_variant = v; return 0; }

// SAM ViT attention on the packed qkv activations [B, grid_h*grid_w, 3*D] (D = heads*hd, per token [3][heads][hd]).
// window == 0 -> global attention.  rel_h/rel_w: [(2S-1), hd] in the activation dtype.  qkv_bias: [3*D] in the
// activation dtype (only read for window pad tokens).  out: [B, grid_h*grid_w, D].
extern "C" int ullsam_vit_attention(int dtype, const void* qkv, void* out, const void* rel_h, const void* rel_w,
                                    const void* qkv_bias, int B, int heads, int hd, int grid_h, int grid_w, int window,
                                    void* stream) {
    ULLSAM_CHECK(dtype == 0 || dtype == 1, "vit_attention: bad dtype");
    ULLSAM_CHECK(grid_h <= 64 && grid_w <= 64 && window <= 14, "vit_attention: grid %dx%d / window %d too large", grid_h, grid_w, window);
    const int esz = dtype == 0 ? 4 : 2;
    ULLSAM_CHECK((hd * esz) % 16 == 0, "vit_attention: head_dim*elem must be a multiple of 16 bytes");
    const long D = (long)heads * hd;
    const long N = (long)grid_h * grid_w;
    AttnArgs a = {};
    const char* base = reinterpret_cast<const char*>(qkv);
    a.q = base; a.k = base + D * esz; a.v = base + 2 * D * esz; a.out = out;
    a.q_bs = a.k_bs = a.v_bs = N * 3 * D; a.q_ts = a.k_ts = a.v_ts = 3 * D; a.q_hs = a.k_hs = a.v_hs = hd;
    a.o_bs = N * D; a.o_ts = D; a.o_hs = hd;
    a.B = B; a.H = heads; a.groups = 1;
    a.scale = 1.0f / sqrtf((float)hd);
    a.rel_h = rel_h; a.rel_w = rel_w; a.grid_h = grid_h; a.grid_w = grid_w;
    const char* bb = reinterpret_cast<const char*>(qkv_bias);
    a.bias_q = bb; a.bias_k = bb + D * esz; a.bias_v = bb + 2 * D * esz;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (window > 0) {
        a.win = window;
        a.nwin_w = (grid_w + window - 1) / window;
        a.nwin = a.nwin_w * ((grid_h + window - 1) / window);
        a.Sq = a.Sk = window * window;
        // window = 196 queries: one 7-wave workgroup (128-key tiles) or two 4-wave workgroups (64-key tiles, 3 resident per CU)
        if (g_attn_variant == 1) return dtype == 0 ? dispatch_hd<float, MODE_VIT_WINDOW, 7>(a, hd, s) : dispatch_hd<bf16, MODE_VIT_WINDOW, 7>(a, hd, s);
        return dtype == 0 ? dispatch_hd<float, MODE_VIT_WINDOW, 7>(a, hd, s) : dispatch_hd<bf16, MODE_VIT_WINDOW, 4>(a, hd, s);
    }
    a.Sq = a.Sk = (int)N;
    return dtype == 0 ? dispatch_hd<float, MODE_VIT_GLOBAL, 4>(a, hd, s) : dispatch_hd<bf16, MODE_VIT_GLOBAL, 4>(a, hd, s);
}

// InternLM2 GQA prefill attention.  q: [B, Sq, H*hd]; k/v: cache layout [B, KVH, k_cap, hd] holding Sk = q_pos0 + Sq
// valid positions; key_mask [B, Sk] (1 = attend) or null; out [B, Sq, H*hd].
extern "C" int ullsam_causal_attention(int dtype, const void* q, const void* k, const void* v, void* out, const int* key_mask,
                                       int B, int H, int KVH, int hd, int Sq, int Sk, int k_cap, int q_pos0, void* stream) {
    ULLSAM_CHECK(dtype == 0 || dtype == 1, "causal_attention: bad dtype");
    ULLSAM_CHECK(H % KVH == 0, "causal_attention: H %% KVH != 0");
    AttnArgs a = {};
    a.q = q; a.k = k; a.v = v; a.out = out;
    a.q_bs = (long)Sq * H * hd; a.q_ts = (long)H * hd; a.q_hs = hd;
    a.k_bs = a.v_bs = (long)KVH * k_cap * hd; a.k_ts = a.v_ts = hd; a.k_hs = a.v_hs = (long)k_cap * hd;
    a.o_bs = a.q_bs; a.o_ts = a.q_ts; a.o_hs = hd;
    a.B = B; a.H = H; a.groups = H / KVH; a.Sq = Sq; a.Sk = Sk; a.key_mask = key_mask; a.q_pos0 = q_pos0;
    a.scale = 1.0f / sqrtf((float)hd);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // 4 waves = 128 queries per workgroup (8 waves / 256 queries measured no faster end to end: 94.9 vs 94.5 ms per step)
    return dtype == 0 ? dispatch_hd<float, MODE_CAUSAL, 4>(a, hd, s) : dispatch_hd<bf16, MODE_CAUSAL, 4>(a, hd, s);
}

// ------------------------------------------------------------------------------------------------------
// Small-shape attention (decoder tokens, decode step).  One workgroup per (batch, head, query).
//   out[b,q,h,:] = softmax(scale * q.k^T + mask) . v      q:[B,Sq,H,hd] k,v:[B,Sk,Hk,hd] via explicit strides
// ------------------------------------------------------------------------------------------------------
struct NaiveArgs {
    const void* q; const void* k; const void* v; void* out;
    long q_bs, q_ts, q_hs, k_bs, k_ts, k_hs, v_bs, v_ts, v_hs, o_bs, o_ts, o_hs;
    int H, groups, Sq, Sk, hd;
    float scale;
    const int* key_mask;  // [B, Sk] or null (decode: finfo.min additive like the reference)
};

template <typename T>
__global__ __launch_bounds__(256) void naive_attn_kernel(NaiveArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sc = reinterpret_cast<float*>(smem);            // [Sk]
    float* qs = sc + ((p.Sk + 3) & ~3);                    // [hd]
    float* red = qs + p.hd;                                // [256]
    const int tid = threadIdx.x;
    const int qi = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
    const int kvh = head / p.groups;
    const T* qp = reinterpret_cast<const T*>(p.q) + (long)b * p.q_bs + (long)qi * p.q_ts + (long)head * p.q_hs;
    for (int d = tid; d < p.hd; d += 256) qs[d] = to_f32(qp[d]);
    __syncthreads();
    const T* kb = reinterpret_cast<const T*>(p.k) + (long)b * p.k_bs + (long)kvh * p.k_hs;
    const T* vb = reinterpret_cast<const T*>(p.v) + (long)b * p.v_bs + (long)kvh * p.v_hs;
    const float FMIN = -3.4028234663852886e38f;
    float mx = -INFINITY;
    for (int kt = tid; kt < p.Sk; kt += 256) {
        const T* kp = kb + (long)kt * p.k_ts;
        float acc = 0.f;
        for (int d = 0; d < p.hd; d += 4) {
            const float4 kv = load4(kp + d);
            acc += qs[d] * kv.x + qs[d + 1] * kv.y + qs[d + 2] * kv.z + qs[d + 3] * kv.w;
        }
        acc *= p.scale;
        if (p.key_mask && p.key_mask[(long)b * p.Sk + kt] == 0) acc += FMIN;
        sc[kt] = acc;
        mx = fmaxf(mx, acc);
    }
    red[tid] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]);
        __syncthreads();
    }
    mx = red[0];
    __syncthreads();
    float sum = 0.f;
    for (int kt = tid; kt < p.Sk; kt += 256) {
        const float e = __expf(sc[kt] - mx);
        sc[kt] = e;
        sum += e;
    }
    red[tid] = sum;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    const float inv = 1.0f / red[0];
    __syncthreads();
    // out[d] = sum_k p[k] v[k][d]; threads = (slice, d): nsl key slices of hd lanes each
    const int hd = p.hd;
    const int nsl = 256 / hd > 0 ? 256 / hd : 1;
    const int d = tid % hd, sl = tid / hd;
    float acc = 0.f;
    if (sl < nsl)
        for (int kt = sl; kt < p.Sk; kt += nsl) acc += sc[kt] * to_f32(vb[(long)kt * p.v_ts + d]);
    red[tid] = (sl < nsl) ? acc : 0.f;
    __syncthreads();
    if (tid < hd) {
        float t = 0.f;
        for (int s2 = 0; s2 < nsl; ++s2) t += red[s2 * hd + tid];
        T* op = reinterpret_cast<T*>(p.out) + (long)b * p.o_bs + (long)qi * p.o_ts + (long)head * p.o_hs;
        op[tid] = from_f32<T>(t * inv);
    }
}

extern "C" int ullsam_naive_attention(int dtype, const void* q, const void* k, const void* v, void* out, const int* key_mask,
                                      int B, int H, int KVH, int hd, int Sq, int Sk,
                                      long q_bs, long q_ts, long q_hs, long k_bs, long k_ts, long k_hs,
                                      long v_bs, long v_ts, long v_hs, long o_bs, long o_ts, long o_hs, float scale, void* stream) {
    ULLSAM_CHECK(hd % 4 == 0 && hd <= 256, "naive_attention: hd=%d", hd);
    ULLSAM_CHECK(Sk > 0 && Sk <= 32768, "naive_attention: Sk=%d", Sk);
    NaiveArgs a{q, k, v, out, q_bs, q_ts, q_hs, k_bs, k_ts, k_hs, v_bs, v_ts, v_hs, o_bs, o_ts, o_hs, H, H / KVH, Sq, Sk, hd, scale, key_mask};
    const size_t lds = (size_t)(((Sk + 3) & ~3) + hd + 256) * 4;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    static bool attr0 = false, attr1 = false;
    if (dtype == 0) {
        if (!attr0) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(naive_attn_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr0 = true; }
        naive_attn_kernel<float><<<dim3(Sq, H, B), dim3(256), lds, s>>>(a);
    } else {
        if (!attr1) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(naive_attn_kernel<bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr1 = true; }
        naive_attn_kernel<bf16><<<dim3(Sq, H, B), dim3(256), lds, s>>>(a);
    }
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// ---- token -> image cross attention of the two-way decoder (transformer.py:160-166, 100-106) -------------------------
// Few queries (T <= 8 tokens) against N = 4096 image keys, P prompts: K and V are streamed once, whole rows, fully coalesced
// (a lane loads 16 B = EPL channels of one head; LPH lanes share a head and reduce the dot product with two DPP shuffles), all
// T queries of all heads are scored against each loaded row, online softmax per (query, head) in registers.  Keys are split
// over `nsplit` workgroups per prompt (flash-decoding): each writes (max, sum, unnormalised o), `tok2img_merge_kernel` combines.
// C = H*16 = 128 only (SAM's decoder); k/v batch stride 0 = one image shared by every prompt.
template <typename TK> struct KVRow;
template <> struct KVRow<float> {
    static constexpr int EPL = 4;
    static __device__ __forceinline__ void ld(const void* base, long elem, float (&o)[4]) {
        const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + elem);
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    }
};
template <> struct KVRow<bf16> {
    static constexpr int EPL = 8;
    static __device__ __forceinline__ void ld(const void* base, long elem, float (&o)[8]) {
        const uint4 v = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16*>(base) + elem);
        const unsigned int d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) { o[2 * j] = __uint_as_float(d[j] << 16); o[2 * j + 1] = __uint_as_float(d[j] & 0xffff0000u); }
    }
};

template <typename TK, int TQ>
__global__ __launch_bounds__(256) void tok2img_partial_kernel(const float* __restrict__ q, const void* __restrict__ k,
                                                              const void* __restrict__ v, float* __restrict__ ws, int T, int N,
                                                              long k_bs, long v_bs, float scale, int nsplit) {
    constexpr int C = 128, HD = 16, H = 8;
    constexpr int EPL = KVRow<TK>::EPL, LPH = HD / EPL, LPR = C / EPL, RPW = 64 / LPR;  // lanes per head / per row, rows per wave load
    __shared__ float so[4][TQ][C];
    __shared__ float sml[4][TQ][H][2];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int split = blockIdx.x, pr = blockIdx.y;
    const int c = lane % LPR, r = lane / LPR;   // channel chunk, row within the wave load
    const int ch0 = c * EPL, head = ch0 / HD;
    const int per = (N + nsplit - 1) / nsplit;
    const int s0 = split * per, s1 = min(N, s0 + per);

    float qf[TQ][EPL];
#pragma unroll
    for (int t = 0; t < TQ; ++t)
#pragma unroll
        for (int e = 0; e < EPL; ++e) qf[t][e] = t < T ? q[((long)pr * T + t) * C + ch0 + e] : 0.f;
    float m[TQ], l[TQ], o[TQ][EPL];
#pragma unroll
    for (int t = 0; t < TQ; ++t) {
        m[t] = -1e30f; l[t] = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) o[t][e] = 0.f;
    }
    const void* kb = reinterpret_cast<const char*>(k) + (long)pr * k_bs * sizeof(TK);
    const void* vb = reinterpret_cast<const char*>(v) + (long)pr * v_bs * sizeof(TK);
    for (int row0 = s0 + wv * RPW; row0 < s1; row0 += 4 * RPW) {
        const int row = row0 + r;
        const bool valid = row < s1;
        const long off = (long)min(row, N - 1) * C + ch0;
        float kf[EPL], vf[EPL];
        KVRow<TK>::ld(kb, off, kf);
        KVRow<TK>::ld(vb, off, vf);
#pragma unroll
        for (int t = 0; t < TQ; ++t) {
            float sc = 0.f;
#pragma unroll
            for (int e = 0; e < EPL; ++e) sc += qf[t][e] * kf[e];
#pragma unroll
            for (int x = 1; x < LPH; x <<= 1) sc += __shfl_xor(sc, x, 64);
            sc *= scale;  // after QK^T, as transformer.py:234
            const float mn = fmaxf(m[t], sc);
            const float al = __expf(m[t] - mn), pv = valid ? __expf(sc - mn) : 0.f;
            l[t] = l[t] * al + pv;
#pragma unroll
            for (int e = 0; e < EPL; ++e) o[t][e] = o[t][e] * al + pv * vf[e];
            m[t] = mn;
        }
    }
    // merge the RPW row groups of the wave (lanes c, c + LPR, ...)
#pragma unroll
    for (int x = LPR; x < 64; x <<= 1) {
#pragma unroll
        for (int t = 0; t < TQ; ++t) {
            const float m2 = __shfl_xor(m[t], x, 64), l2 = __shfl_xor(l[t], x, 64);
            const float mn = fmaxf(m[t], m2);
            const float a1 = __expf(m[t] - mn), a2 = __expf(m2 - mn);
            l[t] = l[t] * a1 + l2 * a2;
#pragma unroll
            for (int e = 0; e < EPL; ++e) o[t][e] = o[t][e] * a1 + __shfl_xor(o[t][e], x, 64) * a2;
            m[t] = mn;
        }
    }
    if (r == 0) {
#pragma unroll
        for (int t = 0; t < TQ; ++t) {
#pragma unroll
            for (int e = 0; e < EPL; ++e) so[wv][t][ch0 + e] = o[t][e];
            if ((c % LPH) == 0) { sml[wv][t][head][0] = m[t]; sml[wv][t][head][1] = l[t]; }
        }
    }
    __syncthreads();
    // merge the 4 waves; partials to ws: o [P][nsplit][T][C], then ml [P][nsplit][T][H][2]
    float* wo = ws + ((long)pr * nsplit + split) * T * C;
    float* wml = ws + (long)gridDim.y * nsplit * T * C + ((long)pr * nsplit + split) * T * H * 2;
    for (int i = tid; i < T * C; i += 256) {
        const int t = i / C, cc = i % C, hh = cc / HD;
        float mn = sml[0][t][hh][0];
#pragma unroll
        for (int w = 1; w < 4; ++w) mn = fmaxf(mn, sml[w][t][hh][0]);
        float acc = 0.f, ll = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float a = __expf(sml[w][t][hh][0] - mn);
            acc += so[w][t][cc] * a;
            ll += sml[w][t][hh][1] * a;
        }
        wo[i] = acc;
        if ((cc % HD) == 0) { wml[(t * H + hh) * 2] = mn; wml[(t * H + hh) * 2 + 1] = ll; }
    }
}

__global__ __launch_bounds__(128) void tok2img_merge_kernel(const float* __restrict__ ws, float* __restrict__ out, int P, int T, int nsplit) {
    constexpr int C = 128, HD = 16, H = 8;
    const int pt = blockIdx.x, pr = pt / T, t = pt % T, cc = threadIdx.x, hh = cc / HD;
    const float* wo = ws + (long)pr * nsplit * T * C;
    const float* wml = ws + (long)P * nsplit * T * C + (long)pr * nsplit * T * H * 2;
    float mn = -1e30f;
    for (int s2 = 0; s2 < nsplit; ++s2) mn = fmaxf(mn, wml[((long)s2 * T + t) * H * 2 + hh * 2]);
    float acc = 0.f, ll = 0.f;
    for (int s2 = 0; s2 < nsplit; ++s2) {
        const float a = __expf(wml[((long)s2 * T + t) * H * 2 + hh * 2] - mn);
        acc += wo[((long)s2 * T + t) * C + cc] * a;
        ll += wml[((long)s2 * T + t) * H * 2 + hh * 2 + 1] * a;
    }
    out[((long)pr * T + t) * C + cc] = acc / ll;
}

// q f32 [P,T,128]; k,v [P or 1, N, 128] in kv_dtype with element batch strides (0 = shared); out f32 [P,T,128];
// workspace f32 [P*nsplit*T*(128+16)].
extern "C" int ullsam_tok2img_attention(int kv_dtype, const float* q, const void* k, const void* v, float* out, int P, int H, int hd,
                                        int T, int N, long k_batch_stride, long v_batch_stride, float scale, float* workspace,
                                        int nsplit, void* stream) {
    ULLSAM_CHECK(H == 8 && hd == 16, "tok2img_attention: H=%d hd=%d (8 x 16 only)", H, hd);
    ULLSAM_CHECK(T >= 1 && T <= 8 && N >= 1 && nsplit >= 1 && P >= 1, "tok2img_attention: T=%d (1..8) N=%d nsplit=%d P=%d", T, N, nsplit, P);
    ULLSAM_CHECK((((uintptr_t)k | (uintptr_t)v) & 15) == 0, "tok2img_attention: k/v must be 16-byte aligned");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid(nsplit, P);
    if (kv_dtype == ULLSAM_DT_F32) tok2img_partial_kernel<float, 8><<<grid, 256, 0, s>>>(q, k, v, workspace, T, N, k_batch_stride, v_batch_stride, scale, nsplit);
    else tok2img_partial_kernel<bf16, 8><<<grid, 256, 0, s>>>(q, k, v, workspace, T, N, k_batch_stride, v_batch_stride, scale, nsplit);
    ULLSAM_LAUNCH_CHECK();
    tok2img_merge_kernel<<<P * T, 128, 0, s>>>(workspace, out, P, T, nsplit);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// Many queries x few keys (image -> token cross attention, transformer.py:178-181): one thread per (b, head, query).
// q:[B,Sq,H*HD] f32, k,v:[B,Sk,H*HD] f32, out:[B,Sq,H*HD] f32.
template <int HD>
__global__ __launch_bounds__(256) void fewkeys_attn_kernel(const float* q, const float* k, const float* v, float* out,
                                                           int B, int H, int Sq, int Sk, float scale, long q_bs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ks = reinterpret_cast<float*>(smem);  // [Sk][HD]
    float* vs = ks + Sk * HD;
    const int head = blockIdx.y, b = blockIdx.z;
    for (int i = threadIdx.x; i < Sk * HD; i += 256) {
        const int kt = i / HD, d = i % HD;
        ks[i] = k[((long)b * Sk + kt) * H * HD + head * HD + d];
        vs[i] = v[((long)b * Sk + kt) * H * HD + head * HD + d];
    }
    __syncthreads();
    const int qi = blockIdx.x * 256 + threadIdx.x;
    if (qi >= Sq) return;
    const float* qp = q + (long)b * q_bs + (long)qi * H * HD + head * HD;
    float qr[HD], o[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) { qr[d] = qp[d]; o[d] = 0.f; }
    float m = -INFINITY, l = 0.f;
    for (int kt = 0; kt < Sk; ++kt) {
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) s += qr[d] * ks[kt * HD + d];
        s *= scale;
        const float mn = fmaxf(m, s);
        const float al = __expf(m - mn), pv = __expf(s - mn);
        l = l * al + pv;
#pragma unroll
        for (int d = 0; d < HD; ++d) o[d] = o[d] * al + pv * vs[kt * HD + d];
        m = mn;
    }
    const float inv = 1.0f / l;
    float* op = out + ((long)b * Sq + qi) * H * HD + head * HD;
#pragma unroll
    for (int d = 0; d < HD; ++d) op[d] = o[d] * inv;
}

// q_batch_stride in elements: Sq*H*hd for per-batch queries, 0 when every batch shares one query set.
extern "C" int ullsam_fewkeys_attention(const float* q, const float* k, const float* v, float* out, int B, int H, int hd,
                                        int Sq, int Sk, float scale, long q_batch_stride, void* stream) {
    ULLSAM_CHECK(hd == 16 || hd == 32, "fewkeys_attention: hd=%d (16 or 32)", hd);
    ULLSAM_CHECK(Sk > 0 && (size_t)Sk * hd * 8 <= 64 * 1024, "fewkeys_attention: Sk=%d too large", Sk);
    const size_t lds = (size_t)Sk * hd * 8;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((Sq + 255) / 256, H, B);
    if (hd == 16) fewkeys_attn_kernel<16><<<grid, dim3(256), lds, s>>>(q, k, v, out, B, H, Sq, Sk, scale, q_batch_stride);
    else fewkeys_attn_kernel<32><<<grid, dim3(256), lds, s>>>(q, k, v, out, B, H, Sq, Sk, scale, q_batch_stride);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}
