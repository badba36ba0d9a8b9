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
//  win14_attn_kernel: SAM's 14x14 windows in bf16 (the production windowed path): whole window staged once per (window, head),
//    rel-pos terms in registers; the generic MODE_VIT_WINDOW path above serves fp32 and other window sizes.
//  decode_attn_*: q_len == 1 against the bf16 KV cache (GQA groups together, keys split over workgroups, softmax merge).
//  tok2img_*: token -> image cross attention of the two-way mask decoder (few queries, 4096 keys streamed once).
//  naive_attn_kernel / fewkeys_attn_kernel: small-shape attention for the two-way mask decoder
//  (transformer.py:220-242) and the fp32 / odd-shape q_len==1 decode step.
#include "common.h"
#include <type_traits>

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
    unsigned long long* dbg; // diagnostic launches only (ullsam_set_attn_debug): s_memtime stamps of causal128_attn_kernel<true>
};

template <typename T, int HD, int TR = 64> struct KVTile {
    static constexpr int RS = HD * (int)sizeof(T) + 16;  // padded row stride in bytes (conflict-free b128 reads)
    static constexpr int ROWS = TR;                      // keys per staged tile: 64, or 128 for the windowed mode (196 keys = 2 tiles)
    static constexpr int BYTES = RS * ROWS;
    static constexpr int CPR = HD * (int)sizeof(T) / 16;  // 16-byte chunks per row
    // V rows are read transposed (ds_read_b64_tr_b16: 32 lanes = 4 rows x 64 contiguous bytes): conflict-free when consecutive
    // rows start 16 banks apart, i.e. row stride = 64 (mod 256) bytes.  With the K stride (HD*2+16) the four rows overlapped
    // in 12 of their 16 banks (SQ_LDS_BANK_CONFLICT was 47 % of the LDS cycles of the causal kernel).
    static constexpr int RSV = sizeof(T) == 2 ? 320 : RS;
    static constexpr int VBYTES = RSV * ROWS;
    static_assert(sizeof(T) != 2 || HD * 2 <= 256, "V row does not fit its 320-byte stride");
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
// Lanes l and l^32 hold the two key halves of one query: combine them with v_permlane32_swap (a VALU op) instead of a
// ds_bpermute round trip through the LDS.  After the swap every lane holds {lower-half value, upper-half value} in (a, b).
__device__ __forceinline__ void halves(float x, float& lo, float& hi) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    lo = __uint_as_float(r[0]);
    hi = __uint_as_float(r[1]);
}

// max of the 16 registers of an MFMA accumulator as ONE asm statement of v_max3_f32: through fmaxf hipcc puts a canonicalising v_max_f32 x, x in
// front of every MFMA result (16 more VALU instructions per 32-key block in loops that are VALU-bound), and it pads every asm statement's boundary
// with an s_nop, so a chain of eight statements would carry eight of them.  The caller guarantees that the MFMAs that wrote `s` are at least a few
// instructions back (no MFMA -> VALU read hazard at this place).
__device__ __forceinline__ float max3_raw(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float max16_raw(const f32x16& s) {
    float mr;
    asm("v_max3_f32 %0, %1, %2, %3\n\tv_max3_f32 %0, %0, %4, %5\n\tv_max3_f32 %0, %0, %6, %7\n\tv_max3_f32 %0, %0, %8, %9\n\t"
        "v_max3_f32 %0, %0, %10, %11\n\tv_max3_f32 %0, %0, %12, %13\n\tv_max3_f32 %0, %0, %14, %15\n\tv_max_f32 %0, %0, %16"
        : "=&v"(mr)
        : "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]), "v"(s[4]), "v"(s[5]), "v"(s[6]), "v"(s[7]), "v"(s[8]), "v"(s[9]), "v"(s[10]), "v"(s[11]),
          "v"(s[12]), "v"(s[13]), "v"(s[14]), "v"(s[15]));
    return mr;
}

template <typename T, int HD, int MODE, int NWAVES, bool FAST64>
__global__ __launch_bounds__(NWAVES * 64, NWAVES == 8 ? 1 : 2) void flash_attn_kernel(AttnArgs p) {  // 8 waves per CU = 2 per SIMD: at most 256 VGPR+AGPR
    constexpr int TR = ((MODE == MODE_VIT_WINDOW && NWAVES == 7) || NWAVES == 8) ? 128 : 64;  // keys per staged K/V tile
    using KT = KVTile<T, HD, TR>;
    constexpr int RS = KT::RS;
    constexpr int RSV = KT::RSV;
    constexpr int NT = NWAVES * 64;
    constexpr int CPR = KT::CPR;
    constexpr int NCH = (TR * CPR + NT - 1) / NT;  // 16-byte chunks per thread per K (or V) tile
    constexpr int KSTEPS = HD / 16;
    constexpr int DT = (HD + 31) / 32;
    constexpr bool REL = (MODE == MODE_VIT_GLOBAL || MODE == MODE_VIT_WINDOW);
    constexpr float LOG2E = 1.4426950408889634f;
    // bf16 with HD % 32 != 0 (ViT-H: 80): the last 32-wide d-tile of O^T = V^T P^T has unused rows.  A column of ones at V[:, HD]
    // makes row HD of O^T the softmax denominator (sum of the bf16 p it is normalised with, rescaled by alpha like every other
    // row), so the 16 adds per lane and block and the running sum disappear from the VALU stream.
    constexpr bool LSUM_MFMA = sizeof(T) == 2 && (HD % 32) != 0 && (HD % 32) <= 16 && (HD % 8) == 0;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = smem + KT::BYTES;
    // V tile gets 16 extra rows of slack: for HD % 32 != 0 the last d-tile's transposed reads run past HD
    int* kms = reinterpret_cast<int*>(smem + KT::BYTES + KT::VBYTES + 16 * RSV);  // key-padding mask of the staged tile (64 ints)
    float* rel_base = reinterpret_cast<float*>(smem + KT::BYTES + KT::VBYTES + 16 * RSV + 512);

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
    // causal: the last query blocks see the most keys -- dispatch them first so the long workgroups do not form the tail
    const int qblk = (MODE == MODE_CAUSAL) ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x;
    // causal: the query blocks are RIGHT-aligned (block b covers queries Sq - (nblk - b) * 128 ...), so the ragged block is the first one,
    // which sees one key tile, instead of the last one, which sees them all (S = 1081: 57 of 128 rows live in the heaviest block = 11 % of
    // the kernel's work spent on padding rows)
    const int qbase = (MODE == MODE_CAUSAL ? p.Sq - (int)gridDim.x * NWAVES * 32 : 0) + qblk * NWAVES * 32;
    const int qi = qbase + wave * 32 + ql;  // index within sequence / window
    bool q_valid = qi >= 0 && qi < p.Sq;
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
        if (MODE == MODE_CAUSAL && tid < TR) {
            const int kt = tile * TR + tid;
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
        if (MODE == MODE_CAUSAL && tid < TR) kms[tid] = km_reg;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int idx = c * NT + tid;
            const int row = idx / CPR, ch = idx - row * CPR;
            if (row < TR) {
                *reinterpret_cast<uint4*>(Ks + row * RS + ch * 16) = kreg[c];
                *reinterpret_cast<uint4*>(Vs + row * RSV + ch * 16) = vreg[c];
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
                    if (kk >= 0 && kk < GHk) relh[kk * 32 + ql] = acc[r] * LOG2E;  // tables are kept in log2 units
                } else if (e < 2 * NE) {
                    const int kk = qw + (G - 1) - (e - NE);
                    if (kk >= 0 && kk < G) relw[kk * 32 + ql] = acc[r] * LOG2E;
                }
            }
        }
        __syncthreads();
    } else     if (REL) {
        const int NE = 2 * G - 1;  // rows per table (<= 127)
#pragma unroll 1
        for (int it = 0; it < 2; ++it) {
            const int tb = FAST64 ? 1 - it : it;  // FAST64: width table first (its slab is then recycled for rel_h)
            const T* const tab = reinterpret_cast<const T*>(tb == 0 ? p.rel_h : p.rel_w);   // (a select, not an indexed pointer array: that array lived in scratch)
            __syncthreads();
            // stage table rows 0..127 into the contiguous K|V region (128 rows of RS bytes)
            for (int idx = tid; idx < 128 * CPR; idx += NT) {
                const int row = idx / CPR, ch = idx - row * CPR;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (row < NE) v = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(tab + (long)row * HD) + ch * 16);
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
                    if (e < NE && kk >= 0 && kk < GK) dst[kk * 32 + ql] = acc[r] * LOG2E;  // tables are kept in log2 units
                }
            }
            if (FAST64 && tb == 1) {
                __syncthreads();  // both lane halves of every wave have scattered their rel_w entries
#pragma unroll
                for (int i = 0; i < 32; ++i) rw[i >> 4][i & 15] = relw[(32 * (i >> 4) + crow32(i & 15, h)) * 32 + ql];
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
        const int last_q = p.q_pos0 + min(p.Sq, qbase + NWAVES * 32) - 1;
        ntiles = min(ntiles, last_q / TR + 1);
    }
    const float FMIN = -3.4028234663852886e38f;  // torch.finfo(float32).min

    if (LSUM_MFMA) {
        for (int row = tid; row < TR; row += NT) {
            *reinterpret_cast<uint4*>(Vs + row * RSV + HD * 2) = make_uint4(0x00003F80u, 0u, 0u, 0u);  // bf16 1.0 at column HD, zeros after
            *reinterpret_cast<uint4*>(Vs + row * RSV + HD * 2 + 16) = make_uint4(0u, 0u, 0u, 0u);
        }
    }
    store_tile();  // tile 0 was fetched before the table phase
    __syncthreads();
    // SAM's windows are always 14x14 (build_sam.py:78): 196 keys = a compile-time tile count, and with the tile loop unrolled
    // the key -> (row, col) split of every accumulator register is a compile-time constant (two candidates, by lane half).
    const bool win14 = (MODE == MODE_VIT_WINDOW) && p.win == 14;
    constexpr int NT14 = (196 + TR - 1) / TR;
    // ---- the three stages of one 32-key block
    auto qk_block = [&](const int sub, f32x16& s) __attribute__((always_inline)) {  // S^T = K . Q^T
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const Frag<T> a = load_frag(reinterpret_cast<const T*>(Ks + (sub * 32 + ql) * RS) + 16 * ks + 8 * h);
            mma32(a, qf[ks], s);
        }
    };
    // scores -> unnormalised probabilities in place (online softmax); returns the factor the running O must be scaled by
    auto soft_block = [&](const int tile, const int sub, const bool W14, const bool tile_pad, const int wave_first_q, f32x16& s)
                          __attribute__((always_inline)) -> float {
        const int kbase = tile * TR + sub * 32;
        float mx = -INFINITY;
        // FAST64: the key row of this 32-key block is the same for all 16 scores of a lane, so its rel_h term is added to the
        // block maximum and folded into the exponent offset instead of into every score
        const float rh64 = FAST64 ? relh[(kbase >> 6) * 32 + ql] : 0.f;
        // causal: blocks entirely in the past of every query of this wave and free of padding need no mask arithmetic
        const bool interior = (MODE == MODE_CAUSAL) && !tile_pad && (kbase + 31 <= wave_first_q) && (kbase + 31 < Sk);
        if (MODE == MODE_CAUSAL && interior) {
            // no mask, no bias: the maximum is taken on the raw scores and the scale rides in the exponent's FMA
            // (16 FMA + 8 max3 instead of 16 mul + 16 sub + the select chains of the masked path)
            float mr = s[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mr = fmaxf(mr, s[r]);
            float lo, hi;
            halves(mr, lo, hi);
            const float mx = fmaxf(lo, hi) * scale2;
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], scale2, -m_new));   // (explicit: the LDS-DMA kernel must round the same way)
                s[r] = pv;
                psum += pv;
            }
            l_run = __builtin_fmaf(l_run, alpha, psum);
            m_run = m_new;
            return alpha;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kt = kbase + crow32(r, h);
            float v;
            if (FAST64) {
                v = s[r] * scale2 + rw[sub & 1][r];  // key column = 32*(sub&1) + crow32(r, h): compile-time register index
            } else if (MODE == MODE_VIT_WINDOW && W14) {
                const int k0 = kbase + crow32(r, 0), k1 = k0 + 4;  // the two lane halves' keys; constants after unrolling
                const int offh = h ? (k1 / 14) * 32 : (k0 / 14) * 32;
                const int offw = h ? (k1 % 14) * 32 : (k0 % 14) * 32;
                const bool valid = h ? (k1 < 196) : (k0 < 196);
                v = valid ? s[r] * scale2 + (relh[(offh < 14 * 32 ? offh : 0) + ql] + relw[offw + ql]) : -INFINITY;
            } else if (REL) {
                const int ky = (int)(((float)kt + 0.5f) * invG);
                const int kx = kt - ky * G;
                const float bias = (kt < Sk) ? (relh[ky * 32 + ql] + relw[kx * 32 + ql]) : 0.f;
                v = s[r] * scale2 + bias;
            } else {
                v = s[r] * scale2;
            }
            if (MODE == MODE_CAUSAL && !interior) {
                // additive masks exactly as the reference builds them (fp32): causal min + padding min
                float add = 0.f;
                if (kt > q_pos) add += FMIN;
                if (kms[sub * 32 + crow32(r, h)] == 0) add += FMIN;
                // = finfo.min (score absorbed) or -inf; deliberately NOT rescaled by log2(e): a row whose
                // keys are all single-masked must stay uniform over them, as in the reference's eager softmax
                if (add != 0.f) v = __builtin_fmaf(s[r], p.scale, add);
            }
            if (!FAST64 && !interior && !(MODE == MODE_VIT_WINDOW && W14) && kt >= Sk) v = -INFINITY;
            if (MODE == MODE_CAUSAL) asm volatile("" : "+v"(v));   // the product is rounded here, not contracted into the exponent's subtraction below (as in causal128_attn_kernel)
            s[r] = v;
            mx = fmaxf(mx, v);
        }
        {
            float lo, hi;
            halves(mx, lo, hi);
            mx = fmaxf(lo, hi) + rh64;
        }
        const float m_new = fmaxf(m_run, mx);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);  // m_run = -inf -> 0
        const float m_off = m_use - rh64;
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pv = __builtin_amdgcn_exp2f(s[r] - m_off);  // raw v_exp_f32: arguments are <= 0, underflow to 0 is the intent
            s[r] = pv;
            if (!LSUM_MFMA) psum += pv;
        }
        l_run = (MODE == MODE_CAUSAL) ? __builtin_fmaf(l_run, alpha, psum) : l_run * alpha + psum;
        m_run = m_new;
        return alpha;
    };
    auto pv_block = [&](const int sub, const Frag<T>& p0, const Frag<T>& p1) __attribute__((always_inline)) {  // O^T += V^T . P^T
#pragma unroll
        for (int d = 0; d < DT; ++d) {
            const Frag<T> v0 = load_vt_frag<RSV>(Vs, sub * 32, d * 32, lane, (const T*)nullptr);
            mma32(v0, p0, o[d]);
            const Frag<T> v1 = load_vt_frag<RSV>(Vs, sub * 32 + 16, d * 32, lane, (const T*)nullptr);
            mma32(v1, p1, o[d]);
        }
    };
    auto scale_o = [&](const float alpha) __attribute__((always_inline)) {
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
    };

    // The ViT kernels (bf16) are VALU-bound: per 32-key block a lane spends ~140 issue slots on the softmax against 11 MFMAs, and a
    // wave that runs QK -> softmax -> PV in sequence leaves the matrix pipe idle during its own softmax.  PIPE software-pipelines
    // the blocks of a staged tile inside the wave: while block j's softmax runs on the VALU, the matrix pipe computes the scores of
    // block j+1 and the PV product of block j-1 (independent registers, one basic block: the O rescale is unconditional there).
    constexpr bool PIPE_OK = sizeof(T) == 2 && REL;
    auto do_tile = [&](const int tile, const bool W14) __attribute__((always_inline)) {
        if (tile + 1 < ntiles) load_tile(tile + 1);
        const bool tile_pad = (MODE == MODE_CAUSAL && kmask_g) ? (__any(kms[lane] == 0 || (TR > 64 && kms[lane + 64] == 0)) != 0) : false;
        const int wave_first_q = p.q_pos0 + qbase + wave * 32;
        constexpr int NSUB = TR / 32;
        if (PIPE_OK && (FAST64 || W14)) {
            // block validity is a compile-time fact here: FAST64 has Sk = 4096 (every block full); W14 has Sk = 196 with the tile
            // index a constant of the unrolled tile loop
            auto valid = [&](const int sub) { return FAST64 || (tile * TR + sub * 32 < 196); };
            if (wave_first_q - p.q_pos0 < p.Sq && valid(0)) {  // wave-uniform: this wave has queries
                f32x16 sc[2];
                Frag<T> p0, p1;
                qk_block(0, sc[0]);
#pragma unroll
                for (int sub = 0; sub < NSUB; ++sub) {
                    if (!valid(sub)) break;
                    if (sub + 1 < NSUB && valid(sub + 1)) qk_block(sub + 1, sc[(sub + 1) & 1]);
                    if (sub > 0) pv_block(sub - 1, p0, p1);
                    const float alpha = soft_block(tile, sub, W14, false, wave_first_q, sc[sub & 1]);
                    scale_o(alpha);
                    p0 = pack_p(sc[sub & 1], 0, (const T*)nullptr);
                    p1 = pack_p(sc[sub & 1], 1, (const T*)nullptr);
                    if (sub + 1 == NSUB || !valid(sub + 1)) pv_block(sub, p0, p1);  // drain before the tile buffer is recycled
                }
            }
        } else {
#pragma unroll
            for (int sub = 0; sub < NSUB; ++sub) {
                const int kbase = tile * TR + sub * 32;
                if (kbase >= Sk) continue;
                if (wave_first_q - p.q_pos0 >= p.Sq) continue;  // wave-uniform: no valid query in this wave
                if (MODE == MODE_CAUSAL) {
                    const int wave_last_q = p.q_pos0 + min(p.Sq, qbase + (wave + 1) * 32) - 1;
                    if (wave_last_q < p.q_pos0) continue;  // wave-uniform: a wave of the ragged first block that lies before query 0
                    if (kbase > wave_last_q) continue;  // wave-uniform: the whole 32-key block is in the future
                }
                f32x16 sc;
                qk_block(sub, sc);
                const float alpha = soft_block(tile, sub, W14, tile_pad, wave_first_q, sc);
                if (__any(alpha != 1.0f)) scale_o(alpha);  // wave-uniform: skip the O rescale when no row maximum moved in this block
                const Frag<T> p0 = pack_p(sc, 0, (const T*)nullptr);
                const Frag<T> p1 = pack_p(sc, 1, (const T*)nullptr);
                pv_block(sub, p0, p1);
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
    float l_tot;
    {
        float lo, hi;
        if (LSUM_MFMA) {  // row HD of O^T: d-tile DT-1, crow32(r, h) == HD % 32 at h = 0
            constexpr int RL = ((HD % 32) & 3) + 4 * ((HD % 32) >> 3);
            halves(o[DT - 1][RL], lo, hi);
            l_tot = lo;
        } else {
            halves(l_run, lo, hi);
            l_tot = lo + hi;
        }
    }
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
    using KT = KVTile<T, HD, ((MODE == MODE_VIT_WINDOW && NWAVES == 7) || NWAVES == 8) ? 128 : 64>;
    constexpr bool REL = (MODE == MODE_VIT_GLOBAL || MODE == MODE_VIT_WINDOW);
    constexpr int RELROWS = (MODE == MODE_VIT_WINDOW) ? 16 : 64;
    const size_t lds = KT::BYTES + KT::VBYTES + 16 * KT::RSV + 512 + (REL ? (size_t)NWAVES * (FAST64 ? 1 : 2) * RELROWS * 32 * 4 : 0);
    ULLSAM_CHECK(lds <= 160 * 1024, "flash_attn: LDS %zu exceeds 160 KiB", lds);
    static PerDeviceOnce attr;
    if (attr.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(flash_attn_kernel<T, HD, MODE, NWAVES, FAST64>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
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

// ------------------------------------------------------------------------------------------------------
//  win14_attn_kernel: SAM's 14x14 windowed attention (build_sam.py:78), bf16, ONE 4-wave workgroup per (image, window, head).
//  The tiled kernel spends two thirds of its time before the first score: every (window, head) was staged by two workgroups
//  (two query tiles), each fetching the 196 K and V slices (160 B out of every 7680-byte token), the rel-pos tables and running
//  the table products.  Here the whole window is fetched and staged in LDS once; each wave handles two 32-query groups
//  (196 = 7 groups over 4 waves) against it; both rel-pos tables live in 28 registers per lane and group (the query's 14 row and
//  14 column terms; the column table is pre-rotated for the upper lane half, whose keys sit 4 further), so LDS holds only K and
//  V: 77.5 KB, two workgroups per CU.  Workgroups are renumbered so that an XCD walks the heads of one window back to back
//  (neighbouring heads share 128-byte lines of the packed qkv row).  The seven 32-key blocks of a group are unrolled and
//  software-pipelined (scores of block j+1 and the PV product of block j-1 on the matrix pipe under block j's softmax); the
//  softmax denominator comes out of the PV product through a column of ones at V[:, HD].
//  Same arithmetic as flash_attn_kernel<.., MODE_VIT_WINDOW, ..>.
// ------------------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256, 2) void win14_attn_kernel(AttnArgs p) {
    typedef bf16 T;
    constexpr int G = 14, NK = 196, NB = 7, NT = 256, NGRP = 2;
    constexpr int KSTEPS = HD / 16, DT = (HD + 31) / 32;
    constexpr int RS = HD * 2 + 16;         // K row stride: conflict-free ds_read_b128 of 32 rows
    constexpr int RSV = 192;                // V row stride = -64 (mod 256): conflict-free ds_read_b64_tr_b16 of 4 rows x 64 B
    constexpr int CPR = HD * 2 / 16;        // 16-byte chunks per row
    constexpr int KBYTES = NK * RS, VROWS = 224;
    constexpr int NCH = (NK * CPR + NT - 1) / NT;
    constexpr float LOG2E = 1.4426950408889634f;
    static_assert(HD == 80, "win14_attn_kernel is laid out for head_dim 80 (ones column at V[:, 80..95], 192-byte V rows)");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = smem + KBYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, ql = lane & 31;
    // problem index: workgroup ids go round robin over the 8 XCDs; give every XCD a contiguous range of (window, head) problems
    const int nprob = gridDim.x;
    int prob;
    {
        const int bid = blockIdx.x, q8 = nprob >> 3, r8 = nprob & 7, xcd = bid & 7;
        prob = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    }
    // The two workgroups resident on a CU start together and would run their fetch phase (80 KB each, a 41 MB burst chip-wide) and
    // their matrix phase in lockstep, round after round (identical durations).  The second resident workgroup of the first round
    // waits a fraction of a workgroup's duration once, so that from then on one fetches while the other computes (measured on the
    // ViT-H shape: 102 us without, 85 us with 6.8 us of stagger, 87 / 92 us with 10 / 14 us).
    if (p.q_pos0 > 0 && (int)blockIdx.x < 512 && (((int)blockIdx.x >> 3) & 32))
        for (int i = 0; i < p.q_pos0; ++i) __builtin_amdgcn_s_sleep(127);
    const int head = prob % p.H;
    int wz = (prob / p.H) % p.nwin, b = prob / (p.H * p.nwin);
    const int wy = wz / p.nwin_w, wx = wz % p.nwin_w;
    const T* Q = reinterpret_cast<const T*>(p.q);
    const T* K = reinterpret_cast<const T*>(p.k);
    const T* V = reinterpret_cast<const T*>(p.v);

    // ---- K / V of the whole window -> registers (their latency hides behind the table phase)
    uint4 kreg[NCH], vreg[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int idx = c * NT + tid;
        const int row = idx / CPR, ch = idx - row * CPR;
        kreg[c] = make_uint4(0, 0, 0, 0);
        vreg[c] = make_uint4(0, 0, 0, 0);
        if (row < NK) {
            const int ky = row / G, kx = row - ky * G;
            const int gy = wy * G + ky, gx = wx * G + kx;
            const T* kp; const T* vp;
            if (gy < p.grid_h && gx < p.grid_w) {
                const long tok = (long)gy * p.grid_w + gx;
                kp = K + (long)b * p.k_bs + tok * p.k_ts + (long)head * p.k_hs;
                vp = V + (long)b * p.v_bs + tok * p.v_ts + (long)head * p.v_hs;
            } else {  // window pad token: LN output padded with zeros => k = bias_k, v = bias_v (live key)
                kp = reinterpret_cast<const T*>(p.bias_k) + (long)head * HD;
                vp = reinterpret_cast<const T*>(p.bias_v) + (long)head * HD;
            }
            kreg[c] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(kp) + ch * 16);
            vreg[c] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(vp) + ch * 16);
        }
    }
    // ---- rel-pos tables: rows [0,27) = rel_h, [27,54) = rel_w, staged in the (still free) V region with the K row stride
    constexpr int NE = 2 * G - 1;
    {
        const T* th = reinterpret_cast<const T*>(p.rel_h);
        const T* tw = reinterpret_cast<const T*>(p.rel_w);
        for (int idx = tid; idx < 64 * CPR; idx += NT) {
            const int row = idx / CPR, ch = idx - row * CPR;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (row < NE) v = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(th + (long)row * HD) + ch * 16);
            else if (row < 2 * NE) v = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(tw + (long)(row - NE) * HD) + ch * 16);
            *reinterpret_cast<uint4*>(Vs + row * RS + ch * 16) = v;
        }
    }
    // ---- per query group: the lane's query fragments and its rel-pos terms
    Frag<T> qf[NGRP][KSTEPS];
    float relh_r[NGRP][G], relw_r[NGRP][G];  // relw_r is rotated by 4 for the upper lane half: its keys are k0 + 4
    bool q_store[NGRP];
    long q_tok[NGRP];
#pragma unroll
    for (int g = 0; g < NGRP; ++g) {
        const int qi = (g * 4 + wave) * 32 + ql;
        const bool q_valid = qi < NK;
        const int qh = qi / G, qw = qi - qh * G;
        const int qgy = wy * G + qh, qgx = wx * G + qw;
        q_store[g] = q_valid && qgy < p.grid_h && qgx < p.grid_w;
        q_tok[g] = (long)qgy * p.grid_w + qgx;
        const T* qp = Q + (long)b * p.q_bs + q_tok[g] * p.q_ts + (long)head * p.q_hs;
        const T* bq = reinterpret_cast<const T*>(p.bias_q) + (long)head * HD;  // window pad token: q = qkv.bias
#pragma unroll
        for (int t = 0; t < KSTEPS; ++t) {
            if (q_store[g]) qf[g][t] = load_frag(qp + 16 * t + 8 * h);
            else if (q_valid) qf[g][t] = load_frag(bq + 16 * t + 8 * h);
            else qf[g][t] = zero_frag<T>();
        }
    }
    __syncthreads();  // table rows staged
    // T^T[e][q] = Table[e][:] . q[:] (unscaled q, image_encoder.py:231-234), scattered per query to a wave-private scratch
    // [28][32] (rows 0..13: rel_h by key row, 14..27: rel_w by key column), in log2 units
    float* scratch = reinterpret_cast<float*>(Vs + 64 * RS) + wave * (2 * G * 32);
    auto rel_group = [&](const int g) __attribute__((always_inline)) {
        const int qi = (g * 4 + wave) * 32 + ql;
        const int qh = qi / G, qw = qi - qh * G;
        if ((g * 4 + wave) * 32 < NK) {  // wave-uniform: the eighth group does not exist
#pragma unroll 1
            for (int t = 0; t < 2; ++t) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < KSTEPS; ++ks) {
                    const Frag<T> a = load_frag(reinterpret_cast<const T*>(Vs + (32 * t + ql) * RS) + 16 * ks + 8 * h);
                    mma32(a, qf[g][ks], acc);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int e = 32 * t + crow32(r, h);
                    if (e < NE) {
                        const int kk = qh + (G - 1) - e;  // rel index e = q - k + (G-1)  (get_rel_pos, image_encoder.py:318-322)
                        if (kk >= 0 && kk < G) scratch[kk * 32 + ql] = acc[r] * LOG2E;
                    } else if (e < 2 * NE) {
                        const int kk = qw + (G - 1) - (e - NE);
                        if (kk >= 0 && kk < G) scratch[(G + kk) * 32 + ql] = acc[r] * LOG2E;
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < G; ++j) {
            relh_r[g][j] = scratch[j * 32 + ql];
            const int jw = h ? (j + 4 < G ? j + 4 : j + 4 - G) : j;
            relw_r[g][j] = scratch[(G + jw) * 32 + ql];
        }
    };
    rel_group(0);
    rel_group(1);
    __syncthreads();  // every wave is done with the table rows and its scratch: the region becomes V
    // ---- stage K and V (+ the ones column, zero pad rows)
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int idx = c * NT + tid;
        const int row = idx / CPR, ch = idx - row * CPR;
        if (row < NK) {
            *reinterpret_cast<uint4*>(Ks + row * RS + ch * 16) = kreg[c];
            *reinterpret_cast<uint4*>(Vs + row * RSV + ch * 16) = vreg[c];
        }
    }
    for (int row = tid; row < VROWS; row += NT) {
        *reinterpret_cast<uint4*>(Vs + row * RSV + HD * 2) = make_uint4(0x00003F80u, 0u, 0u, 0u);  // bf16 1.0 at column HD
        *reinterpret_cast<uint4*>(Vs + row * RSV + HD * 2 + 16) = make_uint4(0u, 0u, 0u, 0u);
        if (row >= NK) {
#pragma unroll
            for (int ch = 0; ch < CPR; ++ch) *reinterpret_cast<uint4*>(Vs + row * RSV + ch * 16) = make_uint4(0u, 0u, 0u, 0u);
        }
    }
    __syncthreads();

    const float scale2 = p.scale * LOG2E;
    auto run_group = [&](const int g) __attribute__((always_inline)) {  // g is a literal at both call sites: register arrays stay static
        if ((g * 4 + wave) * 32 >= NK) return;  // wave-uniform
        // ---- seven 32-key blocks, software-pipelined
        f32x16 o[DT];
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
        float m_run = -INFINITY;
        auto qk_block = [&](const int blk, f32x16& sc) __attribute__((always_inline)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {  // rows >= 196 of the last block read into the V region: finite, masked below
                const Frag<T> a = load_frag(reinterpret_cast<const T*>(Ks + (blk * 32 + ql) * RS) + 16 * ks + 8 * h);
                mma32(a, qf[g][ks], sc);
            }
        };
        auto pv_block = [&](const int blk, const Frag<T>& p0, const Frag<T>& p1) __attribute__((always_inline)) {
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                const Frag<T> v0 = load_vt_frag<RSV>(Vs, blk * 32, d * 32, lane, (const T*)nullptr);
                mma32(v0, p0, o[d]);
                const Frag<T> v1 = load_vt_frag<RSV>(Vs, blk * 32 + 16, d * 32, lane, (const T*)nullptr);
                mma32(v1, p1, o[d]);
            }
        };
        auto soft_block = [&](const int blk, f32x16& sc) __attribute__((always_inline)) -> float {
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k0 = blk * 32 + crow32(r, 0), k1 = k0 + 4;  // keys of the lower / upper lane half: compile-time constants
                const int y0 = k0 / G, y1 = k1 / G, x0 = k0 - y0 * G;
                const bool ok0 = k0 < NK, ok1 = k1 < NK;
                float v;
                if (!ok0 && !ok1) {
                    v = -INFINITY;
                } else {
                    const float bh = (y0 == y1 || !ok1) ? relh_r[g][y0 < G ? y0 : 0] : (h ? relh_r[g][y1 < G ? y1 : 0] : relh_r[g][y0]);
                    v = sc[r] * scale2 + (bh + relw_r[g][x0]);
                    if (!ok1) v = h ? -INFINITY : v;
                }
                sc[r] = v;
                mx = fmaxf(mx, v);
            }
            {
                float lo, hi;
                halves(mx, lo, hi);
                mx = fmaxf(lo, hi);
            }
            const float m_new = fmaxf(m_run, mx);  // every block has at least one live key for every query: m_new is finite
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[r] = __builtin_amdgcn_exp2f(sc[r] - m_new);
            m_run = m_new;
            return alpha;
        };
        {
            f32x16 sc[2];
            Frag<T> p0, p1;
            qk_block(0, sc[0]);
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) {
                if (blk + 1 < NB) qk_block(blk + 1, sc[(blk + 1) & 1]);
                if (blk > 0) pv_block(blk - 1, p0, p1);
                const float alpha = soft_block(blk, sc[blk & 1]);
#pragma unroll
                for (int d = 0; d < DT; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
                p0 = pack_p(sc[blk & 1], 0, (const T*)nullptr);
                p1 = pack_p(sc[blk & 1], 1, (const T*)nullptr);
            }
            pv_block(NB - 1, p0, p1);
        }
        // ---- normalise (row HD of O^T is the denominator: d-tile DT-1, register crow32^-1(HD % 32) of the lower half) and store
        float l_tot;
        {
            constexpr int RL = ((HD % 32) & 3) + 4 * ((HD % 32) >> 3);
            float lo, hi;
            halves(o[DT - 1][RL], lo, hi);
            l_tot = lo;
        }
        const float inv_l = l_tot > 0.f ? 1.0f / l_tot : 0.f;
        if (q_store[g]) {
            T* op = reinterpret_cast<T*>(p.out) + (long)b * p.o_bs + q_tok[g] * p.o_ts + (long)head * p.o_hs;
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
    };
    run_group(0);
    run_group(1);
}

static int launch_win14(const AttnArgs& a, hipStream_t s) {
    constexpr int HD = 80;
    const size_t lds = 196 * (HD * 2 + 16) + 224 * 192;
    static PerDeviceOnce attr;
    if (attr.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(win14_attn_kernel<HD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    win14_attn_kernel<HD><<<dim3(a.H * a.B * a.nwin), dim3(256), lds, s>>>(a);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------
//  win14r_attn_kernel (round 5): the same windowed attention (image_encoder.py:224-289,325-361; bf16, head_dim 80, 14x14 windows) laid out
//  so that no wave runs a dependent chain of key blocks.  win14_attn_kernel gives a wave 32 queries and walks seven 32-key blocks with an
//  online softmax: ~1.7 kilo-cycles of LATENCY per block (fragment reads -> MFMAs -> maximum -> exponentials -> pack -> transposed reads ->
//  MFMAs), two such chains per SIMD, matrix pipe 14 % busy for four rounds.  Here
//    * a query group is ONE WINDOW ROW: 14 queries (qh fixed, qw = lane & 15; lanes 14 / 15 idle) on v_mfma_f32_16x16x32_bf16, 14 groups per
//      (image, window, head), two per wave of a 7-wave workgroup;
//    * a key tile is one window row as well: 16 accumulator rows = the 14 keys (kh = tile, kw = 4 (lane >> 4) + register) + 2 masked rows.  All 14
//      score tiles of a group (the whole 196-key row of S^T, 56 registers) are computed before the softmax: 42 INDEPENDENT MFMAs, one maximum, 56
//      exponentials, 42 MFMAs for O^T -- no running maximum, no rescale, nothing that waits for the previous block;
//    * with these tiles the decomposed rel-pos bias costs no vector instruction per score: rel_w[q, kw] depends on the accumulator ROW only (4 values
//      per lane, the same for every tile) and enters as the MFMA's INITIAL ACCUMULATOR (divided by the scale once; the two masked rows start at -1e30, so
//      they need no select); rel_h[q, kh] is uniform over a tile and rides in the exponent's offset (one fma per score, which the scale needs anyway);
//    * both tables' products with the UNSCALED q (image_encoder.py:354-355) are 9 MFMAs per group on operands read straight from the L2-resident
//      tables: rel_h's rows are picked per accumulator row (e = qh + 13 - kh: wave-uniform shift), its 4 results per lane become 14 through the
//      wave's LDS scratch (four 16-byte broadcast reads; the v_permlane16/32_swap form compiled to copies of rows 0 .. 5 in rows 8 .. 13 -- hipcc dropped
//      the second result of the second swap, tools/probes/win14r_debug2.py); rel_w's shift depends on the lane's query (e = qw + 13 - kw), so its 27
//      products go through the same 2.5 KB per-wave scratch and come back by a per-lane offset;
//    * K and V are the window's 196 head slices (160 B of every 7680-byte token, pad tokens = qkv.bias) gathered back to back by LDS-DMA with
//      per-lane source addresses: plain 160-byte rows are conflict-free both for the 16-row ds_read_b128 of the K operand and for the 4 x 32 B
//      ds_read_b64_tr_b16 blocks of the V^T operand, so the image needs no padding, no swizzle and no staging registers; the softmax denominator comes from
//      a sixth O^T tile whose A operand is a register of ones (same bf16-rounded probabilities as the numerator);
//    * window rows that lie wholly outside the image (bottom windows: 6 of 14) are skipped;
//    * both groups of a wave are prepared before the barrier (every global load of the workgroup is issued in its first microsecond); the A fragments of the score and
//      PV products run through register rings pinned with sched_barriers (left alone hipcc emits read -> wait -> MFMA one at a time); the outputs leave through the scratch as
//      16-byte stores, ten consecutive lanes per token (8-byte scattered stores cost 24 of 86 us).
//  63.5 KB of K / V + 17.9 KB of scratch per workgroup: two workgroups (14 waves) per CU at <= 128 registers, no spills.
//  Measured (ViT-H, batch 4, cold operands): 85 - 89 -> 64 - 66 us; in the bench step 78.32 -> 77.61 ms.  A persistent form (one 14-wave workgroup per CU, K / V double
//  buffered, next problem's pieces requested a whole problem ahead by asm LDS-DMA) was built on the same arithmetic and measured EQUAL (66 - 68 us): its problems cost 17 - 18
//  kilo-cycles each, 9 of them arithmetic, the rest the one barrier per problem (14 waves in lockstep) and request issue against a saturated memory system (~500 cycles per
//  1 KiB piece and wave); removed (git acf202b has it).
// ------------------------------------------------------------------------------------------------------
template <int ABL>   // ABL: 0 = the kernel; diagnostic builds (attention variants 20 .. 22): 1 = fetch + table phase only, 2 = no K / V fetch, 3 = no stores; 4 = s_memtime stamps (ullsam_set_attn_debug)
__global__ __launch_bounds__(448, 4) void win14r_attn_kernel(AttnArgs p) {
    typedef bf16 T;
    constexpr int HD = 80, G = 14, NK = 196, NW = 7;
    constexpr int RS = 160;                    // K / V row: the 160-byte head slice itself
    constexpr int NPIECE = 31;                 // LDS-DMA pieces of 1 KiB per operand: 198 rows (196 keys + 2 rows that only have to be finite) = 31680 B
    constexpr int OPB = NPIECE * 1024;
    constexpr int SCW = 36;                    // scratch row pitch in floats: 144 B = 16-byte aligned, 4 banks further per query (conflict-free b128 stores)
    constexpr int SCRB = 16 * 160;             // per-wave scratch bytes: the table products ([16][36] floats) and, at the end, the group's [16][160 B] output image
    constexpr float LOG2E = 1.4426950408889634f;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = smem + OPB;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & 15, g = lane >> 4;
    float* scr = reinterpret_cast<float*>(smem + 2 * OPB) + wave * (SCRB / 4);
    // ABL == 4: s_memtime stamps of every wave (tools/probes/win_stamps.py): [workgroup][wave][8]
    auto stamp = [&](const int i) __attribute__((always_inline)) {
        if (ABL == 4) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (lane == 0) p.dbg[((size_t)blockIdx.x * NW + wave) * 8 + i] = t;
        }
    };
    stamp(0);
    const int nprob = gridDim.x;
    int prob;
    {   // workgroup ids go round robin over the 8 XCDs: give every XCD a contiguous range of (window, head) problems (the heads of a window share 128-byte lines)
        const int bid = blockIdx.x, q8 = nprob >> 3, r8 = nprob & 7, xcd = bid & 7;
        prob = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    }
    if (p.q_pos0 > 0 && (int)blockIdx.x < 512 && (((int)blockIdx.x >> 3) & 32))   // one-off stagger of the second resident workgroup of the first round (as win14_attn_kernel)
        for (int i = 0; i < p.q_pos0; ++i) __builtin_amdgcn_s_sleep(127);
    const int head = prob % p.H;
    const int wz = (prob / p.H) % p.nwin, b = prob / (p.H * p.nwin);
    const int wy = wz / p.nwin_w, wx = wz % p.nwin_w;
    const T* Q = reinterpret_cast<const T*>(p.q) + (long)b * p.q_bs + (long)head * p.q_hs;
    const T* K = reinterpret_cast<const T*>(p.k) + (long)b * p.k_bs + (long)head * p.k_hs;
    const T* V = reinterpret_cast<const T*>(p.v) + (long)b * p.v_bs + (long)head * p.v_hs;
    const T* bk = reinterpret_cast<const T*>(p.bias_k) + (long)head * HD;
    const T* bv = reinterpret_cast<const T*>(p.bias_v) + (long)head * HD;

    // ---- K / V of the window -> LDS: piece i of an operand = image bytes [1024 i, 1024 i + 1024), lane's chunk j = 64 i + lane = (row j / 10, chunk j % 10)
#pragma unroll
    for (int n = 0; n < (NPIECE + NW - 1) / NW; ++n) {
        const int i = wave + NW * n;
        if (i < NPIECE) {   // wave-uniform
            const int j = i * 64 + lane;
            const int row = j / 10, c = j - row * 10;
            const int ky = row / G, kx = row - ky * G;
            const int gy = wy * G + ky, gx = wx * G + kx;
            const T* ks; const T* vs;
            if (row < NK && gy < p.grid_h && gx < p.grid_w) {
                const long tok = (long)gy * p.grid_w + gx;
                ks = K + tok * p.k_ts;
                vs = V + tok * p.v_ts;
            } else {        // window pad token: LN output padded with zeros => k = bias_k, v = bias_v (a live key); rows >= 196 are masked: any finite bytes
                ks = bk;
                vs = bv;
            }
            if (ABL != 2) {
                __builtin_amdgcn_global_load_lds(GLB_PTR(ks + 8 * c), LDS_PTR(Ks + i * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds(GLB_PTR(vs + 8 * c), LDS_PTR(Vs + i * 1024), 16, 0, 0);
            }
        }
    }

    stamp(1);
    const T* Rh = reinterpret_cast<const T*>(p.rel_h);
    const T* Rw = reinterpret_cast<const T*>(p.rel_w);
    const float scale2 = p.scale * LOG2E, inv_scale = 1.0f / p.scale;
    const int dg2 = 8 * (g & 1);               // third k-step (dims 64 .. 95): lane groups 2 / 3 hold dims that do not exist -- their q is zero, the other operand re-reads dims 64 .. 79
    Frag<T> ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones.v[j] = (bf16)1.0f;

    // ---- both query groups of the wave (window rows qh = wave and wave + 7) are prepared BEFORE the barrier, while the K / V pieces are in flight: every global load of
    // the workgroup (the pieces, the query fragments, the table fragments) is issued in its first microsecond and its latency is paid once.  What a group carries to its
    // main phase: the query fragments (12 registers), the rel_w initial accumulator (4) and the raw rel_h products (4; spread to all four lane groups through the scratch
    // only when the group's turn comes).  (First version: group 1 prepared after group 0's main phase = a second exposed round of load latency per workgroup.)
    Frag<T> qf0[3], qf1[3];                    // (qf1: loaded once for the table products, again during group 0's main phase)
    f32x4 cinit0, cinit1, thr0, thr1;
    bool live[2], qv;
    {
        const int qgx = wx * G + ql;
        qv = ql < G && qgx < p.grid_w;
        const long qx = qv ? qgx : 0;
        Frag<T> rwf[2][3], rhf[2][3], qt1[3];
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            const int qh = wave + NW * gi, qgy = wy * G + qh;
            live[gi] = qgy < p.grid_h;         // wave-uniform: otherwise the whole window row is padding (its outputs are cropped, image_encoder.py:286-288)
            const T* qp = Q + ((long)(live[gi] ? qgy : 0) * p.grid_w + qx) * p.q_ts;
            Frag<T>* qf = gi ? qt1 : qf0;
            const bool ld = qv && live[gi];
            // the lane's query fragments: B operand, lane (q = ql, g) holds q[32 ks + 8 g .. + 7]
            qf[0] = ld ? load_frag(qp + 8 * g) : zero_frag<T>();
            qf[1] = ld ? load_frag(qp + 32 + 8 * g) : zero_frag<T>();
            qf[2] = (ld && g < 2) ? load_frag(qp + 64 + 8 * g) : zero_frag<T>();
            const int eh = min(max(qh + (G - 1) - ql, 0), 2 * G - 2);   // accumulator row m = ql is key row kh = m: rel index e = qh - kh + 13 (get_rel_pos, image_encoder.py:318-322)
            const T* rh = Rh + eh * HD;
            rhf[gi][0] = load_frag(rh + 8 * g);
            rhf[gi][1] = load_frag(rh + 32 + 8 * g);
            rhf[gi][2] = load_frag(rh + 64 + dg2);
        }
        {
            const T* r0 = Rw + ql * HD;
            const T* r1 = Rw + min(16 + ql, 2 * G - 2) * HD;
            rwf[0][0] = load_frag(r0 + 8 * g); rwf[0][1] = load_frag(r0 + 32 + 8 * g); rwf[0][2] = load_frag(r0 + 64 + dg2);
            rwf[1][0] = load_frag(r1 + 8 * g); rwf[1][1] = load_frag(r1 + 32 + 8 * g); rwf[1][2] = load_frag(r1 + 64 + dg2);
        }
        // rel-pos products with the unscaled q: thr[i] = rel_h[q, kh = 4 g + i]; gwa / gwb[i] = q . Rw[e] for e = 4 g + i and 16 + 4 g + i
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            const Frag<T>* qf = gi ? qt1 : qf0;
            f32x4 th = {0.f, 0.f, 0.f, 0.f}, gwa = th, gwb = th;
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                mma16(rhf[gi][ks], qf[ks], th);
                mma16(rwf[0][ks], qf[ks], gwa);
                mma16(rwf[1][ks], qf[ks], gwb);
            }
            // rel_w: the lane's 4 key columns kw = 4 g + i need Rw rows e = qw + 13 - kw with qw = ql: through the wave's scratch [q][e]
            *reinterpret_cast<f32x4*>(scr + ql * SCW + 4 * g) = gwa;
            *reinterpret_cast<f32x4*>(scr + ql * SCW + 16 + 4 * g) = gwb;
            asm volatile("" ::: "memory");
            f32x4 ci;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int kw = 4 * g + i;
                const float v = scr[ql * SCW + max(ql + (G - 1) - kw, 0)];
                ci[i] = kw < G ? v * inv_scale : -1e30f;         // rows 14 / 15 of a tile are not keys of this window row
            }
            asm volatile("" ::: "memory");
            if (gi) { cinit1 = ci; thr1 = th * LOG2E; } else { cinit0 = ci; thr0 = th * LOG2E; }
        }
    }
    const char* kbase = Ks + ql * RS + g * 16;                   // K operand (A): lane (m = ql, g) reads row 14 t + m, chunk 4 ks + g
    const char* kbase2 = Ks + ql * RS + 128 + dg2 * 2;           // third k-step: chunk 8 + (g & 1)
    const char* vbase = Vs + (4 * g + (ql >> 2)) * RS + 8 * (ql & 3);   // V^T operand (A) by ds_read_b64_tr_b16: lane 4 q' + p of a 16-lane group addresses row q', columns 4 p .. 4 p + 3
    auto main_group = [&](const int gi, const Frag<T>* qf, const f32x4& cinit, const f32x4& thr) __attribute__((always_inline)) {
        const int cur_qh = wave + NW * gi;
        // ---- S^T tile t = key row t: 14 independent accumulation chains of 3, walked k-step by k-step (consecutive MFMAs never depend on each other);
        // the A fragments come through a ring of NR registers sets, each read NR MFMAs ahead of its use.  The order is pinned with sched_barriers: left to
        // itself hipcc emits read -> s_waitcnt lgkmcnt(0) -> MFMA one at a time (57.8 us for the launch's arithmetic alone, first version)
        f32x4 s[G];
#pragma unroll
        for (int t = 0; t < G; ++t) s[t] = cinit;
        {
            constexpr int NR = 5, NI = 3 * G;
            auto kread = [&](const int n) __attribute__((always_inline)) -> Frag<T> {
                const int ks = n / G, t = n - ks * G;
                return load_frag(reinterpret_cast<const T*>((ks == 2 ? kbase2 : kbase + 64 * ks) + t * (G * RS)));
            };
            Frag<T> kf[NR];
#pragma unroll
            for (int n = 0; n < NR; ++n) kf[n] = kread(n);
#pragma unroll
            for (int n = 0; n < NI; ++n) {
                __builtin_amdgcn_sched_barrier(0);
                mma16(kf[n % NR], qf[n / G], s[n % G]);
                __builtin_amdgcn_sched_barrier(0);
                if (n + NR < NI) kf[n % NR] = kread(n + NR);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- rel_h: every lane needs all 14 key rows of its query, and they sit in the four lanes q, q + 16, q + 32, q + 48: through the scratch row, back as four 16-byte
        // broadcast reads (log2 units).  Read only now: 14 registers less while the score tiles are built
        *reinterpret_cast<f32x4*>(scr + ql * SCW + 4 * g) = thr;
        asm volatile("" ::: "memory");
        // ---- one maximum over the whole row (log2 units), one exponential per score
        float mx = -INFINITY;
#pragma unroll
        for (int a4 = 0; a4 < 4; ++a4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(scr + ql * SCW + 4 * a4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int t = 4 * a4 + j;
                if (t < G) {
                    float m4;                  // (asm: through fmaxf hipcc canonicalises every MFMA result with a v_max_f32 x, x first -- 4 more instructions per tile)
                    asm("v_max3_f32 %0, %1, %2, %3\n\tv_max_f32 %0, %0, %4" : "=&v"(m4) : "v"(s[t][0]), "v"(s[t][1]), "v"(s[t][2]), "v"(s[t][3]));
                    mx = fmaxf(mx, fmaf(m4, scale2, v[j]));
                }
            }
        }
        mx = fmaxf(mx, lane_xor16(mx));
        mx = fmaxf(mx, lane_xor32(mx));
        // ---- O^T = V^T P^T: dim tiles 0 .. 4, tile 5 = ones^T P^T (the denominator in every row); V^T fragments (two transposed 8-byte reads each) through a ring as well,
        // the first ones requested behind the exponentials (before them they cost the registers that group 1's carried values need)
        constexpr int NV = 8, NVI = 5 * (G / 2);
        auto vread = [&](const int m) __attribute__((always_inline)) -> Frag<T> {
            const int st = m / 5, d = m - 5 * st;
            const char* a0 = vbase + st * (2 * G * RS) + d * 32;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + G * RS));
            typedef __attribute__((ext_vector_type(8))) short s16x8;
            s16x8 tv;
            tv[0] = lo[0]; tv[1] = lo[1]; tv[2] = lo[2]; tv[3] = lo[3];
            tv[4] = hi[0]; tv[5] = hi[1]; tv[6] = hi[2]; tv[7] = hi[3];
            Frag<T> vf;
            vf.v = __builtin_bit_cast(bf16x8_t, tv);
            return vf;
        };
        Frag<T> pf[G / 2];
#pragma unroll
        for (int a4 = 0; a4 < 4; ++a4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(scr + ql * SCW + 4 * a4);   // (read again rather than held: 14 registers)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int t = 4 * a4 + j;
                if (t < G) {
                    const float off = v[j] - mx;
#pragma unroll
                    for (int i = 0; i < 4; ++i) pf[t >> 1].v[4 * (t & 1) + i] = (bf16)__builtin_amdgcn_exp2f(fmaf(s[t][i], scale2, off));
                }
            }
        }
        asm volatile("" ::: "memory");
        // group 1's query fragments are fetched AGAIN here (its table products, before the barrier, had them once: these are cache hits), one PV phase ahead of their use:
        // held from before the barrier they are 12 registers too many next to group 0's 56 score registers (the build parked them in scratch memory)
        if (gi == 0 && live[1]) {
            __builtin_amdgcn_sched_barrier(0);
            const T* qp = Q + ((long)(wy * G + wave + NW) * p.grid_w + (qv ? wx * G + ql : 0)) * p.q_ts;
            qf1[0] = qv ? load_frag(qp + 8 * g) : zero_frag<T>();
            qf1[1] = qv ? load_frag(qp + 32 + 8 * g) : zero_frag<T>();
            qf1[2] = (qv && g < 2) ? load_frag(qp + 64 + 8 * g) : zero_frag<T>();
            __builtin_amdgcn_sched_barrier(0);
        }
        Frag<T> vf[NV];
#pragma unroll
        for (int m = 0; m < NV; ++m) vf[m] = vread(m);
        f32x4 o[6];
#pragma unroll
        for (int d = 0; d < 6; ++d) o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < NVI; ++m) {
            __builtin_amdgcn_sched_barrier(0);
            mma16(vf[m % NV], pf[m / 5], o[m % 5]);
            if (m % 5 == 4) mma16(ones, pf[m / 5], o[5]);
            __builtin_amdgcn_sched_barrier(0);
            if (m + NV < NVI) vf[m % NV] = vread(m + NV);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- normalise; lane (q, g) holds dims 16 d + 4 g .. + 3 of its query: 8-byte pieces.  Through the wave's scratch as a [16 queries][160 B] image and out as
        // 16-byte stores, ten consecutive lanes per token (the 8-byte scattered stores of the first version cost 24 us of an 86 us launch)
        if (ABL != 3 || o[5][0] == 12345.f) {
            const float inv_l = 1.0f / o[5][0];
            char* ob = reinterpret_cast<char*>(scr);
#pragma unroll
            for (int d = 0; d < 5; ++d) {
                bf16x4_t v;
                v[0] = (bf16)(o[d][0] * inv_l); v[1] = (bf16)(o[d][1] * inv_l); v[2] = (bf16)(o[d][2] * inv_l); v[3] = (bf16)(o[d][3] * inv_l);
                *reinterpret_cast<bf16x4_t*>(ob + ql * 160 + 32 * d + 8 * g) = v;
            }
            asm volatile("" ::: "memory");
            T* orow = reinterpret_cast<T*>(p.out) + (long)b * p.o_bs + (long)head * p.o_hs;
            const int qgy = wy * G + cur_qh;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int c = lane + 64 * r;          // 16-byte chunk of the image: query c / 10, chunk c % 10
                const int qq = c / 10, ch = c - 10 * qq;
                const int gx = wx * G + qq;
                if (c < 16 * 10 && qq < G && gx < p.grid_w) {
                    const uint4 v = *reinterpret_cast<const uint4*>(ob + 16 * c);
                    *reinterpret_cast<uint4*>(orow + ((long)qgy * p.grid_w + gx) * p.o_ts + 8 * ch) = v;
                }
            }
            asm volatile("" ::: "memory");
        }
    };
    stamp(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp(3);
    __syncthreads();                           // every wave's pieces have landed
    stamp(4);
    if (ABL == 1) {
        if ((live[0] || live[1]) && thr0[3] + thr1[3] + cinit1[0] == 12345.f && qv) reinterpret_cast<T*>(p.out)[0] = (T)cinit0[0];
        return;
    }
    if (live[0]) main_group(0, qf0, cinit0, thr0);
    stamp(5);
    if (live[1]) main_group(1, qf1, cinit1, thr1);
    stamp(6);
}

static int launch_win14r(const AttnArgs& a, hipStream_t s, int abl = 0) {
    constexpr int LDS = 2 * 31 * 1024 + 7 * 16 * 160;
    static PerDeviceOnce attr;
    if (attr.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(win14r_attn_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(win14r_attn_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(win14r_attn_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(win14r_attn_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(win14r_attn_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    }
    const dim3 grid(a.H * a.B * a.nwin), blk(448);
    if (a.dbg) win14r_attn_kernel<4><<<grid, blk, LDS, s>>>(a);
    else if (abl == 1) win14r_attn_kernel<1><<<grid, blk, LDS, s>>>(a);
    else if (abl == 2) win14r_attn_kernel<2><<<grid, blk, LDS, s>>>(a);
    else if (abl == 3) win14r_attn_kernel<3><<<grid, blk, LDS, s>>>(a);
    else win14r_attn_kernel<0><<<grid, blk, LDS, s>>>(a);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------
//  causal128_attn_kernel: InternLM2 GQA prefill (modeling_internlm2.py:383-419, masks of :96-125,830-851) for bf16, head_dim 128 -- the
//  production path of the LLM's 32 layers.  Same mathematics, mask semantics and MFMA order as flash_attn_kernel<bf16, 128, MODE_CAUSAL, 4>
//  (the results are bit-identical: tests/test_kernels_gpu.py), different data movement.  The tiled kernel stages every 64-key K/V tile
//  global -> registers -> LDS with two barriers per tile and one tile of lead: with two 32-key blocks of work per wave and tile the fetch
//  latency is exposed (4.0 kilo-cycles per block and wave against 0.5 of MFMA + 0.7 of softmax arithmetic: rocprofv3, round 2).  Here
//    * K and V tiles arrive by LDS-DMA (global_load_lds_dwordx4, 1 KiB = 4 key rows per wave instruction) into a DOUBLE-buffered image,
//      requested a whole tile ahead, no staging registers (32 VGPRs less), ONE barrier per tile;
//    * the image has plain 256-byte rows with 16-byte chunk ch of row r stored at slot ch ^ (((r & 3) << 2) | ((r >> 2) & 3)) (cdna guide
//      T10, "one image for row reads AND transposed reads", layout (b)): the K fragments (ds_read_b128, row per lane) and the V^T fragments
//      (ds_read_b64_tr_b16) are both conflict-free, and the permutation is applied on the DMA's per-lane SOURCE address (the DMA writes
//      lane-linear); addresses differ between k-steps / d-tiles by an XOR with a constant;
//    * inside a wave the scores of block 1 are issued before the softmax of block 0 and the PV product of block 0 before the softmax of
//      block 1 (the freed registers hold the second score set), so the wave's own softmax overlaps its matrix work.
//  LDS: 2 x (16 KiB K + 16 KiB V) + the tiles' key-padding masks = 64.5 KiB, two workgroups per CU.
// ------------------------------------------------------------------------------------------------------
template <bool STAMP = false>
__global__ __launch_bounds__(256, 2) void causal128_attn_kernel(AttnArgs p) {
    typedef bf16 T;
    constexpr int HD = 128, TR = 64, KSTEPS = 8, DT = 4, TILE = TR * 256, STG = 2 * TILE;
    constexpr float LOG2E = 1.4426950408889634f;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* kms = reinterpret_cast<int*>(smem + 2 * STG);   // [2][64] key-padding masks of the two staged tiles

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, ql = lane & 31;
    // grid (H * B, query blocks): the block index is the SLOWEST dimension and counts down, so that every workgroup of the heaviest query block
    // (it sees the most keys) is dispatched before any of the next one -- the dispatcher's greedy placement then is longest-first (with the
    // block index fastest, every CU slot drew 2-3 workgroups of random weight: 1152 workgroups on 512 slots finished 25 % apart).
    // Blocks are right-aligned (the ragged block is the one that sees one key tile).
    const int head = blockIdx.x % p.H, b = blockIdx.x / p.H, kvh = head / p.groups;
    const int nqb = gridDim.y;
    const int qblk = nqb - 1 - (int)blockIdx.y;
    const int qbase = p.Sq - nqb * 128 + qblk * 128;
    const int qi = qbase + wave * 32 + ql;
    const bool q_valid = qi >= 0 && qi < p.Sq;
    const int Sk = p.Sk;

    Frag<T> qf[KSTEPS];
    {
        const T* qp = reinterpret_cast<const T*>(p.q) + (long)b * p.q_bs + (long)qi * p.q_ts + (long)head * p.q_hs;
#pragma unroll
        for (int t = 0; t < KSTEPS; ++t) qf[t] = q_valid ? load_frag(qp + 16 * t + 8 * h) : zero_frag<T>();
    }
    const int last_q = p.q_pos0 + min(p.Sq, qbase + 128) - 1;
    const int ntiles = min((Sk + TR - 1) / TR, last_q / TR + 1);
    const int* kmask_g = p.key_mask ? p.key_mask + (long)b * Sk : nullptr;

    // ---- LDS-DMA requests of this wave: K pieces 4w .. 4w+3 and V pieces 4w .. 4w+3 (a piece = 4 key rows); lane -> row 4 piece + lane / 16,
    // physical slot lane % 16, which holds the logical chunk slot ^ sw(row), sw(row) = ((row & 3) << 2) | ((row >> 2) & 3) = ((lane >> 4) << 2) | j.
    // Addresses = wave-uniform tile base (scalar) + one 32-bit byte offset per piece and lane, the same for K and V (both caches have 256-byte rows).
    const char* kg = reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.k) + (long)b * p.k_bs + (long)kvh * p.k_hs);
    const char* vg = reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.v) + (long)b * p.v_bs + (long)kvh * p.v_hs);
    const int prow0 = 16 * wave + (lane >> 4);                    // piece j: row prow0 + 4 j
    const int pcb = ((lane & 15) ^ ((lane >> 4) << 2)) << 4;      // source chunk byte of piece j: pcb ^ (16 j)
    unsigned int poff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) poff[j] = (unsigned int)((prow0 + 4 * j) * 256 + (pcb ^ (16 * j)));
    int km_next = 1;
    auto request = [&](int t) __attribute__((always_inline)) {
        char* stage = smem + (t & 1) * STG;
        const char* kt0 = kg + (size_t)t * (TR * 256);
        const char* vt0 = vg + (size_t)t * (TR * 256);
        if ((t + 1) * TR <= Sk) {   // wave-uniform: a whole tile
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                __builtin_amdgcn_global_load_lds(GLB_PTR(kt0 + poff[j]), LDS_PTR(stage + (4 * wave + j) * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds(GLB_PTR(vt0 + poff[j]), LDS_PTR(stage + TILE + (4 * wave + j) * 1024), 16, 0, 0);
            }
        } else {   // the last, ragged tile: rows past the end re-read the last key (their scores are masked below)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned int off = (unsigned int)(min(prow0 + 4 * j, Sk - 1 - t * TR) * 256 + (pcb ^ (16 * j)));
                __builtin_amdgcn_global_load_lds(GLB_PTR(kt0 + off), LDS_PTR(stage + (4 * wave + j) * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds(GLB_PTR(vt0 + off), LDS_PTR(stage + TILE + (4 * wave + j) * 1024), 16, 0, 0);
            }
        }
        if (tid < TR) {
            const int kt = t * TR + tid;
            km_next = (kmask_g && kt < Sk) ? kmask_g[kt] : 1;
        }
    };
    // ---- fragment addresses (byte offsets inside a stage)
    const int swq = ((ql & 3) << 2) | ((ql >> 2) & 3);
    const int k_off = 256 * ql + ((16 * h) ^ (16 * swq));                 // K[sub * 32 + ql][16 ks + 8 h ..]: (k_off + 8192 sub) ^ (32 ks)
    int v_lo, v_hi;                                                      // V^T fragment rows kv0 + 4 hh + q4 (+ 8): (v_xx + 256 kv0) ^ (2 d0)
    {
        const int grp = lane >> 4, i = lane & 15, hh = grp >> 1, q4 = i >> 2, p4 = i & 3;
        const int c_lo = 2 * (grp & 1) + (p4 >> 1);
        v_lo = 256 * (4 * hh + q4) + 16 * (c_lo ^ ((q4 << 2) | hh)) + 8 * (p4 & 1);
        v_hi = 256 * (4 * hh + q4 + 8) + 16 * (c_lo ^ ((q4 << 2) | ((hh + 2) & 3))) + 8 * (p4 & 1);
    }
    f32x16 o[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float scale2 = p.scale * LOG2E;
    const int q_pos = p.q_pos0 + qi;
    const float FMIN = -3.4028234663852886e38f;  // torch.finfo(float32).min
    const int wave_first_q = p.q_pos0 + qbase + wave * 32;
    const int wave_last_q = p.q_pos0 + min(p.Sq, qbase + (wave + 1) * 32) - 1;
    const bool wave_live = (wave_first_q - p.q_pos0 < p.Sq) && (wave_last_q >= p.q_pos0);   // wave-uniform: this wave has queries

    auto qk_block = [&](const char* ks_, int sub, f32x16& s) __attribute__((always_inline)) {  // S^T = K . Q^T
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const Frag<T> a = load_frag(reinterpret_cast<const T*>(ks_ + ((k_off + 8192 * sub) ^ (32 * ks))));
            mma32(a, qf[ks], s);
        }
    };
    // scores -> unnormalised probabilities in place (online softmax, the reference's additive finfo.min masks); returns the O rescale factor.
    // INTERIOR (compile time; the caller checks it per tile, wave-uniform): the block lies entirely in the past of every query of this wave
    // and holds no padding key -- no mask arithmetic, the maximum is taken on the raw scores and the scale rides in the exponent's FMA.
    auto soft_block = [&](int tile, int sub, auto INTERIOR, f32x16& s) __attribute__((always_inline)) -> float {
        const int kbase = tile * TR + sub * 32;
        if constexpr (decltype(INTERIOR)::value) {
            float mr = s[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mr = fmaxf(mr, s[r]);
            float lo, hi;
            halves(mr, lo, hi);
            const float mx = fmaxf(lo, hi) * scale2;
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], scale2, -m_new));
                s[r] = pv;
                psum += pv;
            }
            l_run = __builtin_fmaf(l_run, alpha, psum);
            m_run = m_new;
            return alpha;
        } else {
            const int* km = kms + (tile & 1) * TR + sub * 32 + 4 * h;
            int kmv[16];   // crow32(r, h) = (r & 3) + 8 (r >> 2) + 4 h: registers 4 g .. 4 g + 3 are four consecutive keys
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int4 w = *reinterpret_cast<const int4*>(km + 8 * g);
                kmv[4 * g] = w.x; kmv[4 * g + 1] = w.y; kmv[4 * g + 2] = w.z; kmv[4 * g + 3] = w.w;
            }
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kt = kbase + crow32(r, h);
                float v = s[r] * scale2;
                float add = 0.f;   // additive masks exactly as the reference builds them (fp32): causal min + padding min
                if (kt > q_pos) add += FMIN;
                if (kmv[r] == 0) add += FMIN;
                if (add != 0.f) v = __builtin_fmaf(s[r], p.scale, add);   // = finfo.min (score absorbed) or -inf; deliberately not rescaled by log2(e)
                if (kt >= Sk) v = -INFINITY;
                asm volatile("" : "+v"(v));   // the product is rounded here, not contracted into the exponent's subtraction below
                s[r] = v;
                mx = fmaxf(mx, v);
            }
            {
                float lo, hi;
                halves(mx, lo, hi);
                mx = fmaxf(lo, hi);
            }
            const float m_new = fmaxf(m_run, mx);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);  // m_run = -inf -> 0
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(s[r] - m_use);
                s[r] = pv;
                psum += pv;
            }
            l_run = __builtin_fmaf(l_run, alpha, psum);
            m_run = m_new;
            return alpha;
        }
    };
    // O^T += V^T . P^T.  The transposed reads are inline asm: through the builtin hipcc waits vmcnt(0) before every ds_read_b64_tr_b16 while an
    // LDS-DMA is in flight (it cannot tell the tile being read from the tile being filled), which would serialise the prefetch.  The reads of
    // two d-tiles are issued together, then one counted wait (cdna guide 5.7: asm loads are not in hipcc's bookkeeping).
    auto pv_block = [&](unsigned int vs_lds, auto SUB, const f32x16& s) __attribute__((always_inline)) {
        constexpr int sub = decltype(SUB)::value;
        const Frag<T> p0 = pack_p(s, 0, (const T*)nullptr), p1 = pack_p(s, 1, (const T*)nullptr);
#pragma unroll
        for (int dp = 0; dp < DT; dp += 2) {
            s16x4 lo[2][2], hi[2][2];
#pragma unroll
            for (int dd = 0; dd < 2; ++dd) {
                const unsigned int a_lo = vs_lds + (unsigned int)(v_lo ^ (64 * (dp + dd))), a_hi = vs_lds + (unsigned int)(v_hi ^ (64 * (dp + dd)));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo[dd][0]) : "v"(a_lo), "i"(256 * (sub * 32)));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi[dd][0]) : "v"(a_hi), "i"(256 * (sub * 32)));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo[dd][1]) : "v"(a_lo), "i"(256 * (sub * 32 + 16)));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi[dd][1]) : "v"(a_hi), "i"(256 * (sub * 32 + 16)));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0][0]), "+v"(hi[0][0]), "+v"(lo[0][1]), "+v"(hi[0][1]), "+v"(lo[1][0]), "+v"(hi[1][0]), "+v"(lo[1][1]), "+v"(hi[1][1]) :: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int dd = 0; dd < 2; ++dd) {
                typedef __attribute__((ext_vector_type(8))) short s16x8;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    s16x8 t;
                    t[0] = lo[dd][half][0]; t[1] = lo[dd][half][1]; t[2] = lo[dd][half][2]; t[3] = lo[dd][half][3];
                    t[4] = hi[dd][half][0]; t[5] = hi[dd][half][1]; t[6] = hi[dd][half][2]; t[7] = hi[dd][half][3];
                    Frag<T> f;
                    f.v = __builtin_bit_cast(bf16x8_t, t);
                    mma32(f, half ? p1 : p0, o[dp + dd]);
                }
            }
        }
    };
    auto scale_o = [&](float alpha) __attribute__((always_inline)) {
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
    };

    // ---- prologue: tile 0 requested and landed
    request(0);
    if (tid < TR) kms[tid] = km_next;
    __syncthreads();   // (vmcnt(0) + barrier: the DMA of every wave has landed)
    unsigned long long st_dma = 0, st_cmp = 0, st_bar = 0, st_t0 = STAMP ? __builtin_amdgcn_s_memtime() : 0ull;
    const unsigned long long st_r0 = STAMP ? __builtin_amdgcn_s_memrealtime() : 0ull;   // (100 MHz, common to all CUs: loop entry)
    for (int tile = 0; tile < ntiles; ++tile) {
        const unsigned long long s0 = STAMP ? __builtin_amdgcn_s_memtime() : 0ull;
        if (tile + 1 < ntiles) request(tile + 1);   // into the buffer tile - 1 was read from: every wave has passed the barrier that ended it
        const unsigned long long s1 = STAMP ? __builtin_amdgcn_s_memtime() : 0ull;
        const char* Kst = smem + (tile & 1) * STG;
        const char* Vst = Kst + TILE;
        const bool tile_pad = kmask_g ? (__any(kms[(tile & 1) * TR + lane] == 0) != 0) : false;
        const int kb0 = tile * TR;
        // wave-uniform: which of the tile's two 32-key blocks this wave needs (a block entirely in the future of all its queries is skipped),
        // and whether the whole tile is interior (three straight-line bodies, no control flow inside a body)
        const bool do0 = wave_live && kb0 < Sk && kb0 <= wave_last_q;
        const bool do1 = wave_live && kb0 + 32 < Sk && kb0 + 32 <= wave_last_q;
        // per block (wave-uniform): entirely in the past of every query of this wave and free of padding -> no mask arithmetic
        const bool in0 = !tile_pad && (kb0 + 31 <= wave_first_q) && (kb0 + 31 < Sk);
        const bool in1 = !tile_pad && (kb0 + 63 <= wave_first_q) && (kb0 + 63 < Sk);   // (in1 implies in0)
        const unsigned int vs_lds = (unsigned int)(uintptr_t)LDS_PTR(Vst);
        auto one_block = [&](auto SUB, auto INTERIOR) __attribute__((always_inline)) {
            f32x16 sc;
            qk_block(Kst, decltype(SUB)::value, sc);
            const float alpha = soft_block(tile, decltype(SUB)::value, INTERIOR, sc);
            if (__any(alpha != 1.0f)) scale_o(alpha);   // wave-uniform: no row maximum moved in this block (the common case after the first tiles) -> 64 multiplies less
            pv_block(vs_lds, SUB, sc);
        };
        const std::integral_constant<int, 0> B0;
        const std::integral_constant<int, 1> B1;
        // five straight-line bodies, no control flow inside a body
        // (Issuing block 1's scores before block 0's softmax -- an in-wave pipeline -- was built and measured equal: with the second score set live
        // hipcc serialises the K fragment reads, one ds_read_b128 -> wait -> MFMA at a time, and the two waves of a SIMD already fill each other's gaps.)
        if (do1) {
            if (in1) { one_block(B0, std::true_type{}); one_block(B1, std::true_type{}); }
            else if (in0) { one_block(B0, std::true_type{}); one_block(B1, std::false_type{}); }
            else { one_block(B0, std::false_type{}); one_block(B1, std::false_type{}); }
        } else if (do0) {
            if (in0) one_block(B0, std::true_type{});
            else one_block(B0, std::false_type{});
        }
        const unsigned long long s2 = STAMP ? __builtin_amdgcn_s_memtime() : 0ull;
        if (tile + 1 < ntiles && tid < TR) kms[((tile + 1) & 1) * TR + tid] = km_next;
        __syncthreads();   // tile + 1 has landed (every wave waited for its own pieces) and nobody reads this tile's buffer any more
        if (STAMP) { const unsigned long long s3 = __builtin_amdgcn_s_memtime(); st_dma += s1 - s0; st_cmp += s2 - s1; st_bar += s3 - s2; }
    }
    if (STAMP && p.dbg && lane == 0) {   // [workgroup][wave][8]: request issue, compute, wait + barrier, whole loop, tiles, block index
        unsigned long long* d = p.dbg + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 8;
        d[0] = st_dma; d[1] = st_cmp; d[2] = st_bar; d[3] = __builtin_amdgcn_s_memtime() - st_t0; d[4] = ntiles; d[5] = qblk;
        d[6] = st_r0; d[7] = __builtin_amdgcn_s_memrealtime();
    }

    // ---- normalise and store: lane q holds O^T[d][q], d = 32 dt + crow32(r, h)
    float lo, hi;
    halves(l_run, lo, hi);
    const float l_tot = lo + hi;
    const float inv_l = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    if (q_valid) {
        T* op = reinterpret_cast<T*>(p.out) + (long)b * p.o_bs + (long)qi * p.o_ts + (long)head * p.o_hs;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq)
                store4(op + 32 * d + 8 * rq + 4 * h, make_float4(o[d][4 * rq] * inv_l, o[d][4 * rq + 1] * inv_l, o[d][4 * rq + 2] * inv_l, o[d][4 * rq + 3] * inv_l));
    }
}

static int launch_causal128(const AttnArgs& a, hipStream_t s) {
    constexpr int LDS = 2 * 2 * 64 * 256 + 2 * 64 * 4;
    static PerDeviceOnce attr;
    if (attr.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(causal128_attn_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(causal128_attn_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    }
    if (a.dbg) causal128_attn_kernel<true><<<dim3(a.H * a.B, (a.Sq + 127) / 128), dim3(256), LDS, s>>>(a);   // stamped diagnostic build (tools/probes/causal_stamps.py)
    else causal128_attn_kernel<false><<<dim3(a.H * a.B, (a.Sq + 127) / 128), dim3(256), LDS, s>>>(a);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------
//  vitglob_attn_kernel: SAM ViT-H global attention (image_encoder.py:224-240 with the decomposed rel-pos bias of :325-361) for bf16,
//  head_dim 80 on the 64 x 64 token grid -- the production path of the four global blocks.  Same mathematics and MFMA order as
//  flash_attn_kernel<bf16, 80, MODE_VIT_GLOBAL, 8, FAST64> (results equal to a rounding of the exponent's argument: tests/test_kernels_gpu.py),
//  different data movement and a leaner softmax.
//  The tiled kernel stages every 128-key K / V tile global -> registers -> LDS with two barriers per tile and drains its in-wave pipeline
//  at every tile end: matrix pipe busy 32 %, 6.6 M LDS bank conflicts per launch, waves parked 28 % of their cycles (round 3).  Here
//    * one 8-wave workgroup per 256 queries of a (image, head) pair; K and V arrive by LDS-DMA (global_load_lds_dwordx4) in 32-key stages
//      (a 32-key block is half a grid row, the unit the FAST64 softmax works in): per stage every wave issues ONE K piece and ONE V piece
//      (4 key rows of 256 bytes each), two to three stages ahead of their use; rings of four K and four V stages;
//      ONE s_barrier per stage behind a counted vmcnt(2) -- nothing is staged through registers and the wait never drains the DMA queue;
//    * the image of a stage is the causal kernel's: plain 256-byte rows, 16-byte chunk ch of row r at slot ch ^ (((r & 3) << 2) | ((r >> 2) & 3))
//      (cdna guide T10, layout (b): K row reads and transposed V reads both conflict-free), the permutation applied on the DMA's per-lane source
//      address.  A head's row is only 160 bytes: lanes whose chunk would be 10 .. 15 fetch a valid dummy (K) or the 32 constant bytes
//      {bf16 1.0, 0, ...} (V chunks 10 / 11: the column of ones at V[:, 80] that makes row 80 of O^T the softmax denominator, zeros after);
//    * a wave's pipeline runs THROUGH the stage boundaries (scores of block j + 1 and a PV product around the softmax of block j), and the two
//      waves of a SIMD (w, w + 4) run the interval's matrix half and softmax half in opposite order: one feeds the matrix pipe while the other is
//      in its exponentials; the O rescale is skipped when no row maximum of the wave moved (a multiply by exactly 1);
//    * workgroups are numbered so that the 16 query blocks of a (image, head) pair run on one XCD (its K / V, 1.3 MB, stay in that L2).
//  LDS: 32 KiB K ring + 32 KiB V ring + 22 KiB rel-pos table staging + 8 x 8 KiB rel_h tables = 150 KiB, one workgroup per CU.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void vitglob_attn_kernel(AttnArgs p) {
    typedef bf16 T;
    constexpr int HD = 80, KSTEPS = 5, DT = 3, G = 64, NB = 128, STAGE = 32 * 256, NE = 2 * G - 1;
    constexpr int RS = HD * 2 + 16;          // row stride of the rel-pos table staging area (the tiled kernel's K stride)
    constexpr int KRING = 0, VRING = 4 * STAGE, TSTAGE = 8 * STAGE, RELH = TSTAGE + 128 * RS;   // K and V rings of four stages: slot = j & 3, a compile-time constant in the loop unrolled by four
    constexpr float LOG2E = 1.4426950408889634f;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, ql = lane & 31;
    // workgroup -> ((image, head) pair, query block): ids go round robin over the 8 XCDs, so id % 8 picks the XCD; an XCD walks its pairs one
    // after the other, the 16 query blocks of a pair on neighbouring CUs of that XCD at the same time (speed only: any placement is correct)
    const int npair = p.B * p.H;
    int pair, qblk;
    {
        const int bid = blockIdx.x;
        if ((npair & 7) == 0) { const int xcd = bid & 7, idx = bid >> 3; pair = xcd + 8 * (idx >> 4); qblk = idx & 15; }
        else { pair = bid >> 4; qblk = bid & 15; }
    }
    const int head = pair % p.H, b = pair / p.H;
    const int qi = qblk * 256 + wave * 32 + ql;
    const int qh = qi >> 6, qw = qi & 63;

    Frag<T> qf[KSTEPS];
    {
        const T* qp = reinterpret_cast<const T*>(p.q) + (long)b * p.q_bs + (long)qi * p.q_ts + (long)head * p.q_hs;
#pragma unroll
        for (int t = 0; t < KSTEPS; ++t) qf[t] = load_frag(qp + 16 * t + 8 * h);
    }

    // ---- LDS-DMA requests of this wave: piece `wave` of a K stage and of a V stage (4 key rows).  lane -> row 4 wave + lane / 16, physical
    // slot lane % 16, which holds the logical chunk slot ^ sw(row), sw(row) = ((row & 3) << 2) | ((row >> 2) & 3) = ((lane >> 4) << 2) | (wave & 3)
    const long k_row = p.k_ts * 2, v_row = p.v_ts * 2;                       // bytes between consecutive keys of a head
    const char* kg = reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.k) + (long)b * p.k_bs + (long)head * p.k_hs);
    const char* vg = reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.v) + (long)b * p.v_bs + (long)head * p.v_hs);
    const int prow = 4 * wave + (lane >> 4);
    const int pch = (lane & 15) ^ (((lane >> 4) << 2) | (wave & 3));         // logical chunk this lane's slot holds
    const unsigned int k_off = (unsigned int)(prow * k_row) + 16u * (unsigned int)(pch < 10 ? pch : pch - 10);   // (chunks 10 .. 15 of a K row are never read: any valid bytes)
    const bool v_real = pch < 10;                                           // chunks 10 .. 15 of a V row: this lane requests nothing (the constant chunks 10 / 11 are written once, below)
    const unsigned int v_off = (unsigned int)(prow * v_row) + 16u * (unsigned int)(v_real ? pch : 0);
    // One request = one global_load_lds_dwordx4 in its SGPR-base form (wave-uniform 64-bit stage address + this lane's 32-bit byte offset), written as
    // asm: through the builtin hipcc builds a 64-bit per-lane address with four VALU instructions per request in a loop that is VALU-bound.  M0 (the
    // LDS base of the piece) is saved and restored inside the statement (cdna guide 5.7); the DMA has no register destination, and its completion is
    // counted by the loop's own s_waitcnt vmcnt.
    auto glds = [&](const char* base, unsigned int off, unsigned int lds_addr) __attribute__((always_inline)) {
        unsigned int keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(off), "s"(base), "s"(lds_addr) : "memory");
    };
    const unsigned int kring_lds = (unsigned int)(uintptr_t)LDS_PTR(smem + KRING) + (unsigned int)wave * 1024u;
    const unsigned int vring_dma = (unsigned int)(uintptr_t)LDS_PTR(smem + VRING) + (unsigned int)wave * 1024u;
    auto req_k = [&](int j) __attribute__((always_inline)) {                 // stage min(j, NB - 1) -> K ring slot j & 3
        const int js = min(j, NB - 1);
        glds(kg + (size_t)js * (32 * k_row), k_off, kring_lds + (unsigned int)((j & 3) * STAGE));
    };
    auto req_v = [&](int j) __attribute__((always_inline)) {
        const int js = min(j, NB - 1);
        if (v_real) glds(vg + (size_t)js * (32 * v_row), v_off, vring_dma + (unsigned int)((j & 3) * STAGE));   // (lanes of chunks >= 10: masked off, LDS untouched)
    };
    // chunks 10 / 11 of every V row of the four ring slots: {bf16 1.0, 0 x 7} and zeros -- the column of ones at V[:, 80] (row 80 of O^T = V^T P^T becomes
    // the softmax denominator) and the zero columns 81 .. 95; the DMA never writes them (its lanes for chunks >= 10 are masked off)
    if (tid < 256) {
        const int slot = tid >> 6, row = (tid >> 1) & 31, c = 10 + (tid & 1);
        *reinterpret_cast<uint4*>(smem + VRING + slot * STAGE + row * 256 + 16 * (c ^ (((row & 3) << 2) | ((row >> 2) & 3)))) = make_uint4(c == 10 ? 0x00003F80u : 0u, 0u, 0u, 0u);
    }
    req_k(0); req_v(0); req_k(1); req_v(1); req_k(2);   // land under the table phase

    // ---- rel-pos tables (the tiled kernel's arithmetic): T^T[e][q] = RelTable[e][:] . q[:] by MFMA, scattered to relh[kh][q]; the width table goes
    // through the same slab first and ends in 32 registers per lane (key columns 32 par + crow32(r, h)), both in log2 units
    float* relh = reinterpret_cast<float*>(smem + RELH) + wave * (64 * 32);
    char* Ts = smem + TSTAGE;
    float rw[2][16];
#pragma unroll 1
    for (int it = 0; it < 2; ++it) {
        const int tb = 1 - it;   // width table first (its slab is then recycled for rel_h)
        const T* const tab = reinterpret_cast<const T*>(tb == 0 ? p.rel_h : p.rel_w);
        __syncthreads();
        for (int idx = tid; idx < 128 * 10; idx += 512) {
            const int row = idx / 10, ch = idx - row * 10;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (row < NE) v = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(tab + (long)row * HD) + ch * 16);
            *reinterpret_cast<uint4*>(Ts + row * RS + ch * 16) = v;
        }
        __syncthreads();
        const int qc = tb == 0 ? qh : qw;
#pragma unroll 1
        for (int t = 0; t < 4; ++t) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                const Frag<T> a = load_frag(reinterpret_cast<const T*>(Ts + (32 * t + ql) * RS) + 16 * ks + 8 * h);
                mma32(a, qf[ks], acc);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int e = 32 * t + crow32(r, h);
                const int kk = qc + (G - 1) - e;  // rel index e = q - k + (G-1)  (get_rel_pos, image_encoder.py:318-322)
                if (e < NE && kk >= 0 && kk < G) relh[kk * 32 + ql] = acc[r] * LOG2E;
            }
        }
        if (tb == 1) {
            __syncthreads();  // both lane halves of every wave have scattered their rel_w entries
#pragma unroll
            for (int i = 0; i < 32; ++i) rw[i >> 4][i & 15] = relh[(32 * (i >> 4) + crow32(i & 15, h)) * 32 + ql];
        }
    }
    __syncthreads();   // (vmcnt(0) + barrier: the first five stages have landed too)

    // ---- fragment addresses inside a stage: one register per k-step / d-tile, the stage's ring slot is an immediate offset
    const int swq = ((ql & 3) << 2) | ((ql >> 2) & 3);
    const int kf_off = 256 * ql + ((16 * h) ^ (16 * swq));                 // K[ql][16 ks + 8 h ..] at kf_off ^ (32 ks)
    const char* kfa[KSTEPS];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) kfa[ks] = smem + KRING + (kf_off ^ (32 * ks));
    unsigned int va_lo[DT], va_hi[DT];                                   // V^T fragment rows 4 hh + q4 (+ 8) of a stage, d-tile d: ^ (64 d)
    {
        const int grp = lane >> 4, i = lane & 15, hh = grp >> 1, q4 = i >> 2, p4 = i & 3;
        const int c_lo = 2 * (grp & 1) + (p4 >> 1);
        const int v_lo = 256 * (4 * hh + q4) + 16 * (c_lo ^ ((q4 << 2) | hh)) + 8 * (p4 & 1);
        const int v_hi = 256 * (4 * hh + q4 + 8) + 16 * (c_lo ^ ((q4 << 2) | ((hh + 2) & 3))) + 8 * (p4 & 1);
        const unsigned int vring_lds = (unsigned int)(uintptr_t)LDS_PTR(smem + VRING);
#pragma unroll
        for (int d = 0; d < DT; ++d) { va_lo[d] = vring_lds + (unsigned int)(v_lo ^ (64 * d)); va_hi[d] = vring_lds + (unsigned int)(v_hi ^ (64 * d)); }
    }

    f32x16 o[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = -INFINITY;
    const float scale2 = p.scale * LOG2E;
    // The rel_w term of a score rides in the MFMA's initial accumulator, in units of the raw score (rw / scale2): the softmax then needs ONE fma
    // and one exponential per score -- p = 2^(s' scale2 - m_off) with s' = k.q + rw / scale2 -- and takes its maximum on the raw s' (scale2 > 0).
    // (The tiled kernel rounds s scale2 + rw first and subtracts the offset after: same mathematics, one rounding apart.)
    {
        const float inv_scale2 = 1.0f / scale2;
#pragma unroll
        for (int i = 0; i < 32; ++i) rw[i >> 4][i & 15] *= inv_scale2;
    }

    auto qk_block = [&](auto SLOT, auto PAR, f32x16& s) __attribute__((always_inline)) {  // S'^T = K . Q^T + rel_w / scale2, K ring slot SLOT
        constexpr int slot = decltype(SLOT)::value, par = decltype(PAR)::value;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = rw[par][r];
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const Frag<T> a = load_frag(reinterpret_cast<const T*>(kfa[ks] + slot * STAGE));
            mma32(a, qf[ks], s);
        }
    };
    // raw scores -> unnormalised probabilities in place (online softmax); returns the factor the running O must be scaled by.  A 32-key block is half
    // a grid row: its rel_h term is uniform over the block (added to the block maximum, folded into the exponent's offset)
    auto soft_block = [&](int j, f32x16& s) __attribute__((always_inline)) -> float {
        const float rh64 = relh[(j >> 1) * 32 + ql];
        // (max16_raw: the scores were written by MFMAs at least one PV product or one barrier earlier)
        const float mr = max16_raw(s);
        float lo, hi;
        halves(mr, lo, hi);
        const float mx = __builtin_fmaf(max3_raw(lo, hi, hi), scale2, rh64);
        // The running maximum only has to keep the exponentials in range, so it FOLLOWS the true maximum lazily: it moves (and O is rescaled) when
        // some row of the wave exceeds it by more than 2^LAZY, not at every new record (random scores set a record in some row of a 32-row wave in
        // ~60 % of the 128 blocks; each rescale is 24 v_pk_mul_f32).  Until then the probabilities of that row are at most 2^LAZY = 8 instead of 1 -- the
        // same relative precision in bf16, fp32 sums -- and the final O / l is unchanged up to rounding (tests: no worse than the tiled kernel
        // against a float64 softmax).  (cdna guide T13; the decision covers the whole block, and PV(j-1) is complete before O is rescaled.)
        constexpr float LAZY = 3.0f;
        float alpha = 1.0f;
        if (__any(mx > m_run + LAZY)) {      // wave-uniform (m_run = -inf at the first block: taken)
            const float m_new = fmaxf(m_run, mx);     // (every block holds 32 finite scores: no -inf bookkeeping)
            alpha = __builtin_amdgcn_exp2f(m_run - m_new);  // m_run = -inf -> 0
            m_run = m_new;
        }
        const float m_off = rh64 - m_run;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], scale2, m_off));  // raw v_exp_f32: underflow to 0 is the intent
        return alpha;
    };
    // O^T += V^T . P^T, V ring slot SLOT.  Transposed reads as inline asm (through the builtin hipcc waits vmcnt(0) before every ds_read_b64_tr_b16
    // while an LDS-DMA is in flight), the reads of a d-tile issued together, one counted wait per d-tile.
    auto pv_block = [&](auto SLOT, const Frag<T>& p0, const Frag<T>& p1) __attribute__((always_inline)) {
        constexpr int slot = decltype(SLOT)::value;
        // all twelve transposed reads of the block are issued first (24 registers), then each d-tile's pair of MFMAs behind a COUNTED wait: LDS
        // operations return in order, so "at most 8 / 4 / 0 outstanding" covers d-tile 0 / 1 / 2 whatever else the compiler has queued behind them
        s16x4 lo[DT][2], hi[DT][2];
#pragma unroll
        for (int d = 0; d < DT; ++d) {
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo[d][0]) : "v"(va_lo[d]), "i"(slot * STAGE));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi[d][0]) : "v"(va_hi[d]), "i"(slot * STAGE));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo[d][1]) : "v"(va_lo[d]), "i"(slot * STAGE + 4096));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi[d][1]) : "v"(va_hi[d]), "i"(slot * STAGE + 4096));
        }
#pragma unroll
        for (int d = 0; d < DT; ++d) {
            if (d == 0) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(lo[0][0]), "+v"(hi[0][0]), "+v"(lo[0][1]), "+v"(hi[0][1]) :: "memory");
            else if (d == 1) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(lo[1][0]), "+v"(hi[1][0]), "+v"(lo[1][1]), "+v"(hi[1][1]) :: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[2][0]), "+v"(hi[2][0]), "+v"(lo[2][1]), "+v"(hi[2][1]) :: "memory");
            __builtin_amdgcn_sched_barrier(0);
            typedef __attribute__((ext_vector_type(8))) short s16x8;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                s16x8 t;
                t[0] = lo[d][half][0]; t[1] = lo[d][half][1]; t[2] = lo[d][half][2]; t[3] = lo[d][half][3];
                t[4] = hi[d][half][0]; t[5] = hi[d][half][1]; t[6] = hi[d][half][2]; t[7] = hi[d][half][3];
                Frag<T> f;
                f.v = __builtin_bit_cast(bf16x8_t, t);
                mma32(f, half ? p1 : p0, o[d]);
            }
        }
    };
    auto scale_o = [&](float alpha) __attribute__((always_inline)) {
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
    };

    // ---- main loop.  Per interval (one s_barrier) every wave requests K(j+3) and V(j+2), computes the scores of block j+1, the softmax of block j
    // and one PV product.  The two waves of a SIMD (wave w and w + 4) run the interval's two halves in OPPOSITE order, so that one is in its matrix
    // half while the other is in its softmax (VALU) half:
    //     waves 0-3:  scores(j+1), PV(j-1) | softmax(j)            waves 4-7:  softmax(j) | scores(j+1), PV(j)
    // (per wave the order of operations on O is the same in both: ... x alpha_j, + PV(j), x alpha_j+1 ...).
    //   RAW: K(j+1) was requested in interval j-2, V(j-1) / V(j) in intervals j-3 / j-2, and the vmcnt(2) that ended interval j-1 left only that
    //        interval's own two requests in flight, behind the barrier every wave passed.
    //   WAR: K(j+3) goes to the slot of K(j-1), last read in interval j-2; V(j+2) to the slot of V(j-2), last read in interval j-1 (waves 0-3).
    //   Past the last stage the requests repeat stage NB-1 into slots nobody reads any more, so the counted wait needs no special case.
    // Four intervals per loop trip: ring slots, rel_w parity and score registers are compile-time constants.
    f32x16 sc0, sc1;
    Frag<T> p0, p1;
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 2> I2;
    typedef std::integral_constant<int, 3> I3;
    qk_block(I0{}, I0{}, sc0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15" ::: "memory");   // the only place where a softmax (waves 4-7, asm v_max3) directly follows the MFMAs that wrote its scores
    __builtin_amdgcn_sched_barrier(0);
    auto finish = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    auto soft_pack = [&](int j, f32x16& s_cur) __attribute__((always_inline)) {
        const float alpha = soft_block(j, s_cur);
        if (__any(alpha != 1.0f)) scale_o(alpha);   // wave-uniform: no row maximum moved (the common case after the first stages): 48 multiplies by 1 skipped
        p0 = pack_p(s_cur, 0, (const T*)nullptr);
        p1 = pack_p(s_cur, 1, (const T*)nullptr);
    };
    if (wave < 4) {
        // interval j with j & 3 == JM: scores of block j+1 (slot (JM+1)&3, parity (JM+1)&1), PV of block j-1 (slot (JM+3)&3), softmax of block j
        auto interval = [&](int j, auto JM, f32x16& s_cur, f32x16& s_next) __attribute__((always_inline)) {
            constexpr int jm = decltype(JM)::value;
            req_k(j + 3);
            req_v(j + 2);
            if (j + 1 < NB) qk_block(std::integral_constant<int, (jm + 1) & 3>{}, std::integral_constant<int, (jm + 1) & 1>{}, s_next);
            if (j > 0) pv_block(std::integral_constant<int, (jm + 3) & 3>{}, p0, p1);
            soft_pack(j, s_cur);
            finish();
        };
#pragma unroll 1
        for (int j = 0; j < NB; j += 4) {
            interval(j, I0{}, sc0, sc1);
            interval(j + 1, I1{}, sc1, sc0);
            interval(j + 2, I2{}, sc0, sc1);
            interval(j + 3, I3{}, sc1, sc0);
        }
        pv_block(std::integral_constant<int, (NB - 1) & 3>{}, p0, p1);
    } else {
        auto interval = [&](int j, auto JM, f32x16& s_cur, f32x16& s_next) __attribute__((always_inline)) {
            constexpr int jm = decltype(JM)::value;
            soft_pack(j, s_cur);
            req_k(j + 3);     // (waves 4-7 issue their requests after their softmax half: the eight waves' sixteen requests of an interval then do not queue
            req_v(j + 2);     //  at the address unit all at once behind the barrier -- 449 -> 424 us per launch)
            if (j + 1 < NB) qk_block(std::integral_constant<int, (jm + 1) & 3>{}, std::integral_constant<int, (jm + 1) & 1>{}, s_next);
            pv_block(JM, p0, p1);
            finish();
        };
#pragma unroll 1
        for (int j = 0; j < NB; j += 4) {
            interval(j, I0{}, sc0, sc1);
            interval(j + 1, I1{}, sc1, sc0);
            interval(j + 2, I2{}, sc0, sc1);
            interval(j + 3, I3{}, sc1, sc0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the two dummy requests of the last interval (nothing may be in flight into LDS at the end)

    // ---- normalise and store: lane q holds O^T[d][q], d = 32 dt + crow32(r, h); row 80 of O^T (d-tile 2, register 8 of the lower lane half) is the denominator
    float lo, hi;
    halves(o[DT - 1][8], lo, hi);
    const float l_tot = lo;
    const float inv_l = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    T* op = reinterpret_cast<T*>(p.out) + (long)b * p.o_bs + (long)qi * p.o_ts + (long)head * p.o_hs;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const int dd = 32 * d + 8 * rq + 4 * h;
            if (dd < HD) store4(op + dd, make_float4(o[d][4 * rq] * inv_l, o[d][4 * rq + 1] * inv_l, o[d][4 * rq + 2] * inv_l, o[d][4 * rq + 3] * inv_l));
        }
}

static int launch_vitglob(const AttnArgs& a, hipStream_t s) {
    constexpr int LDS = 8 * 32 * 256 + 128 * (80 * 2 + 16) + 8 * 64 * 32 * 4;
    static PerDeviceOnce attr;
    if (attr.first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(vitglob_attn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    vitglob_attn_kernel<<<dim3(a.B * a.H * 16), dim3(512), LDS, s>>>(a);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

static int g_attn_variant = 0;
static unsigned long long* g_attn_dbg = nullptr;
extern "C" int ullsam_set_attn_variant(int v) { g_attn_variant = v; return 0; }
extern "C" int ullsam_set_attn_debug(void* stamps) { g_attn_dbg = reinterpret_cast<unsigned long long*>(stamps); return 0; }

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
        // SAM's own shape (14x14 windows, head_dim 80, bf16): the whole-window kernel; variant 2 forces the tiled kernel for A/B
        a.q_pos0 = (g_attn_variant >= 3 && g_attn_variant <= 8) ? g_attn_variant - 3 : 2;  // stagger of the second resident workgroup, units of s_sleep(127) = 3.4 us (A/B: variant 3 + n)
        // round 5: window rows as 16-wide query groups / key tiles, no key-block chain (win14r_attn_kernel); variant 13 keeps round 1's 32-query kernel (A/B, equality test)
        if (dtype == 1 && window == 14 && hd == 80 && g_attn_variant != 1 && g_attn_variant != 2 && g_attn_variant != 13 && (((uintptr_t)qkv | (uintptr_t)out | (uintptr_t)qkv_bias | (uintptr_t)rel_h | (uintptr_t)rel_w) & 15) == 0) {   // (the kernel reads the rel-pos tables 16 bytes at a time too)
            a.q_pos0 = (g_attn_variant >= 3 && g_attn_variant <= 8) ? g_attn_variant - 3 : 0;
            a.dbg = g_attn_dbg;
            return launch_win14r(a, s, (g_attn_variant >= 20 && g_attn_variant <= 22) ? g_attn_variant - 19 : 0);
        }
        if (dtype == 1 && window == 14 && hd == 80 && g_attn_variant != 1 && g_attn_variant != 2) return launch_win14(a, s);
        // window = 196 queries: one 7-wave workgroup (128-key tiles) or two 4-wave workgroups (64-key tiles, 3 resident per CU)
        if (g_attn_variant == 1) return dtype == 0 ? dispatch_hd<float, MODE_VIT_WINDOW, 7>(a, hd, s) : dispatch_hd<bf16, MODE_VIT_WINDOW, 7>(a, hd, s);
        return dtype == 0 ? dispatch_hd<float, MODE_VIT_WINDOW, 7>(a, hd, s) : dispatch_hd<bf16, MODE_VIT_WINDOW, 4>(a, hd, s);
    }
    a.Sq = a.Sk = (int)N;
    // bf16 on SAM's 64x64 grid: one 8-wave workgroup per 256 queries over 128-key tiles (a staged K/V tile feeds twice the queries: half the
    // staging traffic, LDS writes and barriers per score) -- identical results, 543 vs 576 us per ViT-H layer at batch 4 (same-process A/B,
    // tools/probes/attn8_check.py); variant 9 keeps the two 4-wave workgroups with 64-key tiles
    // SAM ViT-H's own shape (head_dim 80): the LDS-DMA kernel; variant 12 keeps the tiled 8-wave kernel (bit-equality test, A/B)
    if (dtype == 1 && grid_h == 64 && grid_w == 64 && hd == 80 && g_attn_variant != 9 && g_attn_variant != 12 && (((uintptr_t)qkv | (uintptr_t)out) & 15) == 0)
        return launch_vitglob(a, s);
    if (dtype == 1 && grid_h == 64 && grid_w == 64 && g_attn_variant != 9) return dispatch_hd<bf16, MODE_VIT_GLOBAL, 8>(a, hd, s);
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
    a.dbg = g_attn_dbg;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // bf16, head_dim 128, contiguous 256-byte key rows (the KV-cache layout): the LDS-DMA kernel (variant 11 keeps the tiled kernel, for the
    // bit-equality test and A/B)
    if (dtype == 1 && hd == 128 && g_attn_variant != 11 && a.k_ts == 128 && a.v_ts == 128 && (((uintptr_t)k | (uintptr_t)v | (uintptr_t)q | (uintptr_t)out) & 15) == 0)
        return launch_causal128(a, s);
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
    static PerDeviceOnce attr0, attr1;
    if (dtype == 0) {
        if (attr0.first()) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(naive_attn_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
        naive_attn_kernel<float><<<dim3(Sq, H, B), dim3(256), lds, s>>>(a);
    } else {
        if (attr1.first()) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(naive_attn_kernel<bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
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
__device__ __forceinline__ void bf16x8_to_f32_attn(const uint4 u, float (&o)[8]) {
    const unsigned int d[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) { o[2 * j] = __uint_as_float(d[j] << 16); o[2 * j + 1] = __uint_as_float(d[j] & 0xffff0000u); }
}
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

// ---- the same partial pass on the matrix pipe (bf16 K / V, T <= 16; round 5).  tok2img_partial_kernel multiplies on the VALU: ~35 instructions per (key row, query)
// put it at 58 - 65 us per call whatever the bytes (one shared 1 MB K / V set in layer 0 or 134 MB of per-prompt keys).  Here a wave walks its keys 32 at a time:
// K rows go straight from HBM into A fragments (lane (m, g) takes the 16 bytes of key m it multiplies: whole 64-byte pieces of a row per instruction), V rows are
// staged coalesced into a wave-private LDS image (288-byte rows: the transposed reads are conflict-free), the NEXT 32 keys' loads of both in flight in registers.
// Scores S^T[key, q] per head on v_mfma_f32_16x16x32_bf16 with a head PAIR's 32 dims as one k-step and the other head's half of the query fragment zeroed; the
// fp32 queries enter as TWO bf16 terms (hi + lo, kept in LDS: two MFMAs per score tile) so that the scores carry K's rounding only, as in the VALU kernel; one
// running maximum per (query, head) agreed over the four lane groups; P^T feeds O^T[dim, q] += V^T P^T from the accumulators.
// Same workspace layout as tok2img_partial_kernel (the merge kernel is shared).
__global__ __launch_bounds__(256, 2) void tok2img_partial_mfma_kernel(const float* __restrict__ q, const bf16* __restrict__ k, const bf16* __restrict__ v, float* __restrict__ ws,
                                                                     int T, int N, long k_bs, long v_bs, float scale, int nsplit) {
    constexpr int C = 128, HD = 16, H = 8, RP = 288, STEP = 32;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) char stage[4][STEP * RP];         // per wave: the V image; afterwards the waves' partial outputs (so)
    __shared__ __attribute__((aligned(16))) bf16 sq[2][H][64][8];             // query B fragments, hi and lo terms, per head, in lane order
    __shared__ float sml[4][16][H][2];
    float (*so)[16][C] = reinterpret_cast<float (*)[16][C]>(&stage[0][0]);     // [4][16][C] floats = 32 KB of the 36 KB
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int split = blockIdx.x, pr = blockIdx.y;
    const int m = lane & 15, g = lane >> 4;
    const int per = (N + nsplit - 1) / nsplit;
    const int s0 = split * per, s1 = min(N, s0 + per);
    const int nst = (s1 - s0 + STEP - 1) / STEP;                               // 32-key steps of this split, dealt to the 4 waves round robin (<= 0: an empty split)
    const bf16* kb = k + (long)pr * k_bs;
    const bf16* vb = v + (long)pr * v_bs;
    char* Vi = stage[wv];
    // query fragments: B operand of head h = 2 hp + (g >> 1) half: lane (q = m, g) holds q[q][32 hp + 8 g .. + 7] where that is head h's, zeros elsewhere; scaled by scale * log2(e)
    const float sc2 = scale * 1.4426950408889634f;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int h = 2 * wv + hh, hp = wv;
        Frag<bf16> fh, fl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = ((g >> 1) == hh && m < T) ? q[((long)pr * T + m) * C + 32 * hp + 8 * g + j] * sc2 : 0.f;
            fh.v[j] = (bf16)x;
            fl.v[j] = (bf16)(x - (float)fh.v[j]);
        }
        *reinterpret_cast<bf16x8_t*>(&sq[0][h][lane][0]) = fh.v;
        *reinterpret_cast<bf16x8_t*>(&sq[1][h][lane][0]) = fl.v;
    }
    __syncthreads();
    f32x4 o[H];
    float mx[H], ls[H];                                                        // running maximum (log2 units of the SCALED scores), this lane's partial sum (its 4 + 4 keys per step)
#pragma unroll
    for (int h = 0; h < H; ++h) { o[h] = f32x4{0.f, 0.f, 0.f, 0.f}; mx[h] = -1e30f; ls[h] = 0.f; }
    u32x4 kr[8], vr[8];
    const bf16* kl = kb + 8 * g;                                               // K fragment j = 2 hp + half: key 16 half + m, dims 32 hp + 8 g .. + 7
    const bf16* vl = vb + 8 * (lane & 15);                                    // 32 rows x 256 B of V per step: lane l takes 16-byte chunk l & 15 of rows (l >> 4) + 4 j
    auto fetch = [&](const int st_) __attribute__((always_inline)) {
        const int r0 = s0 + max(st_, 0) * STEP;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const long rk = min(r0 + 16 * (j & 1) + m, N - 1), rv = min(r0 + g + 4 * j, N - 1);
            kr[j] = *reinterpret_cast<const u32x4*>(kl + rk * C + 32 * (j >> 1));
            vr[j] = *reinterpret_cast<const u32x4*>(vl + rv * C);
        }
    };
    fetch(min(wv, nst - 1));
    for (int st = wv; st < nst; st += 4) {
        Frag<bf16> kf[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            kf[j].v = __builtin_bit_cast(bf16x8_t, kr[j]);
            *reinterpret_cast<u32x4*>(Vi + (g + 4 * j) * RP + 16 * (lane & 15)) = vr[j];
        }
        fetch(min(st + 4, nst - 1));                                           // (the last steps re-fetch a step nobody uses: no branch around the loads)
        const int valid = min(STEP, s1 - (s0 + st * STEP));                    // keys of this step inside the split (wave-uniform)
        int lq = lane * 8;
        asm volatile("" : "+v"(lq));                                           // (opaque per step: otherwise the 16 query fragments are hoisted out of the loop into 64 VGPRs)
        const bf16* sqp = &sq[0][0][0][0] + lq;
        // ---- per head: scores S^T (16 keys x 16 queries, two key halves, two query terms), softmax bookkeeping, O^T += V^T P^T
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const int hp = h >> 1;
            const Frag<bf16> qh = load_frag(sqp + h * 512), ql = load_frag(sqp + (H + h) * 512);
            f32x4 sa = {0.f, 0.f, 0.f, 0.f}, sb = sa;
            mma16(kf[2 * hp], ql, sa);
            mma16(kf[2 * hp + 1], ql, sb);
            mma16(kf[2 * hp], qh, sa);
            mma16(kf[2 * hp + 1], qh, sb);
            // lane (q, g) holds keys 4 g + i of both halves
            float tm = -1e30f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sa[i] = (4 * g + i < valid) ? sa[i] : -1e30f;
                sb[i] = (16 + 4 * g + i < valid) ? sb[i] : -1e30f;
                tm = fmaxf(tm, fmaxf(sa[i], sb[i]));
            }
            tm = fmaxf(tm, lane_xor16(tm));
            tm = fmaxf(tm, lane_xor32(tm));
            const float mn = fmaxf(mx[h], tm);
            const float al = __builtin_amdgcn_exp2f(mx[h] - mn);
            mx[h] = mn;
            float ps = 0.f;
            Frag<bf16> pf, pl;                                                 // the probabilities as two bf16 terms too: the output carries V's rounding only
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float pa = __builtin_amdgcn_exp2f(sa[i] - mn), pb = __builtin_amdgcn_exp2f(sb[i] - mn);
                ps += pa + pb;
                pf.v[i] = (bf16)pa;
                pf.v[4 + i] = (bf16)pb;
                pl.v[i] = (bf16)(pa - (float)pf.v[i]);
                pl.v[4 + i] = (bf16)(pb - (float)pf.v[4 + i]);
            }
            ls[h] = ls[h] * al + ps;
            o[h] *= al;
            // O^T[dim 16 h + 4 g + i, q] += V^T P^T over the 32 keys: A = V^T by transposed reads (lane 4 q' + p of a 16-lane group addresses key row q', dims 4 p .. 4 p + 3)
            const char* a0 = Vi + (4 * g + (m >> 2)) * RP + 32 * h + 8 * (m & 3);
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 16 * RP));
            typedef __attribute__((ext_vector_type(8))) short s16x8;
            s16x8 tv;
            tv[0] = lo[0]; tv[1] = lo[1]; tv[2] = lo[2]; tv[3] = lo[3];
            tv[4] = hi[0]; tv[5] = hi[1]; tv[6] = hi[2]; tv[7] = hi[3];
            Frag<bf16> vf;
            vf.v = __builtin_bit_cast(bf16x8_t, tv);
            mma16(vf, pl, o[h]);
            mma16(vf, pf, o[h]);
        }
    }
    // ---- this wave's partial: sums over the four lane groups, then the 4 waves are merged as in tok2img_partial_kernel (natural-log units in the workspace)
    constexpr float LN2 = 0.6931471805599453f;
    __syncthreads();                                                           // every wave is done with its stage images: the region becomes `so`
#pragma unroll
    for (int h = 0; h < H; ++h) {
        float l = ls[h];
        l += lane_xor16(l);
        l += lane_xor32(l);
        if (m < T) {
#pragma unroll
            for (int i = 0; i < 4; ++i) so[wv][m][HD * h + 4 * g + i] = o[h][i];
            if (g == 0) { sml[wv][m][h][0] = mx[h] * LN2; sml[wv][m][h][1] = l; }
        }
    }
    __syncthreads();
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
    ULLSAM_CHECK(T >= 1 && T <= (kv_dtype == ULLSAM_DT_BF16 ? 16 : 8) && N >= 1 && nsplit >= 1 && P >= 1, "tok2img_attention: T=%d (1..8, bf16: 1..16) N=%d nsplit=%d P=%d", T, N, nsplit, P);
    ULLSAM_CHECK((((uintptr_t)k | (uintptr_t)v) & 15) == 0, "tok2img_attention: k/v must be 16-byte aligned");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid(nsplit, P);
    if (kv_dtype == ULLSAM_DT_F32) tok2img_partial_kernel<float, 8><<<grid, 256, 0, s>>>(q, k, v, workspace, T, N, k_batch_stride, v_batch_stride, scale, nsplit);
    else if (g_attn_variant == 16 && T <= 8) tok2img_partial_kernel<bf16, 8><<<grid, 256, 0, s>>>(q, k, v, workspace, T, N, k_batch_stride, v_batch_stride, scale, nsplit);   // (variant 16: the VALU kernel, for A/B and the equality test)
    else tok2img_partial_mfma_kernel<<<grid, 256, 0, s>>>(q, static_cast<const bf16*>(k), static_cast<const bf16*>(v), workspace, T, N, k_batch_stride, v_batch_stride, scale, nsplit);
    ULLSAM_LAUNCH_CHECK();
    tok2img_merge_kernel<<<P * T, 128, 0, s>>>(workspace, out, P, T, nsplit);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// ---- decode-step attention (q_len == 1, modeling_internlm2.py:383-419 with the mask of :834): one query per head against the cache.
// The G = H / KVH query heads that share a KV head are handled together ("G queries"), the cached keys are split over nsplit
// workgroups (flash-decoding) and merged by decode_attn_merge_kernel.  K / V rows are 256 bytes (hd 128, bf16): 16 lanes per row,
// 4 rows per wave load, fully coalesced; the dot product reduces over the 16 lanes with four DPP steps.  Key padding is the
// reference's additive finfo.min.
// lane ^ 16 / lane ^ 32 exchanges on gfx950's row-swap instructions (one VALU op + a select) instead of a ds_bpermute round trip:
// v_permlane16_swap exchanges the odd rows of its first operand with the even rows of its second, v_permlane32_swap the upper half of
// the first with the lower half of the second; with both operands = v, a lane's partner value is in one of the two results.
// Sum over the 16 lanes of a DPP row, result in every lane: quad exchanges, then the half-row and row mirrors (after the quad steps a
// lane's mirror partner holds the other quad's / half's sum).  Four VALU instructions instead of four ds_bpermute round trips; the
// additions are those of the xor butterfly (own group's sum + partner group's sum).
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));  // row_mirror
    return v;
}
template <int TQ>
__global__ __launch_bounds__(256) void decode_attn_partial_kernel(const bf16* __restrict__ q, const bf16* __restrict__ kc,
                                                                  const bf16* __restrict__ vc, const int* __restrict__ key_mask,
                                                                  float* __restrict__ ws, int G, int KVH, int Sk, long kv_bs,
                                                                  long kv_hs, float scale, int nsplit
#ifdef ULLSAM_STAMP_DECODE
                                                                  , unsigned long long* dbg
#endif
                                                                  ) {
#ifdef ULLSAM_STAMP_DECODE
    unsigned long long st[8]; st[0] = __builtin_amdgcn_s_memtime();
#define DSTAMP(i) st[i] = __builtin_amdgcn_s_memtime()
#else
#define DSTAMP(i)
#endif
    constexpr int HD = 128, EPL = 8, LPR = 16, RPW = 4;
    __shared__ float so[4][TQ][HD];
    __shared__ float sml[4][TQ][2];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int split = blockIdx.x, pr = blockIdx.y;       // pr = b * KVH + kvh
    const int b = pr / KVH, kvh = pr - b * KVH;
    const int c = lane % LPR, r = lane / LPR;
    const int per = (Sk + nsplit - 1) / nsplit;
    const int s0 = split * per, s1 = min(Sk, s0 + per);
    const float FMIN = -3.4028234663852886e38f;
    const bf16* kb = kc + (long)b * kv_bs + (long)kvh * kv_hs;
    const bf16* vb = vc + (long)b * kv_bs + (long)kvh * kv_hs;
    const int* km = key_mask ? key_mask + (long)b * Sk : nullptr;
    // CH row groups (4 rows each) per trip, their K / V / mask loads issued together; the first trip's are requested before the queries
    // are even converted and the next trip's before the current one is multiplied (sched_barriers keep the scheduler from sinking the
    // loads back next to their uses): a workgroup's share of the cache costs one exposed memory latency in all.
    constexpr int CH = 4;
    struct Chunk { uint4 k[CH], v[CH]; int mk[CH]; };
    auto request = [&](Chunk& ck, const int base) {
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int row = min(base + 4 * RPW * j + r, Sk - 1);
            const long off = (long)row * HD + c * EPL;
            ck.k[j] = *reinterpret_cast<const uint4*>(kb + off);
            ck.v[j] = *reinterpret_cast<const uint4*>(vb + off);
            ck.mk[j] = km ? km[row] : 1;
        }
    };
    Chunk cur, nxt;
    uint4 qraw[TQ];
#pragma unroll
    for (int t = 0; t < TQ; ++t) qraw[t] = t < G ? *reinterpret_cast<const uint4*>(q + ((long)pr * G + t) * HD + c * EPL) : make_uint4(0, 0, 0, 0);
    const int base0 = s0 + wv * RPW;
    request(cur, base0);
    __builtin_amdgcn_sched_barrier(0);
    float qf[TQ][EPL];
#pragma unroll
    for (int t = 0; t < TQ; ++t) bf16x8_to_f32_attn(qraw[t], qf[t]);
    float m[TQ], l[TQ], o[TQ][EPL];
#ifdef ULLSAM_STAMP_DECODE
    DSTAMP(1);
#endif
#pragma unroll
    for (int t = 0; t < TQ; ++t) {
        m[t] = -1e30f; l[t] = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) o[t][e] = 0.f;
    }
    for (int base = base0; base < s1; base += 4 * RPW * CH) {
        const bool more = base + 4 * RPW * CH < s1;
        if (more) request(nxt, base + 4 * RPW * CH);
        __builtin_amdgcn_sched_barrier(0);
#ifdef ULLSAM_STAMP_DECODE
        if (base == base0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); DSTAMP(2); }
#endif
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int row0 = base + 4 * RPW * j;
            if (row0 >= s1) break;                      // wave-uniform
            const bool valid = row0 + r < s1;
            float kf[EPL], vf[EPL];
            bf16x8_to_f32_attn(cur.k[j], kf);
            bf16x8_to_f32_attn(cur.v[j], vf);
            const float add = (valid && cur.mk[j] == 0) ? FMIN : 0.f;
#pragma unroll
            for (int t = 0; t < TQ; ++t) {
                float sc = 0.f;
#pragma unroll
                for (int e = 0; e < EPL; ++e) sc += qf[t][e] * kf[e];
                sc = row16_sum(sc);                     // the 16 lanes of a key row (same sums as the xor butterfly, without LDS)
                sc = sc * scale + add;
                const float mn = fmaxf(m[t], sc);
                const float al = __expf(m[t] - mn), pv = valid ? __expf(sc - mn) : 0.f;
                l[t] = l[t] * al + pv;
#pragma unroll
                for (int e = 0; e < EPL; ++e) o[t][e] = o[t][e] * al + pv * vf[e];
                m[t] = mn;
            }
        }
        if (more) cur = nxt;
    }
    DSTAMP(3);
    auto merge_rows = [&](auto partner) {  // merge the RPW row groups of the wave: lane <-> lane ^ 16, then lane ^ 32
#pragma unroll
        for (int t = 0; t < TQ; ++t) {
            const float m2 = partner(m[t]), l2 = partner(l[t]);
            const float mn = fmaxf(m[t], m2);
            const float a1 = __expf(m[t] - mn), a2 = __expf(m2 - mn);
            l[t] = l[t] * a1 + l2 * a2;
#pragma unroll
            for (int e = 0; e < EPL; ++e) o[t][e] = o[t][e] * a1 + partner(o[t][e]) * a2;
            m[t] = mn;
        }
    };
    merge_rows([](float v) { return lane_xor16(v); });
    merge_rows([](float v) { return lane_xor32(v); });
    if (r == 0) {
#pragma unroll
        for (int t = 0; t < TQ; ++t) {
#pragma unroll
            for (int e = 0; e < EPL; ++e) so[wv][t][c * EPL + e] = o[t][e];
            if (c == 0) { sml[wv][t][0] = m[t]; sml[wv][t][1] = l[t]; }
        }
    }
    DSTAMP(4);
    __syncthreads();
    DSTAMP(5);
    // partials: o [P][nsplit][G][HD], then ml [P][nsplit][G][2]
    float* wo = ws + ((long)pr * nsplit + split) * G * HD;
    float* wml = ws + (long)gridDim.y * nsplit * G * HD + ((long)pr * nsplit + split) * G * 2;
    for (int i = tid; i < G * HD; i += 256) {
        const int t = i / HD, cc = i - t * HD;
        float mn = sml[0][t][0];
#pragma unroll
        for (int w = 1; w < 4; ++w) mn = fmaxf(mn, sml[w][t][0]);
        float acc = 0.f, ll = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float a = __expf(sml[w][t][0] - mn);
            acc += so[w][t][cc] * a;
            ll += sml[w][t][1] * a;
        }
        wo[i] = acc;
        if (cc == 0) { wml[t * 2] = mn; wml[t * 2 + 1] = ll; }
    }
#ifdef ULLSAM_STAMP_DECODE
    DSTAMP(6);
    if (dbg && lane == 0) {
        unsigned long long* d = dbg + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wv) * 8;
        for (int i = 0; i < 7; ++i) d[i] = st[i];
        d[7] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}
#undef DSTAMP

__global__ __launch_bounds__(128) void decode_attn_merge_kernel(const float* __restrict__ ws, bf16* __restrict__ out, int P, int G, int nsplit) {
    constexpr int HD = 128, NS = 8;                      // splits per trip: their loads are independent and issued together
    const int pt = blockIdx.x, pr = pt / G, t = pt - pr * G, cc = threadIdx.x;
    const float* wo = ws + (long)pr * nsplit * G * HD;
    const float* wml = ws + (long)P * nsplit * G * HD + (long)pr * nsplit * G * 2;
    float mn = -1e30f;
    for (int s0 = 0; s0 < nsplit; s0 += NS) {
        float mv[NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) mv[j] = s0 + j < nsplit ? wml[((long)(s0 + j) * G + t) * 2] : -1e30f;
#pragma unroll
        for (int j = 0; j < NS; ++j) mn = fmaxf(mn, mv[j]);
    }
    float acc = 0.f, ll = 0.f;
    for (int s0 = 0; s0 < nsplit; s0 += NS) {
        float mv[NS], lv[NS], ov[NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const bool on = s0 + j < nsplit;
            const long s2 = on ? s0 + j : 0;
            mv[j] = on ? wml[(s2 * G + t) * 2] : -1e30f;
            lv[j] = on ? wml[(s2 * G + t) * 2 + 1] : 0.f;
            ov[j] = on ? wo[(s2 * G + t) * HD + cc] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < NS; ++j) {                   // (splits in order, as before: the sums are the same)
            if (s0 + j >= nsplit) break;
            const float a = __expf(mv[j] - mn);
            acc += ov[j] * a;
            ll += lv[j] * a;
        }
    }
    out[((long)pr * G + t) * HD + cc] = (bf16)(acc / ll);
}

// q bf16 [B, H*128]; kc / vc bf16 caches [B, KVH, cap, 128]; key_mask int32 [B, Sk] or NULL; out bf16 [B, H*128];
// workspace f32 [B*KVH*nsplit*G*130], G = H / KVH <= 8.
extern "C" int ullsam_decode_attention(const void* q, const void* kc, const void* vc, const int* key_mask, void* out, int B, int H,
                                       int KVH, int hd, int Sk, int cap, float scale, float* workspace, int nsplit, void* stream) {
    ULLSAM_CHECK(hd == 128 && KVH > 0 && H % KVH == 0 && H / KVH <= 8, "decode_attention: hd=%d H=%d KVH=%d (hd 128, group <= 8)", hd, H, KVH);
    ULLSAM_CHECK(Sk >= 1 && Sk <= cap && nsplit >= 1, "decode_attention: Sk=%d cap=%d nsplit=%d", Sk, cap, nsplit);
    const int G = H / KVH, P = B * KVH;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid(nsplit, P);
    const long kv_bs = (long)KVH * cap * hd, kv_hs = (long)cap * hd;
#ifdef ULLSAM_STAMP_DECODE
#define DDBG , g_attn_dbg
#else
#define DDBG
#endif
    if (G <= 4) decode_attn_partial_kernel<4><<<grid, 256, 0, s>>>((const bf16*)q, (const bf16*)kc, (const bf16*)vc, key_mask, workspace, G, KVH, Sk, kv_bs, kv_hs, scale, nsplit DDBG);
    else decode_attn_partial_kernel<8><<<grid, 256, 0, s>>>((const bf16*)q, (const bf16*)kc, (const bf16*)vc, key_mask, workspace, G, KVH, Sk, kv_bs, kv_hs, scale, nsplit DDBG);
#undef DDBG
    ULLSAM_LAUNCH_CHECK();
    decode_attn_merge_kernel<<<P * G, 128, 0, s>>>(workspace, (bf16*)out, P, G, nsplit);
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
