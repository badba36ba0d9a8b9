// Token side of SAM's two-way mask decoder (modeling/transformer.py:153-242, mask_decoder.py:112-149) as a few fused launches (bf16 models, embedding 256,
// 8 heads, <= 16 tokens per prompt).  Before (round 4): ~22 launches per block and call -- q / k / v / out projections, attention, three LayerNorms, the MLP's
// two linears, add-casts --, each a few microseconds on a few CUs: an automatic-mask-generation tile was 8087 dispatches, 3300 of them these.
//
//   ullsam_dec_tok_attn  : q_in = queries (+ pe); self attention (8 heads x 32: q, k from q_in, v from queries; softmax(q k^T / sqrt(32)) v); out projection
//                          (+ queries, except in layer 0: transformer.py:157-158); norm1; then the q projection of the token -> image attention of (y + pe).
//                          mode 1: only that last projection (the final token -> image attention, transformer.py:99-104).
//   ullsam_dec_tok_mlp   : token -> image output projection + residual, norm2 (norm_final_attn for the last attention: then nothing else), the MLP (256 -> 2048,
//                          ReLU, -> 256) + residual, norm3, and the k (of y + pe) / v (of y) projections the image -> token attention needs.
//   ullsam_dec_heads     : the four hypernetwork MLPs on the mask tokens and the IoU head on the IoU token (mask_decoder.py:141-149,154-176): 15 linears, one launch.
//
// One workgroup (8 waves) per prompt.  A linear is Y^T = W X^T on v_mfma_f32_16x16x32_bf16: W's rows (output features, packed in fragment order) are the A operand, read straight from global
// memory (L2-resident: every prompt's workgroup streams the same 0.6 - 2.2 MB), the prompt's <= 16 token rows the B operand from LDS -- as TWO bf16 terms (x = hi + lo,
// ~17 bits: two MFMAs per weight fragment), so that the token side keeps the accuracy the fp32 launches had (one term, autocast's rounding, cost 0.003 of mask IoU
// on the hard tiles of the full-depth fixture); a lane ends up with 4 consecutive features of one token.  fp32 accumulation, fp32 bias / residual / LayerNorm / softmax.  Weight fragments run through a double-buffered chunk of 8 per wave (64 KB in flight per CU covers the L2 latency at 8 waves).
// A record's outputs do not depend on what it is batched with (one workgroup per prompt, fixed summation order).
#include "common.h"

namespace {

constexpr int C = 256, CP = C + 8;      // embedding; LDS row pitch of a bf16 activation row (528 B: 16-lane groups of a ds_read_b128 cover all 64 banks)
constexpr int NWV = 8;                  // waves per workgroup

// Y^T tile loop: ntiles row tiles (16 output features each) of W [ntiles * 16, K] over the NWV waves; x = LDS bf16 [16][K + 8]; epi(tile, acc): acc[i] = y[token lane & 15][feature 16 tile + 4 (lane >> 4) + i]
// W is given in FRAGMENT order (packed once per weight version by the host, ops.pack_mfma_rows): [row tile][k-step of 32][lane][8 elements], i.e. the 16 bytes lane l of the
// A operand needs for (tile, k-step) sit at ((tile * KST + ks) * 64 + l) * 16 -- a wave's load is ONE contiguous KiB (with W as stored, [out, in], it was sixteen 64-byte
// pieces of sixteen rows: the kernels ran 3 - 4x over their weight-traffic estimate).  KST = k-steps per packed row tile (in / 32); ks0 = first k-step of a K-slice.
// D = chunks of 8 fragments (8 KiB per wave) kept in flight ahead of the MFMAs: a chunk's 16 MFMAs take ~0.1 us, an L2 round trip ~1 us, so with the two buffers of
// the first version (D = 2) a wave moved 8 KiB per round trip and the MLP's 2 MB of weights took ~60 of the kernel's 85 us; the MLP runs D = 4.
template <int K, bool LO, typename Epi, int KST = K / 32, int D = 2>
__device__ __forceinline__ void lin_tiles(const bf16* __restrict__ W, int ntiles, const bf16* x, const bf16* xlo, int wave, int lane, Epi epi, int ks0 = 0) {
    constexpr int KS = K / 32, CH = KS < 8 ? KS : 8, NCH = KS / CH;
    static_assert(KS % CH == 0, "K must be a multiple of 256 (or < 256 and a multiple of 32)");
    const int m = lane & 15, g = lane >> 4;
    const bf16* xr = x + m * (K + 8) + 8 * g;
    const bf16* xl = LO ? xlo + m * (K + 8) + 8 * g : nullptr;
    const int nt = (ntiles - wave + NWV - 1) / NWV;        // this wave's tiles: wave, wave + NWV, ...
    if (nt <= 0) return;
    const int total = nt * NCH;
    Frag<bf16> a[D][CH];
    auto fetch = [&](Frag<bf16> (&buf)[CH], int s) __attribute__((always_inline)) {
        const int tile = wave + NWV * (s / NCH), c0 = (s % NCH) * CH;
        const bf16* wr = W + ((size_t)(tile * KST + ks0 + c0) * 64 + lane) * 8;
#pragma unroll
        for (int j = 0; j < CH; ++j) buf[j] = load_frag(wr + (size_t)j * 512);
    };
    auto run = [&](const Frag<bf16> (&buf)[CH], int s, f32x4& acc) __attribute__((always_inline)) {
        const int c0 = (s % NCH) * CH;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            mma16(buf[j], load_frag(xr + 32 * (c0 + j)), acc);
            if (LO) mma16(buf[j], load_frag(xl + 32 * (c0 + j)), acc);      // the activation's second bf16 term (x = hi + lo to ~2^-17): same weight fragment
        }
        if (s % NCH == NCH - 1) { epi(wave + NWV * (s / NCH), acc); acc = f32x4{0.f, 0.f, 0.f, 0.f}; }
    };
#pragma unroll
    for (int i = 0; i < D - 1; ++i)
        if (i < total) fetch(a[i], i);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < total; s += D) {                   // D steps per trip: the buffers keep static indices
#pragma unroll
        for (int j = 0; j < D; ++j) {
            if (s + j < total) {
                if (s + j + D - 1 < total) fetch(a[(j + D - 1) % D], s + j + D - 1);
                run(a[j], s + j, acc);
            }
        }
    }
}

// LayerNorm over the 256 features of each of the T token rows of y (LDS fp32 [16][C]); wave w takes tokens w, w + 8.  out(token, feature0, float4 of 4 normalised features)
template <typename Out>
__device__ __forceinline__ void ln_rows(const float* y, int T, const float* __restrict__ w, const float* __restrict__ b, float eps, int wave, int lane, Out out) {
    for (int t = wave; t < T; t += NWV) {
        const float4 v = *reinterpret_cast<const float4*>(y + t * C + 4 * lane);
        const float mean = wave_sum(v.x + v.y + v.z + v.w) * (1.0f / C);
        const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
        const float rstd = rsqrtf(wave_sum(dx * dx + dy * dy + dz * dz + dw * dw) * (1.0f / C) + eps);
        const float4 ww = w ? *reinterpret_cast<const float4*>(w + 4 * lane) : make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 bb = b ? *reinterpret_cast<const float4*>(b + 4 * lane) : make_float4(0.f, 0.f, 0.f, 0.f);
        out(t, 4 * lane, make_float4(dx * rstd * ww.x + bb.x, dy * rstd * ww.y + bb.y, dz * rstd * ww.z + bb.z, dw * rstd * ww.w + bb.w));
    }
}

__device__ __forceinline__ void st_bf16x4(bf16* p, float a, float b, float c, float d) {
    bf16x4_t v;
    v[0] = (bf16)a; v[1] = (bf16)b; v[2] = (bf16)c; v[3] = (bf16)d;
    *reinterpret_cast<bf16x4_t*>(p) = v;
}
// x = hi + lo, both bf16 (lo = the rounding error of hi, itself rounded): the pair carries ~17 bits of x, and W x = W hi + W lo costs one more MFMA on the same weight fragment.
// With ONE bf16 term per activation (the first version) the decoder's tokens moved by 3e-3 of their scale and the full-depth bf16 mask IoU fell from 0.9898 to 0.9869 on one tile.
__device__ __forceinline__ void st_split4(bf16* hi, bf16* lo, float a, float b, float c, float d) {
    bf16x4_t h, l;
    h[0] = (bf16)a; h[1] = (bf16)b; h[2] = (bf16)c; h[3] = (bf16)d;
    l[0] = (bf16)(a - (float)h[0]); l[1] = (bf16)(b - (float)h[1]); l[2] = (bf16)(c - (float)h[2]); l[3] = (bf16)(d - (float)h[3]);
    *reinterpret_cast<bf16x4_t*>(hi) = h;
    *reinterpret_cast<bf16x4_t*>(lo) = l;
}

struct TokAttnArgs {
    const float* queries; const float* qpe; float* queries_out; float* q_t2i;       // [P*T, 256] x3, [P*T, 128]
    const bf16 *Wq, *Wk, *Wv, *Wo; const float *bq, *bk, *bv, *bo;                 // self attention, [256, 256] each
    const float *ln_w, *ln_b; float eps;                                           // norm1
    const bf16* Wq2; const float* bq2;                                             // token -> image q projection [128, 256]
    int P, T, skip_pe, mode;                                                       // mode 1: q projection of (queries + pe) only
};

__global__ __launch_bounds__(64 * NWV) void dec_tok_attn_kernel(TokAttnArgs p) {
    __shared__ __attribute__((aligned(16))) bf16 xin[16 * CP], xin_lo[16 * CP];     // q_in = queries (+ pe), later (y + pe); hi and lo terms
    __shared__ __attribute__((aligned(16))) bf16 xv[16 * CP], xv_lo[16 * CP];       // queries (the v input), later the attention output
    __shared__ __attribute__((aligned(16))) float qf[16 * C], kf[16 * C], vf[16 * C];   // q / k / v, then (qf) the block's pre-norm output
    __shared__ float sc[8 * 16 * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, prompt = blockIdx.x, T = p.T;
    const float* q0 = p.queries + (size_t)prompt * T * C;
    const float* pe = p.qpe + (size_t)prompt * T * C;
    for (int e = tid; e < 16 * C / 4; e += 64 * NWV) {
        const int t = e / (C / 4), c = 4 * (e % (C / 4));
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f), qp = q;
        if (t < T) {
            q = *reinterpret_cast<const float4*>(q0 + t * C + c);
            const float4 e4 = *reinterpret_cast<const float4*>(pe + t * C + c);
            qp = (p.skip_pe && p.mode == 0) ? q : make_float4(q.x + e4.x, q.y + e4.y, q.z + e4.z, q.w + e4.w);
        }
        st_split4(xin + t * CP + c, xin_lo + t * CP + c, qp.x, qp.y, qp.z, qp.w);
        st_split4(xv + t * CP + c, xv_lo + t * CP + c, q.x, q.y, q.z, q.w);
    }
    __syncthreads();
    const int m = lane & 15, g = lane >> 4;
    if (p.mode == 0) {
        auto to = [&](float* dst, const float* bias) {
            return [=](int tile, const f32x4& acc) {
                const int f = 16 * tile + 4 * g;
                *reinterpret_cast<float4*>(dst + m * C + f) = make_float4(acc[0] + (bias ? bias[f] : 0.f), acc[1] + (bias ? bias[f + 1] : 0.f), acc[2] + (bias ? bias[f + 2] : 0.f),
                                                                         acc[3] + (bias ? bias[f + 3] : 0.f));
            };
        };
        lin_tiles<C, true>(p.Wq, C / 16, xin, xin_lo, wave, lane, to(qf, p.bq));
        lin_tiles<C, true>(p.Wk, C / 16, xin, xin_lo, wave, lane, to(kf, p.bk));
        lin_tiles<C, true>(p.Wv, C / 16, xv, xv_lo, wave, lane, to(vf, p.bv));
        __syncthreads();
        // scores: (head h, query i, key j), 32-long dot products; scale AFTER the product (transformer.py:233-235)
        for (int e = tid; e < 8 * T * T; e += 64 * NWV) {
            const int h = e / (T * T), i = (e / T) % T, j = e % T;
            const float* a = qf + i * C + 32 * h;
            const float* b = kf + j * C + 32 * h;
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < 32; ++d) s = fmaf(a[d], b[d], s);
            sc[(h * 16 + i) * 16 + j] = s * 0.17677669529663687f;          // 1 / sqrt(32)
        }
        __syncthreads();
        for (int e = tid; e < 8 * T; e += 64 * NWV) {                      // softmax rows
            float* r = sc + (e / T * 16 + e % T) * 16;
            float mx = -INFINITY;
            for (int j = 0; j < T; ++j) mx = fmaxf(mx, r[j]);
            float sum = 0.f;
            for (int j = 0; j < T; ++j) { r[j] = __expf(r[j] - mx); sum += r[j]; }
            const float inv = 1.0f / sum;
            for (int j = 0; j < T; ++j) r[j] *= inv;
        }
        __syncthreads();
        for (int e = tid; e < 16 * C; e += 64 * NWV) {                     // a[i][h * 32 + d] = sum_j p[h][i][j] v[j][h * 32 + d]  -> bf16 (the out projection's input)
            const int i = e / C, f = e % C;
            float s = 0.f;
            if (i < T) {
                const float* r = sc + ((f >> 5) * 16 + i) * 16;
                for (int j = 0; j < T; ++j) s = fmaf(r[j], vf[j * C + f], s);
            }
            const bf16 hi = (bf16)s;
            xv[i * CP + f] = hi;
            xv_lo[i * CP + f] = (bf16)(s - (float)hi);
        }
        __syncthreads();
        lin_tiles<C, true>(p.Wo, C / 16, xv, xv_lo, wave, lane, [=](int tile, const f32x4& acc) {
            const int f = 16 * tile + 4 * g;
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!p.skip_pe && m < T) r = *reinterpret_cast<const float4*>(q0 + m * C + f);     // layer 0 has no residual here (transformer.py:157-158)
            *reinterpret_cast<float4*>(qf + m * C + f) = make_float4(acc[0] + (p.bo ? p.bo[f] : 0.f) + r.x, acc[1] + (p.bo ? p.bo[f + 1] : 0.f) + r.y,
                                                                     acc[2] + (p.bo ? p.bo[f + 2] : 0.f) + r.z, acc[3] + (p.bo ? p.bo[f + 3] : 0.f) + r.w);
        });
        __syncthreads();
        float* qo = p.queries_out + (size_t)prompt * T * C;
        ln_rows(qf, T, p.ln_w, p.ln_b, p.eps, wave, lane, [=](int t, int f, float4 y) {
            *reinterpret_cast<float4*>(qo + t * C + f) = y;
            const float4 e4 = *reinterpret_cast<const float4*>(pe + t * C + f);
            st_split4(xin + t * CP + f, xin_lo + t * CP + f, y.x + e4.x, y.y + e4.y, y.z + e4.z, y.w + e4.w);
        });
        __syncthreads();
    }
    float* q2 = p.q_t2i + (size_t)prompt * T * 128;
    lin_tiles<C, true>(p.Wq2, 128 / 16, xin, xin_lo, wave, lane, [=](int tile, const f32x4& acc) {
        const int f = 16 * tile + 4 * g;
        if (m < T)
            *reinterpret_cast<float4*>(q2 + m * 128 + f) = make_float4(acc[0] + (p.bq2 ? p.bq2[f] : 0.f), acc[1] + (p.bq2 ? p.bq2[f + 1] : 0.f), acc[2] + (p.bq2 ? p.bq2[f + 2] : 0.f),
                                                                       acc[3] + (p.bq2 ? p.bq2[f + 3] : 0.f));
    });
}

struct TokMlpArgs {
    const float* queries; const float* attn; const float* qpe; float* queries_out; float* k_out; float* v_out;   // [P*T, 256], [P*T, 128], [P*T, 256], out [P*T, 256], [P*T, 128] x2
    const bf16* Wo; const float* bo;                         // token -> image out projection [256, 128]
    const float *ln2_w, *ln2_b; float eps2;
    const bf16 *W1, *W2; const float *b1, *b2;               // MLP [2048, 256], [256, 2048]
    const float *ln3_w, *ln3_b; float eps3;
    const bf16 *Wk, *Wv; const float *bk, *bv;               // image -> token k / v projections [128, 256]
    int P, T, do_mlp;                                        // do_mlp 0: out projection + residual + norm only (the final attention)
};

constexpr int HID = 2048, HH = 1024, HHP = HH + 8;   // MLP width; hidden units per pass; LDS pitch of a hidden row
__global__ __launch_bounds__(64 * NWV) void dec_tok_mlp_kernel(TokMlpArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* hid = reinterpret_cast<bf16*>(smem);                                   // [16][HHP] hidden units of one half, hi term   (do_mlp only)
    bf16* hid_lo = hid + 16 * HHP;                                               //           lo term
    bf16* xa = reinterpret_cast<bf16*>(smem + (p.do_mlp ? 2 * 16 * HHP * 2 : 0));   // [16][CP]: the attention output (128 wide, pitch 136), then norm2's output; hi term
    bf16* xa_lo = xa + 16 * CP;                                                  //           its lo term
    bf16* xb = xa_lo + 16 * CP;                                                  // [16][CP]: (y + pe) for the k projection, hi / lo
    bf16* xb_lo = xb + 16 * CP;
    float* yf = reinterpret_cast<float*>(xb_lo + 16 * CP);                       // [16][C] pre-norm sums
    float* q2 = yf + 16 * C;                                                     // [16][C] norm2's output in fp32 (the MLP's residual)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, prompt = blockIdx.x, T = p.T;
    const int m = lane & 15, g = lane >> 4;
    const float* q0 = p.queries + (size_t)prompt * T * C;
    const float* pe = p.qpe + (size_t)prompt * T * C;
    const float* at = p.attn + (size_t)prompt * T * 128;
    for (int e = tid; e < 16 * 128 / 4; e += 64 * NWV) {
        const int t = e / 32, c = 4 * (e % 32);
        const float4 v = t < T ? *reinterpret_cast<const float4*>(at + t * 128 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        st_split4(xa + t * (128 + 8) + c, xa_lo + t * (128 + 8) + c, v.x, v.y, v.z, v.w);
    }
    __syncthreads();
    lin_tiles<128, true>(p.Wo, C / 16, xa, xa_lo, wave, lane, [=](int tile, const f32x4& acc) {
        const int f = 16 * tile + 4 * g;
        const float4 r = m < T ? *reinterpret_cast<const float4*>(q0 + m * C + f) : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(yf + m * C + f) = make_float4(acc[0] + (p.bo ? p.bo[f] : 0.f) + r.x, acc[1] + (p.bo ? p.bo[f + 1] : 0.f) + r.y,
                                                                 acc[2] + (p.bo ? p.bo[f + 2] : 0.f) + r.z, acc[3] + (p.bo ? p.bo[f + 3] : 0.f) + r.w);
    });
    __syncthreads();
    float* qo = p.queries_out + (size_t)prompt * T * C;
    if (!p.do_mlp) {
        ln_rows(yf, T, p.ln2_w, p.ln2_b, p.eps2, wave, lane, [=](int t, int f, float4 y) { *reinterpret_cast<float4*>(qo + t * C + f) = y; });
        return;
    }
    for (int e = tid; e < 2 * 16 * CP / 8; e += 64 * NWV) reinterpret_cast<uint4*>(xa)[e] = make_uint4(0u, 0u, 0u, 0u);   // rows >= T of the next B operand (hi and lo): zeros (finite)
    __syncthreads();
    ln_rows(yf, T, p.ln2_w, p.ln2_b, p.eps2, wave, lane, [=](int t, int f, float4 y) {
        *reinterpret_cast<float4*>(q2 + t * C + f) = y;
        st_split4(xa + t * CP + f, xa_lo + t * CP + f, y.x, y.y, y.z, y.w);
    });
    __syncthreads();
    // the MLP in two halves of the hidden dimension: 1024 hidden units at a time as two bf16 terms (both halves' lin2 sums meet in yf)
    for (int half = 0; half < 2; ++half) {
        auto epi1 = [=](int tile, const f32x4& acc) {      // lin1 + ReLU -> hidden (hi, lo)
            const int f = 16 * tile + 4 * g;
            const float* bb = p.b1 ? p.b1 + half * HH + f : nullptr;
            st_split4(hid + m * HHP + f, hid_lo + m * HHP + f, fmaxf(acc[0] + (bb ? bb[0] : 0.f), 0.f), fmaxf(acc[1] + (bb ? bb[1] : 0.f), 0.f), fmaxf(acc[2] + (bb ? bb[2] : 0.f), 0.f),
                      fmaxf(acc[3] + (bb ? bb[3] : 0.f), 0.f));
        };
        lin_tiles<C, true, decltype(epi1), C / 32, 4>(p.W1 + (size_t)half * (HH / 16) * (C / 32) * 512, HH / 16, xa, xa_lo, wave, lane, epi1);
        __syncthreads();
        auto epi2 = [=](int tile, const f32x4& acc) {                                        // lin2 (+ bias + residual = norm2's output, with the first half)
            const int f = 16 * tile + 4 * g;
            float4 r = half ? *reinterpret_cast<const float4*>(yf + m * C + f) : *reinterpret_cast<const float4*>(q2 + m * C + f);
            if (!half && p.b2) { r.x += p.b2[f]; r.y += p.b2[f + 1]; r.z += p.b2[f + 2]; r.w += p.b2[f + 3]; }
            *reinterpret_cast<float4*>(yf + m * C + f) = make_float4(acc[0] + r.x, acc[1] + r.y, acc[2] + r.z, acc[3] + r.w);
        };
        lin_tiles<HH, true, decltype(epi2), HID / 32, 4>(p.W2, C / 16, hid, hid_lo, wave, lane, epi2, half * (HH / 32));
        __syncthreads();
    }
    for (int e = tid; e < 2 * 16 * CP / 8; e += 64 * NWV) reinterpret_cast<uint4*>(xb)[e] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    ln_rows(yf, T, p.ln3_w, p.ln3_b, p.eps3, wave, lane, [=](int t, int f, float4 y) {
        *reinterpret_cast<float4*>(qo + t * C + f) = y;
        const float4 e4 = *reinterpret_cast<const float4*>(pe + t * C + f);
        st_split4(xa + t * CP + f, xa_lo + t * CP + f, y.x, y.y, y.z, y.w);                                   // v = v_proj(queries)
        st_split4(xb + t * CP + f, xb_lo + t * CP + f, y.x + e4.x, y.y + e4.y, y.z + e4.z, y.w + e4.w);       // k = k_proj(queries + pe)   (transformer.py:176-178)
    });
    __syncthreads();
    float* ko = p.k_out + (size_t)prompt * T * 128;
    float* vo = p.v_out + (size_t)prompt * T * 128;
    auto to = [&](float* dst, const float* bias) {
        return [=](int tile, const f32x4& acc) {
            const int f = 16 * tile + 4 * g;
            if (m < T)
                *reinterpret_cast<float4*>(dst + m * 128 + f) = make_float4(acc[0] + (bias ? bias[f] : 0.f), acc[1] + (bias ? bias[f + 1] : 0.f), acc[2] + (bias ? bias[f + 2] : 0.f),
                                                                            acc[3] + (bias ? bias[f + 3] : 0.f));
        };
    };
    lin_tiles<C, true>(p.Wk, 128 / 16, xb, xb_lo, wave, lane, to(ko, p.bk));
    lin_tiles<C, true>(p.Wv, 128 / 16, xa, xa_lo, wave, lane, to(vo, p.bv));
}

// Heads: chain c of 5 = the four hypernetwork MLPs (on mask token 1 + c of every prompt, -> 32 values) and the IoU head (chain 4, on token 0, -> NM values); three linears with
// ReLU between them (mask_decoder.py:154-176, sigmoid_output False).  A workgroup = one chain x 16 prompts (the MFMA's 16 columns are prompts here).
struct HeadArgs {
    const float* hs;                     // [P, T, 256] decoder output tokens
    const bf16* W[5][3]; const float* b[5][3];
    float* hyper; float* iou;            // [P, nm, 32] (hypernetwork chains m0 .. m0 + nm - 1), [P, n_iou]
    int P, T, n_iou, m0, nm;
};
__global__ __launch_bounds__(64 * NWV) void dec_heads_kernel(HeadArgs p) {
    __shared__ __attribute__((aligned(16))) bf16 x0[16 * CP], x0_lo[16 * CP], x1[16 * CP], x1_lo[16 * CP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, chain = blockIdx.x % 5, pb = blockIdx.x / 5;
    const int m = lane & 15, g = lane >> 4, tok = chain < 4 ? 1 + chain : 0;
    if (chain < 4 && (chain < p.m0 || chain >= p.m0 + p.nm)) return;      // a mask the caller does not ask for (multimask output: masks 1 .. 3)
    for (int e = tid; e < 16 * C / 4; e += 64 * NWV) {
        const int r = e / (C / 4), c = 4 * (e % (C / 4)), prompt = pb * 16 + r;
        const float4 v = prompt < p.P ? *reinterpret_cast<const float4*>(p.hs + ((size_t)prompt * p.T + tok) * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        st_split4(x0 + r * CP + c, x0_lo + r * CP + c, v.x, v.y, v.z, v.w);
    }
    __syncthreads();
    auto relu_to = [&](bf16* dst, bf16* dst_lo, const float* bias) {
        return [=](int tile, const f32x4& acc) {
            const int f = 16 * tile + 4 * g;
            st_split4(dst + m * CP + f, dst_lo + m * CP + f, fmaxf(acc[0] + (bias ? bias[f] : 0.f), 0.f), fmaxf(acc[1] + (bias ? bias[f + 1] : 0.f), 0.f), fmaxf(acc[2] + (bias ? bias[f + 2] : 0.f), 0.f),
                      fmaxf(acc[3] + (bias ? bias[f + 3] : 0.f), 0.f));
        };
    };
    lin_tiles<C, true>(p.W[chain][0], C / 16, x0, x0_lo, wave, lane, relu_to(x1, x1_lo, p.b[chain][0]));
    __syncthreads();
    lin_tiles<C, true>(p.W[chain][1], C / 16, x1, x1_lo, wave, lane, relu_to(x0, x0_lo, p.b[chain][1]));
    __syncthreads();
    const int nout = chain < 4 ? 32 : p.n_iou;
    const float* bias = p.b[chain][2];
    lin_tiles<C, true>(p.W[chain][2], (nout + 15) / 16, x0, x0_lo, wave, lane, [=](int tile, const f32x4& acc) {     // (the last weight is padded to 16 rows by the caller)
        const int prompt = pb * 16 + m;
        if (prompt >= p.P) return;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = 16 * tile + 4 * g + i;
            if (f < nout) {
                const float v = acc[i] + (bias ? bias[f] : 0.f);
                if (chain < 4) p.hyper[((size_t)prompt * p.nm + (chain - p.m0)) * 32 + f] = v;
                else p.iou[(size_t)prompt * p.n_iou + f] = v;
            }
        }
    });
}

// out[p][t] = t < n0 ? prefix[t] : rows[p][t - n0]: the decoder's token matrix [iou token, mask tokens, sparse prompt embeddings] (mask_decoder.py:119-123)
__global__ __launch_bounds__(256) void concat_token_rows_kernel(const float* __restrict__ prefix, int n0, const float* __restrict__ rows, int n1, float* __restrict__ out, int P, int Cc) {
    const long total = (long)P * (n0 + n1) * (Cc / 4);
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(e % (Cc / 4));
        const long r = e / (Cc / 4);
        const int t = (int)(r % (n0 + n1));
        const long pp = r / (n0 + n1);
        reinterpret_cast<float4*>(out)[e] = t < n0 ? reinterpret_cast<const float4*>(prefix)[(long)t * (Cc / 4) + c4]
                                                   : reinterpret_cast<const float4*>(rows)[(pp * n1 + (t - n0)) * (Cc / 4) + c4];
    }
}

}  // namespace

#define AL16(x) ((((uintptr_t)(x)) & 15) == 0)

extern "C" int ullsam_dec_tok_attn(const float* queries, const float* qpe, float* queries_out, float* q_t2i, const void* Wq, const float* bq, const void* Wk, const float* bk,
                                   const void* Wv, const float* bv, const void* Wo, const float* bo, const float* ln_w, const float* ln_b, float eps, const void* Wq2,
                                   const float* bq2, int P, int T, int skip_pe, int mode, void* stream) {
    ULLSAM_CHECK(P > 0 && T >= 1 && T <= 16, "dec_tok_attn: P=%d T=%d (1..16)", P, T);
    ULLSAM_CHECK(queries && qpe && q_t2i && Wq2 && (mode == 1 || (queries_out && Wq && Wk && Wv && Wo)), "dec_tok_attn: null operand");
    ULLSAM_CHECK(AL16(queries) && AL16(qpe) && AL16(queries_out) && AL16(q_t2i) && AL16(Wq) && AL16(Wk) && AL16(Wv) && AL16(Wo) && AL16(Wq2) && AL16(bq) && AL16(bo) && AL16(bk) && AL16(bv) && AL16(bq2) && AL16(ln_w) && AL16(ln_b), "dec_tok_attn: 16-byte aligned operands needed");
    TokAttnArgs a{queries, qpe, queries_out, q_t2i, (const bf16*)Wq, (const bf16*)Wk, (const bf16*)Wv, (const bf16*)Wo, bq, bk, bv, bo, ln_w, ln_b, eps, (const bf16*)Wq2, bq2, P, T, skip_pe, mode};
    dec_tok_attn_kernel<<<dim3(P), 64 * NWV, 0, reinterpret_cast<hipStream_t>(stream)>>>(a);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

extern "C" int ullsam_dec_tok_mlp(const float* queries, const float* attn, const float* qpe, float* queries_out, float* k_out, float* v_out, const void* Wo, const float* bo,
                                  const float* ln2_w, const float* ln2_b, float eps2, const void* W1, const float* b1, const void* W2, const float* b2, const float* ln3_w,
                                  const float* ln3_b, float eps3, const void* Wk, const float* bk, const void* Wv, const float* bv, int P, int T, int do_mlp, void* stream) {
    ULLSAM_CHECK(P > 0 && T >= 1 && T <= 16, "dec_tok_mlp: P=%d T=%d (1..16)", P, T);
    ULLSAM_CHECK(queries && attn && queries_out && Wo && (!do_mlp || (qpe && k_out && v_out && W1 && W2 && Wk && Wv)), "dec_tok_mlp: null operand");
    ULLSAM_CHECK(AL16(queries) && AL16(attn) && AL16(qpe) && AL16(queries_out) && AL16(k_out) && AL16(v_out) && AL16(Wo) && AL16(W1) && AL16(W2) && AL16(Wk) && AL16(Wv) && AL16(bo) && AL16(ln2_w) && AL16(ln2_b) && AL16(b1) && AL16(b2) && AL16(ln3_w) && AL16(ln3_b) && AL16(bk) && AL16(bv),
                 "dec_tok_mlp: 16-byte aligned operands needed");
    TokMlpArgs a{queries, attn, qpe, queries_out, k_out, v_out, (const bf16*)Wo, bo, ln2_w, ln2_b, eps2, (const bf16*)W1, (const bf16*)W2, b1, b2, ln3_w, ln3_b, eps3,
                 (const bf16*)Wk, (const bf16*)Wv, bk, bv, P, T, do_mlp};
    const int lds = (do_mlp ? 2 * 16 * HHP * 2 : 0) + 4 * 16 * CP * 2 + 2 * 16 * C * 4;
    static PerDeviceOnce attr;
    if (attr.first() && hipFuncSetAttribute(reinterpret_cast<const void*>(dec_tok_mlp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 16 * HHP * 2 + 4 * 16 * CP * 2 + 2 * 16 * C * 4) != hipSuccess)
        ULLSAM_CHECK(false, "dec_tok_mlp: hipFuncSetAttribute failed");
    dec_tok_mlp_kernel<<<dim3(P), 64 * NWV, lds, reinterpret_cast<hipStream_t>(stream)>>>(a);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// w[15] / b[15]: chain-major (hypernetwork MLP 0 .. 3, then the IoU head), three layers each; the LAST layer's weight of every chain has >= 16 * ceil(n_out / 16) rows
// (zero padded by the caller: 32 rows for the hypernetworks, 16 for the IoU head).
extern "C" int ullsam_dec_heads(const float* hs, const void* const* w, const float* const* b, float* hyper, float* iou, int P, int T, int n_iou, int m0, int nm, void* stream) {
    ULLSAM_CHECK(P > 0 && T >= 5 && n_iou >= 1 && n_iou <= 16 && hs && w && b && hyper && iou && m0 >= 0 && nm >= 1 && m0 + nm <= 4, "dec_heads: P=%d T=%d n_iou=%d masks %d..+%d", P, T, n_iou, m0, nm);
    HeadArgs a;
    a.hs = hs; a.hyper = hyper; a.iou = iou; a.P = P; a.T = T; a.n_iou = n_iou; a.m0 = m0; a.nm = nm;
    for (int c = 0; c < 5; ++c)
        for (int l = 0; l < 3; ++l) {
            a.W[c][l] = (const bf16*)w[c * 3 + l];
            a.b[c][l] = b[c * 3 + l];
            ULLSAM_CHECK(a.W[c][l] && AL16(a.W[c][l]), "dec_heads: weight %d.%d null / unaligned", c, l);
        }
    dec_heads_kernel<<<dim3(5 * ((P + 15) / 16)), 64 * NWV, 0, reinterpret_cast<hipStream_t>(stream)>>>(a);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

extern "C" int ullsam_concat_token_rows(const float* prefix, int n0, const float* rows, int n1, float* out, int P, int C, void* stream) {
    ULLSAM_CHECK(P > 0 && n0 >= 0 && n1 >= 0 && n0 + n1 > 0 && C > 0 && C % 4 == 0 && out && (n0 == 0 || prefix) && (n1 == 0 || rows), "concat_token_rows: P=%d n0=%d n1=%d C=%d", P, n0, n1, C);
    ULLSAM_CHECK(AL16(prefix) && AL16(rows) && AL16(out), "concat_token_rows: 16-byte aligned operands needed");
    const long total = (long)P * (n0 + n1) * (C / 4);
    const unsigned blocks = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    concat_token_rows_kernel<<<blocks, 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(prefix, n0, rows, n1, out, P, C);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}
