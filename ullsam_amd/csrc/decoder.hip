// Prompt-encoder and mask-decoder kernels (token side is fp32 throughout; the image side uses the MFMA GEMM).
#include "common.h"

// ---- small dense layer: y[M,N] = act(x[M,K] . W[N,K]^T + b) (+ res) -- one wave per output element ----------------
// token-side Linear layers of transformer.py:220-227,171-173 and the hypernetwork / IoU MLPs of mask_decoder.py:139-147
__global__ __launch_bounds__(256) void small_linear_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ W,
                                                           const float* __restrict__ b, const float* __restrict__ res, long ldr,
                                                           float* __restrict__ y, long ldy, int M, int N, int K, int act) {
    const int lane = threadIdx.x & 63;
    const long o = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= (long)M * N) return;
    const int m = (int)(o / N), n = (int)(o % N);
    const float* xr = x + (long)m * ldx;
    const float* wr = W + (long)n * K;
    float acc = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
        const float4 a = *reinterpret_cast<const float4*>(xr + k);
        const float4 w = *reinterpret_cast<const float4*>(wr + k);
        acc += (a.x * w.x + a.y * w.y) + (a.z * w.z + a.w * w.w);
    }
    acc = wave_sum(acc);
    if (lane == 0) {
        if (b) acc += b[n];
        if (act == 2) acc = fmaxf(acc, 0.f);
        else if (act == 1) acc = gelu_erf(acc);
        if (res) acc += res[(long)m * ldr + n];
        y[(long)m * ldy + n] = acc;
    }
}

extern "C" int ullsam_small_linear(const float* x, long ldx, const float* W, const float* b, const float* res, long ldr, float* y,
                                   long ldy, int M, int N, int K, int act, void* stream) {
    ULLSAM_CHECK(K % 4 == 0 && ldx % 4 == 0, "small_linear: K and ldx must be multiples of 4");
    const long outs = (long)M * N;
    if (outs == 0) return 0;
    small_linear_kernel<<<(unsigned)((outs + 3) / 4), 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(x, ldx, W, b, res, ldr, y, ldy, M, N, K, act);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// ---- the same layer for tens to hundreds of rows (many prompts): lanes = 64 outputs, 16 rows per workgroup --------------
// WT is the transposed weight [K, N], so a wave's weight load is one coalesced row segment; the x values of a row block are
// wave-uniform (scalar loads) and feed 16 FMAs per loaded weight.  The four waves of a workgroup split K and reduce through LDS.
template <int KC>
__global__ __launch_bounds__(256) void skinny_linear_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ WT,
                                                            const float* __restrict__ b, const float* __restrict__ res, long ldr,
                                                            float* __restrict__ y, long ldy, int M, int N, int K, int act) {
    constexpr int RM = 16;
    __shared__ float red[4][RM][64];
    __shared__ __attribute__((aligned(16))) float xs[4][RM][KC + 4];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = blockIdx.x * 64 + lane, nc = min(n, N - 1);
    const int m0 = blockIdx.y * RM;
    const int kq = K >> 2, k0 = wv * kq;                 // this wave's K range (kq % KC == 0)
    float acc[RM];
#pragma unroll
    for (int i = 0; i < RM; ++i) acc[i] = 0.f;
    for (int kc = 0; kc < kq; kc += KC) {
        // stage x[m0 .. m0+15][k0+kc .. +KC) for this wave (coalesced), then every lane reads it back as broadcasts
        constexpr int C4 = KC / 4;                       // float4 per row
#pragma unroll
        for (int j = 0; j < RM * C4 / 64; ++j) {
            const int idx = j * 64 + lane, row = idx / C4, c4 = idx % C4;
            const float4 v = *reinterpret_cast<const float4*>(x + (long)min(m0 + row, M - 1) * ldx + k0 + kc + c4 * 4);
            *reinterpret_cast<float4*>(&xs[wv][row][c4 * 4]) = v;
        }
        const float* wp = WT + (long)(k0 + kc) * N + nc;
#pragma unroll
        for (int g = 0; g < KC; g += 16) {
            float w[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) w[u] = wp[(long)(g + u) * N];
#pragma unroll
            for (int u = 0; u < 16; u += 4) {
#pragma unroll
                for (int i = 0; i < RM; ++i) {
                    const float4 xv = *reinterpret_cast<const float4*>(&xs[wv][i][g + u]);
                    acc[i] += (xv.x * w[u] + xv.y * w[u + 1]) + (xv.z * w[u + 2] + xv.w * w[u + 3]);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < RM; ++i) red[wv][i][lane] = acc[i];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RM / 4; ++j) {
        const int i = wv * (RM / 4) + j, m = m0 + i;
        float v = (red[0][i][lane] + red[1][i][lane]) + (red[2][i][lane] + red[3][i][lane]);
        if (m < M && n < N) {
            if (b) v += b[n];
            if (act == 2) v = fmaxf(v, 0.f);
            else if (act == 1) v = gelu_erf(v);
            if (res) v += res[(long)m * ldr + n];
            y[(long)m * ldy + n] = v;
        }
    }
}

// ---- the same layer on the matrix pipe (exact fp32: v_mfma_f32_32x32x2_f32 is an fma chain): a 32 x 32 output tile per workgroup, its four waves
// splitting K.  Nothing goes through LDS on the way in: a lane loads x[m][k0 + 4 h .. +3] as ONE float4 (h = lane half) and the four weight rows
// WT[k0 + 4 h + e][n] (coalesced); MFMA e of the chunk then sums k = k0 + e (lower lanes) and k0 + 4 + e (upper lanes) -- the order of the k sum is free
// as long as both operands agree.  All of a wave's loads (K / 4 values per lane pair) are issued before its first MFMA: one memory latency per launch.
// The token side of the mask decoder at 64 prompts per batch (448 rows x 256 .. 2048: 68 launches per batch) ran 16.6 us per launch on the FMA kernel.
template <int KW, int NW>   // k values per wave = K / NW; NW waves (4 or 16) split K
__global__ __launch_bounds__(64 * NW) void skinny_linear_mfma_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ WT,
                                                                 const float* __restrict__ b, const float* __restrict__ res, long ldr,
                                                                 float* __restrict__ y, long ldy, int M, int N, int act) {
    constexpr int NCH = KW / 8;
    __shared__ float red[NW][16][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, l5 = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
    const float* xp = x + (long)min(m0 + l5, M - 1) * ldx + wv * KW + 4 * h;
    const float* wp = WT + (long)(wv * KW + 4 * h) * N + n0 + l5;
    float4 a[NCH];
    float w[NCH][4];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        a[c] = *reinterpret_cast<const float4*>(xp + 8 * c);
#pragma unroll
        for (int e = 0; e < 4; ++e) w[c][e] = wp[(long)(8 * c + e) * N];
    }
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c].x, w[c][0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c].y, w[c][1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c].z, w[c][2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c].w, w[c][3], acc, 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) red[wv][e][lane] = acc[e];
    __syncthreads();
    // accumulator element e of lane l: row 8 (e / 4) + 4 (l / 32) + e % 4, column l % 32; wave wv finishes elements (16 / NW) wv ...
    constexpr int EPW = 16 / NW;
#pragma unroll
    for (int j = 0; j < EPW; ++j) {
        const int e = EPW * wv + j;
        const int m = m0 + 8 * (e >> 2) + 4 * h + (e & 3), n = n0 + l5;
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < NW; q += 4) v += (red[q][e][lane] + red[q + 1][e][lane]) + (red[q + 2][e][lane] + red[q + 3][e][lane]);
        if (m < M) {
            if (b) v += b[n];
            if (act == 2) v = fmaxf(v, 0.f);
            else if (act == 1) v = gelu_erf(v);
            if (res) v += res[(long)m * ldr + n];
            y[(long)m * ldy + n] = v;
        }
    }
}
static int g_skinny_mfma = 1;
extern "C" int ullsam_set_skinny_linear_mfma(int on) { const int old = g_skinny_mfma; g_skinny_mfma = on; return old; }

// WT fp32 [K, N] (transposed nn.Linear weight); otherwise as ullsam_small_linear.
extern "C" int ullsam_skinny_linear(const float* x, long ldx, const float* WT, const float* b, const float* res, long ldr, float* y,
                                    long ldy, int M, int N, int K, int act, void* stream) {
    ULLSAM_CHECK(K % 128 == 0 && ldx % 4 == 0 && ((uintptr_t)x & 15) == 0, "skinny_linear: K %% 128, ldx %% 4, 16-byte aligned x");
    if ((long)M * N == 0) return 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // (K >= 1024: sixteen waves split K -- with four a wave's share is a chain of 256 dependent MFMAs, 179 us per launch at K = 2048)
    if (g_skinny_mfma && N % 32 == 0 && (K == 128 || K == 256 || K == 512 || K == 1024 || K == 2048)) {
        const dim3 g32((unsigned)(N / 32), (unsigned)((M + 31) / 32));
#define SKINNY_MFMA(KW, NW) skinny_linear_mfma_kernel<KW, NW><<<g32, 64 * NW, 0, s>>>(x, ldx, WT, b, res, ldr, y, ldy, M, N, act)
        if (K == 128) SKINNY_MFMA(32, 4); else if (K == 256) SKINNY_MFMA(64, 4); else if (K == 512) SKINNY_MFMA(128, 4); else if (K == 1024) SKINNY_MFMA(64, 16); else SKINNY_MFMA(128, 16);
#undef SKINNY_MFMA
        ULLSAM_LAUNCH_CHECK();
        return 0;
    }
    const dim3 grid((unsigned)((N + 63) / 64), (unsigned)((M + 15) / 16));
    if (K % 256 == 0) skinny_linear_kernel<64><<<grid, 256, 0, s>>>(x, ldx, WT, b, res, ldr, y, ldy, M, N, K, act);
    else skinny_linear_kernel<32><<<grid, 256, 0, s>>>(x, ldx, WT, b, res, ldr, y, ldy, M, N, K, act);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// ---- sparse prompt embedding (prompt_encoder.py:76-103,220-250) ---------------------------------------------------
// coords [P,Np,2] px, labels int32 [P,Np], boxes [P,4] or null.  out f32 [P, n_out, C], n_out = Np + pad + 2*has_box
// tables: G [2, C/2]; emb rows: 0 not_a_point, 1..4 point_embeddings[0..3]  ([5, C])
__global__ __launch_bounds__(128) void sparse_embed_kernel(const float* __restrict__ coords, const int* __restrict__ labels,
                                                           const float* __restrict__ boxes, const float* __restrict__ G,
                                                           const float* __restrict__ emb, float* __restrict__ out, int P, int Np,
                                                           int pad, int C, float img_w, float img_h) {
    const int n_out = Np + pad + (boxes ? 2 : 0);
    const int p = blockIdx.x / n_out, i = blockIdx.x % n_out;
    const int half = C / 2;
    float cx, cy;
    int label;  // -1 not a point, 0 neg, 1 pos, 2/3 box corners
    if (i < Np) {
        cx = coords[((long)p * Np + i) * 2] + 0.5f;
        cy = coords[((long)p * Np + i) * 2 + 1] + 0.5f;
        label = labels[(long)p * Np + i];
    } else if (i < Np + pad) {
        cx = 0.f; cy = 0.f; label = -1;  // padding point is appended AFTER the +0.5 shift (:81-88)
    } else {
        const int corner = i - Np - pad;
        cx = boxes[(long)p * 4 + 2 * corner] + 0.5f;
        cy = boxes[(long)p * 4 + 2 * corner + 1] + 0.5f;
        label = 2 + corner;
    }
    const float nx = 2.0f * (cx / img_w) - 1.0f, ny = 2.0f * (cy / img_h) - 1.0f;
    float* o = out + ((long)p * n_out + i) * C;
    for (int c = threadIdx.x; c < half; c += blockDim.x) {
        const float ang = 6.283185307179586f * (nx * G[c] + ny * G[half + c]);
        float sv = sinf(ang), cv = cosf(ang);
        if (label == -1) { sv = 0.f; cv = 0.f; }
        const int row = label + 1;  // -1 -> 0, 0 -> 1, 1 -> 2, 2 -> 3, 3 -> 4
        if (row >= 0 && row <= 4) { sv += emb[row * C + c]; cv += emb[row * C + half + c]; }
        o[c] = sv;
        o[half + c] = cv;
    }
}

extern "C" int ullsam_sparse_embed(const float* coords, const int* labels, const float* boxes, const float* G, const float* emb,
                                   float* out, int P, int Np, int pad, int C, float img_w, float img_h, void* stream) {
    const int n_out = Np + pad + (boxes ? 2 : 0);
    if (P * n_out == 0) return 0;
    sparse_embed_kernel<<<P * n_out, 128, 0, reinterpret_cast<hipStream_t>(stream)>>>(coords, labels, boxes, G, emb, out, P, Np, pad, C, img_w, img_h);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// ---- dense positional encoding grid (prompt_encoder.py:230-241), NHWC f32 [H*W, C] -------------------------------
__global__ __launch_bounds__(128) void dense_pe_kernel(const float* __restrict__ G, float* __restrict__ out, int H, int W, int C) {
    const int y = blockIdx.x / W, x = blockIdx.x % W;
    const int half = C / 2;
    const float nx = 2.0f * (((float)x + 0.5f) / (float)W) - 1.0f, ny = 2.0f * (((float)y + 0.5f) / (float)H) - 1.0f;
    float* o = out + (long)blockIdx.x * C;
    for (int c = threadIdx.x; c < half; c += blockDim.x) {
        const float ang = 6.283185307179586f * (nx * G[c] + ny * G[half + c]);
        o[c] = sinf(ang);
        o[half + c] = cosf(ang);
    }
}

extern "C" int ullsam_dense_pe(const float* G, float* out_nhwc, int H, int W, int C, void* stream) {
    dense_pe_kernel<<<H * W, 128, 0, reinterpret_cast<hipStream_t>(stream)>>>(G, out_nhwc, H, W, C);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// ---- mask-input embedding (mask_downscaling, prompt_encoder.py:54-62), fused per output pixel ----------------------
// in f32 [P,1,4H,4W] -> out NHWC f32 [P, H*W, C].  conv k2s2 (1->c1) + LN2d + GELU + conv k2s2 (c1->c2) + LN2d + GELU + 1x1 (c2->C)
__global__ __launch_bounds__(256) void mask_downscale_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W, int C,
                                                             int c1, int c2, const float* w0, const float* b0, const float* g1,
                                                             const float* be1, const float* w3, const float* b3, const float* g4,
                                                             const float* be4, const float* w6, const float* b6) {
    __shared__ float mid[4][16];  // [2x2 position][c1 <= 16]
    __shared__ float hid[64];     // c2 <= 64
    const int pix = blockIdx.x % (H * W), p = blockIdx.x / (H * W);
    const int y = pix / W, x = pix % W;
    const int IW = 4 * W;
    const float* ip = in + (long)p * 16 * H * W;
    const int t = threadIdx.x;
    if (t < 4) {  // stage 1 for sub-position t = (sy, sx)
        const int sy = t >> 1, sx = t & 1;
        float v[16];
        float mean = 0.f;
        for (int c = 0; c < c1; ++c) {
            float a = b0[c];
            for (int ky = 0; ky < 2; ++ky)
                for (int kx = 0; kx < 2; ++kx)
                    a += w0[c * 4 + ky * 2 + kx] * ip[(long)(4 * y + 2 * sy + ky) * IW + 4 * x + 2 * sx + kx];
            v[c] = a; mean += a;
        }
        mean /= (float)c1;
        float var = 0.f;
        for (int c = 0; c < c1; ++c) var += (v[c] - mean) * (v[c] - mean);
        var /= (float)c1;
        for (int c = 0; c < c1; ++c) mid[t][c] = gelu_erf((v[c] - mean) / sqrtf(var + 1e-6f) * g1[c] + be1[c]);
    }
    __syncthreads();
    if (t < c2) {  // conv2: weight [c2, c1, 2, 2]
        float a = b3[t];
        for (int c = 0; c < c1; ++c)
            for (int k = 0; k < 4; ++k) a += w3[(t * c1 + c) * 4 + k] * mid[k][c];
        hid[t] = a;
    }
    __syncthreads();
    if (t == 0) {
        float mean = 0.f;
        for (int c = 0; c < c2; ++c) mean += hid[c];
        mean /= (float)c2;
        float var = 0.f;
        for (int c = 0; c < c2; ++c) var += (hid[c] - mean) * (hid[c] - mean);
        var /= (float)c2;
        for (int c = 0; c < c2; ++c) hid[c] = gelu_erf((hid[c] - mean) / sqrtf(var + 1e-6f) * g4[c] + be4[c]);
    }
    __syncthreads();
    for (int c = t; c < C; c += 256) {
        float a = b6[c];
        for (int k = 0; k < c2; ++k) a += w6[c * c2 + k] * hid[k];
        out[((long)p * H * W + pix) * C + c] = a;
    }
}

extern "C" int ullsam_mask_downscale(const float* masks, float* out_nhwc, int P, int H, int W, int C, int c1, int c2, const float* w0,
                                     const float* b0, const float* g1, const float* be1, const float* w3, const float* b3,
                                     const float* g4, const float* be4, const float* w6, const float* b6, void* stream) {
    ULLSAM_CHECK(c1 <= 16 && c2 <= 64, "mask_downscale: c1<=16, c2<=64");
    if (P == 0) return 0;
    mask_downscale_kernel<<<P * H * W, 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(masks, out_nhwc, H, W, C, c1, c2, w0, b0, g1, be1, w3, b3, g4, be4, w6, b6);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// ---- masks = hyper_in @ upscaled_embedding (mask_decoder.py:143-144) ---------------------------------------------
// up2 f32 or bf16 [NB*H*W*4, 4*CU] : row = ((nb*H + y)*W + x)*4 + (ky*2+kx), col = (ky2*2+kx2)*CU + c  (two k2s2 ConvTranspose as GEMMs)
// hyper f32 [NB, NM, CU];  out f32 [NB, NM, 4H, 4W]
template <typename TU, int CU>
__global__ __launch_bounds__(256) void hyper_masks_kernel(const TU* __restrict__ up2, const float* __restrict__ hyper,
                                                          float* __restrict__ out, int NB, int NM, int H, int W) {
    __shared__ float hs[8 * CU];
    const int nb = blockIdx.y;
    for (int i = threadIdx.x; i < NM * CU; i += 256) hs[i] = hyper[(long)nb * NM * CU + i];
    __syncthreads();
    const long per = (long)H * W * 16;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < per; i += (long)gridDim.x * 256) {
        const int sub2 = i & 3;
        const long row = i >> 2;  // ((y*W + x)*4 + sub1)
        const int sub1 = row & 3;
        const long pix = row >> 2;
        const int y = (int)(pix / W), x = (int)(pix % W);
        const TU* src = up2 + (((long)nb * H * W * 4 + row) * 4 + sub2) * CU;
        float v[CU];
#pragma unroll
        for (int c = 0; c < CU; c += 4) {
            const float4 t = load4(src + c);
            v[c] = t.x; v[c + 1] = t.y; v[c + 2] = t.z; v[c + 3] = t.w;
        }
        const int Y = 4 * y + 2 * (sub1 >> 1) + (sub2 >> 1), X = 4 * x + 2 * (sub1 & 1) + (sub2 & 1);
        for (int m = 0; m < NM; ++m) {
            float a = 0.f;
#pragma unroll
            for (int c = 0; c < CU; ++c) a += hs[m * CU + c] * v[c];
            out[(((long)nb * NM + m) * 4 * H + Y) * 4 * W + X] = a;
        }
    }
}

extern "C" int ullsam_hyper_masks(int dtype, const void* up2, const float* hyper, float* out, int NB, int NM, int H, int W, int CU,
                                  void* stream) {
    ULLSAM_CHECK(CU == 32 && NM <= 8, "hyper_masks: CU must be 32, NM <= 8");
    if (NB == 0) return 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == ULLSAM_DT_F32) hyper_masks_kernel<float, 32><<<dim3(512, NB), 256, 0, s>>>((const float*)up2, hyper, out, NB, NM, H, W);
    else hyper_masks_kernel<bf16, 32><<<dim3(512, NB), 256, 0, s>>>((const bf16*)up2, hyper, out, NB, NM, H, W);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// ---- bilinear resize (align_corners=False) + optional threshold (app.py:635-645, sam.py:154-162,123) ----------------
// in f32 [N, IH(valid of in_ld rows), IW] with row stride in_ld and plane stride in_ps; out f32 [N,OH,OW] and/or mask u8
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ in, long in_ps, int in_ld, int IH, int IW,
                                                              float* __restrict__ out, unsigned char* __restrict__ mask, int N, int OH,
                                                              int OW, float thr) {
    const long total = (long)N * OH * OW;
    const float sy = (float)IH / (float)OH, sx = (float)IW / (float)OW;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ox = (int)(i % OW), oy = (int)((i / OW) % OH);
        const long n = i / ((long)OW * OH);
        const Tap ty = tap_of(oy, sy, IH), tx = tap_of(ox, sx, IW);
        const float* p = in + n * in_ps;
        const float top = lerp_rn(p[(long)ty.i0 * in_ld + tx.i0], p[(long)ty.i0 * in_ld + tx.i1], tx.l);
        const float bot = lerp_rn(p[(long)ty.i1 * in_ld + tx.i0], p[(long)ty.i1 * in_ld + tx.i1], tx.l);
        const float v = lerp_rn(top, bot, ty.l);
        if (out) out[i] = v;
        if (mask) mask[i] = v > thr ? 1 : 0;
    }
}

extern "C" int ullsam_resize_bilinear(const float* in, long in_plane_stride, int in_ld, int IH, int IW, float* out, unsigned char* mask,
                                      int N, int OH, int OW, float thr, void* stream) {
    const long total = (long)N * OH * OW;
    if (total == 0) return 0;
    const int grid = (int)min((total + 255) / 256, (long)2048 * 8);
    resize_bilinear_kernel<<<grid, 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(in, in_plane_stride, in_ld, IH, IW, out, mask, N, OH, OW, thr);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// ---- mask IoU (CalcIoU, train_joint_v2.py:683-694): counts[n] = {intersection, union} over u8 masks ------------------
__global__ __launch_bounds__(256) void mask_iou_kernel(const unsigned char* __restrict__ a, const unsigned char* __restrict__ b,
                                                       unsigned long long* __restrict__ counts, long per) {
    const long n = blockIdx.y;
    unsigned long long inter = 0, uni = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < per; i += (long)gridDim.x * 256) {
        const bool x = a[n * per + i] != 0, y = b[n * per + i] != 0;
        inter += (x && y); uni += (x || y);
    }
    for (int o = 32; o > 0; o >>= 1) { inter += __shfl_xor(inter, o, 64); uni += __shfl_xor(uni, o, 64); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&counts[2 * n], inter); atomicAdd(&counts[2 * n + 1], uni); }
}

extern "C" int ullsam_mask_iou_counts(const unsigned char* a, const unsigned char* b, unsigned long long* counts, int N, long per, void* stream) {
    if (N == 0) return 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (hipMemsetAsync(counts, 0, sizeof(unsigned long long) * 2 * N, s) != hipSuccess) { ullsam_set_error("mask_iou: memset failed"); return -2; }
    mask_iou_kernel<<<dim3(64, N), 256, 0, s>>>(a, b, counts, per);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// The image -> token half of a two-way block (transformer.py:176-182) for MANY prompts, bf16, embedding 256 / internal 128 / 8 heads of 16:
//     q = (keys + pe) Wq^T + bq;  a = softmax_heads(q k_tok^T / 4) v_tok;  upd = keys + a Wo^T + bo;  keys' = norm4(upd)
// as ONE pass over the image-side stream.  The separate launches (q GEMM with fp32 output, the few-keys attention, a cast, the output GEMM with
// its fp32 residual, the fan-out LayerNorm) move 2.1 GB per layer for 64 prompts x 4096 tokens; this kernel reads (keys + pe) in bf16 and keys in
// fp32 and writes keys' in fp32 / bf16 / bf16 (+ pe): 0.94 GB.
// Both weight matrices sit in LDS for the life of the workgroup (128 KiB) in MFMA A-fragment order (a wave's fragment read is one contiguous KiB).
// A wave takes 16 rows at a time with the weights as the FIRST operand (D^T = W X^T), so a lane ends up with its row's values:
//   1. q^T: tile h = head h; lane (row r, group g) holds q[16 h + 4 g .. + 3];
//   2. attention on v_mfma_f32_16x16x16_bf16, whose B operand (4 k-values per lane, k = 4 g + j) IS that accumulator layout: scores S^T[token, row] =
//      K_tok q^T per head (q as two bf16 terms, the tokens' keys pre-scaled by log2(e) / 4 as two terms: three products), softmax over the <= 16 tokens
//      = the lane's four values and the four lane groups (round 4 ran this on the VALU as ~1000 instructions per 16 rows with one wave per SIMD: 273 us;
//      the matrix pipe needs ~200), O^T[dim, row] = V_tok^T P^T the same way (P and V as two terms);
//   3. upd^T = Wo A^T with a head pair's 2 x 4 dims per lane as one 32-deep k-step (Wo's fragments are packed in that k order), the fp32 residual
//      loaded straight into the accumulators at the top; tile t covers columns 16 t + 4 g .. + 3 (the four groups' fp32 loads / stores are 64 contiguous bytes per row), the
//      LayerNorm needs two cross-lane sums over the four groups, and for the bf16 outputs neighbouring groups trade halves of a tile pair (v_permlane16_swap) so
//      that a lane stores 8 consecutive columns = 16 bytes.
// 512 threads: two waves per SIMD share the weights (<= 256 registers), one wave's loads and softmax under the other's products.
// A workgroup works inside ONE prompt (its tokens' k / v fragments); grid = prompts x workgroups per prompt.
// ---------------------------------------------------------------------------------------------------------------
struct I2tArgs {
    const bf16* xin; long in_mod;          // (keys + pe) in bf16 [rows (or in_mod rows shared by every prompt), 256]
    const float* res; long res_mod;        // keys fp32 (the residual), rows as xin
    const bf16* Wq; const float* bq;       // [128, 256], [128]
    const float* ktok; const float* vtok;  // fp32 [P, T, 128]
    const bf16* Wo; const float* bo;       // [256, 128], [256]
    const float* lnw; const float* lnb; float eps;
    const float* key_pe; long pe_rows;     // fp32 [pe_rows, 256]
    float* out_f32; bf16* out_c; bf16* out_c_pe;   // each optional
    int P, T, N, wg_per_prompt;
    float scale;
};
constexpr int I2T_WQ = 0, I2T_WO = 128 * 256 * 2, I2T_KF = I2T_WO + 256 * 128 * 2, I2T_VF = I2T_KF + 8 * 2 * 64 * 8, I2T_PAR = I2T_VF + 8 * 2 * 64 * 8;
constexpr int I2T_LDS = I2T_PAR + (128 + 3 * 256) * 4;                       // Wq | Wo | K fragments [head][term][lane] x 8 B | V fragments | bq, bo, lnw, lnb
constexpr int I2T_WAVES = 8;

__device__ __forceinline__ f32x4 mma16k16(s16x4 a, s16x4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }
__device__ __forceinline__ void split4(const f32x4 x, s16x4& hi, s16x4& lo) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const __bf16 h = (__bf16)x[i];
        hi[i] = __builtin_bit_cast(short, h);
        lo[i] = __builtin_bit_cast(short, (__bf16)(x[i] - (float)h));
    }
}

__global__ __launch_bounds__(64 * I2T_WAVES) void i2t_block_kernel(I2tArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* bq = reinterpret_cast<float*>(smem + I2T_PAR);
    float* bo = bq + 128;
    float* lnw = bo + 256;
    float* lnb = lnw + 256;
    constexpr int NT = 64 * I2T_WAVES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, g = lane >> 4;
    const int prompt = blockIdx.x / p.wg_per_prompt, part = blockIdx.x % p.wg_per_prompt;
    // Wq fragment (tile t, k-step ks): lane (m, gg) <- Wq[16 t + m][32 ks + 8 gg .. + 7]
    for (int c = tid; c < 8 * 8 * 64; c += NT) {
        const int f = c >> 6, ln = c & 63, t = f >> 3, ks = f & 7, m = ln & 15, gg = ln >> 4;
        *reinterpret_cast<uint4*>(smem + I2T_WQ + c * 16) = *reinterpret_cast<const uint4*>(p.Wq + (size_t)(16 * t + m) * 256 + 32 * ks + 8 * gg);
    }
    // Wo fragment (tile t = 2 t' + par, head pair hp): lane m = 4 gq + i <-> output column 32 t' + 16 par + 4 gq + i; its 8 k-values = dims 32 hp + 4 gg .. + 3 and 32 hp + 16 + 4 gg .. + 3
    for (int c = tid; c < 16 * 4 * 64; c += NT) {
        const int f = c >> 6, ln = c & 63, t = f >> 2, hp = f & 3, m = ln & 15, gg = ln >> 4;
        const int col = 32 * (t >> 1) + 16 * (t & 1) + 4 * (m >> 2) + (m & 3);
        const bf16* src = p.Wo + (size_t)col * 128 + 32 * hp + 4 * gg;
        const uint2 lo = *reinterpret_cast<const uint2*>(src), hi = *reinterpret_cast<const uint2*>(src + 16);
        *reinterpret_cast<uint4*>(smem + I2T_WO + c * 16) = uint4{lo.x, lo.y, hi.x, hi.y};
    }
    // the tokens' fragments (A operands of the 16-deep products), two bf16 terms each: K lane (m = token, gg): k_tok[m][16 h + 4 gg + j] * log2(e) * scale;
    // V lane (m = dim, gg): v_tok[4 gg + j][16 h + m]; tokens >= T are zeros (their scores are masked below)
    for (int c = tid; c < 8 * 64; c += NT) {
        const int h = c >> 6, ln = c & 63, m = ln & 15, gg = ln >> 4;
        const float sc2 = p.scale * 1.4426950408889634f;
        f32x4 kx, vx;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            kx[j] = m < p.T ? p.ktok[((size_t)prompt * p.T + m) * 128 + 16 * h + 4 * gg + j] * sc2 : 0.f;
            vx[j] = 4 * gg + j < p.T ? p.vtok[((size_t)prompt * p.T + 4 * gg + j) * 128 + 16 * h + m] : 0.f;
        }
        s16x4 kh, kl, vh, vl;
        split4(kx, kh, kl);
        split4(vx, vh, vl);
        *reinterpret_cast<s16x4*>(smem + I2T_KF + ((h * 2 + 0) * 64 + ln) * 8) = kh;
        *reinterpret_cast<s16x4*>(smem + I2T_KF + ((h * 2 + 1) * 64 + ln) * 8) = kl;
        *reinterpret_cast<s16x4*>(smem + I2T_VF + ((h * 2 + 0) * 64 + ln) * 8) = vh;
        *reinterpret_cast<s16x4*>(smem + I2T_VF + ((h * 2 + 1) * 64 + ln) * 8) = vl;
    }
    for (int c = tid; c < 256; c += NT) {
        if (c < 128) bq[c] = p.bq ? p.bq[c] : 0.f;
        bo[c] = p.bo ? p.bo[c] : 0.f;
        lnw[c] = p.lnw ? p.lnw[c] : 1.f;
        lnb[c] = p.lnb ? p.lnb[c] : 0.f;
    }
    __syncthreads();
    const int groups = (p.N + 15) / 16, stride = p.wg_per_prompt * I2T_WAVES;
    int grp = part * I2T_WAVES + wave;
    if (grp >= groups) return;
    bf16x8_t a[8];
    auto load = [&](int gr) __attribute__((always_inline)) {
        const int r = min(gr * 16 + l16, p.N - 1);                            // (in_mod / res_mod / pe_rows are 0 or N: a shared image's row is the row inside the prompt, no division)
        const bf16* ap = p.xin + ((size_t)(p.in_mod ? 0 : prompt) * p.N + r) * 256 + g * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) a[ks] = *reinterpret_cast<const bf16x8_t*>(ap + ks * 32);
    };
    load(grp);
    for (; grp < groups; grp += stride) {
        int lo_ = lane * 16;
        asm volatile("" : "+v"(lo_));                                         // (opaque per group: the fragment addresses are not loop invariants to hoist into registers)
        const char* wl = smem + lo_;
        const int rin = grp * 16 + l16;
        const bool live = rin < p.N;
        const int rr_ = min(rin, p.N - 1);
        const long row = (long)prompt * p.N + rr_;
        // ---- the fp32 residual rows go straight into the accumulators of the second product: their latency passes under the first product and the attention
        f32x4 u[16];
        {
            const float* rp = p.res + (p.res_mod ? (size_t)rr_ : (size_t)row) * 256 + g * 4;
#pragma unroll
            for (int t = 0; t < 16; ++t) u[t] = *reinterpret_cast<const f32x4*>(rp + 16 * t);
        }
        // ---- q^T = Wq X^T: lane (row l16, group g) gets q[16 h + 4 g + i]
        f32x4 q[8];
#pragma unroll
        for (int h = 0; h < 8; ++h) q[h] = *reinterpret_cast<const f32x4*>(bq + 16 * h + 4 * g);
        {   // batches of four fragment reads run ONE batch ahead of their MFMAs (two register sets pinned by the scheduling fences: hipcc's own order is read, wait, multiply)
            bf16x8_t wr[2][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) wr[0][i] = *reinterpret_cast<const bf16x8_t*>(wl + I2T_WQ + (i * 8) * 1024);
#pragma unroll
            for (int b = 0; b < 16; ++b) {                                    // batch b: k-step b / 2, heads 4 (b & 1) .. + 3
                if (b + 1 < 16) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) wr[(b + 1) & 1][i] = *reinterpret_cast<const bf16x8_t*>(wl + I2T_WQ + ((4 * ((b + 1) & 1) + i) * 8 + ((b + 1) >> 1)) * 1024);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i) q[4 * (b & 1) + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[b & 1][i], a[b >> 1], q[4 * (b & 1) + i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (grp + stride < groups) load(grp + stride);                        // the next group's rows: `a` is free from here, the loads pass under everything below
        // ---- attention per head on the 16-deep MFMA; the outputs leave rounded to bf16 like the cast before the output GEMM
        s16x4 ob[8];
#pragma unroll
        for (int h0 = 0; h0 < 8; h0 += 4) {                                   // four heads abreast: each step is four independent chains
            f32x4 sc[4], o[4];
            float l[4];
            const char* fl = wl - lane * 8;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int h = h0 + j;
                const s16x4 kh = *reinterpret_cast<const s16x4*>(fl + I2T_KF + (h * 2 + 0) * 512);
                const s16x4 kl = *reinterpret_cast<const s16x4*>(fl + I2T_KF + (h * 2 + 1) * 512);
                s16x4 qh, ql;
                split4(q[h], qh, ql);
                sc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                sc[j] = mma16k16(kl, qh, sc[j]);
                sc[j] = mma16k16(kh, ql, sc[j]);
                sc[j] = mma16k16(kh, qh, sc[j]);                              // S^T[token 4 g + i, row l16] in log2 units
            }
            float mx[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                mx[j] = -1e30f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    sc[j][i] = 4 * g + i < p.T ? sc[j][i] : -1e30f;
                    mx[j] = fmaxf(mx[j], sc[j][i]);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) mx[j] = fmaxf(mx[j], lane_xor16(mx[j]));
#pragma unroll
            for (int j = 0; j < 4; ++j) mx[j] = fmaxf(mx[j], lane_xor32(mx[j]));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int h = h0 + j;
                const s16x4 vh = *reinterpret_cast<const s16x4*>(fl + I2T_VF + (h * 2 + 0) * 512);
                const s16x4 vl = *reinterpret_cast<const s16x4*>(fl + I2T_VF + (h * 2 + 1) * 512);
                f32x4 pr;
                l[j] = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) { pr[i] = __builtin_amdgcn_exp2f(sc[j][i] - mx[j]); l[j] += pr[i]; }
                s16x4 ph, pl;
                split4(pr, ph, pl);
                o[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                o[j] = mma16k16(vl, ph, o[j]);
                o[j] = mma16k16(vh, pl, o[j]);
                o[j] = mma16k16(vh, ph, o[j]);                                // O^T[dim 16 h + 4 g + i, row l16]
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) l[j] += lane_xor16(l[j]);
#pragma unroll
            for (int j = 0; j < 4; ++j) l[j] += lane_xor32(l[j]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float inv = 1.0f / l[j];
#pragma unroll
                for (int i = 0; i < 4; ++i) ob[h0 + j][i] = __builtin_bit_cast(short, (__bf16)(o[j][i] * inv));
            }
        }
        // ---- upd^T = keys + Wo A^T (+ bo below): k-step hp of lane group g carries dims 32 hp + 4 g .. + 3 and 32 hp + 16 + 4 g .. + 3 = ob[2 hp], ob[2 hp + 1]
        {
            typedef __attribute__((ext_vector_type(8))) short s16x8;
            bf16x8_t af[4];
#pragma unroll
            for (int hp = 0; hp < 4; ++hp) {
                s16x8 af8;
#pragma unroll
                for (int e = 0; e < 4; ++e) { af8[e] = ob[2 * hp][e]; af8[4 + e] = ob[2 * hp + 1][e]; }
                af[hp] = __builtin_bit_cast(bf16x8_t, af8);
            }
            bf16x8_t wr[2][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) wr[0][i] = *reinterpret_cast<const bf16x8_t*>(wl + I2T_WO + (i * 4) * 1024);
#pragma unroll
            for (int b = 0; b < 16; ++b) {                                    // batch b: head pair b / 4, tiles 4 (b & 3) .. + 3
                if (b + 1 < 16) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) wr[(b + 1) & 1][i] = *reinterpret_cast<const bf16x8_t*>(wl + I2T_WO + ((4 * ((b + 1) & 3) + i) * 4 + ((b + 1) >> 2)) * 1024);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i) u[4 * (b & 3) + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[b & 1][i], af[b >> 2], u[4 * (b & 3) + i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- + bo, LayerNorm over the 256 columns (64 here, the rest in the lanes l16 + 16, + 32, + 48), normalised IN PLACE; then the three outputs ONE BUFFER AT A TIME
        // (fp32, bf16, bf16 + pe): interleaved per tile pair, a wave's 32 stores walked three DRAM streams at once and the positional rows' loads sat between stores
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            u[t] += *reinterpret_cast<const f32x4*>(bo + 16 * t + 4 * g);
            sum += (u[t][0] + u[t][1]) + (u[t][2] + u[t][3]);
        }
        sum += lane_xor16(sum); sum += lane_xor32(sum);
        const float mean = sum * (1.0f / 256.0f);
        float ss = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const f32x4 d = u[t] - mean;
            ss += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
        ss += lane_xor16(ss); ss += lane_xor32(ss);
        const float rstd = 1.0f / sqrtf(ss * (1.0f / 256.0f) + p.eps);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int col = 16 * t + 4 * g;                                    // tile t holds columns 16 t + 4 g .. + 3
            u[t] = (u[t] - mean) * rstd * *reinterpret_cast<const f32x4*>(lnw + col) + *reinterpret_cast<const f32x4*>(lnb + col);
            if ((t & 3) == 3) __builtin_amdgcn_sched_barrier(0);               // (else the parameter reads of all sixteen tiles are hoisted)
        }
        if (p.out_f32 && live) {                                               // the four groups' stores are 64 contiguous bytes per row and instruction
            float* of = p.out_f32 + (size_t)row * 256 + 4 * g;
#pragma unroll
            for (int t = 0; t < 16; ++t) *reinterpret_cast<f32x4*>(of + 16 * t) = u[t];
        }
        // the bf16 outputs want 8 consecutive columns per lane (16-byte stores): groups 0 / 1 and 2 / 3 trade halves of a tile pair (u[t] of the odd group <-> u[t + 1] of the even
        // group), after which group g holds columns 16 t + 16 (g & 1) + 8 (g >> 1) .. + 7 in (u[t], u[t + 1]), t even
#pragma unroll
        for (int t = 0; t < 16; t += 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {                                      // (one exchanged value per lane: hipcc miscompiles the builtin when BOTH of its results are used --
                const float got = lane_xor16((g & 1) ? u[t][e] : u[t + 1][e]); //  it copies the first into the second)
                if (g & 1) u[t][e] = got; else u[t + 1][e] = got;
            }
        }
        const size_t off0 = (size_t)row * 256 + 16 * (g & 1) + 8 * (g >> 1);
        if (p.out_c && live) {
#pragma unroll
            for (int t = 0; t < 16; t += 2) {
                bf16x8_t c;
#pragma unroll
                for (int e = 0; e < 4; ++e) { c[e] = (__bf16)u[t][e]; c[4 + e] = (__bf16)u[t + 1][e]; }
                *reinterpret_cast<bf16x8_t*>(p.out_c + off0 + 16 * t) = c;
            }
        }
        if (p.out_c_pe && live) {
            // the positional rows: all sixteen loads of this pass are issued before its first store (requested before the OTHER passes' stores as well they cost
            // 64 more live registers through those passes and the two-output form ran 147 -> 163 us)
            const float* pe = p.key_pe + (size_t)rr_ * 256 + 16 * (g & 1) + 8 * (g >> 1);
#pragma unroll
            for (int t = 0; t < 16; t += 2) {
                u[t] += *reinterpret_cast<const f32x4*>(pe + 16 * t);
                u[t + 1] += *reinterpret_cast<const f32x4*>(pe + 16 * t + 4);
            }
#pragma unroll
            for (int t = 0; t < 16; t += 2) {
                bf16x8_t c;
#pragma unroll
                for (int e = 0; e < 4; ++e) { c[e] = (__bf16)u[t][e]; c[4 + e] = (__bf16)u[t + 1][e]; }
                *reinterpret_cast<bf16x8_t*>(p.out_c_pe + off0 + 16 * t) = c;
            }
        }
    }
}
// xin bf16 [P*N (or in_mod), 256]; res fp32 alike; Wq bf16 [128, 256]; ktok / vtok fp32 [P, T, 128] (T <= 16); Wo bf16 [256, 128]; key_pe fp32 [pe_rows, 256];
// outputs [P*N, 256], each optional (NULL).  scale = 1 / sqrt(16).
extern "C" int ullsam_i2t_block(const void* xin, long in_mod, const float* res, long res_mod, const void* Wq, const float* bq, const float* ktok,
                                const float* vtok, const void* Wo, const float* bo, const float* lnw, const float* lnb, float eps, const float* key_pe,
                                long pe_rows, float* out_f32, void* out_c, void* out_c_pe, int P, int T, int N, float scale, void* stream) {
    ULLSAM_CHECK(P > 0 && N > 0 && T >= 1 && T <= 16, "i2t_block: P=%d N=%d T=%d (1..16)", P, N, T);
    ULLSAM_CHECK(xin && res && Wq && Wo && ktok && vtok && (!out_c_pe || key_pe), "i2t_block: null operand");
    ULLSAM_CHECK((in_mod == 0 || in_mod == N) && (res_mod == 0 || res_mod == N) && (!out_c_pe || pe_rows == N), "i2t_block: in_mod=%ld res_mod=%ld pe_rows=%ld must be 0 (per prompt) or N=%d",
                 in_mod, res_mod, pe_rows, N);
    ULLSAM_CHECK(((((uintptr_t)xin | (uintptr_t)res | (uintptr_t)Wq | (uintptr_t)Wo | (uintptr_t)out_f32 | (uintptr_t)out_c | (uintptr_t)out_c_pe | (uintptr_t)key_pe)) & 15) == 0,
                 "i2t_block: 16-byte aligned operands needed");
    I2tArgs a{static_cast<const bf16*>(xin), in_mod, res, res_mod, static_cast<const bf16*>(Wq), bq, ktok, vtok, static_cast<const bf16*>(Wo), bo, lnw, lnb, eps,
              key_pe, pe_rows, out_f32, static_cast<bf16*>(out_c), static_cast<bf16*>(out_c_pe), P, T, N, 1, scale};
    const int groups = (N + 15) / 16;
    { const int by_rows = (groups + I2T_WAVES - 1) / I2T_WAVES, by_cus = (256 + P - 1) / P; a.wg_per_prompt = by_rows < by_cus ? by_rows : by_cus; if (a.wg_per_prompt < 1) a.wg_per_prompt = 1; }
    static PerDeviceOnce attr;
    if (attr.first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(i2t_block_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, I2T_LDS);
    i2t_block_kernel<<<dim3(P * a.wg_per_prompt), 64 * I2T_WAVES, I2T_LDS, reinterpret_cast<hipStream_t>(stream)>>>(a);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// The token -> image attention's k and v projections of the image side for MANY prompts in one pass (transformer.py:123-126 / Attention.forward :220-222, bf16):
//     K = (keys + pe) Wk^T + bk,   V = keys Wv^T + bv          [rows, 256] x [128, 256]^T -> [rows, 128] each
// Two launches of the 128x128 tile GEMM (2048 workgroups of 64 KB each, 61 - 77 us apiece for 203 MB) ran at 2.6 - 3.3 TB/s; here both weight matrices sit in LDS as
// MFMA B fragments for the life of the workgroup (128 KiB) and a wave streams 16 rows at a time through both products (rows first: D = X W^T), the next group's rows
// requested under the other product; the weight rows are permuted so that a lane's eight tiles are 8 consecutive features and a store instruction writes whole rows (kv_store).
// ---------------------------------------------------------------------------------------------------------------
struct KvArgs { const bf16* xk; const bf16* xv; const bf16* Wk; const bf16* Wv; const float* bk; const float* bv; bf16* K; bf16* V; long rows; };
constexpr int KV_WAVES = 8;
__device__ __forceinline__ void kv_load(bf16x8_t (&a)[8], const bf16* x, long gr, long rows, int l16, int g) {
    const long row = min(gr * 16 + l16, rows - 1);
    const bf16* ap = x + (size_t)row * 256 + g * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) a[ks] = *reinterpret_cast<const bf16x8_t*>(ap + ks * 32);
}
__device__ __forceinline__ void kv_product(f32x4 (&acc)[8], const bf16x8_t (&a)[8], const char* wl, const float* bias, int l16) {
    {   // the lane's 8 features 64 (l16 >> 3) + 8 (l16 & 7) .. + 7 (tile t = offset t): the same bias for its four rows
        const f32x4 b0v = *reinterpret_cast<const f32x4*>(bias + 8 * l16), b1v = *reinterpret_cast<const f32x4*>(bias + 8 * l16 + 4);
#pragma unroll
        for (int t = 0; t < 4; ++t) { acc[t] = f32x4{b0v[t], b0v[t], b0v[t], b0v[t]}; acc[4 + t] = f32x4{b1v[t], b1v[t], b1v[t], b1v[t]}; }
    }
    bf16x8_t wr[2][4];                                                        // batches of four fragment reads one batch ahead of their MFMAs
#pragma unroll
    for (int i = 0; i < 4; ++i) wr[0][i] = *reinterpret_cast<const bf16x8_t*>(wl + (i * 8) * 1024);
#pragma unroll
    for (int b = 0; b < 16; ++b) {                                            // batch b: k-step b / 2, tiles 4 (b & 1) .. + 3
        if (b + 1 < 16) {
#pragma unroll
            for (int i = 0; i < 4; ++i) wr[(b + 1) & 1][i] = *reinterpret_cast<const bf16x8_t*>(wl + ((4 * ((b + 1) & 1) + i) * 8 + ((b + 1) >> 1)) * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[4 * (b & 1) + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[b >> 1], wr[b & 1][i], acc[4 * (b & 1) + i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}
// rows first (D = X W^T: a lane holds rows 4 g + e of column l16 of every tile) with the weight rows permuted so that column n of tile t is feature
// 64 (n >> 3) + 8 (n & 7) + t: a lane's eight tiles are 8 CONSECUTIVE features of a row, the 16 lanes of a lane group cover the row's 128 features, and one
// store instruction writes four whole rows = 4 x 256 contiguous bytes (round 5 held one row per lane and traded tile halves between lane groups: 64-byte pieces per row and instruction)
__device__ __forceinline__ void kv_store(const f32x4 (&acc)[8], bf16* out, long gr, long rows, int l16, int g) {
    const long row0 = gr * 16 + 4 * g;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        bf16x8_t c;
#pragma unroll
        for (int t = 0; t < 8; ++t) c[t] = (__bf16)acc[t][e];
        if (row0 + e < rows) *reinterpret_cast<bf16x8_t*>(out + (size_t)(row0 + e) * 128 + 8 * l16) = c;
    }
}
__global__ __launch_bounds__(64 * KV_WAVES) void kv_proj_kernel(KvArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* bs = reinterpret_cast<float*>(smem + 2 * 128 * 256 * 2);          // bk [128] | bv [128]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, g = lane >> 4;
    // B fragment (matrix w, tile t, k-step ks) = one contiguous KiB: lane (n, gg) <- W[64 (n >> 3) + 8 (n & 7) + t][32 ks + 8 gg .. + 7]
    for (int c = tid; c < 2 * 8 * 8 * 64; c += 64 * KV_WAVES) {
        const int w = c >> 12, f = (c >> 6) & 63, ln = c & 63, t = f >> 3, ks = f & 7, n = ln & 15, gg = ln >> 4;
        *reinterpret_cast<uint4*>(smem + c * 16) = *reinterpret_cast<const uint4*>((w ? p.Wv : p.Wk) + (size_t)(64 * (n >> 3) + 8 * (n & 7) + t) * 256 + 32 * ks + 8 * gg);
    }
    if (tid < 128) { bs[tid] = p.bk ? p.bk[tid] : 0.f; bs[128 + tid] = p.bv ? p.bv[tid] : 0.f; }
    __syncthreads();
    const long groups = (p.rows + 15) / 16, stride = (long)gridDim.x * KV_WAVES;
    long grp = (long)blockIdx.x * KV_WAVES + wave;
    if (grp >= groups) return;
    bf16x8_t ak[8], av[8];
    kv_load(ak, p.xk, grp, p.rows, l16, g);
    kv_load(av, p.xv, grp, p.rows, l16, g);
    for (; grp < groups; grp += stride) {
        int lo_ = lane * 16;
        asm volatile("" : "+v"(lo_));                                         // (opaque per group: the fragment addresses are not loop invariants to hoist into registers)
        const char* wl = smem + lo_;
        const long nxt = min(grp + stride, groups - 1);                       // (the last trip re-requests its own rows: no branch around the loads)
        f32x4 acc[8], acc2[8];
        kv_product(acc, ak, wl, bs, l16);
        kv_load(ak, p.xk, nxt, p.rows, l16, g);                               // the next group's k rows pass under the v product
        kv_store(acc, p.K, grp, p.rows, l16, g);
        kv_product(acc2, av, wl + 8 * 8 * 1024, bs + 128, l16);
        kv_load(av, p.xv, nxt, p.rows, l16, g);                               // its v rows under the stores and the next k product
        kv_store(acc2, p.V, grp, p.rows, l16, g);
    }
}
// xk, xv bf16 [rows, 256] (keys + pe, keys); Wk, Wv bf16 [128, 256]; bk, bv fp32 [128] | NULL; K, V bf16 [rows, 128]
extern "C" int ullsam_kv_proj(const void* xk, const void* xv, const void* Wk, const void* Wv, const float* bk, const float* bv, void* K, void* V, long rows, void* stream) {
    ULLSAM_CHECK(rows > 0 && xk && xv && Wk && Wv && K && V, "kv_proj: rows=%ld or a null operand", rows);
    ULLSAM_CHECK((((uintptr_t)xk | (uintptr_t)xv | (uintptr_t)Wk | (uintptr_t)Wv | (uintptr_t)K | (uintptr_t)V) & 15) == 0, "kv_proj: 16-byte aligned operands needed");
    KvArgs a{static_cast<const bf16*>(xk), static_cast<const bf16*>(xv), static_cast<const bf16*>(Wk), static_cast<const bf16*>(Wv), bk, bv, static_cast<bf16*>(K), static_cast<bf16*>(V), rows};
    constexpr int LDS = 2 * 128 * 256 * 2 + 256 * 4;
    static PerDeviceOnce attr;
    if (attr.first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kv_proj_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    const long groups = (rows + 15) / 16, want = (groups + KV_WAVES - 1) / KV_WAVES;
    kv_proj_kernel<<<dim3((unsigned)(want < 256 ? want : 256)), 64 * KV_WAVES, LDS, reinterpret_cast<hipStream_t>(stream)>>>(a);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Second transposed convolution + GELU + hypernetwork product in one pass (mask_decoder.py:136-147, bf16):
//     masks[nb][m][Y][X] = sum_c hyper[nb][m][c] * bf16(GELU(u1[row] . w1[tap * 32 + c] + b1[tap * 32 + c]))
// u1 bf16 [NB*H*W*4, 64] (the first transposed convolution after LayerNorm2d + GELU; row = ((nb H + y) W + x) 4 + sub1), w1 bf16 [128 = (ky2, kx2, c), 64],
// hyper fp32 [NB, NM, 32]; out fp32 [NB, NM, 4H, 4W].  The upscaled embedding [NB*H*W*4, 128] (268 MB for 64 prompts) is never written: w1 sits in LDS
// with its rows permuted so that, with the weights as the first MFMA operand, lane (row, g) ends up with the 32 channels of ONE tap (tap = g) -- bias, GELU,
// the rounding to bf16 the separate launches apply, and the NM dot products with the prompt's hypernetwork vectors are lane-local.
// A workgroup works inside one prompt (its hyper vectors in LDS).
// ---------------------------------------------------------------------------------------------------------------
struct Up2Args { const bf16* u1; const bf16* w1; const float* b1; const float* hyper; float* out; int NB, NM, H, W, wg_per_prompt; };
__global__ __launch_bounds__(256, 3) void up2_hyper_kernel(Up2Args p) {
    __shared__ __attribute__((aligned(16))) char wl[128 * 64 * 2];   // w1: physical row n at LDS row 16 t + 4 gq + i (gq = n / 32 = tap, t = (n % 32) / 4), 8 chunks per row
    __shared__ __attribute__((aligned(16))) float hs[8 * 32];
    __shared__ __attribute__((aligned(16))) float bs[128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, g = lane >> 4;
    const int nb = blockIdx.x / p.wg_per_prompt, part = blockIdx.x % p.wg_per_prompt;
    for (int c = tid; c < 128 * 8; c += 256) {
        const int n = c >> 3, ch = c & 7, gq = n >> 5, rem = n & 31, L = 16 * (rem >> 2) + 4 * gq + (rem & 3);
        *reinterpret_cast<uint4*>(wl + (L * 8 + (ch ^ ((L >> 1) & 7))) * 16) = *reinterpret_cast<const uint4*>(p.w1 + (size_t)n * 64 + ch * 8);
    }
    for (int c = tid; c < p.NM * 32; c += 256) hs[c] = p.hyper[(size_t)nb * p.NM * 32 + c];
    if (tid < 128) bs[tid] = p.b1 ? p.b1[tid] : 0.f;
    __syncthreads();
    const long rows = (long)p.H * p.W * 4;                 // rows of this prompt
    const int groups = (int)((rows + 15) / 16), stride = p.wg_per_prompt * 4;
    bf16x8_t a0[2], a1[2];
    auto load = [&](bf16x8_t (&a)[2], int grp) {
        const long row = (long)nb * rows + min((long)grp * 16 + l16, rows - 1);
        const bf16* ap = p.u1 + (size_t)row * 64 + g * 8;
        a[0] = *reinterpret_cast<const bf16x8_t*>(ap);
        a[1] = *reinterpret_cast<const bf16x8_t*>(ap + 32);
    };
    auto compute = [&](const bf16x8_t (&a)[2], int grp) {
        f32x4 acc[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = *reinterpret_cast<const f32x4*>(bs + g * 32 + 4 * t);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int L = 16 * t + l16;
                const bf16x8_t w = *reinterpret_cast<const bf16x8_t*>(wl + (L * 8 + ((ks * 4 + g) ^ ((L >> 1) & 7))) * 16);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, a[ks], acc[t], 0, 0, 0);
            }
        {
            const long rin = (long)grp * 16 + l16;
            if (rin >= rows) return;
            float sm[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) sm[m] = 0.f;
#pragma unroll
            for (int t = 0; t < 8; ++t) {              // channels 4 t .. 4 t + 3 of this lane's tap, in ascending order (the order hyper_masks_kernel sums in)
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = (float)(__bf16)gelu_erfc5(acc[t][i]);   // (the upscaled embedding was a bf16 tensor between the two launches; the one-transcendental GELU of the ring GEMM's bf16 epilogues, within 1.9e-6 of the erf form: the kernel is bound by this arithmetic)
#pragma unroll
                for (int m = 0; m < 8; ++m)
                    if (m < p.NM) {
                        const f32x4 h4 = *reinterpret_cast<const f32x4*>(hs + m * 32 + 4 * t);   // (one broadcast ds_read_b128 instead of four b32)
#pragma unroll
                        for (int i = 0; i < 4; ++i) sm[m] += h4[i] * v[i];
                    }
                __builtin_amdgcn_sched_barrier(0);     // (one GELU group at a time: interleaving all 32 evaluations costs hipcc 280 registers)
            }
            const int sub1 = (int)(rin & 3);
            const long pix = rin >> 2;
            const int y = (int)(pix / p.W), x = (int)(pix % p.W);
            const int Y = 4 * y + 2 * (sub1 >> 1) + (g >> 1), X = 4 * x + 2 * (sub1 & 1) + (g & 1);
#pragma unroll
            for (int m = 0; m < 8; ++m)
                if (m < p.NM) p.out[(((size_t)nb * p.NM + m) * 4 * p.H + Y) * 4 * p.W + X] = sm[m];
        }
    };
    int grp = part * 4 + wave;
    if (grp >= groups) return;
    load(a0, grp);
    while (true) {
        const bool more1 = grp + stride < groups;
        if (more1) load(a1, grp + stride);
        compute(a0, grp);
        if (!more1) break;
        grp += stride;
        const bool more0 = grp + stride < groups;
        if (more0) load(a0, grp + stride);
        compute(a1, grp);
        if (!more0) break;
        grp += stride;
    }
}
// u1 bf16 [NB*H*W*4, 64]; w1 bf16 [128, 64] = (ky2, kx2, c) x cin; b1 fp32 [128] or NULL; hyper fp32 [NB, NM, 32] (NM <= 8); out fp32 [NB, NM, 4H, 4W]
extern "C" int ullsam_up2_hyper_masks(const void* u1, const void* w1, const float* b1, const float* hyper, float* out, int NB, int NM, int H, int W, void* stream) {
    ULLSAM_CHECK(NB > 0 && NM >= 1 && NM <= 8 && H > 0 && W > 0, "up2_hyper_masks: NB=%d NM=%d H=%d W=%d", NB, NM, H, W);
    ULLSAM_CHECK((((uintptr_t)u1 | (uintptr_t)w1) & 15) == 0, "up2_hyper_masks: 16-byte aligned operands needed");
    Up2Args a{static_cast<const bf16*>(u1), static_cast<const bf16*>(w1), b1, hyper, out, NB, NM, H, W, 1};
    const long groups = ((long)H * W * 4 + 15) / 16;
    { const long by_rows = (groups + 15) / 16, by_cus = (2048 + NB - 1) / NB;   /* eight workgroups per CU (84 registers, 18 KB of LDS each), at least four row groups per wave */
      a.wg_per_prompt = (int)(by_rows < by_cus ? by_rows : by_cus); if (a.wg_per_prompt < 1) a.wg_per_prompt = 1; }
    up2_hyper_kernel<<<dim3(NB * a.wg_per_prompt), 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(a);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// First transposed convolution + LayerNorm2d + GELU in one pass (mask_decoder.py:131-138, bf16):
//     u1[row * 4 + tap][c] = bf16(GELU(LayerNorm_c(src[row] . w0[tap * 64 + c] + b0[tap * 64 + c])))
// src bf16 [rows, 256] (the image side of the two-way transformer), w0 bf16 [256 = (ky, kx, c), 256]; out bf16 [rows * 4, 64] = [rows, 256].
// The fp32 result of the convolution ([rows, 256] fp32 = 268 MB for 64 prompts) used to be written by the GEMM and read back by the 64-channel
// LayerNorm; here w0 (128 KiB) sits in LDS as B fragments with its rows permuted so that a lane holds, for FOUR rows (4 g + e), 8 consecutive channels
// 8 (l16 % 8) .. + 7 of tap 2 T + l16 / 8 (tile T * 8 + t = channel offset t): a (row, tap)'s 64 channels sit in 8 neighbouring lanes (mean and variance: an
// 8-value lane sum + three DPP steps), and ONE store instruction writes 16 lanes x 16 B = 256 contiguous bytes per row -- whole 128-byte lines.  (Round 5's
// layout kept a (row, tap)'s 64 channels in one lane: lane-local statistics, but eight 16-byte stores per 128-byte line, each a separate instruction:
// WRITE_SIZE 214 MB for 134 MB of output, lines left the L2 half-written.)
// ---------------------------------------------------------------------------------------------------------------
struct Up1Args { const bf16* src; const bf16* w0; const float* b0; const float* lnw; const float* lnb; float eps; bf16* out; long rows; };
constexpr int UP1_WAVES = 8;       // two per SIMD share the weights (162 registers; twelve waves were measured: 90.9 vs 84.4 us for 64 prompts)
__global__ __launch_bounds__(64 * UP1_WAVES) void up1_ln_gelu_kernel(Up1Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* bs = reinterpret_cast<float*>(smem + 256 * 256 * 2);      // b0 [256] | lnw [64] | lnb [64]
    float* lw = bs + 256;
    float* lb = lw + 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, g = lane >> 4;
    // w0 -> LDS in MFMA B-fragment order: fragment (tile tt, k-step ks) = one contiguous KiB, lane (n, gg) <- w0[row(tt, n)][32 ks + 8 gg .. + 7] with
    // row(tt, n) = 64 (2 (tt >> 3) + (n >> 3)) + 8 (n & 7) + (tt & 7): column n of tile tt is channel 8 (n & 7) + (tt & 7) of tap 2 (tt >> 3) + (n >> 3)
    for (int c = tid; c < 16 * 8 * 64; c += 64 * UP1_WAVES) {
        const int f = c >> 6, ln = c & 63, tt = f >> 3, ks = f & 7, n = ln & 15, gg = ln >> 4;
        *reinterpret_cast<uint4*>(smem + c * 16) = *reinterpret_cast<const uint4*>(p.w0 + (size_t)(64 * (2 * (tt >> 3) + (n >> 3)) + 8 * (n & 7) + (tt & 7)) * 256 + 32 * ks + 8 * gg);
    }
    if (tid < 256) bs[tid] = p.b0 ? p.b0[tid] : 0.f;
    if (tid < 64) { lw[tid] = p.lnw ? p.lnw[tid] : 1.f; lb[tid] = p.lnb ? p.lnb[tid] : 0.f; }
    __syncthreads();
    const long groups = (p.rows + 15) / 16;
    const long stride = (long)gridDim.x * UP1_WAVES;
    const int l8 = l16 & 7, hp = l16 >> 3;                                    // the lane's channel octet and the parity of its taps
    float lwr[8], lbr[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) { lwr[t] = lw[8 * l8 + t]; lbr[t] = lb[8 * l8 + t]; }
    bf16x8_t a[8];
    auto load = [&](long grp) {
        const long row = min(grp * 16 + l16, p.rows - 1);
        const bf16* ap = p.src + (size_t)row * 256 + g * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) a[ks] = *reinterpret_cast<const bf16x8_t*>(ap + ks * 32);
    };
    auto product = [&](f32x4 (&u)[16]) __attribute__((always_inline)) {
        int lo_ = lane * 16;
        asm volatile("" : "+v"(lo_));                                         // (opaque per group: the fragment addresses are not loop invariants to hoist into registers)
        const char* wl = smem + lo_;
#pragma unroll
        for (int T = 0; T < 2; ++T) {                                         // the lane's 8 channels of tap 2 T + hp: the same bias for its four rows
            const f32x4 b0v = *reinterpret_cast<const f32x4*>(bs + (2 * T + hp) * 64 + 8 * l8), b1v = *reinterpret_cast<const f32x4*>(bs + (2 * T + hp) * 64 + 8 * l8 + 4);
#pragma unroll
            for (int t = 0; t < 4; ++t) { u[8 * T + t] = f32x4{b0v[t], b0v[t], b0v[t], b0v[t]}; u[8 * T + 4 + t] = f32x4{b1v[t], b1v[t], b1v[t], b1v[t]}; }
        }
        // batches of four fragment reads run one batch ahead of their MFMAs (two register sets pinned by the scheduling fences)
        bf16x8_t wr[2][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) wr[0][i] = *reinterpret_cast<const bf16x8_t*>(wl + (i * 8) * 1024);
#pragma unroll
        for (int b = 0; b < 32; ++b) {                                        // batch b: k-step b / 4, tiles 4 (b & 3) .. + 3
            if (b + 1 < 32) {
#pragma unroll
                for (int i = 0; i < 4; ++i) wr[(b + 1) & 1][i] = *reinterpret_cast<const bf16x8_t*>(wl + ((4 * ((b + 1) & 3) + i) * 8 + ((b + 1) >> 2)) * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) u[4 * (b & 3) + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[b >> 2], wr[b & 1][i], u[4 * (b & 3) + i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto sum8 = [](float v) __attribute__((always_inline)) {                  // over the 8 lanes l16 % 8 = 0 .. 7 of a (row quad, tap): every lane gets the total
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));  // row_half_mirror
        return v;
    };
    auto finish = [&](f32x4 (&u)[16], long grp) {
        const long row0 = grp * 16 + 4 * g;
        // the eight (tap pair T, row e) statistics of the lane as batches (8 independent DPP chains at a time, not one after the other)
        float mean[8], rstd[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int T = k >> 2, e = k & 3;
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < 8; t += 2) sum += u[8 * T + t][e] + u[8 * T + t + 1][e];
            mean[k] = sum;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) mean[k] = sum8(mean[k]) * (1.0f / 64.0f);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int T = k >> 2, e = k & 3;
            float ss = 0.f;
#pragma unroll
            for (int t = 0; t < 8; t += 2) {
                const float d0 = u[8 * T + t][e] - mean[k], d1 = u[8 * T + t + 1][e] - mean[k];
                ss += d0 * d0 + d1 * d1;
            }
            rstd[k] = ss;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) rstd[k] = 1.0f / sqrtf(sum8(rstd[k]) * (1.0f / 64.0f) + p.eps);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int T = k >> 2, e = k & 3;
            bf16x8_t c;
#pragma unroll
            for (int t = 0; t < 8; ++t) c[t] = (__bf16)gelu_erfc5((u[8 * T + t][e] - mean[k]) * rstd[k] * lwr[t] + lbr[t]);
            if (row0 + e < p.rows) *reinterpret_cast<bf16x8_t*>(p.out + (size_t)(row0 + e) * 256 + T * 128 + l16 * 8) = c;   // 16 lanes: 256 contiguous bytes of the row
        }
    };
    long grp = (long)blockIdx.x * UP1_WAVES + wave;
    if (grp >= groups) return;
    load(grp);
    while (true) {
        f32x4 u[16];
        product(u);
        const long nxt = grp + stride;
        const bool more = nxt < groups;
        if (more) load(nxt);                           // the next group's fragments land under this group's LayerNorm / GELU (one register set: two spilled)
        finish(u, grp);
        if (!more) break;
        grp = nxt;
    }
}
// src bf16 [rows, 256]; w0 bf16 [256, 256] = (ky, kx, c) x cin; b0 fp32 [256] | NULL; lnw / lnb fp32 [64] | NULL; out bf16 [rows * 4, 64]
extern "C" int ullsam_up1_ln_gelu(const void* src, const void* w0, const float* b0, const float* lnw, const float* lnb, float eps, void* out, long rows, void* stream) {
    ULLSAM_CHECK(rows > 0, "up1_ln_gelu: rows=%ld", rows);
    ULLSAM_CHECK((((uintptr_t)src | (uintptr_t)w0 | (uintptr_t)out) & 15) == 0, "up1_ln_gelu: 16-byte aligned operands needed");
    Up1Args a{static_cast<const bf16*>(src), static_cast<const bf16*>(w0), b0, lnw, lnb, eps, static_cast<bf16*>(out), rows};
    constexpr int LDS = 256 * 256 * 2 + (256 + 128) * 4;
    static PerDeviceOnce attr;
    if (attr.first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(up1_ln_gelu_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    const long groups = (rows + 15) / 16;
    const long want = (groups + UP1_WAVES - 1) / UP1_WAVES;
    up1_ln_gelu_kernel<<<dim3((unsigned)(want < 256 ? want : 256)), 64 * UP1_WAVES, LDS, reinterpret_cast<hipStream_t>(stream)>>>(a);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}
