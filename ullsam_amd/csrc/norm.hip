// Row-wise LayerNorm / RMSNorm with fp32 statistics.
//   LayerNorm  : image_encoder.py:151,161 (eps 1e-6 via build_sam.py:72); LayerNorm2d common.py:38-43 (NHWC rows ==
//                per-pixel channel LN, biased variance); nn.LayerNorm in mlp1/mlp2 (modeling_internvl_sam.py:89,96)
//                and the decoder (transformer.py, eps 1e-5); F.layer_norm without affine (prompt_encoder.py:142-145).
//   RMSNorm    : InternLM2RMSNorm.forward modeling_internlm2.py:138-143 (fp32 variance, weight applied last).
// One wave per row, 16-byte loads, two-pass (mean, then centred variance) entirely in registers: HBM-bound,
// algorithmic bytes = rows*D*(sizeof(in)+sizeof(out)).
// Optional fused extras used by the decoder / prompt encoder: post-scale+shift (llm_scale_factor/llm_bias) and
// an activation (GELU) after the affine.
#include "common.h"

struct NormArgs {
    const void* in;
    void* out;
    const float* w;
    const float* b;
    long rows;
    int D;
    long in_stride, out_stride;
    float eps;
    int rms;
    int act;  // 0 none, 1 gelu
    const float* post_scale;  // device scalar or null
    const float* post_shift;  // device scalar or null
};

template <typename TI, typename TO, int MAXV>
__global__ __launch_bounds__(256) void norm_kernel(NormArgs p) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const TI* x = reinterpret_cast<const TI*>(p.in) + row * p.in_stride;
    TO* y = reinterpret_cast<TO*>(p.out) + row * p.out_stride;
    const int nv = p.D >> 2;  // float4 groups
    float4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = i * 64 + lane;
        if (idx < nv) {
            v[i] = load4(x + idx * 4);
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        } else {
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    float mean = 0.f;
    if (!p.rms) mean = wave_sum(s) / (float)p.D;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = i * 64 + lane;
        if (idx < nv) {
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            ss += (a * a + b * b) + (c * c + d * d);
        }
    }
    const float var = wave_sum(ss) / (float)p.D;
    const float rstd = p.rms ? rsqrtf(var + p.eps) : 1.0f / sqrtf(var + p.eps);
    const float ps = p.post_scale ? p.post_scale[0] : 1.f;
    const float pb = p.post_shift ? p.post_shift[0] : 0.f;
    const bool post = p.post_scale || p.post_shift;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = i * 64 + lane;
        if (idx < nv) {
            float4 o = make_float4((v[i].x - mean) * rstd, (v[i].y - mean) * rstd, (v[i].z - mean) * rstd, (v[i].w - mean) * rstd);
            if (p.w) {
                const float4 w = *reinterpret_cast<const float4*>(p.w + idx * 4);
                o.x *= w.x; o.y *= w.y; o.z *= w.z; o.w *= w.w;
            }
            if (p.b) {
                const float4 b = *reinterpret_cast<const float4*>(p.b + idx * 4);
                o.x += b.x; o.y += b.y; o.z += b.z; o.w += b.w;
            }
            if (post) { o.x = o.x * ps + pb; o.y = o.y * ps + pb; o.z = o.z * ps + pb; o.w = o.w * ps + pb; }
            if (p.act == 1) { o.x = gelu_erf(o.x); o.y = gelu_erf(o.y); o.z = gelu_erf(o.z); o.w = gelu_erf(o.w); }
            store4(y + idx * 4, o);
        }
    }
}

// One workgroup per row (rows of >= 1024 elements): a lane holds D/1024 float4 instead of D/256, four times as many rows are in
// flight per CU and no wave waits on sixteen of its own loads before it can start reducing.  The statistics cross the four waves
// through LDS.  Same arithmetic as norm_kernel (the partial sums are combined in wave order).
template <typename TI, typename TO, int MAXV, int WPR = 4>  // WPR waves share a row; a workgroup holds 4 / WPR rows
__global__ __launch_bounds__(256) void norm_block_kernel(NormArgs p) {
    __shared__ float red[2][4];
    constexpr int TPR = WPR * 64;                 // threads per row
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int sub = wv / WPR, t = tid - sub * TPR;  // row slot inside the workgroup, thread index inside the row
    const long row = (long)blockIdx.x * (4 / WPR) + sub;
    const bool live = row < p.rows;
    const TI* x = reinterpret_cast<const TI*>(p.in) + (live ? row : 0) * p.in_stride;
    TO* y = reinterpret_cast<TO*>(p.out) + (live ? row : 0) * p.out_stride;
    const int nv = p.D >> 2;
    float4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = i * TPR + t;
        if (idx < nv && live) {
            v[i] = load4(x + idx * 4);
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        } else {
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    auto row_sum = [&](float part, int slot) -> float {  // sum over the WPR waves of this row
        part = wave_sum(part);
        if (lane == 0) red[slot][wv] = part;
        __syncthreads();
        float tot = red[slot][sub * WPR];
#pragma unroll
        for (int k = 1; k < WPR; ++k) tot += red[slot][sub * WPR + k];
        return tot;
    };
    float mean = 0.f;
    if (!p.rms) mean = row_sum(s, 0) / (float)p.D;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = i * TPR + t;
        if (idx < nv && live) {
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            ss += (a * a + b * b) + (c * c + d * d);
        }
    }
    const float var = row_sum(ss, 1) / (float)p.D;
    const float rstd = p.rms ? rsqrtf(var + p.eps) : 1.0f / sqrtf(var + p.eps);
    const float ps = p.post_scale ? p.post_scale[0] : 1.f;
    const float pb = p.post_shift ? p.post_shift[0] : 0.f;
    const bool post = p.post_scale || p.post_shift;
    if (!live) return;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = i * TPR + t;
        if (idx < nv) {
            float4 o = make_float4((v[i].x - mean) * rstd, (v[i].y - mean) * rstd, (v[i].z - mean) * rstd, (v[i].w - mean) * rstd);
            if (p.w) {
                const float4 w = *reinterpret_cast<const float4*>(p.w + idx * 4);
                o.x *= w.x; o.y *= w.y; o.z *= w.z; o.w *= w.w;
            }
            if (p.b) {
                const float4 b = *reinterpret_cast<const float4*>(p.b + idx * 4);
                o.x += b.x; o.y += b.y; o.z += b.z; o.w += b.w;
            }
            if (post) { o.x = o.x * ps + pb; o.y = o.y * ps + pb; o.z = o.z * ps + pb; o.w = o.w * ps + pb; }
            if (p.act == 1) { o.x = gelu_erf(o.x); o.y = gelu_erf(o.y); o.z = gelu_erf(o.z); o.w = gelu_erf(o.w); }
            store4(y + idx * 4, o);
        }
    }
}

// D <= 64 (LayerNorm2d over the 64 / 32 channels of the decoder's upscaling path, millions of rows): 16 lanes per row, four rows
// per wave, so every lane still moves 16 bytes; statistics reduce over the 16-lane group.
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void norm_narrow_kernel(NormArgs p) {
    const int lane = threadIdx.x & 63, sub = lane & 15;
    const long row = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
    const int nv = p.D >> 2;
    const bool on = row < p.rows && sub < nv;
    const TI* x = reinterpret_cast<const TI*>(p.in) + (on ? row : 0) * p.in_stride;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (on) v = load4(x + sub * 4);
    float s = (v.x + v.y) + (v.z + v.w);
    float mean = 0.f;
    if (!p.rms) {
        for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        mean = s / (float)p.D;
    }
    float ss = 0.f;
    if (on) {
        const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
        ss = (a * a + b * b) + (c * c + d * d);
    }
    for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float var = ss / (float)p.D;
    const float rstd = p.rms ? rsqrtf(var + p.eps) : 1.0f / sqrtf(var + p.eps);
    if (!on) return;
    float4 o = make_float4((v.x - mean) * rstd, (v.y - mean) * rstd, (v.z - mean) * rstd, (v.w - mean) * rstd);
    if (p.w) {
        const float4 w = *reinterpret_cast<const float4*>(p.w + sub * 4);
        o.x *= w.x; o.y *= w.y; o.z *= w.z; o.w *= w.w;
    }
    if (p.b) {
        const float4 b = *reinterpret_cast<const float4*>(p.b + sub * 4);
        o.x += b.x; o.y += b.y; o.z += b.z; o.w += b.w;
    }
    if (p.post_scale || p.post_shift) {
        const float ps = p.post_scale ? p.post_scale[0] : 1.f, pb = p.post_shift ? p.post_shift[0] : 0.f;
        o.x = o.x * ps + pb; o.y = o.y * ps + pb; o.z = o.z * ps + pb; o.w = o.w * ps + pb;
    }
    if (p.act == 1) { o.x = gelu_erf(o.x); o.y = gelu_erf(o.y); o.z = gelu_erf(o.z); o.w = gelu_erf(o.w); }
    store4(reinterpret_cast<TO*>(p.out) + row * p.out_stride + sub * 4, o);
}

static int g_norm_lds = 0;  // A/B switch: non-zero = one wave per row for every shape
extern "C" int ullsam_set_norm_variant(int v) { g_norm_lds = v; return 0; }

template <typename TI, typename TO>
static int launch_norm(const NormArgs& a, hipStream_t s) {
    const dim3 grid((unsigned)((a.rows + 3) / 4)), block(256);
    const int nv = a.D / 4;
    if (nv <= 16 && a.rows >= 4096) {
        norm_narrow_kernel<TI, TO><<<dim3((unsigned)((a.rows + 15) / 16)), block, 0, s>>>(a);
        ULLSAM_LAUNCH_CHECK();
        return 0;
    }
    // rows of >= 2048 elements, many of them (the LLM's RMSNorm): one workgroup per row -- [4324 x 4096] fp32 -> bf16 takes 26.7 us
    // (4.0 TB/s) against 37.9 us with a wave per row; at 1280 elements (ViT LayerNorm) the wave-per-row kernel stays ahead (33.6 vs 37.7)
    // ... and so does a handful of long rows (the decode step: 4 rows x 4096 took 14.6 us with one wave per row)
    if (g_norm_lds == 0 && (a.rows >= 1024 || a.rows <= 64) && nv >= 512) {
        const dim3 g((unsigned)a.rows);
        if (nv <= 512) norm_block_kernel<TI, TO, 2><<<g, block, 0, s>>>(a);
        else norm_block_kernel<TI, TO, 4><<<g, block, 0, s>>>(a);
        ULLSAM_LAUNCH_CHECK();
        return 0;
    }
    const size_t lds = 0;
    if (nv <= 64) norm_kernel<TI, TO, 1><<<grid, block, lds, s>>>(a);
    else if (nv <= 256) norm_kernel<TI, TO, 4><<<grid, block, lds, s>>>(a);
    else if (nv <= 512) norm_kernel<TI, TO, 8><<<grid, block, lds, s>>>(a);
    else norm_kernel<TI, TO, 16><<<grid, block, lds, s>>>(a);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// ---- LayerNorm with fan-out (norm4 of the two-way block, transformer.py:182, feeding the next block) -----------------
// y = LN(in) is needed as the fp32 residual stream, in the compute dtype (v projection / upscaling input) and as (y + pe) in the
// compute dtype (k / q projections): three consumers, one pass -- instead of the norm plus two add_cast passes over [P*N, C].
template <typename TO>
__global__ __launch_bounds__(256) void norm_fanout_kernel(const float* __restrict__ in, long rows, int D, const float* __restrict__ w,
                                                          const float* __restrict__ b, float eps, float* __restrict__ out_f32,
                                                          TO* __restrict__ out_c, TO* __restrict__ out_c_pe,
                                                          const float* __restrict__ pe, long pe_rows) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = D >> 2;
    const bool on = lane < nv;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (on) v = *reinterpret_cast<const float4*>(in + row * D + lane * 4);
    const float mean = wave_sum((v.x + v.y) + (v.z + v.w)) / (float)D;
    float ss = 0.f;
    if (on) {
        const float a = v.x - mean, bb = v.y - mean, c = v.z - mean, d = v.w - mean;
        ss = (a * a + bb * bb) + (c * c + d * d);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)D + eps);
    if (!on) return;
    float4 o = make_float4((v.x - mean) * rstd, (v.y - mean) * rstd, (v.z - mean) * rstd, (v.w - mean) * rstd);
    if (w) { const float4 t = *reinterpret_cast<const float4*>(w + lane * 4); o.x *= t.x; o.y *= t.y; o.z *= t.z; o.w *= t.w; }
    if (b) { const float4 t = *reinterpret_cast<const float4*>(b + lane * 4); o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w; }
    if (out_f32) *reinterpret_cast<float4*>(out_f32 + row * D + lane * 4) = o;
    if (out_c) store4(out_c + row * D + lane * 4, o);
    if (out_c_pe) {
        const float4 t = *reinterpret_cast<const float4*>(pe + (row % pe_rows) * D + lane * 4);
        store4(out_c_pe + row * D + lane * 4, make_float4(o.x + t.x, o.y + t.y, o.z + t.z, o.w + t.w));
    }
}

// in f32 [rows, D] (D <= 256, D % 4 == 0); out_f32 / out_c / out_c_pe each optional; c_dtype 0 f32, 1 bf16; pe f32 [pe_rows, D].
extern "C" int ullsam_norm_fanout(const float* in, long rows, int D, const float* w, const float* b, float eps, float* out_f32,
                                  void* out_c, void* out_c_pe, int c_dtype, const float* pe, long pe_rows, void* stream) {
    ULLSAM_CHECK(D % 4 == 0 && D > 0 && D <= 256, "norm_fanout: D=%d must be a multiple of 4 and <= 256", D);
    ULLSAM_CHECK(!out_c_pe || (pe && pe_rows > 0), "norm_fanout: out_c_pe needs pe");
    if (rows <= 0) return 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)((rows + 3) / 4));
    if (c_dtype == 0) norm_fanout_kernel<float><<<grid, 256, 0, s>>>(in, rows, D, w, b, eps, out_f32, (float*)out_c, (float*)out_c_pe, pe, pe_rows);
    else norm_fanout_kernel<bf16><<<grid, 256, 0, s>>>(in, rows, D, w, b, eps, out_f32, (bf16*)out_c, (bf16*)out_c_pe, pe, pe_rows);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// in_dtype/out_dtype: 0 f32, 1 bf16.  w/b may be null.  rms=1 -> RMSNorm (b ignored).
extern "C" int ullsam_norm(const void* in, int in_dtype, long in_stride, void* out, int out_dtype, long out_stride,
                           const float* w, const float* b, long rows, int D, float eps, int rms, int act,
                           const float* post_scale, const float* post_shift, void* stream) {
    ULLSAM_CHECK(D % 4 == 0 && D > 0 && D <= 4096, "ullsam_norm: D=%d must be a multiple of 4 and <= 4096", D);
    ULLSAM_CHECK(rows >= 0, "ullsam_norm: bad rows");
    if (rows == 0) return 0;
    ULLSAM_CHECK(in_stride % 4 == 0 && out_stride % 4 == 0, "ullsam_norm: strides must be multiples of 4 elements");
    NormArgs a{in, out, w, rms ? nullptr : b, rows, D, in_stride, out_stride, eps, rms, act, post_scale, post_shift};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (in_dtype == 0 && out_dtype == 0) return launch_norm<float, float>(a, s);
    if (in_dtype == 0 && out_dtype == 1) return launch_norm<float, bf16>(a, s);
    if (in_dtype == 1 && out_dtype == 1) return launch_norm<bf16, bf16>(a, s);
    if (in_dtype == 1 && out_dtype == 0) return launch_norm<bf16, float>(a, s);
    ULLSAM_CHECK(false, "ullsam_norm: bad dtypes %d %d", in_dtype, out_dtype);
}

// ---------------------------------------------------------------------------------------------------------------
// fp8 (OCP e4m3) row quantisation for the fp8 ViT path (gemm.hip: gemm256f8_kernel): optional LayerNorm (the ViT's norm1 / norm2,
// image_encoder.py:166,180) followed by a per-row scale = amax / 448 and a saturating round-to-nearest-even conversion.  One wave
// per row, the row stays in registers between the passes.  The same kernel without the norm quantises weight rows (per output
// channel scales), so activations and weights share one rounding.
// ---------------------------------------------------------------------------------------------------------------
template <typename TI, int MAXV>
__global__ __launch_bounds__(256) void rows_fp8_kernel(const TI* __restrict__ in, long in_stride, unsigned char* __restrict__ out, long out_stride,
                                                      float* __restrict__ scale, const float* __restrict__ w, const float* __restrict__ b, long rows, int D,
                                                      float eps, int do_norm) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const TI* x = in + row * in_stride;
    const int nv = D >> 2;
    float4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = i * 64 + lane;
        v[i] = idx < nv ? load4(x + idx * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    if (do_norm) {
        const float mean = wave_sum(s) / (float)D;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i)
            if (i * 64 + lane < nv) {
                const float a = v[i].x - mean, bb = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
                ss += (a * a + bb * bb) + (c * c + d * d);
            }
        const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)D + eps);
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int idx = i * 64 + lane;
            if (idx < nv) {
                const float4 g = *reinterpret_cast<const float4*>(w + idx * 4), be = *reinterpret_cast<const float4*>(b + idx * 4);
                v[i].x = (v[i].x - mean) * rstd * g.x + be.x; v[i].y = (v[i].y - mean) * rstd * g.y + be.y;
                v[i].z = (v[i].z - mean) * rstd * g.z + be.z; v[i].w = (v[i].w - mean) * rstd * g.w + be.w;
            }
        }
    }
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[i].x), fabsf(v[i].y)), fmaxf(fabsf(v[i].z), fabsf(v[i].w))));
    amax = wave_max(amax);
    const float sc = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;   // e4m3: largest finite value 448
    const float inv = 1.0f / sc;
    if (lane == 0) scale[row] = sc;
    unsigned int* y = reinterpret_cast<unsigned int*>(out + row * out_stride);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = i * 64 + lane;
        if (idx < nv) {
            const float q0 = fminf(fmaxf(v[i].x * inv, -448.f), 448.f), q1 = fminf(fmaxf(v[i].y * inv, -448.f), 448.f);
            const float q2 = fminf(fmaxf(v[i].z * inv, -448.f), 448.f), q3 = fminf(fmaxf(v[i].w * inv, -448.f), 448.f);
            int pk = __builtin_amdgcn_cvt_pk_fp8_f32(q0, q1, 0, false);
            pk = __builtin_amdgcn_cvt_pk_fp8_f32(q2, q3, pk, true);
            y[idx] = (unsigned int)pk;
        }
    }
}

// in: fp32 / bf16 [rows, D] (in_dtype 0 / 1, row stride in elements); out: e4m3 bytes [rows, out_stride]; scale fp32 [rows];
// w / b non-null: LayerNorm(eps) first (image_encoder.py:166,180), else plain quantisation (weights).
extern "C" int ullsam_rows_fp8(const void* in, int in_dtype, long in_stride, void* out, long out_stride, float* scale, const float* w,
                               const float* b, long rows, int D, float eps, void* stream) {
    ULLSAM_CHECK(D % 4 == 0 && D <= 8192 && rows >= 0, "ullsam_rows_fp8: D=%d must be a multiple of 4, <= 8192", D);
    ULLSAM_CHECK((w == nullptr) == (b == nullptr), "ullsam_rows_fp8: weight and bias go together");
    ULLSAM_CHECK(out_stride % 4 == 0 && ((uintptr_t)out & 3) == 0, "ullsam_rows_fp8: output rows must be 4-byte aligned");
    if (rows == 0) return 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)((rows + 3) / 4));
    const int nv = D / 4, maxv = (nv + 63) / 64;
    const int dn = w != nullptr;
#define ULLSAM_RF8(TI, MV) rows_fp8_kernel<TI, MV><<<grid, 256, 0, s>>>(reinterpret_cast<const TI*>(in), in_stride, reinterpret_cast<unsigned char*>(out), out_stride, scale, w, b, rows, D, eps, dn)
    if (in_dtype == ULLSAM_DT_F32) {
        if (maxv <= 4) ULLSAM_RF8(float, 4); else if (maxv <= 8) ULLSAM_RF8(float, 8); else if (maxv <= 16) ULLSAM_RF8(float, 16); else ULLSAM_RF8(float, 32);
    } else {
        if (maxv <= 4) ULLSAM_RF8(bf16, 4); else if (maxv <= 8) ULLSAM_RF8(bf16, 8); else if (maxv <= 16) ULLSAM_RF8(bf16, 16); else ULLSAM_RF8(bf16, 32);
    }
#undef ULLSAM_RF8
    ULLSAM_LAUNCH_CHECK();
    return 0;
}
