// HBM-bound data-movement kernels around the ViT encoder and the projector MLPs.
// All are pure gathers / casts: algorithmic bytes = bytes read + bytes written, 16-byte accesses per lane.
#include "common.h"

// ---- patch im2col (PatchEmbed conv k=s=16 as a GEMM, image_encoder.py:387-395) ---------------------------
// in : pixels f32 [B, C, Hs, Ws] (Hs,Ws <= S); optional per-channel (x-mean)/std and zero pad to S x S
//      (Sam.preprocess, sam.py:164-174: normalise THEN pad with zeros)
// out: T [B*g*g, C*p*p], column = c*p*p + ky*p + kx  (== conv weight [D, C, p, p] flattened)
template <typename T>
__global__ __launch_bounds__(256) void patch_im2col_kernel(const float* __restrict__ in, T* __restrict__ out, int B, int C,
                                                           int Hs, int Ws, int S, int p, const float* mean, const float* stdv) {
    const int g = S / p;
    const int kq = p / 4;  // float4 groups per patch row
    const long total = (long)B * g * g * C * p * kq;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        long t = i;
        const int k4 = t % kq; t /= kq;
        const int ky = t % p; t /= p;
        const int c = t % C; t /= C;
        const int gx = t % g; t /= g;
        const int gy = t % g; t /= g;
        const int b = (int)t;
        const int y = gy * p + ky, x = gx * p + k4 * 4;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float val = 0.f;
            if (y < Hs && x + e < Ws) {
                val = in[(((long)b * C + c) * Hs + y) * Ws + x + e];
                if (mean) val = (val - mean[c]) / stdv[c];
            }
            v[e] = val;
        }
        T* dst = out + (((long)b * g + gy) * g + gx) * ((long)C * p * p) + (long)c * p * p + ky * p + k4 * 4;
        store4(dst, make_float4(v[0], v[1], v[2], v[3]));
    }
}

extern "C" int ullsam_patch_im2col(int dtype, const float* pixels, void* out, int B, int C, int Hs, int Ws, int S, int patch,
                                   const float* mean, const float* stdv, void* stream) {
    ULLSAM_CHECK(patch % 4 == 0 && S % patch == 0 && Hs <= S && Ws <= S, "patch_im2col: bad geometry");
    const long total = (long)B * (S / patch) * (S / patch) * C * patch * (patch / 4);
    const int grid = (int)min((total + 255) / 256, (long)2048 * 8);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == 0) patch_im2col_kernel<float><<<grid, 256, 0, s>>>(pixels, (float*)out, B, C, Hs, Ws, S, patch, mean, stdv);
    else patch_im2col_kernel<bf16><<<grid, 256, 0, s>>>(pixels, (bf16*)out, B, C, Hs, Ws, S, patch, mean, stdv);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// ---- 3x3 im2col on NHWC (neck conv 3x3 pad 1, image_encoder.py:96-102) ------------------------------------
// in T [B,H,W,C] -> out T [B*H*W, 9*C], column = (ky*3+kx)*C + c (weight repacked to [Cout][ky][kx][Cin] on the host)
__global__ __launch_bounds__(256) void im2col3x3_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, int B, int H,
                                                        int W, int cq) {  // cq = 16-byte chunks per pixel
    const long total = (long)B * H * W * 9 * cq;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        long t = i;
        const int c = t % cq; t /= cq;
        const int tap = t % 9; t /= 9;
        const int x = t % W; t /= W;
        const int y = t % H; t /= H;
        const int b = (int)t;
        const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = in[(((long)b * H + yy) * W + xx) * cq + c];
        out[i] = v;
    }
}

extern "C" int ullsam_im2col3x3(int dtype, const void* in, void* out, int B, int H, int W, int C, void* stream) {
    const int esz = dtype == 0 ? 4 : 2;
    ULLSAM_CHECK((C * esz) % 16 == 0, "im2col3x3: C*elem must be a multiple of 16 bytes");
    const int cq = C * esz / 16;
    const long total = (long)B * H * W * 9 * cq;
    const int grid = (int)min((total + 255) / 256, (long)2048 * 8);
    im2col3x3_kernel<<<grid, 256, 0, reinterpret_cast<hipStream_t>(stream)>>>((const uint4*)in, (uint4*)out, B, H, W, cq);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// ---- out[r,:] = a[r % a_rows,:] (+ b[r % b_rows,:]) with dtype conversion ------------------------------------
// cast (b == null), keys + key_pe (transformer.py:165,181), src = image_embedding + dense (mask_decoder.py:127)
template <typename TA, typename TO>
__global__ __launch_bounds__(256) void add_cast_kernel(const TA* __restrict__ a, const float* __restrict__ b, TO* __restrict__ out,
                                                       long rows, int cols, long a_rows, long b_rows) {
    const int cq = cols / 4;
    const long total = rows * cq;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / cq;
        const int c = (int)(i - r * cq) * 4;
        float4 v = load4(a + (r % a_rows) * cols + c);
        if (b) {
            const float4 w = load4(b + (r % b_rows) * cols + c);
            v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        }
        store4(out + r * cols + c, v);
    }
}

extern "C" int ullsam_add_cast(const void* a, int a_dtype, long a_rows, const float* b, long b_rows, void* out, int out_dtype,
                               long rows, int cols, void* stream) {
    ULLSAM_CHECK(cols % 4 == 0, "add_cast: cols %% 4 != 0");
    if (rows == 0) return 0;
    const long total = rows * (cols / 4);
    const int grid = (int)min((total + 255) / 256, (long)2048 * 8);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (b_rows <= 0) b_rows = 1;
    if (a_dtype == 0 && out_dtype == 0) add_cast_kernel<float, float><<<grid, 256, 0, s>>>((const float*)a, b, (float*)out, rows, cols, a_rows, b_rows);
    else if (a_dtype == 0 && out_dtype == 1) add_cast_kernel<float, bf16><<<grid, 256, 0, s>>>((const float*)a, b, (bf16*)out, rows, cols, a_rows, b_rows);
    else if (a_dtype == 1 && out_dtype == 0) add_cast_kernel<bf16, float><<<grid, 256, 0, s>>>((const bf16*)a, b, (float*)out, rows, cols, a_rows, b_rows);
    else add_cast_kernel<bf16, bf16><<<grid, 256, 0, s>>>((const bf16*)a, b, (bf16*)out, rows, cols, a_rows, b_rows);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// ---- [B, R, C] <-> [B, C, R] fp32 transpose through a padded LDS tile (NCHW <-> NHWC at the API boundary) ------
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int C) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* ip = in + (long)b * R * C;
    float* op = out + (long)b * R * C;
    for (int j = ty; j < 32; j += 8)
        if (r0 + j < R && c0 + tx < C) tile[j][tx] = ip[(long)(r0 + j) * C + c0 + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (c0 + j < C && r0 + tx < R) op[(long)(c0 + j) * R + r0 + tx] = tile[tx][j];
}

// in [B, R, C] -> out [B, C, R]
extern "C" int ullsam_transpose_f32(const float* in, float* out, int B, int R, int C, void* stream) {
    if (B == 0) return 0;
    transpose_kernel<<<dim3((C + 31) / 32, (R + 31) / 32, B), 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(in, out, R, C);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// ---- [R, C] fp32 or bf16 -> [C, Rp] bf16, transposed, columns R .. Rp - 1 zero: the operands of the training step's dW = dY^T X GEMM (inner dimension =
// the step's row count, padded to the GEMM's K granularity) and the W^T of dX = dY W, cast and transposed in one pass -------------------------------
template <typename Tin>
__global__ __launch_bounds__(256) void transpose_to_bf16_kernel(const Tin* __restrict__ in, bf16* __restrict__ out, int R, int C, int Rp) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int r = r0 + ty + 4 * j, c = c0 + tx;
        tile[ty + 4 * j][tx] = (r < R && c < C) ? to_f32(in[(long)r * C + c]) : 0.f;
    }
    __syncthreads();
    if ((Rp & 3) == 0) {   // output rows start on 8-byte boundaries: a lane writes four consecutive r as one 8-byte store (16 lanes = 128 contiguous bytes of an output row;
        //                    the 2-byte stores of the form below were this kernel's time: 23 -> ~13 us on the training step's 4096 x 1280 ... 5120 operands)
        const int q = threadIdx.x & 15, cy = threadIdx.x >> 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + cy + 16 * j, r = r0 + 4 * q;
            if (c < C && r < Rp) {
                bf16x4_t v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = from_f32<bf16>(tile[4 * q + e][cy + 16 * j]);   // (bank (4 q + e + cy + 16 j) mod 64: the 64 lanes of a wave hit 64 banks)
                *reinterpret_cast<bf16x4_t*>(out + (long)c * Rp + r) = v;
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int c = c0 + ty + 4 * j, r = r0 + tx;
        if (c < C && r < Rp) out[(long)c * Rp + r] = from_f32<bf16>(tile[tx][ty + 4 * j]);
    }
}
// ---- fp32 [R, C] -> bf16 [R, C] (row-major copy, optional) AND bf16 [C, Rp] (transposed, zero columns behind R) AND per-64-row-block column sums (optional) in ONE pass over the input:
// what the training step's bf16 Linear needs of an activation (forward: x for the GEMM, x^T for the later dW) and of a gradient (backward: dY for dX, dY^T for dW, sum_rows dY for db),
// which were a cast, a transpose and a column-sum launch reading the same fp32 tensor three times.  C % 4 == 0, Rp % 4 == 0; colsum: [ceil(R / 64)][C] floats, added in order afterwards.
__global__ __launch_bounds__(256) void cast_transpose_bf16_kernel(const float* __restrict__ in, bf16* __restrict__ out_rm, bf16* __restrict__ out_t, float* __restrict__ colsum,
                                                                  int R, int C, int Rp) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int cq = threadIdx.x & 15, ry = threadIdx.x >> 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = r0 + ry + 16 * j, c = c0 + 4 * cq;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < R && c < C) {                                // (C % 4 == 0: the quad is inside the row)
            v = *reinterpret_cast<const float4*>(in + (long)r * C + c);
            if (out_rm) {
                bf16x4_t o; o[0] = from_f32<bf16>(v.x); o[1] = from_f32<bf16>(v.y); o[2] = from_f32<bf16>(v.z); o[3] = from_f32<bf16>(v.w);
                *reinterpret_cast<bf16x4_t*>(out_rm + (long)r * C + c) = o;
            }
        }
        float* t = &tile[ry + 16 * j][4 * cq];
        t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
    }
    __syncthreads();
    if (colsum && threadIdx.x < 64 && c0 + (int)threadIdx.x < C) {   // rows r0 .. r0 + 63 of column c0 + t, in row order (rows past R hold zeros)
        float s = 0.f;
#pragma unroll 8
        for (int r = 0; r < 64; ++r) s += tile[r][threadIdx.x];
        colsum[(long)blockIdx.y * C + c0 + threadIdx.x] = s;
    }
    const int q = threadIdx.x & 15, cy = threadIdx.x >> 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = c0 + cy + 16 * j, r = r0 + 4 * q;
        if (c < C && r < Rp) {
            bf16x4_t v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = from_f32<bf16>(tile[4 * q + e][cy + 16 * j]);
            *reinterpret_cast<bf16x4_t*>(out_t + (long)c * Rp + r) = v;
        }
    }
}
__global__ __launch_bounds__(256) void colsum_blocks_reduce_kernel(const float* __restrict__ partial, float* __restrict__ out, int C, int nb) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int b = 0; b < nb; ++b) s += partial[(long)b * C + c];
    out[c] = s;
}
// out_rm bf16 [R, C] | NULL; out_t bf16 [C, Rp]; colsum_out fp32 [C] | NULL with colsum_ws fp32 [ceil(Rp / 64) * C]
extern "C" int ullsam_cast_transpose_bf16(const float* in, void* out_rm, void* out_t, float* colsum_out, float* colsum_ws, int R, int C, int Rp, void* stream) {
    ULLSAM_CHECK(R > 0 && C > 0 && Rp >= R && (C & 3) == 0 && (Rp & 3) == 0 && out_t, "cast_transpose_bf16: R=%d C=%d Rp=%d (C, Rp multiples of 4)", R, C, Rp);
    ULLSAM_CHECK((((uintptr_t)in | (uintptr_t)out_rm | (uintptr_t)out_t) & 15) == 0 && (!colsum_out || colsum_ws), "cast_transpose_bf16: 16-byte aligned operands; column sums need their workspace");
    const dim3 grid((C + 63) / 64, (Rp + 63) / 64);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    cast_transpose_bf16_kernel<<<grid, 256, 0, st>>>(in, static_cast<bf16*>(out_rm), static_cast<bf16*>(out_t), colsum_out ? colsum_ws : nullptr, R, C, Rp);
    ULLSAM_LAUNCH_CHECK();
    if (colsum_out) {
        colsum_blocks_reduce_kernel<<<dim3((C + 255) / 256), 256, 0, st>>>(colsum_ws, colsum_out, C, (int)grid.y);
        ULLSAM_LAUNCH_CHECK();
    }
    return 0;
}
extern "C" int ullsam_transpose_to_bf16(int in_dtype, const void* in, void* out, int R, int C, int Rp, void* stream) {
    ULLSAM_CHECK(R > 0 && C > 0 && Rp >= R, "transpose_to_bf16: R=%d C=%d Rp=%d", R, C, Rp);
    const dim3 grid((C + 63) / 64, (Rp + 63) / 64);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (in_dtype == ULLSAM_DT_F32) transpose_to_bf16_kernel<float><<<grid, 256, 0, st>>>(static_cast<const float*>(in), static_cast<bf16*>(out), R, C, Rp);
    else transpose_to_bf16_kernel<bf16><<<grid, 256, 0, st>>>(static_cast<const bf16*>(in), static_cast<bf16*>(out), R, C, Rp);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// ---- pixel_shuffle(0.5, v2) + LayerNorm(4C)  (modeling_internvl_sam.py:226-251, mlp1[0] :89) -------------------
// in : image embedding NHWC f32 [B, H, W, C];  out: T [B*(H/2)*(W/2), 4C]
//      out[(h2,w2), (h&1)*2C + (w&1)*C + c] = in[2h2+(h&1), 2w2+(w&1), c]; one wave per output token
template <typename T>
__global__ __launch_bounds__(256) void pixel_shuffle_ln_kernel(const float* __restrict__ in, T* __restrict__ out, const float* w,
                                                               const float* bta, int B, int H, int W, float eps) {
    constexpr int C = 256;  // SAM out_chans / sam_hidden_size (modeling_internvl_sam.py:84): one float4 per lane per segment
    const int lane = threadIdx.x & 63;
    const long tok = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int H2 = H / 2, W2 = W / 2;
    if (tok >= (long)B * H2 * W2) return;
    const int w2 = tok % W2, h2 = (tok / W2) % H2, b = (int)(tok / ((long)W2 * H2));
    float4 v[4];
    float s = 0.f;
#pragma unroll
    for (int seg = 0; seg < 4; ++seg) {
        v[seg] = *reinterpret_cast<const float4*>(in + ((((long)b * H + 2 * h2 + (seg >> 1)) * W) + 2 * w2 + (seg & 1)) * C + lane * 4);
        s += (v[seg].x + v[seg].y) + (v[seg].z + v[seg].w);
    }
    constexpr int D = 4 * C;
    const float mean = wave_sum(s) / (float)D;
    float ss = 0.f;
#pragma unroll
    for (int seg = 0; seg < 4; ++seg) {
        const float a = v[seg].x - mean, bb = v[seg].y - mean, c = v[seg].z - mean, d = v[seg].w - mean;
        ss += (a * a + bb * bb) + (c * c + d * d);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)D + eps);
    T* dst = out + tok * D;
#pragma unroll
    for (int seg = 0; seg < 4; ++seg) {
        const int col = seg * C + lane * 4;
        const float4 ww = *reinterpret_cast<const float4*>(w + col);
        const float4 bv = *reinterpret_cast<const float4*>(bta + col);
        store4(dst + col, make_float4((v[seg].x - mean) * rstd * ww.x + bv.x, (v[seg].y - mean) * rstd * ww.y + bv.y,
                                      (v[seg].z - mean) * rstd * ww.z + bv.z, (v[seg].w - mean) * rstd * ww.w + bv.w));
    }
}

extern "C" int ullsam_pixel_shuffle_ln(int dtype, const float* in_nhwc, void* out, const float* w, const float* b, int B, int H,
                                       int W, int C, float eps, void* stream) {
    ULLSAM_CHECK(C == 256 && H % 2 == 0 && W % 2 == 0, "pixel_shuffle_ln: needs C == 256 and even H, W");
    const long toks = (long)B * (H / 2) * (W / 2);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int grid = (int)((toks + 3) / 4);
    if (dtype == 0) pixel_shuffle_ln_kernel<float><<<grid, 256, 0, s>>>(in_nhwc, (float*)out, w, b, B, H, W, eps);
    else pixel_shuffle_ln_kernel<bf16><<<grid, 256, 0, s>>>(in_nhwc, (bf16*)out, w, b, B, H, W, eps);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// ---- inverse pixel shuffle (text_aware_dense_feature, modeling_internvl_sam.py:256-268) -------------------------
// in f32 [B, (H/2)*(W/2), 4C] -> out NHWC f32 [B, H, W, C]:  out[Y, X, c] = in[(Y/2)*(W/2) + X/2, (Y&1)*2C + (X&1)*C + c]
__global__ __launch_bounds__(256) void pixel_unshuffle_kernel(const float4* __restrict__ in, float4* __restrict__ out, int B, int H,
                                                              int W, int cq) {
    const long total = (long)B * H * W * cq;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        long t = i;
        const int c = t % cq; t /= cq;
        const int X = t % W; t /= W;
        const int Y = t % H; t /= H;
        const int b = (int)t;
        const long tok = ((long)b * (H / 2) + Y / 2) * (W / 2) + X / 2;
        out[i] = in[tok * 4 * cq + ((Y & 1) * 2 + (X & 1)) * cq + c];
    }
}

extern "C" int ullsam_pixel_unshuffle(const float* in, float* out_nhwc, int B, int H, int W, int C, void* stream) {
    ULLSAM_CHECK(C % 4 == 0 && H % 2 == 0 && W % 2 == 0, "pixel_unshuffle: bad geometry");
    const long total = (long)B * H * W * (C / 4);
    const int grid = (int)min((total + 255) / 256, (long)2048 * 8);
    pixel_unshuffle_kernel<<<grid, 256, 0, reinterpret_cast<hipStream_t>(stream)>>>((const float4*)in, (float4*)out_nhwc, B, H, W, C / 4);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}
