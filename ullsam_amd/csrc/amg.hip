// Automatic-mask-generation helpers (reference: utils/amg.py).  Integer / byte work, HBM-bound: every kernel streams the
// mask stack once with coalesced accesses; results are bit-exact with the reference functions.
#include "common.h"

// ---- calculate_stability_score (amg.py:156-176): per mask, #(x > thr+off) / #(x > thr-off) --------------------------
__global__ __launch_bounds__(256) void stability_count_kernel(const float* __restrict__ m, long per, float hi, float lo,
                                                              unsigned int* __restrict__ counts) {
    const long n = blockIdx.y;
    const float4* p = reinterpret_cast<const float4*>(m + n * per);
    const long nq = per >> 2;
    unsigned int a = 0, b = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nq; i += (long)gridDim.x * 256) {
        const float4 v = p[i];
        a += (v.x > hi) + (v.y > hi) + (v.z > hi) + (v.w > hi);
        b += (v.x > lo) + (v.y > lo) + (v.z > lo) + (v.w > lo);
    }
    if (blockIdx.x == 0)
        for (long i = (nq << 2) + threadIdx.x; i < per; i += 256) { a += m[n * per + i] > hi; b += m[n * per + i] > lo; }
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&counts[2 * n], a); atomicAdd(&counts[2 * n + 1], b); }
}
__global__ void stability_finish_kernel(const unsigned int* __restrict__ counts, float* __restrict__ score, long N) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < N) score[i] = (float)(int)counts[2 * i] / (float)(int)counts[2 * i + 1];  // int32 / int32 true division, as torch
}

// masks fp32 [N, per]; counts u32 [N,2] scratch (zeroed here); score fp32 [N]
extern "C" int ullsam_stability_score(const float* masks, long N, long per, float mask_threshold, float threshold_offset,
                                      unsigned int* counts, float* score, void* stream) {
    if (N == 0) return 0;
    ULLSAM_CHECK(((uintptr_t)masks & 15) == 0 && per % 4 == 0, "stability_score: masks must be 16-byte aligned, H*W %% 4 == 0");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (hipMemsetAsync(counts, 0, sizeof(unsigned int) * 2 * N, s) != hipSuccess) { ullsam_set_error("stability_score: memset failed"); return -2; }
    const int bx = (int)min((per / 4 + 255) / 256, (long)64);
    stability_count_kernel<<<dim3(bx, (unsigned)N), 256, 0, s>>>(masks, per, mask_threshold + threshold_offset, mask_threshold - threshold_offset, counts);
    ULLSAM_LAUNCH_CHECK();
    stability_finish_kernel<<<(unsigned)((N + 255) / 256), 256, 0, s>>>(counts, score, N);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// ---- batched_mask_to_box (amg.py:303-346): XYXY with inclusive maxima; empty mask -> 0,0,0,0 -------------------------
// Streaming pass: 16-byte chunks (one row segment each when W % 16 == 0), (chunks, N) grid, wave reduce, then integer
// atomics on the output box itself ({INT_MAX,INT_MAX,-1,-1} initialised), finalised by a tiny kernel.
__global__ void box_init_kernel(int* __restrict__ out, long N) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < N) reinterpret_cast<int4*>(out)[i] = make_int4(0x7fffffff, 0x7fffffff, -1, -1);
}
__global__ void box_finish_kernel(int* __restrict__ out, long N) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int4 b = reinterpret_cast<int4*>(out)[i];
    if (b.z < b.x || b.w < b.y) reinterpret_cast<int4*>(out)[i] = make_int4(0, 0, 0, 0);
}
template <int VEC>
__global__ __launch_bounds__(256) void mask_to_box_kernel(const unsigned char* __restrict__ m, int H, int W, int* __restrict__ out) {
    const long n = blockIdx.y;
    const long per = (long)H * W;
    const unsigned char* p = m + n * per;
    int x0 = 0x7fffffff, y0 = 0x7fffffff, x1 = -1, y1 = -1;
    if (VEC == 16) {
        const long nchunk = per >> 4;
        const int Wc = W >> 4;
        const uint4* q = reinterpret_cast<const uint4*>(p);
        for (long c = (long)blockIdx.x * 256 + threadIdx.x; c < nchunk; c += (long)gridDim.x * 256) {
            const uint4 v = q[c];
            if (v.x | v.y | v.z | v.w) {
                const int y = (int)(c / Wc), xb = (int)(c - (long)y * Wc) << 4;
                const int lo = v.x ? (__builtin_ctz(v.x) >> 3) : v.y ? 4 + (__builtin_ctz(v.y) >> 3) : v.z ? 8 + (__builtin_ctz(v.z) >> 3) : 12 + (__builtin_ctz(v.w) >> 3);
                const int hi = v.w ? 12 + ((31 - __builtin_clz(v.w)) >> 3) : v.z ? 8 + ((31 - __builtin_clz(v.z)) >> 3) : v.y ? 4 + ((31 - __builtin_clz(v.y)) >> 3) : ((31 - __builtin_clz(v.x)) >> 3);
                x0 = min(x0, xb + lo); x1 = max(x1, xb + hi); y0 = min(y0, y); y1 = max(y1, y);
            }
        }
    } else {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < per; i += (long)gridDim.x * 256) {
            if (p[i]) {
                const int y = (int)(i / W), x = (int)(i - (long)y * W);
                x0 = min(x0, x); x1 = max(x1, x); y0 = min(y0, y); y1 = max(y1, y);
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        x0 = min(x0, __shfl_xor(x0, o, 64)); y0 = min(y0, __shfl_xor(y0, o, 64));
        x1 = max(x1, __shfl_xor(x1, o, 64)); y1 = max(y1, __shfl_xor(y1, o, 64));
    }
    if ((threadIdx.x & 63) == 0 && x1 >= 0) {
        atomicMin(&out[4 * n + 0], x0); atomicMin(&out[4 * n + 1], y0);
        atomicMax(&out[4 * n + 2], x1); atomicMax(&out[4 * n + 3], y1);
    }
}

extern "C" int ullsam_mask_to_box(const unsigned char* masks, long N, int H, int W, int* boxes, void* stream) {
    if (N == 0) return 0;
    ULLSAM_CHECK(((uintptr_t)boxes & 15) == 0, "mask_to_box: boxes must be 16-byte aligned");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long per = (long)H * W;
    box_init_kernel<<<(unsigned)((N + 255) / 256), 256, 0, s>>>(boxes, N);
    ULLSAM_LAUNCH_CHECK();
    if (per > 0) {
        const bool vec = (W % 16 == 0) && (((uintptr_t)masks & 15) == 0);
        const long units = vec ? per / 16 : per;
        const unsigned bx = (unsigned)max(1L, min((units + 256 * 8 - 1) / (256 * 8), 256L));
        if (vec) mask_to_box_kernel<16><<<dim3(bx, (unsigned)N), 256, 0, s>>>(masks, H, W, boxes);
        else mask_to_box_kernel<1><<<dim3(bx, (unsigned)N), 256, 0, s>>>(masks, H, W, boxes);
        ULLSAM_LAUNCH_CHECK();
    }
    box_finish_kernel<<<(unsigned)((N + 255) / 256), 256, 0, s>>>(boxes, N);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// ---- mask_to_rle_pytorch (amg.py:107-135): change positions of the column-major flattened mask ----------------------
// The reference transposes the mask and diffs neighbours of the Fortran-order flattening f = x*H + y.  Here the mask is read once,
// row-major and coalesced: a wave owns 64 rows x (64*VEC) columns, every lane accumulates for its VEC columns a 64-bit word of
// the rows' bits (bit i = mask[64*yb+i][x] != 0), and turns it into a *change word* (bit i set iff t[f] != t[f+1] for
// f = x*H + 64*yb + i), taking the bit after the word from the next row block or, at the column end, from the top of the next
// column.  Change words are stored as words[n][yb][x] (coalesced) and their popcounts summed per mask; `rle_emit_kernel` then
// walks a mask's words in (x, yb) order with a block scan and writes the change positions, ordered, at the mask's offset.
template <int VEC>
__global__ __launch_bounds__(256) void rle_pack_kernel(const unsigned char* __restrict__ m, int H, int W, int nyb,
                                                       unsigned long long* __restrict__ words, int* __restrict__ counts,
                                                       unsigned char* __restrict__ first) {
    const long n = blockIdx.z;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int yb = blockIdx.y * 4 + wv;
    if (yb >= nyb) return;
    const unsigned char* p = m + n * (long)H * W;
    const int xa = (blockIdx.x * 64 + lane) * VEC;  // first column of this lane
    const int ys = yb * 64;
    const int r = min(64, H - ys);                  // rows in this block (wave-uniform)
    if (blockIdx.x == 0 && yb == 0 && lane == 0) first[n] = p[0] ? 1 : 0;
    unsigned long long w[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) w[c] = 0;
    unsigned int nb = 0;  // bit c: the bit that follows the word of column xa+c in Fortran order
    bool has_next[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) has_next[c] = true;
    if (xa < W) {
        if (VEC == 16) {
            for (int j = 0; j < 8; ++j) {
                unsigned int acc[4] = {0, 0, 0, 0};  // byte k of acc[d]: 8 rows' bits of column 4d+k
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int row = 8 * j + i;
                    if (row < r) {
                        const uint4 v = *reinterpret_cast<const uint4*>(p + (long)(ys + row) * W + xa);
                        const unsigned int d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            unsigned int x = d[k];
                            x |= x >> 4; x |= x >> 2; x |= x >> 1;  // any bit of a byte -> its bit 0
                            acc[k] |= (x & 0x01010101u) << i;
                        }
                    }
                }
#pragma unroll
                for (int c = 0; c < 16; ++c) w[c] |= (unsigned long long)((acc[c >> 2] >> (8 * (c & 3))) & 0xffu) << (8 * j);
            }
            if (ys + r < H) {
                const uint4 v = *reinterpret_cast<const uint4*>(p + (long)(ys + r) * W + xa);
                const unsigned int d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int c = 0; c < 16; ++c) nb |= (unsigned int)(((d[c >> 2] >> (8 * (c & 3))) & 0xffu) != 0) << c;
            } else {
                const uint4 v = *reinterpret_cast<const uint4*>(p + xa);  // row 0: the top of columns xa .. xa+15
                const unsigned int d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int c = 0; c < 15; ++c) nb |= (unsigned int)(((d[(c + 1) >> 2] >> (8 * ((c + 1) & 3))) & 0xffu) != 0) << c;
                if (xa + 16 < W) nb |= (unsigned int)(p[xa + 16] != 0) << 15;
                else has_next[15] = false;  // last element of the flattening
            }
        } else {
            for (int row = 0; row < r; ++row) w[0] |= (unsigned long long)(p[(long)(ys + row) * W + xa] != 0) << row;
            if (ys + r < H) nb = p[(long)(ys + r) * W + xa] != 0;
            else if (xa + 1 < W) nb = p[xa + 1] != 0;
            else has_next[0] = false;
        }
    }
    const unsigned long long keep = r == 64 ? ~0ull : ((1ull << r) - 1ull);
    int cnt = 0;
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
        unsigned long long ch = (w[c] ^ ((w[c] >> 1) | ((unsigned long long)((nb >> c) & 1u) << (r - 1)))) & keep;
        if (!has_next[c]) ch &= ~(1ull << (r - 1));
        w[c] = ch;
        cnt += __popcll(ch);
    }
    if (xa < W) {
        unsigned long long* dst = words + ((long)n * nyb + yb) * W + xa;
        if (VEC == 16) {
#pragma unroll
            for (int c = 0; c < 16; c += 2) *reinterpret_cast<ulonglong2*>(dst + c) = make_ulonglong2(w[c], w[c + 1]);
        } else {
            dst[0] = w[0];
        }
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if (lane == 0 && cnt) atomicAdd(&counts[n], cnt);
}

// One workgroup per mask: ordered emission of the change positions from the change words (see rle_pack_kernel).
__global__ __launch_bounds__(256) void rle_emit_kernel(const unsigned long long* __restrict__ words, int H, int W, int nyb,
                                                       const long* __restrict__ offsets, int* __restrict__ out) {
    __shared__ int wsum[2][4];
    const long n = blockIdx.x;
    const unsigned long long* wp = words + n * (long)nyb * W;
    int* dst = out + offsets[n];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long total = (long)W * nyb;
    int base = 0, par = 0;
    for (long c0 = 0; c0 < total; c0 += 1024, par ^= 1) {
        unsigned long long wq[4];
        int fq[4];
        int cnt = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long idx = c0 + tid * 4 + q;  // (x, yb) order
            wq[q] = 0; fq[q] = 0;
            if (idx < total) {
                const int x = (int)(idx / nyb), yb = (int)(idx - (long)x * nyb);
                wq[q] = wp[(long)yb * W + x];
                fq[q] = x * H + yb * 64;
            }
            cnt += __popcll(wq[q]);
        }
        int inc = cnt;
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        if (lane == 63) wsum[par][wv] = inc;
        __syncthreads();
        int wbase = 0;
        for (int k = 0; k < wv; ++k) wbase += wsum[par][k];
        const int tot = wsum[par][0] + wsum[par][1] + wsum[par][2] + wsum[par][3];
        if (cnt) {
            int pos = base + wbase + inc - cnt;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned long long w = wq[q];
                while (w) { dst[pos++] = fq[q] + __builtin_ctzll(w); w &= w - 1; }
            }
        }
        base += tot;
    }
}

// masks u8 [N,H,W] (any non-zero = set); words u64 [N, ceil(H/64), W] scratch; counts int [N] (zeroed here) = number of
// change positions per mask; first u8 [N] = mask[0][0] != 0.
extern "C" int ullsam_rle_pack(const unsigned char* masks, long N, int H, int W, unsigned long long* words, int* counts,
                               unsigned char* first, void* stream) {
    if (N == 0) return 0;
    ULLSAM_CHECK(H > 0 && W > 0 && (long)H * W < (1L << 31), "rle_pack: need 0 < H*W < 2^31");
    ULLSAM_CHECK(((uintptr_t)words & 15) == 0, "rle_pack: words must be 16-byte aligned");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (hipMemsetAsync(counts, 0, sizeof(int) * N, s) != hipSuccess) { ullsam_set_error("rle_pack: memset failed"); return -2; }
    const int nyb = (H + 63) / 64;
    const bool vec = (W % 16 == 0) && (((uintptr_t)masks & 15) == 0);
    const int cols = vec ? 1024 : 64;
    const dim3 grid((unsigned)((W + cols - 1) / cols), (unsigned)((nyb + 3) / 4), (unsigned)N);
    if (vec) rle_pack_kernel<16><<<grid, 256, 0, s>>>(masks, H, W, nyb, words, counts, first);
    else rle_pack_kernel<1><<<grid, 256, 0, s>>>(masks, H, W, nyb, words, counts, first);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}
// offsets i64 [N]: exclusive prefix of counts; out int32 [sum(counts)]: change positions f (t[f] != t[f+1]), ordered per mask.
extern "C" int ullsam_rle_emit(const unsigned long long* words, long N, int H, int W, const long* offsets, int* out, void* stream) {
    if (N == 0) return 0;
    rle_emit_kernel<<<(unsigned)N, 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(words, H, W, (H + 63) / 64, offsets, out);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// ---- box NMS suppression matrix (torchvision.ops.nms semantics): bit j of mask[i][w] = IoU(box i, box 64w+j) > thr -----
// boxes fp32 [N,4] XYXY, already in decreasing-score order; only j > i is filled.  The greedy scan runs on the host.
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ boxes, int N, float thr, unsigned long long* __restrict__ mask) {
    __shared__ float cb[64 * 4];
    const int rb = blockIdx.y, cbk = blockIdx.x;
    const int nw = (N + 63) / 64;
    if (cbk < rb) return;
    const int ncol = min(64, N - cbk * 64);
    if ((int)threadIdx.x < ncol)
        for (int k = 0; k < 4; ++k) cb[threadIdx.x * 4 + k] = boxes[(long)(cbk * 64 + threadIdx.x) * 4 + k];
    __syncthreads();
    const int i = rb * 64 + threadIdx.x;
    if (i >= N) return;
    const float x1 = boxes[(long)i * 4], y1 = boxes[(long)i * 4 + 1], x2 = boxes[(long)i * 4 + 2], y2 = boxes[(long)i * 4 + 3];
    const float ai = (x2 - x1) * (y2 - y1);
    unsigned long long t = 0;
    const int start = (rb == cbk) ? threadIdx.x + 1 : 0;
    for (int j = start; j < ncol; ++j) {
        const float bx1 = cb[j * 4], by1 = cb[j * 4 + 1], bx2 = cb[j * 4 + 2], by2 = cb[j * 4 + 3];
        const float w = fmaxf(fminf(x2, bx2) - fmaxf(x1, bx1), 0.f), h = fmaxf(fminf(y2, by2) - fmaxf(y1, by1), 0.f);
        const float inter = w * h;
        const float iou = inter / (ai + (bx2 - bx1) * (by2 - by1) - inter);
        if (iou > thr) t |= 1ull << j;
    }
    mask[(long)i * nw + cbk] = t;
}

extern "C" int ullsam_nms_mask(const float* boxes, int N, float iou_threshold, unsigned long long* mask, void* stream) {
    if (N == 0) return 0;
    const int nw = (N + 63) / 64;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (hipMemsetAsync(mask, 0, sizeof(unsigned long long) * (size_t)N * nw, s) != hipSuccess) { ullsam_set_error("nms_mask: memset failed"); return -2; }
    nms_mask_kernel<<<dim3(nw, nw), 64, 0, s>>>(boxes, N, iou_threshold, mask);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// ---- threshold fp32 logits -> u8 mask (masks > mask_threshold) ------------------------------------------------------
__global__ __launch_bounds__(256) void threshold_kernel(const float4* __restrict__ in, uchar4* __restrict__ out, long nq, float thr) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nq; i += (long)gridDim.x * 256) {
        const float4 v = in[i];
        out[i] = make_uchar4(v.x > thr, v.y > thr, v.z > thr, v.w > thr);
    }
}
extern "C" int ullsam_threshold_u8(const float* in, unsigned char* out, long n, float thr, void* stream) {
    ULLSAM_CHECK(n % 4 == 0, "threshold_u8: n %% 4 != 0");
    if (n == 0) return 0;
    const int grid = (int)min((n / 4 + 255) / 256, (long)2048 * 8);
    threshold_kernel<<<grid, 256, 0, reinterpret_cast<hipStream_t>(stream)>>>((const float4*)in, (uchar4*)out, n / 4, thr);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}
