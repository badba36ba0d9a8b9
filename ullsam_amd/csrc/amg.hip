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
__global__ __launch_bounds__(256) void mask_to_box_kernel(const unsigned char* __restrict__ m, int H, int W, int* __restrict__ out) {
    __shared__ int red[4][4];
    const long n = blockIdx.x;
    const unsigned char* p = m + n * (long)H * W;
    int x0 = W, y0 = H, x1 = -1, y1 = -1;
    const long per = (long)H * W;
    for (long i = threadIdx.x; i < per; i += 256) {
        if (p[i]) {
            const int y = (int)(i / W), x = (int)(i - (long)y * W);
            x0 = min(x0, x); x1 = max(x1, x); y0 = min(y0, y); y1 = max(y1, y);
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        x0 = min(x0, __shfl_xor(x0, o, 64)); y0 = min(y0, __shfl_xor(y0, o, 64));
        x1 = max(x1, __shfl_xor(x1, o, 64)); y1 = max(y1, __shfl_xor(y1, o, 64));
    }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[wv][0] = x0; red[wv][1] = y0; red[wv][2] = x1; red[wv][3] = y1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) { x0 = min(x0, red[k][0]); y0 = min(y0, red[k][1]); x1 = max(x1, red[k][2]); y1 = max(y1, red[k][3]); }
        const bool empty = x1 < x0 || y1 < y0;
        out[4 * n + 0] = empty ? 0 : x0; out[4 * n + 1] = empty ? 0 : y0;
        out[4 * n + 2] = empty ? 0 : x1; out[4 * n + 3] = empty ? 0 : y1;
    }
}

extern "C" int ullsam_mask_to_box(const unsigned char* masks, long N, int H, int W, int* boxes, void* stream) {
    if (N == 0) return 0;
    mask_to_box_kernel<<<(unsigned)N, 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(masks, H, W, boxes);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// ---- mask_to_rle_pytorch (amg.py:107-135): change positions of the column-major flattened mask ----------------------
// One workgroup per mask walks the mask in Fortran order (f = x*H + y) in 4096-element chunks; `write == 0` only counts
// the changes (counts[n]); `write == 1` stores their positions f (where t[f] != t[f+1]) at out[offsets[n] ...] in order.
__global__ __launch_bounds__(256) void rle_changes_kernel(const unsigned char* __restrict__ m, int H, int W, int write,
                                                          int* __restrict__ counts, const long* __restrict__ offsets,
                                                          int* __restrict__ out, unsigned char* __restrict__ first) {
    __shared__ int wsum[4];
    __shared__ int base_s;
    const long n = blockIdx.x;
    const unsigned char* p = m + n * (long)H * W;
    const long per = (long)H * W;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) { base_s = 0; if (first) first[n] = p[0] ? 1 : 0; }
    __syncthreads();
    int* dst = write ? out + offsets[n] : nullptr;
    for (long c0 = 0; c0 < per - 1; c0 += 4096) {
        const long f0 = c0 + (long)tid * 16;
        unsigned int bits = 0;  // bit k: t[f0+k] != t[f0+k+1]
        if (f0 < per - 1) {
            long f = f0;
            int x = (int)(f / H), y = (int)(f - (long)x * H);
            unsigned char cur = p[(long)y * W + x];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (f + 1 >= per) break;
                int y2 = y + 1, x2 = x;
                if (y2 == H) { y2 = 0; x2 = x + 1; }
                const unsigned char nxt = p[(long)y2 * W + x2];
                if ((cur != 0) != (nxt != 0)) bits |= 1u << k;
                cur = nxt; x = x2; y = y2; ++f;
            }
        }
        const int cnt = __popc(bits);
        int inc = cnt;  // inclusive scan within the wave
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        int wbase = 0;
        for (int k = 0; k < wv; ++k) wbase += wsum[k];
        const int total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        const int base = base_s;
        if (write && cnt) {
            int pos = base + wbase + inc - cnt;
            for (int k = 0; k < 16; ++k) if (bits & (1u << k)) dst[pos++] = (int)(f0 + k);
        }
        __syncthreads();
        if (tid == 0) base_s = base + total;
        __syncthreads();
    }
    if (!write && tid == 0) counts[n] = base_s;
}

extern "C" int ullsam_rle_changes(const unsigned char* masks, long N, int H, int W, int write, int* counts, const long* offsets,
                                  int* out, unsigned char* first, void* stream) {
    if (N == 0) return 0;
    rle_changes_kernel<<<(unsigned)N, 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(masks, H, W, write, counts, offsets, out, first);
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
