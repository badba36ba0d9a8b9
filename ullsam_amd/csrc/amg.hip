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

// One workgroup per mask: ordered emission of the change positions from the change words (see rle_pack_kernel).  1024 threads x 4 words per trip (a 2048^2
// mask is 65536 words = 16 trips of scan + barrier; with 256 threads it was 64 trips and 83 us for the couple of masks an AMG batch keeps: latency, not bytes).
constexpr int EMIT_NT = 1024, EMIT_NW = EMIT_NT / 64;
__global__ __launch_bounds__(EMIT_NT) void rle_emit_kernel(const unsigned long long* __restrict__ words, int H, int W, int nyb,
                                                           const int* __restrict__ select, const long* __restrict__ offsets,
                                                           int* __restrict__ out) {
    __shared__ int wsum[2][EMIT_NW];
    const long n = select ? select[blockIdx.x] : blockIdx.x;
    const unsigned long long* wp = words + n * (long)nyb * W;
    int* dst = out + offsets[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long total = (long)W * nyb;
    int base = 0, par = 0;
    for (long c0 = 0; c0 < total; c0 += 4 * EMIT_NT, par ^= 1) {
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
        int wbase = 0, tot = 0;
#pragma unroll
        for (int k = 0; k < EMIT_NW; ++k) {
            const int v = wsum[par][k];
            if (k < wv) wbase += v;
            tot += v;
        }
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
// select int32 [N]: which masks of `words` to emit (NULL = 0..N-1); offsets i64 [N]: exclusive prefix of the selected masks'
// counts; out int32 [sum(counts)]: change positions f (t[f] != t[f+1]), ordered per mask.
extern "C" int ullsam_rle_emit(const unsigned long long* words, const int* select, long N, int H, int W, const long* offsets, int* out,
                               void* stream) {
    if (N == 0) return 0;
    rle_emit_kernel<<<(unsigned)N, EMIT_NT, 0, reinterpret_cast<hipStream_t>(stream)>>>(words, H, W, (H + 63) / 64, select, offsets, out);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// ---- fused mask post-processing for the generator ---------------------------------------------------------------------
// For every low-res logit map: Sam.postprocess_masks (sam.py:154-162: bilinear to S1 x S1, crop to (nh, nw), bilinear to the crop
// size (CH, CW)) evaluated on the fly, never written; from the virtual CH x CW logits v the kernel derives, in one pass,
//   * calculate_stability_score's two counts  #(v > thr+off), #(v > thr-off)                       (amg.py:156-176)
//   * batched_mask_to_box of (v > thr), in crop coordinates                                         (amg.py:303-346)
//   * the change words of mask_to_rle_pytorch(uncrop_masks(v > thr, crop_box, FH, FW))              (amg.py:107-135, 251-264)
// so the 4 B/pixel logits and 1 B/pixel masks of the separate helpers (12288 masks x 16 MiB per 2048^2 tile) are replaced by
// 1 bit/pixel of change words.  The row taps are wave-uniform, so the x-interpolated low-res rows and intermediate rows are
// memoised in registers (two slots each) while a wave walks down its 64 rows.
__device__ inline Tap utap_of(int o, float scale, int n_in) {  // wave-uniform argument -> scalar indices
    Tap t = tap_of(o, scale, n_in);
    t.i0 = __builtin_amdgcn_readfirstlane(t.i0);
    t.i1 = __builtin_amdgcn_readfirstlane(t.i1);
    return t;
}
struct PostGeom { int LH, LW, S1, nh, nw, CH, CW, FH, FW, cx0, cy0; float s1y, s1x, s2y, s2x; };

__device__ inline float post_eval(const float* __restrict__ lowp, const PostGeom& g, int y, int x) {  // one pixel, no memo (named scalars: two-element arrays here went to scratch)
    const Tap ty = tap_of(y, g.s2y, g.nh), tx = tap_of(x, g.s2x, g.nw);
    const Tap c0 = tap_of(tx.i0, g.s1x, g.LW), c1 = tap_of(tx.i1, g.s1x, g.LW);
    auto inter = [&](int iy) {                                               // the intermediate (S1-grid) row iy at the two column taps, then along x
        const Tap r = tap_of(iy, g.s1y, g.LH);
        const float* r0 = lowp + (long)r.i0 * g.LW;
        const float* r1 = lowp + (long)r.i1 * g.LW;
        const float iv0 = lerp_rn(lerp_rn(r0[c0.i0], r0[c0.i1], c0.l), lerp_rn(r1[c0.i0], r1[c0.i1], c0.l), r.l);
        const float iv1 = lerp_rn(lerp_rn(r0[c1.i0], r0[c1.i1], c1.l), lerp_rn(r1[c1.i0], r1[c1.i1], c1.l), r.l);
        return lerp_rn(iv0, iv1, tx.l);
    };
    const float h0 = inter(ty.i0), h1 = inter(ty.i1);
    return lerp_rn(h0, h1, ty.l);
}

// acc = 2*acc + (v > thr): compare + add-with-carry, exact `>` semantics (false for NaN).  Rows are pushed top first, so after
// 32 pushes the word is bit-reversed (row 0 in bit 31).
__device__ __forceinline__ void push_gt(unsigned int& acc, float v, float thr) {
    asm("v_cmp_lt_f32_e32 vcc, %2, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(acc) : "v"(v), "v"(thr) : "vcc");
}

// A wave owns 64 frame rows x 64*NC frame columns; a lane owns NC adjacent columns, so the wave-uniform row bookkeeping (taps,
// memo lookups) is paid once per 64*NC pixels.  Before walking the rows the wave takes min/max of the low-res footprint of its
// block: both resizes are convex combinations, so a footprint entirely above (below) every threshold by a rounding margin decides
// all of the block's bits without evaluating a pixel -- real masks are flat almost everywhere.
template <int NC>
__global__ __launch_bounds__(256) void amg_postprocess_kernel(const float* __restrict__ low, const int* __restrict__ index, PostGeom g,
                                                              float thr, float off, unsigned long long* __restrict__ words,
                                                              int* __restrict__ rle_counts, unsigned char* __restrict__ first,
                                                              int* __restrict__ boxes, unsigned int* __restrict__ stab) {
    const long n = blockIdx.z;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar for the compiler too: the block's rows, the memo keys and every branch on them below are wave-uniform
    const int nyb = (g.FH + 63) >> 6;
    const int yb = blockIdx.y * 4 + wv;
    if (yb >= nyb) return;
    const float* lowp = low + (long)(index ? index[n] : n) * g.LH * g.LW;
    const int bx0 = blockIdx.x * 64 * NC;             // first frame column of the wave
    const int xf0 = bx0 + lane * NC;                  // first frame column of the lane
    const int ys = yb * 64;
    const int r = min(64, g.FH - ys);                 // frame rows in this block
    const float hi = thr + off, lo = thr - off;
    Tap tx[NC], t0[NC], t1[NC];
    bool col_in[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int x = xf0 + j - g.cx0;
        col_in[j] = xf0 + j < g.FW && x >= 0 && x < g.CW;
        tx[j] = tap_of(min(max(x, 0), g.CW - 1), g.s2x, g.nw);
        t0[j] = tap_of(tx[j].i0, g.s1x, g.LW);
        t1[j] = tap_of(tx[j].i1, g.s1x, g.LW);
    }
    // rows of the block that lie inside the crop: bits [i_lo, i_hi)
    const int i_lo = max(0, g.cy0 - ys), i_hi = min(r, g.cy0 + g.CH - ys);
    unsigned long long rowmask = 0;
    if (i_hi > i_lo) rowmask = ((i_hi - i_lo) == 64 ? ~0ull : ((1ull << (i_hi - i_lo)) - 1ull)) << i_lo;

    // ---- footprint test
    int decided = 0;  // 1: every pixel of the block is above all thresholds, 2: below all
    {
        const int xl = min(max(bx0 - g.cx0, 0), g.CW - 1), xh = min(max(bx0 + 64 * NC - 1 - g.cx0, 0), g.CW - 1);
        const int yl = min(max(ys - g.cy0, 0), g.CH - 1), yh = min(max(ys + r - g.cy0, 0), g.CH - 1);  // incl. the row after the block
        const int lc0 = tap_of(tap_of(xl, g.s2x, g.nw).i0, g.s1x, g.LW).i0, lc1 = tap_of(tap_of(xh, g.s2x, g.nw).i1, g.s1x, g.LW).i1;
        const int lr0 = tap_of(tap_of(yl, g.s2y, g.nh).i0, g.s1y, g.LH).i0, lr1 = tap_of(tap_of(yh, g.s2y, g.nh).i1, g.s1y, g.LH).i1;
        const int nc = lc1 - lc0 + 1, cells = nc * (lr1 - lr0 + 1);
        float mn = INFINITY, mx = -INFINITY;
        int bad = 0;
        for (int i = lane; i < cells; i += 64) {
            const int rr = i / nc, cc = i - rr * nc;
            const float val = lowp[(long)(lr0 + rr) * g.LW + lc0 + cc];
            mn = fminf(mn, val); mx = fmaxf(mx, val);
            bad |= !(val == val);
        }
        for (int o = 32; o > 0; o >>= 1) {
            mn = fminf(mn, __shfl_xor(mn, o, 64)); mx = fmaxf(mx, __shfl_xor(mx, o, 64)); bad |= __shfl_xor(bad, o, 64);
        }
        const float margin = 1e-4f * fmaxf(1.f, fmaxf(fabsf(mn), fabsf(mx)));
        if (!bad && mn > fmaxf(hi, fmaxf(lo, thr)) + margin) decided = 1;
        else if (!bad && mx < fminf(hi, fminf(lo, thr)) - margin) decided = 2;
        decided = __builtin_amdgcn_readfirstlane(decided);
    }

    unsigned long long wm[NC], whi[NC], wlo[NC];
    unsigned int nb = 0;                               // bit j: the element after column j's word in the frame's Fortran order
    if (decided) {
#pragma unroll
        for (int j = 0; j < NC; ++j) wm[j] = whi[j] = wlo[j] = (decided == 1 && col_in[j]) ? rowmask : 0ull;
        if (ys + r < g.FH && decided == 1) {
            const int yc = ys + r - g.cy0;
            if (yc >= 0 && yc < g.CH)
#pragma unroll
                for (int j = 0; j < NC; ++j) nb |= (unsigned int)col_in[j] << j;
        }
    } else {
        // two memo slots per level, addressed by a SCALAR slot number (keys and slot numbers are wave-uniform: scalar branches and selects; the first form
        // handed the slots around as array references and the compiler kept them in scratch behind lane-masked branches, 32 B per lane in the row loop)
        float ga0[NC], ga1[NC], gb0[NC], gb1[NC], ha[NC], hb[NC];
        int gka = -1, gkb = -1, hka = -1, hkb = -1;  // memo keys: low-res row / intermediate row
        auto ensure_g = [&](int lr) -> int {          // -> the slot that holds low-res row lr, x-interpolated at this lane's two column taps
            if (lr == gka) return 0;
            if (lr == gkb) return 1;
            const float* row = lowp + (long)lr * g.LW;
            const int slot = gka <= gkb ? 0 : 1;      // the older row goes (rows only move down)
            if (slot == 0) {
                gka = lr;
#pragma unroll
                for (int j = 0; j < NC; ++j) { ga0[j] = lerp_rn(row[t0[j].i0], row[t0[j].i1], t0[j].l); ga1[j] = lerp_rn(row[t1[j].i0], row[t1[j].i1], t1[j].l); }
            } else {
                gkb = lr;
#pragma unroll
                for (int j = 0; j < NC; ++j) { gb0[j] = lerp_rn(row[t0[j].i0], row[t0[j].i1], t0[j].l); gb1[j] = lerp_rn(row[t1[j].i0], row[t1[j].i1], t1[j].l); }
            }
            return slot;
        };
        auto ensure_h = [&](int iy) -> int {          // -> the slot that holds intermediate row iy at this lane's columns
            if (iy == hka) return 0;
            if (iy == hkb) return 1;
            const Tap rr = utap_of(iy, g.s1y, g.LH);
            const int sp = ensure_g(rr.i0), sq = ensure_g(rr.i1);   // (the second call never evicts the first's row: its key is the newest)
            const int slot = hka <= hkb ? 0 : 1;
            float h[NC];
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const float p0 = sp ? gb0[j] : ga0[j], p1 = sp ? gb1[j] : ga1[j], q0 = sq ? gb0[j] : ga0[j], q1 = sq ? gb1[j] : ga1[j];
                const float i0 = lerp_rn(p0, q0, rr.l), i1 = lerp_rn(p1, q1, rr.l);
                h[j] = lerp_rn(i0, i1, tx[j].l);
            }
            if (slot == 0) {
                hka = iy;
#pragma unroll
                for (int j = 0; j < NC; ++j) ha[j] = h[j];
            } else {
                hkb = iy;
#pragma unroll
                for (int j = 0; j < NC; ++j) hb[j] = h[j];
            }
            return slot;
        };
        auto values = [&](int yc, float (&v)[NC]) {   // crop row yc (wave-uniform), this lane's columns
            const Tap ty = utap_of(yc, g.s2y, g.nh);
            const int s0 = ensure_h(ty.i0), s1 = ensure_h(ty.i1);
#pragma unroll
            for (int j = 0; j < NC; ++j) v[j] = lerp_rn(s0 ? hb[j] : ha[j], s1 ? hb[j] : ha[j], ty.l);
        };
#pragma unroll
        for (int j = 0; j < NC; ++j) wm[j] = whi[j] = wlo[j] = 0;
        for (int half = 0; half < 2; ++half) {
            unsigned int am[NC], ah[NC], al[NC];
#pragma unroll
            for (int j = 0; j < NC; ++j) am[j] = ah[j] = al[j] = 0;
            for (int i = 0; i < 32; ++i) {
                const int row = half * 32 + i;
                if (row >= i_lo && row < i_hi) {
                    float v[NC];
                    values(ys + row - g.cy0, v);
#pragma unroll
                    for (int j = 0; j < NC; ++j) { push_gt(am[j], v[j], thr); push_gt(ah[j], v[j], hi); push_gt(al[j], v[j], lo); }
                } else {
#pragma unroll
                    for (int j = 0; j < NC; ++j) { am[j] += am[j]; ah[j] += ah[j]; al[j] += al[j]; }
                }
            }
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                wm[j] |= (unsigned long long)__brev(am[j]) << (32 * half);
                whi[j] |= (unsigned long long)__brev(ah[j]) << (32 * half);
                wlo[j] |= (unsigned long long)__brev(al[j]) << (32 * half);
            }
        }
#pragma unroll
        for (int j = 0; j < NC; ++j)
            if (!col_in[j]) { wm[j] = 0; whi[j] = 0; wlo[j] = 0; }
        if (ys + r < g.FH) {
            const int yc = ys + r - g.cy0;
            if (yc >= 0 && yc < g.CH) {
                float v[NC];
                values(yc, v);
#pragma unroll
                for (int j = 0; j < NC; ++j) nb |= (unsigned int)(col_in[j] && v[j] > thr) << j;
            }
        }
    }
    unsigned int no_next = 0;
    if (ys + r >= g.FH) {                              // column end: the next element is the top of the next frame column
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const int xf = xf0 + j;
            if (xf + 1 < g.FW) {
                const int x2 = xf + 1 - g.cx0;
                if (g.cy0 == 0 && x2 >= 0 && x2 < g.CW) nb |= (unsigned int)(post_eval(lowp, g, 0, x2) > thr) << j;
            } else {
                no_next |= 1u << j;
            }
        }
    }
    const unsigned long long keep = r == 64 ? ~0ull : ((1ull << r) - 1ull);
    int cnt = 0;
    unsigned int ca = 0, cb = 0;
    int x0 = 0x7fffffff, y0 = 0x7fffffff, x1 = -1, y1 = -1;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int xf = xf0 + j;
        unsigned long long ch = (wm[j] ^ ((wm[j] >> 1) | ((unsigned long long)((nb >> j) & 1u) << (r - 1)))) & keep;
        if ((no_next >> j) & 1u) ch &= ~(1ull << (r - 1));
        if (xf < g.FW) { words[((long)n * nyb + yb) * g.FW + xf] = ch; cnt += __popcll(ch); }
        if (xf == 0 && yb == 0) first[n] = (unsigned char)(wm[j] & 1ull);
        ca += __popcll(whi[j]); cb += __popcll(wlo[j]);
        if (wm[j]) {
            x0 = min(x0, xf - g.cx0); x1 = max(x1, xf - g.cx0);
            y0 = min(y0, ys - g.cy0 + (int)__builtin_ctzll(wm[j]));
            y1 = max(y1, ys - g.cy0 + 63 - (int)__builtin_clzll(wm[j]));
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        cnt += __shfl_xor(cnt, o, 64); ca += __shfl_xor(ca, o, 64); cb += __shfl_xor(cb, o, 64);
        x0 = min(x0, __shfl_xor(x0, o, 64)); y0 = min(y0, __shfl_xor(y0, o, 64));
        x1 = max(x1, __shfl_xor(x1, o, 64)); y1 = max(y1, __shfl_xor(y1, o, 64));
    }
    if (lane == 0) {
        if (cnt) atomicAdd(&rle_counts[n], cnt);
        if (ca) atomicAdd(&stab[2 * n], ca);
        if (cb) atomicAdd(&stab[2 * n + 1], cb);
        if (x1 >= 0) {
            atomicMin(&boxes[4 * n + 0], x0); atomicMin(&boxes[4 * n + 1], y0);
            atomicMax(&boxes[4 * n + 2], x1); atomicMax(&boxes[4 * n + 3], y1);
        }
    }
}

// low f32 [*, LH, LW]; index int32 [M] rows of `low` to process (NULL = 0..M-1); S1 = Sam.image_encoder.img_size; (nh, nw) =
// input_size; (CH, CW) = crop size; the crop sits at (cx0, cy0) of the FH x FW frame.  Outputs: words u64 [M, ceil(FH/64), FW],
// rle_counts int [M], first u8 [M], boxes int [M,4] (crop coordinates, zeros when empty), stab u32 [M,2].
extern "C" int ullsam_amg_postprocess(const float* low, const int* index, long M, int LH, int LW, int S1, int nh, int nw, int CH,
                                      int CW, int FH, int FW, int cx0, int cy0, float mask_threshold, float threshold_offset,
                                      unsigned long long* words, int* rle_counts, unsigned char* first, int* boxes,
                                      unsigned int* stab, void* stream) {
    if (M == 0) return 0;
    ULLSAM_CHECK(LH > 0 && LW > 0 && S1 > 0 && nh > 0 && nw > 0 && nh <= S1 && nw <= S1 && CH > 0 && CW > 0, "amg_postprocess: bad sizes");
    ULLSAM_CHECK(cx0 >= 0 && cy0 >= 0 && cx0 + CW <= FW && cy0 + CH <= FH && (long)FH * FW < (1L << 31), "amg_postprocess: crop outside the frame");
    ULLSAM_CHECK(((uintptr_t)boxes & 15) == 0, "amg_postprocess: boxes must be 16-byte aligned");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (hipMemsetAsync(rle_counts, 0, sizeof(int) * M, s) != hipSuccess || hipMemsetAsync(stab, 0, sizeof(unsigned int) * 2 * M, s) != hipSuccess) {
        ullsam_set_error("amg_postprocess: memset failed");
        return -2;
    }
    box_init_kernel<<<(unsigned)((M + 255) / 256), 256, 0, s>>>(boxes, M);
    ULLSAM_LAUNCH_CHECK();
    PostGeom g;
    g.LH = LH; g.LW = LW; g.S1 = S1; g.nh = nh; g.nw = nw; g.CH = CH; g.CW = CW; g.FH = FH; g.FW = FW; g.cx0 = cx0; g.cy0 = cy0;
    g.s1y = (float)LH / (float)S1; g.s1x = (float)LW / (float)S1; g.s2y = (float)nh / (float)CH; g.s2x = (float)nw / (float)CW;
    const int nyb = (FH + 63) / 64;
    if (FW > 128)
        amg_postprocess_kernel<4><<<dim3((unsigned)((FW + 255) / 256), (unsigned)((nyb + 3) / 4), (unsigned)M), 256, 0, s>>>(
            low, index, g, mask_threshold, threshold_offset, words, rle_counts, first, boxes, stab);
    else
        amg_postprocess_kernel<1><<<dim3((unsigned)((FW + 63) / 64), (unsigned)((nyb + 3) / 4), (unsigned)M), 256, 0, s>>>(
            low, index, g, mask_threshold, threshold_offset, words, rle_counts, first, boxes, stab);
    ULLSAM_LAUNCH_CHECK();
    box_finish_kernel<<<(unsigned)((M + 255) / 256), 256, 0, s>>>(boxes, M);
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
