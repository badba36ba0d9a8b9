// HBM-bound kernels around the InternLM2 decoder stack.
#include "common.h"

// ---- image-token scan (modeling_internvl_sam.py:135-139,194-199) -------------------------------------------------
// ids int64 [B,S] -> rank int32 [B,S] (k-th image token of the sample, or -1) and range int32 [B,2] = [min_idx, max_idx+1)
__global__ __launch_bounds__(64) void scan_image_tokens_kernel(const long long* __restrict__ ids, int* __restrict__ rank,
                                                               int* __restrict__ range, int S, long long img_id) {
    const int b = blockIdx.x, lane = threadIdx.x;
    int count = 0, lo = S, hi = -1;
    for (int s0 = 0; s0 < S; s0 += 64) {
        const int s = s0 + lane;
        const bool is_img = s < S && ids[(long)b * S + s] == img_id;
        const unsigned long long m = __ballot(is_img);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (s < S) rank[(long)b * S + s] = is_img ? count + before : -1;
        if (is_img) { lo = min(lo, s); hi = max(hi, s); }
        count += __popcll(m);
    }
    for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o, 64)); hi = max(hi, __shfl_xor(hi, o, 64)); }
    if (lane == 0) { range[2 * b] = lo; range[2 * b + 1] = hi + 1; }
}

// ---- token embedding gather + image-token scatter (modeling_internvl_sam.py:124-158 / :412-429) -----------------
// out f32 [B*S, D]: row <- table[id] or, where rank >= 0, vit_embeds[b, rank % n_img]   (the "repeat" branch :143-145)
template <typename T>
__global__ __launch_bounds__(256) void embed_tokens_kernel(const T* __restrict__ table, const long long* __restrict__ ids,
                                                           const int* __restrict__ rank, const float* __restrict__ vit, float* __restrict__ out,
                                                           long rows, int S, int D, int n_img, long vocab) {
    const int dq = D / 4;
    const long total = rows * dq;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / dq;
        const int c = (int)(i - r * dq) * 4;
        const int rk = rank ? rank[r] : -1;
        float4 v;
        if (rk >= 0 && vit) {
            const long b = r / S;
            v = *reinterpret_cast<const float4*>(vit + ((long)b * n_img + (rk % n_img)) * D + c);
        } else {
            long long id = ids[r];
            if (id < 0) id = 0;
            if (id >= vocab) id = vocab - 1;
            v = load4(table + id * D + c);
        }
        *reinterpret_cast<float4*>(out + r * D + c) = v;
    }
}

extern "C" int ullsam_scan_image_tokens(const long long* ids, int* rank, int* range, int B, int S, long long img_id, void* stream) {
    if (B == 0) return 0;
    scan_image_tokens_kernel<<<B, 64, 0, reinterpret_cast<hipStream_t>(stream)>>>(ids, rank, range, S, img_id);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

extern "C" int ullsam_embed_tokens(int dtype, const void* table, const long long* ids, const int* rank, const float* vit_embeds,
                                   float* out, int B, int S, int D, int n_img, long vocab, void* stream) {
    ULLSAM_CHECK(D % 4 == 0, "embed_tokens: D %% 4 != 0");
    const long rows = (long)B * S;
    if (rows == 0) return 0;
    const long total = rows * (D / 4);
    const int grid = (int)min((total + 255) / 256, (long)2048 * 8);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == 0) embed_tokens_kernel<float><<<grid, 256, 0, s>>>((const float*)table, ids, rank, vit_embeds, out, rows, S, D, n_img, vocab);
    else embed_tokens_kernel<bf16><<<grid, 256, 0, s>>>((const bf16*)table, ids, rank, vit_embeds, out, rows, S, D, n_img, vocab);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// ---- rows [start_b, start_b + n) of each sample (hidden_states[-1][:, start:end], modeling_internvl_sam.py:198-200) ----
__global__ __launch_bounds__(256) void gather_rows_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, const int* __restrict__ range,
                                                          int B, int S, int n, int rq) {  // rq = 16-byte chunks per row
    const long total = (long)B * n * rq;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        long t = i;
        const int c = t % rq; t /= rq;
        const int r = t % n; t /= n;
        const int b = (int)t;
        int src = range[2 * b] + r;
        if (src >= S) src = S - 1;
        if (src < 0) src = 0;
        out[i] = in[((long)b * S + src) * rq + c];
    }
}

extern "C" int ullsam_gather_rows(const void* in, void* out, const int* range, int B, int S, int n, int row_bytes, void* stream) {
    ULLSAM_CHECK(row_bytes % 16 == 0, "gather_rows: row_bytes %% 16 != 0");
    const long total = (long)B * n * (row_bytes / 16);
    if (total == 0) return 0;
    const int grid = (int)min((total + 255) / 256, (long)2048 * 8);
    gather_rows_kernel<<<grid, 256, 0, reinterpret_cast<hipStream_t>(stream)>>>((const uint4*)in, (uint4*)out, range, B, S, n, row_bytes / 16);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// ---- wqkv de-interleave + RoPE + KV-cache append (modeling_internlm2.py:361-388, rotate_half :233-247) -----------
// qkv T [B*S, KVH*(G+2)*hd] with per-token layout (kv_head, [q x G, k, v], hd)
// q_out T [B*S, H*hd] (head = kv_head*G + g), k_cache/v_cache T [B, KVH, cap, hd] written at cache_pos0 + s
// pos int32 [B,S] -> rows of the fp32 cos/sin tables [n_pos, hd] (cat(freqs,freqs) layout, :166-170)
template <typename T>
__global__ __launch_bounds__(256) void rope_split_kernel(const T* __restrict__ qkv, T* __restrict__ q_out, T* __restrict__ k_cache,
                                                         T* __restrict__ v_cache, const int* __restrict__ pos, const float* __restrict__ cosT,
                                                         const float* __restrict__ sinT, int B, int S, int KVH, int G, int hd, int cap,
                                                         int cache_pos0, int tab_rows) {
    const int half = hd / 2, hq = half / 4;
    const int gs = G + 2;
    const long total = (long)B * S * KVH * gs * hq;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        long t = i;
        const int d4 = (int)(t % hq) * 4; t /= hq;
        const int g = t % gs; t /= gs;
        const int kv = t % KVH; t /= KVH;
        const long tok = t;  // b*S + s
        const int b = (int)(tok / S), s = (int)(tok % S);
        const T* src = qkv + tok * ((long)KVH * gs * hd) + ((long)kv * gs + g) * hd;
        const float4 x1 = load4(src + d4), x2 = load4(src + half + d4);
        if (g == gs - 1) {  // value: no rotation
            T* dst = v_cache + (((long)b * KVH + kv) * cap + cache_pos0 + s) * hd;
            store4(dst + d4, x1);
            store4(dst + half + d4, x2);
            continue;
        }
        const int p = min(max(pos[tok], 0), tab_rows - 1);  // never read outside the tables (the reference would raise IndexError)
        const float4 c1 = *reinterpret_cast<const float4*>(cosT + (long)p * hd + d4);
        const float4 c2 = *reinterpret_cast<const float4*>(cosT + (long)p * hd + half + d4);
        const float4 s1 = *reinterpret_cast<const float4*>(sinT + (long)p * hd + d4);
        const float4 s2 = *reinterpret_cast<const float4*>(sinT + (long)p * hd + half + d4);
        // q_embed = q*cos + rotate_half(q)*sin, rotate_half = cat(-x2, x1)
        const float4 o1 = make_float4(x1.x * c1.x - x2.x * s1.x, x1.y * c1.y - x2.y * s1.y, x1.z * c1.z - x2.z * s1.z, x1.w * c1.w - x2.w * s1.w);
        const float4 o2 = make_float4(x2.x * c2.x + x1.x * s2.x, x2.y * c2.y + x1.y * s2.y, x2.z * c2.z + x1.z * s2.z, x2.w * c2.w + x1.w * s2.w);
        T* dst;
        if (g == gs - 2) dst = k_cache + (((long)b * KVH + kv) * cap + cache_pos0 + s) * hd;
        else dst = q_out + tok * ((long)KVH * G * hd) + ((long)kv * G + g) * hd;
        store4(dst + d4, o1);
        store4(dst + half + d4, o2);
    }
}

extern "C" int ullsam_rope_split(int dtype, const void* qkv, void* q_out, void* k_cache, void* v_cache, const int* pos,
                                 const float* cos_tab, const float* sin_tab, int B, int S, int KVH, int G, int hd, int cap,
                                 int cache_pos0, int tab_rows, void* stream) {
    ULLSAM_CHECK(hd % 8 == 0, "rope_split: hd %% 8 != 0");
    ULLSAM_CHECK(tab_rows > 0, "rope_split: empty cos/sin tables");
    ULLSAM_CHECK(cache_pos0 + S <= cap, "rope_split: cache overflow (%d + %d > %d)", cache_pos0, S, cap);
    const long total = (long)B * S * KVH * (G + 2) * (hd / 8);
    if (total == 0) return 0;
    const int grid = (int)min((total + 255) / 256, (long)2048 * 8);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == 0) rope_split_kernel<float><<<grid, 256, 0, s>>>((const float*)qkv, (float*)q_out, (float*)k_cache, (float*)v_cache, pos, cos_tab, sin_tab, B, S, KVH, G, hd, cap, cache_pos0, tab_rows);
    else rope_split_kernel<bf16><<<grid, 256, 0, s>>>((const bf16*)qkv, (bf16*)q_out, (bf16*)k_cache, (bf16*)v_cache, pos, cos_tab, sin_tab, B, S, KVH, G, hd, cap, cache_pos0, tab_rows);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// ---- greedy argmax over fp32 logits [R, V] -> int64 (first maximum wins, like torch.argmax) ---------------------
__global__ __launch_bounds__(1024) void argmax_kernel(const float* __restrict__ x, long long* __restrict__ out, long V, long ld) {
    __shared__ float sv[16];
    __shared__ long long si[16];
    const float* row = x + (long)blockIdx.x * ld;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float best = -INFINITY;
    long long bi = 0x7fffffffffffffffLL;
    // four independent loads per trip; within a thread indices increase, so `>` keeps the first maximum
    for (long i0 = tid; i0 < V; i0 += 4096) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = (i0 + 1024 * u < V) ? row[i0 + 1024 * u] : -INFINITY;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (v[u] > best) { best = v[u]; bi = i0 + 1024 * u; }
    }
    auto better = [](float v, long long j, float w, long long k) { return v > w || (v == w && j < k); };
    for (int o = 32; o > 0; o >>= 1) {
        const float v = __shfl_xor(best, o, 64);
        const long long j = __shfl_xor(bi, o, 64);
        if (better(v, j, best, bi)) { best = v; bi = j; }
    }
    if (lane == 0) { sv[wv] = best; si[wv] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w)
            if (better(sv[w], si[w], best, bi)) { best = sv[w]; bi = si[w]; }
        out[blockIdx.x] = bi;
    }
}

extern "C" int ullsam_argmax(const float* logits, long long* out, int rows, long V, long ld, void* stream) {
    if (rows == 0) return 0;
    argmax_kernel<<<rows, 1024, 0, reinterpret_cast<hipStream_t>(stream)>>>(logits, out, V, ld);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}
