// Backward kernels of the segmentation branch of the reference's training step (train_joint_v2.py:1026-1100): everything downstream of
// the LLM's last hidden state -- mlp2 (modeling_internvl_sam.py:95-100, 253-270), the prompt encoder's LLM-conditioned dense prompt and
// point embeddings (prompt_encoder.py:131-203), the mask decoder (mask_decoder.py:112-149, transformer.py:62-242), the bilinear
// upsample and the BCE + Dice loss (train_joint_v2.py:605-661, 774-812).  fp32 throughout.  These are correctness-first kernels
// (one output per thread, fp32 FMA): the forward of the same step runs on the inference kernels; what is here is what autograd needs
// on top of them.  ullsam_amd/training.py holds the torch.autograd.Function wrappers.
#include "common.h"

// ---- C[b](m, n) = (accumulate ? C : 0) + sum_k A[b](m, k) * B[b](k, n), every operand with explicit element strides -----------------
// One kernel covers dX = dY W (B = W as stored), dW = dY^T X (A read transposed), the hypernetwork product and its two gradients, the
// attention score / probability products of the matrix-form attention and the rel-pos einsums.
struct MmArgs {
    const float* A; const float* B; float* C;
    int M, N, K, batch;
    long a_b, a_m, a_k, b_b, b_k, b_n, c_b, c_m, c_n;
    int accumulate;
    int ksplit, kc;      // blockIdx.z = batch index * ksplit + s: split s sums k in [s kc, min(K, (s + 1) kc)) into C + s c_s (partials, reduced in order afterwards)
    long c_s;
    // two-level batch index (ullsam_train_matmul_heads): bin > 1 -> entry b = (o, h) = (b / bin, b % bin) of operand X starts at o X_b + (h / X_div) X_h, i.e. the heads of
    // [rows, heads x hd] activations are read / written in place (h / div: grouped KV heads, modeling_internlm2.py:250-259) instead of from head-major copies
    int bin, a_div, b_div;
    long a_h, b_h, c_h;
    // causal structure of a square attention (Sq == Sk, key j visible to query i iff j <= i), MFMA kernels only: 1 = C[m = query][n = key]: tiles with every key behind every query are
    // not formed (the row pass does not read masked entries and writes zeros there); 2 = the sum runs over queries and m is the key: queries before the tile's first key meet zeros,
    // start at it; 3 = the sum runs over keys and m is the query: keys behind the tile's last query meet zeros, stop after it.  0 elsewhere.
    int tri;
};
__device__ __forceinline__ bool mm_tri_clip(MmArgs& p, int m0, int n0) {   // -> true: nothing to do for this tile
    if (p.tri == 1) return n0 >= m0 + 128;
    if (p.tri == 2) { const int kb = min(m0 & ~31, p.K); p.A += (long)kb * p.a_k; p.B += (long)kb * p.b_k; p.K -= kb; }
    if (p.tri == 3) p.K = min(p.K, m0 + 128);
    return false;
}
__device__ __forceinline__ long mm_boff(int b, int bin, long s_o, long s_h, int div) { return bin > 1 ? (long)(b / bin) * s_o + (long)((b % bin) / div) * s_h : (long)b * s_o; }
__global__ __launch_bounds__(256) void matmul_f32_kernel(MmArgs p) {
    // 64 x 64 outputs per workgroup, 4 x 4 per thread, K in steps of 16 through LDS.  The element -> thread assignment of the two tile loads
    // follows the operand's unit stride (k-fastest for a row-major operand, m- / n-fastest for a transposed one), so both are coalesced.
    __shared__ float As[16][65], Bs[16][65];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64, b = blockIdx.z / p.ksplit, ks = blockIdx.z % p.ksplit;
    const float* A = p.A + mm_boff(b, p.bin, p.a_b, p.a_h, p.a_div) + (long)ks * p.kc * p.a_k;
    const float* B = p.B + mm_boff(b, p.bin, p.b_b, p.b_h, p.b_div) + (long)ks * p.kc * p.b_k;
    p.K = min(p.kc, p.K - ks * p.kc);
    p.C += (long)ks * p.c_s;
    const bool a_kfast = p.a_k <= p.a_m, b_nfast = p.b_n <= p.b_k;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int k0 = 0; k0 < p.K; k0 += 16) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            const int am = a_kfast ? idx >> 4 : idx & 63, ak = a_kfast ? idx & 15 : idx >> 6;
            As[ak][am] = (m0 + am < p.M && k0 + ak < p.K) ? A[(long)(m0 + am) * p.a_m + (long)(k0 + ak) * p.a_k] : 0.f;
            const int bn = b_nfast ? idx & 63 : idx >> 4, bk = b_nfast ? idx >> 6 : idx & 15;
            Bs[bk][bn] = (k0 + bk < p.K && n0 + bn < p.N) ? B[(long)(k0 + bk) * p.b_k + (long)(n0 + bn) * p.b_n] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float av[4], bv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { av[i] = As[kk][ty * 4 + i]; bv[i] = Bs[kk][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] += av[i] * bv[j];
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + ty * 4 + i, n = n0 + tx * 4 + j;
            if (m < p.M && n < p.N) {
                float* c = p.C + mm_boff(b, p.bin, p.c_b, p.c_h, 1) + (long)m * p.c_m + (long)n * p.c_n;
                *c = p.accumulate ? *c + acc[i][j] : acc[i][j];
            }
        }
}
// The same product on the matrix pipe (v_mfma_f32_32x32x2_f32: an exact fp32 fma chain, so the result differs from the kernel above only in the
// order of the k sums): 128 x 128 outputs per workgroup, four waves of 64 x 64, K in steps of 32 through two LDS buffers filled from registers
// (the global loads of step t + 1 are in flight under step t's MFMAs; one barrier per step).  Both tiles sit in LDS k-major, [k][m ^ (k >> 1)],
// row stride 160 words: a fragment read (32 consecutive m of row k | 32 of row k + 1) touches 64 distinct banks, and so does the transposing
// write of a k-fastest operand (32 consecutive k of one m).  This is what the matrix-form attention of the train step runs on (the score,
// probability and gradient products of ViT-H's 4096 x 4096 global blocks: 45 % of a step's GPU time on the scalar kernel).
constexpr int MMF_LD = 160;
template <bool AK, bool BN>   // AK: A's unit stride is k (row-major [m][k]); BN: B's unit stride is n (row-major [k][n])
__global__ __launch_bounds__(256, 2) void matmul_f32_mfma_kernel(MmArgs p) {
    extern __shared__ float mmf_lds[];                     // [2 buffers][A | B][32 k][160]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128, b = blockIdx.z / p.ksplit, ks = blockIdx.z % p.ksplit;
    p.A += (long)ks * p.kc * p.a_k;
    p.B += (long)ks * p.kc * p.b_k;
    p.K = min(p.kc, p.K - ks * p.kc);
    p.C += (long)ks * p.c_s;
    if (mm_tri_clip(p, m0, n0)) return;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // element i (0..15) of a thread's share of a 128 x 32 tile: the thread index runs along the operand's unit stride
    //   k-fastest:  row (tid >> 5) + 8 i, k = tid & 31          row-fastest:  row tid & 127, k = (tid >> 7) + 2 i
    const int t5 = tid >> 5, k5 = tid & 31, t7 = tid >> 7, r7 = tid & 127;
    const float* pa = p.A + mm_boff(b, p.bin, p.a_b, p.a_h, p.a_div) + (AK ? (long)(m0 + t5) * p.a_m + (long)k5 * p.a_k : (long)(m0 + r7) * p.a_m + (long)t7 * p.a_k);
    const float* pb = p.B + mm_boff(b, p.bin, p.b_b, p.b_h, p.b_div) + (BN ? (long)(n0 + r7) * p.b_n + (long)t7 * p.b_k : (long)(n0 + t5) * p.b_n + (long)k5 * p.b_k);
    const long a_inc = AK ? 8 * p.a_m : 2 * p.a_k, b_inc = BN ? 2 * p.b_k : 8 * p.b_n, a_step = 32 * p.a_k, b_step = 32 * p.b_k;
    const bool inner = m0 + 128 <= p.M && n0 + 128 <= p.N;
    float ra[16], rb[16];
    auto fetch = [&](int k0) {
        if (inner && k0 + 32 <= p.K) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { ra[i] = pa[i * a_inc]; rb[i] = pb[i * b_inc]; }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const bool va = AK ? (m0 + t5 + 8 * i < p.M && k0 + k5 < p.K) : (m0 + r7 < p.M && k0 + t7 + 2 * i < p.K);
                const bool vb = BN ? (n0 + r7 < p.N && k0 + t7 + 2 * i < p.K) : (n0 + t5 + 8 * i < p.N && k0 + k5 < p.K);
                ra[i] = va ? pa[i * a_inc] : 0.f;
                rb[i] = vb ? pb[i * b_inc] : 0.f;
            }
        }
        pa += a_step;
        pb += b_step;
    };
    auto stash = [&](int buf) {
        float* As = mmf_lds + buf * (2 * 32 * MMF_LD);
        float* Bs = As + 32 * MMF_LD;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (AK) As[k5 * MMF_LD + ((t5 + 8 * i) ^ (k5 >> 1))] = ra[i];
            else    As[(t7 + 2 * i) * MMF_LD + (r7 ^ i)] = ra[i];
            if (BN) Bs[(t7 + 2 * i) * MMF_LD + (r7 ^ i)] = rb[i];
            else    Bs[k5 * MMF_LD + ((t5 + 8 * i) ^ (k5 >> 1))] = rb[i];
        }
    };
    const int steps = (p.K + 31) / 32;
    fetch(0);
    stash(0);
    __syncthreads();
    const int kh = lane >> 5, l5 = lane & 31;
    for (int t = 0; t < steps; ++t) {
        if (t + 1 < steps) fetch(32 * (t + 1));
        const float* As = mmf_lds + (t & 1) * (2 * 32 * MMF_LD) + kh * MMF_LD;
        const float* Bs = As + 32 * MMF_LD;
#pragma unroll
        for (int kp = 0; kp < 16; ++kp) {
            float af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = As[2 * kp * MMF_LD + ((wm * 64 + i * 32 + l5) ^ kp)];
                bf[i] = Bs[2 * kp * MMF_LD + ((wn * 64 + i * 32 + l5) ^ kp)];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (t + 1 < steps) stash((t + 1) & 1);
        __syncthreads();
    }
    // accumulator element e of lane l: row 8 (e / 4) + 4 (l / 32) + e % 4, column l % 32 of the 32 x 32 tile
    float* cb = p.C + mm_boff(b, p.bin, p.c_b, p.c_h, 1);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * 64 + i * 32 + 8 * (e >> 2) + 4 * kh + (e & 3), n = n0 + wn * 64 + j * 32 + l5;
                if (m < p.M && n < p.N) {
                    float* c = cb + (long)m * p.c_m + (long)n * p.c_n;
                    *c = p.accumulate ? *c + acc[i][j][e] : acc[i][j][e];
                }
            }
}
template <bool AK, bool BN>
static void launch_matmul_mfma(const MmArgs& a, hipStream_t stream) {
    constexpr int LDS = 2 * 2 * 32 * MMF_LD * 4;
    static PerDeviceOnce attr;                 // per device: the attribute belongs to the current device's function object (a process may train on several GPUs)
    if (attr.first() && hipFuncSetAttribute(reinterpret_cast<const void*>(matmul_f32_mfma_kernel<AK, BN>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
        ullsam_set_error("train matmul: hipFuncSetAttribute(MaxDynamicSharedMemorySize = %d) failed", LDS);   // (the launch below then fails and is reported by the caller's check)
    matmul_f32_mfma_kernel<AK, BN><<<dim3((a.N + 127) / 128, (a.M + 127) / 128, a.batch * a.ksplit), 256, LDS, stream>>>(a);
}
// The same product with both operands ROUNDED TO bf16 on their way into LDS and the sums on v_mfma_f32_32x32x16_bf16 (fp32 accumulation, fp32 result):
// what torch.autocast(bfloat16) makes of a matmul, i.e. what the reference's trainer computes for the attention products of its bf16 model
// (train_joint_v2.py:1665).  Operands and result stay fp32 in memory, so the kernel is a drop-in for the one above; 16x fewer matrix cycles leave it
// bound by the operand reads.  LDS image: [row][32 k] bf16 with an 80-byte row stride (a 16-lane group of a ds_read_b128 fragment read covers all 64 banks).
constexpr int MMB_LD = 40;
template <bool AK, bool BN, bool VEC = false>   // VEC: every k-fastest operand (A if AK, B if !BN) is fetched as float4 along k -- the launcher has checked unit k stride, 16-byte alignment, K % 4 == 0
__global__ __launch_bounds__(256, 2) void matmul_bf16_mfma_kernel(MmArgs p) {
    extern __shared__ bf16 mmb_lds[];                      // [2 buffers][A | B][128 rows][40]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128, b = blockIdx.z / p.ksplit, ks = blockIdx.z % p.ksplit;
    p.A += (long)ks * p.kc * p.a_k;
    p.B += (long)ks * p.kc * p.b_k;
    p.K = min(p.kc, p.K - ks * p.kc);
    p.C += (long)ks * p.c_s;
    if (mm_tri_clip(p, m0, n0)) return;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int t5 = tid >> 5, k5 = tid & 31, t7 = tid >> 7, r7 = tid & 127;
    // element i (0..15) of a thread's share of a 128 x 32 tile.  Scalar: k-fastest row (tid >> 5) + 8 i, k = tid & 31; row-fastest row tid & 127, k = (tid >> 7) + 2 i.
    // VEC (k-fastest operands only): rows (tid >> 3) + 32 j, j = 0..3, the four k 4 (tid & 7) .. + 3 as one 16-byte load -> one 8-byte LDS store (the LDS image is the same)
    constexpr bool VA = VEC && AK, VB = VEC && !BN;
    const int tr = tid >> 3, kq = 4 * (tid & 7);
    const float* pa = p.A + mm_boff(b, p.bin, p.a_b, p.a_h, p.a_div) + (VA ? (long)(m0 + tr) * p.a_m + kq : AK ? (long)(m0 + t5) * p.a_m + (long)k5 * p.a_k : (long)(m0 + r7) * p.a_m + (long)t7 * p.a_k);
    const float* pb = p.B + mm_boff(b, p.bin, p.b_b, p.b_h, p.b_div) + (VB ? (long)(n0 + tr) * p.b_n + kq : BN ? (long)(n0 + r7) * p.b_n + (long)t7 * p.b_k : (long)(n0 + t5) * p.b_n + (long)k5 * p.b_k);
    const long a_inc = VA ? 32 * p.a_m : AK ? 8 * p.a_m : 2 * p.a_k, b_inc = VB ? 32 * p.b_n : BN ? 2 * p.b_k : 8 * p.b_n, a_step = 32 * p.a_k, b_step = 32 * p.b_k;
    const bool inner = m0 + 128 <= p.M && n0 + 128 <= p.N;
    float ra[16], rb[16];
    auto fetch = [&](int k0) {
        const bool whole = inner && k0 + 32 <= p.K;
        if constexpr (VA) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 v = (whole || (m0 + tr + 32 * j < p.M && k0 + kq < p.K)) ? *reinterpret_cast<const float4*>(pa + j * a_inc) : make_float4(0.f, 0.f, 0.f, 0.f);
                ra[4 * j] = v.x; ra[4 * j + 1] = v.y; ra[4 * j + 2] = v.z; ra[4 * j + 3] = v.w;
            }
        }
        if constexpr (VB) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 v = (whole || (n0 + tr + 32 * j < p.N && k0 + kq < p.K)) ? *reinterpret_cast<const float4*>(pb + j * b_inc) : make_float4(0.f, 0.f, 0.f, 0.f);
                rb[4 * j] = v.x; rb[4 * j + 1] = v.y; rb[4 * j + 2] = v.z; rb[4 * j + 3] = v.w;
            }
        }
        if (whole) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { if constexpr (!VA) ra[i] = pa[i * a_inc]; if constexpr (!VB) rb[i] = pb[i * b_inc]; }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const bool va = AK ? (m0 + t5 + 8 * i < p.M && k0 + k5 < p.K) : (m0 + r7 < p.M && k0 + t7 + 2 * i < p.K);
                const bool vb = BN ? (n0 + r7 < p.N && k0 + t7 + 2 * i < p.K) : (n0 + t5 + 8 * i < p.N && k0 + k5 < p.K);
                if constexpr (!VA) ra[i] = va ? pa[i * a_inc] : 0.f;
                if constexpr (!VB) rb[i] = vb ? pb[i * b_inc] : 0.f;
            }
        }
        pa += a_step;
        pb += b_step;
    };
    auto pack4 = [](const float* r) { bf16x4_t v; v[0] = (__bf16)r[0]; v[1] = (__bf16)r[1]; v[2] = (__bf16)r[2]; v[3] = (__bf16)r[3]; return v; };
    auto stash = [&](int buf) {
        bf16* As = mmb_lds + buf * (2 * 128 * MMB_LD);
        bf16* Bs = As + 128 * MMB_LD;
        if constexpr (VA) {
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<bf16x4_t*>(As + (tr + 32 * j) * MMB_LD + kq) = pack4(ra + 4 * j);
        }
        if constexpr (VB) {
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<bf16x4_t*>(Bs + (tr + 32 * j) * MMB_LD + kq) = pack4(rb + 4 * j);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if constexpr (!VA) { if (AK) As[(t5 + 8 * i) * MMB_LD + k5] = (bf16)ra[i]; else As[r7 * MMB_LD + t7 + 2 * i] = (bf16)ra[i]; }
            if constexpr (!VB) { if (BN) Bs[r7 * MMB_LD + t7 + 2 * i] = (bf16)rb[i]; else Bs[(t5 + 8 * i) * MMB_LD + k5] = (bf16)rb[i]; }
        }
    };
    const int steps = (p.K + 31) / 32;
    fetch(0);
    stash(0);
    __syncthreads();
    const int kh = lane >> 5, l5 = lane & 31;
    for (int t = 0; t < steps; ++t) {
        if (t + 1 < steps) fetch(32 * (t + 1));
        const bf16* As = mmb_lds + (t & 1) * (2 * 128 * MMB_LD) + 8 * kh;
        const bf16* Bs = As + 128 * MMB_LD;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8_t af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = *reinterpret_cast<const bf16x8_t*>(As + (wm * 64 + i * 32 + l5) * MMB_LD + 16 * kk);
                bf[i] = *reinterpret_cast<const bf16x8_t*>(Bs + (wn * 64 + i * 32 + l5) * MMB_LD + 16 * kk);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (t + 1 < steps) stash((t + 1) & 1);
        __syncthreads();
    }
    float* cb = p.C + mm_boff(b, p.bin, p.c_b, p.c_h, 1);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * 64 + i * 32 + 8 * (e >> 2) + 4 * kh + (e & 3), n = n0 + wn * 64 + j * 32 + l5;
                if (m < p.M && n < p.N) {
                    float* c = cb + (long)m * p.c_m + (long)n * p.c_n;
                    *c = p.accumulate ? *c + acc[i][j][e] : acc[i][j][e];
                }
            }
}
static int g_train_matmul_vec = 1;   // ullsam_train_set_matmul_vec: 16-byte fetches of the k-fastest operands of the bf16 product (0: per-element fetches, the round-4 form; same bits)
extern "C" int ullsam_train_set_matmul_vec(int on) { const int old = g_train_matmul_vec; g_train_matmul_vec = on; return old; }
template <bool AK, bool BN>
static void launch_matmul_bf16(const MmArgs& a, hipStream_t stream) {
    constexpr int LDS = 2 * 2 * 128 * MMB_LD * 2;
    const dim3 grid((a.N + 127) / 128, (a.M + 127) / 128, a.batch * a.ksplit);
    // the vector form needs, of every k-fastest operand: unit k stride, rows / batch entries / head offsets / the k-split pieces on 16-byte boundaries, K in whole quads
    auto quad = [](long v) { return (v & 3) == 0; };
    const bool a_ok = !AK || (a.a_k == 1 && quad(a.a_m) && quad(a.a_b) && quad(a.a_h) && ((uintptr_t)a.A & 15) == 0);
    const bool b_ok = BN || (a.b_k == 1 && quad(a.b_n) && quad(a.b_b) && quad(a.b_h) && ((uintptr_t)a.B & 15) == 0);
    if constexpr (AK || !BN) {
        if (g_train_matmul_vec && a_ok && b_ok && quad(a.K) && quad(a.kc)) {
            matmul_bf16_mfma_kernel<AK, BN, true><<<grid, 256, LDS, stream>>>(a);
            return;
        }
    }
    matmul_bf16_mfma_kernel<AK, BN, false><<<grid, 256, LDS, stream>>>(a);
}
static int g_train_matmul_mfma = 1;
extern "C" int ullsam_train_set_matmul_mfma(int on) { const int old = g_train_matmul_mfma; g_train_matmul_mfma = on; return old; }
// C[i] (+)= sum over splits s (in order: deterministic) of part[s][i]
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, float* __restrict__ C, long n, int ksplit, int M, int N, int batch,
                                                            long c_b, long c_m, long c_n, int accumulate) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int k = 0; k < ksplit; ++k) s += part[(long)k * n + i];
    const int nn = (int)(i % N), m = (int)((i / N) % M), b = (int)(i / ((long)M * N));
    float* c = C + (long)b * c_b + (long)m * c_m + (long)nn * c_n;
    *c = accumulate ? *c + s : s;
}
static int train_matmul_launch(MmArgs a, hipStream_t st) {
    if (g_train_matmul_mfma && a.M >= 64 && a.N >= 48 && a.K >= 16) {
        const bool ak = a.a_k <= a.a_m, bn = a.b_n <= a.b_k;
        if (ak && bn) launch_matmul_mfma<true, true>(a, st);
        else if (ak) launch_matmul_mfma<true, false>(a, st);
        else if (bn) launch_matmul_mfma<false, true>(a, st);
        else launch_matmul_mfma<false, false>(a, st);
    } else {
        matmul_f32_kernel<<<dim3((a.N + 63) / 64, (a.M + 63) / 64, a.batch * a.ksplit), 256, 0, st>>>(a);
    }
    ULLSAM_LAUNCH_CHECK();
    return 0;
}
extern "C" int ullsam_train_matmul(const float* A, const float* B, float* C, int M, int N, int K, int batch, long a_b, long a_m, long a_k,
                                   long b_b, long b_k, long b_n, long c_b, long c_m, long c_n, int accumulate, void* stream) {
    ULLSAM_CHECK(M > 0 && N > 0 && K > 0 && batch > 0 && batch < 65536, "train_matmul: M=%d N=%d K=%d batch=%d", M, N, K, batch);
    MmArgs a{A, B, C, M, N, K, batch, a_b, a_m, a_k, b_b, b_k, b_n, c_b, c_m, c_n, accumulate, 1, K, 0, 1, 1, 1, 0, 0, 0, 0};
    return train_matmul_launch(a, reinterpret_cast<hipStream_t>(stream));
}
// fp32 operands rounded to bf16 at the product's door (see matmul_bf16_mfma_kernel); shapes below the MFMA tile fall back to the fp32 product
extern "C" int ullsam_train_matmul_bf16(const float* A, const float* B, float* C, int M, int N, int K, int batch, long a_b, long a_m, long a_k,
                                        long b_b, long b_k, long b_n, long c_b, long c_m, long c_n, int accumulate, void* stream) {
    ULLSAM_CHECK(M > 0 && N > 0 && K > 0 && batch > 0 && batch < 65536, "train_matmul_bf16: M=%d N=%d K=%d batch=%d", M, N, K, batch);
    MmArgs a{A, B, C, M, N, K, batch, a_b, a_m, a_k, b_b, b_k, b_n, c_b, c_m, c_n, accumulate, 1, K, 0, 1, 1, 1, 0, 0, 0, 0};
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (!(M >= 64 && N >= 48 && K >= 16)) return train_matmul_launch(a, st);
    const bool ak = a_k <= a_m, bn = b_n <= b_k;
    if (ak && bn) launch_matmul_bf16<true, true>(a, st);
    else if (ak) launch_matmul_bf16<true, false>(a, st);
    else if (bn) launch_matmul_bf16<false, true>(a, st);
    else launch_matmul_bf16<false, false>(a, st);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}
// The product over (outer, head) pairs whose operands live inside [rows, heads x hd] activations (the matrix-form attention of training.AttentionFn): entry (o, h) of
// operand X starts at o x_o + (h / x_hdiv) x_h -- q / k / v, dO and the gradients are read and written where they are, no head-major copies, no repeat_kv copies.
extern "C" int ullsam_train_matmul_heads(const float* A, const float* B, float* C, int M, int N, int K, int outer, int heads, long a_o, long a_h, int a_hdiv, long a_m,
                                         long a_k, long b_o, long b_h, int b_hdiv, long b_k, long b_n, long c_o, long c_h, long c_m, long c_n, int accumulate, int bf16,
                                         int tri, void* stream) {
    ULLSAM_CHECK(M > 0 && N > 0 && K > 0 && outer > 0 && heads > 0 && (long)outer * heads < 65536 && a_hdiv >= 1 && b_hdiv >= 1,
                 "train_matmul_heads: M=%d N=%d K=%d outer=%d heads=%d", M, N, K, outer, heads);
    ULLSAM_CHECK(tri >= 0 && tri <= 3 && (tri == 0 || M == N || M == K), "train_matmul_heads: tri=%d (a square attention: Sq == Sk)", tri);   // (the one-output-per-thread kernel ignores it: the clipped parts are zeros)
    MmArgs a{A, B, C, M, N, K, outer * heads, a_o, a_m, a_k, b_o, b_k, b_n, c_o, c_m, c_n, accumulate, 1, K, 0, heads, a_hdiv, b_hdiv, a_h, b_h, c_h, tri};
    if (heads == 1) { a.bin = 1; }                                            // (single level: entry b at b x_o)
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (!bf16 || !(M >= 64 && N >= 48 && K >= 16)) return train_matmul_launch(a, st);
    const bool ak = a_k <= a_m, bn = b_n <= b_k;
    if (ak && bn) launch_matmul_bf16<true, true>(a, st);
    else if (ak) launch_matmul_bf16<true, false>(a, st);
    else if (bn) launch_matmul_bf16<false, true>(a, st);
    else launch_matmul_bf16<false, false>(a, st);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}
// The same product with the k range cut into `ksplit` pieces that run as separate workgroups (products with a few output tiles and a long
// sum: the table gradients of the decomposed relative-position bias, the hypernetwork gradient over 65536 pixels).  `partial` holds
// ksplit * batch * M * N floats; the pieces are added in order by a second kernel, so the result does not depend on scheduling.
extern "C" int ullsam_train_matmul_splitk(const float* A, const float* B, float* C, int M, int N, int K, int batch, long a_b, long a_m, long a_k,
                                          long b_b, long b_k, long b_n, long c_b, long c_m, long c_n, int accumulate, int ksplit, float* partial, void* stream) {
    ULLSAM_CHECK(M > 0 && N > 0 && K > 0 && batch > 0 && ksplit >= 1 && (long)batch * ksplit < 65536 && partial, "train_matmul_splitk: M=%d N=%d K=%d batch=%d ksplit=%d", M, N, K, batch, ksplit);
    const int kc = ((K + ksplit - 1) / ksplit + 31) / 32 * 32;
    const int ns = (K + kc - 1) / kc;                      // every split owns at least one k
    const long n = (long)batch * M * N;
    MmArgs a{A, B, partial, M, N, K, batch, a_b, a_m, a_k, b_b, b_k, b_n, (long)M * N, (long)N, 1, 0, ns, kc, n, 1, 1, 1, 0, 0, 0, 0};
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int rc = train_matmul_launch(a, st);
    if (rc) return rc;
    splitk_reduce_kernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, st>>>(partial, C, n, ns, M, N, batch, c_b, c_m, c_n, accumulate);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// ---- out[c] += sum_r x[r][c]  (bias gradients, broadcast-parameter gradients); out is zeroed by the caller.  Row blocks write partial sums
// (partial: colsum_blocks(rows) x cols floats) that a second kernel adds in order: the result does not depend on scheduling -----------------------
static inline int colsum_blocks(long rows) { const long b = (rows + 63) / 64; return (int)(b < 64 ? b : 64); }   // (round 6: 64-row blocks instead of 256 -- a 4096 x 1280 operand is 5 x 64 = 320 workgroups instead of 80: colsum 40 -> ~15 us)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, float* __restrict__ out, float* __restrict__ partial, long rows, int cols,
                                                     long ld, long rows_per_block) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    const long r0 = (long)blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float s = 0.f;
    for (long r = r0; r < r1; ++r) s += x[r * ld + c];
    if (gridDim.y == 1) out[c] += s;
    else partial[(long)blockIdx.y * cols + c] = s;
}
__global__ __launch_bounds__(256) void partial_rows_reduce_kernel(const float* __restrict__ partial, float* __restrict__ out, int cols, int nb) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    float s = 0.f;
    for (int b = 0; b < nb; ++b) s += partial[(long)b * cols + c];
    out[c] += s;
}
extern "C" int ullsam_train_colsum(const float* x, float* out, long rows, int cols, long ld, float* partial, void* stream) {
    ULLSAM_CHECK(rows > 0 && cols > 0, "train_colsum: rows=%ld cols=%d", rows, cols);
    const int nb = colsum_blocks(rows);
    ULLSAM_CHECK(nb == 1 || partial, "train_colsum: %d row blocks need a partial buffer of %d x %d floats", nb, nb, cols);
    const long rpb = (rows + nb - 1) / nb;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    colsum_kernel<<<dim3((cols + 255) / 256, nb), 256, 0, st>>>(x, out, partial, rows, cols, ld, rpb);
    ULLSAM_LAUNCH_CHECK();
    if (nb > 1) {
        partial_rows_reduce_kernel<<<dim3((cols + 255) / 256), 256, 0, st>>>(partial, out, cols, nb);
        ULLSAM_LAUNCH_CHECK();
    }
    return 0;
}

// ---- LayerNorm backward (rows of D elements, biased variance, optional affine): one wave per row -----------------------------------
// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * w;  dw += sum_rows dy * xhat, db += sum_rows dy (zeroed by the caller): a second kernel,
// one thread per column and row block, from the (mean, rstd) pairs the first one leaves in ws; partials added in order (no atomics).
// ws: 2 rows + 2 colsum_blocks(rows) D floats (needed when dw or db is given)
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ dy,
                                                     float* __restrict__ dx, float* __restrict__ stats, long rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * D;
    const float* gr = dy + row * D;
    float s = 0.f;
    for (int i = lane; i < D; i += 64) s += xr[i];
    const float mean = wave_sum(s) / (float)D;
    float ss = 0.f;
    for (int i = lane; i < D; i += 64) { const float d = xr[i] - mean; ss += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)D + eps);
    float sg = 0.f, sgx = 0.f;
    for (int i = lane; i < D; i += 64) {
        const float xh = (xr[i] - mean) * rstd, g = gr[i] * (w ? w[i] : 1.f);
        sg += g; sgx += g * xh;
    }
    const float mg = wave_sum(sg) / (float)D, mgx = wave_sum(sgx) / (float)D;
    for (int i = lane; i < D; i += 64) {
        const float xh = (xr[i] - mean) * rstd, g = gr[i] * (w ? w[i] : 1.f);
        dx[row * D + i] = rstd * (g - mg - xh * mgx);
    }
    if (stats && lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}
__global__ __launch_bounds__(256) void ln_bwd_params_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ stats,
                                                            float* __restrict__ part, long rows, int D, long rows_per_block) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= D) return;
    const long r0 = (long)blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float a = 0.f, b = 0.f;
    for (long r = r0; r < r1; ++r) {
        const float g = dy[r * D + c];
        a += g * ((x[r * D + c] - stats[2 * r]) * stats[2 * r + 1]);
        b += g;
    }
    part[((long)blockIdx.y * 2) * D + c] = a;
    part[((long)blockIdx.y * 2 + 1) * D + c] = b;
}
__global__ __launch_bounds__(256) void ln_bwd_params_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, float* __restrict__ db, int D, int nb) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= D) return;
    float a = 0.f, b = 0.f;
    for (int k = 0; k < nb; ++k) { a += part[((long)k * 2) * D + c]; b += part[((long)k * 2 + 1) * D + c]; }
    if (dw) dw[c] += a;
    if (db) db[c] += b;
}
// The same row backward with the row in registers (D % 4 == 0, D <= 256 VPT: one wave per row, x and dy w fetched once as float4; the kernel above walks the row four times with
// 4-byte loads).  Same formulas; the row sums add a lane's VPT quads first.
template <int VPT>
__global__ __launch_bounds__(256) void ln_bwd_row_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ dy,
                                                         float* __restrict__ dx, float* __restrict__ stats, long rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * D;
    const float* gr = dy + row * D;
    float4 xv[VPT], gv[VPT];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const int i = 4 * (lane + 64 * j);
        if (i < D) {
            xv[j] = *reinterpret_cast<const float4*>(xr + i);
            const float4 g = *reinterpret_cast<const float4*>(gr + i);
            if (w) { const float4 ww = *reinterpret_cast<const float4*>(w + i); gv[j] = make_float4(g.x * ww.x, g.y * ww.y, g.z * ww.z, g.w * ww.w); } else gv[j] = g;
            s += (xv[j].x + xv[j].y) + (xv[j].z + xv[j].w);
        } else { xv[j] = gv[j] = make_float4(0.f, 0.f, 0.f, 0.f); }
    }
    const float mean = wave_sum(s) / (float)D;
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < VPT; ++j)
        if (4 * (lane + 64 * j) < D) { const float a = xv[j].x - mean, b = xv[j].y - mean, c = xv[j].z - mean, d = xv[j].w - mean; ss += (a * a + b * b) + (c * c + d * d); }
    const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)D + eps);
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int j = 0; j < VPT; ++j)
        if (4 * (lane + 64 * j) < D) {
            xv[j] = make_float4((xv[j].x - mean) * rstd, (xv[j].y - mean) * rstd, (xv[j].z - mean) * rstd, (xv[j].w - mean) * rstd);   // xhat
            sg += (gv[j].x + gv[j].y) + (gv[j].z + gv[j].w);
            sgx += (gv[j].x * xv[j].x + gv[j].y * xv[j].y) + (gv[j].z * xv[j].z + gv[j].w * xv[j].w);
        }
    const float mg = wave_sum(sg) / (float)D, mgx = wave_sum(sgx) / (float)D;
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const int i = 4 * (lane + 64 * j);
        if (i < D) *reinterpret_cast<float4*>(dx + row * D + i) = make_float4(rstd * (gv[j].x - mg - xv[j].x * mgx), rstd * (gv[j].y - mg - xv[j].y * mgx),
                                                                               rstd * (gv[j].z - mg - xv[j].z * mgx), rstd * (gv[j].w - mg - xv[j].w * mgx));
    }
    if (stats && lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}
extern "C" int ullsam_train_ln_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, long rows, int D, float eps,
                                   float* ws, void* stream) {
    ULLSAM_CHECK(rows > 0 && D > 0, "train_ln_bwd: rows=%ld D=%d", rows, D);
    ULLSAM_CHECK(!(dw || db) || ws, "train_ln_bwd: parameter gradients need the workspace (2 rows + 2 * %d * D floats)", colsum_blocks(rows));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bool vec = D % 4 == 0 && D <= 2048 && ((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)w)) & 15) == 0;
    if (vec && D <= 512) ln_bwd_row_kernel<2><<<dim3((unsigned)((rows + 3) / 4)), 256, 0, st>>>(x, w, dy, dx, (dw || db) ? ws : nullptr, rows, D, eps);
    else if (vec) ln_bwd_row_kernel<8><<<dim3((unsigned)((rows + 3) / 4)), 256, 0, st>>>(x, w, dy, dx, (dw || db) ? ws : nullptr, rows, D, eps);
    else ln_bwd_kernel<<<dim3((unsigned)((rows + 3) / 4)), 256, 0, st>>>(x, w, dy, dx, (dw || db) ? ws : nullptr, rows, D, eps);
    ULLSAM_LAUNCH_CHECK();
    if (dw || db) {
        const int nb = colsum_blocks(rows);
        const long rpb = (rows + nb - 1) / nb;
        float* part = ws + 2 * rows;
        ln_bwd_params_kernel<<<dim3((D + 255) / 256, nb), 256, 0, st>>>(x, dy, ws, part, rows, D, rpb);
        ULLSAM_LAUNCH_CHECK();
        ln_bwd_params_reduce_kernel<<<dim3((D + 255) / 256), 256, 0, st>>>(part, dw, db, D, nb);
        ULLSAM_LAUNCH_CHECK();
    }
    return 0;
}

// ---- activations: kind 1 = exact (erf) GELU, 2 = ReLU; fwd y = act(x), bwd dx = dy * act'(x) ---------------------------------------
__global__ __launch_bounds__(256) void act_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ out, long n, int kind) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    float r;
    if (kind == 1) {
        const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752f));
        r = dy ? dy[i] * (cdf + v * 0.3989422804014327f * expf(-0.5f * v * v)) : v * cdf;
    } else {
        r = dy ? (v > 0.f ? dy[i] : 0.f) : fmaxf(v, 0.f);
    }
    out[i] = r;
}
extern "C" int ullsam_train_act(const float* x, const float* dy, float* out, long n, int kind, void* stream) {
    ULLSAM_CHECK(n > 0 && (kind == 1 || kind == 2), "train_act: n=%ld kind=%d", n, kind);
    act_kernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(x, dy, out, n, kind);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// ---- y = x * s[0] + t[0] (prompt_encoder.py:148: llm_scale_factor / llm_bias); bwd dx = dy * s, ds += sum dy * x, dt += sum dy -------
__global__ __launch_bounds__(256) void scale_shift_kernel(const float* __restrict__ x, const float* __restrict__ s, const float* __restrict__ t,
                                                          const float* __restrict__ dy, float* __restrict__ out, float* __restrict__ ds,
                                                          float* __restrict__ dt, long n) {
    __shared__ float red[2][4];
    const long i0 = (long)blockIdx.x * 1024 + threadIdx.x;
    const float sc = s[0];
    float a = 0.f, b = 0.f;
    for (int j = 0; j < 4; ++j) {
        const long i = i0 + 256 * j;
        if (i >= n) break;
        if (!dy) { out[i] = x[i] * sc + t[0]; continue; }
        out[i] = dy[i] * sc;
        a += dy[i] * x[i];
        b += dy[i];
    }
    if (!dy) return;
    a = wave_sum(a); b = wave_sum(b);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {   // per-block partials; ordered_sum_kernel adds them in a fixed order
        ds[blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        dt[blockIdx.x] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}
// out[v] += sum_i in[v * stride + i], i < n, for v < nvec: one wave per vector, lane l sums the contiguous chunk l, then the fixed wave tree
__global__ __launch_bounds__(64) void ordered_sum_kernel(const float* __restrict__ in, float* __restrict__ out, long n, long stride) {
    const float* p = in + (long)blockIdx.x * stride;
    const long per = (n + 63) / 64, i0 = threadIdx.x * per, i1 = min(n, i0 + per);
    float s = 0.f;
    for (long i = i0; i < i1; ++i) s += p[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) out[blockIdx.x] += s;
}
extern "C" int ullsam_train_scale_shift(const float* x, const float* s, const float* t, const float* dy, float* out, float* ds, float* dt, long n,
                                        float* partial, void* stream) {
    ULLSAM_CHECK(n > 0 && (!dy || (ds && dt && partial)), "train_scale_shift: n=%ld (the backward needs 2 * ceil(n / 1024) floats of partials)", n);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const long nb = (n + 1023) / 1024;
    scale_shift_kernel<<<dim3((unsigned)nb), 256, 0, st>>>(x, s, t, dy, out, partial, partial ? partial + nb : nullptr, n);
    ULLSAM_LAUNCH_CHECK();
    if (dy) {
        ordered_sum_kernel<<<1, 64, 0, st>>>(partial, ds, nb, 0);
        ordered_sum_kernel<<<1, 64, 0, st>>>(partial + nb, dt, nb, 0);
        ULLSAM_LAUNCH_CHECK();
    }
    return 0;
}

// ---- softmax attention, forward and backward, for the training path ------------------------------------------------------------------
// out = softmax(q k^T * scale + mask) v per head.  Covers the decoder's attention (transformer.py:220-242: no mask) and InternLM2's
// (modeling_internlm2.py:383-419: grouped KV heads, the additive finfo.min causal + padding mask of :834-870).
// q / dq / out [B, Sq, H, hd], k, v / dk, dv [B, Sk, H / groups, hd] with element strides (batch, token, head); causal >= 0: key j is
// visible to query i iff j <= i + causal (causal = Sk - Sq); key_mask int32 [B, Sk] (0 = padding) or NULL.  One workgroup per
// (query, head, batch) computes its row of P; the backward recomputes it, writes dq directly and adds dk / dv by atomics (zeroed by the
// caller).  hd <= 128.
struct AttnTrainArgs {
    const float* q; const float* k; const float* v; const float* dout; float* out; float* dq; float* dk; float* dv;
    long q_bs, q_ts, q_hs, k_bs, k_ts, k_hs, v_bs, v_ts, v_hs, o_bs, o_ts, o_hs;
    int H, groups, Sq, Sk, hd, causal;
    const int* key_mask;
    float scale;
    // decomposed relative-position bias of the ViT (image_encoder.py:325-361): logits += bias_h[q][key / kw] + bias_w[q][key % kw];
    // bias_h [B, H, Sq, Sk / kw], bias_w [B, H, Sq, kw]; the backward writes their gradients (one row each per workgroup)
    const float* bias_h; const float* bias_w; float* dbias_h; float* dbias_w; int kw;
};
template <bool BWD>
__global__ __launch_bounds__(256) void attn_train_kernel(AttnTrainArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MAXHD = 128;
    float* sc = reinterpret_cast<float*>(smem);      // [Sk] scores -> exp
    float* qs = sc + ((p.Sk + 3) & ~3);              // [hd] q
    float* gs = qs + MAXHD;                          // [hd] dO (backward) / output accumulator (forward)
    float* dqs = gs + MAXHD;                         // [hd] dq accumulator
    float* red = dqs + MAXHD;                        // [256]
    float* dbs = red + 256;                          // [128 + 128] bias-gradient rows (backward with bias)
    const int tid = threadIdx.x, hd = p.hd;
    const int qi = blockIdx.x, head = blockIdx.y, b = blockIdx.z, kvh = head / p.groups;
    const int kh_n = p.bias_h ? p.Sk / p.kw : 0;
    const float* bh = p.bias_h ? p.bias_h + (((long)b * p.H + head) * p.Sq + qi) * kh_n : nullptr;
    const float* bw = p.bias_h ? p.bias_w + (((long)b * p.H + head) * p.Sq + qi) * p.kw : nullptr;
    if (BWD && p.bias_h) dbs[tid] = 0.f;
    const float* qp = p.q + (long)b * p.q_bs + (long)qi * p.q_ts + (long)head * p.q_hs;
    if (tid < hd) {
        qs[tid] = qp[tid];
        gs[tid] = BWD ? p.dout[(long)b * p.o_bs + (long)qi * p.o_ts + (long)head * p.o_hs + tid] : 0.f;
        dqs[tid] = 0.f;
    }
    __syncthreads();
    const float* kb = p.k + (long)b * p.k_bs + (long)kvh * p.k_hs;
    const float* vb = p.v + (long)b * p.v_bs + (long)kvh * p.v_hs;
    auto block_reduce = [&](float v, const bool is_max) -> float {
        red[tid] = v;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) red[tid] = is_max ? fmaxf(red[tid], red[tid + s]) : red[tid] + red[tid + s];
            __syncthreads();
        }
        const float r = red[0];
        __syncthreads();
        return r;
    };
    const float FMIN = -3.4028234663852886e38f;
    float mx = -INFINITY;
    for (int kt = tid; kt < p.Sk; kt += 256) {
        const float* kp = kb + (long)kt * p.k_ts;
        float acc = 0.f;
        for (int d = 0; d < hd; ++d) acc += qs[d] * kp[d];
        acc *= p.scale;
        if (bh) acc += bh[kt / p.kw] + bw[kt % p.kw];
        if (p.causal >= 0 && kt > qi + p.causal) acc += FMIN;                              // the reference's additive masks (two finfo.min add to -inf)
        if (p.key_mask && p.key_mask[(long)b * p.Sk + kt] == 0) acc += FMIN;
        sc[kt] = acc;
        mx = fmaxf(mx, acc);
    }
    mx = block_reduce(mx, true);
    float sum = 0.f;
    for (int kt = tid; kt < p.Sk; kt += 256) { const float e = mx == -INFINITY ? 0.f : expf(sc[kt] - mx); sc[kt] = e; sum += e; }
    sum = block_reduce(sum, false);
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;
    // P in place
    for (int kt = tid; kt < p.Sk; kt += 256) sc[kt] *= inv;
    __syncthreads();
    if constexpr (!BWD) {
        // out[d] = sum_j P_j v_j[d]: thread d walks the keys (this kernel serves the small problems; the large ones go through matrices)
        if (tid < hd) {
            float o = 0.f;
            for (int kt = 0; kt < p.Sk; ++kt) o += sc[kt] * vb[(long)kt * p.v_ts + tid];
            p.out[(long)b * p.o_bs + (long)qi * p.o_ts + (long)head * p.o_hs + tid] = o;
        }
        return;
    } else {
        float* dsl = sc + ((p.Sk + 3) & ~3) + 3 * MAXHD + 256 + 256;    // [Sk] dlogit * scale, behind the other LDS arrays
        // dP_j = dO . v_j;  D = sum_j P_j dP_j
        float dsum = 0.f;
        for (int kt = tid; kt < p.Sk; kt += 256) {
            const float* vp = vb + (long)kt * p.v_ts;
            float dp = 0.f;
            for (int d = 0; d < hd; ++d) dp += gs[d] * vp[d];
            dsl[kt] = dp;
            dsum += sc[kt] * dp;
        }
        const float D = block_reduce(dsum, false);
        float* dkb = p.dk + (long)b * p.k_bs + (long)kvh * p.k_hs;
        float* dvb = p.dv + (long)b * p.v_bs + (long)kvh * p.v_hs;
        for (int kt = tid; kt < p.Sk; kt += 256) {
            const float pj = sc[kt];
            const float dlogit = pj * (dsl[kt] - D);
            const float dsj = dlogit * p.scale;            // d loss / d (q . k_j)
            dsl[kt] = dsj;
            if (pj == 0.f) continue;
            if (bh) { atomicAdd(dbs + kt / p.kw, dlogit); atomicAdd(dbs + 128 + kt % p.kw, dlogit); }
            for (int d = 0; d < hd; ++d) {
                atomicAdd(dkb + (long)kt * p.k_ts + d, dsj * qs[d]);
                atomicAdd(dvb + (long)kt * p.v_ts + d, pj * gs[d]);
            }
        }
        __syncthreads();
        if (tid < hd) {                                    // dq[d] = sum_j dS_j k_j[d]
            float a = 0.f;
            for (int kt = 0; kt < p.Sk; ++kt) a += dsl[kt] * kb[(long)kt * p.k_ts + tid];
            p.dq[(long)b * p.q_bs + (long)qi * p.q_ts + (long)head * p.q_hs + tid] = a;
        }
        if (bh) {
            if (tid < kh_n) p.dbias_h[(((long)b * p.H + head) * p.Sq + qi) * kh_n + tid] = dbs[tid];
            if (tid < p.kw) p.dbias_w[(((long)b * p.H + head) * p.Sq + qi) * p.kw + tid] = dbs[128 + tid];
        }
    }
}
extern "C" int ullsam_train_attention(const float* q, const float* k, const float* v, const float* dout, float* out, float* dq, float* dk,
                                      float* dv, int B, int H, int groups, int hd, int Sq, int Sk, int causal, const int* key_mask, long q_bs,
                                      long q_ts, long q_hs, long k_bs, long k_ts, long k_hs, long v_bs, long v_ts, long v_hs, long o_bs,
                                      long o_ts, long o_hs, float scale, const float* bias_h, const float* bias_w, float* dbias_h,
                                      float* dbias_w, int kw, void* stream) {
    ULLSAM_CHECK(!bias_h || (bias_w && kw > 0 && kw <= 128 && Sk % kw == 0 && Sk / kw <= 128 && (!dout || (dbias_h && dbias_w))),
                 "train_attention: decomposed bias needs Sk = kh * kw with kh, kw <= 128 (Sk=%d kw=%d)", Sk, kw);
    ULLSAM_CHECK(hd > 0 && hd <= 128 && Sk > 0 && Sk <= 16000 && Sq > 0 && B > 0 && B < 65536 && H > 0 && H < 65536 && groups > 0 && H % groups == 0,
                 "train_attention: hd=%d Sq=%d Sk=%d H=%d groups=%d", hd, Sq, Sk, H, groups);
    ULLSAM_CHECK((dout != nullptr) == (dq != nullptr) && (dout != nullptr || out != nullptr), "train_attention: forward needs out, backward needs dout / dq / dk / dv");
    AttnTrainArgs a{q, k, v, dout, out, dq, dk, dv, q_bs, q_ts, q_hs, k_bs, k_ts, k_hs, v_bs, v_ts, v_hs, o_bs, o_ts, o_hs, H, groups, Sq, Sk, hd, causal, key_mask, scale, bias_h, bias_w, dbias_h, dbias_w, kw};
    const size_t lds = (size_t)(2 * ((Sk + 3) & ~3) + 128 * 3 + 256 + 256) * 4;
    static PerDeviceOnce attr;
    if (attr.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_train_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_train_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dout) attn_train_kernel<true><<<dim3(Sq, H, B), 256, lds, s>>>(a);
    else attn_train_kernel<false><<<dim3(Sq, H, B), 256, lds, s>>>(a);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// ---- attention backward, materialised form: one wave-row pass between batched matmuls ---------------------------------------------------------
// The backward above adds dk / dv by atomics -- every query's workgroup into the same Sk x hd rows: 1.4 s for one global ViT-B block
// (4096 x 4096 x 12 heads).  Large problems go through matrices instead (ullsam_amd/training.py AttentionFn.backward):
//   forward: S = (q * scale) k^T by ullsam_train_matmul, THIS kernel per row with dP = NULL: logits = S + bias + masks, P = softmax(logits) in place
//   (kept for the backward), out = P v by ullsam_train_matmul;
//   backward: dP = dO v^T, THIS kernel per row with have_p: D = sum_j P_j dP_j, dS = P (dP - D) overwrites dP, the decomposed-bias gradient rows
//   are reduced on the way; then dV = P^T dO, dQ = scale * dS k, dK = dS^T (q * scale).  No global atomics, sums in a fixed order.
// S / dP [BH, Sq, Sk] (BH = batch * heads, b-major); bias_h [BH, Sq, Sk / kw], bias_w [BH, Sq, kw] or NULL; key_mask int32 [B, Sk] or NULL.
__global__ __launch_bounds__(256) void attn_rows_bwd_kernel(float* __restrict__ S, float* __restrict__ dP, const float* __restrict__ bias_h,
                                                            const float* __restrict__ bias_w, float* __restrict__ dbias_h, float* __restrict__ dbias_w,
                                                            const int* __restrict__ key_mask, int H, int Sq, int Sk, int kw, int causal, int have_p) {
    __shared__ float red[256];
    __shared__ float dbs[256];
    const int tid = threadIdx.x, qi = blockIdx.x;
    const long bh = blockIdx.y, row = bh * Sq + qi;
    const int b = (int)(bh / H);
    float* s = S + row * Sk;
    float* g = dP + row * Sk;
    const int kh_n = bias_h ? Sk / kw : 0;
    const float* bhp = bias_h ? bias_h + row * kh_n : nullptr;
    const float* bwp = bias_h ? bias_w + row * kw : nullptr;
    auto block_reduce = [&](float v, const bool is_max) -> float {
        red[tid] = v;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if (tid < st) red[tid] = is_max ? fmaxf(red[tid], red[tid + st]) : red[tid] + red[tid + st];
            __syncthreads();
        }
        const float r = red[0];
        __syncthreads();
        return r;
    };
    const float FMIN = -3.4028234663852886e38f;
    float inv = 1.f;
    if (!have_p) {   // S holds (q * scale) k^T: add bias and masks, softmax in place
        float mx = -INFINITY;
        for (int kt = tid; kt < Sk; kt += 256) {
            const bool behind = causal >= 0 && kt > qi + causal;   // (not read: the products do not form the tiles that hold only such entries; s + FMIN is FMIN for every finite score)
            float v = behind ? FMIN : s[kt];
            if (bhp) v += bhp[kt / kw] + bwp[kt % kw];
            if (key_mask && key_mask[(long)b * Sk + kt] == 0) v += FMIN;
            s[kt] = v;
            mx = fmaxf(mx, v);
        }
        mx = block_reduce(mx, true);
        float sum = 0.f;
        for (int kt = tid; kt < Sk; kt += 256) { const float e = mx == -INFINITY ? 0.f : expf(s[kt] - mx); s[kt] = e; sum += e; }
        sum = block_reduce(sum, false);
        inv = sum > 0.f ? 1.0f / sum : 0.f;
    }
    if (!dP) {   // forward use: P only
        for (int kt = tid; kt < Sk; kt += 256) s[kt] *= inv;
        return;
    }
    float ds = 0.f;
    for (int kt = tid; kt < Sk; kt += 256) {
        const float pj = s[kt] * inv;
        s[kt] = pj;
        if (causal >= 0 && kt > qi + causal) g[kt] = 0.f;        // (dP behind the diagonal may never have been written)
        ds += pj * g[kt];
    }
    const float D = block_reduce(ds, false);
    if (bhp) dbs[tid] = 0.f;
    __syncthreads();
    for (int kt = tid; kt < Sk; kt += 256) {
        const float dl = s[kt] * (g[kt] - D);
        g[kt] = dl;
        if (bhp) { atomicAdd(dbs + kt / kw, dl); atomicAdd(dbs + 128 + kt % kw, dl); }   // LDS, this row only
    }
    if (bhp) {
        __syncthreads();
        if (tid < kh_n) dbias_h[row * kh_n + tid] = dbs[tid];
        if (tid < kw) dbias_w[row * kw + tid] = dbs[128 + tid];
    }
}
// The same row pass with the row held in registers (Sk <= 64 NW VPT): one read and one write of S / dP instead of three of each, wave
// reductions, 4 / NW rows per workgroup (a 196-key window row is one wave's work, a 4096-key row four waves'), and the decomposed-bias gradient rows
// summed from an LDS copy of dS in a fixed order (no LDS atomics).  Arithmetic and its order per element are those of the kernel above
// except for the order of the row sums.
template <int NW, int VPT>
__global__ __launch_bounds__(256) void attn_rows_reg_kernel(float* __restrict__ S, float* __restrict__ dP, const float* __restrict__ bias_h,
                                                            const float* __restrict__ bias_w, float* __restrict__ dbias_h, float* __restrict__ dbias_w,
                                                            const int* __restrict__ key_mask, int H, int Sq, int Sk, int kw, int causal, int have_p, long rows) {
    constexpr int RPB = 4 / NW, T = 64 * NW;
    extern __shared__ float rows_lds[];                    // [RPB][T * VPT] dS of the rows in flight (bias gradients only)
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, rib = wave / NW, tix = (wave % NW) * 64 + lane;
    const long row = (long)blockIdx.x * RPB + rib;
    const bool live = row < rows;
    const int qi = live ? (int)(row % Sq) : 0;
    const long bh = live ? row / Sq : 0;
    const int b = (int)(bh / H);
    float* s = S + row * Sk;
    float* g = dP + row * Sk;
    const int kh_n = bias_h ? Sk / kw : 0;
    const float* bhp = bias_h ? bias_h + row * kh_n : nullptr;
    const float* bwp = bias_h ? bias_w + row * kw : nullptr;
    auto row_reduce = [&](float v, const bool is_max) -> float {
        v = is_max ? wave_max(v) : wave_sum(v);
        if (NW == 1) return v;
        __syncthreads();
        if (lane == 0) red[wave] = v;
        __syncthreads();
        return is_max ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) : (red[0] + red[1]) + (red[2] + red[3]);
    };
    const float FMIN = -3.4028234663852886e38f;
    float p[VPT];
    if (!have_p) {
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int kt = tix + T * i;
            float v = -INFINITY;
            if (live && kt < Sk) {
                v = (causal >= 0 && kt > qi + causal) ? FMIN : s[kt];   // (entries behind the diagonal are not read: their tiles may never have been formed; s + FMIN is FMIN for every finite score)
                if (bhp) v += bhp[kt / kw] + bwp[kt % kw];
                if (key_mask && key_mask[(long)b * Sk + kt] == 0) v += FMIN;
            }
            p[i] = v;
            mx = fmaxf(mx, v);
        }
        mx = row_reduce(mx, true);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < VPT; ++i) { p[i] = (mx == -INFINITY || p[i] == -INFINITY) ? 0.f : expf(p[i] - mx); sum += p[i]; }
        sum = row_reduce(sum, false);
        const float inv = sum > 0.f ? 1.0f / sum : 0.f;
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            p[i] *= inv;
            const int kt = tix + T * i;
            if (live && kt < Sk) s[kt] = p[i];
        }
    } else {
#pragma unroll
        for (int i = 0; i < VPT; ++i) { const int kt = tix + T * i; p[i] = (live && kt < Sk) ? s[kt] : 0.f; }
    }
    if (!dP) return;
    float gr[VPT], ds = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; ++i) { const int kt = tix + T * i; gr[i] = (live && kt < Sk && !(causal >= 0 && kt > qi + causal)) ? g[kt] : 0.f; ds += p[i] * gr[i]; }
    const float D = row_reduce(ds, false);
    float* mine = rows_lds + rib * (T * VPT);
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int kt = tix + T * i;
        const float dl = p[i] * (gr[i] - D);
        if (live && kt < Sk) g[kt] = dl;
        if (bhp) mine[kt] = (live && kt < Sk) ? dl : 0.f;
    }
    if (bhp) {
        __syncthreads();
        if (live)
            for (int j = tix; j < kh_n + kw; j += T) {
                float a = 0.f;
                if (j < kh_n) { for (int c = 0; c < kw; ++c) a += mine[j * kw + c]; dbias_h[row * kh_n + j] = a; }
                else { const int c = j - kh_n; for (int r = 0; r < kh_n; ++r) a += mine[r * kw + c]; dbias_w[row * kw + c] = a; }
            }
    }
}
template <int NW, int VPT>
static void launch_attn_rows_reg(float* S, float* dP, const float* bias_h, const float* bias_w, float* dbias_h, float* dbias_w, const int* key_mask,
                                 int B, int H, int Sq, int Sk, int kw, int causal, int have_p, hipStream_t st) {
    constexpr int RPB = 4 / NW;
    const long rows = (long)B * H * Sq;
    const int lds = (bias_h && dP) ? RPB * 64 * NW * VPT * 4 : 0;
    attn_rows_reg_kernel<NW, VPT><<<dim3((unsigned)((rows + RPB - 1) / RPB)), 256, lds, st>>>(S, dP, bias_h, bias_w, dbias_h, dbias_w, key_mask, H, Sq, Sk, kw, causal, have_p, rows);
}
static int g_train_rows_reg = 1;
extern "C" int ullsam_train_set_rows_reg(int on) { const int old = g_train_rows_reg; g_train_rows_reg = on; return old; }
extern "C" int ullsam_train_attn_rows(float* S, float* dP, const float* bias_h, const float* bias_w, float* dbias_h, float* dbias_w,
                                      const int* key_mask, int B, int H, int Sq, int Sk, int kw, int causal, int have_p, void* stream) {
    ULLSAM_CHECK(B > 0 && H > 0 && (long)B * H < 65536 && Sq > 0 && Sk > 0, "train_attn_rows: bad dims");
    ULLSAM_CHECK(!bias_h || (bias_w && (!dP || (dbias_h && dbias_w)) && kw > 0 && kw <= 128 && Sk % kw == 0 && Sk / kw <= 128), "train_attn_rows: bias needs Sk = kh * kw, kh, kw <= 128");
    if (g_train_rows_reg && Sk <= 4096) {
        hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define ROWS_REG(NW, VPT) launch_attn_rows_reg<NW, VPT>(S, dP, bias_h, bias_w, dbias_h, dbias_w, key_mask, B, H, Sq, Sk, kw, causal, have_p, st)
        if (Sk <= 256) ROWS_REG(1, 4);
        else if (Sk <= 1024) ROWS_REG(4, 4);
        else if (Sk <= 2048) ROWS_REG(4, 8);
        else ROWS_REG(4, 16);
#undef ROWS_REG
        ULLSAM_LAUNCH_CHECK();
        return 0;
    }
    attn_rows_bwd_kernel<<<dim3(Sq, B * H), 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(S, dP, bias_h, bias_w, dbias_h, dbias_w, key_mask, H, Sq, Sk, kw, causal, have_p);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// ---- RMSNorm backward (modeling_internlm2.py:75-89): y = x * rsqrt(mean(x^2) + eps) * w; one wave per row; dw (optional) by atomics ------
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ dy,
                                                          float* __restrict__ dx, float* __restrict__ dw, long rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * D;
    const float* gr = dy + row * D;
    float ss = 0.f;
    for (int i = lane; i < D; i += 64) ss += xr[i] * xr[i];
    const float rstd = rsqrtf(wave_sum(ss) / (float)D + eps);
    float sgx = 0.f;
    for (int i = lane; i < D; i += 64) sgx += gr[i] * w[i] * xr[i] * rstd;
    const float mgx = wave_sum(sgx) / (float)D;
    for (int i = lane; i < D; i += 64) {
        const float xh = xr[i] * rstd;
        dx[row * D + i] = rstd * (gr[i] * w[i] - xh * mgx);
        if (dw) atomicAdd(dw + i, gr[i] * xh);
    }
}
// The same backward with one WORKGROUP per row and the row held in registers (D % 4 == 0, D <= 1024 VPT = 4096): x and dy * w are fetched once as float4 (the kernel above walks the
// row three times with 4-byte loads from one wave: 70 us for 1081 x 4096, latency-bound at one wave per SIMD); sums: wave, then the four waves in order.
template <int VPT>
__global__ __launch_bounds__(256) void rmsnorm_bwd_row_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ dy,
                                                              float* __restrict__ dx, float* __restrict__ dw, int D, float eps) {
    __shared__ float red[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long row = blockIdx.x;
    const float* xr = x + row * D;
    const float* gr = dy + row * D;
    float4 xv[VPT], gv[VPT];
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const int i = 4 * (tid + 256 * j);
        if (i < D) {
            xv[j] = *reinterpret_cast<const float4*>(xr + i);
            const float4 g = *reinterpret_cast<const float4*>(gr + i), ww = *reinterpret_cast<const float4*>(w + i);
            gv[j] = make_float4(g.x * ww.x, g.y * ww.y, g.z * ww.z, g.w * ww.w);
            ss += (xv[j].x * xv[j].x + xv[j].y * xv[j].y) + (xv[j].z * xv[j].z + xv[j].w * xv[j].w);
        } else { xv[j] = gv[j] = make_float4(0.f, 0.f, 0.f, 0.f); }
    }
    ss = wave_sum(ss);
    if (lane == 0) red[0][wave] = ss;
    __syncthreads();
    const float rstd = rsqrtf(((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / (float)D + eps);
    float sgx = 0.f;
#pragma unroll
    for (int j = 0; j < VPT; ++j) sgx += ((gv[j].x * xv[j].x + gv[j].y * xv[j].y) + (gv[j].z * xv[j].z + gv[j].w * xv[j].w)) * rstd;
    sgx = wave_sum(sgx);
    if (lane == 0) red[1][wave] = sgx;
    __syncthreads();
    const float mgx = ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / (float)D;
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const int i = 4 * (tid + 256 * j);
        if (i >= D) continue;
        const float4 xh = make_float4(xv[j].x * rstd, xv[j].y * rstd, xv[j].z * rstd, xv[j].w * rstd);
        *reinterpret_cast<float4*>(dx + row * D + i) = make_float4(rstd * (gv[j].x - xh.x * mgx), rstd * (gv[j].y - xh.y * mgx), rstd * (gv[j].z - xh.z * mgx), rstd * (gv[j].w - xh.w * mgx));
        if (dw) {   // (dw = sum_rows dy * xhat: dy = gv / w is not kept -- re-read)
            const float4 g = *reinterpret_cast<const float4*>(gr + i);
            atomicAdd(dw + i, g.x * xh.x); atomicAdd(dw + i + 1, g.y * xh.y); atomicAdd(dw + i + 2, g.z * xh.z); atomicAdd(dw + i + 3, g.w * xh.w);
        }
    }
}
extern "C" int ullsam_train_rmsnorm_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, long rows, int D, float eps, void* stream) {
    ULLSAM_CHECK(rows > 0 && D > 0, "train_rmsnorm_bwd: rows=%ld D=%d", rows, D);
    if (D % 4 == 0 && D <= 4096 && rows < (1l << 31) && ((((uintptr_t)x | (uintptr_t)w | (uintptr_t)dy | (uintptr_t)dx)) & 15) == 0) {   // (the widths ullsam_norm's forward takes)
        rmsnorm_bwd_row_kernel<4><<<dim3((unsigned)rows), 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(x, w, dy, dx, dw, D, eps);
        ULLSAM_LAUNCH_CHECK();
        return 0;
    }
    rmsnorm_bwd_kernel<<<dim3((unsigned)((rows + 3) / 4)), 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(x, w, dy, dx, dw, rows, D, eps);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// ---- rotary embedding (modeling_internlm2.py:233-247) on rows [B*S, heads, hd]: out = x cos + rotate_half(x) sin, rotate_half = cat(-x2, x1);
// adjoint != 0: the transpose of that map (the backward).  cos / sin tables fp32 [tab_rows, hd] (cat(freqs, freqs)), pos int32 [B*S] ----------
__global__ __launch_bounds__(256) void rope_train_kernel(const float* __restrict__ x, const int* __restrict__ pos, const float* __restrict__ cosT,
                                                         const float* __restrict__ sinT, float* __restrict__ out, long tokens, int heads, int hd,
                                                         int tab_rows, int adjoint) {
    const int half = hd >> 1;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= tokens * heads * half) return;
    const int d = (int)(i % half);
    const long th = i / half, tok = th / heads;
    const int p = min(max(pos[tok], 0), tab_rows - 1);
    const float c1 = cosT[(long)p * hd + d], c2 = cosT[(long)p * hd + half + d], s1 = sinT[(long)p * hd + d], s2 = sinT[(long)p * hd + half + d];
    const float a = x[th * hd + d], b = x[th * hd + half + d];
    if (!adjoint) { out[th * hd + d] = a * c1 - b * s1; out[th * hd + half + d] = b * c2 + a * s2; }
    else { out[th * hd + d] = a * c1 + b * s2; out[th * hd + half + d] = b * c2 - a * s1; }
}
extern "C" int ullsam_train_rope(const float* x, const int* pos, const float* cosT, const float* sinT, float* out, long tokens, int heads, int hd,
                                 int tab_rows, int adjoint, void* stream) {
    ULLSAM_CHECK(tokens > 0 && heads > 0 && hd > 0 && hd % 2 == 0 && tab_rows > 0, "train_rope: bad dims");
    const long n = tokens * heads * (hd / 2);
    rope_train_kernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(x, pos, cosT, sinT, out, tokens, heads, hd, tab_rows, adjoint);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// ---- SwiGLU (modeling_internlm2.py:598-618): out = silu(g) * u; backward dg = dy u (s + g s (1 - s)), du = dy silu(g), s = sigmoid(g) ------
__global__ __launch_bounds__(256) void swiglu_train_kernel(const float* __restrict__ g, const float* __restrict__ u, const float* __restrict__ dy,
                                                           float* __restrict__ out, float* __restrict__ dg, float* __restrict__ du, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gv = g[i], uv = u[i], s = 1.0f / (1.0f + expf(-gv));
    if (!dy) { out[i] = gv * s * uv; return; }
    dg[i] = dy[i] * uv * (s + gv * s * (1.0f - s));
    du[i] = dy[i] * gv * s;
}
extern "C" int ullsam_train_swiglu(const float* g, const float* u, const float* dy, float* out, float* dg, float* du, long n, void* stream) {
    ULLSAM_CHECK(n > 0 && (dy ? (dg && du) : out != nullptr), "train_swiglu: n=%ld", n);
    swiglu_train_kernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(g, u, dy, out, dg, du, n);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// ---- bilinear upsample backward (F.interpolate(align_corners=False), train_joint_v2.py:1073-1078): the adjoint of resize_bilinear_kernel, same taps
// (common.h tap_of), in GATHER form: one thread per input pixel adds the output pixels whose taps touch it, rows then columns in ascending
// order (no atomics: bit-reproducible).  An input row / column iy is touched by outputs y with tap i0 == iy or i1 == iy; for a scale s = ih / oh
// these lie within ((iy - 1) / s - 1, (iy + 2) / s + 1) ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resize_bwd_kernel(const float* __restrict__ dout, float* __restrict__ din, long planes, int ih, int iw, int oh, int ow) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= planes * ih * iw) return;
    const int ix = (int)(i % iw), iy = (int)((i / iw) % ih);
    const long pl = i / ((long)iw * ih);
    const float sy = (float)ih / (float)oh, sx = (float)iw / (float)ow;
    const int y0 = max(0, (int)floorf((float)(iy - 1) / sy) - 1), y1 = min(oh - 1, (int)ceilf((float)(iy + 2) / sy) + 1);
    const int x0 = max(0, (int)floorf((float)(ix - 1) / sx) - 1), x1 = min(ow - 1, (int)ceilf((float)(ix + 2) / sx) + 1);
    const float* g = dout + pl * oh * ow;
    float acc = 0.f;
    for (int y = y0; y <= y1; ++y) {
        const Tap ty = tap_of(y, sy, ih);
        const float wy = (ty.i0 == iy ? 1.f - ty.l : 0.f) + (ty.i1 == iy ? ty.l : 0.f);
        if (wy == 0.f) continue;
        float rowacc = 0.f;
        for (int x = x0; x <= x1; ++x) {
            const Tap tx = tap_of(x, sx, iw);
            const float wx = (tx.i0 == ix ? 1.f - tx.l : 0.f) + (tx.i1 == ix ? tx.l : 0.f);
            if (wx != 0.f) rowacc += g[(long)y * ow + x] * wx;
        }
        acc += rowacc * wy;
    }
    din[i] += acc;
}
extern "C" int ullsam_train_resize_bwd(const float* dout, float* din, long planes, int ih, int iw, int oh, int ow, void* stream) {
    ULLSAM_CHECK(planes > 0 && ih > 0 && iw > 0 && oh > 0 && ow > 0, "train_resize_bwd: bad dims");
    const long n = planes * ih * iw;
    resize_bwd_kernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(dout, din, planes, ih, iw, oh, ow);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// ---- BCE-with-logits (mean over pixels) + Dice (smooth 1e-7) per instance, mean over instances (train_joint_v2.py:605-661, 774-812) ------
// sums [P][4] = (sum bce, sum p t, sum p, sum t) per instance (zeroed by the caller); losses [3] = (total, bce, dice)
__global__ __launch_bounds__(256) void seg_loss_sums_kernel(const float* __restrict__ x, const float* __restrict__ t, float* __restrict__ sums, long npix) {
    __shared__ float red[4][4];
    const int inst = blockIdx.y;
    const long i0 = (long)blockIdx.x * 1024 + threadIdx.x;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < 4; ++j) {
        const long i = i0 + 256 * j;
        if (i >= npix) break;
        const float xv = x[inst * npix + i], tv = t[inst * npix + i];
        const float pr = 1.0f / (1.0f + expf(-xv));
        a[0] += fmaxf(xv, 0.f) - xv * tv + log1pf(expf(-fabsf(xv)));   // BCEWithLogits, the numerically stable form torch uses
        a[1] += pr * tv; a[2] += pr; a[3] += tv;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) { a[c] = wave_sum(a[c]); if ((threadIdx.x & 63) == 0) red[c][threadIdx.x >> 6] = a[c]; }
    __syncthreads();
    // per-block partials [inst][4][blocks], added in order by ordered_sum_kernel
    if (threadIdx.x < 4) sums[((long)inst * 4 + threadIdx.x) * gridDim.x + blockIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}
__global__ void seg_loss_final_kernel(const float* __restrict__ sums, float* __restrict__ losses, int P, long npix, float smooth) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float bce = 0.f, dice = 0.f;
    for (int i = 0; i < P; ++i) {
        const float* s = sums + i * 4;
        bce += s[0] / (float)npix;
        dice += 1.0f - (2.0f * s[1] + smooth) / (s[2] + s[3] + smooth);
    }
    losses[1] = bce / (float)P; losses[2] = dice / (float)P; losses[0] = losses[1] + losses[2];
}
__global__ __launch_bounds__(256) void seg_loss_bwd_kernel(const float* __restrict__ x, const float* __restrict__ t, const float* __restrict__ sums,
                                                           const float* __restrict__ gscale, float* __restrict__ dx, int P, long npix, float smooth) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int inst = blockIdx.y;
    if (i >= npix) return;
    const float* s = sums + inst * 4;
    const float num = 2.0f * s[1] + smooth, den = s[2] + s[3] + smooth;
    const float xv = x[inst * npix + i], tv = t[inst * npix + i];
    const float pr = 1.0f / (1.0f + expf(-xv));
    const float dbce = (pr - tv) / (float)npix;
    const float ddice = -(2.0f * tv * den - num) / (den * den) * pr * (1.0f - pr);
    dx[inst * npix + i] = gscale[0] * (dbce + ddice) / (float)P;
}
extern "C" int ullsam_train_seg_loss(const float* x, const float* t, float* sums, float* losses, int P, long npix, float smooth, float* partial, void* stream) {
    ULLSAM_CHECK(P > 0 && P < 65536 && npix > 0 && partial, "train_seg_loss: P=%d npix=%ld (partial: P * 4 * ceil(npix / 1024) floats)", P, npix);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long nb = (npix + 1023) / 1024;
    seg_loss_sums_kernel<<<dim3((unsigned)nb, P), 256, 0, s>>>(x, t, partial, npix);
    ULLSAM_LAUNCH_CHECK();
    ordered_sum_kernel<<<dim3(P * 4), 64, 0, s>>>(partial, sums, nb, nb);
    ULLSAM_LAUNCH_CHECK();
    seg_loss_final_kernel<<<1, 64, 0, s>>>(sums, losses, P, npix, smooth);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}
extern "C" int ullsam_train_seg_loss_bwd(const float* x, const float* t, const float* sums, const float* gscale, float* dx, int P, long npix,
                                         float smooth, void* stream) {
    ULLSAM_CHECK(P > 0 && P < 65536 && npix > 0, "train_seg_loss_bwd: P=%d npix=%ld", P, npix);
    seg_loss_bwd_kernel<<<dim3((unsigned)((npix + 255) / 256), P), 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(x, t, sums, gscale, dx, P, npix, smooth);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// ---- dst[idx[r]] += src[r] (rows of C floats): gradients of the point-label embedding table (prompt_encoder.py:76-96) and of the relative-position
// tables (image_encoder.py:303-322).  Gather form: one thread per destination element walks the index list in order (tables of 5 ... 127 rows,
// lists of a few thousand entries): no atomics, bit-reproducible (sixteen lanes per element, fixed tree) ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void index_add_rows_kernel(const float* __restrict__ src, const int* __restrict__ idx, float* __restrict__ dst, long rows, int C, int nrows_dst) {
    // (the index list goes through LDS a KiB-entry chunk at a time: walked straight from memory, every step of the loop was a dependent global load -- 50 us for a 196-entry list)
    // SIXTEEN lanes per destination element, lane s taking the list entries j = s (mod 16) in order, their sums added in a fixed tree (bit-reproducible): the 4096-entry lists of the
    // global blocks took 227 us with one thread per element
    __shared__ int sidx[1024];
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) >> 4;
    const int sub = threadIdx.x & 15;
    const bool live = i < (long)nrows_dst * C;
    const int k = live ? (int)(i / C) : -1, c = live ? (int)(i - (long)k * C) : 0;
    float a = 0.f;
    for (long r0 = 0; r0 < rows; r0 += 1024) {
        const int n = (int)min(1024l, rows - r0);
        __syncthreads();
        for (int j = threadIdx.x; j < n; j += 256) sidx[j] = idx[r0 + j];
        __syncthreads();
        if (live)
            for (int j = sub; j < n; j += 16)
                if (sidx[j] == k) a += src[(r0 + j) * C + c];
    }
    a += __shfl_xor(a, 8, 16); a += __shfl_xor(a, 4, 16); a += __shfl_xor(a, 2, 16); a += __shfl_xor(a, 1, 16);
    if (live && sub == 0) dst[i] += a;
}
extern "C" int ullsam_train_index_add_rows(const float* src, const int* idx, float* dst, long rows, int C, int nrows_dst, void* stream) {
    ULLSAM_CHECK(rows > 0 && C > 0 && nrows_dst > 0, "train_index_add_rows: bad dims");
    index_add_rows_kernel<<<dim3((unsigned)(((long)nrows_dst * C * 16 + 255) / 256)), 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(src, idx, dst, rows, C, nrows_dst);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// ---- adjoint of ullsam_im2col3x3 (the neck's 3x3 convolution, image_encoder.py:96-102, as im2col + Linear): dx[b,y,x,c] = sum over the nine
// taps of dcols[(b, y - ty + 1, x - tx + 1)][(ty*3 + tx)*C + c] -------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void col2im3x3_kernel(const float* __restrict__ dcols, float* __restrict__ dx, int B, int H, int W, int C) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * H * W * C) return;
    const int c = (int)(i % C);
    long t = i / C;
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H);
    const int b = (int)(t / H);
    float s = 0.f;
    for (int tap = 0; tap < 9; ++tap) {
        const int yo = y - tap / 3 + 1, xo = x - tap % 3 + 1;
        if (yo >= 0 && yo < H && xo >= 0 && xo < W) s += dcols[((((long)b * H + yo) * W + xo) * 9 + tap) * C + c];
    }
    dx[i] = s;
}
extern "C" int ullsam_train_col2im3x3(const float* dcols, float* dx, int B, int H, int W, int C, void* stream) {
    ULLSAM_CHECK(B > 0 && H > 0 && W > 0 && C > 0, "train_col2im3x3: bad dims");
    const long n = (long)B * H * W * C;
    col2im3x3_kernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(dcols, dx, B, H, W, C);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------
//  Cross entropy of the language-model head (InternLM2ForCausalLM.forward, modeling_internlm2.py:1084-1096: CrossEntropyLoss() over the shifted
//  logits, mean over the labels != ignore_index = -100).  One workgroup per row of fp32 logits [R, V]; every sum in a fixed order (strided partial per
//  thread, then a shared-memory tree): two runs give the same bits.  lse[r] = log sum exp of the row, loss_rows[r] = lse - logit[label] (0 for an ignored row).
// ------------------------------------------------------------------------------------------------------
template <bool MAX>
__device__ __forceinline__ float ce_block_reduce(float v, float* red) {
    const int t = threadIdx.x;
    red[t] = v;
    __syncthreads();
    for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
        if (t < s) red[t] = MAX ? fmaxf(red[t], red[t + s]) : red[t] + red[t + s];
        __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
}
__global__ __launch_bounds__(1024) void ce_rows_kernel(const float* __restrict__ x, long ld, const long long* __restrict__ labels, float* __restrict__ lse,
                                                       float* __restrict__ loss_rows, int V) {
    __shared__ float red[1024];
    const long r = blockIdx.x;
    const float* row = x + r * ld;
    float m = -INFINITY;
    for (int j = threadIdx.x; j < V; j += blockDim.x) m = fmaxf(m, row[j]);
    m = ce_block_reduce<true>(m, red);
    float s = 0.f;
    for (int j = threadIdx.x; j < V; j += blockDim.x) s += __expf(row[j] - m);
    s = ce_block_reduce<false>(s, red);
    if (threadIdx.x == 0) {
        const float l = m + __logf(s);
        const long long lab = labels[r];
        lse[r] = l;
        loss_rows[r] = (lab >= 0 && lab < V) ? l - row[lab] : 0.f;
    }
}
// out[0] = mean of loss_rows over the rows with a label (NaN if there is none, as torch), out[1] = 1 / (number of such rows) (0 if none): ordered sums in one workgroup
__global__ __launch_bounds__(1024) void ce_mean_kernel(const float* __restrict__ loss_rows, const long long* __restrict__ labels, long R, int V, float* __restrict__ out) {
    __shared__ float red[1024];
    float s = 0.f, n = 0.f;
    for (long r = threadIdx.x; r < R; r += blockDim.x) {
        const long long lab = labels[r];
        if (lab >= 0 && lab < V) { s += loss_rows[r]; n += 1.f; }
    }
    s = ce_block_reduce<false>(s, red);
    n = ce_block_reduce<false>(n, red);
    if (threadIdx.x == 0) {
        // no row with a label: the reference's CrossEntropyLoss(mean) divides 0 by 0 (modeling_internlm2.py:1084-1096) and `0 * loss + seg` is NaN there -- the same here
        // (the backward's scale 1 / n stays 0: every row is ignored, its gradient rows are zeros as torch's are)
        out[0] = n > 0.f ? s / n : __builtin_nanf("");
        out[1] = n > 0.f ? 1.f / n : 0.f;
    }
}
// dx[r][j] = (softmax(x[r])[j] - [j == label]) * g / n_valid   (0 for an ignored row)
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ x, long ld, const long long* __restrict__ labels, const float* __restrict__ lse,
                                                     const float* __restrict__ mean_out, const float* __restrict__ g, float* __restrict__ dx, long ldx, int V) {
    const long r = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ldx) return;
    const long long lab = labels[r];
    float d = 0.f;                              // columns V .. ldx - 1 (the padding that makes dlogits a GEMM operand) are written as zeros
    if (j < V && lab >= 0 && lab < V) d = (__expf(x[r * ld + j] - lse[r]) - (j == lab ? 1.f : 0.f)) * (g[0] * mean_out[1]);
    dx[r * ldx + j] = d;
}

extern "C" int ullsam_train_cross_entropy(const float* logits, long ld, const long long* labels, float* lse, float* loss_rows, float* out2, long rows, int V, void* stream) {
    ULLSAM_CHECK(rows > 0 && rows < (1L << 31) && V > 0 && ld >= V, "train_cross_entropy: rows=%ld V=%d ld=%ld", rows, V, ld);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ce_rows_kernel<<<dim3((unsigned)rows), 1024, 0, s>>>(logits, ld, labels, lse, loss_rows, V);
    ULLSAM_LAUNCH_CHECK();
    ce_mean_kernel<<<1, 1024, 0, s>>>(loss_rows, labels, rows, V, out2);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}
extern "C" int ullsam_train_cross_entropy_bwd(const float* logits, long ld, const long long* labels, const float* lse, const float* out2, const float* gscale, float* dlogits,
                                              long ldx, long rows, int V, void* stream) {
    ULLSAM_CHECK(rows > 0 && rows < 65536 && V > 0 && ld >= V && ldx >= V, "train_cross_entropy_bwd: rows=%ld V=%d ld=%ld ldx=%ld", rows, V, ld, ldx);
    ce_bwd_kernel<<<dim3((unsigned)((ldx + 255) / 256), (unsigned)rows), 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(logits, ld, labels, lse, out2, gscale, dlogits, ldx, V);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}
