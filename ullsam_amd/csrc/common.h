// Shared device helpers for the uLLSAM gfx950 kernels.
//
// Element types: activations/weights are either `float` (parity mode, exact fp32 MFMA) or
// `bf16` (throughput mode).  Both use ONE fragment convention so every MFMA kernel is a single
// template:
//   Frag<T> = 8 consecutive k-elements per lane.
//   32x32 tile, K=16 : lane l (r = l&31, h = l>>5) holds A[r][8h+j] / B[8h+j][r], j=0..7
//       bf16 -> one v_mfma_f32_32x32x16_bf16
//       f32  -> eight v_mfma_f32_32x32x2_f32 (MFMA j consumes k = {j, 8+j}); exact fp32 fma chain
//   16x16 tile, K=32 : lane l (r = l&15, g = l>>4) holds A[r][8g+j] / B[8g+j][r]
//       bf16 -> one v_mfma_f32_16x16x32_bf16
//       f32  -> eight v_mfma_f32_16x16x4_f32 (MFMA j consumes k = {8g+j : g=0..3})
// C/D layout is dtype independent on gfx950:
//   32x32: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
//   16x16: col = lane&15, row = 4*(lane>>4) + reg
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define ULLSAM_DT_F32 0
#define ULLSAM_DT_BF16 1

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

template <typename T> struct Frag;
template <> struct Frag<float> { float v[8]; };
template <> struct Frag<bf16> { bf16x8_t v; };

__device__ __forceinline__ void mma32(const Frag<bf16>& a, const Frag<bf16>& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, c, 0, 0, 0);
}
__device__ __forceinline__ void mma32(const Frag<float>& a, const Frag<float>& b, f32x16& c) {
#pragma unroll
    for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[j], b.v[j], c, 0, 0, 0);
}
__device__ __forceinline__ void mma16(const Frag<bf16>& a, const Frag<bf16>& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c, 0, 0, 0);
}
__device__ __forceinline__ void mma16(const Frag<float>& a, const Frag<float>& b, f32x4& c) {
#pragma unroll
    for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], c, 0, 0, 0);
}

// Load a fragment (8 consecutive elements) from a 16-byte aligned address (LDS or global).
__device__ __forceinline__ Frag<bf16> load_frag(const bf16* p) {
    Frag<bf16> f;
    f.v = *reinterpret_cast<const bf16x8_t*>(p);
    return f;
}
__device__ __forceinline__ Frag<float> load_frag(const float* p) {
    Frag<float> f;
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 b = *reinterpret_cast<const float4*>(p + 4);
    f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w;
    f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
    return f;
}
template <typename T> __device__ __forceinline__ Frag<T> zero_frag();
template <> __device__ __forceinline__ Frag<float> zero_frag<float>() {
    Frag<float> f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f.v[j] = 0.f;
    return f;
}
template <> __device__ __forceinline__ Frag<bf16> zero_frag<bf16>() {
    Frag<bf16> f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f.v[j] = (bf16)0.f;
    return f;
}

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16 x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float x) { return (bf16)x; }

// 4-element vector load/store helpers (T = float: 16 B, bf16: 8 B)
__device__ __forceinline__ float4 load4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 load4(const bf16* p) {
    const bf16x4_t v = *reinterpret_cast<const bf16x4_t*>(p);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
__device__ __forceinline__ void store4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void store4(bf16* p, float4 v) {
    bf16x4_t o;
    o[0] = (bf16)v.x; o[1] = (bf16)v.y; o[2] = (bf16)v.z; o[3] = (bf16)v.w;
    *reinterpret_cast<bf16x4_t*>(p) = o;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// lane ^ 16 / lane ^ 32 exchanges on gfx950's row-swap instructions (VALU ops: no ds_bpermute round trip through the LDS; semantics at attention.hip decode_attn)
typedef unsigned int uint2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float lane_xor16(float v) {
    const unsigned int u = __builtin_bit_cast(unsigned int, v);
    const uint2v r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __builtin_bit_cast(float, (threadIdx.x & 16) ? r.x : r.y);
}
__device__ __forceinline__ float lane_xor32(float v) {
    const unsigned int u = __builtin_bit_cast(unsigned int, v);
    const uint2v r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (threadIdx.x & 32) ? r.x : r.y);
}

// erf by Abramowitz-Stegun 7.1.26 (branch-free: 1 rcp, 1 exp, 6 fma).  In fp32 arithmetic |erf_as - erf| <= 6.1e-7 and the
// resulting exact-form GELU is within 4.7e-7 abs of 0.5*x*(1+erf(x/sqrt2)) over [-12, 12] (tools/check_erf.py) -- rounding-noise
// level for an fp32 GELU, at ~1/4 of libm erff's instruction count (which doubled the lin1 GEMM's time in its epilogue).
__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = 1.061405429f;
    p = fmaf(p, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    p *= t;
    const float r = fmaf(-p, __expf(-ax * ax), 1.0f);
    return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f)); }
// GELU for results that are rounded to bf16 (the ring GEMM's epilogue: 160 evaluations per thread of a 256x320 tile): Phi(x) = 0.5 erfc(-x / sqrt2)
// with erfc(z) = exp(z P5(z)) on 0 <= z <= 4 (least-squares fit of log erfc weighted for the absolute error; |erf| error <= 1.0e-6, GELU within
// 1.9e-6 abs of the erf form over [-12, 12] in fp32 arithmetic: tools/check_erf.py).  ONE quarter-rate transcendental (v_exp_f32) per value
// instead of erf_as's two (v_rcp_f32 + v_exp_f32), and the negative tail keeps its relative accuracy (no 1 - erf cancellation).
__device__ __forceinline__ float gelu_erfc5(float x) {
    const float z = fminf(fabsf(x) * 0.70710678118654752440f, 4.0f);
    float p = fmaf(-0.00303853428f, z, 0.0299264971f);   // log2(erfc(z)) / z
    p = fmaf(p, z, -0.149057642f);
    p = fmaf(p, z, -0.918339764f);
    p = fmaf(p, z, -1.62791028f);
    const float q = __builtin_amdgcn_exp2f(fmaf(p, z, -1.0f));   // 0.5 erfc(z)
    return x * (x > 0.f ? 1.0f - q : q);
}
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }

// row of a 32x32 accumulator register for lane-half h
__device__ __forceinline__ int crow32(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// Bilinear resampling (F.interpolate(mode="bilinear", align_corners=False); sam.py:154-162, app.py:635-640) shared by the resize kernel
// (decoder.hip) and the fused mask post-processing kernel (amg.hip).  Both must produce the SAME bits for the same tap -- the generator's
// fused path is checked for equality with the helper chain over 12288 masks x 4 M pixels, where a single a*b+c contracted to an fma in one
// kernel and not in the other flips a pixel that sits on the threshold -- so the arithmetic is written once, with contraction off.
struct Tap { int i0, i1; float l; };
__device__ __forceinline__ Tap tap_of(int o, float scale, int n_in) {
#pragma clang fp contract(off)
    float f = ((float)o + 0.5f) * scale - 0.5f;
    if (f < 0.f) f = 0.f;
    Tap t;
    t.i0 = min((int)f, n_in - 1);
    t.i1 = min(t.i0 + 1, n_in - 1);
    t.l = f - (float)t.i0;
    return t;
}
__device__ __forceinline__ float lerp_rn(float a, float b, float l) {   // a (1 - l) + b l: two products and a sum, each rounded once
#pragma clang fp contract(off)
    return a * (1.f - l) + b * l;
}

// hipFuncSetAttribute applies to the function object of the CURRENT device: one "done" flag per device (a process may drive
// several GPUs).  A benign race (two host threads setting the same attribute) is possible and harmless.
struct PerDeviceOnce {
    bool done[32] = {};
    bool first() {
        int d = 0;
        (void)hipGetDevice(&d);
        d &= 31;
        if (done[d]) return false;
        done[d] = true;
        return true;
    }
};

// Error plumbing for the C ABI (never throws; see include/ullsam_hip.h)
void ullsam_set_error(const char* fmt, ...);
#define ULLSAM_CHECK(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            ullsam_set_error(__VA_ARGS__); \
            return -1;                     \
        }                                  \
    } while (0)
#define ULLSAM_LAUNCH_CHECK()                                                      \
    do {                                                                           \
        hipError_t e_ = hipGetLastError();                                         \
        if (e_ != hipSuccess) {                                                    \
            ullsam_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
            return -2;                                                             \
        }                                                                          \
    } while (0)
