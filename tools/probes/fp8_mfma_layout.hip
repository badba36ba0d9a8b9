#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
// One wave: D[16x16] = A[16x128] * B[128x16], fp8 e4m3, unit scales.  Host checks the lane maps with integer-valued fp8 data.
__global__ void k(const uint8_t* A, const uint8_t* B, float* D, int mapA) {
    const int l = threadIdx.x;
    i32x8 a, b;
    // hypothesis: lane l holds row (l&15), k = 32*(l>>4) .. +31, 32 consecutive bytes
    const uint8_t* ap = A + (l & 15) * 128 + (l >> 4) * 32;
    const uint8_t* bp = B + (l & 15) * 128 + (l >> 4) * 32;   // B stored [n][k]
    memcpy(&a, ap, 32); memcpy(&b, bp, 32);
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}
static uint8_t f2e4m3(float v) {  // small integers only
    if (v == 0) return 0;
    uint8_t s = v < 0 ? 0x80 : 0; v = v < 0 ? -v : v;
    int e = 0; while (v >= 2) { v /= 2; ++e; } while (v < 1) { v *= 2; --e; }
    int m = (int)((v - 1) * 8 + 0.5f);
    return s | ((e + 7) << 3) | m;
}
int main() {
    static uint8_t hA[16 * 128], hB[16 * 128];
    static float fA[16][128], fB[16][128];
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (int)((s >> 24) % 7) - 3; };
    for (int i = 0; i < 16; ++i) for (int k2 = 0; k2 < 128; ++k2) { fA[i][k2] = rnd(); fB[i][k2] = rnd(); hA[i * 128 + k2] = f2e4m3(fA[i][k2]); hB[i * 128 + k2] = f2e4m3(fB[i][k2]); }
    uint8_t *dA, *dB; float* dD; float hD[256];
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dD, 1024);
    hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, dD, 0);
    hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float r = 0; for (int k2 = 0; k2 < 128; ++k2) r += fA[i][k2] * fB[j][k2]; if (r != hD[i * 16 + j]) ++bad; }
    printf("fp8 16x16x128 scaled MFMA, consecutive-32-bytes-per-lane hypothesis: %d mismatches of 256 (D[0][0]=%g)\n", bad, hD[0]);
    return 0;
}
