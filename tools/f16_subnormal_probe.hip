// f16_subnormal_probe.hip -- does v_mfma_f32_32x32x16_f16 on gfx950 honour f16 SUBNORMAL operands, and does the
// f32 -> f16 conversion produce them?  The split-precision mode (NRF_MMA_F16X3: x = hi + lo, lo = f16(x - f16(x)))
// puts the low parts of small values into the f16 subnormal range; if the matrix core flushed them, the mode would
// silently fall back to plain f16 accuracy for |x| < 0.125.
//   hipcc --offload-arch=gfx950 -O2 tools/f16_subnormal_probe.hip -o tools/f16_subnormal_probe && tools/f16_subnormal_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void probe(float a_val, float b_val, float* out) {
    // A = a_val in every element, B = b_val in every element: D[i][j] = 16 * a*b
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)a_val; b[j] = (_Float16)b_val; }
    f32x16 acc = {};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) {
        out[0] = acc[0];
        out[1] = (float)(_Float16)a_val;     // what the conversion produced
        out[2] = (float)(_Float16)b_val;
    }
}

int main() {
    float* d;
    hipMalloc(&d, 16);
    const float cases[][2] = {
        {ldexpf(1.0f, -20), 1024.0f},          // subnormal A (2^-20 < 2^-14), normal B: expect 16 * 2^-10
        {1024.0f, ldexpf(1.0f, -20)},          // subnormal B
        {ldexpf(1.0f, -24), 4096.0f},          // smallest subnormal
        {ldexpf(1.0f, -20), ldexpf(1.0f, -4)}, // product 2^-24 * 16 = 2^-20: small fp32 but normal
        {3.0e-6f, 1.0f},                       // rounds to a multiple of 2^-24
    };
    int bad = 0;
    for (auto& c : cases) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, c[0], c[1], d);
        float h[3];
        hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
        const float expect = 16.0f * h[1] * h[2];
        printf("a=%.9g (f16 %.9g)  b=%.9g (f16 %.9g)  mfma=%.9g  expect=%.9g  %s\n", c[0], h[1], c[1], h[2], h[0], expect,
               h[0] == expect && expect != 0.0f ? "ok" : "FLUSHED/DIFFERENT");
        bad += !(h[0] == expect && expect != 0.0f);
    }
    printf(bad ? "RESULT: f16 subnormals are NOT preserved\n" : "RESULT: f16 subnormals preserved by cvt and by the MFMA\n");
    return 0;
}
