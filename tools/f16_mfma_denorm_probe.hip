#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out, float a_val, float b_val) {
    // A = 32x16 (rows = lanes&31, k = 8 per lane half), B likewise; all entries a_val / b_val
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)a_val; b[i] = (_Float16)b_val; }
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = acc[0]; }
    f32x2 p = {a_val * 1.00048828125f, b_val};
    f16x2 h = __builtin_convertvector(p, f16x2);
    if (threadIdx.x == 0) { out[1] = (float)h[0]; out[2] = (float)h[1]; }
}
int main() {
    float* d; hipMalloc(&d, 64);
    float tests[][2] = {{1.0f, 1.0f}, {3.0e-5f, 1.0f}, {3.0e-5f, 3.0e-5f}, {1.0e-6f, 1000.0f}, {6.0e-8f, 1.0f}};
    for (auto& t : tests) {
        k<<<1, 64>>>(d, t[0], t[1]);
        float h[3]; hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
        double a16 = (double)(float)(_Float16)t[0], b16 = (double)(float)(_Float16)t[1];
        printf("a=%g (f16 %g) b=%g: mfma sum of 16 products = %.9g  expected %.9g   cvt: %.9g %.9g\n", t[0], a16, t[1], h[0], 16.0 * a16 * b16, h[1], h[2]);
    }
    return 0;
}
