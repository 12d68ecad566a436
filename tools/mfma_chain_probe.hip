// Probe: issue rate of v_mfma_f32_32x32x16_bf16 when consecutive MFMAs accumulate into the same registers
// (dependent chain) vs round-robin over independent accumulators.  hipcc --offload-arch=gfx950 -O3 -o probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(512) void probe(const bf16x8* in, f32x16* out, int iters, long long* cycles) {
    bf16x8 a = in[threadIdx.x], b = in[threadIdx.x + 512];
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 24; ++u) {
            const int k = MODE == 0 ? (u & 3) : MODE == 1 ? 0 : MODE == 2 ? (u / 3) & 3 : (u & 1);
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k], 0, 0, 0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    f32x16 s = acc[0];
    for (int i = 1; i < 4; ++i) for (int r = 0; r < 16; ++r) s[r] += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

int main(int argc, char** argv) {
    const bool zeros = argc > 1 && argv[1][0] == 'z';
    bf16x8* in; f32x16* out; long long* cyc;
    hipMalloc(&in, 1024 * sizeof(bf16x8));
    {   // random bf16 operands in [-2, 2): zero operands let the chip clock higher (DVFS) and flatter the result
        unsigned short* h = (unsigned short*)malloc(1024 * 16);
        unsigned x = 12345u;
        for (int i = 0; i < 1024 * 8; ++i) {
            x = x * 1664525u + 1013904223u;
            h[i] = zeros ? 0 : (unsigned short)(((x >> 9) & 0x807f) | 0x3f80 | ((x >> 3) & 0x8000));
        }
        hipMemcpy(in, h, 1024 * 16, hipMemcpyHostToDevice);
        free(h);
    }
    hipMalloc(&out, 256 * 512 * sizeof(f32x16)); hipMalloc(&cyc, 8);
    const int iters = 2000;
    const char* names[4] = {"4 accumulators round-robin", "1 accumulator (24-long chain)", "3 in a row per accumulator", "2 accumulators alternating"};
    for (int threads : {256, 512}) {
        for (int mode = 0; mode < 4; ++mode) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(threads), 0, 0, in, out, iters, cyc);
                if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(threads), 0, 0, in, out, iters, cyc);
                if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(256), dim3(threads), 0, 0, in, out, iters, cyc);
                if (mode == 3) hipLaunchKernelGGL(probe<3>, dim3(256), dim3(threads), 0, 0, in, out, iters, cyc);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            const double n = 24.0 * iters;
            const double tf = 2.0 * 32 * 32 * 16 * n * (threads / 64) * 256 / (ms * 1e-3) / 1e12;
            printf("%s waves/SIMD=%d  %-34s  %.1f ticks/MFMA/wave  %.3f ms  %.0f TFLOP/s  tick rate %.2f GHz\n", zeros ? "zeros " : "random",
                   threads / 256, names[mode], c / n, ms, tf, c / (ms * 1e-3) / 1e9);
        }
    }
    return 0;
}
