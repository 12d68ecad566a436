// Probe: ds_read_b128 throughput of an 8-wave workgroup for the lane -> address patterns of an MFMA A-operand fetch
// (lane (l31, h) reads 16 B; 16 reads per "tile step"), all of them conflict-free by the bank rule ((addr / 4) mod 64):
//   0  padded rows      addr = l31 * 272 + 16 h + 32 j          (K image of split_layout.h)
//   1  chunk-major      addr = l31 * 16 + 512 h + 1024 j        (r02_n experiment: attention +7 %)
//   2  padded rows 80   addr = l31 * 80 + 16 h + 32 j  (+ 2560 c)   (V^T image)
//   3  rows of 528      addr = l31 * 528 + 16 h + 32 j
//   4  chunk-major, XOR-rotated chunks   addr = ((l31 + 2 j) & 31) * 16 + 512 h + 1024 j
// hipcc --offload-arch=gfx950 -O3 -o lds_read_probe tools/lds_read_probe.hip && ./lds_read_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void probe(float* out, int iters, long long* cycles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, h = lane >> 5;
    for (int i = t; i < 65536 / 4; i += 512) reinterpret_cast<float*>(lds)[i] = (float)i;
    __syncthreads();
    int base;
    if (MODE == 0) base = l31 * 272 + 16 * h;
    else if (MODE == 1) base = l31 * 16 + 512 * h;
    else if (MODE == 2) base = l31 * 80 + 16 * h;
    else if (MODE == 3) base = l31 * 528 + 16 * h;
    else base = 512 * h;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const int stage = (it & 1) * 17408;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int a0, a1;
            if (MODE == 0 || MODE == 3) { a0 = base + 32 * j; a1 = a0 + (MODE == 0 ? 8704 : 16896); }
            else if (MODE == 1) { a0 = base + 1024 * j; a1 = a0 + 8192; }
            else if (MODE == 2) { a0 = base + 32 * (j & 1) + 2560 * (j >> 1); a1 = a0 + 10240; }
            else { a0 = base + ((l31 + 2 * j) & 31) * 16 + 1024 * j; a1 = a0 + 8192; }
            const f32x4 x = *reinterpret_cast<const f32x4*>(lds + stage + a0);
            const f32x4 y = *reinterpret_cast<const f32x4*>(lds + stage + a1);
            acc += x;
            acc += y;
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 512 + t] = acc[0] + acc[1] + acc[2] + acc[3];
    if (t == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

template <int MODE>
static void run(const char* name, float* out, long long* cyc) {
    const int iters = 4000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(512), 65536, 0, out, iters, cyc);
    hipDeviceSynchronize();
    long long c;
    hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    // per iteration: 8 waves x 16 ds_read_b128 x 1 KiB = 128 KiB per CU
    printf("%-28s %8.1f cycles per 16 reads of a wave  ->  %6.1f B/clk/CU\n", name, (double)c / iters, 131072.0 * iters / (double)c);
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    hipMalloc(&cyc, sizeof(long long));
    run<0>("padded rows 272 (K image)", out, cyc);
    run<1>("chunk-major 512 B runs", out, cyc);
    run<2>("padded rows 80 (V^T image)", out, cyc);
    run<3>("rows of 528", out, cyc);
    run<4>("chunk-major, rotated", out, cyc);
    return 0;
}
