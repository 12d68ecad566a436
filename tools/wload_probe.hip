// Probe: how long does one wavefront wait for an 8 KiB weight chunk (8 x global_load_dwordx4, 1 KiB each, consecutive)
// when it walks a 42-chunk stream the way layer_wave.hip does?  Variants: no overlap (pure latency), prefetch 1 or 2
// chunks ahead of 32 fp32 MFMAs (512 matrix-pipe cycles) per chunk.  Grids: 628 waves (4 pairs of N=5000: every wave of
// the launch walks the stream in lockstep), 2048 (1 wave / SIMD), 4096 (2 / SIMD).  `cold`: a 512 MiB sweep evicts L2
// and MALL before the launch.     hipcc --offload-arch=gfx950 -O3 -o wload_probe.bin wload_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int NCH = 42;

struct Chunk { f32x4 v[8]; };
__device__ __forceinline__ void load_chunk(Chunk& c, const unsigned char* s, int i, int lane) {
    const unsigned char* p = s + (size_t)i * 8192 + lane * 16;
#pragma unroll
    for (int k = 0; k < 8; ++k) c.v[k] = *reinterpret_cast<const f32x4*>(p + 1024 * k);
}
__device__ __forceinline__ void mma(f32x16& acc, const Chunk& c, float x) {
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(c.v[k][e], x, acc, 0, 0, 0);
}

template <int DEPTH, bool MMA>
__global__ __launch_bounds__(256, 2) void probe(const unsigned char* stream, float* out, long long* cyc, int waves) {
    const int lane = threadIdx.x & 63, gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gw >= waves) return;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    Chunk c[3];
    const long long t0 = __builtin_readcyclecounter();
    if (DEPTH >= 1) load_chunk(c[0], stream, 0, lane);
    if (DEPTH >= 2) load_chunk(c[1], stream, 1, lane);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        if (DEPTH == 0) load_chunk(c[0], stream, i, lane);
        else if (i + DEPTH < NCH) load_chunk(c[(i + DEPTH) % 3], stream, i + DEPTH, lane);
        const Chunk& cur = c[DEPTH == 0 ? 0 : i % 3];
        if (MMA) mma(acc, cur, 1.0f + lane);
        else {
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += cur.v[k][0] + cur.v[k][3];
        }
        asm volatile("" : "+v"(acc));
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[(size_t)gw * 64 + lane] = s;
    if (lane == 0) cyc[gw] = t1 - t0;
}

__global__ void sweep(float* p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = p[i] * 1.0001f + 1.f;
}

int main() {
    unsigned char* stream; float* out; long long* cyc; float* big;
    const size_t big_n = (size_t)128 << 20;   // 512 MiB
    hipMalloc(&stream, NCH * 8192); hipMemset(stream, 0, NCH * 8192);
    hipMalloc(&out, 4096 * 64 * 4); hipMalloc(&cyc, 4096 * 8); hipMalloc(&big, big_n * 4); hipMemset(big, 0, big_n * 4);
    long long* h = (long long*)malloc(4096 * 8);
    for (int waves : {4, 628, 2048, 4096}) {
        for (int variant = 0; variant < 4; ++variant) {
            for (int cold = 1; cold >= 0; --cold) {
                if (cold) hipLaunchKernelGGL(sweep, dim3(2048), dim3(256), 0, 0, big, big_n);
                hipDeviceSynchronize();
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0);
                const dim3 g((waves + 3) / 4), b(256);
                if (variant == 0) hipLaunchKernelGGL((probe<0, false>), g, b, 0, 0, stream, out, cyc, waves);
                if (variant == 1) hipLaunchKernelGGL((probe<0, true>), g, b, 0, 0, stream, out, cyc, waves);
                if (variant == 2) hipLaunchKernelGGL((probe<1, true>), g, b, 0, 0, stream, out, cyc, waves);
                if (variant == 3) hipLaunchKernelGGL((probe<2, true>), g, b, 0, 0, stream, out, cyc, waves);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                hipMemcpy(h, cyc, waves * 8, hipMemcpyDeviceToHost);
                double mean = 0, mx = 0;
                for (int i = 0; i < waves; ++i) { mean += h[i]; if (h[i] > mx) mx = h[i]; }
                mean /= waves;
                const char* names[4] = {"no overlap, no mfma", "no overlap + 32 mfma", "prefetch 1 + 32 mfma", "prefetch 2 + 32 mfma"};
                printf("waves %4d  %-22s %s  kernel %.1f us  ticks/chunk mean %.0f  max %.0f\n", waves, names[variant], cold ? "cold" : "warm",
                       ms * 1e3, mean / NCH, mx / NCH);
            }
        }
    }
    return 0;
}
