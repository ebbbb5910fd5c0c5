// Microbenchmark: what the WHOLE CHIP sustains in back-to-back MFMA issue -- v_mfma_f32_16x16x32_bf16 against
// v_mfma_i32_16x16x64_i8 (nominally twice the bf16 rate) -- from registers only (no LDS, no memory), 256 CUs x 8 waves, for
// ~tens of milliseconds, with operand registers that CHANGE from one MFMA to the next (random bit patterns) or with all-zero
// operands.  The question behind it: the split-bf16 GEMM is power / clock limited at full width; an int8
// slicing scheme is priced at "2 x the bf16 rate" -- does a power-limited chip deliver that factor?
// Build: hipcc -O3 --offload-arch=gfx950 mfma_rate.hip -o mfma_rate.bin ; run: ./mfma_rate.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

constexpr int NACC = 8;     // independent accumulators per wave (no dependent back-to-back MFMAs)
constexpr int NOP = 4;      // distinct A and B fragments rotated through (operands toggle between consecutive MFMAs)

// MODE 0: bf16 16x16x32 (16384 FLOP per wave-instruction), MODE 1: i8 16x16x64 (32768 OP per wave-instruction)
template <int MODE>
__global__ __launch_bounds__(512) void k(int iters, int zero, unsigned long long* cyc, float* sink) {
    const uint32_t seed = hash32(blockIdx.x * 512u + threadIdx.x + 1u);
    i32x4 a[NOP], b[NOP];
#pragma unroll
    for (int i = 0; i < NOP; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            uint32_t va = zero ? 0u : hash32(seed + 17u * i + j), vb = zero ? 0u : hash32(seed * 3u + 29u * i + j);
            if (MODE == 0) {          // keep the bf16 exponents moderate: random sign + mantissa, exponent 0x3f (values in [1, 2) * +-1 / 2^k)
                va = (va & 0x807f807fu) | 0x3f003f00u;
                vb = (vb & 0x807f807fu) | 0x3c003c00u;
            }
            a[i][j] = (int)va;
            b[i][j] = (int)vb;
        }
    f32x4 accf[NACC];
    i32x4 acci[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) {
        accf[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        acci[i] = i32x4{0, 0, 0, 0};
    }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int i = 0; i < NACC; i++) {
                const int ia = (i + u) % NOP, ib = (i * 3 + u) % NOP;
                if (MODE == 0)
                    accf[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[ia]), __builtin_bit_cast(bf16x8, b[ib]), accf[i], 0, 0, 0);
                else
                    acci[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[ia], b[ib], acci[i], 0, 0, 0);
            }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; i++) s += accf[i][0] + accf[i][3] + (float)acci[i][1];
    if (s == 1.2345e-33f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
static int run(const char* name, int zero, int grid, int iters, unsigned long long* d_cyc, float* d_sink) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), 0, 0, iters / 8, zero, d_cyc, d_sink);      // warm-up / clock ramp
    CHK(hipDeviceSynchronize());
    float best = 1e30f;
    double clk = 0;
    for (int rep = 0; rep < 3; rep++) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), 0, 0, iters, zero, d_cyc, d_sink);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long h[8];
        CHK(hipMemcpy(h, d_cyc, sizeof(h), hipMemcpyDeviceToHost));
        if (ms < best) {
            best = ms;
            clk = (double)h[0] / (ms * 1e-3) / 1e9;      // counter ticks per second: the shader clock on gfx9-family parts (s_memtime)
        }
    }
    const double ops = (double)grid * 8 * (double)iters * 4 * NACC * (MODE == 0 ? 16384.0 : 32768.0);
    // cycles per MFMA and SIMD = (launch time x clock) / (MFMAs per SIMD): two waves per SIMD share the pipe
    const double per_simd = (double)iters * 4 * NACC * 2;
    printf("%-34s %s operands: %8.2f ms  %8.1f T%s/s   counter %.3f GHz   %.2f counter ticks per MFMA and SIMD\n", name,
           zero ? "all-zero" : "random  ", best, ops / (best * 1e-3) / 1e12, MODE == 0 ? "FLOP" : "OP", clk, best * 1e-3 * clk * 1e9 / per_simd);
    return 0;
}

int main(int argc, char** argv) {
    hipDeviceProp_t p;
    CHK(hipGetDeviceProperties(&p, 0));
    const int grid = p.multiProcessorCount;
    const int iters = argc > 1 ? atoi(argv[1]) : 60000;
    printf("%s: %d CUs, one 8-wave workgroup per CU, %d x 32 MFMAs per wave\n", p.name, grid, iters);
    unsigned long long* d_cyc;
    float* d_sink;
    CHK(hipMalloc(&d_cyc, (size_t)grid * 8 * sizeof(unsigned long long)));
    CHK(hipMalloc(&d_sink, 64));
    for (int round = 0; round < 2; round++) {
        if (run<0>("v_mfma_f32_16x16x32_bf16", 0, grid, iters, d_cyc, d_sink)) return 1;
        if (run<1>("v_mfma_i32_16x16x64_i8", 0, grid, iters, d_cyc, d_sink)) return 1;
        if (run<0>("v_mfma_f32_16x16x32_bf16", 1, grid, iters, d_cyc, d_sink)) return 1;
        if (run<1>("v_mfma_i32_16x16x64_i8", 1, grid, iters, d_cyc, d_sink)) return 1;
    }
    return 0;
}
