// Microbenchmark: per-CU throughput of global_load_lds_dwordx4 (LDS-DMA, 1 KiB per wave-instruction) against
// global_load_dwordx4 -> VGPR (+ ds_write_b128), from an L2-resident buffer, for 1..8 waves per workgroup and one workgroup per CU.
// Build: hipcc -O3 --offload-arch=gfx950 lds_dma_rate.hip -o lds_dma_rate ; run: ./lds_dma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// each wave: REPS rounds of PIECES pieces; piece = 8 rows x 128 B at `stride` bytes (mode 0) or 1 KiB contiguous (stride = 128)
template <int MODE>   // 0 = LDS-DMA, 1 = global_load -> VGPR -> ds_write, 2 = global_load -> VGPR only
__global__ __launch_bounds__(512) void k(const char* __restrict__ src, size_t wg_bytes, int stride, int pieces, int reps,
                                         unsigned long long* out, float* sink, int cold, int lpr) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const char* base = cold ? src + (size_t)blockIdx.x * (4u << 20) : src + (size_t)(blockIdx.x % 64) * wg_bytes;   // hot: 64 regions, cache resident
    const uint32_t loff = (uint32_t)(lane / lpr) * stride + (lane % lpr) * 16;      // lpr lanes (16 B each) per row: 8 = 128-byte rows, 4 = 64-byte rows
    float acc = 0.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    unsigned long long t_issue = 0;
    for (int r = 0; r < reps; r++) {
        const unsigned long long ti0 = __builtin_readcyclecounter();
        for (int p = 0; p < pieces; p++) {
            const int pi = wave * pieces + p;
            const int rpp = 64 / lpr;                                           // rows per piece
            const uint32_t lin = (uint32_t)((r * nw + wave) * pieces + p);      // cold: every piece of the launch at its own address
            const char* sb = cold ? base + ((lin * rpp * (uint32_t)stride + (lin * rpp * (uint32_t)stride >> 22) * 128u) & ((4u << 20) - 1 - 65536))
                                  : base + (((uint32_t)pi * rpp * (uint32_t)stride + (lpr == 4 ? (uint32_t)(r & 1) * 64u : 0u)) & (uint32_t)(wg_bytes / 2 - 1));
            char* dst = lds + (size_t)pi * 1024;
            if constexpr (MODE == 0) {
                const uint32_t l = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)dst;
                const uint64_t sbu = reinterpret_cast<uint64_t>(sb);
                const uint64_t sbs = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(sbu >> 32)) << 32) |
                                     (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)sbu);
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(loff), "s"(sbs),
                             "s"(__builtin_amdgcn_readfirstlane((int)l)) : "memory");
            } else {
                typedef __attribute__((ext_vector_type(4))) float f4;
                f4 v;
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(loff), "s"(sb) : "memory");
                if constexpr (MODE == 1) {
                    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");   // keep up to 4 loads in flight
                    *reinterpret_cast<f4*>(dst + lane * 16) = v;
                } else {
                    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                    acc += v[0];
                }
            }
        }
        t_issue += __builtin_readcyclecounter() - ti0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    if (lane == 0) out[blockIdx.x * nw + wave] = t1 - t0;
    if (lane == 0) out[2048 + blockIdx.x * nw + wave] = t_issue;
    if (acc == 123.456f) sink[0] = acc + lds[lane];
}

int main() {
    const size_t wg_bytes = 1 << 20;
    char* src;
    unsigned long long* out;
    float* sink;
    CHK(hipMalloc(&src, (size_t)256 * (4u << 20)));
    CHK(hipMemset(src, 1, (size_t)256 * (4u << 20)));
    CHK(hipMalloc(&out, 2 * 2048 * 8));
    CHK(hipMalloc(&sink, 64));
    CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    const int reps = 32;
    for (int cold = 0; cold < 2; cold++)
    for (int lpr : {8, 4})
    for (int stride : {lpr * 16, 2048, 6144}) {
        for (int nw : {4, 8}) {
            const int pieces = 64 / nw > 16 ? 16 : 64 / nw;       // <= 64 KiB of LDS targets per workgroup
            for (int mode = 0; mode < 2; mode++) {
                std::vector<unsigned long long> h(256 * nw), hi(256 * nw);
                for (int it = 0; it < 2; it++) {
                    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(nw * 64), 131072, 0, src, wg_bytes, stride, pieces, reps, out, sink, cold, lpr);
                    if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(nw * 64), 131072, 0, src, wg_bytes, stride, pieces, reps, out, sink, cold, lpr);
                    if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(nw * 64), 131072, 0, src, wg_bytes, stride, pieces, reps, out, sink, cold, lpr);
                    CHK(hipDeviceSynchronize());
                }
                CHK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
                CHK(hipMemcpy(hi.data(), out + 2048, hi.size() * 8, hipMemcpyDeviceToHost));
                double mx = 0, mi = 0;
                for (auto v : h) mx += (double)v;
                for (auto v : hi) mi += (double)v;
                mx /= h.size();
                mi /= hi.size();
                const double bytes = (double)nw * pieces * reps * 1024.0;
                printf("%s rows of %3d B, stride %5d B  waves %d  pieces/wave/round %2d  %s: %7.1f cycles per piece per wave (issue alone %6.1f), %6.1f B/clk per CU\n", cold ? "cold" : "hot ", lpr * 16, stride, nw, pieces,
                       mode == 0 ? "LDS-DMA        " : mode == 1 ? "VGPR + ds_write" : "VGPR only      ", mx / (pieces * reps), mi / (pieces * reps), bytes / mx);
            }
        }
    }
    return 0;
}
