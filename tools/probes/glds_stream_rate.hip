// glds_stream_rate.hip -- diagnostic: how fast ONE wave (and 2 / 4 waves of a workgroup, each with its own stream) takes a table of
// 1-KiB weight quads in when the loads go global -> LDS directly (global_load_lds_dwordx4, no vector-register destination) and the
// wave reads each quad back with ds_read_b128 -- against the 16-18 B/clk per wave of global_load_dwordx4 into registers
// (stream_rate.hip).  R quads in flight per wave (a ring of R KiB of LDS), counted s_waitcnt vmcnt, the LDS reads in inline asm
// so that the compiler does not drain the DMA queue in front of them.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/probes/glds_stream_rate.hip -o glds_rate && ./glds_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int R>
__global__ void k_glds(const f32x4* __restrict__ tab, int nq, float* out, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, W = blockDim.x >> 6;
    char* ring = sm + (size_t)w * R * 1024;                    // this wave's ring: R slots of 1 KiB
    const f32x4* src = tab + lane;                             // every wave streams the whole table (its own stream)
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int r = 0; r < R; ++r)
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + (size_t)r * 64),
                                         (void __attribute__((address_space(3)))*)(ring + r * 1024), 16, 0, 0);
    const unsigned lbase = (unsigned)(size_t)(ring) + 16u * (unsigned)lane;
    for (int q0 = 0; q0 < nq; q0 += R) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int q = q0 + r;
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(R - 1) : "memory");         // quad q has landed (loads complete in order)
            f32x4 x;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(x) : "v"(lbase + (unsigned)r * 1024u) : "memory");
            acc += x;
            const int qn = q + R < nq ? q + R : q;                                 // (tail: re-fetch, keeps the count uniform)
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + (size_t)qn * 64),
                                             (void __attribute__((address_space(3)))*)(ring + r * 1024), 16, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
    if (lane == 0) cyc[(size_t)blockIdx.x * W + w] = t1 - t0;
}
// the register path of stream_rate.hip for comparison, same table, D quads in flight
template <int D>
__global__ void k_reg(const f32x4* __restrict__ tab, int nq, float* out, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, W = blockDim.x >> 6;
    const f32x4* p = tab + lane;
    f32x4 ring[D], acc = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int r = 0; r < D; ++r) ring[r] = p[(size_t)r * 64];
    for (int q0 = 0; q0 < nq; q0 += D) {
#pragma unroll
        for (int r = 0; r < D; ++r) { acc += ring[r]; const int qn = q0 + r + D < nq ? q0 + r + D : q0 + r; ring[r] = p[(size_t)qn * 64]; }
    }
#pragma unroll
    for (int r = 0; r < D; ++r) acc += ring[r];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
    if (lane == 0) cyc[(size_t)blockIdx.x * W + w] = t1 - t0;
}
static f32x4* dtab; static float* dout; static unsigned long long* dcyc;
template <class K> static void run(K kern, const char* tag, int G, int W, int nq, size_t lds, int depth) {
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(kern, dim3(G), dim3(64 * W), lds, 0, dtab, nq, dout, dcyc); (void)hipDeviceSynchronize(); }
    std::vector<unsigned long long> c((size_t)G * W);
    (void)hipMemcpy(c.data(), dcyc, c.size() * 8, hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end());
    const double med = (double)c[c.size() / 2];
    printf("%-30s G=%4d waves/WG=%d in flight=%2d nq=%4d: %8.0f ticks median  %6.1f B/tick per wave  %6.1f per workgroup\n", tag, G, W, depth, nq, med,
           nq * 1024.0 / med, W * nq * 1024.0 / med);
}
int main() {
    const int nq = 576;                         // a six-block chain
    (void)hipMalloc(&dtab, (size_t)nq * 1024 + 65536); (void)hipMemset(dtab, 0, (size_t)nq * 1024 + 65536);
    (void)hipMalloc(&dout, (size_t)2048 * 1024 * 4); (void)hipMalloc(&dcyc, (size_t)2048 * 16 * 8);
    (void)hipFuncSetAttribute((const void*)k_glds<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)k_glds<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)k_glds<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int G : {48, 256}) {
        for (int W : {1, 2, 4}) {
            run(k_reg<12>, "registers (global_load)", G, W, nq, 0, 12);
            run(k_reg<24>, "registers (global_load)", G, W, nq, 0, 24);
            run(k_glds<8>, "LDS-DMA + ds_read_b128", G, W, nq, (size_t)W * 8 * 1024, 8);
            run(k_glds<16>, "LDS-DMA + ds_read_b128", G, W, nq, (size_t)W * 16 * 1024, 16);
            if (W <= 4) run(k_glds<32>, "LDS-DMA + ds_read_b128", G, W, nq, (size_t)W * 32 * 1024, 32);
        }
    }
    return 0;
}
