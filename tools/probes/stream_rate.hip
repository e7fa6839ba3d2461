// stream_rate.hip -- diagnostic: how many bytes per clock a CU takes in when its waves stream an L2-resident table that
// does not fit the 32 KiB vector L1 with global_load_dwordx4 (the access pattern of the weight "quads" of pf_rg.hip:
// one load = 64 lanes x 16 B = 1 KiB contiguous), as a function of waves per CU, loads in flight per wave, how the
// waves of a workgroup divide the table, the number of workgroups reading the SAME table, and the cache policy.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/probes/stream_rate.hip -o stream_rate && ./stream_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
// POLICY 0: plain, 1: nontemporal
template <int POLICY> __device__ __forceinline__ f32x4 ld(const f32x4* p) {
    if constexpr (POLICY == 1) return __builtin_nontemporal_load(p);
    else return *p;
}
// SPLIT 0: every wave of the workgroup reads the whole table in the same order; 1: wave w reads quads w, w + W, ... (a K split);
// 2: every wave reads the whole table, wave w starting at quad w * nq / W
template <int D, int SPLIT, int POLICY>
__global__ void k_stream(const f32x4* __restrict__ tab, int nq, int reps, float* out, unsigned long long* cyc) {
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6, W = blockDim.x >> 6;
    const f32x4* p = tab + l;
    const int mine = SPLIT == 1 ? nq / W : nq;               // quads this wave reads per repetition
    const int step = SPLIT == 1 ? W : 1;
    const int first = SPLIT == 1 ? w : (SPLIT == 2 ? (w * (nq / W)) : 0);
    f32x4 ring[D];
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    int q = first;                                           // table index of the next quad to request
    auto next = [&]() { const f32x4* a = p + (size_t)q * 64; q += step; if (q >= nq) q -= nq; return a; };
    static_for<0, D>([&](auto U) { ring[decltype(U)::value] = ld<POLICY>(next()); });
    const int total = mine * reps;
    for (int i = 0; i + D <= total; i += D) {
        static_for<0, D>([&](auto U) {
            constexpr int u = decltype(U)::value;
            acc += ring[u];
            ring[u] = ld<POLICY>(next());
        });
    }
    static_for<0, D>([&](auto U) { acc += ring[decltype(U)::value]; });
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
    if (l == 0) cyc[(size_t)blockIdx.x * W + w] = t1 - t0;
}
// table staged once into LDS by the whole workgroup, then every wave reads all of it from LDS `reps` times (ds_read_b128)
template <int D>
__global__ void k_lds(const f32x4* __restrict__ tab, int nq, int reps, float* out, unsigned long long* cyc) {
    extern __shared__ f32x4 sm[];
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6, W = blockDim.x >> 6;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = threadIdx.x; i < nq * 64; i += blockDim.x) sm[i] = tab[i];
    __syncthreads();
    const unsigned long long tf = __builtin_amdgcn_s_memtime();
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < reps; ++r)
        for (int i = 0; i + D <= nq; i += D)
            static_for<0, D>([&](auto U) { acc += sm[(size_t)(i + decltype(U)::value) * 64 + l]; });
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
    if (l == 0) { cyc[(size_t)blockIdx.x * W + w] = t1 - tf; if (w == 0) cyc[(size_t)gridDim.x * W + blockIdx.x] = tf - t0; }
}

__global__ void k_evict(f32x4* buf, size_t n) {          // touch 64 MiB: nothing of the table is left in any XCD's L2
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) buf[i] += (f32x4){1.f, 1.f, 1.f, 1.f};
}
static f32x4* dtab; static float* dout; static unsigned long long* dcyc; static f32x4* dbig; static int g_evict = 0; static size_t g_evict_bytes = (size_t)64 << 20;
static double median(std::vector<unsigned long long>& v) { std::sort(v.begin(), v.end()); return (double)v[v.size() / 2]; }

template <int D, int SPLIT, int POLICY> static void run(int G, int W, int nq, int reps, const char* tag) {
    hipLaunchKernelGGL((k_stream<D, SPLIT, POLICY>), dim3(G), dim3(64 * W), 0, 0, dtab, nq, reps, dout, dcyc);   // warm: L2, i-cache
    hipDeviceSynchronize();
    if (g_evict) { hipLaunchKernelGGL(k_evict, dim3(2048), dim3(256), 0, 0, dbig, g_evict_bytes / 16); if (g_evict == 2) hipDeviceSynchronize(); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_stream<D, SPLIT, POLICY>), dim3(G), dim3(64 * W), 0, 0, dtab, nq, reps, dout, dcyc);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c((size_t)G * W);
    hipMemcpy(c.data(), dcyc, c.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long mx = *std::max_element(c.begin(), c.end());
    const double med = median(c);
    const double bytes_wg = (SPLIT == 1 ? 1.0 : (double)W) * nq * 1024.0 * reps;       // bytes requested by one workgroup
    printf("%-34s G=%4d W=%2d D=%2d nq=%4d: %7.0f ticks median (%7llu max)  %6.1f B/tick per workgroup  %6.2f TB/s whole launch (%.1f us)\n",
           tag, G, W, D, nq, med, mx, bytes_wg / med, bytes_wg * G / (ms * 1e-3) / 1e12, ms * 1e3);
    hipEventDestroy(e0); hipEventDestroy(e1);
}
template <int D> static void run_lds(int G, int W, int nq, int reps) {
    const size_t sh = (size_t)nq * 1024;
    hipFuncSetAttribute((const void*)k_lds<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    hipLaunchKernelGGL((k_lds<D>), dim3(G), dim3(64 * W), sh, 0, dtab, nq, reps, dout, dcyc);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((k_lds<D>), dim3(G), dim3(64 * W), sh, 0, dtab, nq, reps, dout, dcyc);
    hipDeviceSynchronize();
    std::vector<unsigned long long> c((size_t)G * W + G);
    hipMemcpy(c.data(), dcyc, c.size() * 8, hipMemcpyDeviceToHost);
    std::vector<unsigned long long> rd(c.begin(), c.begin() + (size_t)G * W), fill(c.begin() + (size_t)G * W, c.end());
    const double mr = median(rd), mf = median(fill);
    printf("LDS: staged once, read %d times       G=%4d W=%2d D=%2d nq=%4d: fill %6.0f ticks = %5.1f B/tick; reads %7.0f ticks = %6.1f B/tick per workgroup\n",
           reps, G, W, D, nq, mf, nq * 1024.0 / mf, mr, (double)W * nq * 1024.0 * reps / mr);
}
int main(int argc, char** argv) {
    const int maxq = 4096;
    hipMalloc(&dtab, (size_t)maxq * 1024); hipMemset(dtab, 0, (size_t)maxq * 1024);
    hipMalloc(&dout, (size_t)2048 * 1024 * 4); hipMalloc(&dcyc, (size_t)2048 * 16 * 8 + 2048 * 8);
    // s_memtime tick against wall time: a spin of known length
    {
        run<12, 0, 0>(256, 1, 96, 64, "calibration");
    }
    const int nq = 96, reps = 40;          // one GVP block's quads; 288 = a three-block chain
    for (int G : {48, 256, 1024}) {
        for (int W : {1, 2, 4, 8, 16}) {
            if (G * W > 8192) continue;
            run<12, 0, 0>(G, W, nq, reps, "same order, plain");
            run<12, 2, 0>(G, W, nq, reps, "staggered starts, plain");
            if (W > 1) run<12, 1, 0>(G, W, nq, reps, "K split, plain");
        }
    }
    for (int W : {1, 4}) {
        run<4, 2, 0>(256, W, nq, reps, "staggered, D=4");
        run<24, 2, 0>(256, W, nq, reps, "staggered, D=24");
        run<12, 2, 1>(256, W, nq, reps, "staggered, nontemporal");
        run<12, 2, 0>(256, W, 288, reps, "staggered, 288-quad table");
        run<12, 2, 0>(256, W, 24, reps, "staggered, 24-quad table (fits L1)");
    }
    for (int W : {4, 8, 16}) run_lds<8>(256, W, 96, 20);
    // ONE pass over a chain's weights, as an item of pf_rg.hip reads them: after a previous launch of the same kernel (the state
    // a step's launches find: same stream, back to back), and after 64 MiB of other traffic
    hipMalloc(&dbig, (size_t)64 << 20); hipMemset(dbig, 0, (size_t)64 << 20);
    for (int ev : {0, 1}) {
        g_evict = ev;
        printf("--- single pass (reps = 1), %s\n", ev ? "after a 64 MiB sweep by another kernel" : "straight after a launch of the same kernel");
        for (int G : {48, 256, 1600}) {
            for (int nq2 : {96, 288, 576}) {
                run<12, 0, 0>(G, 1, nq2, 1, "one wave per workgroup");
                run<12, 1, 0>(G, 4, nq2, 1, "K split over 4 waves");
                run<12, 0, 0>(G, 4, nq2, 1, "4 waves, same order");
            }
        }
    }
    // how much other traffic between two uses does it take to push a 288-quad table out of the XCDs' L2s (4 MiB each; the sweep is
    // spread over the eight of them)?
    g_evict = 1;
    for (size_t mib : {1, 2, 4, 8, 16, 24, 32, 48, 64}) {
        g_evict_bytes = mib << 20;
        printf("--- %zu MiB swept by another kernel in between (%.2f MiB per XCD): ", mib, mib / 8.0);
        run<12, 0, 0>(256, 1, 288, 1, "one wave per workgroup");
    }
    return 0;
}
