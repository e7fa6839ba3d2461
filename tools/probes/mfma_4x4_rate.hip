// mfma_4x4_rate.hip -- diagnostic: issue rate of v_mfma_f32_4x4x1_16b_f32 as a function of where the operands come
// from (same / different A and B registers, CBSZ modes, 1 / 2 / 4 accumulators), v_mfma_f32_16x16x4_f32 for reference.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/probes/mfma_4x4_rate.hip -o mfma_4x4_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
// MODE 0: same A, same B; 1: same A, 16 different B registers; 2: 16 different A and B; 3: same A, B alternates between 2 regs
template <int MODE, int NACC, int CBSZ>
__global__ void k_rate(float* out, unsigned long long* cyc, int iters, const float* in) {
    const int l = threadIdx.x;
    f32x4 acc[NACC];
    for (int q = 0; q < NACC; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a[16], b[16];
    for (int k = 0; k < 16; ++k) { a[k] = in[l + k * 64]; b[k] = in[1024 + l + k * 64]; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        static_for<0, 16>([&](auto U) {
            constexpr int u = decltype(U)::value;
            const float av = MODE == 2 ? a[u] : a[0];
            const float bv = MODE == 0 ? b[0] : (MODE == 3 ? b[u & 1] : b[u]);
            acc[u % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, acc[u % NACC], CBSZ, CBSZ == 4 ? u : (CBSZ == 2 ? u & 3 : 0), 0);
        });
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int q = 0; q < NACC; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    out[blockIdx.x * 64 + l] = s;
    if (l == 0) cyc[blockIdx.x] = t1 - t0;
}
// 16x16x4 f32 (32 cycles) and 32x32x2 (64 cycles) for reference with varying B
template <int NACC>
__global__ void k_rate16(float* out, unsigned long long* cyc, int iters, const float* in) {
    const int l = threadIdx.x;
    f32x4 acc[NACC];
    for (int q = 0; q < NACC; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a[16], b[16];
    for (int k = 0; k < 16; ++k) { a[k] = in[l + k * 64]; b[k] = in[1024 + l + k * 64]; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        static_for<0, 16>([&](auto U) {
            constexpr int u = decltype(U)::value;
            acc[u % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc[u % NACC], 0, 0, 0);
        });
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int q = 0; q < NACC; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    out[blockIdx.x * 64 + l] = s;
    if (l == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    float* din; hipMalloc(&din, 4096 * 4); hipMemset(din, 0, 4096 * 4);
    float* dout; hipMalloc(&dout, 64 * 64 * 4);
    unsigned long long* dc; hipMalloc(&dc, 64 * 8); unsigned long long hc[4];
    const int iters = 200;
#define RUN(MODE, NACC, CBSZ, NAME) hipLaunchKernelGGL((k_rate<MODE, NACC, CBSZ>), 1, 64, 0, 0, dout, dc, iters, din); hipDeviceSynchronize(); \
    hipMemcpy(hc, dc, 8, hipMemcpyDeviceToHost); printf("4x4x1 %-34s acc=%d cbsz=%d : %.2f cyc/mfma\n", NAME, NACC, CBSZ, (double)hc[0] / (iters * 16));
    RUN(0, 2, 4, "same A, same B")
    RUN(1, 2, 4, "same A, 16 different B")
    RUN(3, 2, 4, "same A, B alternates 2 regs")
    RUN(2, 2, 4, "16 different A and B")
    RUN(1, 4, 4, "same A, 16 different B")
    RUN(1, 2, 0, "same A, 16 different B")
    RUN(1, 2, 2, "same A, 16 different B")
    RUN(1, 1, 4, "same A, 16 different B")
    hipLaunchKernelGGL((k_rate16<2>), 1, 64, 0, 0, dout, dc, iters, din); hipDeviceSynchronize();
    hipMemcpy(hc, dc, 8, hipMemcpyDeviceToHost); printf("16x16x4 different A,B acc=2 : %.2f cyc/mfma\n", (double)hc[0] / (iters * 16));
    hipLaunchKernelGGL((k_rate16<4>), 1, 64, 0, 0, dout, dc, iters, din); hipDeviceSynchronize();
    hipMemcpy(hc, dc, 8, hipMemcpyDeviceToHost); printf("16x16x4 different A,B acc=4 : %.2f cyc/mfma\n", (double)hc[0] / (iters * 16));
    return 0;
}
