// linear16_stream.hip -- diagnostic / groundwork: one wave computes Out[16 rows][128] = In[16 rows][K] W^T + on
// v_mfma_f32_16x16x4_f32 with the WEIGHTS as the A operand, streamed from global memory through a register ring
// (the row-group kernels of pf_rg.hip use v_mfma_f32_4x4x1_16b_f32 with 4 / 8 rows per wave and are bound by the issue
// rate: ~115 cycles per 1-KiB quad of weights where the matrix pipe needs 64).  Here a 1-KiB quad feeds 4 MFMAs of 32
// cycles each -- 4x fewer issues per FLOP, 4x the rows per weight byte.
//   layout (gfx950 16x16x4 f32): A[m][k]: lane l holds m = l & 15, k = l >> 4;  B[k][n]: lane l holds k = l >> 4,
//   n = l & 15;  D[m][n]: lane l, register i holds m = 4 (l >> 4) + i, n = l & 15.   m = output feature, n = row.
//   stream: quad q = [64 lanes][4 floats]; image j of quad (ks, half) = A operand of output tile 4 half + j, k-step ks.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/probes/linear16_stream.hip -o linear16_stream && ./linear16_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <type_traits>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
constexpr int KS = 36;                 // k-steps of 4: K = 144 (128 scalars + 16 rbf / sh)
constexpr int NQ = KS * 2;             // quads per Linear: 2 per k-step (8 output tiles of 16)
template <int D, bool STREAM>
__global__ __launch_bounds__(64) void k_lin(const float* __restrict__ wq, const float* __restrict__ in, float* out,
                                            unsigned long long* cyc, int reps) {
    const int l = threadIdx.x;
    const int wave = blockIdx.x;
    float b[KS];                                            // In^T: k = 4 ks + (l >> 4), row = l & 15
    for (int ks = 0; ks < KS; ++ks) b[ks] = in[((size_t)wave * 16 + (l & 15)) * (KS * 4) + 4 * ks + (l >> 4)];
    const f32x4* p = reinterpret_cast<const f32x4*>(wq) + l;
    f32x4 ring[D];
    static_for<0, D>([&](auto I) { ring[decltype(I)::value] = p[decltype(I)::value * 64]; });
    f32x4 acc[8];
    for (int t = 0; t < 8; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        static_for<0, NQ>([&](auto QI) {
            constexpr int qi = decltype(QI)::value;
            constexpr int ks = qi / 2, half = qi % 2;
            const f32x4 w = ring[qi % D];
            if constexpr (STREAM) ring[qi % D] = p[((qi + D) % NQ) * 64];      // wraps: the same Linear again
            static_for<0, 4>([&](auto J) {
                constexpr int j = decltype(J)::value;
                acc[4 * half + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j], b[ks], acc[4 * half + j], 0, 0, 0);
            });
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < 8; ++t)
        for (int i = 0; i < 4; ++i)
            out[((size_t)wave * 16 + (l & 15)) * 128 + 16 * t + 4 * (l >> 4) + i] = acc[t][i];
    if (l == 0) cyc[wave] = t1 - t0;
}
int main() {
    const int K = KS * 4, W = 1024;                                            // up to W waves
    std::vector<float> hw((size_t)128 * K), hin((size_t)W * 16 * K), hq((size_t)NQ * 256);
    srand(1);
    for (auto& x : hw) x = (rand() % 2001 - 1000) * 1e-3f;
    for (auto& x : hin) x = (rand() % 2001 - 1000) * 1e-3f;
    for (int ks = 0; ks < KS; ++ks)
        for (int half = 0; half < 2; ++half)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 4; ++j)
                    hq[(((size_t)(ks * 2 + half)) * 64 + l) * 4 + j] = hw[(size_t)(16 * (4 * half + j) + (l & 15)) * K + 4 * ks + (l >> 4)];
    float *dq, *din, *dout; unsigned long long* dc;
    hipMalloc(&dq, hq.size() * 4); hipMalloc(&din, hin.size() * 4); hipMalloc(&dout, (size_t)W * 16 * 128 * 4); hipMalloc(&dc, W * 8);
    hipMemcpy(dq, hq.data(), hq.size() * 4, hipMemcpyHostToDevice); hipMemcpy(din, hin.data(), hin.size() * 4, hipMemcpyHostToDevice);
    // correctness of the layout (one pass)
    hipLaunchKernelGGL((k_lin<12, true>), 2, 64, 0, 0, dq, din, dout, dc, 1); hipDeviceSynchronize();
    std::vector<float> ho((size_t)2 * 16 * 128);
    hipMemcpy(ho.data(), dout, ho.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int r = 0; r < 32; ++r)
        for (int f = 0; f < 128; ++f) {
            double s = 0;
            for (int k = 0; k < K; ++k) s += (double)hin[(size_t)r * K + k] * hw[(size_t)f * K + k];
            worst = fmax(worst, fabs(s - ho[(size_t)r * 128 + f]));
        }
    printf("layout check: max |error| = %.2e\n", worst);
    const int reps = 20;
    std::vector<unsigned long long> hc(W);
    auto report = [&](const char* name, int waves) {
        hipDeviceSynchronize();
        hipMemcpy(hc.data(), dc, waves * 8, hipMemcpyDeviceToHost);
        double m = 0; for (int i = 0; i < waves; ++i) m = fmax(m, (double)hc[i]);
        printf("%-44s waves %4d : %7.0f cycles per Linear(144->128) on 16 rows = %.1f cyc/mfma, %.0f cyc/row\n", name, waves,
               m / reps, m / reps / (NQ * 4), m / reps / 16);
    };
    for (int waves : {1, 256, 1024}) {
        hipLaunchKernelGGL((k_lin<12, false>), waves, 64, 0, 0, dq, din, dout, dc, reps); report("weights in registers (no stream)", waves);
        hipLaunchKernelGGL((k_lin<6, true>), waves, 64, 0, 0, dq, din, dout, dc, reps); report("streamed, ring of 6 quads", waves);
        hipLaunchKernelGGL((k_lin<12, true>), waves, 64, 0, 0, dq, din, dout, dc, reps); report("streamed, ring of 12 quads", waves);
        hipLaunchKernelGGL((k_lin<24, true>), waves, 64, 0, 0, dq, din, dout, dc, reps); report("streamed, ring of 24 quads", waves);
    }
    return 0;
}
