// mfma_4x4_rate.hip -- diagnostic: issue rate of the f32 matrix instructions the row-group kernels are built from.
//   v_mfma_f32_4x4x1_16b_f32 (512 FLOP; the 4-row form of pf_rg.hip) against v_mfma_f32_16x16x4_f32 (2048 FLOP), with NACC independent
//   accumulators per wave and W waves per SIMD; optionally with one v_fma_f32 / one ds_read_b128 / one global_load_dwordx4
//   between MFMAs (FILL).  Prints cycles per MFMA per SIMD and the fraction of the 64 FLOP/clk/SIMD peak.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/probes/mfma_4x4_rate.hip -o mfma_rate && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND, int NACC, int FILL>
__global__ void k_rate(const float* in, float* out, unsigned long long* cyc, int reps) {
    __shared__ f32x4 sm[256];
    const int lane = threadIdx.x & 63;
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = in[lane], b = in[64 + lane], f = in[128 + lane];
    sm[threadIdx.x & 255] = (f32x4){a, b, f, a};
    const f32x4* gp = reinterpret_cast<const f32x4*>(in) + lane;
    f32x4 side = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if constexpr (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 4, 5, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
                if constexpr (FILL == 1) f = __builtin_fmaf(f, 1.0001f, 0.5f);
                if constexpr (FILL == 2) side += sm[(lane + u + i) & 255];
                if constexpr (FILL == 3) side += gp[((u * NACC + i) & 15) * 64];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = f + side[0] + side[1] + side[2] + side[3];
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[(size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

static float* din; static float* dout; static unsigned long long* dcyc;
template <int KIND, int NACC, int FILL> static void run(int wps, const char* tag) {
    const int reps = 200, G = 256, W = 4 * wps;          // one workgroup per CU, wps waves on each of its four SIMDs
    hipLaunchKernelGGL((k_rate<KIND, NACC, FILL>), dim3(G), dim3(64 * W), 0, 0, din, dout, dcyc, reps);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((k_rate<KIND, NACC, FILL>), dim3(G), dim3(64 * W), 0, 0, din, dout, dcyc, reps);
    hipDeviceSynchronize();
    std::vector<unsigned long long> c((size_t)G * W);
    hipMemcpy(c.data(), dcyc, c.size() * 8, hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end());
    const double med = (double)c[c.size() / 2];
    const double n_mfma = (double)reps * 16 * NACC * wps;                 // MFMAs issued on one SIMD in the window
    const double ticks_per = med / n_mfma;                               // s_memtime ticks (100 MHz) per MFMA -> cycles below
    const double flop = KIND == 0 ? 512.0 : 2048.0;
    printf("%-28s %-34s acc=%2d waves/SIMD=%d: %8.0f ticks  %.4f ticks/MFMA\n", KIND == 0 ? "v_mfma_f32_4x4x1_16b_f32" : "v_mfma_f32_16x16x4_f32", tag, NACC, wps, med, ticks_per);
    (void)flop;
}
int main() {
    hipMalloc(&din, 1 << 20); hipMemset(din, 0, 1 << 20);
    hipMalloc(&dout, (size_t)256 * 1024 * 4); hipMalloc(&dcyc, (size_t)256 * 16 * 8);
    for (int wps : {1, 2, 3}) {
        run<0, 1, 0>(wps, "bare"); run<0, 2, 0>(wps, "bare"); run<0, 4, 0>(wps, "bare"); run<0, 8, 0>(wps, "bare");
        run<1, 1, 0>(wps, "bare"); run<1, 2, 0>(wps, "bare"); run<1, 4, 0>(wps, "bare");
        run<0, 4, 1>(wps, "+ v_fma_f32 each"); run<1, 4, 1>(wps, "+ v_fma_f32 each");
        run<0, 4, 2>(wps, "+ ds_read_b128 each"); run<1, 4, 2>(wps, "+ ds_read_b128 each");
        run<0, 8, 3>(wps, "+ global_load_dwordx4 each"); run<1, 4, 3>(wps, "+ global_load_dwordx4 each");
    }
    // tick -> cycle calibration: the 16x16x4 form issues every 32 cycles per SIMD (MI355X_MICROARCH.md, per-instruction constants)
    return 0;
}
