// mfma_4x4_semantics.hip -- diagnostic: operand / result lane mapping of v_mfma_f32_4x4x1_16b_f32 under the CBSZ / ABID
// broadcast controls (checked against D[i][lane] = A[lane 4 * selected block + i] * B[lane]), its issue rate, and a first
// weight-streaming loop.  The mapping is what pf_rg.hip is built on.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_4x4_semantics.hip -o mfma_4x4_semantics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#include <type_traits>
template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int CBSZ, int ABID>
__global__ void k_sem(const float* a, const float* b, float* d) {
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, CBSZ, ABID, 0);
    for (int i = 0; i < 4; ++i) d[i * 64 + l] = c[i];
}

// timing: NM dependent-or-not MFMAs
template <int NACC>
__global__ void k_rate(float* out, unsigned long long* cyc, int iters) {
    const int l = threadIdx.x;
    f32x4 acc[NACC];
    for (int q = 0; q < NACC; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = l * 0.001f, b = 1.0f + l * 0.002f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        static_for<0, 16>([&](auto U) {
#pragma unroll
            for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[q], 4, decltype(U)::value, 0);
        });
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int q = 0; q < NACC; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    out[blockIdx.x * 64 + l] = s;
    if (l == 0) cyc[blockIdx.x] = t1 - t0;
}

// streaming: one wave = RG row groups; per "GVP" NK k-steps x 2 output halves; weights packed [k/4][half][lane][4]
template <int RG>
__global__ __launch_bounds__(64) void k_stream(const float* __restrict__ w, float* out, int nk4, int ngvp, size_t gvp_stride) {
    const int l = threadIdx.x;
    f32x4 lo[RG], hi[RG];
    float x[RG][9];
    for (int r = 0; r < RG; ++r) {
        lo[r] = (f32x4){0.f, 0.f, 0.f, 0.f}; hi[r] = lo[r];
        for (int m = 0; m < 9; ++m) x[r][m] = 0.001f * (l + m + r);
    }
    for (int g = 0; g < ngvp; ++g) {
        const f32x4* wp = reinterpret_cast<const f32x4*>(w + g * gvp_stride) + l;
        for (int k16 = 0; k16 < nk4 / 4; ++k16) {
            static_for<0, 4>([&](auto Q) {
                constexpr int q = decltype(Q)::value;
                const int k4 = k16 * 4 + q;
                const f32x4 wl = wp[(k4 * 2 + 0) * 64], wh = wp[(k4 * 2 + 1) * 64];
                static_for<0, 4>([&](auto I) {
                    constexpr int i = decltype(I)::value;
#pragma unroll
                    for (int r = 0; r < RG; ++r) {
                        lo[r] = __builtin_amdgcn_mfma_f32_4x4x1f32(x[r][k16], wl[i], lo[r], 4, q * 4 + i, 0);
                        hi[r] = __builtin_amdgcn_mfma_f32_4x4x1f32(x[r][k16], wh[i], hi[r], 4, q * 4 + i, 0);
                    }
                });
            });
        }
        for (int r = 0; r < RG; ++r)
            for (int m = 0; m < 4; ++m) { x[r][m] = lo[r][m] * 1e-3f; x[r][4 + m] = hi[r][m] * 1e-3f; }
    }
    float s = 0.f;
    for (int r = 0; r < RG; ++r) s += lo[r][0] + hi[r][1] + lo[r][2] + hi[r][3];
    out[(size_t)blockIdx.x * 64 + l] = s;
}

int main() {
    // ---- semantics
    std::vector<float> ha(64), hb(64), hd(256);
    for (int l = 0; l < 64; ++l) { ha[l] = 1.0f + l; hb[l] = 100.0f + 3 * l; }
    float *da, *db, *dd;
    CK(hipMalloc(&da, 256)); CK(hipMalloc(&db, 256)); CK(hipMalloc(&dd, 1024));
    CK(hipMemcpy(da, ha.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), 256, hipMemcpyHostToDevice));
    auto check = [&](const char* name, int cbsz, int abid) {
        hipMemcpy(hd.data(), dd, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 4; ++i)
            for (int l = 0; l < 64; ++l) {
                const int b = l >> 2;
                const int group = 1 << cbsz;
                const int ablk = cbsz ? (b / group) * group + abid : b;
                const float exp = ha[4 * ablk + i] * hb[l];
                if (std::fabs(exp - hd[i * 64 + l]) > 1e-3f * std::fabs(exp)) {
                    if (bad < 4) printf("  %s mismatch i=%d lane=%d got %g exp %g\n", name, i, l, hd[i * 64 + l], exp);
                    ++bad;
                }
            }
        printf("semantics %s: %s\n", name, bad ? "MISMATCH" : "ok (D[i][lane] = A[lane 4*blk_sel+i] * B[lane])");
    };
    hipLaunchKernelGGL((k_sem<0, 0>), 1, 64, 0, 0, da, db, dd); CK(hipDeviceSynchronize()); check("cbsz0", 0, 0);
    hipLaunchKernelGGL((k_sem<4, 0>), 1, 64, 0, 0, da, db, dd); CK(hipDeviceSynchronize()); check("cbsz4 abid0", 4, 0);
    hipLaunchKernelGGL((k_sem<4, 5>), 1, 64, 0, 0, da, db, dd); CK(hipDeviceSynchronize()); check("cbsz4 abid5", 4, 5);
    hipLaunchKernelGGL((k_sem<4, 15>), 1, 64, 0, 0, da, db, dd); CK(hipDeviceSynchronize()); check("cbsz4 abid15", 4, 15);
    hipLaunchKernelGGL((k_sem<2, 3>), 1, 64, 0, 0, da, db, dd); CK(hipDeviceSynchronize()); check("cbsz2 abid3", 2, 3);
    // ---- issue rate
    float* dout; unsigned long long* dc;
    CK(hipMalloc(&dout, 4096 * 64 * 4)); CK(hipMalloc(&dc, 4096 * 8));
    unsigned long long hc[4];
    const int iters = 200;
    hipLaunchKernelGGL((k_rate<1>), 1, 64, 0, 0, dout, dc, iters); CK(hipDeviceSynchronize());
    hipMemcpy(hc, dc, 8, hipMemcpyDeviceToHost); printf("rate 1 acc : %.2f cyc/mfma\n", (double)hc[0] / (iters * 16));
    hipLaunchKernelGGL((k_rate<2>), 1, 64, 0, 0, dout, dc, iters); CK(hipDeviceSynchronize());
    hipMemcpy(hc, dc, 8, hipMemcpyDeviceToHost); printf("rate 2 acc : %.2f cyc/mfma\n", (double)hc[0] / (iters * 32));
    hipLaunchKernelGGL((k_rate<4>), 1, 64, 0, 0, dout, dc, iters); CK(hipDeviceSynchronize());
    hipMemcpy(hc, dc, 8, hipMemcpyDeviceToHost); printf("rate 4 acc : %.2f cyc/mfma\n", (double)hc[0] / (iters * 64));
    // ---- streaming
    const int nk4 = 36;                       // 144 k-steps
    const size_t gvp_stride = (size_t)nk4 * 2 * 64 * 4;   // floats
    const int ngvp = 3;
    float* dw; CK(hipMalloc(&dw, gvp_stride * ngvp * 4)); CK(hipMemset(dw, 0, gvp_stride * ngvp * 4));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grids[] = {48, 256, 512, 1024, 1700, 3400};
    for (int rg = 1; rg <= 4; rg *= 2)
        for (int gi = 0; gi < 6; ++gi) {
            const int G = grids[gi];
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0, 0);
                if (rg == 1) hipLaunchKernelGGL((k_stream<1>), G, 64, 0, 0, dw, dout, nk4, ngvp, gvp_stride);
                if (rg == 2) hipLaunchKernelGGL((k_stream<2>), G, 64, 0, 0, dw, dout, nk4, ngvp, gvp_stride);
                if (rg == 4) hipLaunchKernelGGL((k_stream<4>), G, 64, 0, 0, dw, dout, nk4, ngvp, gvp_stride);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            const double flop = (double)G * rg * 4 * ngvp * 144.0 * 128 * 2;
            printf("stream RG=%d waves=%5d rows=%6d : %7.2f us  (%.1f TFLOP/s, %.2f us per GVP level)\n", rg, G, G * rg * 4, best * 1e3,
                   flop / (best * 1e-3) / 1e12, best * 1e3 / ngvp);
        }
    CK(hipDeviceSynchronize());
    return 0;
}
