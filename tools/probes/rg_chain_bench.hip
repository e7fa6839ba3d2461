// rg_chain_bench.hip -- diagnostic: the GVP chain engine of pf_rg.hip in isolation (hot, all-zero weights): 6 pipelined
// generic GVP blocks + flush per wave, at 4 and 8 rows per wave and 48 / 480 / 1700 waves.  Prints wall time per launch
// (HIP events, ~4 us of event overhead included), the mean wave lifetime in s_memtime ticks (= shader cycles) and cycles
// per MFMA.  With -DPF_STAMPS it prints the per-phase stamps of two waves instead.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Ipharmacophore-diffusion_amd/csrc tools/probes/rg_chain_bench.hip -o rg_chain_bench
#include "../../pharmacophore-diffusion_amd/csrc/pf_rg.hip"
#include <cstdio>
template <int RG>
__global__ __launch_bounds__(64) void k_chain(const float* w, float* out, int nblk, unsigned long long* cyc) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __shared__ RgLds lds[RG];
    constexpr int D = RgDepth<RG>::D;
    const int lane = threadIdx.x;
    RgStamp stamp;
    RgRing<D> ring;
    ring_start(ring, (pf_gcf)w, lane);
    float X[RG][8], Va[RG][4];
    const float zero[RG] = {};
    for (int r = 0; r < RG; ++r) {
        for (int m = 0; m < 8; ++m) X[r][m] = 0.01f * (lane + m);
        for (int t = 0; t < 4; ++t) Va[r][t] = 0.02f * (lane - t);
    }
    f32x4 slo[RG], shi[RG], Vd[RG];
    RgCarry<RG> carry;
    RgWave wv{0, 0};
    rg_gvp<SpecGen, RG, D, 0, false>(ring, X, Va, zero, zero, slo, shi, carry, lds, lane, stamp, wv);
    for (int b = 1; b < nblk; ++b) rg_gvp<SpecGen, RG, D, 1, false>(ring, X, Va, zero, zero, slo, shi, carry, lds, lane, stamp, wv);
    rg_flush<RG, D, true, true>(ring, X, Va, Vd, carry, lds, lane, stamp);
    float s = 0.f;
    for (int r = 0; r < RG; ++r) s += X[r][0] + Va[r][1] + Vd[r][2] + slo[r][0] + shi[r][1];
    out[blockIdx.x * 64 + lane] = s;
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    const int nblk = 6;
    const size_t floats = (size_t)(96 * nblk + 24 + 64) * 256;
    float* dw; hipMalloc(&dw, floats * 4); hipMemset(dw, 0, floats * 4);
    float* dout; hipMalloc(&dout, 4096 * 64 * 4);
    unsigned long long* dc; hipMalloc(&dc, 4096 * 8); unsigned long long hc[4096];
#ifdef PF_STAMPS
    unsigned long long* sb; hipMalloc(&sb, 64 * 64 * 8); hipMemset(sb, 0, 64 * 64 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_rg_stamps), &sb, sizeof(sb));
    hipLaunchKernelGGL((k_chain<1>), 48, 64, 0, 0, dw, dout, nblk, dc); hipDeviceSynchronize();
    hipLaunchKernelGGL((k_chain<1>), 48, 64, 0, 0, dw, dout, nblk, dc); hipDeviceSynchronize();
    unsigned long long hs[64 * 64]; hipMemcpy(hs, sb, sizeof(hs), hipMemcpyDeviceToHost);
    for (int w = 0; w < 2; ++w) { printf("RG=1 wave %d:", w); for (int k = 1; k < 64 && hs[w * 64 + k]; ++k) printf(" %llu", hs[w * 64 + k] - hs[w * 64 + k - 1]); printf("\n"); }
    hipMemset(sb, 0, 64 * 64 * 8);
    hipLaunchKernelGGL((k_chain<2>), 48, 64, 0, 0, dw, dout, nblk, dc); hipDeviceSynchronize();
    hipMemcpy(hs, sb, sizeof(hs), hipMemcpyDeviceToHost);
    for (int w = 0; w < 2; ++w) { printf("RG=2 wave %d:", w); for (int k = 1; k < 64 && hs[w * 64 + k]; ++k) printf(" %llu", hs[w * 64 + k] - hs[w * 64 + k - 1]); printf("\n"); }
    return 0;
#endif
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rg = 1; rg <= 2; ++rg)
        for (int waves : {48, 480, 1700}) {
            float tot = 0.f; int cnt = 0;
            for (int rep = 0; rep < 12; ++rep) {
                hipEventRecord(e0, 0);
                if (rg == 1) hipLaunchKernelGGL((k_chain<1>), waves, 64, 0, 0, dw, dout, nblk, dc);
                else hipLaunchKernelGGL((k_chain<2>), waves, 64, 0, 0, dw, dout, nblk, dc);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep >= 2) { tot += ms; ++cnt; }
            }
            hipMemcpy(hc, dc, waves * 8, hipMemcpyDeviceToHost);
            double cs = 0; for (int i = 0; i < waves; ++i) cs += hc[i]; cs /= waves;
            printf("   mean wave lifetime %.0f s_memtime ticks = %.1f per MFMA\n", cs, cs / (nblk * 352.0 * rg));
            const double us = tot / cnt * 1e3 - 6.0;
            printf("RG=%d waves=%4d : %7.2f us total, ~%.2f us per GVP block, %.1f cycles/MFMA at 2.4 GHz (352 MFMAs x RG per block)\n", rg, waves,
                   tot / cnt * 1e3, us / nblk, us * 2400.0 / (nblk * 352.0 * rg));
        }
    return 0;
}
