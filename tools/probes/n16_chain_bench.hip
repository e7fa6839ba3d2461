// n16_chain_bench.hip -- diagnostic: the chain code of pf_n16.hip (n16_block: 16-row items on four waves) in a bare
// harness, with in-kernel cycle stamps.  Answers: what does a GVP block cost per phase, alone on the chip and with 1-4
// items per CU, with the weights L2-cold (after 1 GiB of other traffic) and L2-warm (second launch in a row)?
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 -DN16_STAMPS [-DN16_D=24] \
//         -I pharmacophore-diffusion_amd/csrc tools/probes/n16_chain_bench.hip -o n16_chain_bench && ./n16_chain_bench
// Stamps per wave: item start | per block: start, main k-steps issued, barrier A passed, gate k-steps issued, barrier B
// passed | chain done.  s_memtime ticks at 100 MHz x ... are converted with the measured kernel duration.
#include "pf_n16.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int n_gvps = argc > 1 ? atoi(argv[1]) : 3;
    const int kind = argc > 2 ? atoi(argv[2]) : 16;
    const int nq_chain = (kind == 16 ? n16_sched(N16_M0F).nq : n16_sched(N16_GEN).nq) + (n_gvps - 1) * n16_sched(N16_GEN).nq;
    const size_t stride = (size_t)(nq_chain + N16_TAIL_PAD) * 256;
    const int max_wg = 2048;
    std::vector<float> hs(4 * stride);
    srand(1);
    for (auto& x : hs) x = (rand() % 2001 - 1000) * 5e-5f;
    const int sw = kind == 16 ? 144 : 128, vw = kind == 16 ? 51 : 48;
    std::vector<float> hsin((size_t)max_wg * 16 * sw), hvin((size_t)max_wg * 16 * vw);
    for (auto& x : hsin) x = (rand() % 2001 - 1000) * 1e-3f;
    for (auto& x : hvin) x = (rand() % 2001 - 1000) * 1e-3f;
    float *dstream, *dsin, *dvin, *dso, *dvo, *dflush;
    unsigned long long* dst;
    CK(hipMalloc(&dstream, hs.size() * 4)); CK(hipMalloc(&dsin, hsin.size() * 4)); CK(hipMalloc(&dvin, hvin.size() * 4));
    CK(hipMalloc(&dso, (size_t)max_wg * 16 * 128 * 4)); CK(hipMalloc(&dvo, (size_t)max_wg * 16 * 48 * 4));
    CK(hipMalloc(&dst, 64 * 4 * 64 * 8)); CK(hipMalloc(&dflush, (size_t)1 << 30));
    CK(hipMemcpy(dstream, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsin, hsin.data(), hsin.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dvin, hvin.data(), hvin.size() * 4, hipMemcpyHostToDevice));
#ifdef N16_STAMPS
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_n16_stamps), &dst, sizeof(dst)));
#endif
    UnitParams p{};
    p.s_in = dsin; p.v_in = dvin; p.s_out = dso; p.v_out = dvo; p.kind = kind; p.n_gvps = n_gvps; p.stream = dstream; p.n16_stride = (int)stride;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("chain: kind %d, %d GVPs, %d quads per wave, ring depth %d\n", kind, n_gvps, nq_chain, N16_D);
    std::vector<unsigned long long> st(64 * 4 * 64);
    for (int nwg : {1, 32, 128, 256, 512, 1024}) {
        for (int warm = 0; warm < 2; ++warm) {
            p.n = nwg * 16;
            CK(hipMemset(dst, 0, 64 * 4 * 64 * 8));
            if (!warm) CK(hipMemset(dflush, 1, (size_t)1 << 30));
            else hipLaunchKernelGGL(k_n16_unit, dim3(nwg), dim3(256), 0, 0, p);
            CK(hipDeviceSynchronize());
            CK(hipMemset(dst, 0, 64 * 4 * 64 * 8));
            CK(hipEventRecord(e0));
            for (int rep = 0; rep < (warm ? 10 : 1); ++rep) hipLaunchKernelGGL(k_n16_unit, dim3(nwg), dim3(256), 0, 0, p);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (warm) ms /= 10;
            CK(hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost));
            const int rec = std::min(nwg, 64);
            // s_memtime runs at a fixed 100 MHz on gfx950: report ticks and the chain's share of the launch
            double tot_max = 0, tot_avg = 0;
            int cnt = 0;
            for (int b = 0; b < rec; ++b)
                for (int w = 0; w < 4; ++w) {
                    const unsigned long long* r = &st[((size_t)b * 4 + w) * 64];
                    int n = 0;
                    while (n < 64 && r[n]) ++n;
                    if (n < 2) continue;
                    const double t = (double)(r[n - 1] - r[0]);
                    tot_max = std::max(tot_max, t); tot_avg += t; ++cnt;
                }
            printf("wgs %4d %s: launch %.2f us; item (first -> last stamp) avg %.0f max %.0f ticks\n", nwg, warm ? "warm" : "cold", ms * 1e3, tot_avg / std::max(cnt, 1), tot_max);
            for (int b : {0, rec - 1}) {
                for (int w : {0, 3}) {
                    const unsigned long long* r = &st[((size_t)b * 4 + w) * 64];
                    int n = 0;
                    while (n < 64 && r[n]) ++n;
                    printf("   wg %2d wave %d:", b, w);
                    for (int i = 0; i + 1 < n; ++i) printf(" %llu", r[i + 1] - r[i]);
                    printf("\n");
                }
                if (rec == 1) break;
            }
        }
    }
    return 0;
}
