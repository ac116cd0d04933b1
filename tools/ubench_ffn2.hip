// tools/ubench_ffn2.hip — timing ablation + per-phase timers of hat_ffn2's kernel (NOT part of the library or the tests).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -w tools/ubench_ffn2.hip -o tools/bin/ubench_ffn2 && tools/bin/ubench_ffn2
#define HAT_FFN2_NO_ENTRY
#include "../super_resolution_amd/csrc/hat_ffn2.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 1; } } while (0)

__global__ void fill(float* p, size_t n, float s) { size_t i = blockIdx.x * (size_t)256 + threadIdx.x; if (i < n) p[i] = s * (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f * s; }
__global__ void fillb(bf16_t* p, size_t n, float s) { size_t i = blockIdx.x * (size_t)256 + threadIdx.x; if (i < n) p[i] = (bf16_t)(s * (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f * s); }
__global__ void fillh(_Float16* p, size_t n, float s) { size_t i = blockIdx.x * (size_t)256 + threadIdx.x; if (i < n) p[i] = (_Float16)(s * (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f * s); }

static F2Aggr g_ag = {};
template <int DBG, bool AG = false> float run(const HatFfnDesc& d, int iters) {
    auto kern = ffn2_kernel<DBG, AG>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, F2_LDS);
    dim3 grid((d.W + 15) / 16, (d.H + 7) / 8, d.B);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, grid, dim3(256), F2_LDS, 0, d, g_ag);
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, grid, dim3(256), F2_LDS, 0, d, g_ag);
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    return ms / iters;
}

int main() {
    const int H = 720, W = 1280, C = 144, chunks = 9;
    const size_t N = (size_t)H * W;
    float *tin, *tout, *vec; bf16_t *w1f, *nout; _Float16 *w2f, *dww;
    CK(hipMalloc(&tin, N * C * 4)); CK(hipMalloc(&tout, N * C * 4)); CK(hipMalloc(&vec, 8192 * 4));
    CK(hipMalloc(&w1f, (size_t)chunks * 4 * 5 * 64 * 8 * 2)); CK(hipMalloc(&w2f, (size_t)chunks * 9 * 64 * 8 * 2));
    CK(hipMalloc(&dww, (size_t)chunks * 640 * 2)); CK(hipMalloc(&nout, N * C * 2));
    fill<<<(N * C + 255) / 256, 256>>>(tin, N * C, 2.f);
    fill<<<32, 256>>>(vec, 8192, 0.2f);
    fillb<<<(chunks * 4 * 5 * 64 * 8 + 255) / 256, 256>>>(w1f, (size_t)chunks * 4 * 5 * 64 * 8, 0.15f);
    fillh<<<(chunks * 9 * 64 * 8 + 255) / 256, 256>>>(w2f, (size_t)chunks * 9 * 64 * 8, 0.1f);
    fillh<<<(chunks * 640 + 255) / 256, 256>>>(dww, (size_t)chunks * 640, 0.3f);
    CK(hipDeviceSynchronize());
    HatFfnDesc d = {};
    d.t_in = tin; d.t_out = tout; d.ln_g = vec; d.ln_b = vec + 256; d.w1f = w1f; d.b1 = vec + 512; d.dww = dww; d.dwb = vec + 2048;
    d.w2f = w2f; d.b2 = vec + 4096; d.B = 1; d.H = H; d.W = W; d.C = C; d.chunks = chunks; d.dtype = HAT_BF16;
    d.ln1_g = vec + 5000; d.ln1_b = vec + 5300; d.n_out = nout; d.ldn = C;
    {   // inputs of the fused aggregation stage
        bf16_t *n, *y16, *c1, *wl, *wf; float* bb;
        CK(hipMalloc(&n, N * C * 2)); CK(hipMalloc(&y16, N * 16 * 2)); CK(hipMalloc(&c1, N * 8 * 2));
        CK(hipMalloc(&wl, 9 * 5 * 1024)); CK(hipMalloc(&wf, 9 * 3 * 1024)); CK(hipMalloc(&bb, 144 * 4));
        fillb<<<(N * C + 255) / 256, 256>>>(n, N * C, 2.f); fillb<<<(N * 16 + 255) / 256, 256>>>(y16, N * 16, 2.f);
        fillb<<<(N * 8 + 255) / 256, 256>>>(c1, N * 8, 1.f); fillb<<<(9 * 5 * 512 + 255) / 256, 256>>>(wl, 9 * 5 * 512, 0.15f);
        fillb<<<(9 * 3 * 512 + 255) / 256, 256>>>(wf, 9 * 3 * 512, 0.05f); fill<<<1, 256>>>(bb, 144, 0.1f);
        CK(hipDeviceSynchronize());
        g_ag = F2Aggr{n, y16, c1, (const char*)wl, (const char*)wf, bb, C};
    }
    const int it = 60;
    run<0>(d, 200);   // clocks up
    if (getenv("UB_PHASES")) {
        run<0>(d, 200);
        const size_t nwg = (size_t)((W + 15) / 16) * ((H + 7) / 8);
        float* ph; CK(hipMalloc(&ph, nwg * 4 * 12 * 4));
        d.gap_out = ph;
        const bool ag = getenv("UB_AGGR") != nullptr;
        printf("instrumented run (%s) %.3f ms\n", ag ? "fused aggregation" : "plain", ag ? run<64, true>(d, 2) : run<64>(d, 2));
        std::vector<float> hp(nwg * 4 * 12);
        CK(hipMemcpy(hp.data(), ph, hp.size() * 4, hipMemcpyDeviceToHost));
        const char* names[12] = {"LN stage (rest: LN + Ms)", "barrier after LN", "phase A (fc1)", "barrier after A", "phase B (dw, VALU)", "phase C (gate+fc2)", "barrier after C", "epilogue",
                                 "s0: issue loads", "s0: wait loads+barrier", "s0: MFMA", "s0: barrier 2"};
        double tot[12] = {0}; double all = 0;
        for (size_t i = 0; i < nwg * 4; ++i) for (int k = 0; k < 12; ++k) { tot[k] += hp[i * 12 + k]; all += hp[i * 12 + k]; }
        for (int k = 0; k < 12; ++k) printf("  %-22s %10.0f cycles/wave  (%.1f%%)\n", names[k], tot[k] / (nwg * 4), 100.0 * tot[k] / all);
        printf("  total %.0f cycles/wave\n", all / (nwg * 4));
        return 0;
    }
#define ROW(M, label) printf("%-34s %.3f ms\n", label, run<M>(d, it))
    printf("%-34s %.3f ms\n", "full, fused aggregation stage", run<0, true>(d, it));
    printf("%-34s %.3f ms\n", "  '' , no LN/aggr stage", run<1, true>(d, it));
    ROW(0, "full (+ fused next LN)");
    ROW(1, "no LN stage");
    ROW(2, "no fc1 mfma");
    ROW(4, "no dw fma");
    ROW(8, "no gate math");
    ROW(16, "no fc2 mfma");
    ROW(32, "no weight loads");
    ROW(2 | 16, "no mfma at all");
    ROW(2 | 4 | 8 | 16 | 32, "LN + skeleton + epilogue");
    ROW(1 | 2 | 4 | 8 | 16 | 32, "skeleton + epilogue");
    return 0;
}
