// tools/ubench_tail3.hip — timing + per-phase timers of hat_hab_tail3's kernel (NOT part of the library or the tests).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -w tools/ubench_tail3.hip -o tools/bin/ubench_tail3 && tools/bin/ubench_tail3
#define HAT_TAIL3_NO_ENTRY
#include "../super_resolution_amd/csrc/hat_tail3.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 1; } } while (0)

__global__ void fill(float* p, size_t n, float s) { size_t i = blockIdx.x * (size_t)256 + threadIdx.x; if (i < n) p[i] = s * (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f * s; }
__global__ void fillb(bf16_t* p, size_t n, float s) { size_t i = blockIdx.x * (size_t)256 + threadIdx.x; if (i < n) p[i] = (bf16_t)(s * (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f * s); }
__global__ void fillh(_Float16* p, size_t n, float s) { size_t i = blockIdx.x * (size_t)256 + threadIdx.x; if (i < n) p[i] = (_Float16)(s * (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f * s); }

static T3Aggr g_ag = {};
static int g_lds = T3_LDS;     // > 81 920: one workgroup per CU (occupancy experiment)
template <int DBG> float run(const HatFfnDesc& d, int iters) {
    auto kern = tail3_kernel<DBG>;
    const int T3_LDS = g_lds;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, T3_LDS);
    dim3 grid((d.W + 15) / 16, (d.H + 7) / 8, d.B);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, grid, dim3(256), T3_LDS, 0, d, g_ag);
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, grid, dim3(256), T3_LDS, 0, d, g_ag);
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    return ms / iters;
}

int main(int argc, char** argv) {
    if (argc > 1) g_lds = atoi(argv[1]);
    printf("dynamic LDS %d B per workgroup\n", g_lds);
    const int H = 720, W = 1280, C = 144, chunks = 9;
    const size_t N = (size_t)H * W;
    float *tin, *tout, *vec; bf16_t *w1f, *nout; _Float16 *w2f, *dww;
    CK(hipMalloc(&tin, N * C * 4)); CK(hipMalloc(&tout, N * C * 4)); CK(hipMalloc(&vec, 8192 * 4));
    CK(hipMalloc(&w1f, (size_t)chunks * 4 * 5 * 64 * 8 * 2)); CK(hipMalloc(&w2f, (size_t)chunks * 9 * 64 * 8 * 2));
    CK(hipMalloc(&dww, (size_t)chunks * 2048)); CK(hipMalloc(&nout, N * C * 2));
    fill<<<(N * C + 255) / 256, 256>>>(tin, N * C, 2.f);
    fill<<<32, 256>>>(vec, 8192, 0.2f);
    fillb<<<(chunks * 4 * 5 * 64 * 8 + 255) / 256, 256>>>(w1f, (size_t)chunks * 4 * 5 * 64 * 8, 0.15f);
    fillh<<<(chunks * 9 * 64 * 8 + 255) / 256, 256>>>(w2f, (size_t)chunks * 9 * 64 * 8, 0.1f);
    fillh<<<(chunks * 1024 + 255) / 256, 256>>>(dww, (size_t)chunks * 1024, 0.3f);
    CK(hipDeviceSynchronize());
    HatFfnDesc d = {};
    d.t_in = tin; d.t_out = tout; d.ln_g = vec; d.ln_b = vec + 256; d.w1f = w1f; d.b1 = vec + 512; d.dww = dww; d.dwb = vec + 2048;
    d.w2f = w2f; d.b2 = vec + 4096; d.B = 1; d.H = H; d.W = W; d.C = C; d.chunks = chunks; d.dtype = HAT_BF16;
    d.ln1_g = vec + 5000; d.ln1_b = vec + 5300; d.n_out = nout; d.ldn = C;
    {
        bf16_t *n, *y16, *c1, *wl, *wf; float* bb;
        CK(hipMalloc(&n, N * C * 2)); CK(hipMalloc(&y16, N * 16 * 2)); CK(hipMalloc(&c1, N * 8 * 2));
        CK(hipMalloc(&wl, 9 * 5 * 1024)); CK(hipMalloc(&wf, 9 * 3 * 1024)); CK(hipMalloc(&bb, 144 * 4));
        fillb<<<(N * C + 255) / 256, 256>>>(n, N * C, 2.f); fillb<<<(N * 16 + 255) / 256, 256>>>(y16, N * 16, 2.f);
        fillb<<<(N * 8 + 255) / 256, 256>>>(c1, N * 8, 1.f); fillb<<<(9 * 5 * 512 + 255) / 256, 256>>>(wl, 9 * 5 * 512, 0.15f);
        fillb<<<(9 * 3 * 512 + 255) / 256, 256>>>(wf, 9 * 3 * 512, 0.05f); fill<<<1, 256>>>(bb, 144, 0.1f);
        CK(hipDeviceSynchronize());
        g_ag = T3Aggr{n, y16, c1, (const char*)wl, (const char*)wf, bb, C};
    }
    run<0>(d, 200);   // clocks up
    printf("tail3 full                          %.3f ms\n", run<0>(d, 60));
    printf("tail3 full                          %.3f ms\n", run<0>(d, 60));
    const size_t nwg = (size_t)((W + 15) / 16) * ((H + 7) / 8);
    float* ph; CK(hipMalloc(&ph, nwg * 4 * 12 * 4));
    d.gap_out = ph;
    printf("instrumented run %.3f ms\n", run<64>(d, 2));
    std::vector<float> hp(nwg * 4 * 12);
    CK(hipMemcpy(hp.data(), ph, hp.size() * 4, hipMemcpyDeviceToHost));
    const char* names[12] = {"s0: issue fc1(0) copies", "barrier before loop", "phase A (fc1)", "barrier after A", "phase B (dw, VALU)", "phase C (gate+fc2)", "barrier after C", "epilogue",
                             "s0: issue loads", "s0: wait copies+barrier", "s0: MFMA + LN x3", "s0: barrier 2"};
    double tot[12] = {0}; double all = 0;
    for (size_t i = 0; i < nwg * 4; ++i) for (int k = 0; k < 12; ++k) { tot[k] += hp[i * 12 + k]; all += hp[i * 12 + k]; }
    for (int k = 0; k < 12; ++k) printf("  %-26s %10.0f cycles/wave  (%.1f%%)\n", names[k], tot[k] / (nwg * 4), 100.0 * tot[k] / all);
    printf("  total %.0f cycles/wave\n", all / (nwg * 4));
    return 0;
}
