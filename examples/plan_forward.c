/* examples/plan_forward.c — a host without Python: load a forward plan, run it, read the result.
 *
 *   python -m super_resolution_amd.plan -opt options/test/HAT-S_SRx4.yml --shape 1 720 1280 -o hats_720p.hatplan   (once)
 *   gcc examples/plan_forward.c -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -Lsuper_resolution_amd -lhat_mi355x \
 *       -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/super_resolution_amd -o plan_forward
 *   ./plan_forward hats_720p.hatplan
 *
 * Only the C ABI of include/hat_mi355x.h and the HIP runtime are used.
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "hat_mi355x.h"

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s net.hatplan [iterations]\n", argv[0]); return 2; }
    const int iters = argc > 2 ? atoi(argv[2]) : 5;
    hat_plan* plan = NULL;
    int rc = hat_plan_load(argv[1], &plan);
    if (rc) { fprintf(stderr, "hat_plan_load failed: %d\n", rc); return 1; }
    int32_t d[8];
    int64_t launches = 0, bytes = 0;
    hat_plan_info(plan, d, &launches, &bytes);
    const size_t nx = (size_t)d[0] * d[1] * d[2] * d[3], ny = (size_t)d[0] * d[5] * d[2] * d[4] * d[3] * d[4];
    printf("plan: input %dx%dx%dx%d, x%d, %lld launches, %.1f MB on the device\n", d[0], d[1], d[2], d[3], d[4], (long long)launches, bytes / 1e6);
    float *hx = (float*)malloc(nx * 4), *hy = (float*)malloc(ny * 4), *dx = NULL, *dy = NULL;
    for (size_t i = 0; i < nx; ++i) hx[i] = (float)((i * 2654435761u) % 1000) / 1000.0f;   /* any image in [0, 1) */
    if (hipMalloc((void**)&dx, nx * 4) || hipMalloc((void**)&dy, ny * 4) || hipMemcpy(dx, hx, nx * 4, hipMemcpyHostToDevice)) return 1;
    hipStream_t s;
    hipEvent_t e0, e1;
    hipStreamCreate(&s); hipEventCreate(&e0); hipEventCreate(&e1);
    rc = hat_plan_forward(plan, dx, dy, s);                  /* warm-up */
    hipEventRecord(e0, s);
    for (int i = 0; i < iters && !rc; ++i) rc = hat_plan_forward(plan, dx, dy, s);
    hipEventRecord(e1, s);
    if (rc || hipStreamSynchronize(s)) { fprintf(stderr, "forward failed: %d\n", rc); return 1; }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(hy, dy, ny * 4, hipMemcpyDeviceToHost);
    double sum = 0;
    for (size_t i = 0; i < ny; ++i) sum += hy[i];
    printf("%.3f ms per forward, mean of the output %.6f\n", ms / iters, sum / (double)ny);
    hat_plan_free(plan);
    hipFree(dx); hipFree(dy); free(hx); free(hy);
    return 0;
}
