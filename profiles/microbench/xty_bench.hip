// mvf_xty alone on 2 M rows: whole kernel, memory side alone (-DMVF_XTY_NO_MFMA), matrix-core
// side alone (-DMVF_XTY_NO_LOAD).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form \
//        [-DMVF_XTY_NO_MFMA | -DMVF_XTY_NO_LOAD] profiles/microbench/xty_bench.hip -o xty_bench
#include "../../multimodal-fusion_amd/csrc/mvf_xty.hip"
#include <vector>

namespace mvf {
void set_error(const char *fmt, ...) { (void)fmt; }
}

int main()
{
    const int64_t rows = 2002000;
    const int shapes[][2] = {{64, 64}, {32, 32}, {32, 64}, {16, 8}, {64, 1}};
    float *x, *y, *out;
    void *ws;
    hipMalloc(&x, rows * 64 * 4);
    hipMalloc(&y, rows * 64 * 4);
    hipMalloc(&out, 64 * 64 * 4);
    hipMalloc(&ws, mvf_xty_workspace_bytes());
    hipMemset(x, 0, rows * 64 * 4);
    hipMemset(y, 0, rows * 64 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (auto &sh : shapes) {
        const int A = sh[0], B = sh[1];
        for (int k = 0; k < 3; ++k) mvf_xty(x, A, A, y, B, B, rows, out, B, ws, mvf_xty_workspace_bytes(), nullptr);
        hipEventRecord(e0, nullptr);
        for (int k = 0; k < 20; ++k) mvf_xty(x, A, A, y, B, B, rows, out, B, ws, mvf_xty_workspace_bytes(), nullptr);
        hipEventRecord(e1, nullptr);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double us = ms / 20 * 1e3, gb = 4.0 * (A + B) * rows / 1e9;
        printf("%2d x %2d: %7.1f us  %5.2f TB/s  %6.1f TFLOP/s\n", A, B, us, gb / us * 1e3 / 1e3 * 1e3 / 1e3,
               2.0 * A * B * rows / us / 1e6);
    }
    return 0;
}
