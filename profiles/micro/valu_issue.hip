// VALU issue-rate probe: W waves per SIMD, each running a stream of independent v_fma_f32 /
// v_cndmask / v_cmp.  Prints wave-instructions per cycle per SIMD for W = 1..8.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int KIND>
__global__ void __launch_bounds__(256) probe(float *out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (KIND == 0) {
                x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
                x4 = fmaf(x4, a, b); x5 = fmaf(x5, a, b); x6 = fmaf(x6, a, b); x7 = fmaf(x7, a, b);
            } else {
                x0 = (x1 < a) ? x2 : x0; x1 = (x2 < b) ? x3 : x1; x2 = (x3 < a) ? x4 : x2; x3 = (x4 < b) ? x5 : x3;
                x4 = (x5 < a) ? x6 : x4; x5 = (x6 < b) ? x7 : x5; x6 = (x7 < a) ? x0 : x6; x7 = (x0 < b) ? x1 : x7;
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float *d;
    hipMalloc(&d, sizeof(float) * 256 * cus * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int kind = 0; kind < 2; kind++)
        for (int w = 1; w <= 8; w++) {
            // w workgroups of 256 threads per CU = w waves per SIMD
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (kind == 0) hipLaunchKernelGGL(probe<0>, dim3(cus * w), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
                else hipLaunchKernelGGL(probe<1>, dim3(cus * w), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double insts = (double)iters * 64 * (kind == 0 ? 1 : 2);   // wave-instructions per wave
            const double clk = p.clockRate * 1e3;                             // Hz
            printf("kind=%d waves/SIMD=%d  %.3f ms  -> %.3f wave-instr/cycle/SIMD (clock %.0f MHz)\n", kind, w, ms,
                   insts * w / (ms * 1e-3 * clk), clk / 1e6);
        }
    return 0;
}
