// Issue rate of v_mfma_f64_16x16x4_f64 on gfx950: back-to-back MFMAs on two independent accumulators per wave,
// no memory traffic.  Prints TFLOP/s for 1, 2 and 4 waves per SIMD (2048 flop per instruction).
//   hipcc -O3 --offload-arch=gfx950 scripts/mfma_f64_probe.hip -o build/mfma_f64_probe && build/mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double f64x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) probe(double* out, int iters, double a0, double b0) {
    f64x4 x = {0, 0, 0, 0}, y = {0, 0, 0, 0};
    double a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, x, 0, 0, 0);
            y = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, y, 0, 0, 0);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x[0] + x[1] + x[2] + x[3] + y[0] + y[1] + y[2] + y[3];
}

int main() {
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    double* out;
    hipMalloc(&out, sizeof(double) * 256 * cus * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    for (int per_cu = 1; per_cu <= 4; per_cu *= 2) {
        const int grid = cus * per_cu;
        probe<<<grid, 256>>>(out, 100, 1.0, 2.0);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        probe<<<grid, 256>>>(out, iters, 1.0, 2.0);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double n_mfma = (double)grid * 4 * iters * 16;
        printf("%d wave(s)/SIMD: %.3f ms, %.1f TFLOP/s, %.1f ns per MFMA per SIMD\n", per_cu, ms, n_mfma * 2048 / (ms * 1e-3) / 1e12,
               ms * 1e6 / ((double)per_cu * iters * 16));
    }
    return 0;
}
