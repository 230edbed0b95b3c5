// Measures wave64 VALU issue rate on gfx950: plain v_fma_f32 vs v_pk_fma_f32, 1..8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int PK>
__global__ void __launch_bounds__(256) probe(float* out, int iters, unsigned long long* cyc) {
    float a[16];
    f2 v[8];
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
    for (int i = 0; i < 8; ++i) v[i] = f2{a[2*i], a[2*i+1]};
    const float m = 1.0001f, c = 0.0001f;
    const f2 m2 = {m, m}, c2 = {c, c};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (PK) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __builtin_elementwise_fma(v[i], m2, c2);
        } else {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) a[i] = __builtin_fmaf(a[i], m, c);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a[i];
    for (int i = 0; i < 8; ++i) s += v[i].x + v[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
    float* d; unsigned long long* dc;
    (void)hipMalloc(&d, 256 * 8 * 256 * 4 * 8); (void)hipMalloc(&dc, 8);
    const int iters = 20000;
    for (int pk = 0; pk < 2; ++pk)
        for (int wps = 1; wps <= 8; wps *= 2) {
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            const int blocks = 256 * wps;   // 256-thread blocks = 1 wave per SIMD each
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipEventRecord(e0);
                if (pk) probe<1><<<blocks, 256>>>(d, iters, dc); else probe<0><<<blocks, 256>>>(d, iters, dc);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            }
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            unsigned long long cyc; (void)hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost);
            const double instr_per_wave = (double)iters * 32;
            printf("pk=%d waves/SIMD=%d: %.3f ms, memtime ticks %llu (%.2f ticks per instr per wave), wall-based SIMD cycles/instr at 2.4GHz: %.2f\n",
                   pk, wps, ms, cyc, cyc / instr_per_wave, ms * 1e-3 * 2.4e9 / (instr_per_wave * wps));
        }
    return 0;
}
