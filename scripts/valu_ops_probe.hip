// Issue rate of individual wave64 VALU operations on gfx950, 4 waves per SIMD, 16 independent chains per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ void __launch_bounds__(256) probe(float* out, int iters) {
    float a[16];
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
    float m = 1.0001f + threadIdx.x * 1e-9f, c = 0.0001f;
    const float sm = __builtin_amdgcn_readfirstlane(iters) * 1e-9f + 1.0f;
    float sw[8];
    for (int i = 0; i < 8; ++i) sw[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sm + i * 1e-6f)));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                if (OP == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (OP == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (OP == 3) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (OP == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c));
                if (OP == 5) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (OP == 6) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 1) & 15]));
                if (OP == 7) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (OP == 8) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                if (OP == 9) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 15]));
                if (OP == 10) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(*reinterpret_cast<double*>(&a[2 * (i & 7)])) : "v"(*reinterpret_cast<double*>(&m)));
                if (OP == 11) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
                if (OP == 12) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(sm), "v"(c));      // scalar-register multiplicand
                if (OP == 13) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(sw[i & 7]), "v"(c));  // eight different scalar registers in turn
            }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP>
void run(const char* name, float* d) {
    const int iters = 20000, wps = 4, blocks = 256 * wps;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        probe<OP><<<blocks, 256>>>(d, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double instr = (double)iters * 32 * wps;   // per SIMD
    printf("%-14s %.3f ms  %.3f ns per wave64 instruction per SIMD\n", name, ms, ms * 1e6 / instr);
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 4 * 256 * 4);
    run<0>("v_fma_f32", d); run<1>("v_add_f32", d); run<2>("v_mul_f32", d); run<3>("v_sub_f32", d);
    run<4>("v_cndmask_b32", d); run<5>("v_add_u32", d); run<6>("v_mov_dpp", d); run<7>("v_max_f32", d);
    run<8>("v_fmac_f32", d); run<9>("v_mov_b32", d); run<10>("v_fma_f64", d); run<11>("v_log_f32", d);
    run<12>("v_fmac s,v", d); run<13>("v_fmac s[i],v", d);
    run<0>("v_fma_f32", d);
    return 0;
}
