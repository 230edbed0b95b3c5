// Delivery rate of the scalar cache on gfx950: every wave of a full chip streams 32-byte scalar loads (s_load_dwordx8) from a table
// of a given size, five in flight per wait, the way the resampler's multiply-add loop reads its tap weights.
// Prints bytes per clock per CU at 16 waves per CU for table sizes below and above the cache.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(4))) const f32x8 const_f32x8;
__global__ void __launch_bounds__(256) probe(const float* table, int n_vec /* 32-byte entries, a multiple of 5 */, int iters, int per_wave_offset, float* out) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6) + 4 * (int)blockIdx.x);
    const_f32x8* w = (const_f32x8*)table;
    float acc = 0.f;
    int pos = (wave * per_wave_offset) % n_vec;
    for (int it = 0; it < iters; ++it) {
        const f32x8 a = w[pos], b = w[pos + 1], c = w[pos + 2], d = w[pos + 3], e = w[pos + 4];
        pos += 5;
        if (pos >= n_vec) pos = 0;
        float s = a[0] + b[1] + c[2] + d[3] + e[4];     // scalar adds: the loads must complete, no vector work to speak of
        acc += s;
    }
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}
int main() {
    const int n_max = 1 << 20;
    float* t; float* o;
    (void)hipMalloc(&t, n_max * 32); (void)hipMemset(t, 0, n_max * 32); (void)hipMalloc(&o, 4096 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 256 * 4, iters = 20000;
    const int sizes[] = {45, 180, 450, 1440, 4500, 45000};      // entries of 32 B: 1.4 KB, 5.8 KB, 14 KB, 46 KB, 144 KB, 1.4 MB
    for (int off : {0, 45}) {
        for (int n : sizes) {
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipEventRecord(e0);
                probe<<<blocks, 256>>>(t, n, iters, off, o);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&ms, e0, e1);
            }
            const double bytes_per_cu = 16.0 * iters * 5 * 32;
            printf("table %8.1f KB  wave offset %2d entries: %.3f ms  %.2f GB/s per CU (%.2f B/clk at 2.4 GHz), %.1f ns per 5-load round per wave\n", n * 32 / 1024.0, off, ms,
                   bytes_per_cu / ms / 1e6, bytes_per_cu / (ms * 1e-3) / 2.4e9, ms * 1e6 / iters);
        }
    }
    return 0;
}
