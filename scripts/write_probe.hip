// Pure-write and mixed read/write streaming rates on gfx950 (16-byte accesses, grid-stride).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>  // 0: write only, 1: copy (read n, write n), 2: read 1 : write 4, 3: nontemporal write only
__global__ void __launch_bounds__(256) k(const float4* in, float4* out, size_t n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 v = make_float4(1.f, 2.f, 3.f, (float)i);
        if (MODE == 1) v = in[i];
        if (MODE == 2) { if ((i & 3) == 0) v = in[i >> 2]; }
        if (MODE == 3) { __builtin_nontemporal_store(v.x, &out[i].x); __builtin_nontemporal_store(v.y, &out[i].y); __builtin_nontemporal_store(v.z, &out[i].z); __builtin_nontemporal_store(v.w, &out[i].w); }
        else out[i] = v;
    }
}
int main() {
    const size_t n4 = (size_t)5 * 1024 * 1024 * 1024 / 16;   // 5 GiB
    float4 *in, *out; (void)hipMalloc(&in, n4 * 16); (void)hipMalloc(&out, n4 * 16);
    (void)hipMemset(in, 0, n4 * 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const char* names[4] = {"write only", "copy 1:1", "read 1 : write 4", "nontemporal write only"};
    for (int mode = 0; mode < 4; ++mode)
        for (int bpc = 4; bpc <= 16; bpc *= 2) {
            float best = 1e9;
            for (int rep = 0; rep < 4; ++rep) {
                (void)hipEventRecord(e0);
                if (mode == 0) k<0><<<256 * bpc, 256>>>(in, out, n4); else if (mode == 1) k<1><<<256 * bpc, 256>>>(in, out, n4);
                else if (mode == 2) k<2><<<256 * bpc, 256>>>(in, out, n4); else k<3><<<256 * bpc, 256>>>(in, out, n4);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            const double bytes = (double)n4 * 16 * (mode == 1 ? 2.0 : mode == 2 ? 1.25 : 1.0);
            printf("%-24s %2d blocks/CU: %.3f ms  %.0f GB/s\n", names[mode], bpc, best, bytes / best / 1e6);
        }
    return 0;
}
