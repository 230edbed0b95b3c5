// Probe of the lane layout of v_mfma_f32_4x4x1_16b_f32 on gfx950: prints D for a = 1+lane, b = 100*(1+lane).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out) {
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    const float a = 1.0f + l, b = 1000.0f + 13.0f * l;
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
int main() {
    float* d; hipMalloc(&d, 256 * 4);
    probe<<<1, 64>>>(d);
    float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // hypothesis: block = l/4; D[i][j] (i = a-row, j = b-col) in lane 4*block + j, register i; A[i] from lane 4*block+i, B[j] from lane 4*block+j
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        const int blk = l / 4, j = l % 4, i = r;
        const float expect = (1.0f + (4 * blk + i)) * (1000.0f + 13.0f * (4 * blk + j));
        if (h[l * 4 + r] != expect) { if (bad < 8) printf("lane %d reg %d: got %g expect %g\n", l, r, h[l*4+r], expect); ++bad; }
    }
    printf("mismatches: %d\n", bad);
    for (int l = 0; l < 8; ++l) printf("lane %d: %g %g %g %g\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    return 0;
}
