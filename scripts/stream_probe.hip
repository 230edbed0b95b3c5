// Achievable HBM rate for the fbank access pattern: each 16-lane group reads a 1600-byte frame
// (NA loads of W bytes per lane) and writes 512 bytes, persistent waves, one frame-quad prefetched ahead.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int W, int ST>  // W: bytes per lane per load (8 or 16); ST: 0 = 8 dword stores/lane (64-B runs), 1 = 2 dwordx4 stores/lane (256-B runs), 2 = no stores
__global__ void __launch_bounds__(256) probe(const float* in, float* out, int n_quads, int work) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fl = lane >> 4, b = lane & 15;
    constexpr int NL = (W == 8) ? 13 : 7;
    float acc[16 / 4 * (W / 4)];  // W/4 floats per load
    int q = blockIdx.x * 4 + wave;
    const int qs = gridDim.x * 4;
    float nxt[NL][W / 4];
    auto load = [&](int qq) {
        const float* fp = in + ((size_t)qq * 4 + fl) * 400;
#pragma unroll
        for (int a = 0; a < NL; ++a) {
            const int idx = (W / 4) * 16 * a + (W / 4) * b;
            const bool ok = idx + (W / 4) <= 400;
            const float* p = ok ? fp + idx : in;
            if (W == 8) { float2 r = *(const float2*)p; nxt[a][0] = ok ? r.x : 0; nxt[a][1] = ok ? r.y : 0; }
            else { float4 r = *(const float4*)p; nxt[a][0] = ok ? r.x : 0; nxt[a][1] = ok ? r.y : 0; nxt[a][2] = ok ? r.z : 0; nxt[a][3] = ok ? r.w : 0; }
        }
    };
    if (q < n_quads) load(q);
    for (; q < n_quads; q += qs) {
        float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int a = 0; a < NL; ++a)
#pragma unroll
            for (int e = 0; e < W / 4; ++e) s[(a + e) & 7] += nxt[a][e];
        if (q + qs < n_quads) load(q + qs);
        // synthetic dependent ALU work
        for (int it = 0; it < work; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) s[i] = __builtin_fmaf(s[i], 1.0001f, 0.5f);
        if (ST == 0) {
            float* o = out + ((size_t)q * 4 + fl) * 128 + b;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[16 * i] = s[i];
        } else if (ST == 1) {
            float4* o = (float4*)(out + ((size_t)q * 4 + fl) * 128) + b;
            o[0] = make_float4(s[0], s[1], s[2], s[3]);
            o[16] = make_float4(s[4], s[5], s[6], s[7]);
        } else if (ST == 3) {
            float* o = out + ((size_t)q * 4 + fl) * 128 + 4 * b;
#pragma unroll
            for (int i = 0; i < 4; ++i) __builtin_nontemporal_store(s[i], o + i);
#pragma unroll
            for (int i = 0; i < 4; ++i) __builtin_nontemporal_store(s[4 + i], o + 64 + i);
        } else if (ST == 4) {
            float* o = out + ((size_t)q * 4 + fl) * 128 + b;
#pragma unroll
            for (int i = 0; i < 8; ++i) __builtin_nontemporal_store(s[i], o + 16 * i);
        } else {
            if (s[0] == 123.456f) out[q] = s[1] + s[2] + s[3] + s[4] + s[5] + s[6] + s[7];
        }
    }
}
int main() {
    const int B = 4096, frames = 384;
    const size_t n_in = (size_t)B * frames * 400 + 4096, n_out = (size_t)B * frames * 128;
    float *in, *out;
    (void)hipMalloc(&in, n_in * 4); (void)hipMalloc(&out, n_out * 4);
    (void)hipMemset(in, 0, n_in * 4);
    const int n_quads = B * frames / 4;
    const double bytes = (double)B * frames * (1600 + 512);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int st = 3; st < 5; ++st)
    for (int w = 8; w <= 16; w += 8)
        for (int bpc = 2; bpc <= 4; bpc += 1) {
            for (int work = 0; work <= 0; work += 30) {
                float best = 1e9;
                for (int rep = 0; rep < 5; ++rep) {
                    (void)hipEventRecord(e0);
                    if (st == 0) { if (w == 8) probe<8, 0><<<256 * bpc, 256>>>(in, out, n_quads, work); else probe<16, 0><<<256 * bpc, 256>>>(in, out, n_quads, work); }
                    else if (st == 1) { if (w == 8) probe<8, 1><<<256 * bpc, 256>>>(in, out, n_quads, work); else probe<16, 1><<<256 * bpc, 256>>>(in, out, n_quads, work); }
                    else if (st == 3) { if (w == 8) probe<8, 3><<<256 * bpc, 256>>>(in, out, n_quads, work); else probe<16, 3><<<256 * bpc, 256>>>(in, out, n_quads, work); }
                    else if (st == 4) { if (w == 8) probe<8, 4><<<256 * bpc, 256>>>(in, out, n_quads, work); else probe<16, 4><<<256 * bpc, 256>>>(in, out, n_quads, work); }
                    else { if (w == 8) probe<8, 2><<<256 * bpc, 256>>>(in, out, n_quads, work); else probe<16, 2><<<256 * bpc, 256>>>(in, out, n_quads, work); }
                    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                    float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
                }
                printf("store mode %d, load %2d B/lane, %d blocks/CU (%2d waves/CU), alu work %2d: %.3f ms  %.0f GB/s\n", st, w, bpc, bpc * 4, work, best, (st == 2 ? (double)B * frames * 1600 : bytes) / best / 1e6);
            }
        }
    return 0;
}
