// Memory-only ceiling for the log-mel access pattern on gfx950: per frame quad 6400 contiguous bytes are read
// and 2048 contiguous bytes written (4096 clips x 96 quads: 2.517 GB in, 0.805 GB out).  No arithmetic beyond a
// sum that keeps the loads alive.  Axes:
//   loader  0  frame-shaped 8-byte loads (16 lanes own a frame: 13 loads per lane, 128-byte runs)
//           1  wave-linear 16-byte loads (64 lanes x 16 B = 1 KiB per instruction, 7 per quad)
//           2  wave-linear LDS-DMA (global_load_lds_dwordx4) into a per-wave LDS ring, read back by ds_read_b64
//           3  frame-shaped 16-byte loads (16 lanes own a frame, lane b takes samples 64a + 4b .. + 3: 7 loads per lane, 256-byte
//              runs): the pattern of a 256 = 8 x 32 split of the transform, where a lane's 16-byte load is its first-stage input
//   depth   quads in flight per wave beyond the one being consumed (1 or 2)
//   store   0  4 x 256-byte runs per instruction (the kernel's layout), nontemporal
//           1  1 KiB contiguous per instruction, nontemporal
//           2  as 0, plain stores
//           3  none (read-only)
//   blocks per CU (256 threads each), persistent grid-stride over quads; chunk = consecutive quads per wave visit
// Prints ms and TB/s of (read + written) bytes, best and median of 7 launches.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define LDS_AS __attribute__((address_space(3)))

constexpr int kQuadFloats = 1600;       // 4 frames x 400 samples
constexpr int kSlotBytes = 7 * 1024;    // LDS-DMA lands 7 x 1 KiB per quad (last piece a quarter used)

template <int ST>
__device__ __forceinline__ void store_quad(float* out, size_t q, int lane, const float (&s)[8]) {
    if (ST == 3) {
        if (s[0] == 123.456f) out[q] = s[1] + s[2] + s[3] + s[4] + s[5] + s[6] + s[7];
        return;
    }
    float* base = out + q * 512;
    if (ST == 1) {
        f32x4* o = reinterpret_cast<f32x4*>(base) + lane;
        __builtin_nontemporal_store(f32x4{s[0], s[1], s[2], s[3]}, o);
        __builtin_nontemporal_store(f32x4{s[4], s[5], s[6], s[7]}, o + 64);
    } else {
        const int q4 = lane & 3, f0 = lane & ~3;
        f32x4* o = reinterpret_cast<f32x4*>(base + q4 * 128 + f0);
        if (ST == 0) {
            __builtin_nontemporal_store(f32x4{s[0], s[1], s[2], s[3]}, o);
            __builtin_nontemporal_store(f32x4{s[4], s[5], s[6], s[7]}, o + 16);
        } else {
            o[0] = f32x4{s[0], s[1], s[2], s[3]};
            o[16] = f32x4{s[4], s[5], s[6], s[7]};
        }
    }
}

// loaders 0 and 1: register staging
template <int LOADER, int DEPTH, int ST, bool NT>
__global__ void __launch_bounds__(256) probe_reg(const float* __restrict__ in, float* __restrict__ out, int n_quads, int chunk) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NL = LOADER == 0 ? 13 : 7;
    constexpr int W = LOADER == 0 ? 2 : 4;
    static_assert(LOADER == 0 || LOADER == 1 || LOADER == 3, "register loaders");
    float buf[DEPTH][NL][W];
    // quads are visited in runs of `chunk` consecutive quads per wave
    const int wid = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    auto quad_of = [&](int i) { return ((i / chunk) * nw + wid) * chunk + (i % chunk); };
    auto load = [&](int slot, int q) {
        const float* qp = in + (size_t)q * kQuadFloats;
#pragma unroll
        for (int a = 0; a < NL; ++a) {
            if constexpr (LOADER == 0) {
                const int fl = lane >> 4, b = lane & 15;
                const int idx = 32 * a + 2 * b;
                const bool ok = idx < 400;
                const float* p = ok ? qp + fl * 400 + idx : in;
                f32x2 r = NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(p)) : *reinterpret_cast<const f32x2*>(p);
                buf[slot][a][0] = ok ? r.x : 0.f; buf[slot][a][1] = ok ? r.y : 0.f;
            } else if constexpr (LOADER == 3) {
                const int fl = lane >> 4, b = lane & 15;
                const int idx = 64 * a + 4 * b;
                const bool ok = idx < 400;
                const float* p = ok ? qp + fl * 400 + idx : in;
                f32x4 r = NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)) : *reinterpret_cast<const f32x4*>(p);
                buf[slot][a][0] = ok ? r.x : 0.f; buf[slot][a][1] = ok ? r.y : 0.f; buf[slot][a][2] = ok ? r.z : 0.f; buf[slot][a][3] = ok ? r.w : 0.f;
            } else {
                const int idx = 4 * (64 * a + lane);
                const bool ok = idx < kQuadFloats;
                const float* p = ok ? qp + idx : in;
                f32x4 r = NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)) : *reinterpret_cast<const f32x4*>(p);
                buf[slot][a][0] = ok ? r.x : 0.f; buf[slot][a][1] = ok ? r.y : 0.f; buf[slot][a][2] = ok ? r.z : 0.f; buf[slot][a][3] = ok ? r.w : 0.f;
            }
        }
    };
    int i = 0;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) { const int q = quad_of(d); if (q < n_quads) load(d, q); }
    for (;; i += DEPTH) {
        bool done = false;
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int q = quad_of(i + d);
            if (q >= n_quads) { done = true; break; }
            float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int a = 0; a < NL; ++a)
#pragma unroll
                for (int e = 0; e < W; ++e) s[(a + e) & 7] += buf[d][a][e];
            const int qn = quad_of(i + d + DEPTH);
            if (qn < n_quads) load(d, qn);
            store_quad<ST>(out, (size_t)q, lane, s);
        }
        if (done) break;
    }
}

// loader 2: LDS-DMA into a per-wave ring of DEPTH + 1 slots; the consumer reads its frame-shaped slices back
template <int DEPTH, int ST, bool NT>
__global__ void __launch_bounds__(256) probe_dma(const float* __restrict__ in, float* __restrict__ out, int n_quads, int chunk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int NS = DEPTH + 1;
    char* ring = smem + wave * NS * kSlotBytes;
    const int wid = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    auto quad_of = [&](int i) { return ((i / chunk) * nw + wid) * chunk + (i % chunk); };
    auto issue = [&](int slot, int q) {
        const char* qp = reinterpret_cast<const char*>(in + (size_t)q * kQuadFloats);
        char* dst = ring + slot * kSlotBytes;
#pragma unroll
        for (int a = 0; a < 7; ++a) {
            const int off = 16 * (64 * a + lane);
            const char* src = off < kQuadFloats * 4 ? qp + off : reinterpret_cast<const char*>(in);
            // inline asm: the builtin makes hipcc drain vmcnt(0) before every new LDS-DMA (it orders them as LDS writes)
            const unsigned lds_dst = (unsigned)(uintptr_t)(LDS_AS char*)(dst + 1024 * a);
            unsigned keep;
            if (NT) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
            else asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) { const int q = quad_of(d); if (q < n_quads) issue(d, q); }
    int slot = 0;
    for (int i = 0;; ++i) {
        const int q = quad_of(i);
        if (q >= n_quads) break;
        const int qn = quad_of(i + DEPTH);
        int nslot = slot + DEPTH; if (nslot >= NS) nslot -= NS;
        if (qn < n_quads) {
            issue(nslot, qn);
            // all but the DEPTH youngest quads (7 pieces each) have landed
            // (stores count in vmcnt too: 2 per consumed quad unless ST == 3)
            constexpr int kAllowed = DEPTH * (7 + (ST == 3 ? 0 : 2));
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kAllowed) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const float* sp = reinterpret_cast<const float*>(ring + slot * kSlotBytes);
        const int fl = lane >> 4, b = lane & 15;
        float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int a = 0; a < 13; ++a) {
            const int idx = 32 * a + 2 * b;
            const f32x2 r = *reinterpret_cast<const f32x2*>(sp + fl * 400 + (idx < 400 ? idx : 0));
            s[a & 7] += r.x; s[(a + 1) & 7] += r.y;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the slot is re-filled DEPTH iterations later at the earliest
        store_quad<ST>(out, (size_t)q, lane, s);
        slot += 1; if (slot >= NS) slot = 0;
    }
}

struct Result { float best, med; };

template <typename F>
Result time_it(F launch) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(); launch();
    std::vector<float> t;
    for (int rep = 0; rep < 7; ++rep) {
        (void)hipEventRecord(e0);
        launch();
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return {t[0], t[t.size() / 2]};
}

int main() {
    const int B = 4096, frames = 384;
    const size_t n_in = (size_t)B * frames * 400 + 4096, n_out = (size_t)B * frames * 128;
    float *in, *out;
    if (hipMalloc(&in, n_in * 4) != hipSuccess || hipMalloc(&out, n_out * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    // random-ish contents (zero pages could be special-cased by the memory system)
    {
        std::vector<float> h(1 << 22);
        unsigned x = 12345u;
        for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (float)(x >> 8) * (1.0f / 16777216.0f) - 0.5f; }
        for (size_t off = 0; off < n_in; off += h.size())
            (void)hipMemcpy(in + off, h.data(), std::min(h.size(), n_in - off) * 4, hipMemcpyHostToDevice);
    }
    const int n_quads = B * frames / 4;
    const double rd = (double)B * frames * 1600, wr = (double)B * frames * 512;
    auto report = [&](const char* name, int depth, int st, bool nt, int bpc, int chunk, Result r) {
        const double bytes = rd + (st == 3 ? 0.0 : wr);
        printf("%-10s depth %d store %d nt %d  %2d blocks/CU chunk %2d : best %.3f ms  med %.3f ms  %.2f TB/s\n", name, depth, st, (int)nt, bpc, chunk,
               r.best, r.med, bytes / r.med / 1e9);
        fflush(stdout);
    };
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_dma<1, 0, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_dma<1, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_dma<2, 0, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_dma<2, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_dma<2, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_dma<2, 3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_dma<1, 3, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);

#define RUN_REG(L, D, ST, NT, bpc, chunk) report(L == 0 ? "frame8B" : (L == 3 ? "frame16B" : "linear16B"), D, ST, NT, bpc, chunk, time_it([&] { probe_reg<L, D, ST, NT><<<256 * (bpc), 256>>>(in, out, n_quads, chunk); }))
#define RUN_DMA(D, ST, NT, bpc, chunk) report("lds-dma", D, ST, NT, bpc, chunk, time_it([&] { probe_dma<D, ST, NT><<<256 * (bpc), 256, 4 * (D + 1) * kSlotBytes>>>(in, out, n_quads, chunk); }))

    // the three register patterns side by side at the kernel's occupancy (3 blocks of 4 waves per CU) and one above it
    for (int rep = 0; rep < 2; ++rep)
        for (int bpc : {3, 4}) {
            RUN_REG(0, 1, 0, false, bpc, 1);
            RUN_REG(0, 1, 0, true, bpc, 1);
            RUN_REG(3, 1, 0, false, bpc, 1);
            RUN_REG(3, 1, 0, true, bpc, 1);
            RUN_REG(1, 1, 0, false, bpc, 1);
            RUN_REG(1, 1, 0, true, bpc, 1);
            RUN_REG(0, 1, 3, true, bpc, 1);
            RUN_REG(3, 1, 3, true, bpc, 1);
        }
    if (getenv("MMFE_PROBE_SHORT")) { (void)hipDeviceSynchronize(); return 0; }
    for (int bpc : {2, 3, 4, 6, 8}) {
        RUN_REG(0, 1, 0, false, bpc, 1);
        RUN_REG(1, 1, 0, false, bpc, 1);
        RUN_REG(1, 1, 0, true, bpc, 1);
        RUN_REG(1, 2, 0, false, bpc, 1);
        RUN_REG(1, 2, 0, true, bpc, 1);
        RUN_REG(1, 2, 1, true, bpc, 1);
        RUN_REG(1, 2, 2, false, bpc, 1);
        RUN_REG(1, 2, 3, false, bpc, 1);
    }
    for (int chunk : {4, 24, 96}) {
        RUN_REG(1, 2, 0, true, 3, chunk);
        RUN_REG(1, 2, 0, true, 4, chunk);
        RUN_REG(1, 2, 0, false, 4, chunk);
    }
    // LDS-DMA: 4 waves x (depth + 1) x 7 KiB per block
    for (int bpc : {2, 3}) {
        RUN_DMA(1, 0, false, bpc, 1);
        RUN_DMA(1, 0, true, bpc, 1);
        RUN_DMA(1, 3, false, bpc, 1);
    }
    for (int bpc : {1, 2}) {      // depth 2: 84 KiB per block
        RUN_DMA(2, 0, false, bpc, 1);
        RUN_DMA(2, 0, true, bpc, 1);
        RUN_DMA(2, 1, true, bpc, 1);
        RUN_DMA(2, 3, true, bpc, 1);
    }
    (void)hipDeviceSynchronize();
    printf("last error: %s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
