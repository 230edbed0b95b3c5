// Patch embedding of a spectrogram: the 16 x 16 / stride-16 convolution with which the encoders turn the front-end's
// output into tokens (PatchEmbed.forward, classification/vggsound/models/vit_model.py:193-199, with the projection
// AudioTransformer installs at vit_model.py:435):
//     tokens[b][ph * Pw + pw][o] = bias[o] + sum_{i,j} weight[o][0][i][j] * x[b][16 ph + i][16 pw + j]
// i.e. one (B * P) x 256 by 256 x E matrix product, run on the matrix pipe in the 16-bit type the encoders' @autocast()
// (vit_model.py:388) computes the convolution in, with float32 accumulation.
//
// Shape of the work on an MI355X.  K = 256 is tiny and E <= 768, so the WHOLE weight matrix (E x 256 x 2 bytes <= 384 KB) fits
// the register file of one CU (512 KB): a block of 8 waves keeps it there for the life of the kernel, wave w holding the
// MFMA operands of output columns [w E/8, (w + 1) E/8).  Blocks are persistent; per step a block takes 64 consecutive patches.
// Their 64 x 256 input values arrive by LDS-DMA (one global_load_lds_dwordx4 per patch: the 16 image-row segments of 64 bytes
// that make up a patch land as one 1 KB row of LDS; rows are 1040 bytes apart, which puts the 16 patches a wave reads together
// on 16 different bank groups) into one of two buffers, a step ahead of the arithmetic.  Every wave reads all 64 patches
// (float32, converted to the 16-bit type on the way into the matrix pipe) and multiplies them by its own columns:
//     D[16 columns x 16 patches] += Wfrag[16 columns x 32 k] * patches[32 k x 16 patches]        (v_mfma_f32_16x16x32)
// so a lane ends with 4 consecutive output columns of one patch per 16-column tile and stores 16 bytes; the 4 lanes that
// share a patch write 64 contiguous bytes.  Traffic: the image is read once, the tokens are written once, the weights are read
// once per block (from L2).  The kernel is bound by its token stores (3 x the input bytes for float32 tokens), the matrix pipe
// is about a quarter busy.
//
// k order.  The MFMA sums over 32 k values per instruction, 8 per lane quarter h = lane >> 4.  Which k they are is free as
// long as both operands agree: lane quarter h takes k = 64 h + 8 s + j in step s (j = 0..7), so that a lane's 8 values are 32
// contiguous bytes of a patch row and the two quarters that share an LDS access group read at offsets 256 bytes apart (the same
// bank group, which the 16 different rows then spread).
#include "mmfe_internal.h"

#include <cmath>
#include <cstring>
#include <new>

namespace mmfe {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kPatch = 16;                 // patch side (kernel size and stride of the convolution)
constexpr int kK = kPatch * kPatch;        // 256 values per patch
constexpr int kWaves = 8;
constexpr int kThreads = 64 * kWaves;
constexpr int kTileM = 64;                 // patches per block step
constexpr int kRowBytes = kK * 4 + 16;     // LDS row of one patch: 1 KB of float32 + 16 bytes (bank spreading)
constexpr int kBufBytes = kTileM * kRowBytes;
// 16-bit images: a patch is 512 bytes, one DMA brings a PAIR of patches with their 16-byte pieces interleaved (piece c of the even
// patch at 32 c, of the odd patch at 32 c + 16); pairs are 1056 bytes apart, which puts patch m's piece on bank group (m + 2 s) mod 16
constexpr int kPairBytes = 2 * kK * 2 + 32;
constexpr int kBuf16Bytes = (kTileM / 2) * kPairBytes;
constexpr int kBiasBytes = 4096;           // E <= 768 floats, padded

struct PatchArgs {
    const void* x;             // (B, H, W) float32, or the 16-bit type of the products
    int64_t x_batch_stride;    // elements
    const u32x4* wpack;        // MFMA operand order: [(wave * NT + tile)][step][lane] 16 bytes (8 values of the 16-bit type)
    const float* bias;         // E floats (zeros when the layer has no bias)
    void* out;                 // (B, P, E)
    int64_t out_batch_stride;  // elements
    int W, Pw, P, E;
    int64_t M;                 // B * P patches
    int n_tiles;
};

// One LDS-DMA: every lane's 16 bytes at `gsrc` go to LDS byte address lds_dst + 16 * lane.  M0 holds the destination base for
// the instruction; it is the compiler's register, so it is saved and restored inside the statement.  Not counted by the
// compiler's s_waitcnt bookkeeping: completion is waited for by hand (vmcnt, see the step loop).
#ifndef MMFE_PE_NTLOAD
#define MMFE_PE_NTLOAD 0      // image loads bypass the caches (tuning switch)
#endif
#if MMFE_PE_NTLOAD
#define MMFE_PE_LD_HINT " nt"
#else
#define MMFE_PE_LD_HINT ""
#endif
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
#ifdef MMFE_PE_NOLOAD         // ablation build: no image traffic (the arithmetic runs on whatever the buffers hold)
    asm volatile("" : "=s"(keep) : "v"(gsrc), "s"(lds_dst));
#else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" MMFE_PE_LD_HINT "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
#endif
}

// Token stores are inline asm so that their number per step is exactly what the step loop's vmcnt count assumes (the compiler
// could split or merge plain stores).  The trailing s_nop keeps the data registers intact until the store has read them.
#ifndef MMFE_PE_NT
#define MMFE_PE_NT 0          // 1: token stores bypass the caches.  Measured slower (1.01 vs 0.66 ms per 4096 clips, float32 tokens): a store instruction
                              // writes 64-byte halves of 16 lines, and the other halves come from the next instruction; cached, they merge in L2
#endif
#if MMFE_PE_NT
#define MMFE_PE_ST_HINT " nt"
#else
#define MMFE_PE_ST_HINT ""
#endif
__device__ __forceinline__ void store16(void* p, u32x4 v) {
#ifdef MMFE_PE_NOSTORE        // ablation build: what the kernel takes without its token stores (results are kept alive, not written)
    asm volatile("" :: "v"(p), "v"(v));
#else
    asm volatile("global_store_dwordx4 %0, %1, off" MMFE_PE_ST_HINT "\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
#endif
}
__device__ __forceinline__ void store8(void* p, u32x2 v) {
#ifdef MMFE_PE_NOSTORE
    asm volatile("" :: "v"(p), "v"(v));
#else
    asm volatile("global_store_dwordx2 %0, %1, off" MMFE_PE_ST_HINT "\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
#endif
}

template <bool BF>
__device__ __forceinline__ u32x4 to16(f32x8 v) {
    if constexpr (BF) return __builtin_bit_cast(u32x4, __builtin_convertvector(v, bf16x8));
    else return __builtin_bit_cast(u32x4, __builtin_convertvector(v, f16x8));
}

template <bool BF>
__device__ __forceinline__ f32x4 mfma(u32x4 a, u32x4 b, f32x4 c) {
    if constexpr (BF) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// Number of token-store instructions a wave issues per step (4 patch groups of 16): what the vmcnt wait skips over.
template <int NT, int OUT>
constexpr int stores_per_step() {
#ifdef MMFE_PE_NOSTORE
    return 0;
#else
    return 4 * (OUT == MMFE_OUT_F32 ? NT : (NT % 2 == 0 ? NT / 2 : NT));
#endif
}

// NT: 16-column tiles per wave (E = 128 NT).  BF: bfloat16 (else float16) operands.  OUT: MMFE_OUT_* of the tokens.
// IN16: the image already holds the 16-bit type of the products (what the front-end kernels store with out_dtype float16 / bfloat16).
template <int NT, bool BF, int OUT, bool IN16>
__global__ void __launch_bounds__(kThreads) patch_embed_kernel(const PatchArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // the one LDS object: [bias 4 KB][buffer 0][buffer 1]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lm = lane & 15, lh = lane >> 4;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

    // this wave's weights: NT tiles x 8 steps x 16 bytes per lane, resident in registers from here on
    u32x4 wf[NT][8];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int s = 0; s < 8; ++s) wf[t][s] = p.wpack[((size_t)(wave * NT + t) * 8 + s) * 64 + lane];
    for (int i = threadIdx.x; i < kBiasBytes / 4; i += kThreads) reinterpret_cast<float*>(smem)[i] = i < p.E ? p.bias[i] : 0.f;

    // lane part of a patch's source address.  float32: piece `lane` of the 64 sixteen-byte pieces is image row lane >> 2, columns
    // 4 (lane & 3)..+3.  16-bit: lane = 2 c + e fetches piece c (image row c >> 1, columns 8 (c & 1)..+7) of the pair's patch e.
    const int64_t lane_off = IN16 ? (int64_t)(lane >> 2) * p.W + 8 * ((lane >> 1) & 1) : (int64_t)(lane >> 2) * p.W + 4 * (lane & 3);
    const int Ph = p.P / p.Pw;
    constexpr int kBuf = IN16 ? kBuf16Bytes : kBufBytes;
    auto stage = [&](int tile, int buf) {           // request the 64 patches of step `tile` into buffer `buf`: 8 consecutive ones per wave
        unsigned m = (unsigned)tile * kTileM + wave * (kTileM / kWaves);
        if ((int64_t)m > p.M - 1) m = (unsigned)(p.M - 1);           // a last, partial step: waves beyond the batch re-read its last patch
        unsigned b = m / (unsigned)p.P;
        unsigned pp = m - b * (unsigned)p.P;
        int ph = (int)(pp / (unsigned)p.Pw), pw = (int)(pp - (pp / (unsigned)p.Pw) * (unsigned)p.Pw);
        auto patch_base = [&]() { return (int64_t)b * p.x_batch_stride + (int64_t)(ph * kPatch) * p.W + pw * kPatch; };
        auto advance = [&](int r) {                                  // to the next patch (a last, partial step stays on the last one)
            if ((int64_t)m + r + 1 < p.M) {
                if (++pw == p.Pw) { pw = 0; if (++ph == Ph) { ph = 0; ++b; } }
            }
        };
        if constexpr (!IN16) {
#pragma unroll
            for (int r = 0; r < kTileM / kWaves; ++r) {
                const int ml = wave * (kTileM / kWaves) + r;
                const float* src = static_cast<const float*>(p.x) + patch_base() + lane_off;
                dma16(src, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + kBiasBytes + buf * kBuf + ml * kRowBytes)));
                advance(r);
            }
        } else {
#pragma unroll
            for (int r = 0; r < kTileM / kWaves; r += 2) {
                const int pair = (wave * (kTileM / kWaves) + r) >> 1;
                const int64_t base0 = patch_base();
                advance(r);
                const int64_t base1 = patch_base();
                advance(r + 1);
                const unsigned short* src = static_cast<const unsigned short*>(p.x) + ((lane & 1) ? base1 : base0) + lane_off;
                dma16(src, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + kBiasBytes + buf * kBuf + pair * kPairBytes)));
            }
        }
    };

    int tile = blockIdx.x;
    if (tile >= p.n_tiles) return;
    stage(tile, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // first tile, weights and the bias image: all there
    // The weights are complete; saying so here keeps the compiler from placing its own wait for them (a vmcnt(0), which would
    // also drain the DMAs of the step ahead) at their first use inside the loop.
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int s = 0; s < 8; ++s) asm volatile("" : "+v"(wf[t][s]));
    int buf = 0;
    const int col0 = wave * (16 * NT);               // this wave's first output column
    for (;;) {
        // Every wave has waited for its own DMAs of this step, and has finished reading the other buffer: after the barrier the
        // whole tile is in LDS and the other buffer may be overwritten.
        __builtin_amdgcn_s_barrier();
        const int next = tile + gridDim.x;
        if (next < p.n_tiles) stage(next, buf ^ 1);

        const char* tile_base = smem + kBiasBytes + buf * kBuf;
#pragma unroll 1
        for (int mt = 0; mt < 4; ++mt) {
            f32x4 acc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {            // start from the bias of the 4 columns this lane ends with (see the column maps below)
                int c;
                if constexpr (OUT == MMFE_OUT_F32 || NT % 2 != 0) c = col0 + 16 * t + 4 * lh;
                else c = col0 + 32 * (t >> 1) + 8 * lh + 4 * (t & 1);
                acc[t] = *reinterpret_cast<const f32x4*>(smem + 4 * c);
            }
            if constexpr (IN16) {
                const char* row = tile_base + ((mt * 16 + lm) >> 1) * kPairBytes + (lm & 1) * 16 + lh * 256;
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const u32x4 b = *reinterpret_cast<const u32x4*>(row + 32 * s);
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = mfma<BF>(wf[t][s], b, acc[t]);
                }
            } else {
                const char* row = tile_base + (mt * 16 + lm) * kRowBytes + lh * 256;
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(row + 32 * s);
                    const f32x4 hi = *reinterpret_cast<const f32x4*>(row + 32 * s + 16);
                    const u32x4 b = to16<BF>(f32x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w});
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = mfma<BF>(wf[t][s], b, acc[t]);
                }
            }
            // epilogue: lane (lm, lh) holds, per tile t, 4 output columns of patch mt * 16 + lm.  The float32 stores read the
            // accumulators from inside asm statements, where the compiler pads no hazards: the wait states a vector-memory read of
            // a matrix-pipe result needs are spent here, once per accumulator.
            if constexpr (OUT == MMFE_OUT_F32) {
#pragma unroll
                for (int t = 0; t < NT; ++t) asm volatile("s_nop 7\n\ts_nop 4" : "+v"(acc[t]));
            }
            const unsigned m = (unsigned)tile * kTileM + mt * 16 + lm;
            if ((int64_t)m < p.M) {
                const int64_t b = m / (unsigned)p.P;
                const int64_t pp = m - (unsigned)b * (unsigned)p.P;
                if constexpr (OUT == MMFE_OUT_F32) {
                    // column map: tile t, row 4 lh + r  ->  column col0 + 16 t + 4 lh + r
                    float* o = static_cast<float*>(p.out) + b * p.out_batch_stride + pp * p.E + col0 + 4 * lh;
#pragma unroll
                    for (int t = 0; t < NT; ++t) store16(o + 16 * t, __builtin_bit_cast(u32x4, acc[t]));
                } else if constexpr (NT % 2 == 0) {
                    // column map: tiles 2u, 2u + 1, row 4 lh + r  ->  column col0 + 32 u + 8 lh + 4 (t & 1) + r: 8 consecutive columns per lane
                    unsigned short* o = static_cast<unsigned short*>(p.out) + b * p.out_batch_stride + pp * p.E + col0 + 8 * lh;
#pragma unroll
                    for (int u = 0; u < NT / 2; ++u) {
                        const f32x4 a0 = acc[2 * u], a1 = acc[2 * u + 1];
                        store16(o + 32 * u, u32x4{pack2_bits(a0.x, a0.y, OUT), pack2_bits(a0.z, a0.w, OUT), pack2_bits(a1.x, a1.y, OUT), pack2_bits(a1.z, a1.w, OUT)});
                    }
                } else {
                    unsigned short* o = static_cast<unsigned short*>(p.out) + b * p.out_batch_stride + pp * p.E + col0 + 4 * lh;
#pragma unroll
                    for (int t = 0; t < NT; ++t) store8(o + 16 * t, u32x2{pack2_bits(acc[t].x, acc[t].y, OUT), pack2_bits(acc[t].z, acc[t].w, OUT)});
                }
            }
        }
        if (next >= p.n_tiles) break;
        tile = next;
        buf ^= 1;
        // This wave's DMAs of the next step were issued before this step's token stores: they are complete once all but the
        // stores_per_step youngest vector-memory operations are (only a full step can be followed by another one).
        if constexpr (stores_per_step<NT, OUT>() == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else if constexpr (stores_per_step<NT, OUT>() == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if constexpr (stores_per_step<NT, OUT>() == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if constexpr (stores_per_step<NT, OUT>() == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if constexpr (stores_per_step<NT, OUT>() == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

// Re-packs float32 weights that live on the device (the parameters of a model in training) into operand order, and rounds the
// bias: one thread per weight.  Index arithmetic as in mmfe_patch_embed_create.
__global__ void __launch_bounds__(256) patch_pack_kernel(const float* __restrict__ weight, const float* __restrict__ bias,
                                                         unsigned short* __restrict__ pack, float* __restrict__ bias_out, int E, int NT, int bf, int paired) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < E * kK) {
        const int j = i & 7, l = (i >> 3) & 63, s = (i >> 9) & 7, wt = i >> 12;      // [(wave * NT + t)][s][lane][j]
        const int w = wt / NT, t = wt - w * NT, rho = l & 15, h = l >> 4;
        const int col = w * 16 * NT + (paired ? 32 * (t >> 1) + 8 * (rho >> 2) + 4 * (t & 1) + (rho & 3) : 16 * t + rho);
        const float v = weight[(size_t)col * kK + 64 * h + 8 * s + j];
        pack[i] = bf ? bf16_bits(v) : f16_bits(v);
    }
    if (i < E) {
        const float v = bias != nullptr ? bias[i] : 0.f;
        bias_out[i] = bf ? __uint_as_float((unsigned)bf16_bits(v) << 16) : (float)static_cast<_Float16>(v);
    }
}

unsigned short host_bf16(float v) {
    unsigned u;
    std::memcpy(&u, &v, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
unsigned short host_f16(float v) {
    const _Float16 h = static_cast<_Float16>(v);       // round to nearest even
    unsigned short b;
    std::memcpy(&b, &h, 2);
    return b;
}

typedef void (*kernel_fn)(const PatchArgs);
template <int NT, bool BF, bool IN16>
kernel_fn pick_out(int out) {
    return out == MMFE_OUT_F32 ? &patch_embed_kernel<NT, BF, MMFE_OUT_F32, IN16> : out == MMFE_OUT_F16 ? &patch_embed_kernel<NT, BF, MMFE_OUT_F16, IN16>
                                                                                                       : &patch_embed_kernel<NT, BF, MMFE_OUT_BF16, IN16>;
}
template <int NT>
kernel_fn pick_kernel(bool bf, int out, bool in16) {
    if (bf) return in16 ? pick_out<NT, true, true>(out) : pick_out<NT, true, false>(out);
    return in16 ? pick_out<NT, false, true>(out) : pick_out<NT, false, false>(out);
}

}  // namespace

struct PatchEmbed {
    int device = -1;
    int height = 0, width = 0, embed = 0, nt = 0;
    int compute = MMFE_OUT_BF16, out_dtype = MMFE_OUT_F32;
    bool in16 = false;
    u32x4* d_wpack = nullptr;
    float* d_bias = nullptr;
    kernel_fn fn = nullptr;
};

}  // namespace mmfe

using namespace mmfe;

extern "C" {

int mmfe_patch_embed_create(int32_t height, int32_t width, int32_t patch, int32_t embed_dim, const float* weight, const float* bias,
                            int32_t in_dtype, int32_t compute_dtype, int32_t out_dtype, int device, mmfe_patch_embed** out_handle) {
    if (out_handle == nullptr) { set_error("out_handle is NULL"); return MMFE_EINVAL; }
    *out_handle = nullptr;
    if (weight == nullptr) { set_error("weight is NULL"); return MMFE_EINVAL; }
    if (patch != kPatch) { set_error("patch_embed: the kernel is built for 16 x 16 patches with stride 16 (vit_model.py:435), got %d", patch); return MMFE_EINVAL; }
    if (height < kPatch || width < kPatch) { set_error("patch_embed: image %d x %d smaller than one patch", height, width); return MMFE_EINVAL; }
    if (in_dtype != MMFE_IN_F32 && in_dtype != MMFE_IN_F16 && in_dtype != MMFE_IN_BF16) { set_error("patch_embed: in_dtype is MMFE_IN_F32, MMFE_IN_F16 or MMFE_IN_BF16"); return MMFE_EINVAL; }
    const bool in16 = in_dtype != MMFE_IN_F32;
    if (in16 && (in_dtype == MMFE_IN_F16) != (compute_dtype == MMFE_OUT_F16)) { set_error("patch_embed: a 16-bit image must hold the type of the products (compute_dtype)"); return MMFE_EINVAL; }
    if (width % (in16 ? 8 : 4) != 0) { set_error("patch_embed: image width must be a multiple of %d (16-byte pieces), got %d", in16 ? 8 : 4, width); return MMFE_EINVAL; }
    if (embed_dim != 384 && embed_dim != 768) { set_error("patch_embed: embed_dim 384 or 768 (8 waves x 3 or 6 tiles of 16 columns), got %d", embed_dim); return MMFE_EINVAL; }
    if (compute_dtype != MMFE_OUT_F16 && compute_dtype != MMFE_OUT_BF16) { set_error("patch_embed: compute_dtype is MMFE_OUT_F16 or MMFE_OUT_BF16"); return MMFE_EINVAL; }
    if (out_dtype < MMFE_OUT_F32 || out_dtype > MMFE_OUT_BF16) { set_error("patch_embed: unknown out_dtype"); return MMFE_EINVAL; }
    const int ndev = mmfe_device_count();
    if (ndev <= 0) { set_error("no HIP device available: this library has no CPU path"); return MMFE_ENODEVICE; }
    if (device < 0 || device >= ndev) { set_error("device %d out of range (%d devices)", device, ndev); return MMFE_ENODEVICE; }

    PatchEmbed* pe = new (std::nothrow) PatchEmbed();
    if (pe == nullptr) { set_error("out of host memory"); return MMFE_ENOMEM; }
    pe->device = device; pe->height = height; pe->width = width; pe->embed = embed_dim; pe->nt = embed_dim / 128;
    pe->compute = compute_dtype; pe->out_dtype = out_dtype; pe->in16 = in16;
    const int NT = pe->nt;
    const bool bf = compute_dtype == MMFE_OUT_BF16;
    const bool paired = out_dtype != MMFE_OUT_F32 && NT % 2 == 0;
    // operand order: [(wave * NT + t)][s][lane][j] = weight[column(wave, t, lane & 15)][64 (lane >> 4) + 8 s + j]
    std::vector<unsigned short> pack((size_t)embed_dim * kK);
    for (int w = 0; w < kWaves; ++w)
        for (int t = 0; t < NT; ++t)
            for (int s = 0; s < 8; ++s)
                for (int l = 0; l < 64; ++l) {
                    const int rho = l & 15, h = l >> 4;
                    const int col = w * 16 * NT + (paired ? 32 * (t >> 1) + 8 * (rho >> 2) + 4 * (t & 1) + (rho & 3) : 16 * t + rho);
                    for (int j = 0; j < 8; ++j) {
                        const float v = weight[(size_t)col * kK + 64 * h + 8 * s + j];
                        pack[((((size_t)(w * NT + t) * 8 + s) * 64) + l) * 8 + j] = bf ? host_bf16(v) : host_f16(v);
                    }
                }
    std::vector<float> hb(embed_dim, 0.f);
    // autocast casts every floating-point argument of the convolution, the bias too: it enters the float32 sum as its 16-bit value
    if (bias != nullptr)
        for (int i = 0; i < embed_dim; ++i) {
            if (bf) { const unsigned u = (unsigned)host_bf16(bias[i]) << 16; std::memcpy(&hb[i], &u, 4); }
            else hb[i] = (float)static_cast<_Float16>(bias[i]);
        }
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess || (prev != device && hipSetDevice(device) != hipSuccess)) { set_error("cannot make device %d current", device); delete pe; return MMFE_EHIP; }
    int rc = MMFE_OK;
    if (hipMalloc((void**)&pe->d_wpack, pack.size() * 2) != hipSuccess || hipMalloc((void**)&pe->d_bias, hb.size() * 4) != hipSuccess ||
        hipMemcpy(pe->d_wpack, pack.data(), pack.size() * 2, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(pe->d_bias, hb.data(), hb.size() * 4, hipMemcpyHostToDevice) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        set_error("patch_embed: uploading the weights failed");
        rc = MMFE_EHIP;
    }
    if (rc == MMFE_OK) pe->fn = NT == 6 ? pick_kernel<6>(bf, out_dtype, in16) : pick_kernel<3>(bf, out_dtype, in16);
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    if (rc) { (void)hipFree(pe->d_wpack); (void)hipFree(pe->d_bias); delete pe; return rc; }
    *out_handle = reinterpret_cast<mmfe_patch_embed*>(pe);
    return MMFE_OK;
}

int mmfe_patch_embed_update(mmfe_patch_embed* handle, const float* weight_dev, const float* bias_dev, void* hip_stream) {
    PatchEmbed* pe = reinterpret_cast<PatchEmbed*>(handle);
    if (pe == nullptr || weight_dev == nullptr) { set_error("patch_embed_update: NULL handle or weight"); return MMFE_EINVAL; }
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess || (prev != pe->device && hipSetDevice(pe->device) != hipSuccess)) { set_error("cannot make device %d current", pe->device); return MMFE_EHIP; }
    const int n = pe->embed * kK;
    const int paired = pe->out_dtype != MMFE_OUT_F32 && pe->nt % 2 == 0;
    hipLaunchKernelGGL(patch_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(hip_stream), weight_dev, bias_dev,
                       reinterpret_cast<unsigned short*>(pe->d_wpack), pe->d_bias, pe->embed, pe->nt, pe->compute == MMFE_OUT_BF16 ? 1 : 0, paired);
    const hipError_t e = hipGetLastError();
    if (prev >= 0 && prev != pe->device) (void)hipSetDevice(prev);
    if (e != hipSuccess) { set_error("patch_embed_update launch failed: %s", hipGetErrorString(e)); return MMFE_EHIP; }
    return MMFE_OK;
}

int mmfe_patch_embed_destroy(mmfe_patch_embed* handle) {
    PatchEmbed* pe = reinterpret_cast<PatchEmbed*>(handle);
    if (pe == nullptr) return MMFE_OK;
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (prev != pe->device) (void)hipSetDevice(pe->device);
    (void)hipFree(pe->d_wpack); (void)hipFree(pe->d_bias);
    if (prev >= 0 && prev != pe->device) (void)hipSetDevice(prev);
    delete pe;
    return MMFE_OK;
}

int mmfe_patch_embed_shape(const mmfe_patch_embed* handle, int64_t* n_patches, int64_t* embed_dim) {
    const PatchEmbed* pe = reinterpret_cast<const PatchEmbed*>(handle);
    if (pe == nullptr || n_patches == nullptr || embed_dim == nullptr) { set_error("bad argument"); return MMFE_EINVAL; }
    *n_patches = (int64_t)((pe->height - kPatch) / kPatch + 1) * ((pe->width - kPatch) / kPatch + 1);
    *embed_dim = pe->embed;
    return MMFE_OK;
}

int mmfe_patch_embed_run(const mmfe_patch_embed* handle, const void* x, int64_t x_batch_stride, int64_t n_batch, void* out,
                         int64_t out_batch_stride, void* hip_stream) {
    const PatchEmbed* pe = reinterpret_cast<const PatchEmbed*>(handle);
    if (pe == nullptr) { set_error("handle is NULL"); return MMFE_EINVAL; }
    if (n_batch < 0) { set_error("negative batch"); return MMFE_EINVAL; }
    if (n_batch == 0) return MMFE_OK;
    if (x == nullptr || out == nullptr) { set_error("NULL data pointer"); return MMFE_EINVAL; }
    const int Ph = (pe->height - kPatch) / kPatch + 1, Pw = (pe->width - kPatch) / kPatch + 1;
    const int64_t P = (int64_t)Ph * Pw;
    if (x_batch_stride < (int64_t)pe->height * pe->width && n_batch > 1) { set_error("x_batch_stride smaller than one image"); return MMFE_EINVAL; }
    if (out_batch_stride < P * pe->embed && n_batch > 1) { set_error("out_batch_stride smaller than one image's tokens"); return MMFE_EINVAL; }
    if ((reinterpret_cast<uintptr_t>(x) & 15) != 0 || (x_batch_stride & (pe->in16 ? 7 : 3)) != 0) { set_error("patch_embed: images must start on 16-byte boundaries"); return MMFE_EINVAL; }
    const int esz = pe->out_dtype == MMFE_OUT_F32 ? 4 : 2;
    if ((reinterpret_cast<uintptr_t>(out) & 15) != 0 || (out_batch_stride * esz) % 16 != 0) { set_error("patch_embed: token rows must start on 16-byte boundaries"); return MMFE_EINVAL; }
    const int64_t M = n_batch * P;
    if (M > 0x7fffff00ll) { set_error("patch_embed: batch too large"); return MMFE_EINVAL; }
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess || (prev != pe->device && hipSetDevice(pe->device) != hipSuccess)) { set_error("cannot make device %d current", pe->device); return MMFE_EHIP; }
    const size_t lds = kBiasBytes + 2 * (pe->in16 ? kBuf16Bytes : kBufBytes);
    int blocks_per_cu = 1, cus = 256;
    int rc = kernel_prepare(reinterpret_cast<const void*>(pe->fn), kThreads, lds, &blocks_per_cu, &cus);
    if (rc == MMFE_OK) {
        PatchArgs a{};
        a.x = x; a.x_batch_stride = x_batch_stride; a.wpack = pe->d_wpack; a.bias = pe->d_bias; a.out = out; a.out_batch_stride = out_batch_stride;
        a.W = pe->width; a.Pw = Pw; a.P = (int)P; a.E = pe->embed; a.M = M; a.n_tiles = (int)((M + kTileM - 1) / kTileM);
        const int grid = a.n_tiles < cus ? a.n_tiles : cus;
        hipLaunchKernelGGL(pe->fn, dim3(grid), dim3(kThreads), lds, reinterpret_cast<hipStream_t>(hip_stream), a);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) { set_error("patch_embed launch failed: %s", hipGetErrorString(e)); rc = MMFE_EHIP; }
    }
    if (prev >= 0 && prev != pe->device) (void)hipSetDevice(prev);
    return rc;
}

}  // extern "C"
