// Rational-ratio polyphase FIR resampler on gfx950 (MMFE_KIND_RESAMPLE): the step the reference runs offline, on the CPU,
// in front of the spectrogram path: librosa.load(file, sr=16000, duration=10) turns the 44.1 / 48 kHz VGGSound tracks into
// the 16 kHz clips the dataset stores (classification/vggsound/dataset/vggsound.py:114), and ffmpeg -ar 24000 does the same
// for EPIC (classification/recognition/dataset/extract_audio.py:6-13).
//
// out[m] = sum_j bank[phase(m)][j] * x[i(m) - j],   t = (m + pre) * down,  i(m) = t / up,  phase(m) = t % up
// with bank[p][j] = h_padded[p + j * up]: upsample by `up`, FIR-filter with h, keep every `down`-th sample, with the
// alignment and output count of scipy.signal.resample_poly (h centred on the output grid, ceil(n * up / down) outputs,
// zeros outside the row).
//
// Two kernels.
//
// resample_phase_kernel (ratios whose phase pattern lines up with the wave, e.g. 160/441, 1/3, 1/2): outputs m and m + up
// use the same filter phase and sit exactly `down` input samples apart.  So the 64 lanes of a wave take outputs that are a
// multiple L of `up` apart: every lane then needs the SAME tap weights, which makes them wave-uniform values that arrive
// through the scalar cache and ride on the multiply-adds as scalar operands - no LDS or vector-memory traffic for the
// filter at all.  A lane produces Q = 4 consecutive outputs; their input windows overlap, so it walks ONE window of
// taps + (Q - 1) * down / up samples, reads each sample once from LDS (lane stride L * down / up: odd for 160/441, i.e.
// conflict-free) and feeds it to the Q accumulators.  Per output that is (taps + 8) / 4 LDS reads instead of taps.  A block
// stages the input of a tile of NB * 64 * L outputs in LDS (fetched one tile ahead into registers, 16-bit PCM decoded
// there), collects the tile's outputs in a second LDS area (a lane's outputs are L floats apart) and stores them as whole
// contiguous rows.
//
// resample_kernel (any other ratio): a block takes a tile of consecutive outputs of one row, input samples and the
// polyphase bank in LDS (bank rows padded to whole 16-byte groups, four taps per read), one thread per output.
#include "mmfe_internal.h"
#include "fft_regs.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <type_traits>
#include <vector>

namespace mmfe {

namespace {

constexpr int kThreads = 256;
constexpr int kMaxTileIn = 12288;        // floats of staged input per tile (48 KiB)
constexpr int kMaxBankBytes = 96 * 1024;

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ResampleArgs {
    RowSource src;
    float* out;
    int n_rows;
    int n_in;                  // samples per row (ragged batches: of the longest row)
    int64_t n_out;             // outputs computed per row
    int64_t out_stride;        // floats between output rows
    int up, down;
    int taps;                  // J: taps per phase
    int taps_padded;           // J rounded up to a multiple of 4
    int pre;                   // outputs of the full filter response dropped in front (n_pre_remove)
    int tile_out;              // outputs per tile
    int tile_in;               // staged input samples per tile
    int tiles_per_row;
    const float* bank;         // [up][taps_padded]
};

template <typename SMP, bool BANK_IN_LDS>
__global__ void __launch_bounds__(kThreads) resample_kernel(const ResampleArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_bank = reinterpret_cast<float*>(smem);
    float* s_x = s_bank + (BANK_IN_LDS ? p.up * p.taps_padded : 0);
    if (BANK_IN_LDS) {
        const int n4 = p.up * p.taps_padded / 4;
        for (int i = threadIdx.x; i < n4; i += kThreads) reinterpret_cast<f32x4*>(s_bank)[i] = reinterpret_cast<const f32x4*>(p.bank)[i];
    }
    const float* bank = BANK_IN_LDS ? s_bank : p.bank;
    const int64_t n_tiles = (int64_t)p.n_rows * p.tiles_per_row;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int row = (int)(tile / p.tiles_per_row);
        const int64_t m0 = (tile - (int64_t)row * p.tiles_per_row) * p.tile_out;
        const SMP* x = row_ptr_as<SMP>(p.src, row);
        const int n_row = row_samples(p.src, row, p.n_in);
        // staged range: from the oldest sample the tile's first output reads
        const int64_t i_lo = ((m0 + p.pre) * p.down) / p.up - (p.taps_padded - 1);
        __syncthreads();                                   // the previous tile's readers are done (and the bank is in place)
        for (int i = threadIdx.x; i < p.tile_in; i += kThreads) {
            const int64_t g = i_lo + i;
            s_x[i] = (g >= 0 && g < n_row) ? sample_value(__builtin_nontemporal_load(x + g)) : 0.f;
        }
        __syncthreads();
        float* orow = p.out + (int64_t)row * p.out_stride;
        for (int k = threadIdx.x; k < p.tile_out; k += kThreads) {
            const int64_t m = m0 + k;
            if (m >= p.n_out) break;
            const int64_t t = (m + p.pre) * p.down;
            const int64_t im = t / p.up;
            const int phase = (int)(t - im * p.up);
            const float* xp = s_x + (int)(im - i_lo);      // x[i(m)]; tap j pairs with xp[-j]
            const float* bp = bank + (size_t)phase * p.taps_padded;
            float acc = 0.f;
            for (int j = 0; j < p.taps_padded; j += 4) {    // padded taps are zero and pair with staged (finite) samples
                const f32x4 w = *reinterpret_cast<const f32x4*>(bp + j);
                acc = fmaf(w.x, xp[-j], acc);
                acc = fmaf(w.y, xp[-j - 1], acc);
                acc = fmaf(w.z, xp[-j - 2], acc);
                acc = fmaf(w.w, xp[-j - 3], acc);
            }
            __builtin_nontemporal_store(acc, orow + m);
        }
    }
}


// ---- wave-uniform-phase kernel --------------------------------------------------------------------------------------
#ifndef MMFE_RS_STAGE_FIRST
#define MMFE_RS_STAGE_FIRST 0     // 1: the next tile is staged before this tile's outputs are stored (A/B builds)
#endif
#ifndef MMFE_RS_LOAD_ALL
#define MMFE_RS_LOAD_ALL 0      // 1: every thread requests all its pieces of the next tile, staged or not (A/B builds)
#endif
constexpr int kMaxPhaseThreads = 1024;
constexpr int kPhasePF = 36;             // samples of the next tile held per thread (9 four-sample pieces in the vector form)
constexpr int kPhaseLdsBytes = 158 * 1024;

typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(4))) const f32x8 const_f32x8;     // read through the scalar cache
typedef __attribute__((address_space(4))) const int const_int;

struct ResamplePhaseArgs {
    RowSource src;
    float* out;
    int n_rows;
    int n_in;
    int64_t n_out;             // outputs computed per row
    int64_t out_stride;
    int up, down;
    int L;                     // outputs between neighbouring lanes (a multiple of up and of Q)
    int D;                     // input samples between neighbouring lanes: L * down / up
    int nb;                    // lane blocks (64 lanes each) per tile
    int groups;                // L / Q: runs of Q consecutive outputs inside L
    int wlen;                  // samples staged in front of a tile's first lane: the longest window of any run (a multiple of 8)
    int wchunks;               // eight-sample chunks per run in the weight table (the longest run's count)
    int tile_in;               // staged input samples per tile
    int lds_in;                // floats of LDS they take (tile_in, plus one per D samples when padded)
    int pad;                   // 1: one unused float follows every D staged samples, see below
    unsigned inv_D;            // ceil(2^32 / D): (k * inv_D) >> 32 == k / D for every staged index k
    int tile_out;              // nb * 64 * L
    int ls;                    // floats between the output runs of neighbouring lanes in LDS (padded so that 64 lanes fall on distinct banks)
    int round_out;             // 0: a tile's outputs are collected whole.  Else: outputs of a lane per round of the block's waves (waves * Q),
                               // stored round by round (one lane block per tile)
    int tiles_per_row;
    int vec_out;               // output rows start 16-byte aligned: whole tiles are stored 16 bytes per lane
    const float* wt;           // [groups][wchunks][Q][8]: weight of window position s for output q of the run, zero outside its taps
    const int* gi;             // [groups]: LDS offset of window position 0 of lane 0's run;  [groups]: chunks of the run
};
// Padded staging.  The lanes of a wave read LDS D floats apart; an even D puts them on a few of the 32 banks (D = 160, the 16 kHz ->
// 44.1 kHz ratio: all 32 lanes of a half on ONE bank, every read serialised 32 times).  With pad = 1 staged sample k sits at
// k + k / D, so the lane stride is D + 1, odd, conflict-free.  A run's window then walks over the unused floats where it crosses a
// multiple of D; since all lanes start a whole number of D apart they cross at the same window position, and the host gives that
// position a zero weight: the multiply-add loop is the same, its window one position longer per crossing.  The unused floats are
// zeroed once per block (a zero weight times whatever LDS held could be NaN).

// four consecutive samples as they lie in memory
template <typename SMP> struct Piece;
template <> struct Piece<float> {
    typedef f32x4 type;
    static __device__ __forceinline__ f32x4 decode(f32x4 v) { return v; }
};
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <> struct Piece<int16_t> {
    typedef u32x2 type;
    static __device__ __forceinline__ f32x4 decode(u32x2 v) {
        constexpr float k = 1.0f / 32768.0f;
        return f32x4{(float)(int16_t)(v.x & 0xffffu) * k, (float)((int32_t)v.x >> 16) * k, (float)(int16_t)(v.y & 0xffffu) * k, (float)((int32_t)v.y >> 16) * k};
    }
};

// Q: consecutive outputs per lane.  VEC: rows start on 4-sample boundaries of suitably aligned memory and hold a multiple of 4
// samples each (no per-row lengths); every tile's staged range starts at a multiple of 4 samples by construction, so the input
// moves in whole four-sample pieces (16-byte / 8-byte loads, 16-byte LDS writes, one in-row test per piece).
template <typename SMP, int Q, bool VEC>
__global__ void __launch_bounds__(kMaxPhaseThreads) resample_phase_kernel(const ResamplePhaseArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_x = reinterpret_cast<float*>(smem);                       // lds_in (rounded up to 4)
    float* s_o = s_x + ((p.lds_in + 3) & ~3);                          // nb * 64 * ls, or 64 * ls of one round
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_waves = blockDim.x >> 6;
    const unsigned n_tiles = (unsigned)p.n_rows * (unsigned)p.tiles_per_row;      // < 2^31 (checked by the launcher)
    typedef typename Piece<SMP>::type piece_t;
    constexpr int kPieces = kPhasePF / 4;

    SMP pf[VEC ? 1 : kPhasePF];
    piece_t pv[VEC ? kPieces : 1];
    const int last_piece = (p.tile_in - 1) & ~3;                     // staged index of the tile's last four-sample piece
    // tile -> (row, first output); staged sample k of a tile is input sample lo + k of the row
    auto fetch = [&](unsigned tile) {
        if (tile >= n_tiles) return;
        const unsigned row = tile / (unsigned)p.tiles_per_row;
        const int m0 = (int)(tile - row * (unsigned)p.tiles_per_row) * p.tile_out;      // < n_out + tile_out < 2^31 (checked by the launcher)
        const int lo = (m0 / p.up) * p.down - p.wlen;
        const SMP* x = row_ptr_as<SMP>(p.src, row);
        const int n_row = row_samples(p.src, row, p.n_in);
        // every load reads inside the row (index clamped); samples outside the row are zeroed when the registers are staged, so
        // that nothing here waits for the loads
        if (VEC) {
#pragma unroll
            for (int q = 0; q < kPieces; ++q) {
                // The block's threads cover 36 samples each, a tile may stage a third of that: a piece beyond the staged range re-reads
                // the tile's last piece (one address per wave, a cache hit) instead of pulling the next tiles' samples from memory.
                // (Skipping those loads under a branch instead serialises all nine: the wait-count pass cannot see that the branch
                // around a load and the branch around its use go the same way, and waits for each load before issuing the next.)
                const int k = MMFE_RS_LOAD_ALL ? 4 * ((int)threadIdx.x + q * (int)blockDim.x) : min(4 * ((int)threadIdx.x + q * (int)blockDim.x), last_piece);
                const int g = lo + k;
                // non-temporal: the input streams through once; cached, it pushes the filter weights out of L2 and every scalar-cache
                // miss of the multiply-add loop then goes to memory (float32 input: 0.86 ms per 1024 clips cached, 0.66 non-temporal).
                pv[q] = __builtin_nontemporal_load(reinterpret_cast<const piece_t*>(x + (g < 0 ? 0 : (g + 4 <= n_row ? g : n_row - 4))));
            }
            return;
        }
        if (n_row <= 0) {            // block-uniform
#pragma unroll
            for (int q = 0; q < kPhasePF; ++q) pf[q] = SMP(0);      // (staged as zeros whatever they hold: every index is outside the row)
            return;
        }
#pragma unroll
        for (int q = 0; q < kPhasePF; ++q) {
            const int k = min((int)threadIdx.x + q * (int)blockDim.x, p.tile_in - 1);
            const int g = lo + k;
            pf[q] = __builtin_nontemporal_load(x + (g < 0 ? 0 : (g < n_row ? g : n_row - 1)));
        }
    };
    // ---- drain: lane run j holds outputs m0 + j * L .. + L - 1 of `row` ----
    auto drain = [&](unsigned row, int m0) {
        float* orow = p.out + (int64_t)row * p.out_stride;
        if (p.ls == p.L) {
            for (int k = threadIdx.x; k < p.tile_out; k += blockDim.x) {
                const int64_t m = m0 + k;
                if (m < p.n_out) __builtin_nontemporal_store(s_o[k], orow + m);
            }
        } else if (p.vec_out && m0 + p.tile_out <= p.n_out) {
            // a lane run is L contiguous floats in LDS and in memory, both 16-byte aligned
            const int l4 = p.L >> 2;
            for (int j = wave; j < p.nb * 64; j += n_waves) {
                for (int k = lane; k < l4; k += 64)
                    __builtin_nontemporal_store(*reinterpret_cast<const f32x4*>(s_o + j * p.ls + 4 * k), reinterpret_cast<f32x4*>(orow + m0 + (int64_t)j * p.L) + k);
            }
        } else {
            for (int j = wave; j < p.nb * 64; j += n_waves) {
                for (int k = lane; k < p.L; k += 64) {
                    const int64_t m = m0 + (int64_t)j * p.L + k;
                    if (m < p.n_out) __builtin_nontemporal_store(s_o[j * p.ls + k], orow + m);
                }
            }
        }
    };
    // ---- the outputs of one round (round_out > 0): lane run j holds outputs m0 + j * L + r * round_out .. + round_out - 1 ----
    auto drain_round = [&](unsigned row, int m0, int r) {
        float* orow = p.out + (int64_t)row * p.out_stride + m0;
        const int n_valid = min(p.round_out, p.L - r * p.round_out);      // (the last round of a lane run may be short)
        if (p.vec_out && m0 + p.tile_out <= p.n_out && (n_valid & 3) == 0 && (p.round_out & 3) == 0) {
            const int r4 = n_valid >> 2;                                  // 16-byte pieces per lane run
            for (int i = threadIdx.x; i < 64 * r4; i += blockDim.x) {
                const int j = i / r4, k = i - j * r4;
                __builtin_nontemporal_store(*reinterpret_cast<const f32x4*>(s_o + j * p.ls + 4 * k),
                                            reinterpret_cast<f32x4*>(orow + (int64_t)j * p.L + r * p.round_out) + k);
            }
        } else {
            for (int i = threadIdx.x; i < 64 * n_valid; i += blockDim.x) {
                const int j = i / n_valid, k = i - j * n_valid;
                const int64_t m = (int64_t)j * p.L + r * p.round_out + k;
                if (m0 + m < p.n_out) __builtin_nontemporal_store(s_o[j * p.ls + k], orow + m);
            }
        }
    };
#ifdef MMFE_RS_STAMPS
    unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0}, st_prev = 0, st_now = 0;
#define RS_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_now)::"memory"); __builtin_amdgcn_sched_barrier(0); if ((i) >= 0) st_acc[(i)] += st_now - st_prev; st_prev = st_now; } while (0)
#else
#define RS_STAMP(i) do { } while (0)
#endif
    if (p.pad) {                                                        // the unused floats, once: staging never writes them
        for (int j = threadIdx.x; j * (p.D + 1) + p.D < p.lds_in; j += blockDim.x) s_x[j * (p.D + 1) + p.D] = 0.f;
    }
    // ---- stage: the registers `fetch` filled -> LDS ----
    auto stage = [&](unsigned tile) {
        const unsigned row = tile / (unsigned)p.tiles_per_row;
        const int m0 = (int)(tile - row * (unsigned)p.tiles_per_row) * p.tile_out;
        const int lo = (m0 / p.up) * p.down - p.wlen;
        const int n_row = row_samples(p.src, row, p.n_in);
        // (a thread's LDS positions are the same in every tile; left visible, the optimiser keeps all of them in registers across
        // the loop - 36 per thread in the padded layout - and spills)
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        if (VEC) {
#pragma unroll
            for (int q = 0; q < kPieces; ++q) {
                const int k = 4 * (tid + q * (int)blockDim.x);
                const int g = lo + k;
                const f32x4 v = (g >= 0 && g + 4 <= n_row) ? Piece<SMP>::decode(pv[q]) : f32x4{0.f, 0.f, 0.f, 0.f};
                if (k < p.tile_in) {
                    if (!p.pad) {
                        *reinterpret_cast<f32x4*>(s_x + k) = v;
                    } else {                                      // (lds_in has room for the whole last piece)
#pragma unroll
                        for (int i = 0; i < 4; ++i) s_x[k + i + (int)__umulhi((unsigned)(k + i), p.inv_D)] = v[i];
                    }
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < kPhasePF; ++q) {
                const int k = tid + q * (int)blockDim.x;
                const int g = lo + k;
                if (k < p.tile_in) s_x[p.pad ? k + (int)__umulhi((unsigned)k, p.inv_D) : k] = (g >= 0 && g < n_row) ? sample_value(pf[q]) : 0.f;
            }
        }
    };
    // A step: request the next tile's samples, run this tile's multiply-adds, barrier, store this tile's outputs, stage the next
    // tile, barrier.  Loads and stores return in order, so the stage's wait for its samples also covers the stores issued a moment
    // before it.  Staging first (-DMMFE_RS_STAGE_FIRST=1: everything the wait covers is then a whole multiply-add phase old) measured
    // the same to 4 % slower over eight ratios: the block's 16 waves meet at the barrier behind both anyway.
    fetch(blockIdx.x);
    if (blockIdx.x < n_tiles) stage(blockIdx.x);
    lds_barrier();
    for (unsigned tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        RS_STAMP(-1);
        const unsigned row = tile / (unsigned)p.tiles_per_row;
        const int m0 = (int)(tile - row * (unsigned)p.tiles_per_row) * p.tile_out;
        // The next tile's samples, requested here all at once.  s_memtime stamps (-DMMFE_RS_STAMPS) show the later waves of a block
        // sitting in the issue of these loads for up to 4 000 of a step's 24 000 cycles (16 waves x 9 loads pass the CU's 64 B/clk
        // path to L1 one after the other).  Having each wave request one piece per chunk of its first run instead moves the waiting
        // into the runs and the step takes as long (measured: 16-bit rows +0.6 %, float32 rows +5 %).
        fetch(tile + gridDim.x);
        RS_STAMP(0);      // fetch issue
        // ---- runs: (lane block, group) pairs spread over the waves.  Every wave takes the same number of turns (rounds end in
        //      barriers when the outputs leave round by round); a turn beyond the last run is empty. ----
        const int n_tasks = p.nb * p.groups;
        int r = 0;
        for (int t0 = 0; t0 < n_tasks; t0 += n_waves, ++r) {
            const int task = t0 + wave;
            if (task < n_tasks) {
                const int blk = task / p.groups;
                const int u = task - blk * p.groups;
                const float* xl = s_x + (blk * 64 + lane) * (p.D + p.pad) + ((const_int*)p.gi)[u];
                const int chunks = ((const_int*)p.gi)[p.groups + u];
#ifdef MMFE_RS_ABL_SAMEW      // diagnostic build: every run reads group 0's weights (wrong results; shows what scalar-cache misses cost)
                const_f32x8* w = ((const_f32x8*)p.wt);
#else
                const_f32x8* w = ((const_f32x8*)p.wt) + (int64_t)u * p.wchunks * Q;
#endif
                float acc[Q];
#pragma unroll
                for (int q = 0; q < Q; ++q) acc[q] = 0.f;
                // (requesting the next chunk's weights and samples ahead of the multiply-adds, in a second register set, measured 3 %
                // slower: scalar loads and LDS reads share one counter and scalar loads return out of order, so any wait is a wait for
                // everything, and the other waves of the SIMD already fill the gaps)
                // (skipping, in the window's first and last chunks, the outputs that have no taps there - 13 % of the multiply-adds - measured
                // 5 to 23 % slower: the wave-uniform branches cost the loop its overlap of loads and arithmetic)
                for (int c = 0; c < chunks; ++c) {
                    f32x8 wq[Q];
#pragma unroll
                    for (int q = 0; q < Q; ++q) wq[q] = w[c * Q + q];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float xv = xl[-(c * 8 + e)];
#pragma unroll
                        for (int q = 0; q < Q; ++q) acc[q] = fmaf(wq[q][e], xv, acc[q]);
                    }
                }
                float* so = p.round_out > 0 ? s_o + lane * p.ls + Q * wave : s_o + (blk * 64 + lane) * p.ls + Q * u;
                if (Q == 4) {
                    *reinterpret_cast<f32x4*>(so) = f32x4{acc[0], acc[1], acc[2], acc[3]};
                } else {
#pragma unroll
                    for (int q = 0; q < Q; ++q) so[q] = acc[q];
                }
            }
            if (p.round_out > 0 && t0 + n_waves < n_tasks) {          // every round but the last: its outputs leave here
                lds_barrier();
                drain_round(row, m0, r);
                lds_barrier();                                        // (the output area is written again at the end of the next round's runs)
            }
        }
        RS_STAMP(1);      // runs
        lds_barrier();    // the tile's readers are done with the staged samples; its (last round of) outputs are complete
        RS_STAMP(2);      // barrier after runs
#if MMFE_RS_STAGE_FIRST
        if (tile + gridDim.x < n_tiles) stage(tile + gridDim.x);
        RS_STAMP(3);      // stage
#endif
        if (p.round_out > 0) drain_round(row, m0, r - 1);
        else drain(row, m0);
        RS_STAMP(4);      // drain
#if !MMFE_RS_STAGE_FIRST
        if (tile + gridDim.x < n_tiles) stage(tile + gridDim.x);
#endif
        lds_barrier();    // the next tile's samples are in place; the output area is free
        RS_STAMP(5);      // barrier after stage and drain
    }
#ifdef MMFE_RS_STAMPS
    if (blockIdx.x == 3 && lane == 0 && (wave == 0 || wave == 9))
        printf("wave %d, shader-clock cycles: load issue %llu  runs %llu  barrier %llu  stage %llu  drain %llu  barrier %llu\n", wave, st_acc[0], st_acc[1], st_acc[2], st_acc[3], st_acc[4], st_acc[5]);
#endif
}

// ---- the same for 16-bit PCM rows, two blocks per CU ------------------------------------------------------------------
// With 113 KB of staged float32 samples per 64 lanes only one block of the kernel above fits a CU, and its SIMDs issue nothing
// while it stages, requests loads or stores.  PCM rows can stay 16-bit in LDS (56 KB per tile) and a block can store its outputs
// round by round (16 runs of Q outputs per lane: 21 KB instead of 41 KB): two blocks then share a CU.  The samples are converted
// where they are read (one 2-byte LDS read and one v_cvt each) and the 1 / 32768 of the decode rides on the weights, which is
// exact (a power-of-two scale commutes with every rounding of the multiply-adds): the outputs equal the other kernel's bit for
// bit.  Measured gain: 4 % (0.637 against 0.663 ms per 1024 ten-second clips) - the runs are bound by instruction issue (every
// instruction of any kind costs a SIMD about 2.5 cycles: a pure-arithmetic build scales with its instruction count), the 16
// extra instructions per chunk of 40 multiply-adds (reads, converts) eat most of what the overlap returns.
// Needs: whole four-sample pieces (VEC above), one lane block per tile, a whole number of rounds (groups % waves == 0).
struct ResamplePhase16Args {
    ResamplePhaseArgs a;       // wt: the weights scaled by 1 / 32768;  ls: floats between the lane runs of one ROUND in LDS
    int rounds;                // groups / waves
    int round_out;             // outputs of a lane per round: waves * Q
};

template <int Q>
__global__ void __launch_bounds__(kMaxPhaseThreads, 8) resample_phase16_kernel(const ResamplePhase16Args pp) {
    const ResamplePhaseArgs& p = pp.a;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int16_t* s_x = reinterpret_cast<int16_t*>(smem);                                           // tile_in (rounded up to 8)
    float* s_o = reinterpret_cast<float*>(smem + 2 * (size_t)((p.tile_in + 7) & ~7));          // 64 * ls
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_waves = blockDim.x >> 6;
    const unsigned n_tiles = (unsigned)p.n_rows * (unsigned)p.tiles_per_row;
    const int chunks = p.wlen >> 3;
    constexpr int kPieces = kPhasePF / 4;
    u32x2 pv[kPieces];
    const int last_piece = (p.tile_in - 1) & ~3;
    // -DMMFE_RS16_INTERIOR=1: tiles whose staged range lies inside the row (all but a row's first and last) take paths without
    // per-piece tests.  Fewer instructions, but at this kernel's 64-register cap the second code path costs more scalar-register
    // spills than the tests did: measured 1.3 % slower, so it is off.
#ifndef MMFE_RS16_INTERIOR
#define MMFE_RS16_INTERIOR 0
#endif
    const int span = 4 * kPieces * (int)blockDim.x;                  // samples the block's loads cover
    auto fetch = [&](unsigned tile) {
        if (tile >= n_tiles) return;
        const unsigned row = tile / (unsigned)p.tiles_per_row;
        const int m0 = (int)(tile - row * (unsigned)p.tiles_per_row) * p.tile_out;
        const int lo = (m0 / p.up) * p.down - p.wlen;
        const int16_t* x = row_ptr_as<int16_t>(p.src, row);
        if (MMFE_RS16_INTERIOR && lo >= 0 && lo + span <= p.n_in) {   // block-uniform
            const u32x2* xp = reinterpret_cast<const u32x2*>(x + lo) + threadIdx.x;
#pragma unroll
            for (int q = 0; q < kPieces; ++q) pv[q] = __builtin_nontemporal_load(xp + q * (int)blockDim.x);
            return;
        }
#pragma unroll
        for (int q = 0; q < kPieces; ++q) {
            // (a piece beyond the staged range re-reads the tile's last piece, see resample_phase_kernel)
            const int k = MMFE_RS_LOAD_ALL ? 4 * ((int)threadIdx.x + q * (int)blockDim.x) : min(4 * ((int)threadIdx.x + q * (int)blockDim.x), last_piece);
            const int g = lo + k;
            pv[q] = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(x + (g < 0 ? 0 : (g + 4 <= p.n_in ? g : p.n_in - 4))));
        }
    };
    // ---- stage: the pieces as they are, zeros outside the row ----
    auto stage = [&](unsigned tile) {
        const unsigned row = tile / (unsigned)p.tiles_per_row;
        const int m0 = (int)(tile - row * (unsigned)p.tiles_per_row) * p.tile_out;
        const int lo = (m0 / p.up) * p.down - p.wlen;
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));       // (keeps the nine LDS positions from being held in registers across the loop)
        if (MMFE_RS16_INTERIOR && lo >= 0 && lo + span <= p.n_in) {
#pragma unroll
            for (int q = 0; q < kPieces; ++q) {
                const int k = 4 * (tid + q * (int)blockDim.x);
                if (k < p.tile_in) *reinterpret_cast<u32x2*>(s_x + k) = pv[q];
            }
        } else {
#pragma unroll
            for (int q = 0; q < kPieces; ++q) {
                const int k = 4 * (tid + q * (int)blockDim.x);
                const int g = lo + k;
                if (k < p.tile_in) *reinterpret_cast<u32x2*>(s_x + k) = (g >= 0 && g + 4 <= p.n_in) ? pv[q] : u32x2{0u, 0u};
            }
        }
    };
    // ---- a round's outputs: lane run j holds outputs m0 + j * L + r * round_out .. + round_out - 1 ----
    auto drain_round = [&](float* orow, int m0, int r, bool whole) {
        if (whole) {
            const int r4 = pp.round_out >> 2;                   // 16-byte pieces per lane run
            for (int i = threadIdx.x; i < 64 * r4; i += blockDim.x) {
                const int j = i / r4, k = i - j * r4;
                __builtin_nontemporal_store(*reinterpret_cast<const f32x4*>(s_o + j * p.ls + 4 * k),
                                            reinterpret_cast<f32x4*>(orow + (int64_t)j * p.L + r * pp.round_out) + k);
            }
        } else {
            for (int i = threadIdx.x; i < 64 * pp.round_out; i += blockDim.x) {
                const int j = i / pp.round_out, k = i - j * pp.round_out;
                const int64_t m = (int64_t)j * p.L + r * pp.round_out + k;
                if (m0 + m < p.n_out) __builtin_nontemporal_store(s_o[j * p.ls + k], orow + m);
            }
        }
    };
    // step order as in resample_phase_kernel
    fetch(blockIdx.x);
    if (blockIdx.x < n_tiles) stage(blockIdx.x);
    lds_barrier();
    for (unsigned tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const unsigned row = tile / (unsigned)p.tiles_per_row;
        const int m0 = (int)(tile - row * (unsigned)p.tiles_per_row) * p.tile_out;
        fetch(tile + gridDim.x);
        float* orow = p.out + (int64_t)row * p.out_stride + m0;
        const bool whole = p.vec_out && m0 + p.tile_out <= p.n_out;
        for (int r = 0; r < pp.rounds; ++r) {
            const int u = r * n_waves + wave;
            const int16_t* xl = s_x + p.wlen + lane * p.D + ((const_int*)p.gi)[u];
            typedef __attribute__((address_space(3))) const int16_t lds_i16;
            lds_i16* xl3 = (lds_i16*)xl;
            const_f32x8* w = ((const_f32x8*)p.wt) + (int64_t)u * chunks * Q;
            float acc[Q];
#pragma unroll
            for (int q = 0; q < Q; ++q) acc[q] = 0.f;
            for (int c = 0; c < chunks; ++c) {
                f32x8 wq[Q];
#pragma unroll
                for (int q = 0; q < Q; ++q) wq[q] = w[c * Q + q];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    // one 2-byte LDS read per sample (this file is compiled with the IR load/store vectoriser off, see the Makefile:
                    // left on, it fetches the chunk's 8 samples with one ds_read_b128 at a 2-byte aligned address, which the LDS
                    // serves in 73 cycles per wave)
                    const float xv = (float)xl3[-(c * 8 + e)];
#pragma unroll
                    for (int q = 0; q < Q; ++q) acc[q] = fmaf(wq[q][e], xv, acc[q]);
                }
            }
            float* so = s_o + lane * p.ls + Q * wave;
            if (Q == 4) {
                *reinterpret_cast<f32x4*>(so) = f32x4{acc[0], acc[1], acc[2], acc[3]};
            } else {
#pragma unroll
                for (int q = 0; q < Q; ++q) so[q] = acc[q];
            }
            lds_barrier();
            if (r + 1 < pp.rounds) {
                drain_round(orow, m0, r, whole);
                lds_barrier();                      // (the output area is written again at the end of the next round's runs)
            }
        }
        // (the barrier behind the last round: the tile's readers are done with the staged samples)
#if MMFE_RS_STAGE_FIRST
        if (tile + gridDim.x < n_tiles) stage(tile + gridDim.x);
#endif
        drain_round(orow, m0, pp.rounds - 1, whole);
#if !MMFE_RS_STAGE_FIRST
        if (tile + gridDim.x < n_tiles) stage(tile + gridDim.x);
#endif
        lds_barrier();
    }
}

}  // namespace

int resample_prepare(Plan& plan) {
    const mmfe_spec& s = plan.spec;
    const int up = s.n_fft, down = s.hop, n_taps = s.win_len;
    if (up < 1 || down < 1 || up > 4096 || down > 65536) { set_error("RESAMPLE: up (n_fft) must lie in [1, 4096], down (hop) in [1, 65536]"); return MMFE_EINVAL; }
    int a = up, b = down;
    while (b) { const int t = a % b; a = b; b = t; }
    if (a != 1) { set_error("RESAMPLE: up and down must be coprime (divide both by %d)", a); return MMFE_EINVAL; }
    if (n_taps < 1 || n_taps % 2 == 0) { set_error("RESAMPLE: the filter needs an odd number of taps (win_len), got %d", n_taps); return MMFE_EINVAL; }
    if (s.n_mel != 0 || s.log_kind != MMFE_LOG_NONE || s.out_dtype != MMFE_OUT_F32 || s.clamp_below_max > 0.f || s.drop_last ||
        s.layout != MMFE_LAYOUT_TIME_MAJOR) {
        set_error("RESAMPLE: no mel bank, log, clamp or 16-bit output");
        return MMFE_EINVAL;
    }
    // alignment of scipy.signal.resample_poly: h is padded in front so that the kept outputs start on the input grid
    const int half = (n_taps - 1) / 2;
    const int pre_pad = down - half % down;
    const int padded_len = pre_pad + n_taps;
    const int taps = (padded_len + up - 1) / up;
    const int taps_padded = (taps + 3) & ~3;
    std::vector<float> bank((size_t)up * taps_padded, 0.f);
    for (int p = 0; p < up; ++p)
        for (int j = 0; j < taps; ++j) {
            const int idx = p + j * up - pre_pad;
            if (idx >= 0 && idx < n_taps) bank[(size_t)p * taps_padded + j] = plan.window[idx];
        }
    // the staged input of even a one-thread-per-output tile must fit
    if ((int64_t)kThreads * down / up + taps_padded + 8 > kMaxTileIn) {
        set_error("RESAMPLE: ratio %d/%d with %d taps per phase needs more than %d staged samples per tile", up, down, taps, kMaxTileIn);
        return MMFE_EINVAL;
    }
    int rc = upload(&plan.d_window, bank);
    if (rc) return rc;

    // ---- wave-uniform-phase kernel: applicable when lanes L = up * ceil(Q / up) outputs apart leave whole runs of Q ----
    plan.rs_L = 0;
    const char* force = std::getenv("MMFE_RESAMPLE_FORM");          // tuning / test knob, read once per plan: "general" forces the general kernel
    if (force != nullptr && std::string(force) == "general") return MMFE_OK;
    const char* force_q = std::getenv("MMFE_RESAMPLE_Q");           // tuning knob, read once per plan: outputs per lane run (4 or 5)
    const int pre = (half + pre_pad) / down;
    // Q = 4 or 5 consecutive outputs per lane: whichever fills the rounds of a block's waves better.  The multiply-adds of a
    // tile take `rounds` turns of (waves per SIMD) runs of wlen * Q each.
    struct Layout {
        int Q = 0, L = 0, D = 0, groups = 0, wlen = 0, wchunks = 0, i_max = 0, nb = 0, threads = 0, tile_in = 0, lds_in = 0, ls = 0, pad = 0, round_out = 0;
        double cost = 0;
        std::vector<int32_t> table;      // [groups] LDS offset of window position 0, [groups] chunks
        std::vector<int> io, ph, last;   // per output: newest input sample and phase;  per run: its newest input sample
    };
    Layout best;
    // Q = 3 and 7 serve the odd `up` of upsampling ratios (441/160, 147/80, 3/1, 3/2: lanes `up` outputs apart leave whole runs of 3 or 7);
    // their lane runs are not whole 16-byte pieces, so those tiles are stored 4 bytes per lane (still contiguous rows)
    for (int Q : {4, 5, 7, 3}) {
        if (force_q != nullptr && std::atoi(force_q) != Q) continue;
        Layout c;
        c.Q = Q;
        const int64_t L = (int64_t)up * ((Q + up - 1) / up);
        if (L % Q != 0 || L > 4096) continue;
        if (L % 4 != 0 && Q != 3 && Q != 7) continue;
        c.L = (int)L;
        c.D = (int)(L * down / up);
        c.groups = (int)(L / Q);
        c.io.resize(c.groups * Q);
        c.ph.resize(c.groups * Q);
        c.last.resize(c.groups);
        int max_back = 0;
        for (int u = 0; u < c.groups; ++u) {
            for (int q = 0; q < Q; ++q) {
                const int64_t t = (int64_t)(Q * u + q + pre) * down;
                c.io[u * Q + q] = (int)(t / up);
                c.ph[u * Q + q] = (int)(t % up);
            }
            c.last[u] = c.io[u * Q + Q - 1];
            c.i_max = std::max(c.i_max, c.last[u]);
            max_back = std::max(max_back, c.last[u] - c.io[u * Q]);
        }
        // An even lane stride D puts the 32 lanes of a read on few banks: one unused float per D staged samples makes it odd.  Every
        // unused float a window walks over is a chunk position without work, so a small D pays more than it gains (measured, 256
        // clips: D = 2: 0.099 ms plain, 0.119 padded; D = 8: 0.210 / 0.219; D = 12: 0.251 / 0.202; D = 160: 0.325 / 0.096).
        const char* force_pad = std::getenv("MMFE_RESAMPLE_PAD");      // tuning / test knob, read once per plan: 0 or 1
        c.pad = c.D < 2 ? 0 : force_pad != nullptr ? (std::atoi(force_pad) != 0 ? 1 : 0) : (c.D % 2 == 0 && c.D > 8 ? 1 : 0);
        // Samples staged in front of lane 0's first run.  A run reads whole chunks of 8 LDS floats downwards from its newest sample:
        // at most taps + max_back samples, the unused floats among them, and up to 7 floats of the last chunk's rounding.
        const int crossings = c.pad ? (taps + max_back) / c.D + 1 : 0;
        c.wlen = (taps + max_back + crossings + 6 + 7) & ~7;
        auto lds_pos = [&](int64_t k) { return k + (c.pad ? k / c.D : 0); };
        // per run: LDS offset of its window position 0 (lane 0) and its chunk count; position s of the window is the float s below
        c.table.resize(2 * c.groups);
        int64_t fma_chunks = 0;
        for (int u = 0; u < c.groups; ++u) {
            const int64_t top = c.wlen + c.last[u];                               // staged index of the run's newest sample (lane 0)
            const int64_t bottom = top - (taps - 1 + c.last[u] - c.io[u * Q]);     // ... and of its oldest (>= 0: wlen covers the longest window)
            const int n_pos = (int)(lds_pos(top) - lds_pos(bottom)) + 1;
            c.table[u] = (int32_t)lds_pos(top);
            c.table[c.groups + u] = (n_pos + 7) / 8;
            c.wchunks = std::max(c.wchunks, (n_pos + 7) / 8);
            fma_chunks += (n_pos + 7) / 8;
        }
        // lane runs in the output area: a multiple of 4 floats whose quarter is odd keeps the 16-byte accesses of 64 lanes on distinct banks
        // (runs that are not a multiple of 4 floats are stored float by float: any odd stride is conflict-free)
        auto lane_stride = [](int64_t n) { return n % 4 == 0 ? ((n / 4) % 2 == 1 ? (int)n : (int)n + 4) : (n % 2 == 1 ? (int)n : (int)n + 1); };
        c.ls = lane_stride(L);
        // lane blocks per tile: as many as LDS and the prefetch registers hold
        for (int cand = 1; cand <= 64; ++cand) {
            const int64_t t_in = c.wlen + (int64_t)cand * 64 * c.D + c.i_max + 1;
            const int64_t l_in = lds_pos(((t_in + 3) & ~(int64_t)3) - 1) + 1;      // (samples move in pieces of 4)
            const int64_t bytes = 4 * (((l_in + 3) & ~(int64_t)3) + (int64_t)cand * 64 * c.ls);
            const int tasks = cand * c.groups;
            const int rounds = (tasks + 15) / 16;
            const int nw = (tasks + rounds - 1) / rounds;
            if (bytes > kPhaseLdsBytes || t_in > (int64_t)kPhasePF * 64 * nw || t_in * c.D >= (int64_t)1 << 31) break;
            c.nb = cand; c.threads = 64 * nw; c.tile_in = (int)t_in; c.lds_in = (int)l_in;
            c.cost = (double)rounds * ((nw + 3) / 4) * 8.0 * fma_chunks / c.groups * Q / ((double)cand * 64 * L);
        }
        if (c.nb == 0) {
            // One lane block's outputs do not fit beside its staged input (22.05 -> 16 kHz: 112 + 80 KB): the outputs leave round by
            // round of the block's waves instead, through an area of 64 lanes x (waves * Q) floats.
            const int64_t t_in = c.wlen + (int64_t)64 * c.D + c.i_max + 1;
            const int64_t l_in = lds_pos(((t_in + 3) & ~(int64_t)3) - 1) + 1;
            for (int nw = 16; nw >= 4 && c.nb == 0; --nw) {
                if (nw > c.groups) continue;
                const int round_out = nw * Q;
                const int ls_r = lane_stride(round_out);
                const int64_t bytes = 4 * (((l_in + 3) & ~(int64_t)3) + (int64_t)64 * ls_r);
                if (bytes > kPhaseLdsBytes || t_in > (int64_t)kPhasePF * 64 * nw || t_in * c.D >= (int64_t)1 << 31) continue;
                const int rounds = (c.groups + nw - 1) / nw;
                c.nb = 1; c.threads = 64 * nw; c.tile_in = (int)t_in; c.lds_in = (int)l_in; c.ls = ls_r; c.round_out = round_out;
                c.cost = 1.05 * (double)rounds * ((nw + 3) / 4) * 8.0 * fma_chunks / c.groups * Q / ((double)64 * L);      // (the rounds' barriers)
            }
        }
        if (c.nb > 0 && (best.Q == 0 || c.cost < best.cost)) best = c;
    }
    if (best.Q == 0) return MMFE_OK;
    const int Q = best.Q;
    {
        // weights by LDS position: position s of run u holds staged sample top - s', where s' counts only the samples (the unused
        // floats of a padded layout keep weight zero)
        const Layout& c = best;
        auto lds_pos = [&](int64_t k) { return k + (c.pad ? k / c.D : 0); };
        std::vector<float> wt((size_t)c.groups * c.wchunks * 8 * Q, 0.f);
        for (int u = 0; u < c.groups; ++u) {
            const int64_t top = c.wlen + c.last[u];
            for (int back = 0; back <= taps - 1 + c.last[u] - c.io[u * Q]; ++back) {          // sample top - back
                const int pos = (int)(lds_pos(top) - lds_pos(top - back));
                for (int q = 0; q < Q; ++q) {
                    const int t = back - (c.last[u] - c.io[u * Q + q]);
                    if (t >= 0 && t < taps)
                        wt[(((size_t)u * c.wchunks + pos / 8) * Q + q) * 8 + pos % 8] = bank[(size_t)c.ph[u * Q + q] * taps_padded + t];
                }
            }
        }
        if ((rc = upload(&plan.d_tw_a, wt))) return rc;
        if ((rc = upload(&plan.d_mel_start, c.table))) return rc;
    }
    plan.rs_q = Q; plan.rs_L = best.L; plan.rs_D = best.D; plan.rs_nb = best.nb; plan.rs_groups = best.groups; plan.rs_wlen = best.wlen;
    plan.rs_wchunks = best.wchunks; plan.rs_tile_in = best.tile_in; plan.rs_lds_in = best.lds_in; plan.rs_pad = best.pad;
    plan.rs_round_out = best.round_out; plan.rs_ls = best.ls; plan.rs_threads = best.threads;

    // ---- 16-bit rows: the two-blocks-per-CU form (samples stay 16-bit in LDS, outputs leave round by round) ----
    plan.rs16_threads = 0;
    if (s.in_dtype != MMFE_IN_S16 || (force != nullptr && std::string(force) == "wide")) return MMFE_OK;   // "wide": the one-block form for PCM too
    for (int Q16 = 5; Q16 >= 4 && plan.rs16_threads == 0; --Q16) {
        if (force_q != nullptr && std::atoi(force_q) != Q16) continue;
        const int64_t L = (int64_t)up * ((Q16 + up - 1) / up);
        if (L % Q16 != 0 || L % 4 != 0 || L > 4096) continue;
        const int groups = (int)(L / Q16);
        const int D = (int)(L * down / up);
        std::vector<int32_t> gi(groups);
        std::vector<int> io(groups * Q16), ph(groups * Q16);
        int max_back = 0, i_max = 0;
        for (int u = 0; u < groups; ++u) {
            for (int q = 0; q < Q16; ++q) {
                const int64_t t = (int64_t)(Q16 * u + q + pre) * down;
                io[u * Q16 + q] = (int)(t / up);
                ph[u * Q16 + q] = (int)(t % up);
            }
            gi[u] = io[u * Q16 + Q16 - 1];
            i_max = std::max(i_max, gi[u]);
            max_back = std::max(max_back, gi[u] - io[u * Q16]);
        }
        const int wlen16 = (taps + max_back + 7) & ~7;
        const int64_t t_in = wlen16 + (int64_t)64 * D + i_max + 1;
        for (int nw = 16; nw >= 8; --nw) {          // the most waves that divide the groups into whole rounds
            if (groups % nw != 0 || (nw * Q16) % 4 != 0) continue;
            if (groups / nw > 2) continue;              // more rounds per tile mean more barriers around the rounds' stores than the second block returns
            const int round_out = nw * Q16;
            const int ls16 = (round_out / 4) % 2 == 1 ? round_out : round_out + 4;
            const int64_t bytes = 2 * ((t_in + 7) & ~(int64_t)7) + (int64_t)64 * ls16 * 4;
            if (bytes > kPhaseLdsBytes / 2 || t_in > (int64_t)kPhasePF * 64 * nw) continue;
            std::vector<float> wt16((size_t)groups * wlen16 * Q16, 0.f);
            for (int u = 0; u < groups; ++u)
                for (int sidx = 0; sidx < wlen16; ++sidx)
                    for (int q = 0; q < Q16; ++q) {
                        const int t = sidx - (gi[u] - io[u * Q16 + q]);
                        if (t >= 0 && t < taps)
                            wt16[(((size_t)u * (wlen16 / 8) + sidx / 8) * Q16 + q) * 8 + sidx % 8] =
                                bank[(size_t)ph[u * Q16 + q] * taps_padded + t] * (1.0f / 32768.0f);
                    }
            if ((rc = upload(&plan.d_tw_b, wt16))) return rc;
            if ((rc = upload(&plan.d_mel_niter, gi))) return rc;
            plan.rs16_q = Q16; plan.rs16_L = (int)L; plan.rs16_D = D; plan.rs16_groups = groups; plan.rs16_wlen = wlen16;
            plan.rs16_tile_in = (int)t_in; plan.rs16_ls = ls16; plan.rs16_threads = 64 * nw; plan.rs16_round_out = round_out;
            break;
        }
    }
    return MMFE_OK;
}

int64_t resample_out_len(const mmfe_spec& s, int64_t n) {
    const int64_t v = n * s.n_fft;
    return v / s.hop + (v % s.hop ? 1 : 0);
}

int resample_launch(const Plan& plan, const RowSource& src, int64_t n_rows, int64_t n_samples, int64_t n_out_natural, int64_t out_len,
                    float* out, hipStream_t stream) {
    const mmfe_spec& s = plan.spec;
    const int64_t n_out = std::min(n_out_natural, out_len);
    if (n_rows <= 0 || n_out <= 0) return MMFE_OK;
    if (n_samples >= ((int64_t)1 << 31) / 2 || n_rows >= (int64_t)1 << 31) { set_error("RESAMPLE: row too long for one launch"); return MMFE_EINVAL; }
    const int up = s.n_fft, down = s.hop, n_taps = s.win_len;
    const bool pcm = s.in_dtype == MMFE_IN_S16;
    if (plan.rs16_threads > 0 && src.offsets == nullptr && src.row_len == nullptr && n_samples % 4 == 0 && n_samples >= 4 && src.stride % 4 == 0 &&
        reinterpret_cast<uintptr_t>(src.base) % 8 == 0) {
        ResamplePhase16Args b{};
        ResamplePhaseArgs& a = b.a;
        a.src = src;
        a.out = out;
        a.n_rows = (int)n_rows;
        a.n_in = (int)n_samples;
        a.n_out = n_out;
        a.out_stride = out_len;
        a.up = up; a.down = down;
        a.L = plan.rs16_L; a.D = plan.rs16_D; a.nb = 1; a.groups = plan.rs16_groups; a.wlen = plan.rs16_wlen;
        a.tile_in = plan.rs16_tile_in; a.ls = plan.rs16_ls;
        a.tile_out = 64 * a.L;
        a.tiles_per_row = (int)((n_out + a.tile_out - 1) / a.tile_out);
        a.wt = plan.d_tw_b;
        a.gi = plan.d_mel_niter;
        a.vec_out = (reinterpret_cast<uintptr_t>(out) % 16 == 0 && out_len % 4 == 0) ? 1 : 0;
        b.rounds = a.groups / (plan.rs16_threads / 64);
        b.round_out = plan.rs16_round_out;
        const size_t lds = 2 * (size_t)((a.tile_in + 7) & ~7) + (size_t)64 * a.ls * 4;
        typedef void (*phase16_fn)(const ResamplePhase16Args);
        const phase16_fn kfn = plan.rs16_q == 5 ? &resample_phase16_kernel<5> : &resample_phase16_kernel<4>;
        int per_cu = 1, cus = 256;
        const int rc = kernel_prepare(reinterpret_cast<const void*>(kfn), plan.rs16_threads, lds, &per_cu, &cus);
        if (rc) return rc;
        const int64_t n_tiles = n_rows * a.tiles_per_row;
        if (n_tiles >= ((int64_t)1 << 31) - 65536 || n_out + a.tile_out >= (int64_t)1 << 31) { set_error("RESAMPLE: batch too large for one launch"); return MMFE_EINVAL; }
        const int grid = (int)std::min<int64_t>(n_tiles, (int64_t)cus * std::max(1, per_cu));
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(plan.rs16_threads), lds, stream, b);
        MMFE_HIP_CHECK(hipGetLastError());
        return MMFE_OK;
    }
    if (plan.rs_L > 0) {
        ResamplePhaseArgs a{};
        a.src = src;
        a.out = out;
        a.n_rows = (int)n_rows;
        a.n_in = (int)n_samples;
        a.n_out = n_out;
        a.out_stride = out_len;
        a.up = up; a.down = down;
        a.L = plan.rs_L; a.D = plan.rs_D; a.nb = plan.rs_nb; a.groups = plan.rs_groups; a.wlen = plan.rs_wlen; a.wchunks = plan.rs_wchunks;
        a.tile_in = plan.rs_tile_in; a.lds_in = plan.rs_lds_in; a.pad = plan.rs_pad; a.ls = plan.rs_ls; a.round_out = plan.rs_round_out;
        a.inv_D = (unsigned)(((uint64_t)1 << 32) / (uint64_t)a.D + 1);
        a.tile_out = a.nb * 64 * a.L;
        a.tiles_per_row = (int)((n_out + a.tile_out - 1) / a.tile_out);
        a.wt = plan.d_tw_a;
        a.gi = plan.d_mel_start;
        a.vec_out = (reinterpret_cast<uintptr_t>(out) % 16 == 0 && out_len % 4 == 0 && a.L % 4 == 0) ? 1 : 0;
        const size_t lds = 4 * ((size_t)((a.lds_in + 3) & ~3) + (size_t)a.nb * 64 * a.ls);
        // whole four-sample pieces: rows on 4-sample boundaries of aligned memory, a multiple of 4 samples each, no per-row lengths
        const bool vec = src.offsets == nullptr && src.row_len == nullptr && n_samples % 4 == 0 && n_samples >= 4 && src.stride % 4 == 0 &&
                         reinterpret_cast<uintptr_t>(src.base) % (pcm ? 8 : 16) == 0;
        typedef void (*phase_fn)(const ResamplePhaseArgs);
        auto pick = [&](auto smp_tag) -> phase_fn {
            typedef decltype(smp_tag) SMP;
            switch (plan.rs_q) {
                case 3: return vec ? &resample_phase_kernel<SMP, 3, true> : &resample_phase_kernel<SMP, 3, false>;
                case 4: return vec ? &resample_phase_kernel<SMP, 4, true> : &resample_phase_kernel<SMP, 4, false>;
                case 5: return vec ? &resample_phase_kernel<SMP, 5, true> : &resample_phase_kernel<SMP, 5, false>;
                case 7: return vec ? &resample_phase_kernel<SMP, 7, true> : &resample_phase_kernel<SMP, 7, false>;
                default: return nullptr;
            }
        };
        const phase_fn kfn = pcm ? pick(int16_t{}) : pick(float{});
        if (kfn == nullptr) { set_error("RESAMPLE: no kernel for runs of %d outputs", plan.rs_q); return MMFE_EINVAL; }
        int per_cu = 1, cus = 256;
        const int rc = kernel_prepare(reinterpret_cast<const void*>(kfn), plan.rs_threads, lds, &per_cu, &cus);
        if (rc) return rc;
        const int64_t n_tiles = n_rows * a.tiles_per_row;
        if (n_tiles >= ((int64_t)1 << 31) - 65536 || n_out + a.tile_out >= (int64_t)1 << 31) { set_error("RESAMPLE: batch too large for one launch"); return MMFE_EINVAL; }
        const int grid = (int)std::min<int64_t>(n_tiles, (int64_t)cus * std::max(1, per_cu));
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(plan.rs_threads), lds, stream, a);
        MMFE_HIP_CHECK(hipGetLastError());
        return MMFE_OK;
    }
    const int half = (n_taps - 1) / 2;
    const int pre_pad = down - half % down;
    ResampleArgs a{};
    a.src = src;
    a.out = out;
    a.n_rows = (int)n_rows;
    a.n_in = (int)n_samples;
    a.n_out = n_out;
    a.out_stride = out_len;
    a.up = up;
    a.down = down;
    a.taps = (pre_pad + n_taps + up - 1) / up;
    a.taps_padded = (a.taps + 3) & ~3;
    a.pre = (half + pre_pad) / down;
    // largest tile (a multiple of the block size, at most 16 outputs per thread) whose staged input fits
    int per_thread = 16;
    auto tile_in_of = [&](int tile_out) { return (int)(((int64_t)tile_out * down + up - 1) / up) + a.taps_padded + 8; };
    while (per_thread > 1 && tile_in_of(per_thread * kThreads) > kMaxTileIn) per_thread /= 2;
    a.tile_out = per_thread * kThreads;
    a.tile_in = tile_in_of(a.tile_out);
    a.tiles_per_row = (int)((n_out + a.tile_out - 1) / a.tile_out);
    a.bank = plan.d_window;
    const size_t bank_bytes = (size_t)up * a.taps_padded * 4;
    const bool bank_lds = bank_bytes <= (size_t)kMaxBankBytes;
    const size_t lds = (bank_lds ? bank_bytes : 0) + (size_t)a.tile_in * 4;
    const void* fn = pcm ? (bank_lds ? reinterpret_cast<const void*>(&resample_kernel<int16_t, true>) : reinterpret_cast<const void*>(&resample_kernel<int16_t, false>))
                         : (bank_lds ? reinterpret_cast<const void*>(&resample_kernel<float, true>) : reinterpret_cast<const void*>(&resample_kernel<float, false>));
    int per_cu = 1, cus = 256;
    const int rc = kernel_prepare(fn, kThreads, lds, &per_cu, &cus);
    if (rc) return rc;
    const int64_t n_tiles = n_rows * a.tiles_per_row;
    const int grid = (int)std::min<int64_t>(n_tiles, (int64_t)cus * std::max(1, per_cu));
    if (pcm) {
        if (bank_lds) hipLaunchKernelGGL((resample_kernel<int16_t, true>), dim3(grid), dim3(kThreads), lds, stream, a);
        else hipLaunchKernelGGL((resample_kernel<int16_t, false>), dim3(grid), dim3(kThreads), lds, stream, a);
    } else {
        if (bank_lds) hipLaunchKernelGGL((resample_kernel<float, true>), dim3(grid), dim3(kThreads), lds, stream, a);
        else hipLaunchKernelGGL((resample_kernel<float, false>), dim3(grid), dim3(kThreads), lds, stream, a);
    }
    MMFE_HIP_CHECK(hipGetLastError());
    return MMFE_OK;
}

}  // namespace mmfe
