// Fused kaldi-style log-mel filterbank for n_fft = 512 on gfx950.
//
// Stands behind torchaudio.compliance.kaldi.fbank as the reference calls it
// (classification/vggsound/dataset/vggsound.py:45-47) plus the pad/cut/normalise of
// vggsound.py:48-59: per frame  DC removal -> pre-emphasis -> window -> zero-pad to 512 ->
// real FFT -> |X|^2 -> banded mel -> ln(max(.,floor)) -> (x+shift)*scale, stored time-major.
//
// Mapping: 16 lanes own one frame, so a wave64 carries 4 consecutive frames of one clip through the
// whole pipeline.
//   * the 512-point real FFT is a 256-point complex FFT of z[n] = y[2n] + i*y[2n+1], split 16 x 16:
//     lane b runs a register-resident 16-point FFT over z[16a+b], twiddles by W256^(b*c), the 16
//     lanes transpose through a padded LDS tile, lane c runs the second 16-point FFT and holds
//     Z[c+16d];
//   * Z goes back to LDS in natural order, each lane picks up the pairs (k, 256-k) it needs for the
//     real-FFT recombination, shares E = Z[k]+conj(Z[256-k]) and D = Z[k]-conj(Z[256-k]) between the
//     two bins, and leaves |X|^2 in LDS;
//   * banded mel projection, two forms:
//       - matrix-pipe form: v_mfma_f32_4x4x1 runs 16 independent 4x4 outer-product blocks; block
//         `blk` accumulates (4 frames of the wave) x (filters 4*blk..4*blk+3) over ITS OWN run of
//         bins, one bin per instruction, weights resident in VGPRs.  Two instruction streams cover
//         128 filters.  The f32 MFMA is a k-ordered fmaf chain, so the numerics equal the scalar
//         form; it only moves the banded projection off the VALU and off the LDS latency chain.
//       - VALU form (any filter bank): 16 filters at a time, one per lane, zero-padded tap tables.
// HBM traffic is the algorithmic minimum: every kept sample is read once (8-byte lane loads that
// cover each frame's 128-byte lines exactly once) and every output float is written once.
#include "mmfe_internal.h"
#include "fft_regs.h"
#include "log_affine.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

// Contraction is explicit in this file (fmaf where a fused multiply-add is meant), see fft_regs.h.
#pragma clang fp contract(off)

namespace mmfe {

namespace {

constexpr int kFrameLds = 2176;              // bytes of LDS work space per frame: the 16x16 complex transpose tile with a 16-byte skew per row pair
                                             // (270 entries of 8 bytes), rounded up so that consecutive tiles start 32 banks apart
constexpr int kFrameFloats = kFrameLds / 4;  // 544
constexpr int kPGuard = 8;                   // floats in front of the power spectrum (bin runs may start below 0)
#ifndef MMFE_FBANK_WAVES
#define MMFE_FBANK_WAVES 4
#endif
#ifndef MMFE_FBANK_MIN_WAVES_PER_SIMD
#define MMFE_FBANK_MIN_WAVES_PER_SIMD 3      // 3 blocks of 4 waves per CU, up to 168 VGPRs per lane: room for a prefetch that is requested a whole
                                             // iteration ahead.  4 (128 VGPRs) runs the FFT phases faster but has to request the prefetch half an
                                             // iteration later; measured 1.5-2 % slower overall, and with the early prefetch it spills
#endif
// output stores: the feature tensor is far larger than the caches, so it can bypass them (1) or not (0)
#ifndef MMFE_FBANK_NTLOAD
#define MMFE_FBANK_NTLOAD 1      // samples are read once: non-temporal loads (16 interleaved rounds of 30 launches: 0.668 against 0.673-0.675 ms)
#endif
#ifndef MMFE_FBANK_NT
#define MMFE_FBANK_NT 1
#endif
typedef float mmfe_f32x4 __attribute__((ext_vector_type(4)));
#if MMFE_FBANK_NT
#define MMFE_OUT_STORE(v, ptr) __builtin_nontemporal_store((v), (ptr))
#define MMFE_OUT_STORE4(a, b, c, d, ptr) __builtin_nontemporal_store(mmfe_f32x4{(a), (b), (c), (d)}, reinterpret_cast<mmfe_f32x4*>(ptr))
#else
#define MMFE_OUT_STORE(v, ptr) (*(ptr) = (v))
#define MMFE_OUT_STORE4(a, b, c, d, ptr) (*reinterpret_cast<mmfe_f32x4*>(ptr) = mmfe_f32x4{(a), (b), (c), (d)})
#endif

// LDS instruction selection is part of this kernel's design (cycles of the LDS array per wave instruction on gfx950:
// ds_read_b64 2, ds_read_b128 4, but ds_read2_b64 8 and ds_read2_b32 4; ds_write_b64 6, ds_write2_b64 13), so every
// LDS and global access below is written with the width it is to be issued with, the kernel opts out of the
// backend's pairing of DS instructions (target attribute "no-load-store-opt") and this file is compiled with the IR
// load/store vectoriser off (-mllvm -amdgpu-load-store-vectorizer=0, see the Makefile): left on, it fuses neighbouring
// 8-byte LDS reads of 8-byte alignment into ds_read2_b64.
#ifdef __HIP_DEVICE_COMPILE__
#define MMFE_NO_DS_PAIRING __attribute__((target("no-load-store-opt")))
#else
#define MMFE_NO_DS_PAIRING
#endif
#ifndef MMFE_PREFETCH_EARLY
#define MMFE_PREFETCH_EARLY 1  // 1: the next quad's samples are requested as soon as this quad's have been windowed (a whole iteration ahead)
#endif
#ifndef MMFE_PREFETCH_DEPTH
#define MMFE_PREFETCH_DEPTH 1  // 2 (whole-quad variant only): TWO quads' samples in flight per wave in two register sets that swap roles every
#endif                         // iteration (the loop body is instantiated twice); needs MMFE_PREFETCH_EARLY 1
#ifndef MMFE_MEL_PAIR
#define MMFE_MEL_PAIR 1       // matrix-pipe mel: every bin run starts on an even bin, so a lane fetches the spectrum operands of two
#endif                        // steps with one 8-byte LDS read (18 ds_read_b64 = 36 LDS cycles against 36 ds_read_b32 = 72)
#ifndef MMFE_MEL_ACC_A
#define MMFE_MEL_ACC_A 1
#endif
#ifndef MMFE_MEL_ACC_B
#define MMFE_MEL_ACC_B 1
#endif
constexpr int kWavesPerBlock = MMFE_FBANK_WAVES;
constexpr int kDefaultChunk = 1;
#ifndef MMFE_PREFETCH_SPLIT
#define MMFE_PREFETCH_SPLIT 6
#endif
constexpr int kPrefetchSplit = MMFE_PREFETCH_SPLIT;
constexpr int kTwPostEntries = 144;          // -i*W512^k for k = 0..143 (the kernel uses the first 128)
constexpr int kMaxPass = MelPassTable::kMaxPass;
constexpr int kTw4Entries = 8 * 16, kPost4Entries = 4 * 16, kWin4Entries = 8 * 16;     // 16-byte entries of the LDS tables
constexpr int kTableBytes = 16 * (kTw4Entries + kPost4Entries + kWin4Entries);         // 5120
constexpr int kFmStride = 20;                // floats per filter row of the feature-major output tile: 16 frames + 4 (a quarter-odd stride: 16-byte
                                             // accesses of 16 consecutive rows fall on distinct banks)
constexpr int kFmTileBytes = 128 * kFmStride * 4 + 64;   // tile + per-wave (row, first frame, valid frames)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// Explicit-width accesses (see the note on LDS instruction selection above).
__device__ __forceinline__ float2 ld8(const void* p) {
    const f32x2 v = *static_cast<const f32x2*>(p);
    return make_float2(v.x, v.y);
}
__device__ __forceinline__ f32x4 ld16(const void* p) { return *static_cast<const f32x4*>(p); }
__device__ __forceinline__ void st8(void* p, float2 v) { *static_cast<f32x2*>(p) = f32x2{v.x, v.y}; }

struct FbankArgs {
    RowSource src;
    void* out;                 // elements of out_dtype
    int out_dtype;             // MMFE_OUT_*
    int n_rows;
    int n_samples;         // samples per row (ragged batches: of the longest row)
    int frames_valid;      // frames computed per row (<= natural frame count, <= out_rows); ragged batches: upper limit
    int out_rows;          // frames per row in the output tensor
    int hop;
    int win_len;
    int n_mel;
    int vec_out;               // n_mel % 4 == 0 and a 16-byte aligned output tensor: 16-byte stores
    int chunk;                 // consecutive quads a wave takes per visit (>= 1)
    int feat_major;            // output laid out (row, filter, frame) instead of (row, frame, filter)
    int fm_tile;               // feature-major: the four waves of a block pool a round's outputs in LDS and store 64-byte pieces of every filter row
    int vec_out_fm;            // feature-major: a lane's 4 frames of one filter may go out as one 16-byte (8-byte for 16-bit types) store
    float preemph;
    float mean_scale;          // 1/win_len when the per-frame mean is removed, else 0
    LogAffine la;
    const float* window;       // 512 floats, pre-halved, zero beyond win_len
    const float2* tw_stage;    // 256: W256^n
    const float2* tw_post;     // 144: -i*W512^k
    // VALU mel form
    int n_pass;
    int mel_rows;              // total tap rows (of 16 weights)
    const int32_t* mel_start;  // n_pass*16
    const int32_t* mel_niter;  // n_pass
    const int32_t* mel_woff;   // n_pass
    const float* mel_w;        // mel_rows*16
    // matrix-pipe mel form
    const int32_t* mel4_kstart;  // 2*64: first bin of the run of (set, lane)
    const float* mel4_w;         // (4*NGA + 4*NGB) * 64 weights, step-major
    unsigned long long* stamps;  // diagnostic build only
};

// Input element types.  A lane holds its 2-sample slices as they come from memory: two floats, or one 32-bit word
// with two 16-bit PCM samples (half the prefetch registers).  PCM samples enter the pipeline as their integer values;
// the 1/32768 of the decode is folded into the window table, which is exact (every stage before the window is
// linear and a power-of-two scale commutes with float rounding), so the results equal those of decoded floats.
template <bool PCM> struct Input;
template <> struct Input<false> {
    typedef float elem;
    typedef float2 slice;
    static __device__ __forceinline__ float2 unpack(float2 v) { return v; }
    static __device__ __forceinline__ float2 perturb(float2 v, int nt) { return make_float2(v.y * 1.0001f, v.x + 1e-3f * (float)nt); }
};
template <> struct Input<true> {
    typedef int16_t elem;
    typedef uint32_t slice;
    static __device__ __forceinline__ float2 unpack(uint32_t v) {
        return make_float2((float)(int16_t)(v & 0xffffu), (float)((int32_t)v >> 16));
    }
    static __device__ __forceinline__ uint32_t perturb(uint32_t v, int nt) { return v * 2654435761u + (uint32_t)nt; }
};

// Loads the 2-sample slices (32a+2b, 32a+2b+1), a = 0..NA-1, of one frame.  Branch-free, so the
// surrounding loop stays one basic block and the loads can be scheduled freely: a frame slot beyond
// the row's last frame reads row 0 instead (`safe`; its results are never stored), and lanes of a
// slice that sits in the zero padding read `safe` and select zero.  EXACT promises
// 32*(NA-1) < win_len <= 32*NA, so only the last slice needs the per-lane test and all other loads
// are one base register plus an immediate offset.  The VEC2 form needs an even win_len.
template <int NA, bool VEC2, bool EXACT, int A0 = 0, int A1 = NA>
__device__ __forceinline__ void load_frame(float2 (&v)[NA], const float* fp, const float* safe, bool valid, int b, int win_len) {
    const float* base = (valid ? fp : safe) + 2 * b;
#pragma unroll
    for (int a = A0; a < A1; ++a) {
        const int idx = 32 * a + 2 * b;
        const bool whole = EXACT && a < NA - 1;        // compile-time: slice entirely inside the window
        if (VEC2) {
            const bool ok = whole || idx + 1 < win_len;
#if MMFE_FBANK_NTLOAD
            const f32x2 rr = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(ok ? base + 32 * a : safe));
            const float2 r = make_float2(rr.x, rr.y);
#else
            const float2 r = ld8(ok ? base + 32 * a : safe);
#endif
            v[a] = whole ? r : make_float2(ok ? r.x : 0.f, ok ? r.y : 0.f);
        } else {
            const bool ok0 = whole || idx < win_len, ok1 = whole || idx + 1 < win_len;
            const float r0 = *(ok0 ? base + 32 * a : safe);
            const float r1 = *(ok1 ? base + 32 * a + 1 : safe);
            v[a] = make_float2(ok0 ? r0 : 0.f, ok1 ? r1 : 0.f);
        }
    }
}

// The same for 16-bit PCM: one 4-byte load per slice (VEC2: needs 4-byte aligned frame starts) or two 2-byte loads.
template <int NA, bool VEC2, bool EXACT, int A0 = 0, int A1 = NA>
__device__ __forceinline__ void load_frame(uint32_t (&v)[NA], const int16_t* fp, const int16_t* safe, bool valid, int b, int win_len) {
    const int16_t* base = (valid ? fp : safe) + 2 * b;
#pragma unroll
    for (int a = A0; a < A1; ++a) {
        const int idx = 32 * a + 2 * b;
        const bool whole = EXACT && a < NA - 1;
        if (VEC2) {
            const bool ok = whole || idx + 1 < win_len;
            const uint32_t r = *reinterpret_cast<const uint32_t*>(ok ? base + 32 * a : safe);
            v[a] = whole ? r : (ok ? r : 0u);
        } else {
            const bool ok0 = whole || idx < win_len, ok1 = whole || idx + 1 < win_len;
            const uint32_t r0 = *reinterpret_cast<const uint16_t*>(ok0 ? base + 32 * a : safe);
            const uint32_t r1 = *reinterpret_cast<const uint16_t*>(ok1 ? base + 32 * a + 1 : safe);
            v[a] = (ok0 ? r0 : 0u) | (ok1 ? r1 : 0u) << 16;
        }
    }
}

// How the rows of a launch end (template parameter MODE of the kernel):
//   kRowsWhole   every row has the same number of frames, a multiple of 4, and the output is float32 with whole 16-byte
//                filter blocks: every lane of every quad stores, unconditionally.  This matters beyond the saved compares:
//                loads and stores share one in-order counter (vmcnt), so a wave can wait for "my prefetched samples, but
//                not the stores I issued after them" only if the number of those stores is the same on every path.
//   kRowsAny     every row has the same number of frames; stores are predicated
//   kRowsRagged  per-row lengths (FbankArgs::src.row_len)
constexpr int kRowsWhole = 0, kRowsAny = 1, kRowsRagged = 2;

// Frames computed for row r (wave-uniform): the launch-wide count, or in a ragged batch what the row's own length gives.
template <bool RAGGED>
__device__ __forceinline__ int frames_of(const FbankArgs& p, unsigned r) {
    if (!RAGGED) return p.frames_valid;
    if (r >= (unsigned)p.n_rows) return 0;
    const int n = row_samples(p.src, r, p.n_samples);
    const int nat = n < p.win_len ? 0 : 1 + (n - p.win_len) / p.hop;
    return nat < p.frames_valid ? nat : p.frames_valid;
}

// Diagnostic build only (-DMMFE_STAMPS): per-phase s_memtime shares of the main loop, printed by the launcher.
#ifdef MMFE_STAMPS
#define MMFE_STAMP(i)                                                                                   \
    do {                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        unsigned long long _t;                                                                          \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");                      \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        if ((i) > 0) st_acc[(i) - 1] += _t - st_prev;                                                   \
        st_prev = _t;                                                                                   \
    } while (0)
#else
#define MMFE_STAMP(i) do { } while (0)
#endif

template <bool PCM, int NA /* ceil(win_len/32) */, bool VEC2, bool EXACT, int NGA, int NGB /* groups of 4 bins per mel set; NGA == 0: VALU mel */,
          int MODE /* kRowsWhole, kRowsAny or kRowsRagged */>
__global__ void __launch_bounds__(64 * kWavesPerBlock, MMFE_FBANK_MIN_WAVES_PER_SIMD) MMFE_NO_DS_PAIRING fbank512_kernel(const FbankArgs p) {
    typedef Input<PCM> In;
    typedef typename In::elem Elem;
    constexpr bool RAGGED = MODE == kRowsRagged, WHOLE = MODE == kRowsWhole;
    constexpr bool kMfmaMel = NGA > 0;
    constexpr int kStepsA = 4 * NGA, kStepsB = 4 * NGB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // constant tables, laid out so that each lane fetches two entries with one 16-byte read (the 16 lanes of a frame read 256
    // contiguous bytes, and the four frame groups read the same addresses):
    f32x4* s_tw4 = reinterpret_cast<f32x4*>(smem);        // [8][16]: (W256^(b*2j), W256^(b*(2j+1))), the twiddles between the FFT halves
    f32x4* s_post4 = s_tw4 + kTw4Entries;                 // [4][16]: (-i*W512^(b+32j), -i*W512^(b+32j+16)), the real-FFT recombination
    f32x4* s_win4 = s_post4 + kPost4Entries;              // [8][16]: window[64j+2b], [64j+2b+1], [64j+32+2b], [64j+32+2b+1]
    int32_t* s_mel_start = reinterpret_cast<int32_t*>(s_win4 + kWin4Entries);  // VALU mel only: kMaxPass*16
    float* s_mel_w = reinterpret_cast<float*>(s_mel_start + (kMfmaMel ? 0 : kMaxPass * 16));   // VALU mel only
    char* s_work = reinterpret_cast<char*>(s_mel_w + (kMfmaMel ? 0 : p.mel_rows * 16));

    for (int i = threadIdx.x; i < kTw4Entries; i += blockDim.x) {
        const int j = i >> 4, bb = i & 15;
        const float2 t0 = p.tw_stage[bb * (2 * j)], t1 = p.tw_stage[bb * (2 * j + 1)];
        s_tw4[i] = f32x4{t0.x, t0.y, t1.x, t1.y};
    }
    for (int i = threadIdx.x; i < kPost4Entries; i += blockDim.x) {
        const int j = i >> 4, bb = i & 15;
        const float2 t0 = p.tw_post[bb + 32 * j], t1 = p.tw_post[bb + 32 * j + 16];
        s_post4[i] = f32x4{t0.x, t0.y, t1.x, t1.y};
    }
    for (int i = threadIdx.x; i < kWin4Entries; i += blockDim.x) {
        const int j = i >> 4, bb = i & 15;
        s_win4[i] = f32x4{p.window[64 * j + 2 * bb], p.window[64 * j + 2 * bb + 1], p.window[64 * j + 32 + 2 * bb], p.window[64 * j + 32 + 2 * bb + 1]};
    }
    if (!kMfmaMel) {
        for (int i = threadIdx.x; i < p.n_pass * 16; i += blockDim.x) s_mel_start[i] = p.mel_start[i];
        for (int i = threadIdx.x; i < p.mel_rows * 16; i += blockDim.x) s_mel_w[i] = p.mel_w[i];
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fl = lane >> 4;      // frame slot within the wave
    const int b = lane & 15;       // lane within the frame group
    char* wave_work = s_work + wave * 4 * kFrameLds;
    float2* fbuf = reinterpret_cast<float2*>(wave_work + fl * kFrameLds);
    // power spectrum of frame slot f lives at float offset kPGuard + 8*f of its tile: the 8*f skew keeps
    // same-index 4-byte accesses of different frame slots on different banks
    float* pbuf = reinterpret_cast<float*>(fbuf) + kPGuard + 8 * fl;

    // matrix-pipe mel: lane = 4*blk + q supplies the weight of filter (64*set + lane) at bin run(blk) + s (operand A,
    // row q of the block) and P[frame slot q][run(blk) + s] (operand B, column q); weights stay in registers for
    // the life of the kernel
    float wreg[kMfmaMel ? kStepsA + kStepsB : 1];
    const float* pa_A = nullptr;
    const float* pa_B = nullptr;
    if (kMfmaMel) {
#pragma unroll
        for (int s = 0; s < kStepsA + kStepsB; ++s) wreg[s] = p.mel4_w[s * 64 + lane];
        const float* slot = reinterpret_cast<const float*>(wave_work + (lane & 3) * kFrameLds) + kPGuard + 8 * (lane & 3);
        pa_A = slot + p.mel4_kstart[lane];
        pa_B = slot + p.mel4_kstart[64 + lane];
    }

    // work distribution: quad = 4 consecutive frames of one row.  The quads are cut into runs of p.chunk consecutive quads
    // and the runs dealt round-robin to the grid's waves (the waves of a block take neighbouring runs), so that a wave
    // streams p.chunk * 4 frames of contiguous input and output per visit and a block one contiguous region.
    const unsigned qpr = (unsigned)(p.frames_valid + 3) >> 2;      // quads per row
    const unsigned chunk = (unsigned)p.chunk;
    const unsigned jump = (gridDim.x * kWavesPerBlock - 1) * chunk + 1;        // from the last quad of a run to the first of the wave's next
    const unsigned d_row = jump / qpr, d_q = jump - d_row * qpr;
    const unsigned quad0 = (blockIdx.x * kWavesPerBlock + wave) * chunk;
    unsigned row = quad0 / qpr, q = quad0 - row * qpr;              // wave-uniform
    unsigned in_run = 0;
    auto advance = [&](unsigned& r, unsigned& qq) {
        in_run += 1;
        const bool last = in_run == chunk;
        if (last) in_run = 0;
        r += last ? d_row : 0u;
        qq += last ? d_q : 1u;
        if (qq >= qpr) { qq -= qpr; r += 1; }
    };

    // where masked lanes read: the first win_len samples of row 0, or (ragged batches: row 0 may be shorter than a window) of the buffer
    const Elem* safe = RAGGED ? static_cast<const Elem*>(p.src.base) : row_ptr_as<Elem>(p.src, 0);
    int fv = frames_of<RAGGED>(p, row);                                       // of the quad whose samples are in `nxt`
    float none_flag = p.la.none_flag;
    asm volatile("" : "+v"(none_flag));         // a per-lane value from here on: selects on it stay v_cndmask
    constexpr bool DEPTH2 = MMFE_PREFETCH_DEPTH == 2 && MODE == kRowsWhole && MMFE_PREFETCH_EARLY == 1;
    typename In::slice nxt[NA];
    typename In::slice nx2[DEPTH2 ? NA : 1];
    // Quads whose samples have been requested but not yet transformed, oldest first: (row_h, q_h, fv_h) is the one the next
    // iteration works on; with two quads in flight (row_m, q_m, fv_m) follows it; (row, q, fv) is the most recent request.
    unsigned row_h = row, q_h = q, row_m = 0, q_m = 0;
    int fv_h = fv, fv_m = 0;
    {
        const int t = (int)q * 4 + fl;
        const bool valid = row < (unsigned)p.n_rows && t < fv;
        const Elem* fp = row_ptr_as<Elem>(p.src, valid ? row : 0) + (int64_t)t * p.hop;
        load_frame<NA, VEC2, EXACT>(nxt, fp, safe, valid, b, p.win_len);
    }
    // The lane's share of the frame sum (DC removal) is taken at the END of an iteration, after the output stores have been
    // issued: it is the first use of the prefetched samples, so the wait for them stands behind the stores in program order
    // and can leave exactly those outstanding.  At the top of the loop the same wait would also have to cover the path
    // from the prologue, i.e. drain every store of the previous iteration before any arithmetic starts.
    auto lane_sum = [&](const typename In::slice (&nxt)[NA]) {
        if constexpr (PCM) {
            // 16-bit samples: the sum of a lane's <= 32 samples stays below 2^24, so integer addition gives exactly the float sum
            // (every float partial sum is exact too) and the samples are converted once, where they are unpacked
            // (v_dot2c_i32_i16 with the constant pair (1, 1) adds both halves of a word to the running sum in one instruction)
            typedef short i16x2 __attribute__((ext_vector_type(2)));
            int s = 0;
#pragma unroll
            for (int a = 0; a < NA; ++a) s = __builtin_amdgcn_sdot2(__builtin_bit_cast(i16x2, nxt[a]), i16x2{1, 1}, s, false);
            return (float)s;
        } else {
            float s = 0.f;
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                const float2 v = In::unpack(nxt[a]);
                s += v.x + v.y;
            }
            return s;
        }
    };
    float dc_part = lane_sum(nxt);

#ifdef MMFE_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev = 0;
#endif
    // Feature-major tile mode (never in the whole-quad variant): the waves of a block take neighbouring quads (chunk 1) and meet at two
    // LDS barriers per round, so every wave runs the block's number of rounds; a wave whose quad lies beyond the batch computes on
    // row 0's samples like any masked frame slot and contributes nothing.
    unsigned rounds_left = 0;
    float* fm_tile = nullptr;
    int* fm_meta = nullptr;
    bool tile_mode = false;
    if constexpr (!WHOLE) {
        tile_mode = p.fm_tile != 0;
        if (tile_mode) {
            fm_tile = reinterpret_cast<float*>(s_work + kWavesPerBlock * 4 * kFrameLds);
            fm_meta = reinterpret_cast<int*>(fm_tile + 128 * kFmStride);
            const unsigned total = (unsigned)p.n_rows * qpr, first = blockIdx.x * kWavesPerBlock, stride = gridDim.x * kWavesPerBlock;
            rounds_left = first < total ? (total - first + stride - 1) / stride : 0;
        }
    }
    if constexpr (DEPTH2) {                 // second request: the quad after the first one
        advance(row, q);
        fv = frames_of<RAGGED>(p, row);
        const int t = (int)q * 4 + fl;
        const bool valid = row < (unsigned)p.n_rows && t < fv;
        const Elem* fp = row_ptr_as<Elem>(p.src, row < (unsigned)p.n_rows ? row : 0) + (int64_t)t * p.hop;
        load_frame<NA, VEC2, EXACT>(nx2, fp, safe, valid, b, p.win_len);
        row_m = row; q_m = q; fv_m = fv;
    }
    // One iteration: transforms the quad whose samples are in `cur_s`, requests the quad after the most recent request into the same
    // registers (free once the samples have been windowed), and ends by taking the first look at `nxt_s`, the samples of the quad the
    // next iteration works on (one register set: cur_s and nxt_s are the same array).
    auto iteration = [&](typename In::slice (&nxt)[NA], const typename In::slice (&nxt_s)[NA]) __attribute__((always_inline)) {
        if (tile_mode) --rounds_left;
        MMFE_STAMP(0);
        const int t0 = (int)q_h * 4;        // first frame of the quad
        const unsigned cur_row = row_h;
        const int cur_fv = fv_h;            // frames of this row that exist

        float2 g[NA];
#pragma unroll
        for (int a = 0; a < NA; ++a) g[a] = In::unpack(nxt[a]);

        // ---- per-frame DC removal (kaldi remove_dc_offset) ------------------------------------
        {
            const float mean = row16_sum(dc_part) * p.mean_scale;       // mean_scale = 0 switches the removal off
#pragma unroll
            for (int a = 0; a < NA; ++a) { g[a].x -= mean; g[a].y -= mean; }
        }

        // ---- pre-emphasis y[j] = g[j] - c*g[max(j-1,0)], then window.  The window table is zero
        //      beyond win_len, which also clears the lanes that sit in the zero padding. ---------
        float2 y[16];
        {
            const float c = p.preemph;
            f32x4 w4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                // the sample before g[a].x lives in lane b-1 (.y, same a); for lane 0 it is lane 15's
                // .y of slice a-1; the very first sample of the frame uses itself (replicate padding)
                float edge;
                if (a == 0) edge = g[0].x;
                else edge = dpp_mov<DPP_ROW_ROR1>(g[a - 1 < 0 ? 0 : a - 1].y);
                const float prev = dpp_upd<DPP_ROW_SHR1>(edge, g[a].y);
                const float y0 = fmaf(-c, prev, g[a].x);
                const float y1 = fmaf(-c, g[a].x, g[a].y);
                if ((a & 1) == 0) w4 = ld16(s_win4 + (a >> 1) * 16 + b);
                y[a] = (a & 1) == 0 ? make_float2(y0 * w4.x, y1 * w4.y) : make_float2(y0 * w4.z, y1 * w4.w);
            }
#pragma unroll
            for (int a = NA; a < 16; ++a) y[a] = make_float2(0.f, 0.f);
        }

        // prefetch of the wave's next quad: requested here (MMFE_PREFETCH_EARLY 1: a whole iteration ahead, 26 registers live
        // through both FFT halves), after the second FFT half (0: half an iteration ahead, where register pressure is
        // lowest), or the first kPrefetchSplit slices here and the rest there (2)
        const Elem* nfp = nullptr;
        bool nvalid = false;
        if (MMFE_PREFETCH_EARLY) {
            advance(row, q);
            const int nt = (int)q * 4 + fl;
            fv = frames_of<RAGGED>(p, row);
            nvalid = row < (unsigned)p.n_rows && nt < fv;
            nfp = row_ptr_as<Elem>(p.src, row < (unsigned)p.n_rows ? row : 0) + (int64_t)nt * p.hop;
#ifdef MMFE_ABL_NOLOAD
#pragma unroll
            for (int a = 0; a < NA; ++a) nxt[a] = In::perturb(nxt[a], nt);
#else
            load_frame<NA, VEC2, EXACT, 0, MMFE_PREFETCH_EARLY == 2 ? kPrefetchSplit : NA>(nxt, nfp, safe, nvalid, b, p.win_len);
#endif
        }
        // ---- 256-point complex FFT, first half: 16-point FFT over a, twiddle, transpose --------
        MMFE_STAMP(1);
        fft16(y);
        // Transpose tile: row c starts at float2 index 16c + 2(c>>1).  A lane writes 8 bytes per row (16 lanes: 128 contiguous
        // bytes) and reads its own row back with eight 16-byte reads; the 2-entry skew per row pair spreads the 16 rows that one
        // 16-byte read touches (lanes {0-3, 12-15} of one frame group with lanes {4-11} of the next) over all 64 banks.
        {
            f32x4 tw4 = f32x4{1.f, 0.f, 1.f, 0.f};
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                float2 val = y[fft16_pos(c)];
                if ((c & 1) == 0) tw4 = ld16(s_tw4 + (c >> 1) * 16 + b);
                if (c > 0) val = cmul(val, (c & 1) == 0 ? make_float2(tw4.x, tw4.y) : make_float2(tw4.z, tw4.w));
                st8(fbuf + c * 16 + 2 * (c >> 1) + b, val);
            }
        }
        wave_lds_fence();
        MMFE_STAMP(2);
        float2 u[16];
        {
            const float2* trow = fbuf + 16 * b + 2 * (b >> 1);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x4 v = ld16(trow + 2 * j);
                u[2 * j] = make_float2(v.x, v.y);
                u[2 * j + 1] = make_float2(v.z, v.w);
            }
        }
        wave_lds_fence();

        // ---- second half: lane c = b now owns Z[c + 16 d]; park Z in natural order --------------
        MMFE_STAMP(3);
        fft16(u);
        // Lane b recombines k = b + 16 i, i < 8: Z[k] is its own register i, and the partner Z[256 - k] is register 15 - i of lane
        // 16 - b (register 16 - i of lane 0 itself when b = 0), i.e. always one of the upper eight registers of some lane:
        // only those go through LDS.  The one exception, Z[256 - 0] = Z[0], is lane 0's own register 0.
#pragma unroll
        for (int d = 8; d < 16; ++d) st8(fbuf + b + 16 * d, u[fft16_pos(d)]);
        // bin 128 pairs with itself: X[128] = 2*conj(Z[128]) (window pre-halved), and Z[128] is lane 0's d = 8
        const float p128 = 4.0f * fmaf(u[fft16_pos(8)].x, u[fft16_pos(8)].x, u[fft16_pos(8)].y * u[fft16_pos(8)].y);
        wave_lds_fence();

        // ---- real-FFT recombination on pairs (k, 256-k), power spectrum ------------------------
        // prefetch the next quad of this wave here, where register pressure is lowest (both FFT halves are done
        // and only Z in LDS is live); the loads stay in flight under the rest of this iteration
        if (!MMFE_PREFETCH_EARLY) {
            advance(row, q);
            const int nt = (int)q * 4 + fl;
            fv = frames_of<RAGGED>(p, row);
            nvalid = row < (unsigned)p.n_rows && nt < fv;
            nfp = row_ptr_as<Elem>(p.src, row < (unsigned)p.n_rows ? row : 0) + (int64_t)nt * p.hop;
#ifdef MMFE_ABL_NOLOAD
#pragma unroll
            for (int a = 0; a < NA; ++a) nxt[a] = In::perturb(nxt[a], nt);
#endif
        }
#ifndef MMFE_ABL_NOLOAD
        if (MMFE_PREFETCH_EARLY != 1)
            load_frame<NA, VEC2, EXACT, MMFE_PREFETCH_EARLY == 2 ? kPrefetchSplit : 0, NA>(nxt, nfp, safe, nvalid, b, p.win_len);
#endif
        MMFE_STAMP(4);
        float pk[8], pq[8];
        f32x4 post4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = b + 16 * i;
            const float2 zk = u[fft16_pos(i)];
            float2 zp = ld8(fbuf + ((256 - k) & 255));
            if (i == 0) zp = make_float2(b == 0 ? zk.x : zp.x, b == 0 ? zk.y : zp.y);
            if ((i & 1) == 0) post4 = ld16(s_post4 + (i >> 1) * 16 + b);
            const float2 tw = (i & 1) == 0 ? make_float2(post4.x, post4.y) : make_float2(post4.z, post4.w);
            const float2 E = make_float2(zk.x + zp.x, zk.y - zp.y);
            const float2 D = make_float2(zk.x - zp.x, zk.y + zp.y);
            const float2 G = cmul(tw, D);
            const float2 xa = cadd(E, G), xb = csub(E, G);
            pk[i] = fmaf(xa.x, xa.x, xa.y * xa.y);      // |X[k]|^2
            pq[i] = fmaf(xb.x, xb.x, xb.y * xb.y);      // |X[256-k]|^2
        }
        wave_lds_fence();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = b + 16 * i;
            pbuf[k] = pk[i];
            pbuf[256 - k] = pq[i];
        }
        pbuf[b == 0 ? 128 : 272 + b] = p128;       // only lane 0 holds bin 128; the other lanes park their value in unused slots
        wave_lds_fence();

        MMFE_STAMP(5);
        // ---- banded mel projection, log, affine, store ----------------------------------------
        if (kMfmaMel) {
            // several independent accumulators per filter set: a dependent v_mfma chain pays the full MFMA
            // latency per step, independent ones issue back to back; the partial sums are added at the end
            constexpr int kAccA = MMFE_MEL_ACC_A, kAccB = MMFE_MEL_ACC_B;
            f32x4 pa[kAccA], pb[kAccB];
#pragma unroll
            for (int i = 0; i < kAccA; ++i) pa[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < kAccB; ++i) pb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            // Operand order decides which way round the 4x4 block comes out.  Weights as operand A (row i = filter 4*blk + i, taken
            // from lane 4*blk + i) and the spectrum as B (column q = frame slot q, from lane 4*blk + q) leave lane 4*blk + q
            // with 4 CONSECUTIVE FILTERS of ONE frame: the time-major store.  Swapped, lane 4*blk + j holds 4 CONSECUTIVE
            // FRAMES of ONE filter: the feature-major store.
            auto run_mel = [&](auto swapped) {
                constexpr bool FM = decltype(swapped)::value;
                auto step = [&](float w, float pv, f32x4 acc) {
                    return FM ? __builtin_amdgcn_mfma_f32_4x4x1f32(pv, w, acc, 0, 0, 0) : __builtin_amdgcn_mfma_f32_4x4x1f32(w, pv, acc, 0, 0, 0);
                };
#if MMFE_MEL_PAIR
#pragma unroll
                for (int s = 0; s < kStepsA; s += 2) {
                    const float2 pv = ld8(pa_A + s);
                    pa[s % kAccA] = step(wreg[s], pv.x, pa[s % kAccA]);
                    pa[(s + 1) % kAccA] = step(wreg[s + 1], pv.y, pa[(s + 1) % kAccA]);
                }
#pragma unroll
                for (int s = 0; s < kStepsB; s += 2) {
                    const float2 pv = ld8(pa_B + s);
                    pb[s % kAccB] = step(wreg[kStepsA + s], pv.x, pb[s % kAccB]);
                    pb[(s + 1) % kAccB] = step(wreg[kStepsA + s + 1], pv.y, pb[(s + 1) % kAccB]);
                }
#else
#pragma unroll
                for (int s = 0; s < kStepsA; ++s) pa[s % kAccA] = step(wreg[s], pa_A[s], pa[s % kAccA]);
#pragma unroll
                for (int s = 0; s < kStepsB; ++s) pb[s % kAccB] = step(wreg[kStepsA + s], pa_B[s], pb[s % kAccB]);
#endif
            };
            const bool fm = !WHOLE && p.feat_major != 0;        // wave-uniform
            if (fm) run_mel(std::true_type{});
            else run_mel(std::false_type{});
            f32x4 accA = pa[0], accB = pb[0];
#pragma unroll
            for (int i = 1; i < kAccA; ++i) accA += pa[i];
#pragma unroll
            for (int i = 1; i < kAccB; ++i) accB += pb[i];
            // time-major: one 16-byte store per filter set, and the 16 lanes that share a frame write 256 contiguous bytes of its row
            MMFE_STAMP(6);
            float vab[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) { vab[r] = accA[r]; vab[4 + r] = NGB > 0 ? accB[r] : 1.f; }
            apply_log_affine_block(vab, p.la);
            const int n_ok = cur_fv - t0;                 // wave-uniform; <= 0 for a quad past the end of a short row
            const int q4 = lane & 3, f0 = lane & ~3;      // frame slot, first filter of the lane's block
            const int64_t oidx = ((int64_t)cur_row * p.out_rows + t0 + q4) * p.n_mel + f0;
            if constexpr (WHOLE) {
                float* orow = static_cast<float*>(p.out) + oidx;
#ifdef MMFE_ABL_NOSTORE
                if (vab[0] == 123.456f && vab[5] == 654.321f)
#endif
                {
                    MMFE_OUT_STORE4(vab[0], vab[1], vab[2], vab[3], orow);
                    if (NGB > 0) MMFE_OUT_STORE4(vab[4], vab[5], vab[6], vab[7], orow + 64);
                }
            } else
#ifdef MMFE_ABL_NOSTORE
            if (q4 < n_ok && vab[0] == 123.456f && vab[5] == 654.321f) {
#else
            if (fm && tile_mode) {
                // lane holds filter m = lane (and 64 + m) for frames t0 .. t0+3.  The block's four quads are neighbours: pooled in LDS
                // they are 16 consecutive frames, 64 contiguous bytes, of every filter row
                const int nok = cur_row < (unsigned)p.n_rows ? (n_ok < 0 ? 0 : (n_ok > 4 ? 4 : n_ok)) : 0;
                *reinterpret_cast<f32x4*>(fm_tile + lane * kFmStride + 4 * wave) = f32x4{vab[0], vab[1], vab[2], vab[3]};
                if (NGB > 0) *reinterpret_cast<f32x4*>(fm_tile + (64 + lane) * kFmStride + 4 * wave) = f32x4{vab[4], vab[5], vab[6], vab[7]};
                if (lane == 0) { fm_meta[4 * wave] = (int)cur_row; fm_meta[4 * wave + 1] = t0; fm_meta[4 * wave + 2] = nok; }
                lds_barrier();
#pragma unroll
                for (int j = 0; j < (NGB > 0 ? 2 : 1); ++j) {
                    const int idx = (int)threadIdx.x + 256 * j;
                    const int m = idx >> 2, pc = idx & 3;              // filter row, piece = the wave that produced it
                    const int prow = fm_meta[4 * pc], pt0 = fm_meta[4 * pc + 1], pnok = fm_meta[4 * pc + 2];
                    if (m >= p.n_mel || pnok <= 0) continue;
                    const f32x4 v = ld16(fm_tile + m * kFmStride + 4 * pc);
                    const int64_t o = ((int64_t)prow * p.n_mel + m) * p.out_rows + pt0;
                    if (p.out_dtype == MMFE_OUT_F32) {
                        float* op = static_cast<float*>(p.out) + o;
                        // (whole 64-byte pieces: non-temporal like the time-major rows; cached stores measured 0.86 against 0.80 ms)
                        if (p.vec_out_fm && pnok >= 4) MMFE_OUT_STORE4(v[0], v[1], v[2], v[3], op);
                        else {
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                if (i < pnok) op[i] = v[i];
                        }
                    } else {
                        unsigned short* op = static_cast<unsigned short*>(p.out) + o;
                        if (p.vec_out_fm && pnok >= 4)
                            // (32-byte pieces: cached; non-temporal measured 1.00 against 0.83 ms)
                            *reinterpret_cast<u32x2*>(op) = u32x2{pack2_bits(v[0], v[1], p.out_dtype), pack2_bits(v[2], v[3], p.out_dtype)};
                        else {
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                if (i < pnok) op[i] = p.out_dtype == MMFE_OUT_F16 ? f16_bits(v[i]) : bf16_bits(v[i]);
                        }
                    }
                }
                lds_barrier();      // the tile is free again
            } else if (fm) {
                // lane holds filter m (and 64 + m) for frames t0 .. t0+3; a wave that takes `chunk` consecutive quads fills
                // 16 * chunk contiguous bytes of each filter row over its visit
                const int m = lane;
                const int64_t fidx = ((int64_t)cur_row * p.n_mel + m) * p.out_rows + t0;
#pragma unroll
                for (int h = 0; h < (NGB > 0 ? 2 : 1); ++h) {
                    if (64 * h + m >= p.n_mel || n_ok <= 0) continue;
                    const int64_t o = fidx + (int64_t)64 * h * p.out_rows;
                    const float* v = vab + 4 * h;
                    if (p.out_dtype == MMFE_OUT_F32) {
                        float* op = static_cast<float*>(p.out) + o;
                        // cached, not non-temporal: a 16-byte piece is a fraction of a line, and the line is only whole once the quads next
                        // to this one have stored theirs; pieces that bypass the cache reach HBM as partial writes (4x the launch time)
                        if (p.vec_out_fm && n_ok >= 4) *reinterpret_cast<mmfe_f32x4*>(op) = mmfe_f32x4{v[0], v[1], v[2], v[3]};
                        else {
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                if (i < n_ok) op[i] = v[i];
                        }
                    } else {
                        unsigned short* op = static_cast<unsigned short*>(p.out) + o;
                        if (p.vec_out_fm && n_ok >= 4)
                            *reinterpret_cast<u32x2*>(op) = u32x2{pack2_bits(v[0], v[1], p.out_dtype), pack2_bits(v[2], v[3], p.out_dtype)};
                        else {
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                if (i < n_ok) op[i] = p.out_dtype == MMFE_OUT_F16 ? f16_bits(v[i]) : bf16_bits(v[i]);
                        }
                    }
                }
            } else if (q4 < n_ok) {
#endif
                if (p.out_dtype == MMFE_OUT_F32) {
                    float* orow = static_cast<float*>(p.out) + oidx;
                    if (p.vec_out) {        // n_mel % 4 == 0 and an aligned tensor: blocks are whole
                        if (f0 < p.n_mel) MMFE_OUT_STORE4(vab[0], vab[1], vab[2], vab[3], orow);
                        if (NGB > 0 && 64 + f0 < p.n_mel) MMFE_OUT_STORE4(vab[4], vab[5], vab[6], vab[7], orow + 64);
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            if (f0 + i < p.n_mel) orow[i] = vab[i];
                            if (NGB > 0 && 64 + f0 + i < p.n_mel) orow[64 + i] = vab[4 + i];
                        }
                    }
                } else {                    // 16-bit output: four values = one 8-byte store
                    unsigned short* orow = static_cast<unsigned short*>(p.out) + oidx;
                    if (p.vec_out) {
                        if (f0 < p.n_mel)
                            *reinterpret_cast<u32x2*>(orow) = u32x2{pack2_bits(vab[0], vab[1], p.out_dtype), pack2_bits(vab[2], vab[3], p.out_dtype)};
                        if (NGB > 0 && 64 + f0 < p.n_mel)
                            *reinterpret_cast<u32x2*>(orow + 64) = u32x2{pack2_bits(vab[4], vab[5], p.out_dtype), pack2_bits(vab[6], vab[7], p.out_dtype)};
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            if (f0 + i < p.n_mel) orow[i] = p.out_dtype == MMFE_OUT_F16 ? f16_bits(vab[i]) : bf16_bits(vab[i]);
                            if (NGB > 0 && 64 + f0 + i < p.n_mel) orow[64 + i] = p.out_dtype == MMFE_OUT_F16 ? f16_bits(vab[4 + i]) : bf16_bits(vab[4 + i]);
                        }
                    }
                }
            }
        } else {
            const int t = t0 + fl;
            const bool valid = t < cur_fv;
            const int64_t oidx = ((int64_t)cur_row * p.out_rows + t) * p.n_mel;
            for (int j = 0; j < p.n_pass; ++j) {
                const int m = 16 * j + b;
                const float* pp = pbuf + s_mel_start[m];
                const float* wp = s_mel_w + p.mel_woff[j] * 16 + b;
                const int n = p.mel_niter[j];
                float acc = 0.f;
                for (int i = 0; i < n; ++i) acc = fmaf(wp[16 * i], pp[i], acc);
                const float val = apply_log_affine(acc, p.la, none_flag);
                if (valid && m < p.n_mel) {
                    const int64_t o = p.feat_major ? ((int64_t)cur_row * p.n_mel + m) * p.out_rows + t : oidx + m;
                    if (p.out_dtype == MMFE_OUT_F32) static_cast<float*>(p.out)[o] = val;
                    else static_cast<unsigned short*>(p.out)[o] = p.out_dtype == MMFE_OUT_F16 ? f16_bits(val) : bf16_bits(val);
                }
            }
        }
        MMFE_STAMP(7);
#ifdef MMFE_STAMPS
        st_acc[7] += 1;
#endif
        wave_lds_fence();
        __builtin_amdgcn_sched_barrier(0);      // the sum below stays behind the stores
        dc_part = lane_sum(nxt_s);
        if constexpr (DEPTH2) { row_h = row_m; q_h = q_m; fv_h = fv_m; row_m = row; q_m = q; fv_m = fv; }
        else { row_h = row; q_h = q; fv_h = fv; }
    };
    if constexpr (DEPTH2) {
        while (row_h < (unsigned)p.n_rows) {
            iteration(nxt, nx2);
            if (row_h >= (unsigned)p.n_rows) break;
            iteration(nx2, nxt);
        }
    } else {
        while (tile_mode ? rounds_left > 0 : row_h < (unsigned)p.n_rows) iteration(nxt, nxt);
    }
#ifdef MMFE_STAMPS
    if (lane == 0)
        for (int i = 0; i < 8; ++i) p.stamps[((size_t)blockIdx.x * kWavesPerBlock + wave) * 8 + i] = st_acc[i];
#endif
}

// `want` blocks cover the batch once; the grid is persistent, at most `per_cu` blocks on each CU of the current device.
template <bool PCM, int NA, bool VEC2, bool EXACT, int NGA, int NGB, int MODE>
int launch_variant(const FbankArgs& args, int64_t want, int per_cu, size_t lds, hipStream_t stream) {
    int resident = 0, cus = 0;      // raises the variant's dynamic-LDS limit on this device (once) and returns the cached CU count
    const int prc = kernel_prepare(reinterpret_cast<const void*>(&fbank512_kernel<PCM, NA, VEC2, EXACT, NGA, NGB, MODE>), 64 * kWavesPerBlock, lds,
                                   &resident, &cus);
    if (prc) return prc;
    const int grid = (int)std::min<int64_t>(want, (int64_t)cus * per_cu);
#ifdef MMFE_STAMPS
    FbankArgs targs = args;
    const size_t n_st = (size_t)grid * kWavesPerBlock * 8;
    MMFE_HIP_CHECK(hipMalloc((void**)&targs.stamps, n_st * 8));
    MMFE_HIP_CHECK(hipMemset(targs.stamps, 0, n_st * 8));
    hipLaunchKernelGGL((fbank512_kernel<PCM, NA, VEC2, EXACT, NGA, NGB, MODE>), dim3(grid), dim3(64 * kWavesPerBlock), lds, stream, targs);
    MMFE_HIP_CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(n_st);
    MMFE_HIP_CHECK(hipMemcpy(h.data(), targs.stamps, n_st * 8, hipMemcpyDeviceToHost));
    (void)hipFree(targs.stamps);
    static int printed = 0;
    if (printed++ < 2) {
        double sum[8] = {0};
        for (size_t w = 0; w < n_st / 8; ++w) for (int i = 0; i < 8; ++i) sum[i] += (double)h[w * 8 + i];
        const char* names[7] = {"load+preproc", "fft1+twiddle+Twrite", "prefetch+Tread", "fft2+Zwrite", "pairs+Pwrite", "mel(mfma)", "log+store"};
        double tot = 0; for (int i = 0; i < 7; ++i) tot += sum[i];
        for (int i = 0; i < 7; ++i) fprintf(stderr, "[stamps] %-22s %8.1f ticks/iter  %5.1f%%\n", names[i], sum[i] / sum[7], 100.0 * sum[i] / tot);
        fprintf(stderr, "[stamps] total %.1f ticks/iter, %.0f iterations\n", tot / sum[7], sum[7]);
    }
    return MMFE_OK;
#else
    hipLaunchKernelGGL((fbank512_kernel<PCM, NA, VEC2, EXACT, NGA, NGB, MODE>), dim3(grid), dim3(64 * kWavesPerBlock), lds, stream, args);
    MMFE_HIP_CHECK(hipGetLastError());
    return MMFE_OK;
#endif
}

// Instantiated (NGA, NGB) shapes of the matrix-pipe mel form, smallest first.
struct Mel4Shape { int nga, ngb; };
constexpr Mel4Shape kMel4Shapes[] = {{2, 7}, {3, 8}, {4, 0}, {8, 0}, {4, 8}};

}  // namespace

// Builds the per-(set, lane) bin runs and the step-major weight table of the matrix-pipe mel form.
// Returns false when the bank does not fit any instantiated shape (the VALU form is used then).
static bool build_mel4(Plan& plan, bool align) {
    const mmfe_spec& s = plan.spec;
    if (s.n_mel > 128) return false;
    const int nsets = (s.n_mel + 63) / 64;
    std::vector<int> boff(s.n_mel + 1, 0);
    for (int m = 0; m < s.n_mel; ++m) boff[m + 1] = boff[m] + plan.band_len[m];
    int kstart[2][16];
    int steps[2] = {0, 0};
    for (int h = 0; h < 2; ++h) {
        for (int blk = 0; blk < 16; ++blk) {
            int kmin = 1 << 30, kmax = -1;
            for (int j = 0; j < 4; ++j) {
                const int m = 64 * h + 4 * blk + j;
                if (h >= nsets || m >= s.n_mel || plan.band_len[m] == 0) continue;
                kmin = std::min(kmin, (int)plan.band_start[m]);
                kmax = std::max(kmax, (int)(plan.band_start[m] + plan.band_len[m] - 1));
            }
            if (kmax < 0) { kstart[h][blk] = 0; continue; }
            int ks = kmin;
            if (MMFE_MEL_PAIR) ks &= ~1;                             // 8-byte operand reads: even run starts
            if (align && !MMFE_MEL_PAIR) ks = kmin - (((kmin - blk) % 8 + 8) % 8);     // run start = blk (mod 8): conflict-free LDS reads
            if (ks < -kPGuard) ks = kmin;
            kstart[h][blk] = ks;
            steps[h] = std::max(steps[h], kmax - ks + 1);
        }
    }
    const int ga = (steps[0] + 3) / 4, gb = (steps[1] + 3) / 4;
    int pick = -1;
    for (size_t i = 0; i < sizeof(kMel4Shapes) / sizeof(kMel4Shapes[0]); ++i) {
        if (ga <= kMel4Shapes[i].nga && gb <= kMel4Shapes[i].ngb && ga >= 1) { pick = (int)i; break; }
    }
    if (pick < 0) return false;
    const int NGA = kMel4Shapes[pick].nga, NGB = kMel4Shapes[pick].ngb;
    const int total_steps = 4 * (NGA + NGB);
    // Free bank-conflict reduction: a block whose run is shorter than its set's step count may start its run a
    // few bins early (the extra leading steps carry zero weights).  The A operands of the 8 blocks that share
    // a 32-lane half land on distinct LDS banks when their run starts are distinct modulo 8, so slide each
    // run, within its slack, to a residue its half has not used yet (least flexible blocks choose first).
    if (!align && !(std::getenv("MMFE_MEL_SLIDE") != nullptr && std::string(std::getenv("MMFE_MEL_SLIDE")) == "0")) {
        for (int h = 0; h < 2; ++h) {
            const int budget = 4 * (h == 0 ? NGA : NGB);
            for (int half = 0; half < 2; ++half) {
                int run[8], order[8];
                bool used[8] = {false, false, false, false, false, false, false, false};
                for (int i = 0; i < 8; ++i) {
                    const int blk = 8 * half + i;
                    int kmax = -1;
                    for (int j = 0; j < 4; ++j) {
                        const int m = 64 * h + 4 * blk + j;
                        if (h >= nsets || m >= s.n_mel || plan.band_len[m] == 0) continue;
                        kmax = std::max(kmax, (int)(plan.band_start[m] + plan.band_len[m] - 1));
                    }
                    run[i] = kmax < 0 ? 0 : kmax - kstart[h][blk] + 1;
                    order[i] = i;
                }
                std::sort(order, order + 8, [&](int x, int y) { return run[x] > run[y]; });   // longest run = least slack first
                for (int oi = 0; oi < 8; ++oi) {
                    const int i = order[oi], blk = 8 * half + i;
                    const int slack = run[i] == 0 ? 7 : std::min(7, budget - run[i]);
                    int chosen = 0;
                    for (int sh = 0; sh <= slack; sh += MMFE_MEL_PAIR ? 2 : 1) {
                        const int cand = kstart[h][blk] - sh;
                        if (cand < -kPGuard) break;
                        if (!used[((cand % 8) + 8) % 8]) { chosen = sh; break; }
                    }
                    kstart[h][blk] -= chosen;
                    used[((kstart[h][blk] % 8) + 8) % 8] = true;
                }
            }
        }
    }
    // every run must stay inside the frame's LDS tile
    for (int h = 0; h < 2; ++h)
        for (int blk = 0; blk < 16; ++blk)
            if (kPGuard + 24 + kstart[h][blk] + 4 * std::max(NGA, NGB) > kFrameFloats) return false;
    std::vector<int32_t> ks(128, 0);
    std::vector<float> w((size_t)total_steps * 64, 0.f);
    for (int h = 0; h < nsets; ++h) {
        const int nsteps = 4 * (h == 0 ? NGA : NGB);
        const int soff = h == 0 ? 0 : 4 * NGA;
        for (int lane = 0; lane < 64; ++lane) {
            const int blk = lane / 4, m = 64 * h + lane;
            ks[h * 64 + lane] = kstart[h][blk];
            if (m >= s.n_mel) continue;
            for (int st = 0; st < nsteps; ++st) {
                const int k = kstart[h][blk] + st;
                const int rel = k - plan.band_start[m];
                if (rel >= 0 && rel < plan.band_len[m]) w[(size_t)(soff + st) * 64 + lane] = plan.band_weights[boff[m] + rel];
            }
        }
    }
    plan.mel4_nga = NGA;
    plan.mel4_ngb = NGB;
    if (upload(&plan.d_mel4_kstart, ks) != MMFE_OK) return false;
    if (upload(&plan.d_mel4_w, w) != MMFE_OK) return false;
    return true;
}

int fbank512_prepare(Plan& plan) {
    const mmfe_spec& s = plan.spec;
    if (s.n_fft != 512 || s.win_off != 0 || s.win_len > 512 || s.win_len < 2) {
        set_error("fbank512: needs n_fft == 512, win_off == 0, 2 <= win_len <= 512");
        return MMFE_EINVAL;
    }
    if (s.n_mel <= 0 || s.n_mel > 16 * kMaxPass) {
        set_error("fbank512: n_mel must be in [1, %d]", 16 * kMaxPass);
        return MMFE_EINVAL;
    }
    // window: pre-halved (folds the 1/2 of the real-FFT recombination; exact in binary fp), zero padded
    // 16-bit PCM samples run through the pipeline as integers: their 1/32768 rides on the window as well
    const float wscale = s.in_dtype == MMFE_IN_S16 ? 0.5f / 32768.0f : 0.5f;
    std::vector<float> win(512, 0.f);
    for (int i = 0; i < s.win_len; ++i) win[i] = wscale * plan.window[i];
    int rc = upload(&plan.d_window, win);
    if (rc) return rc;
    std::vector<float> tws(2 * 256), twp(2 * kTwPostEntries);
    const double pi = 3.14159265358979323846;
    for (int n = 0; n < 256; ++n) {
        tws[2 * n] = (float)std::cos(2.0 * pi * n / 256.0);
        tws[2 * n + 1] = (float)-std::sin(2.0 * pi * n / 256.0);
    }
    for (int k = 0; k < kTwPostEntries; ++k) {   // -i * exp(-2 pi i k / 512)
        twp[2 * k] = (float)-std::sin(2.0 * pi * k / 512.0);
        twp[2 * k + 1] = (float)-std::cos(2.0 * pi * k / 512.0);
    }
    if ((rc = upload(&plan.d_tw_a, tws))) return rc;
    if ((rc = upload(&plan.d_tw_b, twp))) return rc;

    for (int m = 0; m < s.n_mel; ++m) {
        const int st = plan.band_start[m], len = plan.band_len[m];
        if (len > 0 && (st < 0 || st + len > 257)) {
            set_error("fbank512: mel band %d = [%d, %d) leaves the 257-bin spectrum", m, st, st + len);
            return MMFE_EINVAL;
        }
    }
    // matrix-pipe mel form when the bank fits an instantiated shape (MMFE_MEL_FORM=valu forces the VALU form,
    // MMFE_MEL_ALIGN=1 starts every bin run at blk (mod 8), which makes the LDS operand reads bank-conflict-free
    // at the price of longer runs; measured slower than the shorter unaligned runs, so it is off by default)
    // tuning knobs are read here, once per plan, never on the launch path
    if (const char* e = std::getenv("MMFE_FBANK_BLOCKS_PER_CU")) plan.tune_blocks_per_cu = std::atoi(e);     // fewer resident blocks per CU
    if (const char* e = std::getenv("MMFE_FBANK_CHUNK")) plan.tune_chunk = std::atoi(e);                     // quads per wave visit
    const char* form = std::getenv("MMFE_MEL_FORM");
    const char* al = std::getenv("MMFE_MEL_ALIGN");
    const bool want_mfma = !(form != nullptr && std::string(form) == "valu");
    const bool align = al != nullptr && std::string(al) == "1";
    plan.mel4_nga = plan.mel4_ngb = 0;
    if (want_mfma && !build_mel4(plan, align) && align) (void)build_mel4(plan, false);

    // VALU form tables (always built: they are small and serve as the general path)
    MelPassTable& mt = plan.mel16;
    mt.n_pass = (s.n_mel + 15) / 16;
    mt.start.assign(mt.n_pass * 16, 0);
    int rows = 0;
    std::vector<int> boff(s.n_mel + 1, 0);
    for (int m = 0; m < s.n_mel; ++m) boff[m + 1] = boff[m] + plan.band_len[m];
    for (int j = 0; j < mt.n_pass; ++j) {
        int longest = 0;
        for (int l = 0; l < 16; ++l) {
            const int m = 16 * j + l;
            if (m < s.n_mel) longest = std::max(longest, (int)plan.band_len[m]);
        }
        mt.niter[j] = longest;
        mt.woff[j] = rows;
        rows += longest;
    }
    mt.total_rows = rows;
    mt.weights.assign((size_t)std::max(rows, 1) * 16, 0.f);
    for (int j = 0; j < mt.n_pass; ++j) {
        for (int l = 0; l < 16; ++l) {
            const int m = 16 * j + l;
            if (m >= s.n_mel) continue;
            int st = plan.band_start[m];
            const int len = plan.band_len[m];
            if (len == 0) st = 0;
            // every lane walks niter[j] taps: the walk must stay inside the frame's LDS tile
            if (kPGuard + 24 + st + mt.niter[j] > kFrameFloats) {
                set_error("fbank512: mel pass %d too long", j);
                return MMFE_EINVAL;
            }
            mt.start[m] = st;
            for (int i = 0; i < len; ++i) mt.weights[(size_t)(mt.woff[j] + i) * 16 + l] = plan.band_weights[boff[m] + i];
        }
    }
    std::vector<int32_t> niter(mt.niter, mt.niter + kMaxPass), woff(mt.woff, mt.woff + kMaxPass);
    if ((rc = upload(&plan.d_mel_start, mt.start))) return rc;
    if ((rc = upload(&plan.d_mel_w, mt.weights))) return rc;
    if ((rc = upload(&plan.d_mel_niter, niter))) return rc;
    if ((rc = upload(&plan.d_mel_woff, woff))) return rc;
    return MMFE_OK;
}

int fbank512_launch(const Plan& plan, const RowSource& src, int64_t n_rows, int64_t n_samples, int64_t n_frames_natural,
                    int64_t out_frames, void* out, hipStream_t stream) {
    const mmfe_spec& s = plan.spec;
    const int64_t frames_valid = std::min(n_frames_natural, out_frames);
    if (n_rows <= 0 || frames_valid <= 0) return MMFE_OK;
    const int64_t quads = n_rows * ((frames_valid + 3) / 4);
    if (quads >= (int64_t)1 << 31 || n_rows >= (int64_t)1 << 31) {
        set_error("fbank512: batch too large for one launch (%lld frame quads)", (long long)quads);
        return MMFE_EINVAL;
    }
    FbankArgs a{};
    a.src = src;
    a.out = out;
    a.n_rows = (int)n_rows;
    a.n_samples = (int)std::min<int64_t>(n_samples, INT32_MAX);
    a.frames_valid = (int)frames_valid;
    a.out_rows = (int)out_frames;
    a.hop = s.hop;
    a.win_len = s.win_len;
    a.n_mel = s.n_mel;
    a.out_dtype = s.out_dtype;
    a.vec_out = (s.n_mel % 4 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0) ? 1 : 0;
    a.feat_major = s.layout == MMFE_LAYOUT_FEAT_MAJOR ? 1 : 0;
    a.vec_out_fm = (out_frames % 4 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0) ? 1 : 0;
    a.preemph = s.preemph;
    a.mean_scale = s.remove_dc ? 1.0f / (float)s.win_len : 0.0f;
    a.la = make_log_affine(s);
    a.window = plan.d_window;
    a.tw_stage = reinterpret_cast<const float2*>(plan.d_tw_a);
    a.tw_post = reinterpret_cast<const float2*>(plan.d_tw_b);
    a.n_pass = plan.mel16.n_pass;
    a.mel_rows = std::max(plan.mel16.total_rows, 1);
    a.mel_start = plan.d_mel_start;
    a.mel_niter = plan.d_mel_niter;
    a.mel_woff = plan.d_mel_woff;
    a.mel_w = plan.d_mel_w;
    a.mel4_kstart = plan.d_mel4_kstart;
    a.mel4_w = plan.d_mel4_w;

    const bool mfma = plan.mel4_nga > 0;
    const size_t tables = kTableBytes + (mfma ? 0 : kMaxPass * 16 * 4 + (size_t)a.mel_rows * 64);
    // feature-major tile mode: matrix-pipe mel (a lane holds 4 frames of one filter), rows of equal length, neighbouring quads per block
    a.fm_tile = (a.feat_major && mfma && src.row_len == nullptr && plan.tune_chunk <= 1 && s.n_mel <= 128 && kWavesPerBlock == 4) ? 1 : 0;
    const size_t lds = tables + (size_t)kWavesPerBlock * 4 * kFrameLds + (a.fm_tile ? kFmTileBytes : 0);
    if (lds > 160 * 1024) {
        set_error("fbank512: mel tap table does not fit in LDS (%zu bytes)", lds);
        return MMFE_EINVAL;
    }
    const int blocks_per_cu = std::max<int>(1, (int)((160 * 1024) / lds));
    const int by_waves = std::max(1, (4 * MMFE_FBANK_MIN_WAVES_PER_SIMD) / kWavesPerBlock);
    int per_cu = std::min(blocks_per_cu, by_waves);
    if (plan.tune_blocks_per_cu >= 1 && plan.tune_blocks_per_cu < per_cu) per_cu = plan.tune_blocks_per_cu;
    // feature-major rows outside the tile mode: 8 consecutive quads per visit, so that a wave completes 128-byte lines of every filter row
    a.chunk = std::max(1, std::min(plan.tune_chunk > 0 ? plan.tune_chunk : (a.feat_major && !a.fm_tile ? 8 : kDefaultChunk), 1 << 16));
    const int64_t want = (quads + (int64_t)kWavesPerBlock * a.chunk - 1) / ((int64_t)kWavesPerBlock * a.chunk);

    // one load per 2-sample slice needs every frame start aligned to the slice (8 bytes of floats, 4 of PCM)
    const bool pcm = s.in_dtype == MMFE_IN_S16;
    bool vec2 = (s.hop % 2 == 0) && (s.win_len % 2 == 0) && (reinterpret_cast<uintptr_t>(src.base) % (pcm ? 4 : 8) == 0);
    if (src.offsets == nullptr) vec2 = vec2 && (src.stride % 2 == 0);
    else vec2 = false;   // slice offsets are data: take the scalar-load variant
    const int na = (s.win_len + 31) / 32;
    const int nga = plan.mel4_nga, ngb = plan.mel4_ngb;
    const bool ragged = src.row_len != nullptr;
#define MMFE_FBANK_DISPATCH(NGA_, NGB_)                                                                                  \
    if (nga == NGA_ && ngb == NGB_) {                                                                                    \
        if (ragged) {      /* ragged batches: the forms without a compile-time window length */                          \
            if (pcm) return vec2 ? launch_variant<true, 16, true, false, NGA_, NGB_, kRowsRagged>(a, want, per_cu, lds, stream)  \
                                 : launch_variant<true, 16, false, false, NGA_, NGB_, kRowsRagged>(a, want, per_cu, lds, stream); \
            if (na == 13 && vec2) return launch_variant<false, 13, true, true, NGA_, NGB_, kRowsRagged>(a, want, per_cu, lds, stream); \
            return vec2 ? launch_variant<false, 16, true, false, NGA_, NGB_, kRowsRagged>(a, want, per_cu, lds, stream)         \
                        : launch_variant<false, 16, false, false, NGA_, NGB_, kRowsRagged>(a, want, per_cu, lds, stream);       \
        }                                                                                                                \
        const bool whole = NGA_ > 0 && frames_valid % 4 == 0 && a.vec_out && s.out_dtype == MMFE_OUT_F32 &&              \
                           s.n_mel == (NGB_ > 0 ? 128 : 64) && !a.feat_major;                                            \
        if (pcm) {                                                                                                       \
            if (na == 13 && vec2) return whole ? launch_variant<true, 13, true, true, NGA_, NGB_, kRowsWhole>(a, want, per_cu, lds, stream) \
                                               : launch_variant<true, 13, true, true, NGA_, NGB_, kRowsAny>(a, want, per_cu, lds, stream); \
            return vec2 ? launch_variant<true, 16, true, false, NGA_, NGB_, kRowsAny>(a, want, per_cu, lds, stream)      \
                        : launch_variant<true, 16, false, false, NGA_, NGB_, kRowsAny>(a, want, per_cu, lds, stream);    \
        }                                                                                                                \
        if (na == 13 && vec2) return whole ? launch_variant<false, 13, true, true, NGA_, NGB_, kRowsWhole>(a, want, per_cu, lds, stream) \
                                           : launch_variant<false, 13, true, true, NGA_, NGB_, kRowsAny>(a, want, per_cu, lds, stream); \
        return vec2 ? launch_variant<false, 16, true, false, NGA_, NGB_, kRowsAny>(a, want, per_cu, lds, stream)         \
                    : launch_variant<false, 16, false, false, NGA_, NGB_, kRowsAny>(a, want, per_cu, lds, stream);       \
    }
    MMFE_FBANK_DISPATCH(2, 7)
    MMFE_FBANK_DISPATCH(3, 8)
    MMFE_FBANK_DISPATCH(4, 0)
    MMFE_FBANK_DISPATCH(8, 0)
    MMFE_FBANK_DISPATCH(4, 8)
    MMFE_FBANK_DISPATCH(0, 0)
#undef MMFE_FBANK_DISPATCH
    set_error("fbank512: no kernel variant for mel shape (%d, %d)", nga, ngb);
    return MMFE_EINVAL;
}

}  // namespace mmfe
