// EXPERIMENT, not part of the library: the n_fft 400 kernel restructured for three waves per SIMD (8 KB of LDS work area per wave through a
// two-shot XOR-swizzled transpose tile, window and mel weights as LDS tables, <= 168 VGPRs).  Bit-identical outputs; measured SLOWER than
// two waves per SIMD (profiles/r03_rfft400_occupancy.txt): the same build limited to 2 blocks per CU beats itself at 3.
// Build: SRCFILE=scripts/experiments/rfft400_three_waves.hip scripts/build_variant_of.sh rfft400 r4late -DMMFE_R400_EARLY_FETCH=0
// Fused log-mel front-end for n_fft = 400 (25 ms frames at 16 kHz) on gfx950: framing with centre
// reflect/zero padding or edge snipping -> window -> 400-point real FFT -> |X|^2 -> banded mel ->
// log -> affine, spectra never leave the CU.  This is the long-utterance configuration
// (n_fft 400, hop 160, 80 mel bands, 30 s utterances); its parameters are those of
// transformers' WhisperFeatureExtractor, the reference itself has no n_fft = 400 caller.
//
// Mapping: 16 lanes own a PAIR of consecutive frames (A, B); a wave64 carries 8 consecutive frames
// of one row per pass.
//   * every lane loads its own strided samples (n = 16a + b of its two frames) straight from global memory, one pass
//     ahead; the 2.5x overlap of neighbouring frames is served by L1 / L2, so HBM sees every sample about once;
//   * z[n] = w[n]*(A[n] + i*B[n]) is transformed by ONE 400-point complex FFT, split 25 x 16:
//     lane b runs a register-resident 25-point FFT (radix 5 x 5) over z[16a+b], twiddles by
//     W400^(b*c), the 16 lanes transpose through an XOR-swizzled LDS tile in two shots (rows 0..15, then
//     rows 16..24 in the same place), and each lane runs the 16-point FFTs of columns c = b and c = b+16
//     (the second only for b <= 8), leaving Z[c + 25d];
//   * the two real spectra separate as X_A[k] = (Z[k] + conj Z[400-k])/2 and
//     X_B[k] = (Z[k] - conj Z[400-k])/(2i); the window is pre-halved, so |X_A|^2 and |X_B|^2 are two
//     sums of squares per bin, written to LDS;
//   * banded mel on the matrix pipe: v_mfma_f32_4x4x1 runs 16 independent blocks of (4 frames) x
//     (4 filters); a block walks ITS OWN run of bins, one bin per instruction, weights resident in
//     an LDS table.  The (filter block, frame group) units are spread over up to four instruction streams
//     in order of frequency, so the short low-frequency runs do not wait for the long high ones; a
//     last stream that is at most half full splits its runs in two and adds the halves.
// Three waves per SIMD: 8 KB of LDS work area per wave (the two-shot tile) and at most 168 VGPRs per lane, which is why
// the window and the mel weights are LDS tables (16-byte reads) and only the prefetched samples stay in registers.
#include "mmfe_internal.h"
#include "fft_regs.h"
#include "log_affine.h"

#include <algorithm>
#include <functional>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <type_traits>
#include <vector>

namespace mmfe {

namespace {

constexpr int kN = 400;
constexpr int kBins = 201;
constexpr int kWaves = 4;                           // waves per block
constexpr int kMaxHop = 160;
constexpr int kPS = 216;                            // floats between the power spectra of consecutive frames (= 24 mod 64: 8 frames on 8 bank groups)
constexpr int kPGuard = 8;                          // floats in front of bin 0 (a bin run may start below 0)
constexpr int kTwBytes = 25 * 16 * 8;               // W400^(b*c) table, [c][b]
constexpr int kStreams = 4;

#ifndef MMFE_R400_EARLY_FETCH
#define MMFE_R400_EARLY_FETCH 1     // shared-load form (hop a multiple of 16): request the next octet's samples right after the first tile shot has
                                    // been written instead of after the 16-point transforms (35 values in flight fit the register budget there)
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Rfft400Args {
    RowSource src;
    float* out;
    int* rowmax;               // per-row running maximum of the stored values (ordered-int form), or nullptr
    int* rowmin;               // per-row running minimum, likewise
    int n_rows;
    int n_samples;
    int frames_valid;          // frames computed per row
    int out_frames;            // frames per row in the output tensor
    int hop;
    int shift;                 // DFT input position n of frame t holds sample t*hop + n + shift
    int pad_mode;
    int n_mel;
    int layout;
    int out_dtype;             // MMFE_OUT_*: 16-bit kinds store the rounded value (what .to(dtype) of the float32 result gives)
    int vec_store;             // feature-major rows may be stored 16 bytes at a time
    LogAffine la;
    const float* window;       // 400 floats: window pre-halved at win_off, zeros elsewhere
    const float2* tw;          // [25][16]: W400^(b*c)
    const int32_t* mel_poff;   // [kStreams][64]: float offset of the lane's bin run in the wave's spectrum area
    const int32_t* mel_dst;    // [kStreams][64]: filter | frame group << 16, or -1 for an idle lane
    const float* mel_w;        // [total steps][64], step-major
};

// 5-point forward DFT in place.
__device__ __forceinline__ void dft5(float2& x0, float2& x1, float2& x2, float2& x3, float2& x4) {
    constexpr float C1 = 0.30901699437494745f, C2 = -0.80901699437494734f;   // cos(2pi/5), cos(4pi/5)
    constexpr float S1 = 0.95105651629515353f, S2 = 0.58778525229247325f;    // sin(2pi/5), sin(4pi/5)
    const float2 t1 = cadd(x1, x4), t2 = cadd(x2, x3), t3 = csub(x1, x4), t4 = csub(x2, x3);
    const float2 a1 = make_float2(fmaf(C2, t2.x, fmaf(C1, t1.x, x0.x)), fmaf(C2, t2.y, fmaf(C1, t1.y, x0.y)));
    const float2 a2 = make_float2(fmaf(C1, t2.x, fmaf(C2, t1.x, x0.x)), fmaf(C1, t2.y, fmaf(C2, t1.y, x0.y)));
    const float2 b1 = make_float2(fmaf(S2, t4.x, S1 * t3.x), fmaf(S2, t4.y, S1 * t3.y));
    const float2 b2 = make_float2(fmaf(-S1, t4.x, S2 * t3.x), fmaf(-S1, t4.y, S2 * t3.y));
    x0 = make_float2(x0.x + t1.x + t2.x, x0.y + t1.y + t2.y);
    x1 = make_float2(a1.x + b1.y, a1.y - b1.x);      // a1 - i*b1
    x4 = make_float2(a1.x - b1.y, a1.y + b1.x);
    x2 = make_float2(a2.x + b2.y, a2.y - b2.x);
    x3 = make_float2(a2.x - b2.y, a2.y + b2.x);
}

// W25^m = exp(-2 pi i m / 25), m a compile-time constant in [0, 16]
__device__ __forceinline__ float2 w25(int m) {
    constexpr float C[17] = {1.f, 0.96858316112863108f, 0.87630668004386358f, 0.72896862742141155f, 0.53582679497899655f,
                             0.30901699437494745f, 0.062790519529313527f, -0.1873813145857246f, -0.42577929156507272f,
                             -0.63742398974868975f, -0.80901699437494734f, -0.92977648588825135f, -0.99211470131447776f,
                             -0.99211470131447788f, -0.92977648588825146f, -0.80901699437494778f, -0.63742398974868952f};
    constexpr float S[17] = {0.f, 0.24868988716485479f, 0.48175367410171532f, 0.68454710592868862f, 0.84432792550201508f,
                             0.95105651629515353f, 0.99802672842827156f, 0.98228725072868872f, 0.90482705246601947f,
                             0.77051324277578925f, 0.58778525229247325f, 0.36812455268467814f, 0.12533323356430454f,
                             -0.12533323356430429f, -0.36812455268467792f, -0.58778525229247269f, -0.77051324277578936f};
    return make_float2(C[m], -S[m]);
}

// 25-point forward DFT in place, radix 5 x 5.  Input z[a] in natural order; output Z[c] is left at
// z[fft25_pos(c)].
__host__ __device__ constexpr int fft25_pos(int c) { return 5 * (c % 5) + c / 5; }

__device__ __forceinline__ void fft25(float2 (&z)[25]) {
    // a = 5*a1 + a2, c = c1 + 5*c2:  W25^(a*c) = W5^(a1*c1) * W25^(a2*c1) * W5^(a2*c2)
#pragma unroll
    for (int a2 = 0; a2 < 5; ++a2) dft5(z[a2], z[5 + a2], z[10 + a2], z[15 + a2], z[20 + a2]);    // c1 lands at z[5*c1 + a2]
#pragma unroll
    for (int c1 = 1; c1 < 5; ++c1)
#pragma unroll
        for (int a2 = 1; a2 < 5; ++a2) z[5 * c1 + a2] = cmul(z[5 * c1 + a2], w25(a2 * c1));
#pragma unroll
    for (int c1 = 0; c1 < 5; ++c1) dft5(z[5 * c1], z[5 * c1 + 1], z[5 * c1 + 2], z[5 * c1 + 3], z[5 * c1 + 4]);   // c2 lands at z[5*c1 + c2]
}

// Stored element i of a row of n >= 1 under the plan's padding rule (i may lie outside the row), undecoded.  Straight-line code (an
// index that always lies inside the row, then a select): with a branch per sample the 50 calls of an edge octet turn into 50 small
// basic blocks whose loaded values the register allocator parks in scratch.
template <typename S>
__device__ __forceinline__ S fetch_raw(const S* row, int i, int n, int pad_mode) {
    const int period = n > 1 ? 2 * (n - 1) : 1;
    int m = i % period;
    m = m < 0 ? m + period : m;
    m = m < n ? m : period - m;                       // the reflected position
    const bool inside = i >= 0 && i < n;
    const S v = row[inside ? i : m];
    return (inside || pad_mode == MMFE_PAD_REFLECT) ? v : S(0);
}

// One output element / four consecutive ones in the plan's output type; `i` counts elements.
__device__ __forceinline__ void store1(const Rfft400Args& p, int64_t i, float v) {
    if (p.out_dtype == MMFE_OUT_F32) p.out[i] = v;
    else reinterpret_cast<unsigned short*>(p.out)[i] = p.out_dtype == MMFE_OUT_BF16 ? bf16_bits(v) : f16_bits(v);
}
__device__ __forceinline__ void store4(const Rfft400Args& p, int64_t i, float a, float b, float c, float d) {     // i % 4 == 0, base 16-byte aligned
    if (p.out_dtype == MMFE_OUT_F32) *reinterpret_cast<float4*>(p.out + i) = make_float4(a, b, c, d);
    else *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(p.out) + i) = make_uint2(pack2_bits(a, b, p.out_dtype), pack2_bits(c, d, p.out_dtype));
}

// LDS of a block: twiddles | window | mel weights | one work area per wave.
//   work area of a wave: 4 pairs x kPairBytes.  Every phase of a pass reuses it:
//     T1  rows c = 0..15 of the pair's 25 x 16 transpose tile (16 rows x 128 bytes), T2 rows 16..24 in the same place once T1 has been
//         read (the tile in two shots: 2 KB per pair instead of 3.2 KB is what lets a third wave per SIMD fit the CU's LDS);
//         row r keeps its 16-byte column pairs XOR-swizzled by (r >> 1) & 7, so that the 16 rows one 16-byte read touches fall on 16
//         distinct bank groups without padding the 128-byte row stride
//     Z   the upper half of the spectrum, Z[200 .. 400], parked at index k - 200 for the partners of the bins k <= 200
//     P   the 8 power spectra of the wave (kPS floats apart), operands of the matrix-pipe mel projection
constexpr int kPairBytes = 2048;
constexpr int kWaveBytes = 4 * kPairBytes;          // 8192
constexpr int kWinStride = 28;                      // floats per lane in the LDS window table: 25 taps + 3; 16-byte reads of 16 lanes conflict-free
constexpr int kWinBytes = 16 * kWinStride * 4;      // 1792

// Floats per lane in the LDS mel-weight table: every stream's steps padded to a multiple of 4 (16-byte reads), and a stride whose
// quarter is odd (the 16-byte reads of 16 consecutive lanes fall on distinct bank groups).
constexpr int r400_pad4(int v) { return (v + 3) & ~3; }
constexpr int r400_wstride(int s0, int s1, int s2, int s3) {
    const int t = r400_pad4(s0) + r400_pad4(s1) + r400_pad4(s2) + r400_pad4(s3);
    return (t / 4) % 2 == 1 ? t : t + 4;
}

template <typename SMP /* input element: float or int16_t PCM */, int HS /* hop / 16 when the hop is a multiple of 16 (at most 160), else 0 */,
          int S0, int S1, int S2, int S3 /* bins walked by the four mel instruction streams */, bool SPLIT /* last stream: runs split in two halves */>
__global__ void __launch_bounds__(64 * kWaves, 3) rfft400_kernel(const Rfft400Args p) {
    constexpr int kSteps[kStreams] = {S0, S1, S2, S3};
    constexpr int kStepOff[kStreams] = {0, S0, S0 + S1, S0 + S1 + S2};                                        // in the step-major device table
    constexpr int kLdsOff[kStreams] = {0, r400_pad4(S0), r400_pad4(S0) + r400_pad4(S1), r400_pad4(S0) + r400_pad4(S1) + r400_pad4(S2)};
    constexpr int kWStride = r400_wstride(S0, S1, S2, S3);
    constexpr int kLastStream = S3 > 0 ? 3 : (S2 > 0 ? 2 : (S1 > 0 ? 1 : 0));
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* s_tw = reinterpret_cast<float2*>(smem);                                  // [25][16]: W400^(b*c)
    float* s_win = reinterpret_cast<float*>(smem + kTwBytes);                        // [16][kWinStride]: window[16a + b] at [b][a]
    float* s_w = reinterpret_cast<float*>(smem + kTwBytes + kWinBytes);              // [64][kWStride]: mel weights of lane l, stream by stream
    for (int i = threadIdx.x; i < 25 * 16; i += blockDim.x) s_tw[i] = p.tw[i];
    for (int i = threadIdx.x; i < 16 * kWinStride; i += blockDim.x) {
        const int bb = i / kWinStride, a = i - bb * kWinStride;
        s_win[i] = a < 25 ? p.window[16 * a + bb] : 0.f;
    }
    for (int i = threadIdx.x; i < 64 * kWStride; i += blockDim.x) {
        const int l = i / kWStride, o = i - l * kWStride;
        float w = 0.f;
#pragma unroll
        for (int st = 0; st < kStreams; ++st)
            if (kSteps[st] > 0 && o >= kLdsOff[st] && o < kLdsOff[st] + kSteps[st]) w = p.mel_w[(kStepOff[st] + o - kLdsOff[st]) * 64 + l];
        s_w[i] = w;
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* const wave_area = smem + kTwBytes + kWinBytes + 64 * kWStride * 4 + wave * kWaveBytes;
    float* const parea = reinterpret_cast<float*>(wave_area);

    for (int i = lane; i < kWaveBytes / 16; i += 64) reinterpret_cast<float4*>(wave_area)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    wave_lds_fence();

    const int hop = p.hop;
    // what a lane keeps across passes: where its mel operands start and which (filter, frame group) each stream leaves it with
    int poff[kStreams], dst[kStreams];
#pragma unroll
    for (int st = 0; st < kStreams; ++st) {
        poff[st] = kSteps[st] > 0 ? p.mel_poff[st * 64 + lane] : 0;
        dst[st] = kSteps[st] > 0 ? p.mel_dst[st * 64 + lane] : -1;
    }
    // ---- work distribution: octet = 8 consecutive frames of one row; a block owns a contiguous range of rounds of kWaves consecutive
    //      octets, so a wave stays inside a row for many passes (the per-row extremes of clamp_below_max are reduced once per row) ------
    const unsigned opr = (unsigned)(p.frames_valid + 7) >> 3;      // octets per row
    const unsigned n_rounds = ((unsigned)p.n_rows * opr + kWaves - 1) / kWaves;
    const unsigned per_block = (n_rounds + gridDim.x - 1) / gridDim.x;
    const unsigned round0 = blockIdx.x * per_block;
    unsigned rounds_left = round0 < n_rounds ? (n_rounds - round0 < per_block ? n_rounds - round0 : per_block) : 0;
    const unsigned ostride = kWaves;
    const unsigned oct0 = round0 * kWaves + wave;
    const unsigned d_row = ostride / opr, d_o = ostride - d_row * opr;
    unsigned row = oct0 / opr, oq = oct0 - row * opr;               // wave-uniform
    auto advance = [&](unsigned& r, unsigned& o) {
        r += d_row;
        o += d_o;
        if (o >= opr) { o -= opr; r += 1; }
    };

    // the samples of the lane's two frames under octet (r, o): n = 16a + b of frame A = 2*pr and of frame B = A + 1, straight from
    // global memory one pass ahead (the 2.5x overlap of neighbouring frames is served by L1 / L2; kept as loaded, 16-bit PCM is decoded
    // where the values are used, so that nothing waits for the loads here).  With a hop of 16 * HS, frame B's sample a IS frame A's
    // sample a + HS of the same lane: one run g[0 .. 24 + HS] serves both frames (35 loads instead of 50 at hop 160).
    constexpr int kNG = HS > 0 ? 25 + HS : 50;
    constexpr bool kEarlyFetch = HS > 0 && MMFE_R400_EARLY_FETCH != 0;
    SMP g[kNG];
    auto fetch = [&](unsigned r, unsigned o, int pr, int b) {
        const unsigned rr = r < (unsigned)p.n_rows ? r : 0;
        const SMP* rp = row_ptr_as<SMP>(p.src, rr);
        const int n_row = row_samples(p.src, rr, p.n_samples);
        const int first = (int)o * 8 * hop + p.shift;
        const bool interior = first >= 0 && first + 7 * hop + kN <= n_row;     // wave-uniform
        const int ia = first + 2 * pr * hop + b;
        if (interior) {
            const SMP* qa = rp + ia;
            if (HS > 0) {
#pragma unroll
                for (int a = 0; a < kNG; ++a) g[a] = qa[16 * a];
            } else {
                const SMP* qb = qa + hop;
#pragma unroll
                for (int a = 0; a < 25; ++a) { g[a] = qa[16 * a]; g[25 + a] = qb[16 * a]; }
            }
        } else if (n_row > 0) {    // the octet touches an end of the row: padding rule per sample
#pragma unroll
            for (int a = 0; a < (HS > 0 ? kNG : 25); ++a) {
                g[a] = fetch_raw(rp, ia + 16 * a, n_row, p.pad_mode);
                if (HS == 0) g[25 + a] = fetch_raw(rp, ia + hop + 16 * a, n_row, p.pad_mode);
                __builtin_amdgcn_sched_barrier(0);      // one position at a time: interleaved, the index arithmetic of all of them needs more
                                                        // registers than the pass has, and the values loaded first go to scratch
            }
        } else {
#pragma unroll
            for (int a = 0; a < kNG; ++a) g[a] = SMP(0);
        }
    };
    fetch(row, oq, lane >> 4, lane & 15);

    float lmax = -INFINITY, lmin = INFINITY;      // running extremes of the row the wave is in (clamp_below_max plans)
    auto flush_extremes = [&](unsigned r) {
        const float wmax = wave_max(lmax), wmin = wave_min(lmin);
        if (lane == 0 && wmax > -INFINITY) {
            atomicMax(p.rowmax + r, float_to_ordered(wmax));
            atomicMin(p.rowmin + r, float_to_ordered(wmin));
        }
        lmax = -INFINITY;
        lmin = INFINITY;
    };
    for (; rounds_left > 0 && row < (unsigned)p.n_rows; --rounds_left) {
        const int t0 = (int)oq * 8;          // first frame of the octet
        const unsigned cur_row = row;
        // Everything a lane derives from its number (tile addresses, swizzle keys, table rows) is recomputed per pass from a copy the
        // optimiser cannot see through: kept across the loop these two dozen values cost more registers than the 168 a third wave
        // per SIMD leaves, and went to scratch.
        int l = lane;
        asm volatile("" : "+v"(l));
        const int pr = l >> 4;         // frame pair within the wave
        const int b = l & 15;          // lane within the pair's group
        char* const area = wave_area + pr * kPairBytes;
        float2* const zpark = reinterpret_cast<float2*>(area);
        const int wr_col = 16 * (b >> 1) + 8 * (b & 1);               // byte offset of column b inside an unswizzled tile row
        const int rd_row0 = 128 * b, rd_key0 = ((b >> 1) & 7) << 4;   // the row this lane reads in shot 1, and its swizzle key
        const int b1 = b <= 8 ? b : 8;                                // shot 2 holds rows 16..24: lanes 9..15 repeat row 24, result unused
        const int rd_row1 = 128 * b1, rd_key1 = ((b1 >> 1) & 7) << 4;
        const float* const win_l = s_win + b * kWinStride;
        const float* const w_l = s_w + l * kWStride;
        // the slots of the spectrum area that no bin is written to (8 in front of the first frame, 15 behind every frame's bin 200):
        // cleared every pass, because the area also serves as tile and parking space and the mel runs may touch them with zero weights
        float* const gap0 = parea + (l < 8 ? l : kPGuard + (l >> 3) * kPS + 201 + (l & 7));
        float* const gap1 = parea + kPGuard + (l >> 3) * kPS + (l < 8 ? 201 + l : 208 + (l & 7));
        float* const pw_a = parea + (2 * pr) * kPS + kPGuard;
        float* const pw_b = pw_a + kPS;

        float2 z[25];
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(win_l + 4 * j);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int a = 4 * j + e;
                if (a < 25) z[a] = make_float2(w4[e] * sample_value(g[a]), w4[e] * sample_value(g[HS > 0 ? a + HS : 25 + a]));
            }
        }

        // ---- 400-point complex FFT, first half: 25-point FFT over a, twiddle, transpose ------------
        fft25(z);
        // The twiddle table and the tile are both LDS, and the compiler cannot tell them apart: a twiddle read placed after a tile
        // write stays behind it and then costs a full LDS round trip before its product can be written.  So the reads run one group
        // of 8 ahead of the writes.  Rows 0..15 go to the tile now (shot 1); rows 16..24 wait, twiddled, in registers.
        {
            float2 tw_cur[8], tw_nxt[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) tw_cur[j] = s_tw[(1 + j) * 16 + b];
            *reinterpret_cast<float2*>(area + wr_col) = z[fft25_pos(0)];
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                if (g < 2) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) tw_nxt[j] = s_tw[(9 + 8 * g + j) * 16 + b];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int c = 1 + 8 * g + j;
                    const float2 v = cmul(z[fft25_pos(c)], tw_cur[j]);
                    if (c < 16) *reinterpret_cast<float2*>(area + 128 * c + (wr_col ^ (((c >> 1) & 7) << 4))) = v;
                    else z[fft25_pos(c)] = v;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) tw_cur[j] = tw_nxt[j];
            }
        }
        if (kEarlyFetch) {
            advance(row, oq);
            fetch(row, oq, pr, b);
        }
        wave_lds_fence();
        float2 u0[16], u1[16];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(area + rd_row0 + ((q << 4) ^ rd_key0));
            u0[2 * q] = make_float2(v.x, v.y);
            u0[2 * q + 1] = make_float2(v.z, v.w);
        }
        wave_lds_fence();
        // shot 2: rows 16..24 into the same place
#pragma unroll
        for (int c = 16; c < 25; ++c) *reinterpret_cast<float2*>(area + 128 * (c - 16) + (wr_col ^ ((((c - 16) >> 1) & 7) << 4))) = z[fft25_pos(c)];
        wave_lds_fence();
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(area + rd_row1 + ((q << 4) ^ rd_key1));
            u1[2 * q] = make_float2(v.x, v.y);
            u1[2 * q + 1] = make_float2(v.z, v.w);
        }
        wave_lds_fence();

        // ---- second half: 16-point FFTs of columns b and b+16 leave Z[c + 25 d] in registers.  The two real spectra separate
        //      from the pairs (k, 400 - k), k <= 200: the lane owns k = c + 25 d for d <= 7 (and k = 200 = column 0, d = 8); the
        //      partners 400 - k >= 200 are the d >= 8 half of some column, which is all that goes to LDS, at index k - 200 -------
        fft16(u0);
        __builtin_amdgcn_sched_barrier(0);      // one transform after the other: interleaved, their temporaries do not fit next to the prefetched samples
        fft16(u1);
#pragma unroll
        for (int d = 8; d < 16; ++d) zpark[b + 25 * d - 200] = u0[fft16_pos(d)];
        if (b <= 8) {
#pragma unroll
            for (int d = 8; d < 16; ++d) zpark[16 + b + 25 * d - 200] = u1[fft16_pos(d)];
        }
        if (b == 0) zpark[200] = u0[fft16_pos(0)];      // k = 0 pairs with itself (index 400 - 200); k = 200 finds itself at 200 - 200 (written above)
        wave_lds_fence();

        // ---- next octet's samples: requested here, with the last third of the pass to arrive (the spectra' separation, the mel projection,
        //      the logarithms and the stores).  Earlier, the 50 values in flight do not fit next to the transforms' registers in the 168 a
        //      third wave per SIMD leaves; the other two waves of the SIMD cover what is left of the wait.
        if (!kEarlyFetch) {
            advance(row, oq);
            fetch(row, oq, pr, b);
        }
        float p_a[17], p_b[17];
#pragma unroll
        for (int i = 0; i < 17; ++i) {
            // i < 9: column b, d = i (d = 8 is bin 200 + b: a real bin for b = 0 only); i >= 9: column 16 + b, d = i - 9 (b <= 8 only)
            const int k = i < 9 ? b + 25 * i : 16 + b + 25 * (i - 9);
            const float2 zk = i < 9 ? u0[fft16_pos(i)] : u1[fft16_pos(i - 9)];
            const int km = 200 - k;                                       // partner 400 - k, parked at (400 - k) - 200
            const float2 zm = zpark[km < 0 ? 0 : km];                     // (k > 200: lanes whose value is never stored)
            const float2 E = make_float2(zk.x + zm.x, zk.y - zm.y);     // X_A[k]
            const float2 D = make_float2(zk.x - zm.x, zk.y + zm.y);     // i * X_B[k]
            p_a[i] = E.x * E.x + E.y * E.y;
            p_b[i] = D.x * D.x + D.y * D.y;
        }
        wave_lds_fence();
        *gap0 = 0.f;
        *gap1 = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            pw_a[b + 25 * i] = p_a[i];
            pw_b[b + 25 * i] = p_b[i];
        }
        if (b == 0) {
            pw_a[200] = p_a[8];
            pw_b[200] = p_b[8];
        }
        if (b <= 8) {
#pragma unroll
            for (int i = 9; i < 17; ++i) {
                pw_a[16 + b + 25 * (i - 9)] = p_a[i];
                pw_b[16 + b + 25 * (i - 9)] = p_b[i];
            }
        }
        wave_lds_fence();

        // ---- banded mel on the matrix pipe (weights from the LDS table, four steps per 16-byte read), log, affine, store ------
        float v[4 * kStreams];
#pragma unroll
        for (int st = 0; st < kStreams; ++st) {
            f32x4 acc = f32x4{1.f, 1.f, 1.f, 1.f};
            if (kSteps[st] > 0) {
                acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s4 = 0; s4 < kSteps[st]; s4 += 4) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(w_l + kLdsOff[st] + s4);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (s4 + e < kSteps[st]) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(parea[poff[st] + s4 + e], w4[e], acc, 0, 0, 0);
                }
                if (SPLIT && st == kLastStream) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] += upper_half_value(acc[i]);      // blocks 8..15 hold the second halves of the runs
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) v[4 * st + i] = acc[i];
        }
        apply_log_affine_block(v, p.la);
        if (p.layout == MMFE_LAYOUT_FEAT_MAJOR && p.vec_store && t0 + 8 <= p.frames_valid) {
            // whole octet, 16-byte feature-major stores (the regular case): no per-frame predicates, one 64-bit scalar base per pass
            const int64_t obase = (int64_t)cur_row * p.n_mel * p.out_frames + t0;
#pragma unroll
            for (int st = 0; st < kStreams; ++st) {
                if (kSteps[st] == 0) continue;
                int d = dst[st];
                asm volatile("" : "+v"(d));      // (the address arithmetic below stays inside the pass, see the note on `l` above)
                if (d >= 0) {
                    if (p.rowmax != nullptr) {
                        lmax = fmaxf(fmaxf(lmax, v[4 * st]), fmaxf(v[4 * st + 1], fmaxf(v[4 * st + 2], v[4 * st + 3])));
                        lmin = fminf(fminf(lmin, v[4 * st]), fminf(v[4 * st + 1], fminf(v[4 * st + 2], v[4 * st + 3])));
                    }
                    store4(p, obase + ((d & 0xffff) * p.out_frames + 4 * (d >> 16)), v[4 * st], v[4 * st + 1], v[4 * st + 2], v[4 * st + 3]);
                }
            }
        } else {
#pragma unroll
            for (int st = 0; st < kStreams; ++st) {
                if (kSteps[st] == 0) continue;
                // lane = 4*blk + j holds filter j of the block's unit for its 4 frames in v[4*st .. 4*st+3]
                int d = dst[st];
                asm volatile("" : "+v"(d));
                const int m = d & 0xffff;
                const int tf = t0 + 4 * (d >> 16);          // first of the lane's 4 frames
                if (d >= 0) {
                    if (p.rowmax != nullptr) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            lmax = tf + i < p.frames_valid ? fmaxf(lmax, v[4 * st + i]) : lmax;
                            lmin = tf + i < p.frames_valid ? fminf(lmin, v[4 * st + i]) : lmin;
                        }
                    }
                    if (p.layout == MMFE_LAYOUT_FEAT_MAJOR) {
                        const int64_t o = ((int64_t)cur_row * p.n_mel + m) * p.out_frames + tf;
                        if (p.vec_store && tf + 4 <= p.frames_valid) {
                            store4(p, o, v[4 * st], v[4 * st + 1], v[4 * st + 2], v[4 * st + 3]);
                        } else {
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                if (tf + i < p.frames_valid) store1(p, o + i, v[4 * st + i]);
                        }
                    } else {
                        const int64_t o = ((int64_t)cur_row * p.out_frames + tf) * p.n_mel + m;
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (tf + i < p.frames_valid) store1(p, o + (int64_t)i * p.n_mel, v[4 * st + i]);
                    }
                }
            }
        }
        // the extremes go to memory when the wave leaves the row (once per row, not once per pass)
        if (p.rowmax != nullptr && row != cur_row) flush_extremes(cur_row);
        wave_lds_fence();
        if (p.rowmax != nullptr && row == cur_row && rounds_left == 1) flush_extremes(cur_row);     // the range ends inside this row
    }
}

// clamp_below_max on the finished features.  The affine map (x + shift) * scale with scale > 0 is monotone, so it
// commutes with the clamp: out = max(out, rowmax_out - c * scale).  Rows whose minimum already clears the threshold
// (any signal with less than c decades of dynamic range) are skipped, so the pass normally reads two ints per row.
// 16-bit outputs: rounding is monotone, so max(round(x), round(thr)) is the rounding of max(x, thr), the value a cast of the float32
// result holds; the extremes were tracked on the unrounded values.
__global__ void __launch_bounds__(256) clamp_rows_kernel(float* out, int out_dtype, const int* rowmax, const int* rowmin, int64_t per_row,
                                                         int n_rows, int chunks_per_row, float delta) {
    const int64_t n_work = (int64_t)n_rows * chunks_per_row;
    for (int64_t wk = blockIdx.x; wk < n_work; wk += gridDim.x) {
        const int row = (int)(wk / chunks_per_row);
        const int chunk = (int)(wk - (int64_t)row * chunks_per_row);
        const int omx = rowmax[row], omn = rowmin[row];
        const float mx = __int_as_float(omx >= 0 ? omx : omx ^ 0x7fffffff);
        const float mn = __int_as_float(omn >= 0 ? omn : omn ^ 0x7fffffff);
        const float thr = mx - delta;
        if (!(mn < thr)) continue;
        const int64_t len = (per_row + chunks_per_row - 1) / chunks_per_row;
        const int64_t lo = (int64_t)chunk * len, hi = lo + len < per_row ? lo + len : per_row;
        if (out_dtype == MMFE_OUT_F32) {
            float* o = out + (int64_t)row * per_row;
            for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) o[i] = fmaxf(o[i], thr);
        } else {
            unsigned short* o = reinterpret_cast<unsigned short*>(out) + (int64_t)row * per_row;
            const unsigned short tb = out_dtype == MMFE_OUT_BF16 ? bf16_bits(thr) : f16_bits(thr);
            const float tr = out_dtype == MMFE_OUT_BF16 ? __uint_as_float((unsigned)tb << 16) : (float)__builtin_bit_cast(_Float16, tb);
            for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
                const float x = out_dtype == MMFE_OUT_BF16 ? __uint_as_float((unsigned)o[i] << 16) : (float)__builtin_bit_cast(_Float16, o[i]);
                if (x < tr) o[i] = tb;
            }
        }
    }
}

// instantiated stream shapes (bins walked per stream, split flag), tightest first
struct MelShape { int s[kStreams]; bool split; };
constexpr MelShape kShapes[] = {
    {{5, 19, 17, 0}, true},       // 80 Slaney bands on 201 bins (16 kHz)
    {{8, 20, 20, 0}, true},       // 80 kaldi bands
    {{4, 8, 12, 24}, false},      // 128 bands
    {{12, 44, 0, 0}, false},      // 64 bands
    {{36, 32, 0, 0}, true},       // 40 bands
};

template <typename SMP, int HS, int S0, int S1, int S2, int S3, bool SPLIT>
int launch_variant(const Rfft400Args& a, int64_t octets, int max_blocks_per_cu /* tuning, 0 = no limit */, int* cus_out, hipStream_t stream) {
    const void* fn = reinterpret_cast<const void*>(&rfft400_kernel<SMP, HS, S0, S1, S2, S3, SPLIT>);
    const size_t lds = kTwBytes + kWinBytes + (size_t)64 * r400_wstride(S0, S1, S2, S3) * 4 + (size_t)kWaves * kWaveBytes;
    int blocks_per_cu = 1, cus = 256;
    int rc = kernel_prepare(fn, 64 * kWaves, lds, &blocks_per_cu, &cus);
    if (rc) return rc;
    if (max_blocks_per_cu >= 1 && max_blocks_per_cu < blocks_per_cu) blocks_per_cu = max_blocks_per_cu;
    *cus_out = cus;
    const int64_t want = (octets + kWaves - 1) / kWaves;
    const int grid = (int)std::min<int64_t>(want, (int64_t)cus * std::max(1, blocks_per_cu));
    hipLaunchKernelGGL((rfft400_kernel<SMP, HS, S0, S1, S2, S3, SPLIT>), dim3(grid), dim3(64 * kWaves), lds, stream, a);
    MMFE_HIP_CHECK(hipGetLastError());
    return MMFE_OK;
}

}  // namespace

bool rfft400_supports(const mmfe_spec& s) {
    if (const char* f = std::getenv("MMFE_FORCE_GENERIC")) {
        if (std::string(f) == "1") return false;
    }
    return s.n_fft == kN && s.n_mel > 0 && s.n_mel <= 128 && s.power == 2 && s.spec_scale == 1.0f && !s.window_f64 &&
           !s.remove_dc && s.preemph == 0.f && s.hop >= 1 && s.hop <= kMaxHop && !(s.clamp_below_max > 0.f && !(s.out_scale > 0.f)) &&
           (s.frames_mode == MMFE_FRAMES_CENTER || s.frames_mode == MMFE_FRAMES_SNIP);
}

// Builds the twiddle / window tables and the mel instruction streams.  Returns MMFE_EINVAL (with the
// error text set) when the filter bank fits none of the instantiated stream shapes; the caller then
// falls back to the any-length kernel.
int rfft400_prepare(Plan& plan) {
    const mmfe_spec& s = plan.spec;
    if (!rfft400_supports(s)) { set_error("rfft400: unsupported spec"); return MMFE_EINVAL; }
    for (int m = 0; m < s.n_mel; ++m) {
        const int st = plan.band_start[m], len = plan.band_len[m];
        if (len > 0 && (st < 0 || st + len > kBins)) {
            set_error("rfft400: mel band %d = [%d, %d) leaves the 201-bin spectrum", m, st, st + len);
            return MMFE_EINVAL;
        }
    }
    // ---- (filter block, frame group) units in order of frequency, 16 per stream ---------------------
    const int nfb = (s.n_mel + 3) / 4;
    const int units = 2 * nfb;
    const int n_streams = (units + 15) / 16;
    std::vector<int> kmin(nfb, 0), run(nfb, 0);
    for (int fb = 0; fb < nfb; ++fb) {
        int lo = 1 << 30, hi = -1;
        for (int j = 0; j < 4; ++j) {
            const int m = 4 * fb + j;
            if (m >= s.n_mel || plan.band_len[m] == 0) continue;
            lo = std::min(lo, (int)plan.band_start[m]);
            hi = std::max(hi, (int)(plan.band_start[m] + plan.band_len[m] - 1));
        }
        if (hi >= 0) { kmin[fb] = lo; run[fb] = hi - lo + 1; }
    }
    const int last_units = units - 16 * (n_streams - 1);
    const bool split = last_units <= 8;
    int steps[kStreams] = {0, 0, 0, 0};
    for (int u = 0; u < units; ++u) {
        const int st = u / 16, fb = u >> 1;
        const int need = (split && st == n_streams - 1) ? (run[fb] + 1) / 2 : run[fb];
        steps[st] = std::max(steps[st], std::max(need, 1));
    }
    int pick = -1;
    for (size_t i = 0; i < sizeof(kShapes) / sizeof(kShapes[0]); ++i) {
        bool ok = kShapes[i].split == split;
        for (int st = 0; st < kStreams; ++st) {
            if ((kShapes[i].s[st] > 0) != (steps[st] > 0) || kShapes[i].s[st] < steps[st]) ok = false;
        }
        if (ok) { pick = (int)i; break; }
    }
    if (pick < 0) {
        set_error("rfft400: no kernel variant for mel stream shape (%d, %d, %d, %d)%s", steps[0], steps[1], steps[2], steps[3],
                  split ? " split" : "");
        return MMFE_EINVAL;
    }
    const MelShape& shape = kShapes[pick];
    int step_off[kStreams + 1] = {0, 0, 0, 0, 0};
    for (int st = 0; st < kStreams; ++st) step_off[st + 1] = step_off[st] + shape.s[st];
    std::vector<int> boff(s.n_mel + 1, 0);
    for (int m = 0; m < s.n_mel; ++m) boff[m + 1] = boff[m] + plan.band_len[m];

    // Bank conflicts of the matrix-pipe operand reads.  Lane (block, j) of a stream reads bin ks + step of frame 4g + j; the 8 frames
    // of an octet sit on 8 different residues of bank / 8 (kPS = 24 mod 64), so two filter blocks of a stream collide - all their
    // lanes, on every step - exactly when their first bins agree mod 8.  A run may start up to 7 bins early (zero weights, the
    // guard in front of bin 0 covers it) as long as it still ends within the stream's steps: the shifts below use that slack to
    // give the filter blocks of a stream distinct residues where possible.
    std::vector<int> early(nfb, 0);
    {
        const char* e = std::getenv("MMFE_R400_DECONFLICT");      // tuning knob, read once per plan: 0 keeps every run at its first bin
        const bool enabled = e == nullptr || std::atoi(e) != 0;
        for (int st = 0; enabled && st < n_streams; ++st) {
            // a split stream walks each run in two halves S bins apart: its filter blocks hold two residues each
            const bool split_here = split && st == n_streams - 1;
            const int S = shape.s[st];
            std::vector<int> fbs;
            for (int u = 16 * st; u < std::min(units, 16 * st + (split_here ? 8 : 16)); u += 2) fbs.push_back(u >> 1);
            const int n = (int)fbs.size();
            auto slack_of = [&](int fb) { return std::max(0, std::min(7, (split_here ? 2 * S : S) - std::max(run[fb], 1))); };
            std::sort(fbs.begin(), fbs.end(), [&](int a, int b) { return slack_of(a) < slack_of(b); });   // least slack first
            std::vector<int> pickd(n, 0), best;
            int best_cost = 1 << 30;
            // depth-first over the shifts: cost = residues shared with an earlier filter block, then the total shift
            std::function<void(int, unsigned, int, int)> go = [&](int i, unsigned used, int clashes, int total) {
                if (clashes * 1000 + total >= best_cost) return;
                if (i == n) { best_cost = clashes * 1000 + total; best = pickd; return; }
                const int fb = fbs[i];
                for (int d = 0; d <= slack_of(fb); ++d) {
                    const int r0 = ((kmin[fb] - d) % 8 + 8) % 8, r1 = ((kmin[fb] - d + S) % 8 + 8) % 8;
                    unsigned mine = 1u << r0;
                    int c = (used >> r0) & 1u ? 1 : 0;
                    if (split_here && r1 != r0) { c += (used >> r1) & 1u ? 1 : 0; mine |= 1u << r1; }
                    else if (split_here) c += 1;
                    pickd[i] = d;
                    go(i + 1, used | mine, clashes + c, total + d);
                }
            };
            go(0, 0u, 0, 0);
            for (int i = 0; i < n; ++i) early[fbs[i]] = best.empty() ? 0 : best[i];
        }
    }
    std::vector<int32_t> poff(kStreams * 64, kPGuard), dst(kStreams * 64, -1);
    std::vector<float> w((size_t)std::max(step_off[kStreams], 1) * 64, 0.f);
    for (int st = 0; st < n_streams; ++st) {
        const bool split_here = split && st == n_streams - 1;
        for (int blk = 0; blk < 16; ++blk) {
            const int part = split_here ? blk / 8 : 0;
            const int u = 16 * st + (split_here ? blk % 8 : blk);
            if (u >= units) continue;
            const int fb = u >> 1, g = u & 1;
            // a split stream walks [kmin, kmin + S) in blocks 0..7 and [kmin + S, kmin + 2S) in blocks 8..15
            const int ks = kmin[fb] - early[fb] + part * shape.s[st];
            for (int j = 0; j < 4; ++j) {
                const int lane = 4 * blk + j;
                poff[st * 64 + lane] = (4 * g + j) * kPS + kPGuard + ks;      // operand A: frame 4g+j of the octet
                const int m = 4 * fb + j;                                      // operand B / result: filter m
                if (m >= s.n_mel) continue;
                if (part == 0) dst[st * 64 + lane] = m | (g << 16);
                for (int sp = 0; sp < shape.s[st]; ++sp) {
                    const int rel = ks + sp - plan.band_start[m];
                    if (rel >= 0 && rel < plan.band_len[m]) w[(size_t)(step_off[st] + sp) * 64 + lane] = plan.band_weights[boff[m] + rel];
                }
            }
        }
    }
    plan.r400_shape = pick;
    std::vector<float> win(kN, 0.f);
    for (int i = 0; i < s.win_len; ++i) win[s.win_off + i] = 0.5f * plan.window[i];
    std::vector<float> tw(2 * 25 * 16);
    const double pi = 3.14159265358979323846;
    for (int c = 0; c < 25; ++c)
        for (int b = 0; b < 16; ++b) {
            tw[2 * (c * 16 + b)] = (float)std::cos(2.0 * pi * (b * c) / 400.0);
            tw[2 * (c * 16 + b) + 1] = (float)-std::sin(2.0 * pi * (b * c) / 400.0);
        }
    int rc;
    if ((rc = upload(&plan.d_window, win))) return rc;
    if ((rc = upload(&plan.d_tw_a, tw))) return rc;
    if ((rc = upload(&plan.d_mel4_kstart, poff))) return rc;
    if ((rc = upload(&plan.d_r400_dst, dst))) return rc;
    if ((rc = upload(&plan.d_mel4_w, w))) return rc;
    if (const char* e = std::getenv("MMFE_R400_BLOCKS_PER_CU")) plan.tune_blocks_per_cu = std::atoi(e);     // tuning knob, read once per plan
    return MMFE_OK;
}

int rfft400_launch(const Plan& plan, const RowSource& src, int64_t n_rows, int64_t n_samples, int64_t n_frames_natural,
                   int64_t out_frames, float* out, const ClampScratch& cs, hipStream_t stream) {
    const mmfe_spec& s = plan.spec;
    const int64_t frames_valid = std::min(n_frames_natural, out_frames);
    if (n_rows <= 0 || frames_valid <= 0) return MMFE_OK;
    const int64_t octets = n_rows * ((frames_valid + 7) / 8);
    if (octets >= (int64_t)1 << 31 || n_rows >= (int64_t)1 << 31 || n_samples >= (int64_t)1 << 30) {
        set_error("rfft400: batch too large for one launch");
        return MMFE_EINVAL;
    }
    const bool clamp = s.clamp_below_max > 0.f;
    Rfft400Args a{};
    a.src = src;
    a.out = out;
    a.rowmax = nullptr;
    a.n_rows = (int)n_rows;
    a.n_samples = (int)n_samples;
    a.frames_valid = (int)frames_valid;
    a.out_frames = (int)out_frames;
    a.hop = s.hop;
    a.shift = s.frames_mode == MMFE_FRAMES_CENTER ? -kN / 2 : -s.win_off;
    a.pad_mode = s.pad_mode;
    a.n_mel = s.n_mel;
    a.layout = s.layout;
    a.out_dtype = s.out_dtype;
    a.vec_store = (out_frames % 4 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0 && out_frames * (int64_t)s.n_mel < ((int64_t)1 << 31)) ? 1 : 0;
    a.la = make_log_affine(s);
    a.window = plan.d_window;
    a.tw = reinterpret_cast<const float2*>(plan.d_tw_a);
    a.mel_poff = plan.d_mel4_kstart;
    a.mel_dst = plan.d_r400_dst;
    a.mel_w = plan.d_mel4_w;
    int cus = 256;
    if (clamp) {
        int rc = rowmax_init_launch(cs, n_rows, stream);
        if (rc) return rc;
        a.rowmax = reinterpret_cast<int*>(cs.base);
        a.rowmin = a.rowmax + cs.capacity;
    }
    const bool pcm = s.in_dtype == MMFE_IN_S16;
    int rc = MMFE_EINVAL;
    bool found = false;
#define MMFE_R400_SHAPE(IDX, A, B, C, D, SP)                                                                     \
    if (plan.r400_shape == IDX) {                                                                                \
        found = true;                                                                                            \
        if (s.hop == 160) {   /* the long-utterance configuration: frames 10 x 16 samples apart share their loads */                \
            if (pcm) rc = launch_variant<int16_t, 10, A, B, C, D, SP>(a, octets, plan.tune_blocks_per_cu, &cus, stream); \
            else rc = launch_variant<float, 10, A, B, C, D, SP>(a, octets, plan.tune_blocks_per_cu, &cus, stream);       \
        } else if (pcm) rc = launch_variant<int16_t, 0, A, B, C, D, SP>(a, octets, plan.tune_blocks_per_cu, &cus, stream); \
        else rc = launch_variant<float, 0, A, B, C, D, SP>(a, octets, plan.tune_blocks_per_cu, &cus, stream);   \
    }
    MMFE_R400_SHAPE(0, 5, 19, 17, 0, true)
    MMFE_R400_SHAPE(1, 8, 20, 20, 0, true)
    MMFE_R400_SHAPE(2, 4, 8, 12, 24, false)
    MMFE_R400_SHAPE(3, 12, 44, 0, 0, false)
    MMFE_R400_SHAPE(4, 36, 32, 0, 0, true)
#undef MMFE_R400_SHAPE
    if (!found) { set_error("rfft400: plan has no kernel variant"); return MMFE_EINVAL; }
    if (rc) return rc;
    if (clamp) {
        const int64_t per_row = out_frames * (int64_t)s.n_mel;
        const int chunks = (int)std::max<int64_t>(1, std::min<int64_t>(64, per_row / 4096));
        const int grid = (int)std::min<int64_t>(n_rows * chunks, (int64_t)cus * 8);
        hipLaunchKernelGGL(clamp_rows_kernel, dim3(grid), dim3(256), 0, stream, out, s.out_dtype, a.rowmax, a.rowmin, per_row, (int)n_rows, chunks,
                           s.clamp_below_max * s.out_scale);
        MMFE_HIP_CHECK(hipGetLastError());
    }
    return rc;
}

}  // namespace mmfe
