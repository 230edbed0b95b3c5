// Fused log-mel front-end for n_fft = 400 (25 ms frames at 16 kHz) on gfx950: framing with centre
// reflect/zero padding or edge snipping -> window -> 400-point real FFT -> |X|^2 -> banded mel ->
// log -> affine, spectra never leave the CU.  This is the long-utterance configuration
// (n_fft 400, hop 160, 80 mel bands, 30 s utterances); its parameters are those of
// transformers' WhisperFeatureExtractor, the reference itself has no n_fft = 400 caller.
//
// Mapping: 16 lanes own a PAIR of consecutive frames (A, B); a wave64 carries 8 consecutive frames
// of one row per pass.
//   * the wave's samples (7*hop + 400 of them) arrive by coalesced 16-byte loads, one pass ahead,
//     and are laid out in LDS once, so the 2.5x overlap of neighbouring frames never re-reads HBM;
//   * z[n] = w[n]*(A[n] + i*B[n]) is transformed by ONE 400-point complex FFT, split 25 x 16:
//     lane b runs a register-resident 25-point FFT (radix 5 x 5) over z[16a+b], twiddles by
//     W400^(b*c), the 16 lanes transpose through a padded LDS tile, and each lane runs the 16-point
//     FFTs of columns c = b and c = b+16 (the second only for b <= 8), leaving Z[c + 25d];
//   * the two real spectra separate as X_A[k] = (Z[k] + conj Z[400-k])/2 and
//     X_B[k] = (Z[k] - conj Z[400-k])/(2i); the window is pre-halved, so |X_A|^2 and |X_B|^2 are two
//     sums of squares per bin, written to LDS;
//   * banded mel on the matrix pipe: v_mfma_f32_4x4x1 runs 16 independent blocks of (4 frames) x
//     (4 filters); a block walks ITS OWN run of bins, one bin per instruction, weights resident in
//     VGPRs.  The (filter block, frame group) units are spread over up to four instruction streams
//     in order of frequency, so the short low-frequency runs do not wait for the long high ones; a
//     last stream that is at most half full splits its runs in two and adds the halves.
// Two waves per SIMD (LDS-bound occupancy) leave 256 VGPRs per lane: the window, the mel weights
// and the prefetched samples all stay in registers.
#include "mmfe_internal.h"
#include "fft_regs.h"
#include "log_affine.h"

#include <algorithm>
#include <functional>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

namespace mmfe {

namespace {

constexpr int kN = 400;
constexpr int kBins = 201;
#ifndef MMFE_R400_WAVES
#define MMFE_R400_WAVES 4
#endif
constexpr int kWaves = MMFE_R400_WAVES;             // waves per block
constexpr int kMaxHop = 160;
constexpr int kMaxSpan = 7 * kMaxHop + kN;          // samples under one wave pass: 1520
[[maybe_unused]] constexpr int kPF = (kMaxSpan / 4 + 63) / 64;       // prefetch registers (16-byte) per lane: 6
constexpr int kTileStride = 18;                     // complex numbers per transpose-tile row (16 + padding, keeps rows 16-byte aligned)
constexpr int kPairBytes = 25 * kTileStride * 8 + 112;   // 3712: transpose tile / natural-order spectrum of one frame pair; = 32 dwords
                                                         // mod 64, so the two pairs of a 32-lane half sit on disjoint LDS banks
constexpr int kWaveBytes = 4 * kPairBytes;          // 14848: one LDS work area per wave, reused by every phase
constexpr int kPS = 216;                            // floats between the power spectra of consecutive frames (= 24 mod 64: 8 frames on 8 bank groups)
constexpr int kPGuard = 8;                          // floats in front of bin 0 (a bin run may start below 0)
constexpr int kTwBytes = 25 * 16 * 8;               // W400^(b*c) table, [c][b]
constexpr int kStreams = 4;

#ifndef MMFE_R400_DIRECT
#define MMFE_R400_DIRECT 1      // 1: every lane loads its own strided samples (n = 16a + b of frames A and B) straight from global memory, one
                                //    pass ahead; the 2.5x overlap of neighbouring frames is served by L1/L2.  0: the wave's samples are staged in LDS
                                //    by 16-byte loads and gathered from there
#endif
#ifndef MMFE_R400_HALFPARK
#define MMFE_R400_HALFPARK 1    // 1: a lane separates the bins k <= 200 of its own FFT columns, Z[k] from registers; only Z[201..399] is parked
                                //    in LDS for the partners Z[400 - k].  0: all of Z is parked in natural order and both members are read back
#endif

#ifndef MMFE_R400_TWGROUP
#define MMFE_R400_TWGROUP 8
#endif
#ifndef MMFE_R400_RANGES
#define MMFE_R400_RANGES 1      // 1: a block works through one contiguous range of octets, so a wave stays inside a row for many passes and the
                                //    per-row extremes of clamp_below_max are reduced across the wave once per row instead of once per pass.
                                //    0: blocks take octets round-robin over the whole batch
#endif

#ifndef MMFE_R400_PIN
#define MMFE_R400_PIN 1
#endif
#ifndef MMFE_R400_ROTATE
#define MMFE_R400_ROTATE 1      // direct-load form: see "Rotated pass" in the kernel
#endif
#ifndef MMFE_R400_POOL
#define MMFE_R400_POOL 0        // 1 (feature-major 16-byte stores, blocks of 4 waves): the block pools a round's 32 frames of every filter in an LDS tile
#endif                          //    and stores whole 128-byte runs of each filter row (two block barriers per round)
constexpr int kPoolStride = 36;  // floats per filter row of that tile: 32 frames + 4
#ifndef MMFE_R400_NTLOAD
#define MMFE_R400_NTLOAD 0
#endif
#if MMFE_R400_NTLOAD
#define MMFE_R400_LOAD(ptr) __builtin_nontemporal_load(ptr)
#else
#define MMFE_R400_LOAD(ptr) (*(ptr))
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
    int pool;                  // feature-major 16-byte stores go through the block's LDS tile (MMFE_R400_POOL)
    LogAffine la;
    const float* window;       // 400 floats: window pre-halved at win_off, zeros elsewhere
    const float2* tw;          // [25][16]: W400^(b*c)
    const int32_t* mel_poff;   // [kStreams][64]: float offset of the lane's bin run in the wave's spectrum area
    const int32_t* mel_dst;    // [kStreams][64]: filter | frame group << 16, or -1 for an idle lane
    const float* mel_w;        // [total steps][64], step-major
    unsigned long long* stamps;  // diagnostic build only
};

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

// Position of staged sample i (relative to the wave's first sample) in the LDS staging buffer.  With
// hop 160 the four frame pairs of a wave start 320 floats apart, i.e. on the same LDS banks; 16
// floats of padding per 320 samples put them on four different bank groups.  A run of 16 samples
// that starts at a multiple of 16 never straddles a padding gap.
template <int HOP>
__device__ __forceinline__ int stage_pos(int i) {
    if constexpr (HOP == 160) return i + 16 * (i / 320);
    else return i;
}

// Four consecutive samples as floats: one 16-byte load of floats, or one 8-byte load of 16-bit PCM decoded as s / 32768.
__device__ __forceinline__ float4 load4(const float* sp) { return *reinterpret_cast<const float4*>(sp); }
__device__ __forceinline__ float4 load4(const int16_t* sp) {
    const uint2 r = *reinterpret_cast<const uint2*>(sp);
    constexpr float k = 1.0f / 32768.0f;
    return make_float4((float)(int16_t)(r.x & 0xffffu) * k, (float)((int32_t)r.x >> 16) * k,
                       (float)(int16_t)(r.y & 0xffffu) * k, (float)((int32_t)r.y >> 16) * k);
}

// Stored element i of a row of n under the plan's padding rule (i may lie outside the row), undecoded.
template <typename S>
__device__ __forceinline__ S fetch_raw(const S* row, int i, int n, int pad_mode) {
    if (i >= 0 && i < n) return row[i];
    if (n <= 0 || pad_mode != MMFE_PAD_REFLECT) return S(0);
    if (n == 1) return row[0];
    const int period = 2 * (n - 1);
    int m = i % period;
    if (m < 0) m += period;
    return row[m < n ? m : period - m];
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

template <typename SMP /* input element: float or int16_t PCM */, int HOP /* 160: the two frames of a pair share their loads (and the staging layout of the LDS-staged form is specialised); 0: any hop <= 160 */,
          bool VEC /* row starts aligned to 4 samples and hop % 4 == 0 */,
          int S0, int S1, int S2, int S3 /* bins walked by the four mel instruction streams */, bool SPLIT /* last stream: runs split in two halves */>
__global__ void __launch_bounds__(64 * kWaves, 2 * kWaves / 4 > 2 ? 2 : 2) rfft400_kernel(const Rfft400Args p) {
    constexpr int kSteps[kStreams] = {S0, S1, S2, S3};
    constexpr int kStepOff[kStreams] = {0, S0, S0 + S1, S0 + S1 + S2};
    constexpr int kTotalSteps = S0 + S1 + S2 + S3;
    constexpr int kLastStream = S3 > 0 ? 3 : (S2 > 0 ? 2 : (S1 > 0 ? 1 : 0));
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* s_tw = reinterpret_cast<float2*>(smem);
    for (int i = threadIdx.x; i < 25 * 16; i += blockDim.x) s_tw[i] = p.tw[i];
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pr = lane >> 4;      // frame pair within the wave
    const int b = lane & 15;       // lane within the pair's group
    char* wave_area = smem + kTwBytes + wave * kWaveBytes;
#if !MMFE_R400_DIRECT
    float* stage = reinterpret_cast<float*>(wave_area);
#endif
    float2* tile = reinterpret_cast<float2*>(wave_area + pr * kPairBytes);
    float* parea = reinterpret_cast<float*>(wave_area);
    float* otile = reinterpret_cast<float*>(smem + kTwBytes + kWaves * kWaveBytes);      // [n_mel][kPoolStride], pooled stores only
    int* ometa = reinterpret_cast<int*>(otile + p.n_mel * kPoolStride);                  // per wave: row of its octet (-1: nothing to store), first frame

    // Bin runs of the mel stage may read a few slots that no phase writes (tile padding, staging gaps) with
    // zero weights; start from zeros so that those slots can never hold a non-finite bit pattern.
    for (int i = lane; i < kWaveBytes / 16; i += 64) reinterpret_cast<float4*>(wave_area)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    wave_lds_fence();

    const int hop = HOP > 0 ? HOP : p.hop;
#if !MMFE_R400_DIRECT
    const int span4 = (7 * hop + kN + 3) & ~3;      // samples staged per pass, rounded up to whole 16-byte pieces
#endif

    // ---- per-lane constants, resident for the life of the kernel -----------------------------------
    float win[25];
#pragma unroll
    for (int a = 0; a < 25; ++a) win[a] = p.window[16 * a + b];
    float wreg[kTotalSteps];
#pragma unroll
    for (int s = 0; s < kTotalSteps; ++s) wreg[s] = p.mel_w[s * 64 + lane];
    const float* pa[kStreams];
    int dst[kStreams];
#pragma unroll
    for (int st = 0; st < kStreams; ++st) {
        pa[st] = parea + (kSteps[st] > 0 ? p.mel_poff[st * 64 + lane] : 0);
        dst[st] = kSteps[st] > 0 ? p.mel_dst[st * 64 + lane] : -1;
    }
    // ---- work distribution: octet = 8 consecutive frames of one row; a block takes kWaves consecutive
    //      octets per round, so its waves stream one contiguous region and fill whole output lines ------
    const unsigned opr = (unsigned)(p.frames_valid + 7) >> 3;      // octets per row
#if MMFE_R400_RANGES
    // rounds of kWaves consecutive octets; block b owns rounds [b * per_block, (b + 1) * per_block)
    const unsigned n_rounds = ((unsigned)p.n_rows * opr + kWaves - 1) / kWaves;
    const unsigned per_block = (n_rounds + gridDim.x - 1) / gridDim.x;
    const unsigned round0 = blockIdx.x * per_block;
    unsigned rounds_left = round0 < n_rounds ? (n_rounds - round0 < per_block ? n_rounds - round0 : per_block) : 0;
    const unsigned ostride = kWaves;
    const unsigned oct0 = round0 * kWaves + wave;
#else
    const unsigned ostride = gridDim.x * kWaves;
    const unsigned oct0 = blockIdx.x * kWaves + wave;
#endif
    const unsigned d_row = ostride / opr, d_o = ostride - d_row * opr;
    unsigned row = oct0 / opr, oq = oct0 - row * opr;               // wave-uniform
    auto advance = [&](unsigned& r, unsigned& o) {
        r += d_row;
        o += d_o;
        if (o >= opr) { o -= opr; r += 1; }
    };

#if MMFE_R400_DIRECT
    // the samples of the lane's two frames under octet (r, o): n = 16a + b of frame A = 2*pr and of frame B = A + 1
    // (kept as loaded; 16-bit PCM is decoded where the values are used, so that nothing waits for the loads here).
    // With a hop of 16 * HS (HOP = 160: HS = 10), frame B's sample a IS frame A's sample a + HS of the same lane: one run
    // g[0 .. 24 + HS] serves both frames, 35 loads per lane and pass instead of 50 (-7.5 % on the whole call at hop 160).
    constexpr int HS = HOP / 16;
    constexpr int kNG = HS > 0 ? 25 + HS : 50;
    SMP g[kNG];
    auto fetch = [&](unsigned r, unsigned o) {
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
                for (int a = 0; a < kNG; ++a) g[a] = MMFE_R400_LOAD(qa + 16 * a);
            } else {
                const SMP* qb = qa + hop;
#pragma unroll
                for (int a = 0; a < 25; ++a) { g[a] = qa[16 * a]; g[25 + a] = qb[16 * a]; }
            }
        } else {    // the octet touches an end of the row: padding rule per sample
#pragma unroll
            for (int a = 0; a < (HS > 0 ? kNG : 25); ++a) {
                g[a] = fetch_raw(rp, ia + 16 * a, n_row, p.pad_mode);
                if (HS == 0) g[25 + a] = fetch_raw(rp, ia + hop + 16 * a, n_row, p.pad_mode);
            }
        }
    };
    fetch(row, oq);
#else
    // the samples under octet (r, o): lane holds pieces lane, lane+64, ... of 4 consecutive samples
    float4 pf[kPF];
    auto fetch = [&](unsigned r, unsigned o) {
        const unsigned rr = r < (unsigned)p.n_rows ? r : 0;
        const SMP* rp = row_ptr_as<SMP>(p.src, rr);
        const int n_row = row_samples(p.src, rr, p.n_samples);
        const int first = (int)o * 8 * hop + p.shift;
        const bool interior = first >= 0 && first + span4 <= n_row;     // wave-uniform
        if (interior) {
#pragma unroll
            for (int q = 0; q < kPF; ++q) {
                const int i = 4 * (lane + 64 * q);
                const SMP* sp = rp + first + (i < span4 ? i : 0);
                if (VEC) pf[q] = load4(sp);
                else pf[q] = make_float4(sample_value(sp[0]), sample_value(sp[1]), sample_value(sp[2]), sample_value(sp[3]));
            }
        } else {    // the octet touches an end of the row: padding rule per sample
#pragma unroll
            for (int q = 0; q < kPF; ++q) {
                const int i = 4 * (lane + 64 * q);
                const int s0 = first + (i < span4 ? i : 0);
                pf[q] = make_float4(fetch_sample(rp, s0, n_row, p.pad_mode), fetch_sample(rp, s0 + 1, n_row, p.pad_mode),
                                    fetch_sample(rp, s0 + 2, n_row, p.pad_mode), fetch_sample(rp, s0 + 3, n_row, p.pad_mode));
            }
        }
    };
    fetch(row, oq);

    // sample n = 16a + b of frames A = 2*pr and B = 2*pr + 1 in the staging buffer
    const float* st_a = stage + stage_pos<HOP>(2 * pr * hop) + b;
    const float* st_b = HOP == 160 ? st_a + 160 : stage + (2 * pr + 1) * hop + b;
#endif
    float* const pw_a = parea + (2 * pr) * kPS + kPGuard;
    float* const pw_b = pw_a + kPS;

#ifdef MMFE_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev = 0;
#endif
    // feature-major 16-byte stores: the lane's part of the address that does not change from pass to pass
    int coff[kStreams];
#pragma unroll
    for (int st = 0; st < kStreams; ++st) coff[st] = dst[st] >= 0 ? (dst[st] & 0xffff) * p.out_frames + 4 * (dst[st] >> 16) : 0;
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
#if MMFE_R400_DIRECT && MMFE_R400_ROTATE
    // Rotated pass: the window products z of the octet a pass transforms are formed at the END of the pass before (from the samples
    // requested a whole pass earlier), right in front of the request for the octet after it and in front of that pass's stores.  The
    // one wait of a pass for global memory then stands where the loads it covers are a pass old and the previous stores long done;
    // with the products at the top of a pass the same wait also covers the stores issued a moment before.
    float2 z[25];
    auto window_products = [&]() {
#pragma unroll
        for (int a = 0; a < 25; ++a) z[a] = make_float2(win[a] * sample_value(g[a]), win[a] * sample_value(g[HS > 0 ? a + HS : 25 + a]));
        // pinned in front of the next request: left free, the optimiser sinks the products below it (towards their use), and because
        // loads return in order the wait for the old samples becomes a wait for the new ones as well
#pragma unroll
        for (int a = 0; a < 25; ++a) asm volatile("" : "+v"(z[a].x), "+v"(z[a].y));
    };
    unsigned zrow = row, zoq = oq;          // the octet whose products are in z; (row, oq) is the octet whose samples are in flight
    window_products();
    advance(row, oq);
    fetch(row, oq);
    // pooled stores: every wave of the block runs every round (two block barriers per round); a wave whose octet lies beyond the batch
    // works on row 0's samples like any masked slot and stores nothing
    for (; rounds_left > 0 && (p.pool || zrow < (unsigned)p.n_rows); --rounds_left) {
        const int t0 = (int)zoq * 8;          // first frame of the octet
        const unsigned cur_row = zrow;
        const bool live = zrow < (unsigned)p.n_rows;
#else
#if MMFE_R400_RANGES
    for (; rounds_left > 0 && row < (unsigned)p.n_rows; --rounds_left) {
#else
    while (row < (unsigned)p.n_rows) {
#endif
        MMFE_STAMP(0);
        const int t0 = (int)oq * 8;          // first frame of the octet
        const unsigned cur_row = row;
        const bool live = true;

#if MMFE_R400_DIRECT
        float2 z[25];
#pragma unroll
        for (int a = 0; a < 25; ++a) z[a] = make_float2(win[a] * sample_value(g[a]), win[a] * sample_value(g[HS > 0 ? a + HS : 25 + a]));
#if MMFE_R400_PIN
        // The products are pinned HERE, in front of the request for the next octet's samples.  Left free, the optimiser sinks them below
        // that request (towards their use in the transform): the first use of the samples loaded one pass ago then stands behind the new
        // loads in program order, and because loads return in order the wait for the old ones becomes a wait for ALL of them - the
        // "prefetch" is waited for right after it is issued, every pass.
#pragma unroll
        for (int a = 0; a < 25; ++a) asm volatile("" : "+v"(z[a].x), "+v"(z[a].y));
#endif
#else
        // ---- stage the wave's samples -----------------------------------------------------------
#pragma unroll
        for (int q = 0; q < kPF; ++q) {
            const int i = 4 * (lane + 64 * q);
            if (i < span4) *reinterpret_cast<float4*>(stage + stage_pos<HOP>(i)) = pf[q];
        }
        wave_lds_fence();
        float2 z[25];
#pragma unroll
        for (int a = 0; a < 25; ++a) {
            // hop 160: frame A crosses the padding gap at sample 320 of its pair (a >= 20), frame B at its sample 160 (a >= 10)
            const int off_a = 16 * a + (HOP == 160 && a >= 20 ? 16 : 0);
            const int off_b = 16 * a + (HOP == 160 && a >= 10 ? 16 : 0);
            z[a] = make_float2(win[a] * st_a[off_a], win[a] * st_b[off_b]);
        }
        wave_lds_fence();

#endif

        MMFE_STAMP(1);
        // ---- next octet's samples: in flight under the rest of this pass ----------------------------
        advance(row, oq);
        fetch(row, oq);

#endif
        // ---- 400-point complex FFT, first half: 25-point FFT over a, twiddle, transpose ------------
        fft25(z);
        // The twiddle table and the tile are both LDS, and the compiler cannot tell them apart: a twiddle read
        // placed after a tile write stays behind it and then costs a full LDS round trip before its product can be
        // written.  So the reads run one group of MMFE_R400_TWGROUP ahead of the writes: 24 / group round trips instead of 24.
        {
            constexpr int G = MMFE_R400_TWGROUP;      // twiddles fetched per LDS round trip (24 = 3 x 8 = 6 x 4)
            float2 tw_cur[G], tw_nxt[G];
#pragma unroll
            for (int j = 0; j < G; ++j) tw_cur[j] = s_tw[(1 + j) * 16 + b];
            tile[b] = z[fft25_pos(0)];
#pragma unroll
            for (int g = 0; g < 24 / G; ++g) {
                if (g + 1 < 24 / G) {
#pragma unroll
                    for (int j = 0; j < G; ++j) tw_nxt[j] = s_tw[(1 + G * (g + 1) + j) * 16 + b];
                }
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    const int c = 1 + G * g + j;
                    tile[c * kTileStride + b] = cmul(z[fft25_pos(c)], tw_cur[j]);
                }
#pragma unroll
                for (int j = 0; j < G; ++j) tw_cur[j] = tw_nxt[j];
            }
        }
        wave_lds_fence();
        MMFE_STAMP(2);
        float2 u0[16], u1[16];
        {
            const float2* r0 = tile + b * kTileStride;
            const float2* r1 = tile + (b <= 8 ? 16 + b : 24) * kTileStride;     // lanes 9..15 repeat column 24, result unused
#pragma unroll
            for (int bb = 0; bb < 16; ++bb) { u0[bb] = r0[bb]; u1[bb] = r1[bb]; }
        }
        wave_lds_fence();

        MMFE_STAMP(3);
#if MMFE_R400_HALFPARK
        // ---- second half: 16-point FFTs of columns b and b+16 leave Z[c + 25 d] in registers.  The two real spectra separate
        //      from the pairs (k, 400 - k), k <= 200: the lane owns k = c + 25 d for d <= 7 (and k = 200 = column 0, d = 8); the
        //      partners 400 - k > 200 are the d >= 8 half of some column, which is all that goes to LDS -------------------------
        fft16(u0);
        fft16(u1);
#pragma unroll
        for (int d = 8; d < 16; ++d) tile[b + 25 * d] = u0[fft16_pos(d)];
        if (b <= 8) {
#pragma unroll
            for (int d = 8; d < 16; ++d) tile[16 + b + 25 * d] = u1[fft16_pos(d)];
        }
        if (b == 0) tile[400] = u0[fft16_pos(0)];      // k = 0 pairs with itself; k = 200 finds itself at 400 - 200 (written above)
        wave_lds_fence();

        MMFE_STAMP(4);
        float p_a[17], p_b[17];
#pragma unroll
        for (int i = 0; i < 17; ++i) {
            // i < 9: column b, d = i (d = 8 is bin 200 + b: a real bin for b = 0 only); i >= 9: column 16 + b, d = i - 9 (b <= 8 only)
            const int k = i < 9 ? b + 25 * i : 16 + b + 25 * (i - 9);
            const float2 zk = i < 9 ? u0[fft16_pos(i)] : u1[fft16_pos(i - 9)];
            const float2 zm = tile[400 - k];
            const float2 E = make_float2(zk.x + zm.x, zk.y - zm.y);     // X_A[k]
            const float2 D = make_float2(zk.x - zm.x, zk.y + zm.y);     // i * X_B[k]
            p_a[i] = E.x * E.x + E.y * E.y;
            p_b[i] = D.x * D.x + D.y * D.y;
        }
        wave_lds_fence();
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
#else
        // ---- second half: 16-point FFTs of columns b and b+16; Z[c + 25 d] parked in natural order ----
        fft16(u0);
        fft16(u1);
#pragma unroll
        for (int d = 0; d < 16; ++d) tile[b + 25 * d] = u0[fft16_pos(d)];
        if (b <= 8) {
#pragma unroll
            for (int d = 0; d < 16; ++d) tile[16 + b + 25 * d] = u1[fft16_pos(d)];
        }
        wave_lds_fence();

        MMFE_STAMP(4);
        // ---- separate the two real spectra, power ------------------------------------------------
        float p_a[13], p_b[13];
#pragma unroll
        for (int i = 0; i < 13; ++i) {
            int k = b + 16 * i;
            if (i == 12) k = k < 200 ? k : 200;
            int km = 400 - k;
            if (i == 0) km = k == 0 ? 0 : km;
            const float2 zk = tile[k];
            const float2 zm = tile[km];
            const float2 E = make_float2(zk.x + zm.x, zk.y - zm.y);     // X_A[k]
            const float2 D = make_float2(zk.x - zm.x, zk.y + zm.y);     // i * X_B[k]
            p_a[i] = E.x * E.x + E.y * E.y;
            p_b[i] = D.x * D.x + D.y * D.y;
        }
        wave_lds_fence();
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            pw_a[b + 16 * i] = p_a[i];
            pw_b[b + 16 * i] = p_b[i];
        }
        if (b <= 8) {
            pw_a[192 + b] = p_a[12];
            pw_b[192 + b] = p_b[12];
        }
        wave_lds_fence();

#endif

        MMFE_STAMP(5);
        // ---- banded mel on the matrix pipe, log, affine, store ---------------------------------------
        float v[4 * kStreams];
#pragma unroll
        for (int st = 0; st < kStreams; ++st) {
            f32x4 acc = f32x4{1.f, 1.f, 1.f, 1.f};
            if (kSteps[st] > 0) {
                acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < kSteps[st]; ++s)
                    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(pa[st][s], wreg[kStepOff[st] + s], acc, 0, 0, 0);
                if (SPLIT && st == kLastStream) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] += upper_half_value(acc[i]);      // blocks 8..15 hold the second halves of the runs
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) v[4 * st + i] = acc[i];
        }
        MMFE_STAMP(6);
#if MMFE_R400_DIRECT && MMFE_R400_ROTATE
        window_products();
        zrow = row;
        zoq = oq;
        advance(row, oq);
        fetch(row, oq);
#define MMFE_R400_NEXT_ROW zrow
#else
#define MMFE_R400_NEXT_ROW row
#endif
        apply_log_affine_block(v, p.la);
        const bool whole = live && t0 + 8 <= p.frames_valid;        // wave-uniform
        if (p.pool) {
            // Pooled feature-major stores.  The block's four octets of a round are neighbours (32 consecutive frames while they stay in one
            // row): every lane drops its 4 frames of one filter into the tile, and after a barrier the block stores 16-byte pieces in filter
            // order, 8 consecutive lanes = the 32 frames = 128 contiguous bytes of one filter row.
            if (whole) {
#pragma unroll
                for (int st = 0; st < kStreams; ++st) {
                    if (kSteps[st] == 0) continue;
                    if (dst[st] >= 0) {
                        if (p.rowmax != nullptr) {
                            lmax = fmaxf(fmaxf(lmax, v[4 * st]), fmaxf(v[4 * st + 1], fmaxf(v[4 * st + 2], v[4 * st + 3])));
                            lmin = fminf(fminf(lmin, v[4 * st]), fminf(v[4 * st + 1], fminf(v[4 * st + 2], v[4 * st + 3])));
                        }
                        *reinterpret_cast<f32x4*>(otile + (dst[st] & 0xffff) * kPoolStride + 8 * wave + 4 * (dst[st] >> 16)) =
                            f32x4{v[4 * st], v[4 * st + 1], v[4 * st + 2], v[4 * st + 3]};
                    }
                }
            }
            if (lane == 0) { ometa[2 * wave] = whole ? (int)cur_row : -1; ometa[2 * wave + 1] = t0; }
            lds_barrier();
            for (int idx = threadIdx.x; idx < p.n_mel * 8; idx += 64 * kWaves) {
                const int m = idx >> 3, pc = idx & 7;
                const int prow = ometa[2 * (pc >> 1)], pt0 = ometa[2 * (pc >> 1) + 1];
                if (prow < 0) continue;
                const f32x4 q = *reinterpret_cast<const f32x4*>(otile + m * kPoolStride + 4 * pc);
#ifdef MMFE_ABL_NOSTORE
                if (q.x == 123.456f && q.y == 654.321f)
#endif
                store4(p, ((int64_t)prow * p.n_mel + m) * p.out_frames + pt0 + 4 * (pc & 1), q.x, q.y, q.z, q.w);
            }
            lds_barrier();          // the tile is free again
        }
        if (p.pool && (whole || !live)) {
        } else if (p.layout == MMFE_LAYOUT_FEAT_MAJOR && p.vec_store && t0 + 8 <= p.frames_valid) {
            // whole octet, 16-byte feature-major stores (the regular case): no per-frame predicates, one 64-bit scalar base per pass
            const int64_t obase = (int64_t)cur_row * p.n_mel * p.out_frames + t0;
#pragma unroll
            for (int st = 0; st < kStreams; ++st) {
                if (kSteps[st] == 0) continue;
                if (dst[st] >= 0) {
                    if (p.rowmax != nullptr) {
                        lmax = fmaxf(fmaxf(lmax, v[4 * st]), fmaxf(v[4 * st + 1], fmaxf(v[4 * st + 2], v[4 * st + 3])));
                        lmin = fminf(fminf(lmin, v[4 * st]), fminf(v[4 * st + 1], fminf(v[4 * st + 2], v[4 * st + 3])));
                    }
#ifdef MMFE_ABL_NOSTORE      // diagnostic build: what the output stores cost (the condition is never true, the values stay live)
                    if (v[4 * st] == 123.456f && v[4 * st + 1] == 654.321f)
#endif
                    store4(p, obase + coff[st], v[4 * st], v[4 * st + 1], v[4 * st + 2], v[4 * st + 3]);
                }
            }
        } else {
#pragma unroll
            for (int st = 0; st < kStreams; ++st) {
                if (kSteps[st] == 0) continue;
                // lane = 4*blk + j holds filter j of the block's unit for its 4 frames in v[4*st .. 4*st+3]
                const int d = dst[st];
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
        // the extremes go to memory when the wave leaves the row (with contiguous block ranges: once per row, not once per pass)
        if (live && p.rowmax != nullptr && (MMFE_R400_NEXT_ROW != cur_row || MMFE_R400_RANGES == 0)) flush_extremes(cur_row);
        MMFE_STAMP(7);
#ifdef MMFE_STAMPS
        st_acc[7] += 1;
#endif
        wave_lds_fence();
#if MMFE_R400_RANGES
        if (live && p.rowmax != nullptr && MMFE_R400_NEXT_ROW == cur_row && rounds_left == 1) flush_extremes(cur_row);     // the range ends inside this row
#endif
    }
#ifdef MMFE_STAMPS
    if (lane == 0)
        for (int i = 0; i < 8; ++i) p.stamps[((size_t)blockIdx.x * kWaves + wave) * 8 + i] = st_acc[i];
#endif
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

template <typename SMP, int HOP, bool VEC, int S0, int S1, int S2, int S3, bool SPLIT>
int launch_variant(const Rfft400Args& a, int64_t octets, int max_blocks_per_cu /* tuning, 0 = no limit */, int* cus_out, hipStream_t stream) {
    const void* fn = reinterpret_cast<const void*>(&rfft400_kernel<SMP, HOP, VEC, S0, S1, S2, S3, SPLIT>);
    const size_t lds = kTwBytes + (size_t)kWaves * kWaveBytes + (a.pool ? (size_t)a.n_mel * kPoolStride * 4 + 64 : 0);
    int blocks_per_cu = 1, cus = 256;
    int rc = kernel_prepare(fn, 64 * kWaves, lds, &blocks_per_cu, &cus);
    if (rc) return rc;
    if (max_blocks_per_cu >= 1 && max_blocks_per_cu < blocks_per_cu) blocks_per_cu = max_blocks_per_cu;
    *cus_out = cus;
    const int64_t want = (octets + kWaves - 1) / kWaves;
    const int grid = (int)std::min<int64_t>(want, (int64_t)cus * std::max(1, blocks_per_cu));
#ifdef MMFE_STAMPS
    Rfft400Args ta = a;
    const size_t n_st = (size_t)grid * kWaves * 8;
    MMFE_HIP_CHECK(hipMalloc((void**)&ta.stamps, n_st * 8));
    MMFE_HIP_CHECK(hipMemset(ta.stamps, 0, n_st * 8));
    hipLaunchKernelGGL((rfft400_kernel<SMP, HOP, VEC, S0, S1, S2, S3, SPLIT>), dim3(grid), dim3(64 * kWaves), lds, stream, ta);
    MMFE_HIP_CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(n_st);
    MMFE_HIP_CHECK(hipMemcpy(h.data(), ta.stamps, n_st * 8, hipMemcpyDeviceToHost));
    (void)hipFree(ta.stamps);
    static int printed = 0;
    if (printed++ < 2) {
        double sum[8] = {0};
        for (size_t w = 0; w < n_st / 8; ++w) for (int i = 0; i < 8; ++i) sum[i] += (double)h[w * 8 + i];
        const char* names[7] = {"stage+gather", "prefetch+fft25+Twrite", "Tread", "fft16x2+Zwrite", "separate+Pwrite", "mel(mfma)", "log+store+extremes"};
        double tot = 0; for (int i = 0; i < 7; ++i) tot += sum[i];
        for (int i = 0; i < 7; ++i) fprintf(stderr, "[stamps] %-22s %8.1f ticks/iter  %5.1f%%\n", names[i], sum[i] / sum[7], 100.0 * sum[i] / tot);
        fprintf(stderr, "[stamps] total %.1f ticks/iter, %.0f iterations\n", tot / sum[7], sum[7]);
    }
    return MMFE_OK;
#else
    hipLaunchKernelGGL((rfft400_kernel<SMP, HOP, VEC, S0, S1, S2, S3, SPLIT>), dim3(grid), dim3(64 * kWaves), lds, stream, a);
    MMFE_HIP_CHECK(hipGetLastError());
    return MMFE_OK;
#endif
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
    a.pool = (MMFE_R400_POOL && MMFE_R400_DIRECT && MMFE_R400_ROTATE && MMFE_R400_RANGES && kWaves == 4 && s.layout == MMFE_LAYOUT_FEAT_MAJOR && a.vec_store &&
              s.n_mel <= 128) ? 1 : 0;
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
    [[maybe_unused]] const bool vec = s.hop % 4 == 0 && a.shift % 4 == 0 && src.offsets == nullptr && src.stride % 4 == 0 &&
                     reinterpret_cast<uintptr_t>(src.base) % (pcm ? 8 : 16) == 0;
    int rc = MMFE_EINVAL;
    bool found = false;
#if MMFE_R400_DIRECT
#define MMFE_R400_SHAPE(IDX, A, B, C, D, SP)                                                                     \
    if (plan.r400_shape == IDX) {                                                                                \
        found = true;                                                                                            \
        if (s.hop == 160) {                                                                                      \
            if (pcm) rc = launch_variant<int16_t, 160, false, A, B, C, D, SP>(a, octets, plan.tune_blocks_per_cu, &cus, stream);                \
            else rc = launch_variant<float, 160, false, A, B, C, D, SP>(a, octets, plan.tune_blocks_per_cu, &cus, stream);                      \
        } else if (pcm) rc = launch_variant<int16_t, 0, false, A, B, C, D, SP>(a, octets, plan.tune_blocks_per_cu, &cus, stream);               \
        else rc = launch_variant<float, 0, false, A, B, C, D, SP>(a, octets, plan.tune_blocks_per_cu, &cus, stream);                            \
    }
#else
#define MMFE_R400_SHAPE(IDX, A, B, C, D, SP)                                                                     \
    if (plan.r400_shape == IDX) {                                                                                \
        found = true;                                                                                            \
        if (pcm) {                                                                                               \
            if (s.hop == 160 && vec) rc = launch_variant<int16_t, 160, true, A, B, C, D, SP>(a, octets, plan.tune_blocks_per_cu, &cus, stream);     \
            else if (vec) rc = launch_variant<int16_t, 0, true, A, B, C, D, SP>(a, octets, plan.tune_blocks_per_cu, &cus, stream);                  \
            else rc = launch_variant<int16_t, 0, false, A, B, C, D, SP>(a, octets, plan.tune_blocks_per_cu, &cus, stream);                          \
        } else if (s.hop == 160 && vec) rc = launch_variant<float, 160, true, A, B, C, D, SP>(a, octets, plan.tune_blocks_per_cu, &cus, stream);    \
        else if (vec) rc = launch_variant<float, 0, true, A, B, C, D, SP>(a, octets, plan.tune_blocks_per_cu, &cus, stream);                        \
        else rc = launch_variant<float, 0, false, A, B, C, D, SP>(a, octets, plan.tune_blocks_per_cu, &cus, stream);                                \
    }
#endif
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
