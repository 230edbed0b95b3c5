// Log-(mel-)spectrogram for ANY DFT length on gfx950: direct DFT with an exact-index twiddle table.
//
// Stands behind  librosa.stft(audio, n_fft=511, hop_length=120, win_length=240, window='hann',
// pad_mode='constant')  followed by  np.log(|X|^2 + eps)  as the reference calls them
// (classification/recognition/dataset/epic_kitchen.py:119-125, same code at
// classification/retrieval/dataset/epic_kitchen.py:146-152), with the /255 of
// classification/recognition/dataset/transform.py:285 folded into out_scale.  n_fft = 511 = 7*73 has
// no radix-2 structure and only win_len = 240 of its 511 inputs are non-zero, so the transform is
// evaluated as  X[k] = sum_j u[j] * W[(win_off + j) * k mod n_fft]  with the twiddle index kept as
// an exact integer (no angle accumulation).  The same kernel serves centred log-mel with reflect
// padding (n_fft = 400) and snip-edges fbank for window sizes the radix kernels do not cover.
//
// Two forms.  Vector form (dftgen_kernel, every option): a block owns FR consecutive frames of one row; the windowed
// frames are staged once in LDS as u[tap][frame] and folded about the window centre; each thread accumulates 4 bins
// x 4 frames in float64 registers (4 broadcast reads of the folded taps and 4 twiddle reads per tap pair), then the
// block applies |.|^2, the banded mel projection (from LDS), the log and the affine map and stores in the requested
// layout.  Matrix-pipe form (dftmm_kernel, plain spectrograms with at most 120 tap pairs, which is the EPIC
// configuration): the same folded DFT as two float64 MFMA products per 16 frames x 16 bins tile; see the comment
// above that kernel.
#include "mmfe_internal.h"
#include "fft_regs.h"
#include "log_affine.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <string>

namespace mmfe {

namespace {

constexpr int FR = 16;           // frames per block
constexpr int kThreads = 256;
static_assert(FR == 4 * (kThreads / 64), "the DFT loop gives each of the block's waves 4 frames");

struct DftArgs {
    RowSource src;
    float* out;
    float* rowmax;         // per-row running max as ordered int (clamp_below_max), or nullptr
    int n_rows;
    int n_samples;
    int frames_valid;      // frames computed per row
    int out_frames;        // frames per row in the output tensor
    int n_fft, win_len, win_off, hop;
    int centre_shift;      // frame t, tap j reads sample t*hop + j + centre_shift
    int pad_mode;
    int n_bins, n_mel, n_feat;
    int remove_dc;
    float preemph;
    int power;
    float spec_scale;
    LogAffine la;          // log + affine stage (plain log when a clamp pass follows)
    int layout;
    int out_dtype;         // MMFE_OUT_*: 16-bit kinds store the rounded value (what .to(dtype) of the float32 result gives)
    int p_alias;           // the power spectra reuse the LDS space of the staged taps
    const double* window;      // win_len doubles
    const double2* tw;         // n_fft entries exp(-2 pi i n / n_fft), float64
    const int32_t* band_start;
    const int32_t* band_len;
    const int32_t* band_off;
    const float* band_w;
};

// One output element in the plan's output type (float32, or float16 / bfloat16 rounded to nearest even).
__device__ __forceinline__ void store_out(const DftArgs& p, int64_t i, float v) {
    if (p.out_dtype == MMFE_OUT_F32) p.out[i] = v;
    else reinterpret_cast<unsigned short*>(p.out)[i] = p.out_dtype == MMFE_OUT_BF16 ? bf16_bits(v) : f16_bits(v);
}

template <typename S /* input element: float, or int16_t PCM decoded as s / 32768 */>
__global__ void __launch_bounds__(kThreads) dftgen_kernel(const DftArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double2* s_tw = reinterpret_cast<double2*>(smem);                     // 2*n_fft half-step twiddles
    double* s_u = reinterpret_cast<double*>(s_tw + 2 * p.n_fft);          // win_len * FR, [tap][frame], float64
    // power spectra [frame][bin]: they take over the tap array once the DFT loop is done with it when they fit
    // (fewer bytes per block = one more resident block per CU), else they get their own space behind it
    float* s_p = p.p_alias ? reinterpret_cast<float*>(s_u) : reinterpret_cast<float*>(s_u + (p.win_len + 1) * FR);                                     // FR * (n_bins + 1), [frame][bin]
    const int pstride = p.n_bins + 1;

    for (int i = threadIdx.x; i < 2 * p.n_fft; i += kThreads) s_tw[i] = p.tw[i];
    if (threadIdx.x < FR) s_u[p.win_len * FR + threadIdx.x] = 0.0;        // row of zeros behind the taps
    // Only |X[k]| is ever used, and it does not change when the DFT is taken about the centre of the window
    // instead of tap 0.  About the centre the cosine is even and the sine is odd, so with
    //   e[i] = u[hi_i] + u[lo_i],  o[i] = u[hi_i] - u[lo_i]   (taps mirrored about the centre, distance m_i)
    // the spectrum is  sum_i e[i] cos(2 pi k m_i / N)  -  i * sum_i o[i] sin(2 pi k m_i / N):
    // half the multiply-adds of the plain sum.  m_i = i + 1/2 for an even window length, i for an odd one;
    // the twiddle table holds half steps exp(-2 pi i n / 2N), indexed with k * (2 m_i) mod 2N.
    const int L = p.win_len;
    const int H = (L + 1) >> 1;              // mirrored tap pairs (the centre tap of an odd window pairs with itself)
    const int odd = L & 1;
    const int hi0 = L >> 1;                  // first tap at or above the centre
    const int lo0 = (L - 1) >> 1;            // first tap at or below the centre (== hi0 when L is odd)
    const int two_n = 2 * p.n_fft;

    const int blocks_per_row = (p.frames_valid + FR - 1) / FR;
    const int64_t n_work = (int64_t)p.n_rows * blocks_per_row;

    for (int64_t work = blockIdx.x; work < n_work; work += gridDim.x) {
        const int row = (int)(work / blocks_per_row);
        const int t0 = (int)(work - (int64_t)row * blocks_per_row) * FR;
        const int nfr = min(FR, p.frames_valid - t0);
        const S* rp = row_ptr_as<S>(p.src, row);
        const int n_row = row_samples(p.src, row, p.n_samples);

        // barriers order LDS only: the previous item's output stores stay in flight instead of being waited for
        lds_barrier();   // previous work item is done with s_u / s_p (and s_tw is loaded)
        // ---- stage windowed frames: u[j][f].  16 consecutive threads take one tap of the 16 frames, so a wave
        //      writes 4 whole rows of the [tap][frame] array (512 contiguous bytes, no LDS bank conflict); its
        //      global reads are 16 runs of 4 consecutive samples, which the overlapping frames keep in L1/L2 ----
        {
            const int f = threadIdx.x & (FR - 1);
            const int base = (t0 + f) * p.hop + p.centre_shift;
            constexpr int kRows = kThreads / FR;       // taps staged per sweep of the block
            constexpr int kSweep = 16;                 // loads per thread in flight
            // Blocks whose 16 frames lie inside the row (all but the first and last few of a row) read without
            // any padding test, 16 taps per thread in flight at once (one memory round trip for windows up to 256 taps); a sweep that waits for one load at a
            // time would leave the block idle for a memory round trip per 16 taps.
            const int lo_s = t0 * p.hop + p.centre_shift - 1;                                  // incl. the pre-emphasis neighbour
            const int hi_s = (t0 + FR - 1) * p.hop + p.centre_shift + p.win_len;
            const bool interior = nfr == FR && lo_s >= 0 && hi_s <= n_row;              // block-uniform
            if (interior) {
                const S* fp = rp + base;
                const bool pre = p.preemph != 0.f && !p.remove_dc;
                for (int j0 = threadIdx.x / FR; j0 < p.win_len; j0 += kSweep * kRows) {
                    float x[kSweep], xp[kSweep];
#pragma unroll
                    for (int u = 0; u < kSweep; ++u) {
                        const int j = j0 + u * kRows;
                        const int jc = j < p.win_len ? j : 0;
                        x[u] = sample_value(fp[jc]);
                        xp[u] = pre ? sample_value(fp[jc > 0 ? jc - 1 : 0]) : 0.f;
                    }
#pragma unroll
                    for (int u = 0; u < kSweep; ++u) {
                        const int j = j0 + u * kRows;
                        if (j < p.win_len) {
                            const float xv = pre ? x[u] - p.preemph * xp[u] : x[u];
                            double v = xv;
                            if (!p.remove_dc) v *= p.window[j];      // float64 window x float32 sample, rounded in float64
                            s_u[j * FR + f] = v;
                        }
                    }
                }
            } else {
                for (int j = threadIdx.x / FR; j < p.win_len; j += kRows) {
                    double v = 0.0;
                    if (f < nfr) {
                        float x = fetch_sample(rp, base + j, n_row, p.pad_mode);
                        if (p.preemph != 0.f && !p.remove_dc) {
                            const float prev = fetch_sample(rp, base + (j > 0 ? j - 1 : 0), n_row, p.pad_mode);
                            x = x - p.preemph * prev;
                        }
                        v = x;
                        if (!p.remove_dc) v *= p.window[j];
                    }
                    s_u[j * FR + f] = v;
                }
            }
        }
        lds_barrier();
        if (p.remove_dc) {
            // kaldi order: subtract the frame mean, pre-emphasise the centred samples, then window.
            // One thread per frame walks its taps (rarely used path: only for window sizes the radix
            // kernels do not cover).
            if (threadIdx.x < FR) {
                const int f = threadIdx.x;
                float s = 0.f;
                for (int j = 0; j < p.win_len; ++j) s += (float)s_u[j * FR + f];
                const float mean = s / (float)p.win_len;
                float prev = (float)s_u[f] - mean;
                for (int j = 0; j < p.win_len; ++j) {
                    const float g = (float)s_u[j * FR + f] - mean;
                    s_u[j * FR + f] = (double)((g - p.preemph * prev) * (float)p.window[j]);
                    prev = g;
                }
            }
            lds_barrier();
        }

        // ---- fold the windowed taps about the window centre: e at the upper tap's slot, o at the lower one's -----
        for (int i = threadIdx.x; i < (H - odd) * FR; i += kThreads) {
            const int q = i / FR + odd;          // pair index (the self-paired centre tap of an odd window is left as it is)
            const int f = i - (i / FR) * FR;
            const double a = s_u[(lo0 - q) * FR + f], b = s_u[(hi0 + q) * FR + f];
            s_u[(hi0 + q) * FR + f] = b + a;
            s_u[(lo0 - q) * FR + f] = b - a;
        }
        lds_barrier();

        // ---- direct DFT about the window centre.  Register blocking: wave w owns frames 4w..4w+3, lane l owns
        //      bins l, l+64, l+128, l+192 (of each group of 256 bins): 32 float64 accumulators per thread, and per
        //      tap pair 4 broadcast reads of e/o plus 4 twiddle reads feed 32 multiply-adds, so the LDS pipe
        //      keeps pace with the float64 FMA pipe.  Products and sums are float64, the arithmetic type of the
        //      librosa path (fp64 window x fp32 frame, fp64 FFT, result rounded to complex64); gfx950 issues
        //      v_fma_f64 at the fp32 FMA rate. -------------------------------------------------------------
        const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
        for (int kb = 0; kb < p.n_bins; kb += 256) {
            double re[4][4], im[4][4];
            // twiddle cursors are byte offsets into s_tw, which sits at LDS offset 0; min(a, a - span) in unsigned
            // arithmetic is the wrap-around (a - span underflows to a huge value while a < span)
            unsigned tw_at[4], tw_step[4];
            const unsigned tw_span = (unsigned)two_n * 16u;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = kb + ln + 64 * j;
                const int kk = k < p.n_bins ? k : 0;        // lanes past the last bin compute bin 0 and drop it
                tw_step[j] = (unsigned)((2 * kk) % two_n) * 16u;
                tw_at[j] = odd ? 0u : (unsigned)(kk % two_n) * 16u;      // k * (2 m_0) mod 2N
#pragma unroll
                for (int f = 0; f < 4; ++f) { re[j][f] = 0.0; im[j][f] = 0.0; }
            }
            for (int i = 0; i < H; ++i) {
                // the centre tap of an odd window pairs with itself: its odd part is the row of zeros behind the taps
                const double2* ep = reinterpret_cast<const double2*>(s_u + (hi0 + i) * FR + 4 * wv);
                const double2* op = reinterpret_cast<const double2*>(s_u + ((odd && i == 0) ? L : lo0 - i) * FR + 4 * wv);
                const double2 e01 = ep[0], e23 = ep[1], o01 = op[0], o23 = op[1];
                const double e[4] = {e01.x, e01.y, e23.x, e23.y};
                const double o[4] = {o01.x, o01.y, o23.x, o23.y};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const double2 w = *reinterpret_cast<const double2*>(smem + tw_at[j]);
                    tw_at[j] += tw_step[j];
                    tw_at[j] = min(tw_at[j], tw_at[j] - tw_span);
#pragma unroll
                    for (int f = 0; f < 4; ++f) {
                        re[j][f] = fma(e[f], w.x, re[j][f]);
                        im[j][f] = fma(o[f], w.y, im[j][f]);
                    }
                }
            }
            if (p.p_alias) lds_barrier();        // every wave is done reading the taps (one bin group only in this mode)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = kb + ln + 64 * j;
                if (k >= p.n_bins) continue;
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    // complex64 rounding of the (centre-referenced) spectrum, then fp32 |.|^2 like the reference's numpy expression
                    const float xr = (float)(re[j][f] * (double)p.spec_scale), xi = (float)(im[j][f] * (double)p.spec_scale);
                    float v = xr * xr + xi * xi;
                    if (p.power == 1) v = sqrtf(v);
                    s_p[(4 * wv + f) * pstride + k] = v;
                }
            }
        }
        lds_barrier();

        // ---- mel (optional), log, affine, store ------------------------------------------------
        const int nfeat = p.n_feat;
        float local_max = -INFINITY;
        for (int i = threadIdx.x; i < FR * nfeat; i += kThreads) {
            int f, m;
            if (p.layout == MMFE_LAYOUT_FEAT_MAJOR) { m = i / FR; f = i - m * FR; }    // frames fastest: matches out[row][m][t]
            else { f = i / nfeat; m = i - f * nfeat; }                                  // features fastest: matches out[row][t][m]
            if (f >= nfr) continue;
            float v;
            if (p.n_mel > 0) {
                const int st = p.band_start[m], len = p.band_len[m];
                const float* w = p.band_w + p.band_off[m];
                const float* sp = s_p + f * pstride + st;
                v = 0.f;
                for (int q = 0; q < len; ++q) v = fmaf(w[q], sp[q], v);
            } else {
                v = s_p[f * pstride + m];
            }
            float vv[1] = {v};
            apply_log_affine_block(vv, p.la);      // hardware log2 + one fma; floored cells take the host-rounded log(floor)
            v = vv[0];
            int64_t o;
            if (p.layout == MMFE_LAYOUT_FEAT_MAJOR) o = ((int64_t)row * nfeat + m) * p.out_frames + (t0 + f);
            else o = ((int64_t)row * p.out_frames + (t0 + f)) * nfeat + m;
            store_out(p, o, v);
            if (p.rowmax != nullptr) local_max = fmaxf(local_max, v);      // clamp + affine happen in the second pass
        }
        if (p.rowmax != nullptr) {
            // wave max, then one atomic per wave
            for (int off = 32; off > 0; off >>= 1) local_max = fmaxf(local_max, __shfl_xor(local_max, off));
            if ((threadIdx.x & 63) == 0 && local_max > -INFINITY)
                atomicMax(reinterpret_cast<int*>(p.rowmax) + row, float_to_ordered(local_max));
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Matrix-pipe form of the same transform for plain spectrograms (no mel bank, no DC removal / pre-emphasis, no
// per-row clamp) with at most 120 mirrored tap pairs: EPIC's n_fft 511 / 240-tap configuration.
//
// About the window centre the spectrum of a block of frames is two real matrix products,
//     Re[k][f] = sum_i cos(2 pi k m_i / N) * e[i][f],     Im[k][f] = sum_i -sin(2 pi k m_i / N) * o[i][f],
// (bins x pairs) x (pairs x frames), and v_mfma_f64_16x16x4_f64 runs them at the float64 FMA rate of the vector
// pipe without that pipe's per-FMA operand traffic: a wave owns 16 frames, keeps their folded taps e / o in
// registers as the B operands (lane l: pair 4s + (l >> 4) of frame l & 15 for k-step s), walks the 16-bin tiles
// of the spectrum and fetches the A operands (one cos / -sin pair per lane and step, bin 16*tile + (l & 15),
// pair 4s + (l >> 4)) from the half-step twiddle table in LDS with an incrementally updated index.  The
// accumulators hold bins 16*tile + (l >> 4) + 4j, j = 0..3, of frame l & 15: rounded to complex64, |.|^2, log,
// affine and stored from registers.  Waves are independent (no LDS staging of samples, no barriers after the
// table load): each lane reads its 2*KS samples straight from global memory (the 16 overlapping frames of a
// tile live in L1/L2) and multiplies by the float64 window.
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <typename S, int KS /* k-steps of 4 pairs */, bool OUT16 /* float16 / bfloat16 stores (a compile-time choice: as a run-time branch in
                                                                  the epilogue it cost this kernel 4 %) */>
__global__ void __launch_bounds__(kThreads, 2) dftmm_kernel(const DftArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double2* s_tw = reinterpret_cast<double2*>(smem);                     // 2*n_fft half-step twiddles (cos, -sin)
    double* s_win = reinterpret_cast<double*>(s_tw + 2 * p.n_fft);        // win_len
    for (int i = threadIdx.x; i < 2 * p.n_fft; i += kThreads) s_tw[i] = p.tw[i];
    for (int i = threadIdx.x; i < p.win_len; i += kThreads) s_win[i] = p.window[i];
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef MMFE_DFTMM_CLOCK
    unsigned long long clk0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(clk0)::"memory");
#endif
    const int f = lane & 15, c = lane >> 4;
    const int L = p.win_len;
    const int H = (L + 1) >> 1;
    const int odd = L & 1;
    const int hi0 = L >> 1, lo0 = (L - 1) >> 1;
    const int two_n = 2 * p.n_fft;
    const unsigned tw_span = (unsigned)two_n * 16u;
    const int n_tiles = (p.n_bins + 15) >> 4;
    const int tiles_per_row = (p.frames_valid + 15) >> 4;
    const int64_t n_items = (int64_t)p.n_rows * tiles_per_row;

    for (int64_t item = (int64_t)blockIdx.x * (kThreads / 64) + wv; item < n_items; item += (int64_t)gridDim.x * (kThreads / 64)) {
        const int row = (int)(item / tiles_per_row);
        const int t0 = (int)(item - (int64_t)row * tiles_per_row) * 16;
        const S* rp = row_ptr_as<S>(p.src, row);
        const int n_row = row_samples(p.src, row, p.n_samples);
        const int base = (t0 + f) * p.hop + p.centre_shift;
        const int first = t0 * p.hop + p.centre_shift;
        const bool interior = first >= 0 && first + 15 * p.hop + L <= n_row;      // wave-uniform: no padding test needed

        // folded, windowed taps of this lane's pairs (float64 window x float32 sample, like the vector form)
        // (the pair index is made opaque per item: otherwise the tap offsets, masks and window values of all KS steps
        // are hoisted out of the item loop and held in registers next to e / o)
        int cq = c;
        asm volatile("" : "+v"(cq));
        // all 2*KS samples of the lane are requested before the first one is used: one memory round trip per item
        float xh[KS], xl[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int i = 4 * s + cq;
            const bool live = i < H;
            const int jh = live ? hi0 + i : hi0, jl = live ? lo0 - i : hi0;
            if (interior) {
                xh[s] = sample_value(rp[base + jh]);
                xl[s] = sample_value(rp[base + jl]);
            } else {
                xh[s] = fetch_sample(rp, base + jh, n_row, p.pad_mode);
                xl[s] = fetch_sample(rp, base + jl, n_row, p.pad_mode);
            }
        }
        double e[KS], o[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int i = 4 * s + cq;
            const bool live = i < H;
            const int jh = live ? hi0 + i : hi0, jl = live ? lo0 - i : hi0;
            const double uh = (double)xh[s] * s_win[jh], ul = (double)xl[s] * s_win[jl];
            const bool self = odd && i == 0;          // the centre tap of an odd window pairs with itself
            e[s] = live ? (self ? uh : uh + ul) : 0.0;
            o[s] = (live && !self) ? uh - ul : 0.0;
        }

        const bool frame_ok = t0 + f < p.frames_valid;
        // The A operands run R - 1 k-steps ahead of the MFMAs that use them, through a ring of R register pairs, and
        // keep running across tile boundaries (the reads for the first steps of tile mt + 1 are issued under the
        // last steps of tile mt), so neither the LDS latency nor its bank conflicts reach the matrix pipe.
        constexpr int R = (KS % 6 == 0) ? 6 : 4;
        static_assert(KS % R == 0, "ring slots must line up across tiles");
        // byte offset into the twiddle table (LDS offset 0) of k * (2 m_i) mod 2N for bin k = 16*tile + (lane & 15) and
        // pair i = 4s + c; one step of s advances 2 m_i by 8.  min(a, a - span) in unsigned arithmetic is the wrap-around.
        auto cursor = [&](int mt, unsigned& at, unsigned& step) {
            const int k = 16 * mt + f;
            const int kk = k < p.n_bins ? k : 0;
            at = (unsigned)((kk * (2 * c + 1 - odd)) % two_n) * 16u;
            step = (unsigned)((8 * kk) % two_n) * 16u;
        };
        unsigned at, step;
        cursor(0, at, step);
        double2 ring[R];
#pragma unroll
        for (int s = 0; s < R - 1; ++s) {
            ring[s] = *reinterpret_cast<const double2*>(smem + at);
            at += step;
            at = min(at, at - tw_span);
        }
        for (int mt = 0; mt < n_tiles; ++mt) {
            unsigned at_n, step_n;
            cursor(mt + 1 < n_tiles ? mt + 1 : mt, at_n, step_n);
            f64x4 re = {0.0, 0.0, 0.0, 0.0}, im = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (s + R - 1 == KS) { at = at_n; step = step_n; }          // from here on the reads belong to the next tile
                ring[(s + R - 1) % R] = *reinterpret_cast<const double2*>(smem + at);
                at += step;
                at = min(at, at - tw_span);
                const double2 w = ring[s % R];
                re = __builtin_amdgcn_mfma_f64_16x16x4f64(w.x, e[s], re, 0, 0, 0);
                im = __builtin_amdgcn_mfma_f64_16x16x4f64(w.y, o[s], im, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);       // keeps the read-ahead distance as written
            }
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // complex64 rounding of the (centre-referenced) spectrum, then fp32 |.|^2 like the reference's numpy expression
                const float xr = (float)(re[j] * (double)p.spec_scale), xi = (float)(im[j] * (double)p.spec_scale);
                v[j] = xr * xr + xi * xi;
                if (p.power == 1) v[j] = sqrtf(v[j]);
            }
            apply_log_affine_block(v, p.la);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int bin = 16 * mt + c + 4 * j;
                if (!frame_ok || bin >= p.n_bins) continue;
                int64_t oi;
                if (p.layout == MMFE_LAYOUT_FEAT_MAJOR) oi = ((int64_t)row * p.n_feat + bin) * p.out_frames + (t0 + f);
                else oi = ((int64_t)row * p.out_frames + (t0 + f)) * p.n_feat + bin;
                if (OUT16) reinterpret_cast<unsigned short*>(p.out)[oi] = p.out_dtype == MMFE_OUT_BF16 ? bf16_bits(v[j]) : f16_bits(v[j]);
                else p.out[oi] = v[j];
            }
        }
    }
#ifdef MMFE_DFTMM_CLOCK
    unsigned long long clk1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(clk1)::"memory");
    if (blockIdx.x == 0 && threadIdx.x == 0) printf("dftmm block 0 wave 0: %llu shader-clock ticks\n", clk1 - clk0);
#endif
}

template <typename S, int KS, bool OUT16>
int dftmm_launch_typed(const DftArgs& a, int64_t items, size_t lds, hipStream_t stream) {
    int resident = 1, cus = 256;
    int rc = kernel_prepare(reinterpret_cast<const void*>(&dftmm_kernel<S, KS, OUT16>), kThreads, lds, &resident, &cus);
    if (rc) return rc;
    const int64_t blocks = (items + kThreads / 64 - 1) / (kThreads / 64);
    const int grid = (int)std::min<int64_t>(blocks, (int64_t)cus * resident);
    hipLaunchKernelGGL((dftmm_kernel<S, KS, OUT16>), dim3(grid), dim3(kThreads), lds, stream, a);
    MMFE_HIP_CHECK(hipGetLastError());
    return MMFE_OK;
}

template <typename S, int KS>
int dftmm_launch_variant(const DftArgs& a, int64_t items, size_t lds, hipStream_t stream) {
    return a.out_dtype == MMFE_OUT_F32 ? dftmm_launch_typed<S, KS, false>(a, items, lds, stream) : dftmm_launch_typed<S, KS, true>(a, items, lds, stream);
}

// Plans the matrix-pipe form serves (MMFE_DFT_FORM=valu keeps them on the vector form).
bool dftmm_supports(const mmfe_spec& s) {
    if (const char* f = std::getenv("MMFE_DFT_FORM")) {
        if (std::string(f) == "valu") return false;
    }
    const int pairs = (s.win_len + 1) / 2;
    const size_t lds = (size_t)s.n_fft * 32 + (size_t)s.win_len * 8;
    return s.n_mel == 0 && !s.remove_dc && s.preemph == 0.f && !(s.clamp_below_max > 0.f) && pairs <= 120 && lds <= 64 * 1024;
}

// extremes[0..cap) = running maximum, extremes[cap..2cap) = running minimum, both in ordered-int form
__global__ void rowmax_init_kernel(int* extremes, int n, int cap) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        extremes[i] = float_to_ordered(-INFINITY);
        extremes[cap + i] = float_to_ordered(INFINITY);
    }
}

// second pass of clamp_below_max: x = max(x, rowmax - c), then affine
__global__ void clamp_affine_kernel(float* out, const int* rowmax, int64_t per_row, int64_t total, float c, float shift,
                                    float scale) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / per_row);
        const int oi = rowmax[row];
        const float mx = __int_as_float(oi >= 0 ? oi : oi ^ 0x7fffffff);
        out[i] = (fmaxf(out[i], mx - c) + shift) * scale;
    }
}

}  // namespace

int rowmax_init_launch(const ClampScratch& cs, int64_t n_rows, hipStream_t stream) {
    if (cs.base == nullptr || n_rows > cs.capacity) {
        set_error("clamp_below_max: %lld rows in one call, this entry point handles at most %lld", (long long)n_rows, (long long)cs.capacity);
        return MMFE_EINVAL;
    }
    hipLaunchKernelGGL(rowmax_init_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, stream,
                       reinterpret_cast<int*>(cs.base), (int)n_rows, (int)cs.capacity);
    MMFE_HIP_CHECK(hipGetLastError());
    return MMFE_OK;
}

int clamp_affine_launch(const Plan& plan, const ClampScratch& cs, float* out, int64_t n_rows, int64_t per_row, hipStream_t stream) {
    const mmfe_spec& s = plan.spec;
    int dev = 0, cus = 256;
    MMFE_HIP_CHECK(hipGetDevice(&dev));
    MMFE_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int64_t total = per_row * n_rows;
    const int grid = (int)std::min<int64_t>((total + 255) / 256, (int64_t)cus * 16);
    hipLaunchKernelGGL(clamp_affine_kernel, dim3(grid), dim3(256), 0, stream, out, reinterpret_cast<const int*>(cs.base),
                       per_row, total, s.clamp_below_max, s.out_shift, s.out_scale);
    MMFE_HIP_CHECK(hipGetLastError());
    return MMFE_OK;
}

int dftgen_prepare(Plan& plan) {
    const mmfe_spec& s = plan.spec;
    if (s.n_fft < 2 || s.n_fft > 4096 || s.win_len < 1 || s.win_off < 0 || s.win_off + s.win_len > s.n_fft) {
        set_error("dftgen: needs 2 <= n_fft <= 4096 and the window inside the DFT input");
        return MMFE_EINVAL;
    }
    // LDS need of the vector form (twiddles, tap array of 16 frames, power spectra when they cannot reuse the tap array):
    // refuse here what no launch could serve.  In practice win_len close to n_fft stops fitting above n_fft of about 850.
    if (!dftmm_supports(s)) {
        const size_t p_bytes = (size_t)FR * (plan.n_bins + 1) * 4;
        const bool alias = plan.n_bins <= 256 && p_bytes <= (size_t)s.win_len * FR * 8;
        const size_t lds = (size_t)s.n_fft * 32 + (size_t)(s.win_len + 1) * FR * 8 + (alias ? 0 : p_bytes);
        if (lds > 160 * 1024) {
            set_error("dftgen: n_fft %d with %d window taps needs %zu bytes of LDS, a CU has %d", s.n_fft, s.win_len, lds, 160 * 1024);
            return MMFE_EINVAL;
        }
    }
    // half-step twiddles exp(-2 pi i n / 2N), n < 2N
    std::vector<double> tw(4 * (size_t)s.n_fft);
    const double pi = 3.14159265358979323846;
    for (int n = 0; n < 2 * s.n_fft; ++n) {
        tw[2 * n] = std::cos(pi * n / s.n_fft);
        tw[2 * n + 1] = -std::sin(pi * n / s.n_fft);
    }
    int rc;
    if ((rc = upload(&plan.d_tw64, tw))) return rc;
    if ((rc = upload(&plan.d_window64, plan.window64))) return rc;
    if (s.n_mel > 0) {
        std::vector<int32_t> off(s.n_mel, 0);
        int acc = 0;
        for (int m = 0; m < s.n_mel; ++m) {
            off[m] = acc;
            acc += plan.band_len[m];
            if (plan.band_len[m] > 0 && (plan.band_start[m] < 0 || plan.band_start[m] + plan.band_len[m] > plan.n_bins)) {
                set_error("dftgen: mel band %d leaves the spectrum", m);
                return MMFE_EINVAL;
            }
        }
        std::vector<int32_t> st = plan.band_start;
        for (int m = 0; m < s.n_mel; ++m) if (plan.band_len[m] == 0) st[m] = 0;
        if ((rc = upload(&plan.d_band_start, st))) return rc;
        if ((rc = upload(&plan.d_band_len, plan.band_len))) return rc;
        if ((rc = upload(&plan.d_band_off, off))) return rc;
        std::vector<float> w = plan.band_weights;
        if (w.empty()) w.push_back(0.f);
        if ((rc = upload(&plan.d_band_w, w))) return rc;
    }
    return MMFE_OK;
}

int dftgen_launch(const Plan& plan, const RowSource& src, int64_t n_rows, int64_t n_samples, int64_t n_frames_natural,
                  int64_t out_frames, float* out, const ClampScratch& cs, hipStream_t stream) {
    const mmfe_spec& s = plan.spec;
    const int64_t frames_valid = std::min(n_frames_natural, out_frames);
    if (n_rows <= 0 || frames_valid <= 0) return MMFE_OK;
    if (n_rows >= (int64_t)1 << 31 || n_samples >= (int64_t)1 << 30) {
        set_error("dftgen: batch too large for one launch");
        return MMFE_EINVAL;
    }
    DftArgs a{};
    a.src = src;
    a.out = out;
    a.rowmax = nullptr;
    a.n_rows = (int)n_rows;
    a.n_samples = (int)n_samples;
    a.frames_valid = (int)frames_valid;
    a.out_frames = (int)out_frames;
    a.n_fft = s.n_fft; a.win_len = s.win_len; a.win_off = s.win_off; a.hop = s.hop;
    // frame t, tap j reads sample t*hop + j + centre_shift: librosa centres the n_fft-long frame on t*hop, scipy's zero
    // boundary extension puts win_len/2 zeros in front of the row, snip-edges framing starts at the row's first sample
    a.centre_shift = s.frames_mode == MMFE_FRAMES_CENTER ? s.win_off - s.n_fft / 2 : (s.frames_mode == MMFE_FRAMES_SCIPY ? -(s.win_len / 2) : 0);
    a.pad_mode = s.pad_mode;
    a.n_bins = plan.n_bins; a.n_mel = s.n_mel; a.n_feat = s.n_mel > 0 ? s.n_mel : plan.n_bins;
    a.remove_dc = s.remove_dc; a.preemph = s.preemph; a.power = s.power; a.spec_scale = s.spec_scale;
    a.la = make_log_affine(s, s.clamp_below_max > 0.f);
    a.layout = s.layout;
    a.out_dtype = s.out_dtype;
    a.window = plan.d_window64;
    a.tw = reinterpret_cast<const double2*>(plan.d_tw64);
    a.band_start = plan.d_band_start; a.band_len = plan.d_band_len; a.band_off = plan.d_band_off; a.band_w = plan.d_band_w;

    const bool clamp = s.clamp_below_max > 0.f;
    if (clamp) {
        int rc = rowmax_init_launch(cs, n_rows, stream);
        if (rc) return rc;
        a.rowmax = cs.base;
    }
    if (dftmm_supports(s)) {
        const size_t mm_lds = (size_t)s.n_fft * 32 + (size_t)s.win_len * 8;
        const int64_t items = n_rows * ((frames_valid + 15) / 16);
        const int ksteps = ((s.win_len + 1) / 2 + 3) / 4;
        const bool pcm16 = s.in_dtype == MMFE_IN_S16;
#define MMFE_DFTMM_DISPATCH(KS_)                                                                     \
        if (ksteps <= KS_) return pcm16 ? dftmm_launch_variant<int16_t, KS_>(a, items, mm_lds, stream)  \
                                        : dftmm_launch_variant<float, KS_>(a, items, mm_lds, stream);
        MMFE_DFTMM_DISPATCH(8)
        MMFE_DFTMM_DISPATCH(16)
        MMFE_DFTMM_DISPATCH(30)
#undef MMFE_DFTMM_DISPATCH
    }
    const size_t p_bytes = (size_t)FR * (plan.n_bins + 1) * 4;
    a.p_alias = (plan.n_bins <= 256 && p_bytes <= (size_t)s.win_len * FR * 8) ? 1 : 0;
    const size_t lds = (size_t)s.n_fft * 32 + (size_t)(s.win_len + 1) * FR * 8 + (a.p_alias ? 0 : p_bytes);
    if (lds > 160 * 1024) {
        set_error("dftgen: n_fft %d / win_len %d need %zu bytes of LDS", s.n_fft, s.win_len, lds);
        return MMFE_EINVAL;
    }
    int resident = 1, cus = 256;      // persistent blocks: as many as fit at once, so that every block gets the same share of the work
    const bool pcm = s.in_dtype == MMFE_IN_S16;
    const void* fn = pcm ? reinterpret_cast<const void*>(&dftgen_kernel<int16_t>) : reinterpret_cast<const void*>(&dftgen_kernel<float>);
    int rc = kernel_prepare(fn, kThreads, lds, &resident, &cus);
    if (rc) return rc;
    const int64_t work = n_rows * ((frames_valid + FR - 1) / FR);
    const int grid = (int)std::min<int64_t>(work, (int64_t)cus * resident);
    if (pcm) hipLaunchKernelGGL(dftgen_kernel<int16_t>, dim3(grid), dim3(kThreads), lds, stream, a);
    else hipLaunchKernelGGL(dftgen_kernel<float>, dim3(grid), dim3(kThreads), lds, stream, a);
    MMFE_HIP_CHECK(hipGetLastError());
    if (clamp) return clamp_affine_launch(plan, cs, out, n_rows, out_frames * a.n_feat, stream);
    return MMFE_OK;
}

}  // namespace mmfe
