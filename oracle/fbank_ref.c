/*
 * CPU restatement in C of the VGGSound front-end: TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the library built from this file,
 * and only as a checker / as the timed CPU baseline.  Nothing under multi_modal_amd/ links or loads it.
 *
 * What it restates: torchaudio.compliance.kaldi.fbank as the reference calls it
 * (classification/vggsound/dataset/vggsound.py:45-47: htk_compat=True, sample_frequency=16000, use_energy=False,
 * window_type='hanning', num_mel_bins=128, dither=0.0, frame_shift=25) followed by the pad / cut to 384 frames and the
 * (x + 5.1991) / 4.4049 of vggsound.py:48-59.  torchaudio is absent from this image and unpinned in the reference:
 * PARITY UNPINNED, exactly as for oracle/frontend_oracle.py, whose operation order this file follows (snip-edges framing,
 * per-frame mean removal, pre-emphasis with a replicated first sample, symmetric window, zero padding to the next power of
 * two, |rfft|^2, banded mel projection, ln(max(., eps))).  It shares no code with that module: the FFT below is a
 * textbook iterative radix-2 transform with float64 twiddles, the window and the mel weights are passed in by the caller
 * (the Python oracle's float32 tables), so the two restatements check each other's framing, pre-processing, transform and
 * projection, and differ only by float32 rounding (p99.9 <= 1e-4 on the log-mel values, tests/test_oracle_cpu.py).
 *
 * Clips are independent: a batch runs as one OpenMP task per clip.  (The benchmark's CPU baseline stays the Python / torch
 * restatement, whose pocketfft transform is several times faster per core than the plain radix-2 code here.)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    int n_fft;          /* padded window (power of two) */
    int win_len;        /* samples per frame */
    int hop;            /* samples between frames */
    int n_mel;
    int out_frames;     /* > 0: rows are cut / padded to this many frames */
    float preemph;
    int remove_dc;
    float log_floor;    /* ln(max(mel, log_floor)) */
    float pad_value;    /* value of padded frames BEFORE the affine map */
    float out_shift, out_scale;   /* stored value = (x + out_shift) * out_scale */
} fbank_ref_params;

/* in-place iterative radix-2 complex FFT of length n (power of two).  Twiddles are laid out stage by stage: the stage with
 * butterflies of span `half` reads tw[half - 1 + k] = exp(-2 pi i k / (2 half)), k < half (n - 1 entries in all), so the inner
 * loop walks contiguous memory and vectorises. */
static void fft_radix2(float* restrict re, float* restrict im, int n, const float* restrict tw_re, const float* restrict tw_im) {
    for (int i = 1, j = 0; i < n; ++i) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            float t = re[i]; re[i] = re[j]; re[j] = t;
            t = im[i]; im[i] = im[j]; im[j] = t;
        }
    }
    for (int half = 1; half < n; half <<= 1) {
        const float* wr = tw_re + half - 1;
        const float* wi = tw_im + half - 1;
        for (int i = 0; i < n; i += 2 * half) {
            float* restrict ar = re + i;
            float* restrict ai = im + i;
            float* restrict br = re + i + half;
            float* restrict bi = im + i + half;
            for (int k = 0; k < half; ++k) {
                const float tr = br[k] * wr[k] - bi[k] * wi[k], ti = br[k] * wi[k] + bi[k] * wr[k];
                br[k] = ar[k] - tr;
                bi[k] = ai[k] - ti;
                ar[k] += tr;
                ai[k] += ti;
            }
        }
    }
}

/* Frames a clip of n samples yields (kaldi snip_edges). */
int64_t fbank_ref_num_frames(const fbank_ref_params* p, int64_t n) { return n < p->win_len ? 0 : 1 + (n - p->win_len) / p->hop; }

/* One clip: wave[n] -> out[(out_frames or natural frames)][n_mel].  window[win_len]; mel bank band by band:
 * filter m covers bins [band_start[m], band_start[m] + band_len[m]) with weights band_w[band_off[m] ...].
 * scratch: 2 * n_fft floats.  Returns the number of rows written. */
static int64_t fbank_one(const fbank_ref_params* p, const float* wave, int64_t n, const float* window, const int32_t* band_start,
                         const int32_t* band_len, const int32_t* band_off, const float* band_w, const float* tw_re,
                         const float* tw_im, float* scratch, float* out) {
    const int N = p->n_fft, W = p->win_len;
    float* re = scratch;
    float* im = scratch + N;
    const int64_t natural = fbank_ref_num_frames(p, n);
    const int64_t rows = p->out_frames > 0 ? p->out_frames : natural;
    const int64_t computed = natural < rows ? natural : rows;
    for (int64_t t = 0; t < computed; ++t) {
        const float* x = wave + t * p->hop;
        float mean = 0.f;
        if (p->remove_dc) {
            float s = 0.f;
            for (int i = 0; i < W; ++i) s += x[i];
            mean = s / (float)W;
        }
        /* pre-emphasis on the mean-free frame, first sample replicated; window; zero padding */
        float prev = x[0] - mean;
        for (int i = 0; i < W; ++i) {
            const float cur = x[i] - mean;
            re[i] = (cur - p->preemph * prev) * window[i];
            im[i] = 0.f;
            prev = cur;
        }
        for (int i = W; i < N; ++i) { re[i] = 0.f; im[i] = 0.f; }
        fft_radix2(re, im, N, tw_re, tw_im);
        float* o = out + t * p->n_mel;
        for (int m = 0; m < p->n_mel; ++m) {
            const float* w = band_w + band_off[m];
            const int k0 = band_start[m];
            float acc = 0.f;
            for (int j = 0; j < band_len[m]; ++j) {
                const int k = k0 + j;
                acc += w[j] * (re[k] * re[k] + im[k] * im[k]);
            }
            const float v = logf(acc > p->log_floor ? acc : p->log_floor);
            o[m] = (v + p->out_shift) * p->out_scale;
        }
    }
    const float padv = (p->pad_value + p->out_shift) * p->out_scale;
    for (int64_t t = computed; t < rows; ++t)
        for (int m = 0; m < p->n_mel; ++m) out[t * p->n_mel + m] = padv;
    return rows;
}

/* Batch of clips, one OpenMP task per clip.  waves: (n_clips, n) row-major with row stride `stride`; out: (n_clips, rows, n_mel).
 * threads <= 0: the OpenMP default.  Returns 0, or -1 on a bad argument / allocation failure. */
int fbank_ref_batch(const fbank_ref_params* p, const float* waves, int64_t n_clips, int64_t n, int64_t stride, const float* window,
                    const int32_t* band_start, const int32_t* band_len, const float* band_w, float* out, int threads) {
    if (p == NULL || waves == NULL || out == NULL || p->n_fft < 2 || (p->n_fft & (p->n_fft - 1)) || p->win_len > p->n_fft ||
        p->win_len < 1 || p->hop < 1 || p->n_mel < 1)
        return -1;
    const int N = p->n_fft;
    const double pi = 3.14159265358979323846;
    float* tw_re = (float*)malloc(sizeof(float) * 2 * (size_t)N);
    int32_t* band_off = (int32_t*)malloc(sizeof(int32_t) * (size_t)p->n_mel);
    if (tw_re == NULL || band_off == NULL) { free(tw_re); free(band_off); return -1; }
    float* tw_im = tw_re + N;
    for (int half = 1; half < N; half <<= 1)
        for (int k = 0; k < half; ++k) {
            tw_re[half - 1 + k] = (float)cos(pi * k / half);
            tw_im[half - 1 + k] = (float)-sin(pi * k / half);
        }
    int32_t off = 0;
    for (int m = 0; m < p->n_mel; ++m) { band_off[m] = off; off += band_len[m]; }
    const int64_t natural = fbank_ref_num_frames(p, n);
    const int64_t rows = p->out_frames > 0 ? p->out_frames : natural;
    int failed = 0;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel
    {
        float* scratch = (float*)malloc(sizeof(float) * 2 * (size_t)N);
        if (scratch == NULL) {
#pragma omp atomic write
            failed = 1;
        }
#pragma omp for schedule(dynamic, 1)
        for (int64_t c = 0; c < n_clips; ++c) {
            if (scratch != NULL)
                fbank_one(p, waves + c * stride, n, window, band_start, band_len, band_off, band_w, tw_re, tw_im, scratch,
                          out + c * rows * p->n_mel);
        }
        free(scratch);
    }
    free(tw_re);
    free(band_off);
    return failed ? -1 : 0;
}

int fbank_ref_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
