/*
 * mmfe — MI355X (gfx950) signal front-end, C ABI.
 *
 * The reference (lixinghe1999/multi_modal) is pure Python and has no FFI of its own: the
 * boundary it offers for this path is a set of Python call sites that hand a waveform to a
 * third-party CPU routine.  Each entry point below names the call site(s) it stands behind;
 * the Python wrappers in multi_modal_amd/ keep those call signatures and bind this library
 * with ctypes.
 *
 *   kind MMFE_KIND_FBANK     torchaudio.compliance.kaldi.fbank(...) called from
 *                            classification/vggsound/dataset/vggsound.py:45-47, with the
 *                            pad/cut to 384 frames and the affine normalisation of
 *                            vggsound.py:48-59 fused in (out_frames / pad_value / out_shift /
 *                            out_scale).
 *   kind MMFE_KIND_LOGSPEC   librosa.stft(...) + np.log(|X|^2 + eps) of
 *                            classification/recognition/dataset/epic_kitchen.py:114-126 and
 *                            classification/retrieval/dataset/epic_kitchen.py:141-153, with the
 *                            /255 of classification/recognition/dataset/transform.py:268-285
 *                            fused in (out_scale).  Also serves the long-utterance log-mel
 *                            configuration (centre reflect padding, mel, log10, per-clip clamp).
 *   kind MMFE_KIND_STFT_MAG  np.abs(scipy.signal.stft(x, fs, nperseg, noverlap)[-1]) of
 *                            classification/har/adbox.py:20.
 *   kind MMFE_KIND_FRAME     frame_signal(signal, frame_length, hop_length, window) of
 *                            classification/vggsound/utils/transforms.py:19-61 (identical copies
 *                            under recognition/ and retrieval/).
 *   kind MMFE_KIND_RESAMPLE  the sample-rate conversion in front of the path: librosa.load(file,
 *                            sr=16000, duration=10) of classification/vggsound/dataset/vggsound.py:114
 *                            and ffmpeg -ar 24000 of
 *                            classification/recognition/dataset/extract_audio.py:6-13 (filter design,
 *                            alignment and output length of scipy.signal.resample_poly).
 *
 * Conventions
 *   - Every function returns 0 on success and a negative MMFE_E* code on failure; the message is
 *     available from mmfe_last_error() (thread-local).  Nothing throws across the ABI, nothing
 *     calls exit().
 *   - All data pointers passed to mmfe_run* are DEVICE pointers owned by the caller (e.g.
 *     tensor.data_ptr()).  The library allocates only plan-lifetime constant tables.
 *   - Input samples are float32, or (spec.in_dtype = MMFE_IN_S16) the signed 16-bit PCM the reference's audio
 *     files hold, decoded in the kernel's loads; strides and offsets always count input elements.
 *   - mmfe_run* is asynchronous on the given HIP stream (NULL = the default stream); it never
 *     synchronises and never allocates.
 *   - Plans are immutable after creation and may be shared between threads and streams, with one
 *     exception: a plan with clamp_below_max > 0 owns a per-row scratch (row maxima and minima), so
 *     two mmfe_run* calls on such a plan must not overlap on the device (issue them on one stream, or
 *     create one plan per stream); a call handles at most 2^20 rows.
 *   - There is no CPU path: creating a plan without a usable gfx950 device fails.
 */
#ifndef MMFE_H
#define MMFE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMFE_ABI_VERSION 7

/* error codes */
#define MMFE_OK            0
#define MMFE_EINVAL       -1   /* bad argument / unsupported parameter combination */
#define MMFE_ENODEVICE    -2   /* no usable HIP device */
#define MMFE_EHIP         -3   /* a HIP runtime call failed */
#define MMFE_ENOMEM       -4

/* mmfe_spec.kind */
#define MMFE_KIND_FBANK     1
#define MMFE_KIND_LOGSPEC   2
#define MMFE_KIND_STFT_MAG  3
#define MMFE_KIND_FRAME     4
#define MMFE_KIND_RESAMPLE  5  /* rational-ratio polyphase FIR resampler: rows of n samples -> rows of ceil(n * up / down) samples.
                                  Stands where the reference resamples on the CPU before anything else happens:
                                  librosa.load(file, sr=16000, duration=10) (classification/vggsound/dataset/vggsound.py:114) and
                                  ffmpeg -ar 24000 (classification/recognition/dataset/extract_audio.py:6-13).
                                  Fields: n_fft = up, hop = down (coprime), win_len = number of FIR taps (odd), `window` = the taps
                                  (low-pass of cutoff 1/max(up, down), already multiplied by up), out_frames = 0 or the number of
                                  output samples rows are cut / zero-padded to (duration=...), in_dtype float32 or 16-bit PCM;
                                  every other field 0 / MMFE_*_NONE.  Alignment, output count and edge handling (zeros beyond the
                                  row) are those of scipy.signal.resample_poly(x, up, down).  Output: (n_rows, samples), float32 */

/* mmfe_spec.pad_mode: how samples outside [0, n) are read */
#define MMFE_PAD_NONE     0    /* never read outside (snip-edges framing) */
#define MMFE_PAD_ZERO     1
#define MMFE_PAD_REFLECT  2

/* mmfe_spec.frames_mode: number of frames for an n-sample row */
#define MMFE_FRAMES_SNIP    0  /* n < win_len ? 0 : 1 + (n - win_len) / hop          (kaldi snip_edges) */
#define MMFE_FRAMES_CENTER  1  /* 1 + n / hop, frame t centred on sample t*hop        (librosa center=True) */
#define MMFE_FRAMES_SCIPY   2  /* 1 + ceil(n / hop), zero-extended by win_len/2       (scipy boundary='zeros', padded) */
#define MMFE_FRAMES_CEIL    3  /* n <= win_len ? 1 : 1 + ceil((n - win_len) / hop), zero pad split floor/ceil (frame_signal) */

/* mmfe_spec.log_kind */
#define MMFE_LOG_NONE       0
#define MMFE_LOG_LN_MAX     1  /* ln(max(x, log_floor))      kaldi */
#define MMFE_LOG_LN_ADD     2  /* ln(x + log_floor)          EPIC  */
#define MMFE_LOG_LOG10_MAX  3  /* log10(max(x, log_floor))   long-utterance log-mel */

/* mmfe_spec.layout */
#define MMFE_LAYOUT_TIME_MAJOR  0  /* out[row][frame][feature] */
#define MMFE_LAYOUT_FEAT_MAJOR  1  /* out[row][feature][frame] */

/* mmfe_spec.out_dtype: element type of the output tensor */
#define MMFE_OUT_F32   0
#define MMFE_OUT_F16   1  /* IEEE half, round to nearest even: what the encoders' @autocast() casts the spectrogram to
                           (classification/vggsound/models/vit_model.py:388); FBANK and LOGSPEC kinds (the any-length DFT kernel not
                           together with clamp_below_max) */
#define MMFE_OUT_BF16  2  /* bfloat16, round to nearest even; same kinds */

/* mmfe_spec.in_dtype: element type of the input rows */
#define MMFE_IN_F32   0
#define MMFE_IN_S16   1  /* signed 16-bit PCM, sample value = s / 32768 (exact in float32): what torchaudio.load
                           (classification/vggsound/dataset/vggsound.py:66) and librosa.load
                           (classification/recognition/dataset/epic_kitchen.py:135) make of the 16-bit FLAC / pcm_s16le
                           WAV files the reference stores (vggsound.py:120, recognition/dataset/extract_audio.py:9);
                           FBANK, LOGSPEC and RESAMPLE kinds */
#define MMFE_IN_F64   2  /* FRAME kind only: frame_signal multiplies in the dtype of its operands
                           (classification/vggsound/utils/transforms.py:24-25, 58-59), so float64 rows take a float64 window
                           (spec.window_f64 = 1) and give float64 frames, one correctly rounded product each */
#define MMFE_IN_F16   3  /* FRAME kind only: IEEE half rows, float window holding half values, half frames (the exact product
                           rounded once to nearest even, as torch multiplies halves) */
#define MMFE_IN_BF16  4  /* FRAME kind only: bfloat16 rows, likewise */

typedef struct mmfe_spec {
    int32_t struct_size;   /* = sizeof(mmfe_spec) */
    int32_t kind;          /* MMFE_KIND_* */
    int32_t n_fft;         /* DFT length in [2, 4096]; 512, 400 and (STFT_MAG) 32 have radix kernels.  Other lengths run on the
                              any-length kernel, which needs 32*n_fft + 128*(win_len + 1) bytes of LDS (plus 64*(n_fft/2 + 2) when
                              n_fft > 510): plans beyond the CU's 160 KiB (win_len near n_fft above about 850) are refused at creation */
    int32_t win_len;       /* number of window taps (<= n_fft); FRAME: frame_length */
    int32_t win_off;       /* position of tap 0 inside the n_fft-long DFT input (zeros elsewhere) */
    int32_t hop;
    int32_t pad_mode;      /* MMFE_PAD_* */
    int32_t frames_mode;   /* MMFE_FRAMES_* */
    int32_t n_mel;         /* 0 = no mel projection, features are the n_fft/2+1 DFT bins */
    int32_t remove_dc;     /* subtract the per-frame mean before pre-emphasis (kaldi) */
    float   preemph;       /* 0 = off; y[j] = x[j] - preemph * x[max(j-1,0)] */
    int32_t power;         /* 1 = |X|, 2 = |X|^2 */
    float   spec_scale;    /* multiplies X before |.| (scipy 'spectrum' scaling = 1/sum(window)); 1 = off */
    int32_t log_kind;      /* MMFE_LOG_* */
    float   log_floor;
    float   out_shift;     /* result = (value + out_shift) * out_scale */
    float   out_scale;
    int32_t out_frames;    /* 0 = natural frame count; else rows are cut / padded to this many frames */
    float   pad_value;     /* value (before shift/scale) of padded frames */
    int32_t drop_last;     /* drop the last natural frame (long-utterance log-mel) */
    float   clamp_below_max; /* > 0: per-row clamp to (row max - clamp_below_max) before shift/scale */
    int32_t layout;        /* MMFE_LAYOUT_* */
    int32_t window_f64;    /* 1: the `window` argument of mmfe_plan_create points to doubles (librosa / scipy build
                              their windows in float64 and multiply in float64); 0: floats */
    int32_t out_dtype;     /* MMFE_OUT_*: the arithmetic stays float32, only the stored value is rounded */
    int32_t in_dtype;      /* MMFE_IN_*: 16-bit PCM results are bit-identical to running on the decoded float32 samples; the FRAME
                              kind stores frames of its rows' own type */
} mmfe_spec;

typedef struct mmfe_plan mmfe_plan;

/* Library / ABI version (MMFE_ABI_VERSION of the build). */
int mmfe_abi_version(void);

/* Last error message of the calling thread ("" if none). */
const char* mmfe_last_error(void);

/* Number of usable devices (0 when there is none; never fails). */
int mmfe_device_count(void);

/* Output shape for rows of n_samples samples: number of frames and features per frame.
 * Pure host arithmetic, no device needed.
 * Stands for the shape rules of kaldi snip-edges framing, librosa centred framing,
 * scipy.signal.stft and frame_signal (transforms.py:32-37). */
int mmfe_spec_out_shape(const mmfe_spec* spec, int64_t n_samples, int64_t* n_frames, int64_t* n_feat);

/* Build a plan: validates the spec, uploads the constant tables to `device`.
 *   window        host, win_len floats (doubles when spec->window_f64), the values exactly as the
 *                 reference library builds them
 *   band_start    host, n_mel ints: first DFT bin of each mel filter          (NULL when n_mel == 0)
 *   band_len      host, n_mel ints: number of consecutive bins of each filter (0 allowed)
 *   band_weights  host, sum(band_len) floats, filter after filter
 * Replaces the per-call table construction inside kaldi.fbank / librosa.stft / scipy.signal.stft. */
int mmfe_plan_create(const mmfe_spec* spec, const void* window,
                     const int32_t* band_start, const int32_t* band_len, const float* band_weights,
                     int device, mmfe_plan** out_plan);

int mmfe_plan_destroy(mmfe_plan* plan);

/* Run the transform on a batch.
 *   in         device, n_rows rows of n_samples samples (float, or int16_t when spec.in_dtype = MMFE_IN_S16), row r
 *              starting in_stride elements after row r - 1
 *   out        device, contiguous (n_rows, frames, feat) or (n_rows, feat, frames) per spec.layout,
 *              frames/feat as reported by mmfe_spec_out_shape (after out_frames / drop_last); elements of
 *              spec.out_dtype (float unless stated otherwise)
 *   hip_stream hipStream_t or NULL
 * Stands where the reference calls kaldi.fbank (vggsound.py:45), librosa.stft + log
 * (epic_kitchen.py:119-125), scipy.signal.stft + abs (adbox.py:20) or frame_signal. */
int mmfe_run(const mmfe_plan* plan, const void* in, int64_t in_stride, int64_t n_rows, int64_t n_samples,
             void* out, void* hip_stream);

/* mmfe_run for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc): same arguments, same results.  No counterpart in the reference.  The one scratch buffer the
 * kernels need, the per-row maxima and minima of clamp_below_max plans, is a static device array inside the library
 * (2 x MMFE_NOALLOC_MAX_ROWS floats = 512 KiB per GPU, held until the process ends) instead of the plan's own buffer;
 * the call allocates and frees nothing.  Limits: a clamp_below_max plan takes at most MMFE_NOALLOC_MAX_ROWS rows per
 * call (more: MMFE_EINVAL, nothing is launched); every call on a GPU uses that GPU's one array, so two calls with
 * clamp_below_max plans that overlap on one GPU race (issue them on one stream).  Plans without clamp_below_max use no
 * scratch and run exactly as under mmfe_run. */
#define MMFE_NOALLOC_MAX_ROWS 65536
int mmfe_run_noalloc(const mmfe_plan* plan, const void* in, int64_t in_stride, int64_t n_rows, int64_t n_samples,
                     void* out, void* hip_stream);

/* Same transform, but row r is the slice [offsets[r], offsets[r] + n_samples) of recording
 * rec_index[r] in a packed device buffer (recording i starts rec_start[i] elements into recs).  Stands for
 * the slicing in EPICKitchen._extract_sound_feature (epic_kitchen.py:128-154) feeding _log_specgram. */
int mmfe_run_slices(const mmfe_plan* plan, const void* recs, const int64_t* rec_start, const int32_t* rec_index,
                    const int64_t* offsets, int64_t n_rows, int64_t n_samples, void* out, void* hip_stream);

/* Ragged batches in ONE launch: row r holds row_len[r] samples (device array of n_rows ints, each clamped to
 * [0, n_samples]); n_samples is the length of the longest row and fixes the output shape exactly as in mmfe_run.  Every
 * row is transformed as if it were passed alone with its own length (its own padding at its own end, its own frame
 * count); frames a shorter row does not have are filled like the frames beyond out_frames' natural count
 * ((pad_value + out_shift) * out_scale).  Stands for a DataLoader batch that mixes lengths: the reference feeds
 * interior (30 672-sample) and edge (30 696-sample) EPIC slices through the same _log_specgram
 * (epic_kitchen.py:143-154, both give 256 frames) and pads VGGSound clips of any length to 384 frames
 * (vggsound.py:48-56).  FBANK and LOGSPEC kinds without clamp_below_max.  The first win_len elements of `in` / `recs`
 * must be readable (the kernels park masked lanes there). */
int mmfe_run_ragged(const mmfe_plan* plan, const void* in, int64_t in_stride, const int32_t* row_len, int64_t n_rows,
                    int64_t n_samples, void* out, void* hip_stream);
int mmfe_run_slices_ragged(const mmfe_plan* plan, const void* recs, const int64_t* rec_start, const int32_t* rec_index,
                           const int64_t* offsets, const int32_t* row_len, int64_t n_rows, int64_t n_samples, void* out,
                           void* hip_stream);

/* Sum and sum of squares (float64) of n device floats, accumulated into sums[0], sums[1] (device,
 * caller zeroes them).  Stands for the two DataLoader passes of cal_norm (vggsound.py:85-107). */
int mmfe_feature_sums(const float* x, int64_t n, double* sums, int device, void* hip_stream);

/* Name of the kernel family that serves the plan ("fbank512", "rfft400", "dftgen", "stft32", "frame"); a static
 * string, never NULL.  Diagnostic: the reference has one code path per transform, so nothing there corresponds;
 * tests use it to make sure a parameter set reaches the kernel they mean to exercise. */
const char* mmfe_plan_kernel(const mmfe_plan* plan);

/* ---- patch embedding: the consumer's first layer --------------------------------------------------------------
 * The encoders turn the spectrogram into tokens with one 16 x 16 / stride-16 convolution, PatchEmbed.forward
 * (classification/vggsound/models/vit_model.py:193-199) with the projection AudioTransformer installs at vit_model.py:435:
 *     tokens = proj(x).flatten(2).transpose(1, 2)          x: (B, 1, H, W)  ->  (B, P, embed_dim), P = (H/16) * (W/16)
 * computed under @autocast() (vit_model.py:388) in a 16-bit type.  mmfe_patch_embed_create packs Conv2d.weight
 * ((embed_dim, 1, 16, 16) contiguous = embed_dim rows of 256 floats) and Conv2d.bias (embed_dim floats or NULL), rounded to
 * compute_dtype (MMFE_OUT_F16: what autocast uses on a GPU; MMFE_OUT_BF16), into matrix-pipe operand order on `device`;
 * the image (in_dtype MMFE_IN_F32) is rounded to compute_dtype on the way in, or already holds that type (in_dtype MMFE_IN_F16 /
 * MMFE_IN_BF16: the spectrogram as the front-end kernels store it with out_dtype float16 / bfloat16, bit-identical to the cast);
 * products are accumulated in float32, tokens are stored as out_dtype.
 * Limits: patch = 16, embed_dim 384 or 768, width a multiple of 4 (8 for 16-bit images); rows / columns beyond the last whole
 * patch are ignored, as the convolution ignores them. */
typedef struct mmfe_patch_embed mmfe_patch_embed;
int mmfe_patch_embed_create(int32_t height, int32_t width, int32_t patch, int32_t embed_dim, const float* weight, const float* bias,
                            int32_t in_dtype, int32_t compute_dtype, int32_t out_dtype, int device, mmfe_patch_embed** out_handle);
int mmfe_patch_embed_destroy(mmfe_patch_embed* handle);
/* New weights for an existing handle, read from DEVICE memory (the Conv2d parameters of a model in training, after an optimizer
 * step: vanilla_single.py:75-80): weight_dev (embed_dim, 256) float32 contiguous, bias_dev embed_dim floats or NULL (zeros).  One
 * small kernel on hip_stream re-packs and rounds them as mmfe_patch_embed_create does on the host: no allocation, no
 * synchronisation; calls of mmfe_patch_embed_run issued later on the same stream use the new weights. */
int mmfe_patch_embed_update(mmfe_patch_embed* handle, const float* weight_dev, const float* bias_dev, void* hip_stream);
/* P and embed_dim of the tokens one image yields. */
int mmfe_patch_embed_shape(const mmfe_patch_embed* handle, int64_t* n_patches, int64_t* embed_dim);
/* x: device, n_batch images (elements of in_dtype) of height x width (unit column stride, row stride = width), image b starting
 * x_batch_stride elements after image b - 1 (16-byte aligned); out: device, (n_batch, P, embed_dim) of out_dtype, image b's
 * tokens starting out_batch_stride elements after image b - 1's (so that a caller can leave room for a class token in
 * front, vit_model.py:382-383).  Asynchronous on hip_stream, allocates nothing. */
int mmfe_patch_embed_run(const mmfe_patch_embed* handle, const void* x, int64_t x_batch_stride, int64_t n_batch, void* out,
                         int64_t out_batch_stride, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* MMFE_H */
