/*
 * afx.h -- C ABI of libafx.so, the MI355X (gfx950) MFCC / RMS engine.
 *
 * The reference (chiy48308/audio_feature_extraction) has no FFI: its hot path is
 * three Python methods that call librosa.  This header is the seam a maintainer
 * binds with ctypes; each entry point names the reference
 * call(s) it replaces.  Paths are relative to the reference root,
 *   F = audio_feature_extraction_toolkit/core/feature_extractor.py
 *
 * Conventions
 *   - plain C, plain pointers and sizes, no torch / HIP types in signatures;
 *   - every function returns 0 (AFX_OK) or a negative afx_status; nothing throws
 *     across the ABI; afx_last_error() gives the text of the last failure on the
 *     calling thread;
 *   - the caller owns every buffer it passes; the library never frees them;
 *   - an afx_ctx is bound to one HIP device and owns one stream; it is not
 *     thread-safe -- use one ctx per worker thread (one per GPU);
 *   - all entry points are synchronous on return, except afx_extract_submit (its results are
 *     handed out by afx_extract_collect);
 *   - a per-clip failure is reported in out_status[] and never fails the batch
 *     (maps onto batch_process's per-file try/except, F:229-235).
 */
#ifndef AFX_H
#define AFX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AFX_VERSION 104      /* 104: afx_batch_geometry, afx_host_alloc / afx_host_free */

typedef enum afx_status {
  AFX_OK = 0,
  AFX_ERR_INVALID = -1,     /* bad argument / unsupported parameter combination */
  AFX_ERR_NO_DEVICE = -2,   /* no usable HIP device */
  AFX_ERR_HIP = -3,         /* a HIP runtime call or kernel launch failed */
  AFX_ERR_NOMEM = -4,
  AFX_ERR_UNSUPPORTED = -5  /* e.g. n_fft not a power of two in [256, 2048] */
} afx_status;

/* per-clip status written to out_status[] */
typedef enum afx_clip_status {
  AFX_CLIP_OK = 0,
  AFX_CLIP_TOO_SHORT = 1,  /* fewer than delta_width frames, or < 2 samples:
                              librosa.feature.delta raises ParameterError (F:137) */
  AFX_CLIP_NONFINITE = 2   /* NaN/Inf sample: librosa.util.valid_audio raises */
} afx_clip_status;

enum { AFX_WINDOW_HAMMING = 0, AFX_WINDOW_HANN = 1 };
enum { AFX_FMT_F32 = 0, AFX_FMT_S16 = 1 };           /* S16: value / 32768 (libsndfile) */
enum { AFX_MEM_HOST = 0, AFX_MEM_DEVICE = 1 };

/* flags for afx_extract_batch */
enum {
  AFX_FLAG_PREEMPH = 1,    /* apply pre-emphasis (F:69); off for extract_mfcc(y)/extract_energy(y) */
  AFX_FLAG_TRIM = 2        /* apply silence trim (F:72) */
};

/* Constructor arguments of AudioFeatureExtractor (F:10-17) plus the values the
 * reference hard-codes at its librosa call sites. */
typedef struct afx_params {
  int32_t sr;            /* F:11  */
  int32_t n_fft;         /* frame_length, F:12 -> n_fft of librosa.feature.mfcc (F:131) */
  int32_t hop;           /* F:13  */
  int32_t n_mfcc;        /* F:14  */
  int32_t n_mels;        /* librosa default 128 (the reference never passes it) */
  int32_t window;        /* AFX_WINDOW_HAMMING: F:133 */
  float   preemph;       /* F:17, 0.97 */
  float   trim_top_db;   /* 30, F:72 */
  int32_t trim_frame;    /* 2048, librosa.effects.trim default */
  int32_t trim_hop;      /* 512 */
  float   top_db;        /* 80, librosa.power_to_db default */
  float   amin;          /* 1e-10 */
  int32_t delta_width;   /* 9, librosa.feature.delta default */
  int32_t reserved;
  /* mel / MFCC option variants the reference's experiment extractors pass to librosa.filters.mel and
   * librosa.feature.mfcc (04_feature_extraction_experiment/audio_feature_extraction 2/audio_feature_extraction/
   * feature_extractor.py:148-181); the packaged class passes none, i.e. the defaults below */
  float   fmin;          /* 0: lowest mel band edge, Hz */
  float   fmax;          /* 0 = sr / 2: highest band edge, Hz */
  int32_t htk;           /* 0: Slaney mel scale; 1: HTK formula */
  float   lifter;        /* 0: none; L > 0: coefficient n (1-based) scaled by 1 + (L / 2) sin(pi n / L) */
} afx_params;

typedef struct afx_ctx afx_ctx;
typedef struct afx_plan afx_plan;

/* ---- library / device ---------------------------------------------------- */
int afx_version(void);
int afx_device_count(void);                       /* 0 when no GPU is visible */
const char* afx_last_error(void);                 /* thread-local, never NULL */

int afx_init(int device, afx_ctx** out);          /* binds device, creates the stream */
void afx_destroy(afx_ctx* ctx);

/* device memory helpers so that a binding needs no HIP API of its own */
int afx_malloc(afx_ctx* ctx, size_t bytes, void** out_dptr);
int afx_free(afx_ctx* ctx, void* dptr);
/* Page-locked host memory for batch buffers that are uploaded (batch_process packs a window of decoded files into one,
 * F:228-235): afx_memcpy_h2d from it is one DMA at link rate, from pageable memory the runtime stages it through its own
 * pinned buffer first (a host copy at a fifth of that).  Free with afx_host_free before afx_destroy. */
int afx_host_alloc(afx_ctx* ctx, size_t bytes, void** out);
int afx_host_free(afx_ctx* ctx, void* hptr);
int afx_memcpy_h2d(afx_ctx* ctx, void* dst_d, const void* src_h, size_t bytes);
int afx_memcpy_d2h(afx_ctx* ctx, void* dst_h, const void* src_d, size_t bytes);
int afx_synchronize(afx_ctx* ctx);

/* ---- plan: window / twiddle / sparse-mel / DCT tables for one parameter set  */
void afx_default_params(afx_params* p);           /* reference defaults (F:10-17, F:72, F:133) */
int afx_plan_create(afx_ctx* ctx, const afx_params* p, afx_plan** out);
void afx_plan_destroy(afx_plan* plan);

/* Host-only table builders (no device needed) -- what afx_plan_create uploads.
 * window[n_fft]; mel_dense[n_mels * (n_fft/2+1)] row-major (librosa.filters.mel
 * float32 values); dct[n_mfcc * n_mels] (ortho DCT-II rows).  Any pointer may be
 * NULL.  Replaces scipy.signal.get_window / librosa.filters.mel / scipy.fft.dct
 * table construction under librosa.feature.mfcc (F:127). */
int afx_build_tables(const afx_params* p, float* window, float* mel_dense, float* dct);

/* Host-only: the mel schedule of the wave-level frame kernel (k_frames3), for inspection and tests.
 * One frame pair at a time, a lane accumulates 4 * nb consecutive taps of one filter of librosa.filters.mel
 * (F:127); `width` adjacent lanes share a filter.  info[26] = rounds, weight floats, then per round (8 slots)
 * nb, width, weight offset; weights[info[1]] as [round][batch][lane][4]; meta[64 * rounds] per lane:
 * first bin | filter << 11 | owner << 20.  Any pointer may be NULL.  AFX_ERR_UNSUPPORTED when the
 * configuration has no such schedule. */
int afx_build_mel_schedule(const afx_params* p, int32_t* info, float* weights, int32_t* meta);

/* Host-only: the frame geometry of a ragged batch, as every batch entry point lays it out (batch_process hands over
 * files of any length, F:228-235).  records[4 * n_clips]: per clip the first frame slot, Tmax = 1 + length / hop, Tmax
 * padded to whole 16-frame blocks, first block index; totals[4]: frame slots, 16-frame blocks, trim blocks, largest
 * Tmax.  Either pointer may be NULL.  AFX_ERR_INVALID for negative or oversized offsets / lengths (the same check the
 * batch entry points make before anything is uploaded). */
int afx_batch_geometry(const afx_params* p, const int64_t* offsets, const int64_t* lengths, int n_clips,
                       int64_t* records, int64_t* totals);

/* ---- the hot path ---------------------------------------------------------
 * One pass of preprocess_audio -> extract_mfcc + extract_energy (F:194,198,199)
 * over a ragged batch of clips.
 *
 *   samples   packed clips (AFX_FMT_F32 float or AFX_FMT_S16 int16_t), in host
 *             or device memory (mem_kind); clip i is
 *             samples[offsets[i] .. offsets[i]+lengths[i])  (element units)
 *   offsets, lengths   host arrays, n_clips entries
 *   flags     AFX_FLAG_PREEMPH | AFX_FLAG_TRIM for extract_features semantics
 *   out_stats host, n_clips * (4*n_mfcc + 3) floats per clip:
 *             mfcc_mean[K] mfcc_std[K] mfcc_delta_mean[K] mfcc_delta2_mean[K]
 *             energy_mean energy_std energy_range        (F:141-150, F:171-178)
 *             A clip whose status is AFX_CLIP_TOO_SHORT because it has fewer than nine frames (the
 *             width-9 delta of F:137 cannot be formed) but at least two samples still has its three
 *             energy statistics, on every plan shape: extract_energy (F:153-179) only calls
 *             librosa.feature.rms.  Its MFCC entries are zero.  Every other non-OK clip: all zero.
 *   out_status host int32[n_clips]  (afx_clip_status)
 *   out_trim  host int64[2*n_clips] (start, end) of the kept span, or NULL
 *   out_nframes host int32[n_clips] T = 1 + (end-start)/hop, or NULL
 *   out_frames  NULL, or host float buffer for the per-frame matrices the
 *             reference computes and then reduces (F:127-138, F:164): clip i
 *             occupies rows of stride Tmax_i = 1 + lengths[i]/hop starting at
 *             float index frame_offsets[i]: (3*n_mfcc + 1) rows
 *             [mfcc K | delta K | delta2 K | rms 1], first T_i entries valid.
 *   frame_offsets host int64[n_clips] (ignored when out_frames is NULL)
 */
int afx_extract_batch(afx_plan* plan,
                      const void* samples, int sample_fmt, int mem_kind,
                      const int64_t* offsets, const int64_t* lengths, int n_clips,
                      int flags,
                      float* out_stats, int32_t* out_status,
                      int64_t* out_trim, int32_t* out_nframes,
                      float* out_frames, const int64_t* frame_offsets);

/* The same pass in two halves, for callers that keep the device busy across batches (batch_process walks its
 * files in windows, F:204-211): afx_extract_submit queues everything a batch needs -- kernels and the copies of its
 * results -- on the plan's stream and returns; afx_extract_collect waits for THAT batch only and fills the out_*
 * arrays given at submit (which, like `samples`, `offsets` and `lengths`, must stay valid until then).  Two plans of
 * one context submitted alternately from one thread run back to back on the device: the host's share of a batch
 * (collect, hand-out, next submit) falls under the other plan's kernels.  One batch per plan may be pending; n_clips
 * must be in [1, 32768]; afx_extract_batch(plan, ...) == submit + collect per 32768-clip chunk.
 * A batch whose clip offsets / lengths differ from the plan's previous batch costs the host one 48-byte record per
 * clip (pinned staging, asynchronous upload); the per-block work list is built on the device.  No stream
 * synchronisation happens inside submit unless workspace has to grow.  With out_frames given, the copy of the
 * per-frame matrices into the caller's (pageable) buffer is queued by submit and may make it block. */
int afx_extract_submit(afx_plan* plan,
                       const void* samples, int sample_fmt, int mem_kind,
                       const int64_t* offsets, const int64_t* lengths, int n_clips,
                       int flags,
                       float* out_stats, int32_t* out_status,
                       int64_t* out_trim, int32_t* out_nframes,
                       float* out_frames, const int64_t* frame_offsets);
int afx_extract_collect(afx_plan* plan);

/* extract_f0 (F:76-114): librosa.pyin(y, fmin, fmax, frame_length=n_fft, hop_length=hop, sr)
 * at librosa's defaults, of the same preprocessed clips (flags as for
 * afx_extract_batch: the reference feeds y_processed, F:195), reduced as F:97-107.
 *   out_f0stats host double[4 * n_clips]: f0_mean, f0_std, f0_missing_rate, f0_quality
 *             (0, 0, 1, 0 when no frame is voiced -- the reference's own branch F:103-107)
 *   out_status host int32[n_clips]: AFX_CLIP_OK or AFX_CLIP_NONFINITE
 *   out_f0    NULL, or host double buffer: clip i's per-frame f0 (NaN = unvoiced) in
 *             out_f0[f0_offsets[i] .. + T_i), T_i = 1 + kept_length_i / hop; the rest of the clip's
 *             1 + length_i / hop slots (frames trimmed away) is set to NaN
 * Device workspace inside the plan: about 14 KB per frame (the Viterbi value columns, 2 * n_bins
 * doubles per frame, are kept for back-tracking); batches are processed in chunks of at most
 * 1.28 M frames (about 18 GB).
 * Float64 throughout, except the running frame energy, which numpy accumulates in
 * float32 and which is reproduced add for add. */
int afx_f0_batch(afx_plan* plan,
                 const void* samples, int sample_fmt, int mem_kind,
                 const int64_t* offsets, const int64_t* lengths, int n_clips,
                 int flags, double fmin, double fmax,
                 double* out_f0stats, int32_t* out_status,
                 double* out_f0, const int64_t* f0_offsets);

/* Zero-crossing rate per frame of the same preprocessed clips (librosa.feature.zero_crossing_rate with
 * frame_length = n_fft, hop_length = hop, center=True): the frame-level feature the reference's experiment
 * scripts store beside mfcc / f0 / energy (04_feature_extraction_experiment/feature_extraction.py:340-352).
 *   out_zcr   host double buffer: clip i's T_i rates at out_zcr[zcr_offsets[i] ..]; at most 32768 clips per call */
int afx_zcr_batch(afx_plan* plan,
                  const void* samples, int sample_fmt, int mem_kind,
                  const int64_t* offsets, const int64_t* lengths, int n_clips, int flags,
                  double* out_zcr, const int64_t* zcr_offsets, int32_t* out_status);

/* Frame-level spectral descriptors the reference's experiment extractor takes from librosa at its defaults
 * (04_feature_extraction_experiment/feature_extractor.py:497-506: spectral_centroid, spectral_bandwidth,
 * spectral_rolloff, spectral_contrast; n_fft 2048, hop 512, centred, magnitude spectrum) of the given clips
 * (flags: AFX_FLAG_PREEMPH or 0; no trim -- pass the preprocessed signal, as the reference does).  The plan must have
 * n_fft 2048 / hop 512 (the reference calls librosa with its default Hann window: create the plan with AFX_WINDOW_HANN).
 *   out_desc  host float buffer: clip i has T_i = 1 + length_i / 512 rows of 17 floats at out_desc[desc_offsets[i] ..]:
 *             centroid (Hz), bandwidth (Hz, p = 2), rolloff (Hz, 85 %), then per octave band k = 0..6 of
 *             spectral_contrast(fmin = 200, n_bands = 6, quantile = 0.02) the valley[k] (mean of the smallest
 *             magnitudes) and after those the peak[k]; contrast = power_to_db(peak) - power_to_db(valley) with
 *             librosa's clip-global top_db clamp is left to the caller (it needs the maximum over the whole clip).
 *   AFX_ERR_UNSUPPORTED when sr <= 12800 (librosa: "Frequency band exceeds Nyquist") or the plan has another shape. */
int afx_spectral_batch(afx_plan* plan,
                       const void* samples, int sample_fmt, int mem_kind,
                       const int64_t* offsets, const int64_t* lengths, int n_clips, int flags,
                       float* out_desc, const int64_t* desc_offsets, int32_t* out_status);

/* Host-only (no device needed): the tables afx_f0_batch uploads, for inspection and tests.
 * info[8] = min_period, max_period, n_pitch_bins, band (transition half-width), candidate
 * capacity, lags kept, lags per lane, trough slots per lane.  beta[100] = Beta(2,18) mass of
 * each threshold interval; lt[2][2*band+1][2*band+1] = log(switch * local + tiny) per source-row
 * class (0 interior, 1..band low edge, band+1..2*band high edge) and offset; freqs[n_pitch_bins].
 * Any pointer may be NULL. */
int afx_f0_build_tables(int sr, int n_fft, int hop, double fmin, double fmax, int32_t* info,
                        double* beta, double* lt, double* freqs);

/* Host-only ingest for batch_process (load_audio, F:52 -> librosa.load -> soundfile), by `threads` native threads.
 * afx_wav_probe walks the RIFF chunks of n files: info[4 i ..] = format tag (1 PCM, 3 IEEE float; the sub-format of
 * WAVE_FORMAT_EXTENSIBLE), channels, sample rate, bits per sample; frames[i], data_off[i] = sample frames and byte offset
 * of the data chunk (cut at the file's end); status[i] = 0 ok, 1 not a usable RIFF/WAVE file, 2 cannot be opened / read.
 * afx_wav_read_s16 copies frames[i] 16-bit samples of file i from data_off[i] to out[offsets[i] ..] (for files the probe
 * found to be 16-bit PCM mono: a batch packed in place, uploaded as AFX_FMT_S16); status[i] = 0 or 2.  Every other sample
 * type, channel count or rate goes through the caller's own decoder. */
int afx_wav_probe(const char* const* paths, int n, int threads, int32_t* info /*[4n]*/, int64_t* frames,
                  int64_t* data_off, int32_t* status);
int afx_wav_read_s16(const char* const* paths, int n, int threads, const int64_t* data_off, const int64_t* frames,
                     int16_t* out, int64_t out_len, const int64_t* offsets, int32_t* status);

/* preprocess_audio(y) (F:58-74): pre-emphasis + trim of ONE host clip.
 * out_y receives the n pre-emphasised samples (host, n floats); the kept span
 * is out_y[*start .. *end). */
int afx_preprocess(afx_plan* plan, const float* y, int64_t n,
                   float* out_y, int64_t* start, int64_t* end, int32_t* status);

/* ---- measurement ---------------------------------------------------------
 * When enabled, afx_extract_batch brackets every kernel with HIP events on the
 * plan's stream.  afx_plan_get_timings returns, per kernel slot, the summed
 * milliseconds and launch count since the last reset. */
enum {
  AFX_K_TRIM_BLOCKS = 0,   /* two-pass pipeline: per-512-sample block sums of squares of y_pre; samples-read-once pipeline
                              (n_fft 1024 / hop 256): the second frame-kernel launch over the frames a trim cut touches */
  AFX_K_TRIM_DECIDE = 1,   /* per-clip max / threshold scan -> [start,end), T */
  AFX_K_FRAMES = 2,        /* fused framing+window+rFFT+power+mel+dB (+RMS)  -- dominant */
  AFX_K_DCT = 3,           /* top_db clamp + DCT-II */
  AFX_K_STATS = 4,         /* delta/delta2 + per-clip statistics */
  AFX_K_COUNT = 5
};
/* enable: 0 off, 1 every kernel, 2 the frame kernel (AFX_K_FRAMES) only -- an event pair per kernel costs stream time
 * (about 5 us each, more with several batches in flight), so a throughput run times the one kernel it reports */
int afx_plan_set_timing(afx_plan* plan, int enable);
int afx_plan_get_timings(afx_plan* plan, float* ms_sum /*[AFX_K_COUNT]*/,
                         int32_t* launches /*[AFX_K_COUNT]*/, int reset);
/* The same launches as (start, end) intervals in milliseconds on a clock common to every plan of the device (one
 * reference event per device), newest `cap` of them, oldest first; *count = intervals held since the last reset.
 * With several plans (streams) in flight on one GPU their kernels overlap: the time the GPU spends in a kernel is
 * the union of the plans' intervals, not their sum -- bench.py merges them for `roofline.avg_launch_ms`. */
int afx_plan_get_intervals(afx_plan* plan, int kernel_slot, double* start_ms, double* end_ms, int cap, int32_t* count);

#ifdef __cplusplus
}
#endif
#endif /* AFX_H */
