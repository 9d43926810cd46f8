// extract_f0 (pYIN) on the device: structures shared by afx_f0.hip, afx_tables.cpp and afx_api.cpp.
// Reference: audio_feature_extraction_toolkit/core/feature_extractor.py:76-114 (librosa.pyin at its defaults).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "afx_device.h"

namespace afx {

constexpr int kF0Thresholds = 100;     // librosa.pyin n_thresholds
constexpr int kF0FramesPerBlock = 16;  // frames one workgroup of k_f0_yin owns at most (8 where that fits a third workgroup per CU)

struct F0Params {
  int32_t n_fft, hop, W;               // frame_length, hop_length, win_length = frame_length / 2
  int32_t min_period, max_period;      // floor(sr / fmax), min(ceil(sr / fmin), n_fft - W - 1)
  int32_t n_tau;                       // max_period + 1 lags of the difference function
  int32_t n_tau_pad;                   // row stride of the energy rows (multiple of 64)
  int32_t n_lag;                       // max_period - min_period + 1 lags kept after normalisation
  int32_t n_bins;                      // pitch bins (10 per semitone from fmin)
  int32_t R;                           // ceil(n_tau / 64): lags per lane
  int32_t slots;                       // ceil(n_lag / 64): trough slots per lane
  int32_t cap;                         // candidate capacity per frame (>= number of possible troughs)
  int32_t band;                        // max_semitones_per_frame * bins_per_semitone (transition half-width)
  int32_t epb;                         // frames per block of k_f0_energy (64, 32 or 16)
  int32_t debug;                       // AFX_F0_DEBUG: timing-only ablation switches (results are wrong when set)
  double sr, fmin;
  double tiny;                         // np.finfo(float64).tiny
  double c0;                           // log(tiny): log of a zero probability
  double no_trough_prob;
  double bins_per_octave;              // 12 * bins_per_semitone
};

struct F0Tables {
  const double* thr;      // [101] np.linspace(0, 1, 101)
  const double* beta;     // [100] diff(beta.cdf(thr, 2, 18))
  const double* cumbeta;  // [101] cumbeta[n] = sum(beta[:n])
  const double* bfact;    // [cap + 1] (1 - e^-2) / (1 - e^-2n)  (scipy.stats.boltzmann.pmf normaliser)
  const double* bexp;     // [cap + 1] e^-2k
  const double* lt;       // [2][2 * band + 1][2 * band + 1] log(switch * local[row class][d] + tiny)
  const double* ltw;      // [2][2 * band + 1] the interior row as the band walk meets it: entry 2 band - e; stay row, then switch row
                          // (contiguous, so that the walk's scalar loads take several weights at once)
  const double* freqs;    // [n_bins] fmin * 2^(b / 120)
};

// (maximum, first arg-max) of one Viterbi value column: the best out-of-band source of the next step
struct VitBest { double value; int32_t arg; int32_t pad; };

struct HostF0Tables {
  F0Params p{};
  std::vector<double> thr, beta, cumbeta, bfact, bexp, lt, ltw, freqs;
};

// builds every table from (sr, n_fft, hop, fmin, fmax); returns false when the combination is unsupported
bool build_f0_tables(int sr, int n_fft, int hop, double fmin, double fmax, HostF0Tables& t, std::string& why);

size_t f0_energy_lds_bytes(const F0Params& fp);
size_t f0_yin_lds_bytes(const F0Params& fp);
size_t f0_viterbi_lds_bytes(const F0Params& fp);
size_t f0_backtrack_lds_bytes(const F0Params& fp);   // also requires 2 band + 1 <= 64 (a lane per source of the band)

// per-frame candidate record sizes (device workspace)
inline size_t f0_cand_bins_bytes(const F0Params& fp, int64_t frames) { return (size_t)frames * fp.cap * sizeof(int16_t); }
inline size_t f0_cand_prob_bytes(const F0Params& fp, int64_t frames) { return (size_t)frames * fp.cap * sizeof(double); }
// Viterbi value columns kept for back-tracking: 2 n_bins doubles per frame
inline size_t f0_vrows_bytes(const F0Params& fp, int64_t frames) { return (size_t)frames * 2 * fp.n_bins * sizeof(double); }

hipError_t launch_f0_prep(hipStream_t s, const void* samples, const ClipDesc* clips, const ClipInfo* info,
                          float* ysig, int n_clips, int64_t max_len, const KParams& kp);
hipError_t launch_f0_energy(hipStream_t s, const float* ysig, const ClipDesc* clips, const ClipInfo* info,
                            float* energy, int n_clips, int max_tmax, const F0Params& fp);
hipError_t launch_f0_yin(hipStream_t s, const float* ysig, const ClipDesc* clips, const ClipInfo* info,
                         const float* energy, const F0Tables& tb, const F0Params& fp,
                         int32_t* cand_cnt, double* cand_vp, int16_t* cand_bin, double* cand_prob /* nullable: diagnostics */,
                         double* cand_lp, double* cand_lu, int n_clips, int max_tmax);
hipError_t launch_f0_viterbi(hipStream_t s, const ClipDesc* clips, const ClipInfo* info, const F0Tables& tb,
                             const F0Params& fp, const int32_t* cand_cnt,
                             const int16_t* cand_bin, const double* cand_lp, const double* cand_lu,
                             double* vrows, VitBest* vbest,
                             uint16_t* states, double* out_stats, double* out_f0, const int64_t* f0_offsets,
                             int n_clips);

hipError_t launch_zcr(hipStream_t s, const float* ysig, const ClipDesc* clips, const ClipInfo* info, int n_fft, int hop,
                      double* out, const int64_t* out_offsets, int n_clips, int max_tmax);

}  // namespace afx
