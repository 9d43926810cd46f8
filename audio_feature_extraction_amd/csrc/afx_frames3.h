// k_frames3 (afx_frames3.hip): tables, LDS geometry and launcher.  Internal to libafx.so.
#pragma once
#include <cstdint>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "afx_device.h"

namespace afx {

constexpr int kF3ExFloats = 2176;          // per-wave LDS image: 64 rows x 17 float2 (exchange 1, padded) = 8704 B
constexpr int kF3TabFloats = 2048;         // pass-2 twiddles 128 float2 + two last-pass tables of 7 x 64 float2
constexpr int kF3MaxRounds = 8;            // mel schedule rounds (table capacity)
constexpr int kF3KernelRounds = 4;         // rounds the kernel unrolls; build_f3_mel never plans more
constexpr int kF3MaxBatches = 8;           // batches of 4 taps per lane and round

// Mel schedule of k_frames3.  One frame pair at a time, a lane accumulates `4 * nb` consecutive taps of one
// filter; `width` adjacent lanes share a filter (contiguous tap chunks) and are summed by DPP.  Rounds are
// chosen by build_f3_mel (afx_tables.cpp) so that the padded tap count over all rounds is small.
struct HostF3Mel {
  int32_t rounds = 0;
  int32_t nb[kF3MaxRounds] = {};           // batches per lane
  int32_t width[kF3MaxRounds] = {};        // lanes per filter: 1, 2, 4, 8
  int32_t woff[kF3MaxRounds] = {};         // float offset of the round's weights
  std::vector<float> w;                    // [round][batch][lane][4]
  std::vector<int32_t> meta;               // [round][lane]: first bin | filter << 11 | owner << 20
  bool usable = false;
};

struct F3Tables {
  const float* window;      // n_fft floats
  const float* w1024;       // twiddle source of the complex FFT: exp(-2 pi i k / 1024), k < 512 (n_fft 1024, 2048);
                            // exp(-2 pi i k / 512), k < 256 (n_fft 512), as float2
  const float* w2048;       // n_fft 2048 only: exp(-2 pi i k / 2048), k < 1024 (real-FFT split)
  const float* mel_w;
  const int32_t* mel_meta;
  int32_t mel_rounds;
  int32_t mel_wfloats;
  int32_t mel_all_own;      // every (round, lane) owns a filter: the straight-line schedule stores without an owner test
  int32_t mel_own_w1;       // the same for the rounds of width 1 alone (k_frames3s's compiled-in schedule mixes widths)
  uint32_t mel_rp[kF3MaxRounds];   // per round: batches | width << 4 | weight offset (floats) << 8
};

// mel_dense: n_mels x n_bins (librosa float32 values); max_slot: highest bin slot of the image a padded tap may read;
// lanes: lanes that share one spectrum (64, or 32 when a wave holds two); align: bins per 16-byte read (2: (A, B) pairs
// of a frame pair, 4: one frame)
void build_f3_mel(const std::vector<float>& mel_dense, int n_mels, int n_bins, int max_slot, int lanes, int align, HostF3Mel& out);

size_t frames3_lds_bytes(int waves, const F3Tables& ft);
bool frames3_eligible(const KParams& kp, const F3Tables& ft);
int frames3_waves(const F3Tables& ft);
// spec = true: the speculative first launch over the host-built absolute blocks (emits bsum / blockmax);
// spec = false: blocks come from a device-built list whose length is *nblocks_dev (nblocks = its capacity).
// work_ctr: a zeroed device int -> the waves take the second half of the list by ticket (runs that shrink towards the
// end) instead of equal shares; nullptr -> equal contiguous shares
hipError_t launch_frames3(hipStream_t s, const void* samples, ClipInfo* info, const BlockDesc* blocks, int nblocks,
                          const int* nblocks_dev, const F3Tables& ft, const KParams& kp, float* logmel,
                          float* blockmax, float* bsum, bool spec, int* work_ctr, int n_cu);
// the same for n_fft 2048 / hop 512 (afx_frames3s.hip: one frame per FFT, real-FFT split) and n_fft 512 / hop 128
// (afx_frames3d.hip: two frame pairs per wave)
size_t frames3s_lds_bytes(int waves, const F3Tables& ft);
size_t frames3d_lds_bytes(int waves, const F3Tables& ft);
hipError_t launch_frames3s(hipStream_t s, const void* samples, ClipInfo* info, const BlockDesc* blocks, int nblocks,
                           const int* nblocks_dev, const F3Tables& ft, const KParams& kp, float* logmel,
                           float* blockmax, float* bsum, bool spec, int* work_ctr, int n_cu);
hipError_t launch_frames3d(hipStream_t s, const void* samples, ClipInfo* info, const BlockDesc* blocks, int nblocks,
                           const int* nblocks_dev, const F3Tables& ft, const KParams& kp, float* logmel,
                           float* blockmax, float* bsum, bool spec, int* work_ctr, int n_cu);
// spectral descriptors (k_frames3s<DESC>): librosa.feature.spectral_contrast's octave bands as bin ranges [lo, hi] of the
// sub-band and the number of magnitudes averaged at either end
struct SpecBands {
  int32_t lo[8], hi[8], cnt[8];
  float hz_per_bin, roll_percent;
};
constexpr int kSpecFloats = 17;            // per frame: centroid, bandwidth, rolloff, valley[7], peak[7]
hipError_t launch_spectral(hipStream_t s, const void* samples, ClipInfo* info, const BlockDesc* blocks, int nblocks,
                           const F3Tables& ft, const KParams& kp, float* desc_out, const int64_t* desc_offs,
                           const SpecBands& sb, int n_cu);
// dispatch on kp.n_fft
hipError_t launch_frames3_any(hipStream_t s, const void* samples, ClipInfo* info, const BlockDesc* blocks, int nblocks,
                              const int* nblocks_dev, const F3Tables& ft, const KParams& kp, float* logmel,
                              float* blockmax, float* bsum, bool spec, int* work_ctr, int n_cu);
// the speculative launch's block list (every absolute 16-frame block of every clip) from the clip records, on the device
hipError_t launch_build_blocks3(hipStream_t s, const ClipDesc* clips, int n_clips, int nblocks, BlockDesc* blocks,
                                const KParams& kp);
constexpr int kF3ItemsPerClip = 6;         // redo-list capacity per clip (k_trim_decide3)
hipError_t launch_trim_decide3(hipStream_t s, const ClipDesc* clips, ClipInfo* info, const float* bsum, const float* blockmax,
                               BlockDesc* items, int* n_items, int max_items, float* rms_rows, int n_clips, const KParams& kp);

}  // namespace afx
