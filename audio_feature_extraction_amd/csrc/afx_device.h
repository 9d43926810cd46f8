// Device-visible structures and kernel launcher prototypes (host callable).
#pragma once
#include <cstdint>

#include <hip/hip_runtime_api.h>

#include "afx.h"
#include "afx_consts.h"

namespace afx {

constexpr int kFramesPerBlock = 16;   // F: frames one workgroup of k_frames owns
constexpr int kWaves = 4;             // waves per workgroup in k_frames

// Host-built, per clip.
struct ClipDesc {
  int64_t off;         // element offset of the clip in the packed sample buffer
  int64_t len;         // N samples
  int64_t frame_base;  // first padded frame slot of this clip (multiple of 16)
  int64_t tblk_base;   // first trim-block slot of this clip
  int32_t tmax;        // 1 + N / hop   (frame count before trim)
  int32_t tpad;        // tmax rounded up to a multiple of 16
  int32_t blk_base;    // first k_frames block (16 frames) of this clip
  int32_t pad_;
};

// One per 16-frame block of k_frames, written by k_trim_decide once the kept span is known,
// so that k_frames needs a single 64-byte fetch per block (no dependent clip lookups).
// Staged index j (0 .. 15*hop + n_fft) is sample g0 + j of the clip, g0 = start + t0*hop - n_fft/2.
struct BlockDesc {
  int64_t sample_base;   // element index of staged sample 0 in the packed buffer (clip off + g0)
  int64_t frame_slot;    // clip frame_base + t0
  int64_t clip_off;      // element index of the clip's sample 0
  int32_t keep_lo, keep_hi;   // j kept by trim iff keep_lo <= j < keep_hi (else staged as zero)
  int32_t have_lo, have_hi;   // sample exists iff have_lo <= j < have_hi   (have_lo = -g0)
  int32_t clip, t0, T, active;
  int32_t pad_[2];
};
static_assert(sizeof(BlockDesc) == 64, "BlockDesc must be 64 bytes");

// Device-written, per clip (k_trim_decide), read by every later kernel.
struct ClipInfo {
  int64_t start, end;   // kept span [start, end) after trim (clip-relative)
  int32_t T;            // frames = 1 + (end - start) / hop
  int32_t status;       // afx_clip_status
  uint32_t lmax_ord;    // order-preserving uint image of max log-mel (atomicMax)
  uint32_t nonfinite;   // set by k_trim_blocks
};

struct DevTables {
  const float* window;   // n_fft
  const float* tw;       // n_fft/2 complex
  const float* post;     // n_fft/2 complex
  const float4* mel_coef; // per (group, row): a_lo, b_lo, a_hi, b_hi of the filter's triangle
  const float* mel_koff;  // per (group, row): kmin - kc
  const int4* mel_grp;   // per 16-filter group: kmin, nblk, first block, group id
  const int4* mel_items;  // [4 waves][8 items][2 x int4]: group, b0, nb, role | slot, nslots, 0, 0
  int32_t mel_item_cnt[4];
  int32_t mel_n_slots;    // LDS partial-sum slots in use (0: no group is split)
  const float* mel_taps;  // k_frames2: oct-padded tap weights
  const int32_t* mel_meta; // k_frames2: per filter k0 | n4 << 10 | offset << 15
  int32_t mel_ntaps;      // floats in mel_taps; 0: tables unusable, generic kernel
  const float* dctA;     // DCT-II rows as MFMA A images
  const float* dctP;     // the same, permuted for k_dct16's 16-byte tile loads (nullptr: not applicable)
  const float* dctS;     // k_tail<., SYM>: A images of the folded contraction -- even coefficients' groups, then odd ones' (nullptr: n/a)
  int32_t n_groups;      // ceil(n_mels / 16)
  int32_t n_cgroups;     // ceil(n_mfcc / 16)
};

struct KParams {
  int32_t n_fft, hop, n_mels, n_mfcc;
  int32_t trim_frame, trim_hop;
  float preemph_b1;      // float32(-coef)
  float trim_top_db, top_db, amin;
  int32_t flags;         // AFX_FLAG_*
  int32_t fmt;           // AFX_FMT_*
  int32_t rms_sub;       // > 0: k_trim_blocks keeps sums per hop-sized sub-block (rms_sub per trim block) and
                         // k_trim_decide derives the RMS rows from them (k_frames2 path); 0: the frame kernel computes RMS
};

// dynamic LDS bytes k_frames needs for (n_fft, hop); 0 if n_fft unsupported
size_t frames_lds_bytes(int n_fft, int hop);

hipError_t launch_trim_blocks(hipStream_t s, const void* samples, const ClipDesc* clips, ClipInfo* info,
                              float* bsum, int n_clips, int max_tblocks, const KParams& kp);
// samples: the batch's packed samples when rms_rows is wanted (a clip of fewer than nine frames gets its RMS rows here
// on the shapes whose frame kernel computes them); may be nullptr otherwise
hipError_t launch_trim_decide(hipStream_t s, const ClipDesc* clips, ClipInfo* info, const float* bsum,
                              BlockDesc* blocks, float* rms_rows, int n_clips, const KParams& kp,
                              const void* samples = nullptr);
// true when launch_frames will take the n_fft = 1024 / hop = 256 kernel (k_frames2) for this plan
bool frames2_eligible(const KParams& kp, const DevTables& tb);
hipError_t launch_frames(hipStream_t s, const void* samples, ClipInfo* info,
                         const BlockDesc* blocks, int nblocks, const DevTables& tb, const KParams& kp,
                         float* logmel, float* rms_rows, int grid, unsigned long long* stamps = nullptr);
constexpr int kStampPhases = 12;
hipError_t launch_dct(hipStream_t s, const ClipDesc* clips, const ClipInfo* info, const DevTables& tb,
                      const KParams& kp, const float* logmel, float* mfcc, int n_clips, int max_tmax,
                      bool frame_major = false, bool spec = false);
hipError_t launch_finish(hipStream_t s, ClipInfo* info, int n_clips, int* counters, unsigned* flag_dev, unsigned seq);
// stats / info_out may be host memory the device can write (the batch's results land where the caller reads them)
hipError_t launch_stats(hipStream_t s, const ClipDesc* clips, const ClipInfo* info, const KParams& kp,
                        const float* mfcc, const float* rms_rows, float* stats, float* frames_out,
                        const int64_t* frame_offsets, int n_clips, ClipInfo* info_out = nullptr);
hipError_t launch_preemph(hipStream_t s, const float* y, float* out, int64_t n, float b1);
// afx_tail.hip: clamp + DCT + statistics of a clip in one workgroup, from the frame-major log-mel spill of the wave-level
// frame kernels; the MFCC rows are never written.  Replaces launch_dct + launch_stats when no per-frame output is wanted.
bool tail_eligible(const KParams& kp, const DevTables& tb);
hipError_t launch_tail(hipStream_t s, const ClipDesc* clips, const ClipInfo* info, const DevTables& tb, const KParams& kp,
                       const float* logmel, const float* rms_rows, float* stats, ClipInfo* info_out, int n_clips, int spec,
                       int n_cu);

}  // namespace afx
