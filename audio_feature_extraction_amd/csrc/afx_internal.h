// Internal declarations shared by the host table builder, the kernel launchers
// and the C-ABI layer of libafx.so.  Not installed; include/afx.h is the ABI.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "afx.h"
#include "afx_consts.h"
#include "afx_frames3.h"

namespace afx {

// librosa.filters.mel for v_mfma_f32_16x16x4_f32: filters are taken 16 at a time (a
// "group"); a group touches the contiguous bin range [kmin, kmin + 4*nblk) and only
// those 16x4 blocks are visited (136 of 1032 at 22050 Hz / 1024).  The A operand is not
// tabulated: each lane evaluates its filter's triangle on the fly,
//   w(k) = max(0, min(a_lo + b_lo*(k - kc), a_hi + b_hi*(k - kc)))      (Slaney norm folded in)
// with the intercepts taken at the filter's peak bin kc, which keeps the float32 result
// within ~3e-7 of the filter peak of librosa's float32 table (tests/test_native_cpu.py).
struct MelBlocks {
  int32_t n_groups = 0;
  std::vector<int32_t> grp;     // 4 ints per group: kmin, nblk, first block, group id
  // k_frames' mel schedule: the groups' block ranges cut into work items balanced over the 4 waves.
  // A group much larger than a wave's fair share is split along its bins; all but one of its parts
  // ("writers") park their partial 16x16 sums in an LDS slot, the last ("reader") adds them.
  // 8 ints per item: group, first block, n blocks, role (0 whole, 1 writer, 2 reader), slot, n slots, 0, 0
  std::vector<int32_t> items;   // [4 waves][kMelMaxItems][8]
  int32_t item_cnt[4] = {0, 0, 0, 0};
  int32_t n_slots = 0;
  std::vector<float> coef;      // per (group, row): a_lo, b_lo, a_hi, b_hi
  std::vector<float> koff;      // per (group, row): kmin - kc  (k - kc = koff + 4*blk + q)
};

// Tap tables for k_frames2's mel stage: filters in groups of eight ("octs"), every filter of an oct padded
// with zero weights to the oct's longest tap count (a multiple of 4).
struct MelTaps {
  std::vector<float> taps;        // per oct: 8 filters x (4 * n4) weights
  std::vector<int32_t> meta;      // per filter: k0 | n4 << 10 | tap offset << 15
  bool usable = false;            // false: the generic kernel is used instead
};

// ortho DCT-II rows as MFMA A-operand images: dctA[(c * (n_mels/4) + i) * 64 + l] =
// D[16c + (l & 15)][4i + (l >> 4)], zero for coefficient rows >= n_mfcc.
struct DctBlocks {
  int32_t n_cgroups = 0;        // ceil(n_mfcc / 16)
  std::vector<float> A;
  std::vector<float> P;         // k_dct16 (n_mels % 16 == 0, n_mfcc <= 16): A images for 16-byte tile loads
};

struct HostTables {
  std::vector<float> window;      // n_fft, periodic (fftbins=True)
  std::vector<float> mel_dense;   // n_mels x (n_fft/2+1), librosa float32 values
  std::vector<float> dct;         // n_mfcc x n_mels, ortho DCT-II rows
  std::vector<float> tw;          // n_fft/2 complex: exp(-2*pi*i*n/(n_fft/2))
  std::vector<float> post;        // n_fft/2 complex: exp(-2*pi*i*k/n_fft)
  MelBlocks mel;
  MelTaps taps;
  DctBlocks dctb;
  HostF3Mel f3mel;
};

// returns AFX_OK or a negative status; msg set on failure
int validate_params(const afx_params& p, std::string& msg);
void build_host_tables(const afx_params& p, HostTables& t);

void set_error(const std::string& s);

// afx_host.cpp: the clip records of a ragged batch (host-only; what prepare_descriptors uploads)
struct ClipDesc;
struct BatchGeom {
  int64_t total_tpad = 0, total_tblk = 0;
  int nblocks = 0, max_tblocks = 1, max_tmax = 0;
};
bool build_clip_descs(int hop, int trim_hop, const int64_t* offsets, const int64_t* lengths, int n, ClipDesc* out,
                      BatchGeom& g, std::string& why);

}  // namespace afx
