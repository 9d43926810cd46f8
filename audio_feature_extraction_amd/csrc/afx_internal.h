// Internal declarations shared by the host table builder, the kernel launchers
// and the C-ABI layer of libafx.so.  Not installed; include/afx.h is the ABI.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "afx.h"

namespace afx {

// librosa.filters.mel rows stored sparsely: the non-zeros of filter m are the
// contiguous bins [k0[m], k0[m]+nnz); taps are padded with zeros to a multiple
// of 4 so that the four lane-quarters of a wave split a filter's taps evenly.
struct MelSparse {
  std::vector<int32_t> k0, ntap4, woff;   // per filter
  std::vector<float> taps;                // sum(ntap4)*4 floats
};

struct HostTables {
  std::vector<float> window;      // n_fft, periodic (fftbins=True)
  std::vector<float> mel_dense;   // n_mels x (n_fft/2+1), librosa float32 values
  std::vector<float> dct;         // n_mfcc x n_mels, ortho DCT-II rows
  std::vector<float> tw;          // n_fft/2 complex: exp(-2*pi*i*n/(n_fft/2))
  std::vector<float> post;        // n_fft/2 complex: exp(-2*pi*i*k/n_fft)
  MelSparse mel;
};

// returns AFX_OK or a negative status; msg set on failure
int validate_params(const afx_params& p, std::string& msg);
void build_host_tables(const afx_params& p, HostTables& t);

void set_error(const std::string& s);

}  // namespace afx
