// Host-only part of libafx.so: the error string, the version, and the builders that need no device -- the clip records of
// a batch (what prepare_descriptors uploads) and the pYIN tables.  Together with afx_tables.cpp, afx_f0_tables.cpp and
// afx_wav.cpp this is everything that parses caller- or file-supplied data on the host; `make asan` builds exactly these
// files (plus afx_host_stubs.cpp) with g++ -fsanitize=address,undefined as libafx_host_asan.so.
#include <algorithm>
#include <cstring>
#include <string>

#include "afx_device.h"
#include "afx_f0.h"
#include "afx_internal.h"

namespace afx {

static thread_local std::string g_err;
void set_error(const std::string& s) { g_err = s; }

// The per-clip records of a ragged batch (reference: batch_process walks files of any length,
// audio_feature_extraction_toolkit/core/feature_extractor.py:228-235): frame slots padded to whole 16-frame blocks,
// trim-block slots, block indices.  false + `why` on input the kernels' 32-bit frame arithmetic cannot hold.
bool build_clip_descs(int hop, int trim_hop, const int64_t* offsets, const int64_t* lengths, int n, ClipDesc* out,
                      BatchGeom& g, std::string& why) {
  g = BatchGeom{};
  if (hop <= 0 || trim_hop <= 0) { why = "hop / trim hop must be positive"; return false; }
  int64_t fb = 0, tb = 0, nblk = 0;
  int max_tb = 0, max_tm = 0;
  const int64_t kMaxElems = (int64_t)1 << 46;                 // offsets + lengths stay far from overflow
  for (int i = 0; i < n; ++i) {
    if (lengths[i] < 0 || offsets[i] < 0) { why = "negative clip offset/length"; return false; }
    if (offsets[i] > kMaxElems) { why = "clip offset too large"; return false; }
    if (lengths[i] / hop > (int64_t)1 << 30 || lengths[i] > kMaxElems) { why = "clip too long"; return false; }
    ClipDesc& c = out[i];
    c.off = offsets[i]; c.len = lengths[i];
    c.tmax = (int32_t)(1 + lengths[i] / hop);
    c.tpad = (c.tmax + kFramesPerBlock - 1) / kFramesPerBlock * kFramesPerBlock;
    c.frame_base = fb; c.tblk_base = tb;
    fb += c.tpad;
    const int64_t ntb = (lengths[i] + trim_hop - 1) / trim_hop;
    tb += ntb;
    max_tb = (int)std::max<int64_t>(max_tb, std::min<int64_t>(ntb, INT32_MAX));
    max_tm = std::max<int>(max_tm, c.tmax);
    c.blk_base = (int32_t)nblk; c.pad_ = 0;
    nblk += c.tpad / kFramesPerBlock;
    if (nblk > (int64_t)1 << 30 || tb > (int64_t)1 << 40) { why = "batch too large"; return false; }
  }
  g.total_tpad = fb; g.total_tblk = tb; g.max_tblocks = std::max(max_tb, 1); g.max_tmax = max_tm; g.nblocks = (int)nblk;
  return true;
}

}  // namespace afx

using namespace afx;

extern "C" int afx_version(void) { return AFX_VERSION; }

extern "C" const char* afx_last_error(void) { return g_err.c_str(); }

extern "C" int afx_batch_geometry(const afx_params* p, const int64_t* offsets, const int64_t* lengths, int n_clips,
                                  int64_t* records, int64_t* totals) {
  if (!p || n_clips < 0 || (n_clips > 0 && (!offsets || !lengths))) { set_error("afx_batch_geometry: null/invalid argument"); return AFX_ERR_INVALID; }
  std::string msg;
  int st = validate_params(*p, msg);
  if (st != AFX_OK) { set_error(msg); return st; }
  std::vector<ClipDesc> cd((size_t)std::max(n_clips, 1));
  BatchGeom g;
  if (!build_clip_descs(p->hop, p->trim_hop, offsets, lengths, n_clips, cd.data(), g, msg)) { set_error("afx_batch_geometry: " + msg); return AFX_ERR_INVALID; }
  if (records)
    for (int i = 0; i < n_clips; ++i) {
      int64_t* r = records + 4 * (size_t)i;
      r[0] = cd[i].frame_base; r[1] = cd[i].tmax; r[2] = cd[i].tpad; r[3] = cd[i].blk_base;
    }
  if (totals) { totals[0] = g.total_tpad; totals[1] = g.nblocks; totals[2] = g.total_tblk; totals[3] = g.max_tmax; }
  return AFX_OK;
}

extern "C" int afx_f0_build_tables(int sr, int n_fft, int hop, double fmin, double fmax, int32_t* info,
                                   double* beta, double* lt, double* freqs) {
  HostF0Tables t;
  std::string why;
  if (!build_f0_tables(sr, n_fft, hop, fmin, fmax, t, why)) { set_error("afx_f0_build_tables: " + why); return AFX_ERR_UNSUPPORTED; }
  if (info) {
    const int32_t v[8] = {t.p.min_period, t.p.max_period, t.p.n_bins, t.p.band, t.p.cap, t.p.n_lag, t.p.R, t.p.slots};
    std::memcpy(info, v, sizeof(v));
  }
  if (beta) std::memcpy(beta, t.beta.data(), t.beta.size() * sizeof(double));
  if (lt) std::memcpy(lt, t.lt.data(), t.lt.size() * sizeof(double));
  if (freqs) std::memcpy(freqs, t.freqs.data(), t.freqs.size() * sizeof(double));
  return AFX_OK;
}

