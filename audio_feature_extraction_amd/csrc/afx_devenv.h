// Developer switches of libafx.so, read ONCE per process when the first caller asks (never on the per-call path:
// getenv is not safe against a host program that writes os.environ from other threads, and a batch must not pay for it).
// They exist for same-box A/B runs (tools/ab_env.sh) and for tests that exercise the multi-chunk paths with small
// batches; none of them is part of the ABI.  Internal to libafx.so.
#pragma once
#include <cstdint>

namespace afx {

struct DevEnv {
  int debug_skip = 0, f0_debug = 0;
  bool stamps = false, f3_debug = false, no_spec = false, no_tickets = false;
  bool no_frames3 = false, no_frames3s = false, no_frames3d = false;   // fall back to round 1's kernels (A/B)
  bool f3_generic_mel = false;        // the generic mel walk instead of the compiled-in schedule (A/B)
  bool generic_1024 = false;          // k_frames<1024> instead of k_frames2 on the two-pass path (A/B)
  bool no_dct16l = false;             // k_dct16<2|3> instead of k_dct16l (A/B)
  bool host_blocks = false;           // build the speculative block list on the host and upload it (A/B of k_build_blocks3)
  bool no_fused_tail = false;         // k_dct16* + k_stats instead of the fused per-clip DCT + statistics kernel (A/B)
  int tail_mode = 0;                  // k_tail: 1 registers, 2 / 3 / 4 LDS-DMA ring of 4 waves x 4 tiles / 8 x 2 / 4 x 2 with two workgroups per CU (A/B)
  int f3_waves = 0;                   // 12 / 16: force the workgroup size of the wave-level frame kernels
  int chunk_clips = 32768;
  int64_t f0_chunk_frames = 1280 * 1024;
  const char* f0_dump = nullptr;
  DevEnv();
};
const DevEnv& dev_env();             // afx_api.cpp

}  // namespace afx
