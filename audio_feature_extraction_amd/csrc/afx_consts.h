// Constants shared by the host table builder and the device code.
#pragma once
namespace afx {
constexpr int kMelMaxItems = 8;      // mel work items per wave (the first 3 stay in registers)
constexpr int kMelRegItems = 3;
constexpr int kMelMaxSlots = 2;      // 1 KB LDS partial-sum slots for groups split across waves
constexpr int kPbPadRows = 3;        // zero rows past the Nyquist bin in the power-spectrum buffer (mel taps come in fours)
constexpr int kMelMaxQuads = 8;      // quads per wave in k_frames2 (n_mels <= 128)
constexpr int kMelTapCap = 1440;     // floats of LDS reserved for the padded tap table
}  // namespace afx
