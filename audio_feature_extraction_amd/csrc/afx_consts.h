// Constants shared by the host table builder and the device code.
#pragma once
namespace afx {
constexpr int kMelMaxItems = 8;      // mel work items per wave (the first 3 stay in registers)
constexpr int kMelRegItems = 3;
constexpr int kMelMaxSlots = 2;      // 1 KB LDS partial-sum slots for groups split across waves
constexpr int kPbPadRows = 3;        // zero rows past the Nyquist bin in the power-spectrum buffer (mel taps come in fours)
constexpr int kMelMaxOcts = 16;      // filter octs k_frames2 handles (n_mels <= 128), four per wave
}  // namespace afx
