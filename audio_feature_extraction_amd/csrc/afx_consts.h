// Constants shared by the host table builder and the device code.
#pragma once
namespace afx {
constexpr int kMelMaxItems = 8;      // mel work items per wave (the first 3 stay in registers)
constexpr int kMelRegItems = 3;
constexpr int kMelMaxSlots = 2;      // 1 KB LDS partial-sum slots for groups split across waves
}  // namespace afx
