// Tuning / test knobs of the launchers (run lengths, which occupancy build of a kernel is launched, ...).  Defaults are
// the measured optima; a knob is read from its DDSP_HIP_<NAME> environment variable ONCE, when the library first needs
// it, and can be changed afterwards only through ddsp_hip_set_tuning() (include/ddsp_hip.h) -- measurement tools and the
// run-split tests use that; the launch path itself never touches the environment.
#pragma once

namespace ddsp {

enum Knob {
  KNOB_BLK_WPS = 0, KNOB_BLK_RUN, KNOB_BLK_PADLDS, KNOB_FFT_RUN, KNOB_STFT_WPS, KNOB_STFT_RUN, KNOB_MEL_WPS,
  KNOB_MEL_RUN, KNOB_FIR_MAX_SLOTS, KNOB_SINS_V1, KNOB_TAPS_GEMM, KNOB_STREAM_LAYOUT, KNOB_BLK_TURNS, KNOB_CZT_ROUNDS, KNOB_CZT_TURNS,
  KNOB_SINS_NOSKIP, KNOB_SMALL_PATH, KNOB_LANE_ROWS, KNOB_LANES, KNOB_FIR_BWD_DIRECT, KNOB_BWD_WPS, KNOB_TAPS_FULL,
  KNOB_AP_BWD_SPLIT, KNOB_SINS_SEQ,
  KNOB_COUNT
};

// a step with fewer frames than this (B F) takes the streaming-shape forms: fewer, fused launches (knob SMALL_PATH = 1: never)
constexpr long kSmallRows = 4096;

long knob(Knob k);                       // current value; 0 = unset (use the built-in default)
int knob_set(const char* name, long v);  // 0 on success, -1 for an unknown name
long knob_get(const char* name);         // value, or -1 for an unknown name

}  // namespace ddsp
