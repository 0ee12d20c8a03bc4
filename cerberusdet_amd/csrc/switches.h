// Process-wide switches of the library: kernel-form pins for A/B timing and for the parity tests that compare two forms of one kernel.
// Read from the environment ONCE, when the library is loaded (core.hip); afterwards only cdet_set_switch() changes them -- no launch path calls
// getenv. cdet_active_switches() lists every switch that is not at its default, so a run's configuration is visible in the bench line.
#pragma once
#include <limits.h>

namespace cdet {

enum Switch {
    SW_CONV_PP = 0,    // CDET_CONV_PP     0: the 4-wave 3x3 form everywhere; 1 (default): the 8-wave ping-pong form for single-round launches; 2: wherever it fits
    SW_CONV_PAIR,      // CDET_CONV_PAIR   0: never the 320-cout pair tile; 1 (default): by geometry; 2: wherever it fits (tests)
    SW_HALO_NG,        // CDET_HALO_NG     1 | 2: pin the pixel tile (128 / 256); 3 | 4: the 384- / 512-pixel forms (CDET_EXPERIMENTS builds only)
    SW_HALO_WG3,       // CDET_HALO_WG3    0 / 1: never / always three workgroups per CU for the 96-cout patch tile (default: by grid size)
    SW_HALO_KS,        // CDET_HALO_KS     1: one K chain per half tile; 2: the in-workgroup K split wherever it fits (default: by grid size)
    SW_WGRAD_HALO,     // CDET_WGRAD_HALO  0: im2col weight-gradient kernels only; 3: the 1x1 tap-resident form for every Cout >= 128; 4: 64-cin tile
    SW_WGRAD_PATCH,    // CDET_WGRAD_PATCH 0: linear-halo weight gradient; 1: patch form for the 80-cout tile only (default: patch form where the map splits)
    SW_WGRAD_S2,       // CDET_WGRAD_S2    0: stride-2 weight gradients on the im2col kernel
    SW_PEER_SPIN_MS,   // CDET_PEER_SPIN_MS wall-clock budget of one in-kernel SyncBatchNorm exchange (default 600000)
    SW_COUNT
};
constexpr int SW_AUTO = INT_MIN;  // not pinned: the geometry rules decide

int sw(int id);                   // current value, or SW_AUTO
inline bool sw_is(int id) { return sw(id) != SW_AUTO; }

}  // namespace cdet
