// ba_profile.h -- measurement and test hooks of the host mirror's BundleAdjustment::adjustBundle.  NOT part of the
// reference's interface (include/BundleAdjustment.h:19-20 has the one static function, and so has the mirror's header): what
// bench.py's `adjust_bundle_call` and tests/test_gpu_host_cpp.py read through the self-test executable.
#pragma once
#include "sfmhip.h"

struct SfmBaCallProfile {
  int n_cam = 0, n_pt = 0, n_obs = 0;
  double pack_ms = 0;       // containers -> flat arrays (poses to angle-axis, the std::map tracks to observation triples)
  double solve_ms = 0;      // sfmhip_ba_solve, whose stages are in `solve`
  double writeback_ms = 0;  // flat arrays -> containers (only on CONVERGENCE)
  double total_ms = 0;
  sfmhip_ba_solve_profile solve = {};
};
// the last adjustBundle call of this thread
const SfmBaCallProfile& sfm_ba_last_call_profile();

// Test hook: the reference's two hard-coded limits (src/BundleAdjustment.cpp:118,120: 500 iterations, 10 s) made reachable on
// problems that converge in milliseconds; a negative value keeps the reference's.  The product never calls it (the mirror
// does not read the environment): csrc/host/ba_selftest.cpp does, for tests/test_gpu_host_cpp.py.
void sfm_ba_set_test_limits(int max_iterations, double max_time_s);
