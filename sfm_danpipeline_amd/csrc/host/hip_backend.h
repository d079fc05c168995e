// hip_backend.h -- process-wide sfmhip context shared by the host classes.
#pragma once
#include "sfmhip.h"

// Lazily created on device SFM_HIP_DEVICE (default 0); aborts with a message if no MI355X is
// present -- there is no CPU fallback behind the drop-in.
sfmhip_ctx* sfm_hip_context();
