"""ctypes binding of libsfmhip.so (include/sfmhip.h).  There is no fallback: if the library is
missing or no gfx950 device is present, the product path raises."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# (SFMHIP_SO: a diagnostic build of the same library, e.g. scripts/build_match_variants.py)
SO = os.environ.get("SFMHIP_SO") or os.path.join(HERE, "libsfmhip.so")

F32, U8 = 0, 1
L2, HAMMING = 0, 1
BA_CONVERGENCE, BA_NO_CONVERGENCE, BA_FAILURE = 0, 1, 2


class SfmHipError(RuntimeError):
    pass


class BaOpts(C.Structure):
    _fields_ = [
        ("max_iterations", C.c_int), ("max_time_s", C.c_double), ("function_tolerance", C.c_double),
        ("gradient_tolerance", C.c_double), ("parameter_tolerance", C.c_double), ("initial_radius", C.c_double),
        ("max_radius", C.c_double), ("min_radius", C.c_double), ("min_relative_decrease", C.c_double),
        ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double), ("jacobi_scaling", C.c_int),
        ("max_consecutive_invalid", C.c_int), ("verbose", C.c_int),
    ]


class BaSummary(C.Structure):
    _fields_ = [
        ("termination", C.c_int), ("iterations", C.c_int), ("successful_steps", C.c_int),
        ("initial_cost", C.c_double), ("final_cost", C.c_double), ("final_radius", C.c_double),
        ("gradient_max_norm", C.c_double), ("time_s", C.c_double), ("spin_timeouts", C.c_int),
    ]


class BaSolveProfile(C.Structure):
    """sfmhip_ba_solve_profile (include/sfmhip.h): where the last one-shot solve on a context spent its host time."""
    _fields_ = [("create_ms", C.c_double), ("set_params_ms", C.c_double), ("run_ms", C.c_double), ("get_params_ms", C.c_double),
                ("keep_ms", C.c_double), ("total_ms", C.c_double), ("plan_reused", C.c_int), ("front_plan_reused", C.c_int)]


class LmState(C.Structure):
    """sfmhip_lm_state (include/sfmhip.h): options + trust-region state of sfmhip_ba_lm_decide, the host-side test hook of the
    decision the device takes at the end of every step evaluation."""
    _fields_ = [
        ("gradient_tolerance", C.c_double), ("parameter_tolerance", C.c_double), ("function_tolerance", C.c_double),
        ("min_relative_decrease", C.c_double), ("max_radius", C.c_double), ("min_radius", C.c_double),
        ("max_consecutive_invalid", C.c_int), ("max_iterations", C.c_int), ("timing_only", C.c_int), ("pad", C.c_int),
        ("radius", C.c_double), ("decrease_factor", C.c_double), ("cost", C.c_double), ("gradient_max_norm", C.c_double),
        ("x_norm", C.c_double),
        ("iterations", C.c_int), ("successful_steps", C.c_int), ("invalid_steps", C.c_int), ("lin_unread", C.c_int),
        ("accepted", C.c_int), ("stop", C.c_int),
    ]


class LmInputs(C.Structure):
    _fields_ = [
        ("lin_cost", C.c_double), ("lin_failed_blocks", C.c_double), ("lin_gradient_max", C.c_double),
        ("candidate_cost", C.c_double), ("model_cost_change", C.c_double), ("step_norm2", C.c_double),
        ("candidate_norm2", C.c_double), ("solve_info", C.c_int), ("pad", C.c_int),
    ]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_void_p)

# every symbol include/sfmhip.h declares (tests check the built library exports all of them)
SYMBOLS = [
    "sfmhip_device_count",
    "sfmhip_init", "sfmhip_init_on_stream", "sfmhip_shutdown", "sfmhip_synchronize", "sfmhip_device", "sfmhip_stream", "sfmhip_set_timing", "sfmhip_error_string",
    "sfmhip_last_hip_error", "sfmhip_version", "sfmhip_match_knn2", "sfmhip_imageset_create",
    "sfmhip_imageset_upload", "sfmhip_imageset_adopt_device", "sfmhip_imageset_prepare_async",
    "sfmhip_imageset_destroy", "sfmhip_matchplan_create", "sfmhip_matchplan_set_pairs", "sfmhip_matchplan_run_async", "sfmhip_matchplan_fetch",
    "sfmhip_matchplan_fetch_knn", "sfmhip_matchplan_pipeline", "sfmhip_matchplan_fetch_wait", "sfmhip_matchplan_last_timing", "sfmhip_matchplan_last_knn_kernel_time",
    "sfmhip_matchplan_destroy",
    "sfmhip_triangulate", "sfmhip_find_2d3d", "sfmhip_merge_new_points", "sfmhip_ba_default_opts", "sfmhip_ba_solve", "sfmhip_ba_create",
    "sfmhip_ba_set_allreduce", "sfmhip_ba_set_params", "sfmhip_ba_get_params", "sfmhip_ba_run",
    "sfmhip_ba_iterate", "sfmhip_ba_reduced_system", "sfmhip_ba_linearize_obs", "sfmhip_ba_last_timing", "sfmhip_ba_reduced_layout", "sfmhip_ba_reduced_tree", "sfmhip_probe_i8_mfma_peak", "sfmhip_probe_clock_start", "sfmhip_probe_clock_read", "sfmhip_ba_reduced_step", "sfmhip_ba_lm_decide", "sfmhip_score_essential", "sfmhip_score_last_flags", "sfmhip_score_five_point", "sfmhip_score_homography_kernel", "sfmhip_score_homography", "sfmhip_sift_detect_and_compute", "sfmhip_sift_detect_and_compute_device", "sfmhip_sift_batch", "sfmhip_device_free", "sfmhip_host_free", "sfmhip_device_download", "sfmhip_ba_destroy", "sfmhip_ba_last_solve_profile", "sfmhip_host_parallel_for",
]

_lib = None


def lib():
    """Load libsfmhip.so and declare prototypes.  Raises if the extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO):
        raise SfmHipError(f"{SO} is missing: run `python -m sfm_danpipeline_amd.build` "
                          "(the HIP extension is the only implementation; there is no CPU fallback)")
    L = C.CDLL(SO)
    vp, i32, f64, cint = C.c_void_p, C.c_int32, C.c_double, C.c_int
    L.sfmhip_init.argtypes = [cint, C.POINTER(vp)]
    L.sfmhip_init_on_stream.argtypes = [cint, vp, C.POINTER(vp)]
    L.sfmhip_shutdown.argtypes = [vp]
    L.sfmhip_shutdown.restype = None
    L.sfmhip_synchronize.argtypes = [vp]
    L.sfmhip_set_timing.argtypes = [vp, C.c_int]
    L.sfmhip_error_string.argtypes = [cint]
    L.sfmhip_error_string.restype = C.c_char_p
    L.sfmhip_matchplan_set_pairs.argtypes = [vp, vp, cint]
    L.sfmhip_find_2d3d.argtypes = [vp, vp, vp, vp, cint, cint, cint, vp, vp, cint, vp, vp, vp]
    L.sfmhip_merge_new_points.argtypes = [vp, vp, cint, vp, cint, C.c_float, vp, vp]
    L.sfmhip_match_knn2.argtypes = [vp, vp, cint, vp, cint, cint, cint, cint, C.c_float, vp, vp, vp, vp]
    L.sfmhip_imageset_create.argtypes = [vp, cint, vp, cint, cint, cint, C.POINTER(vp)]
    L.sfmhip_imageset_upload.argtypes = [vp, cint, vp]
    L.sfmhip_imageset_adopt_device.argtypes = [vp, cint, vp]
    L.sfmhip_imageset_prepare_async.argtypes = [vp]
    L.sfmhip_imageset_destroy.argtypes = [vp]
    L.sfmhip_imageset_destroy.restype = None
    L.sfmhip_matchplan_create.argtypes = [vp, vp, cint, C.POINTER(vp)]
    L.sfmhip_matchplan_run_async.argtypes = [vp, C.c_float]
    L.sfmhip_matchplan_fetch.argtypes = [vp, vp, vp, vp, vp, C.c_int64, C.POINTER(C.c_int64)]
    L.sfmhip_matchplan_fetch_knn.argtypes = [vp, cint, vp, vp]
    L.sfmhip_matchplan_pipeline.argtypes = [vp, C.c_int64]
    L.sfmhip_matchplan_fetch_wait.argtypes = [vp, cint, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_int64)]
    L.sfmhip_matchplan_last_timing.argtypes = [vp, vp]
    L.sfmhip_matchplan_last_knn_kernel_time.argtypes = [vp, C.POINTER(C.c_double)]
    L.sfmhip_matchplan_destroy.argtypes = [vp]
    L.sfmhip_matchplan_destroy.restype = None
    if hasattr(L, "sfmhip_triangulate"):
        L.sfmhip_triangulate.argtypes = [vp, vp, vp, vp, vp, vp, vp, cint, C.c_float, vp, vp, vp]
    if hasattr(L, "sfmhip_ba_solve"):
        L.sfmhip_ba_default_opts.argtypes = [C.POINTER(BaOpts)]
        L.sfmhip_ba_default_opts.restype = None
        L.sfmhip_ba_solve.argtypes = [vp, cint, cint, cint, vp, vp, vp, vp, vp, vp, C.POINTER(BaOpts),
                                      C.POINTER(BaSummary)]
        L.sfmhip_ba_create.argtypes = [vp, cint, cint, cint, vp, vp, vp, C.POINTER(vp)]
        L.sfmhip_ba_set_allreduce.argtypes = [vp, ALLREDUCE_FN, vp, cint, cint]
        L.sfmhip_ba_set_params.argtypes = [vp, vp, vp, f64]
        L.sfmhip_ba_get_params.argtypes = [vp, vp, vp, vp]
        L.sfmhip_ba_run.argtypes = [vp, C.POINTER(BaOpts), C.POINTER(BaSummary)]
        L.sfmhip_ba_iterate.argtypes = [vp, cint, C.POINTER(BaSummary)]
        L.sfmhip_ba_reduced_system.argtypes = [vp, f64, vp, vp, vp]
        L.sfmhip_ba_linearize_obs.argtypes = [vp, cint, vp, vp, f64, vp, vp, vp, vp, vp]
        L.sfmhip_ba_last_timing.argtypes = [vp, vp, vp]
        L.sfmhip_ba_reduced_layout.argtypes = [vp, vp]
        L.sfmhip_ba_reduced_tree.argtypes = [vp, vp]
        L.sfmhip_probe_i8_mfma_peak.argtypes = [vp, C.c_double, vp, vp]
        L.sfmhip_probe_clock_start.argtypes = [vp, C.c_double]
        L.sfmhip_probe_clock_read.argtypes = [vp, vp]
        L.sfmhip_score_essential.argtypes = [vp, cint, vp, vp, vp, f64, f64, f64, f64, f64, f64, vp, vp, vp]
        L.sfmhip_score_last_flags.argtypes = [vp]
        L.sfmhip_score_five_point.argtypes = [vp, cint, vp, vp, vp, vp]
        L.sfmhip_score_homography_kernel.argtypes = [vp, cint, vp, vp, vp, vp]
        L.sfmhip_score_homography.argtypes = [vp, cint, vp, vp, vp, vp, f64, cint, vp, vp, vp]
        L.sfmhip_sift_detect_and_compute.argtypes = [vp, vp, cint, cint, cint, f64, f64, f64, cint, vp, vp, vp]
        L.sfmhip_sift_detect_and_compute_device.argtypes = [vp, vp, cint, cint, cint, f64, f64, f64, cint, vp, C.POINTER(vp), vp]
        L.sfmhip_sift_batch.argtypes = [vp, cint, vp, vp, vp, cint, f64, f64, f64, vp, vp, vp]
        L.sfmhip_device_free.argtypes = [vp]
        L.sfmhip_device_free.restype = None
        L.sfmhip_host_free.argtypes = [vp]
        L.sfmhip_host_free.restype = None
        L.sfmhip_device_download.argtypes = [vp, vp, vp, C.c_size_t]
        L.sfmhip_ba_reduced_step.argtypes = [vp, f64, vp, vp]
        L.sfmhip_ba_destroy.argtypes = [vp]
        L.sfmhip_ba_destroy.restype = None
        if hasattr(L, "sfmhip_ba_last_solve_profile"):  # (diagnostic builds of older revisions lack it)
            L.sfmhip_ba_last_solve_profile.argtypes = [vp, C.POINTER(BaSolveProfile)]
    _lib = L
    return L


def check(rc, what="sfmhip"):
    if rc != 0:
        msg = lib().sfmhip_error_string(rc).decode()
        raise SfmHipError(f"{what}: {msg} (status {rc}, hip error {lib().sfmhip_last_hip_error()})")


class Context:
    """One sfmhip context = one HIP device (+ optionally an existing stream, e.g. torch's)."""

    def __init__(self, device=0, stream=None):
        self.h = C.c_void_p()
        if stream is None:
            check(lib().sfmhip_init(device, C.byref(self.h)), "sfmhip_init")
        else:
            check(lib().sfmhip_init_on_stream(device, C.c_void_p(stream), C.byref(self.h)), "sfmhip_init_on_stream")
        self.device = device

    def synchronize(self):
        check(lib().sfmhip_synchronize(self.h), "sfmhip_synchronize")

    def set_timing(self, enable):
        """Record hipEvents between the stages of a run (MatchPlan.last_timing / BaProblem.last_timing);
        off by default: each event costs a few microseconds of stream bubble."""
        check(lib().sfmhip_set_timing(self.h, int(bool(enable))), "sfmhip_set_timing")

    def probe_i8_mfma_peak(self, seconds=0.05):
        """(operations/s, shader GHz) the chip sustains for bare i8 MFMAs on random operands (sfmhip_probe_i8_mfma_peak):
        the measured ceiling bench.py prints next to the nominal one."""
        ops, ghz = C.c_double(0.0), C.c_double(0.0)
        check(lib().sfmhip_probe_i8_mfma_peak(self.h, float(seconds), C.addressof(ops), C.addressof(ghz)), "sfmhip_probe_i8_mfma_peak")
        return ops.value, ghz.value

    def probe_clock_start(self, seconds):
        """A one-wave sampler on THIS context's stream for `seconds`; run the load to be qualified on another stream."""
        check(lib().sfmhip_probe_clock_start(self.h, float(seconds)), "sfmhip_probe_clock_start")

    def probe_clock_read(self):
        ghz = C.c_double(0.0)
        check(lib().sfmhip_probe_clock_read(self.h, C.addressof(ghz)), "sfmhip_probe_clock_read")
        return ghz.value

    def close(self):
        if self.h:
            lib().sfmhip_shutdown(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(int(os.environ.get("LOCAL_RANK", "0")))
    return _default_ctx
