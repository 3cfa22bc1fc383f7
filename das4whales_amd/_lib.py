"""ctypes binding of libd4w.so -- the C-ABI boundary declared in include/d4w.h.

The HIP library is the product: there is no CPU or PyTorch fallback.  If the shared object is
missing (not built) the import fails loudly with the build command.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_PACKAGED = os.path.join(_HERE, "lib", "libd4w.so")
# D4W_LIB: another build of the same library (probe builds with extra instrumentation, scripts/probe/fp_timing.sh).  Never
# silently: a variable left over from a probe session would make every result come from that build.
LIB_PATH = os.environ.get("D4W_LIB") or _PACKAGED
if os.path.abspath(LIB_PATH) != os.path.abspath(_PACKAGED):
    import warnings
    warnings.warn("das4whales_amd: D4W_LIB=%s replaces the packaged library %s for this process" % (LIB_PATH, _PACKAGED),
                  RuntimeWarning, stacklevel=2)

D4W_OK = 0
_ERRORS = {-1: ValueError, -2: MemoryError, -3: RuntimeError}


class D4WLibraryMissing(ImportError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise D4WLibraryMissing(
            "das4whales_amd: native library %s not found. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    c_int, c_void_p, c_char_p = ctypes.c_int, ctypes.c_void_p, ctypes.c_char_p
    P = ctypes.POINTER
    sigs = {
        "d4w_last_error": (c_char_p, []),
        "d4w_version": (c_char_p, []),
        "d4w_fk_plan_create": (c_int, [c_int, c_int, P(c_void_p)]),
        "d4w_fk_plan_create_ex": (c_int, [c_int, c_int, P(c_int), P(c_void_p)]),
        "d4w_fk_plan_destroy": (c_int, [c_void_p]),
        "d4w_fk_shape_is_specialised": (c_int, [c_int, c_int]),
        "d4w_fk_register_shape": (c_int, [c_void_p, ctypes.c_size_t]),
        "d4w_fk_plan_info": (c_int, [c_void_p, P(c_int)]),
        "d4w_fk_set_mask_dense_f32": (c_int, [c_void_p, c_void_p, c_void_p]),
        "d4w_fk_set_mask_dense_pruned_f32": (c_int, [c_void_p, c_void_p, ctypes.c_double, c_void_p]),
        "d4w_fk_set_mask_design_f32": (c_int, [c_void_p, c_int, ctypes.c_double, ctypes.c_double, P(ctypes.c_double),
                                               c_int, c_int, c_void_p, ctypes.c_double, c_void_p]),
        "d4w_fk_plan_live_rows": (c_int, [c_void_p]),
        "d4w_fk_plan_order": (c_int, [c_void_p, P(c_int), P(ctypes.c_double)]),
        "d4w_fk_apply_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
        "d4w_fk_apply_timed_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, P(ctypes.c_float)]),
        "d4w_fkd_plan_create": (c_int, [c_int, c_int, c_int, c_int, P(c_void_p)]),
        "d4w_fkd_plan_destroy": (c_int, [c_void_p]),
        "d4w_fkd_plan_info": (c_int, [c_void_p, P(c_int)]),
        "d4w_fkd_plan_q1_owner": (c_int, [c_void_p, P(c_int)]),
        "d4w_fkd_plan_live_columns": (c_int, [c_void_p, P(c_int)]),
        "d4w_fkd_set_mask_dense_f32": (c_int, [c_void_p, c_void_p, c_void_p]),
        "d4w_fkd_set_mask_design_f32": (c_int, [c_void_p, c_int, ctypes.c_double, ctypes.c_double, P(ctypes.c_double),
                                                c_int, c_int, c_void_p, c_void_p]),
        "d4w_fkd_time_fwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
        "d4w_fkd_chan_apply_f32": (c_int, [c_void_p, c_void_p, c_void_p]),
        "d4w_fkd_time_inv_f32": (c_int, [c_void_p, c_void_p, c_void_p]),
        "d4w_fkd_plan_is_packed": (c_int, [c_void_p]),
        "d4w_fkd_time_fwd_packed_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
        "d4w_fkd_time_inv_packed_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
        "d4w_fkd_time_fwd_packed_rows_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
        "d4w_fkd_time_inv_packed_rows_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
        "d4w_fkd_time_inv_packed_rows_stats_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
        "d4w_fk_apply_stats_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
        "d4w_fk_apply_timed_stats_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                                 P(ctypes.c_float)]),
        "d4w_taper_f32": (c_int, [c_void_p, c_int, c_int, c_void_p]),
        "d4w_sosfiltfilt_ws_bytes": (ctypes.c_size_t, [c_int, c_int, c_int]),
        "d4w_sosfiltfilt_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, P(ctypes.c_double), P(ctypes.c_double),
                                        c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
        "d4w_sosfiltfilt_ends_ws_bytes": (ctypes.c_size_t, [c_int, c_int, c_int]),
        "d4w_sosfiltfilt_ends_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                             c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
        "d4w_sosfiltfilt_ends_sides_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, P(ctypes.c_double), P(ctypes.c_double), c_int, c_int,
                                             c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
        "d4w_design_mask_f32": (c_int, [c_int, c_int, c_int, ctypes.c_double, ctypes.c_double, P(ctypes.c_double),
                                        c_int, c_int, c_void_p, c_void_p, c_void_p]),
        "d4w_flip_sum_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
        "d4w_gaussian_filter_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, ctypes.c_double, c_void_p]),
        "d4w_minmax_normalise_f32": (c_int, [c_void_p, ctypes.c_size_t, c_void_p]),
        "d4w_fk_set_mask_dense_affine_f32": (c_int, [c_void_p, c_void_p, ctypes.c_float, ctypes.c_float, c_void_p]),
        "d4w_raw2strain_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, ctypes.c_double, c_void_p, c_void_p]),
        "d4w_row_stats_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
        "d4w_copy_cols_f32": (c_int, [c_void_p, ctypes.c_size_t, c_void_p, ctypes.c_size_t, c_int, c_int, c_void_p]),
        "d4w_xcorr_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                  c_void_p, c_void_p, c_void_p]),
        "d4w_xcorr_lens_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                       c_void_p, c_void_p, c_void_p]),
        "d4w_xcorr_dc_tail_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, ctypes.c_double, c_int, c_void_p,
                                          c_void_p]),
        "d4w_row_prefix_max_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
        "d4w_row_stats_prefix_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
        "d4w_fk_stats_in_epilogue": (c_int, [c_void_p]),
        "d4w_xcorr_dc_tail_rows_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, ctypes.c_double, c_int, c_void_p,
                                               c_void_p, c_void_p, ctypes.c_double, c_void_p]),
        "d4w_xcorr_mm_max_support": (c_int, []),
        "d4w_xcorr_mm_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                     c_int, c_int, c_void_p, c_void_p, c_void_p]),
        "d4w_xcorr_mm_rowmax_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                            c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
        "d4w_xcorr_mm_tail_max_support": (c_int, []),
        "d4w_xcorr_mm_tail_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                          c_int, c_int, ctypes.c_double, ctypes.c_double, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p]),
        "d4w_xcorr_fft_max_support": (c_int, []),
        "d4w_xcorr_fft_ws_bytes": (ctypes.c_size_t, []),
        "d4w_xcorr_fft_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                      c_void_p, c_void_p, c_void_p, c_void_p]),
        "d4w_xcorr_fft_cont_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                           c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
        "d4w_fir_fft_max_halfwidth": (c_int, []),
        "d4w_fir_fft_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, ctypes.c_double, c_void_p, c_void_p,
                                    c_void_p]),
        "d4w_fir_fft_cols_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, ctypes.c_double, c_void_p, c_int, c_int,
                                         c_void_p, c_void_p]),
        "d4w_fir_fft_halo_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int,
                                         c_void_p, ctypes.c_double, c_void_p, c_void_p, c_void_p]),
        "d4w_analytic_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, ctypes.c_double, c_void_p]),
        "d4w_analytic_row_fits_lds": (c_int, [c_int]),
        "d4w_analytic_long_clear": (c_int, []),
        "d4w_analytic_long_ws_bytes": (ctypes.c_size_t, [c_int, c_int]),
        "d4w_analytic_long_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, ctypes.c_double, c_void_p, c_void_p]),
        "d4w_row_var_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
        "d4w_snr_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
        "d4w_fx_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
        "d4w_stft_frames": (c_int, [c_int, c_int]),
        "d4w_stft_mag_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
        "d4w_stft_mm_eligible": (c_int, [c_int, c_int, c_int, c_int]),
        "d4w_stft_mag_mm_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
        "d4w_scale_rows_f32": (c_int, [c_void_p, c_int, ctypes.c_size_t, c_void_p, c_int, c_void_p]),
        "d4w_row_median_f32": (c_int, [c_void_p, c_int, ctypes.c_size_t, c_void_p, c_void_p]),
        "d4w_spectrocorr_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p,
                                        c_int, c_void_p, c_void_p]),
        "d4w_find_peaks_f32": (c_int, [c_void_p, c_int, c_int, ctypes.c_double, c_void_p, c_void_p, c_int, c_void_p]),
        "d4w_find_peaks_dthr_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, ctypes.c_double, c_void_p, c_void_p, c_int, c_void_p]),
        "d4w_pick_offsets_i64": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
        "d4w_pack_picks_i64": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, ctypes.c_int64, c_void_p, c_void_p]),
        "d4w_minmax_f32": (c_int, [c_void_p, ctypes.c_size_t, c_void_p, c_void_p]),
        "d4w_scale_pixels_f32": (c_int, [c_void_p, c_void_p, ctypes.c_size_t, c_void_p, ctypes.c_double, c_void_p]),
        "d4w_threshold_f32": (c_int, [c_void_p, c_void_p, ctypes.c_size_t, ctypes.c_double, c_void_p]),
        "d4w_mask_mul_f32": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_size_t, c_void_p]),
        "d4w_resize_ws_bytes": (ctypes.c_size_t, [c_int, c_int, c_int, c_int]),
        "d4w_resize_bilinear_aa_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p]),
        "d4w_filter2d_ws_bytes": (ctypes.c_size_t, [c_int, c_int]),
        "d4w_filter2d_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p]),
        "d4w_filter2d_mm_eligible": (c_int, [c_int, c_int]),
        "d4w_filter2d_mm_ws_bytes": (ctypes.c_size_t, [c_int, c_int]),
        "d4w_filter2d_mm_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)      # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    return lib, sigs


lib, SIGNATURES = _load()


def check(rc):
    if rc != D4W_OK:
        msg = lib.d4w_last_error().decode("utf-8", "replace")
        raise _ERRORS.get(rc, RuntimeError)("d4w: " + msg)


def version():
    return lib.d4w_version().decode()
