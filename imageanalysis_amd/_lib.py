"""ctypes loader for libiamx.so (the C ABI declared in include/iamx.h).

The product path has no CPU fallback: if the HIP library is missing, or a kernel is
asked to run without a GPU, this raises.  torch is imported first so that the HIP
runtime already mapped by torch (its bundled libamdhip64, soname libamdhip64.so.7) is
the one libiamx.so binds to -- one runtime per process, device pointers interchangeable.
"""
import contextlib
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('IAMX_LIB') or os.path.join(_HERE, 'libiamx.so')   # IAMX_LIB: A/B builds
_lib = None

c_void_p = ctypes.c_void_p
c_int = ctypes.c_int
c_int64 = ctypes.c_int64
c_double = ctypes.c_double

# name -> (restype, argtypes); mirrors include/iamx.h one to one
SIGNATURES = {
    'iamx_version': (c_int, []),
    'iamx_last_error': (ctypes.c_char_p, []),
    'iamx_arch': (ctypes.c_char_p, []),
    'iamx_desc_padded_rows': (c_int64, [c_int64]),
    'iamx_desc_pack_u8': (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    'iamx_desc_pack_f32': (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    'iamx_desc_unpack_u8': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'iamx_knn2_wg_per_pair': (c_int, [c_int]),
    'iamx_knn2_l2_pairs': (c_int, [c_void_p] * 8 + [c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'iamx_knn2_l2_u8': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                c_void_p, c_void_p, c_void_p]),
    'iamx_match_metric': (c_int, [c_void_p, c_void_p, c_int, c_double, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p]),
    'iamx_match_compact': (c_int, [c_void_p, c_int] + [c_void_p] * 4 + [c_int] + [c_void_p] * 4),
    'iamx_desc2_rows_cap': (c_int64, [c_int64]),
    'iamx_desc2_pack_u8': (c_int, [c_void_p, c_int64] + [c_void_p] * 7),
    'iamx_desc2_pack_f32': (c_int, [c_void_p, c_int64] + [c_void_p] * 7),
    'iamx_desc2_pack_batch_u8': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int]
                                 + [c_void_p] * 7),
    'iamx_knn2v2_pairs': (c_int, [c_void_p] * 11 + [c_int, c_int, c_int, c_int] + [c_void_p] * 3),
    'iamx_knn2v2_resolve': (c_int, [c_void_p] * 13 + [c_int, c_void_p, c_void_p]),
    'iamx_knn2v2_finish': (c_int, [c_void_p] * 10 + [c_double] + [c_void_p] * 5 + [c_int]
                           + [c_void_p] * 3),
    'iamx_desc3_rows_cap': (c_int64, [c_int64]),
    'iamx_knn2sym_rows_per_wg': (c_int, [c_int]),
    'iamx_knn2sym_kernel_id': (ctypes.c_char_p, [c_int]),
    'iamx_desc3_pack_u8': (c_int, [c_void_p, c_int64] + [c_void_p] * 7),
    'iamx_desc3_pack_f32': (c_int, [c_void_p, c_int64] + [c_void_p] * 7),
    'iamx_desc3_pack_batch_u8': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int]
                                 + [c_void_p] * 7),
    'iamx_knn2sym_narrow_bytes': (c_int64, [c_int64, c_int]),
    'iamx_knn2sym_sweep': (c_int, [c_void_p] * 9 + [c_int, c_int, c_int] + [c_void_p] * 4),
    'iamx_knn2sym_candidates': (c_int, [c_void_p] * 12 + [c_int, c_double] + [c_void_p] * 8
                                + [c_int64, c_int, c_int, c_void_p]),
    'iamx_knn2sym_exact': (c_int, [c_void_p] * 4 + [c_int64] + [c_void_p] * 8 + [c_int, c_double]
                           + [c_void_p] * 13 + [c_int64, c_int, c_void_p]),
    'iamx_match_postfilter_clip': (c_int, []),
    'iamx_match_pack_results': (c_int, [c_void_p] * 4 + [c_int, c_int, c_int64] + [c_void_p] * 4),
    'iamx_match_postfilter': (c_int, [c_void_p] * 9 + [c_int, c_double, c_double, c_double, c_double]
                              + [c_void_p] * 6),
    'iamx_thp_pays': (c_int, []),
    'iamx_link_matches': (c_int64, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                    c_void_p, c_void_p]),
    'iamx_chain_members_uv': (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int]),
    'iamx_kp_key2': (c_int, [c_void_p, c_int64, c_void_p]),
    'iamx_kp_dup_remap': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int]),
    'iamx_match_lists_scan': (c_int, [c_void_p] * 4 + [c_int64, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int]),
    'iamx_link_pair_blocks': (c_int64, [c_void_p] * 3 + [c_int64] + [c_void_p] * 4),
    'iamx_chains_longest_first': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int]),
    'iamx_first_occurrence': (c_int, [c_void_p, c_int64, c_void_p]),
    'iamx_ledger_index': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    'iamx_pairs_fwd_rev': (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int]),
    'iamx_touch_pages': (c_int, [c_void_p, c_int64, c_int]),
    'iamx_segment_mean_std': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int]),
    'iamx_hbm_copy16': (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    'iamx_yaw_feedback_new': (c_void_p, [c_int, c_void_p]),
    'iamx_yaw_feedback_free': (None, [c_void_p]),
    'iamx_yaw_feedback_seed': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'iamx_yaw_feedback_feed': (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'iamx_yaw_feedback_state': (c_int, [c_void_p, c_void_p, c_void_p]),
    'iamx_group_level': (c_int64, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_int,
                                   c_int, c_int, c_void_p]),
    'iamx_triangulate_ground': (c_int, [c_void_p] * 3 + [c_int] + [c_void_p] * 3 + [c_int64]
                                + [c_void_p] * 3),
    'iamx_triangulate_pairs': (c_int, [c_void_p] * 7 + [c_int, c_int, c_void_p, c_void_p]),
    'iamx_triangulate_pairs_xyz': (c_int, [c_void_p] * 7 + [c_int, c_int, c_void_p, c_void_p]),
    'iamx_triangulate_packed': (c_int, [c_void_p] * 7 + [c_int, c_int64, c_void_p, c_void_p]),
    'iamx_similarity_pairs': (c_int, [c_void_p] * 5 + [c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'iamx_exclusive_scan_i32': (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    'iamx_ba_residual': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                 c_int64, c_void_p, c_void_p, c_void_p]),
    'iamx_ba_residual_prepared': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p,
                                          c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    'iamx_ba_residual_jac': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p,
                                     c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p]),
    'iamx_image_prep_workspace_bytes': (c_int64, [c_int, c_int]),
    'iamx_image_resized_dims': (c_int, [c_int, c_int, c_double, c_void_p, c_void_p]),
    'iamx_image_equalize_resize': (c_int, [c_void_p, c_int, c_int, c_int, ctypes.c_float, c_double,
                                           c_void_p, c_int64, c_void_p, c_void_p]),
    'iamx_jpeg_info': (c_int, [c_void_p, c_int64, c_void_p]),
    'iamx_jpeg_decode_coefficients': (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p]),
    'iamx_jpeg_workspace_bytes': (c_int64, [c_void_p]),
    'iamx_jpeg_reconstruct': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    'iamx_gzip_f32_from_u8_bound': (c_int64, [c_int64, c_int64]),
    'iamx_gzip_f32_from_u8': (c_int64, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64]),
    'iamx_gzip_members_bound': (c_int64, [c_int64, c_int64]),
    'iamx_gzip_members': (c_int64, [c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_void_p, c_int64]),
    'iamx_gzip_records_bound': (c_int64, [c_int64, c_int64]),
    'iamx_gzip_records': (c_int64, [c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_void_p, c_int64]),
    'iamx_u8_to_f32': (c_int, [c_void_p, c_void_p, c_int64, c_int]),
    'iamx_f32_to_u8_many': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int]),
    'iamx_u8_gather_many': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int]),
    'iamx_feat_records': (c_int, [c_void_p] * 7 + [c_int64, c_void_p]),
    'iamx_pickle_pair_lists': (c_int64, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p]),
    'iamx_sift_workspace_bytes': (c_int64, [c_int, c_int]),
    'iamx_sift_detect': (c_int, [c_void_p, c_int, c_int, c_int, ctypes.c_float, ctypes.c_float,
                                 ctypes.c_float, c_void_p, c_int64, c_void_p, c_void_p, c_int,
                                 c_void_p, c_void_p]),
    'iamx_sift_pyramid_level': (c_int, [c_int] * 5 + [c_void_p] * 4),
    'iamx_sift_sort_workspace_bytes': (c_int64, [c_int]),
    'iamx_sift_sort': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                               c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    'iamx_ba_jv': (c_int, [c_void_p] * 5 + [c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'iamx_ba_jtv': (c_int, [c_void_p] * 6 + [c_int64, c_int, c_int, c_void_p, c_int, c_void_p,
                            c_void_p, c_void_p]),
    'iamx_ba_lsmr_state_size': (c_int, []),
    'iamx_ba_lsmr_partials_size': (c_int64, [c_int, c_int]),
    'iamx_ba_lsmr_prepare': (c_int, [c_void_p] * 3 + [c_int, c_int] + [c_void_p] * 3),
    'iamx_ba_lsmr_iterate': (c_int, [c_void_p] * 8 + [c_int64, c_int, c_int] + [c_void_p] * 12
                             + [c_int, c_void_p]),
    'iamx_ba_lsmr_phase': (c_int, [c_void_p] * 8 + [c_int64, c_int, c_int, c_int, c_int]
                           + [c_void_p] * 12 + [c_int, c_int, c_void_p]),
    'iamx_ba_accumulate': (c_int, [c_void_p] * 6 + [c_int64, c_int, c_int] + [c_void_p] * 5),
    'iamx_ba_block_diag': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'iamx_ba_schur_state_size': (c_int, []),
    'iamx_ba_schur_prepare': (c_int, [c_void_p] * 6 + [c_int64, c_int, c_int] + [c_void_p] * 10),
    'iamx_ba_schur_factor': (c_int, [c_void_p] * 3 + [c_int, c_int, c_int, c_double, c_double, c_int]
                             + [c_void_p] * 8),
    'iamx_ba_schur_iterate': (c_int, [c_void_p] * 8 + [c_int64, c_int, c_int] + [c_void_p] * 15
                              + [c_int, c_int, c_int, c_void_p]),
    'iamx_ba_schur_finish': (c_int, [c_void_p] * 6 + [c_int64, c_int, c_int, c_int, c_int]
                             + [c_void_p] * 8),
    'iamx_comm_unique_id': (c_int, [c_void_p]),
    'iamx_comm_init': (c_int, [c_int, c_int, c_void_p, c_void_p]),
    'iamx_comm_destroy': (c_int, [c_void_p]),
    'iamx_comm_allgather': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    'iamx_comm_allreduce_f64': (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    'iamx_vec_axpby': (c_int, [c_int64, c_double, c_void_p, c_double, c_void_p, c_void_p]),
    'iamx_vec_mul2': (c_int, [c_int64] + [c_void_p] * 6),
    'iamx_vec_dot': (c_int, [c_int64] + [c_void_p] * 5),
    'iamx_vec_lsmr_update': (c_int, [c_int64] + [c_void_p] * 4 + [c_double] * 3 + [c_void_p]),
    'iamx_vec_lincomb': (c_int, [c_int64, c_double, c_void_p, c_double, c_void_p, c_double, c_void_p,
                                 c_void_p, c_void_p]),
    'iamx_vec_mul': (c_int, [c_int64, c_double, c_void_p, c_void_p, c_void_p, c_void_p]),
    'iamx_vec_gather': (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    'iamx_vec_sqrt_shift': (c_int, [c_int64, c_void_p, c_double, c_void_p, c_void_p]),
    'iamx_vec_scratch_doubles': (c_int, []),
    'iamx_vec_dots': (c_int, [c_int64, c_int] + [c_void_p] * 6),
    'iamx_vec_absmax_prod': (c_int, [c_int64] + [c_void_p] * 5),
    'iamx_trf_cl_scaling': (c_int, [c_int64] + [c_void_p] * 7),
    'iamx_trf_scale': (c_int, [c_int64] + [c_void_p] * 9),
    'iamx_trf_jac_scale': (c_int, [c_int64, c_void_p, c_void_p, c_int, c_void_p]),
    'iamx_trf_step_to_bound': (c_int, [c_int64] + [c_void_p] * 7),
    'iamx_trf_reflect': (c_int, [c_int64] + [c_void_p] * 4 + [c_double] + [c_void_p] * 4),
    'iamx_trf_count_outside': (c_int, [c_int64] + [c_void_p] * 7),
    'iamx_trf_strictly_feasible': (c_int, [c_int64] + [c_void_p] * 6),
    'iamx_trf_active': (c_int, [c_int64] + [c_void_p] * 3 + [c_double] + [c_void_p] * 2),
    'iamx_trf_feasible_start': (c_int, [c_int64] + [c_void_p] * 3 + [c_double] + [c_void_p] * 2),
    'iamx_trf_scaled_start': (c_int, [c_int64] + [c_void_p] * 6),
}


class IamxError(RuntimeError):
    pass


def lib():
    """Load libiamx.so (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        import torch  # noqa: F401  (maps the HIP runtime first, see module docstring)
        if not os.path.exists(LIB_PATH):
            raise IamxError(
                "libiamx.so not found at %s -- build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` "
                "(imageanalysis_amd/csrc/build.sh); there is no CPU fallback" % LIB_PATH)
        h = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)          # AttributeError if the ABI and the header drift apart
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


def check(rc, what=''):
    if rc != 0:
        msg = lib().iamx_last_error()
        raise IamxError("%s failed (%d): %s" % (what or 'libiamx call', rc,
                                                msg.decode() if msg else '?'))


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise IamxError("no HIP device visible: the iamx hot path runs on an MI355X only "
                        "(no CPU fallback)")
    return torch.device('cuda', torch.cuda.current_device())


_held = threading.local()


def stream_ptr():
    held = getattr(_held, 'ptr', None)
    if held is not None:
        return held
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


@contextlib.contextmanager
def hold_stream(fresh=False):
    """Inside the block stream_ptr() answers with the stream that was current on entry (per
    thread) instead of asking torch each time -- torch.cuda.current_stream() costs ~8 us, more
    than many of the kernels the TRF loop launches.  Code that switches streams inside (a graph
    capture) wraps its launches in hold_stream(fresh=True): the stream current THERE."""
    import torch
    prev = getattr(_held, 'ptr', None)
    if fresh or prev is None:
        _held.ptr = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    try:
        yield
    finally:
        _held.ptr = prev
