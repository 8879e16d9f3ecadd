"""ctypes binding of libsvdss_hip.so (the C-ABI declared in include/svdss_hip.h).

The library is the product: there is no Python or CPU fallback.  If the shared
object is missing this module raises at import time; if no GPU is present the
compute entry points return SVDSS_EHIP and the wrappers raise SvdssError.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SVDSS_LIB: another build of the same library (developer builds such as the op-counting one, `make count`)
LIB_PATH = os.environ.get("SVDSS_LIB") or os.path.join(_HERE, "libsvdss_hip.so")

SVDSS_OK = 0
SVDSS_SFS_ASSEMBLE = 1
SVDSS_BAM_PUTATIVE = 0x100


class SvdssError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        msg = lib.svdss_strerror(code).decode()
        hip = lib.svdss_last_hip_error().decode()
        super().__init__(f"{where}: {msg} (code {code})" + (f" [{hip}]" if hip else ""))


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C svdss_amd/csrc` (hipcc, gfx950). There is no CPU fallback."
        )
    return C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)


lib = _load()

_p = C.c_void_p
_i64 = C.c_int64
_i32 = C.c_int32
_pi64 = C.POINTER(C.c_int64)
_pi32 = C.POINTER(C.c_int32)
_pu8 = C.POINTER(C.c_uint8)

# every symbol declared in include/svdss_hip.h: (restype, argtypes)
SIGNATURES = {
    "svdss_strerror": (C.c_char_p, [C.c_int]),
    "svdss_last_hip_error": (C.c_char_p, []),
    "svdss_nt6_encode": (C.c_int, [C.c_char_p, _i64, _p]),
    "svdss_index_build": (C.c_int, [_p, _p, _i32, _i32, C.POINTER(_p)]),
    "svdss_index_build_device": (C.c_int, [_p, _p, _i32, _i32, _i32, C.POINTER(_p)]),
    "svdss_index_save": (C.c_int, [_p, C.c_char_p]),
    "svdss_index_save_records": (C.c_int, [_p, C.c_char_p]),
    "svdss_index_save_fmd": (C.c_int, [_p, C.c_char_p]),
    "svdss_fmd_read_bwt": (C.c_int, [C.c_char_p, _p, _i64, _p]),
    "svdss_index_load": (C.c_int, [C.c_char_p, C.POINTER(_p)]),
    "svdss_index_free": (None, [_p]),
    "svdss_index_size": (_i64, [_p]),
    "svdss_index_acc": (C.c_int, [_p, _p]),
    "svdss_index_bwt": (C.c_int, [_p, _p]),
    "svdss_index_device_bytes": (_i64, [_p]),
    "svdss_index_kmer": (_i32, [_p]),
    "svdss_index_kmer_limit": (None, [_i32]),
    "svdss_index_attach_blocks": (C.c_int, [_p, C.c_char_p]),
    "svdss_index_load_blocks": (C.c_int, [C.c_char_p, C.POINTER(_p)]),
    "svdss_index_append_blocks": (C.c_int, [_p, C.c_char_p]),
    "svdss_index_deep_frac": (C.c_double, [_p]),
    "svdss_index_to_device": (C.c_int, [_p, _i32]),
    "svdss_index_count": (_i64, [_p, _p, _i64]),
    "svdss_index_verify_device": (C.c_int, [_p, _i64, _p]),
    "svdss_device_count": (C.c_int, []),
    "svdss_search_stream_create": (C.c_int, [_i32, C.POINTER(_p)]),
    "svdss_stream_destroy": (C.c_int, [_p]),
    "svdss_index_replicate": (C.c_int, [_p, _i32, C.POINTER(_p)]),
    "svdss_sfs_search_batch": (C.c_int, [_p, _p, _p, _i64, _i32, C.POINTER(_p)]),
    "svdss_sfs_search_batch_bam": (C.c_int, [_p, _p, _p, _p, _i64, _i32, C.POINTER(_p)]),
    "svdss_host_alloc": (C.c_int, [_i64, C.POINTER(_p)]),
    "svdss_host_free": (None, [_p]),
    "svdss_bgzf_inflate": (C.c_int, [C.POINTER(_p), _i32, _p, _i64, _p, _i64, _p, _p, _i64, _pi64]),
    "svdss_inflate_kernel_ms": (C.c_double, [_p]),
    "svdss_inflate_free": (None, [_p]),
    "svdss_bgzf_deflate": (C.c_int, [C.POINTER(_p), _i32, _p, _i64, _i32, _p, _i64, _p]),
    "svdss_deflate_kernel_ms": (C.c_double, [_p]),
    "svdss_deflate_free": (None, [_p]),
    "svdss_device_alloc": (C.c_int, [_i32, _i64, C.POINTER(_p)]),
    "svdss_device_free": (None, [_i32, _p]),
    "svdss_device_memset": (C.c_int, [_i32, _p, C.c_int, _i64]),
    "svdss_device_download": (C.c_int, [_i32, _p, _p, _i64]),
    "svdss_sfs_search_batch_device": (C.c_int, [_p, _p, _p, _i64, _i64, _i32, _p, C.POINTER(_p)]),
    "svdss_sfs_batch_nreads": (_i64, [_p]),
    "svdss_sfs_batch_total": (_i64, [_p]),
    "svdss_sfs_batch_total_ext": (_i64, [_p]),
    "svdss_sfs_batch_kernel_ms": (C.c_double, [_p]),
    "svdss_sfs_batch_search_kernel_ms": (C.c_double, [_p]),
    "svdss_sfs_batch_segments": (_i32, [_p]),
    "svdss_sfs_batch_fallbacks": (_i64, [_p]),
    "svdss_sfs_batch_used_bs": (_i32, [_p]),
    "svdss_sfs_batch_fetch": (C.c_int, [_p, _p, _p, _p, _p]),
    "svdss_sfs_batch_device_ptrs": (C.c_int, [_p, C.POINTER(_p), C.POINTER(_p), C.POINTER(_p), C.POINTER(_p)]),
    "svdss_sfs_batch_free": (None, [_p]),
    "svdss_bam_stream_create": (C.c_int, [_i32, C.POINTER(_p)]),
    "svdss_bam_stream_free": (None, [_p]),
    "svdss_bam_stream_error": (C.c_char_p, [_p]),
    "svdss_bam_stream_rewalked": (_i64, [_p, _pi64]),
    "svdss_bam_stream_region": (C.c_int, [_p, _i32, _i32, _p, _i64]),
    "svdss_bam_stream_head": (_i64, [_p, C.POINTER(_p)]),
    "svdss_bam_stream_tail": (_i64, [_p, C.POINTER(_p)]),
    "svdss_bam_batch_run": (C.c_int, [_p, _i64, _i32, _i64, _p, _i32, _p, _p, _p, _p, _p, _i32, C.POINTER(_p)]),
    "svdss_bam_store_create": (C.c_int, [_i32, _i64, _i64, C.POINTER(_p)]),
    "svdss_bam_store_free": (None, [_p]),
    "svdss_bam_store_reset": (C.c_int, [_p]),
    "svdss_bam_store_batches": (_i64, [_p, _p, _pi64, _pi64]),
    "svdss_bam_select_store_run": (C.c_int, [_p, _i64, _i32, _i64, _p, _p, _i32, _p, _p, _p, _p, _p, C.POINTER(_p)]),
    "svdss_bam_store_select": (C.c_int, [_p, _i64, _p, C.POINTER(_p)]),
    "svdss_bam_batch_front": (C.c_int, [_p, _i64, _i32, _i64, _i32, _p, _i32, _p, _p, _p, _p, _p, _i32, C.POINTER(_p)]),
    "svdss_bam_batch_parked": (C.c_int, [_p, _pi64, _pi64, _pi64]),
    "svdss_bam_batch_search": (C.c_int, [_p, _p]),
    "svdss_bam_park_create": (C.c_int, [_i32, _i64, _i64, C.POINTER(_p)]),
    "svdss_bam_park_free": (None, [_p]),
    "svdss_bam_park_close": (C.c_int, [_p]),
    "svdss_bam_park_groups": (_i64, [_p]),
    "svdss_bam_park_group": (C.c_int, [_p, _i64, _pi64, _pi64, _pi64]),
    "svdss_bam_park_group_ready": (_i32, [_p, _i64]),
    "svdss_bam_park_search": (C.c_int, [_p, _i64, _p, _i32, C.POINTER(_p)]),
    "svdss_bam_batch_result": (C.c_int, [_p, _p]),
    "svdss_bam_smooth_create": (C.c_int, [_p, _p, C.c_int32, C.c_int32, _p]),
    "svdss_bam_smooth_free": (None, [_p]),
    "svdss_bam_stream_set_output_prefix": (C.c_int, [_p, _p, _i64]),
    "svdss_bam_smooth_measure": (C.c_int, [_p, _i64, C.c_int32, _i64, _p, C.c_int32, _p, _p, _p, _p, _p, _p]),
    "svdss_bam_smooth_run": (C.c_int, [_p, _i64, C.c_int32, _i64, _p, C.c_double, _p, _i64, C.c_int32, _p, _p, _p, _p, _p, _p]),
    "svdss_bam_batch_smoothed": (C.c_int, [_p, _p]),
    "svdss_bam_batch_error": (C.c_char_p, [_p]),
    "svdss_bam_filter_create": (C.c_int, [_i32, _i32, _i32, _p, _p, _i64, _p, _p, _p, _i64, C.POINTER(_p)]),
    "svdss_bam_filter_free": (None, [_p]),
    "svdss_bam_select_run": (C.c_int, [_p, _i64, _i32, _i64, _p, _i32, _p, _p, _p, _p, _p, C.POINTER(_p)]),
    "svdss_bam_batch_selection": (C.c_int, [_p, _p]),
    "svdss_bam_batch_free": (None, [_p]),
    "svdss_ref_upload": (C.c_int, [_p, _p, _i32, _i32, C.POINTER(_p)]),
    "svdss_ref_upload_parts": (C.c_int, [_p, _p, _i32, _i32, C.POINTER(_p)]),
    "svdss_ref_free": (None, [_p]),
    "svdss_place_sfs_batch": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _p, _p, _p]),
    "svdss_smooth_batch": (C.c_int, [_p] * 11 + [_i64] + [_p] * 7),
    "svdss_align_global_batch": (C.c_int, [_p, _p, _p, _p, _i64, _i32, _p, _i32, _i32, _i32, _i32, _i32,
                                           C.POINTER(_p)]),
    "svdss_aln_batch_npairs": (_i64, [_p]),
    "svdss_aln_batch_total_cigar": (_i64, [_p]),
    "svdss_aln_batch_cells": (_i64, [_p]),
    "svdss_aln_batch_kernel_ms": (C.c_double, [_p]),
    "svdss_aln_batch_fetch": (C.c_int, [_p, _p, _p, _p]),
    "svdss_aln_batch_free": (None, [_p]),
    "svdss_poa_consensus_batch": (C.c_int, [_p, _p, _p, _i64, _i32, C.POINTER(_p)]),
    "svdss_poa_batch_nclusters": (_i64, [_p]),
    "svdss_poa_batch_total": (_i64, [_p]),
    "svdss_poa_batch_cells": (_i64, [_p]),
    "svdss_poa_batch_kernel_ms": (C.c_double, [_p]),
    "svdss_poa_batch_hbm": (_i64, [_p]),
    "svdss_poa_batch_quad_back": (_i64, [_p]),
    "svdss_poa_quad_selftest": (C.c_int, [_p, _p, C.c_int32]),
    "svdss_poa_batch_fetch": (C.c_int, [_p, _p, _p]),
    "svdss_poa_batch_free": (None, [_p]),
    "svdss_indel_ratio_batch": (C.c_int, [_p, _p, _p, _p, _i64, _i32, _p, _p]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here == symbol missing from the .so
    _fn.restype = _res
    _fn.argtypes = _args


def check(code, where):
    if code != SVDSS_OK:
        raise SvdssError(code, where)
