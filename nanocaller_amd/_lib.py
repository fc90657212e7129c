"""ctypes binding of libnanocaller_hip.so (the C ABI declared in include/nanocaller_hip.h).

There is no CPU fallback: if the shared library is missing, or no MI355X is visible when a context is
created, this raises -- the product path never routes through oracle/.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NANOCALLER_HIP_LIB") or os.path.join(_HERE, "libnanocaller_hip.so")   # env: another build of the same ABI

NC_OK = 0
NC_ERR_CAPACITY = -2
NC_ERR_NOMEM = -3
NC_ERR_UNSUPPORTED = -7
FLAG_REFSKIP = 0x10000   # nc_decoded_arrays.flag bit: the CIGAR holds a reference skip
ABI_VERSION = 11         # include/nanocaller_hip.h NC_ABI_VERSION this binding was written for
MODEL_SNP, MODEL_SNP_HAP, MODEL_INDEL, MODEL_INDEL_HAP = 0, 1, 2, 3
SEQ_MODES = {"ont": 0, "short_ont": 1, "ul_ont": 2, "ul_ont_extreme": 3, "pacbio": 4}
CODE_ABSENT = 7
FLANK = 50000
SNP_TENSOR = 1025
# Scoring of the product's own star alignment (gap open, gap extend, match, mismatch): a gap must cost more than a mismatch
# or sequencing noise next to a real indel is absorbed as extra gaps and the consensus length comes out wrong; with (25, 1, 20,
# -10) 90 % of planted indels come back with their exact length against 77 % with parasail's call-site values (9, 1, 20, -10),
# which stay in use where the reference uses them: allele_prediction (tools/exp_concordance.py).
STAR_SCORING = (25, 1, 20, -10)

# every symbol include/nanocaller_hip.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "nc_abi_version", "nc_device_count", "nc_ctx_create", "nc_ctx_destroy", "nc_ctx_set_stream", "nc_ctx_sync",
    "nc_last_error", "nc_malloc", "nc_free", "nc_memcpy_h2d", "nc_memcpy_d2h", "nc_last_kernel_ms",
    "nc_enable_timing", "nc_pack_plan", "nc_pack_fill", "nc_snp_scan", "nc_snp_scan_begin", "nc_snp_scan_end", "nc_snp_scan_fetch", "nc_snp_featurize",
    "nc_snp_scale", "nc_load_weights", "nc_snp_forward", "nc_indel_forward", "nc_indel_tensor", "nc_indel_scan",
    "nc_bam_open", "nc_bam_close", "nc_bam_n_refs", "nc_bam_ref", "nc_bam_error", "nc_bam_decode", "nc_decoded_view",
    "nc_decoded_free", "nc_snp_vcf_format", "nc_set_cnn_precision", "nc_snp_scan_fetch_async", "nc_snp_forward_drain", "nc_argsort4", "nc_indel_slices", "nc_slices_view", "nc_slices_free",
    "nc_nw_cigar", "nc_allele_prediction", "nc_bgzf_compress", "nc_bam_set_threads", "nc_indel_scan_batch", "nc_timing_sums", "nc_snp_chunk_depth_async", "nc_bam_decode_regions", "nc_star_msa", "nc_star_msa_tensor", "nc_star_msa_tensor_dup", "nc_set_tensor_format", "nc_allele_prediction_batch", "nc_allele_prediction_device", "nc_bgzf_read_file", "nc_consensus_strings",
    "nc_wire_build", "nc_wire_build_del", "nc_wire_apply_deletions", "nc_wire_ref_unpack", "nc_wire_view", "nc_wire_free", "nc_wire_expand", "nc_wire_expand_del", "nc_snp_set_mates", "nc_decoded_name_groups", "nc_wire_ins_unpack", "nc_indel_events_pack8", "nc_indel_events_expand8", "nc_wire_build2", "nc_wire_expand2", "nc_d2h_async",
    "nc_indel_pass2_sets", "nc_pass2_view", "nc_pass2_free",
    "nc_decoded_check", "nc_indel_pack_build", "nc_indel_pack_view", "nc_indel_pack_free", "nc_indel_sites_plan", "nc_indel_sites_run",
    "nc_indel_sites_fetch", "nc_indel_sites_fetch_alt", "nc_indel_sites_stage_ms", "nc_indel_sites_band_stats", "nc_indel_sites_band", "nc_indel_events_pack", "nc_indel_events_expand", "nc_inflate_device", "nc_inflate_device_phase", "nc_bgzf_crc_device", "nc_bgzf_members", "nc_bgzf_scan", "nc_bam_walk", "nc_bam_meta", "nc_bam_codes", "nc_bam_indel_reads", "nc_indel_sites_scoring", "nc_indel_vcf_format", "nc_synth_indel_truth", "nc_synth_indel_reads", "nc_cnn_x_limit", "nc_cnn_range_watch", "nc_snp_trunk_info",
]


class TileEntry(C.Structure):
    _fields_ = [("start", C.c_int32), ("end", C.c_int32), ("base_flag", C.c_int64)]


TILE_ENTRY_DTYPE = np.dtype([("start", "<i4"), ("end", "<i4"), ("base_flag", "<i8")])
assert TILE_ENTRY_DTYPE.itemsize == C.sizeof(TileEntry) == 16


class ReadPackC(C.Structure):
    _fields_ = [("codes_len", C.c_int64), ("codes", C.c_void_p), ("tile_size", C.c_int32), ("tile_pos0", C.c_int32),
                ("n_tiles", C.c_int32), ("tile_off", C.c_void_p), ("tile_ent", C.c_void_p), ("n_entries", C.c_int64)]


class ScanParamsC(C.Structure):
    _fields_ = [("mincov", C.c_int32), ("min_allele_freq", C.c_double), ("nbr_t0", C.c_double), ("nbr_t1", C.c_double),
                ("haploid", C.c_int32)]


class IndelEventsC(C.Structure):
    _fields_ = [("n_reads", C.c_int32), ("ev_off", C.c_void_p), ("ev_pos", C.c_void_p), ("ev_len", C.c_void_p),
                ("read_hap", C.c_void_p)]


class IndelScanParamsC(C.Structure):
    _fields_ = [("mincov", C.c_int32), ("win_size", C.c_int32), ("small_win_size", C.c_int32), ("ins_t", C.c_double),
                ("del_t", C.c_double), ("haploid", C.c_int32), ("impute", C.c_int32)]


class DecodedArraysC(C.Structure):
    _fields_ = [("n_reads", C.c_int32), ("start", C.c_void_p), ("end", C.c_void_p), ("flag", C.c_void_p), ("off", C.c_void_p),
                ("codes", C.c_void_p), ("n_codes", C.c_int64), ("ev_off", C.c_void_p), ("ev_pos", C.c_void_p),
                ("ev_len", C.c_void_p), ("n_events", C.c_int64), ("hap", C.c_void_p), ("ps", C.c_void_p),
                ("seq_off", C.c_void_p), ("seq", C.c_void_p), ("n_seq", C.c_int64), ("name_off", C.c_void_p),
                ("names", C.c_void_p), ("qstart", C.c_void_p)]


class SlicesArraysC(C.Structure):
    _fields_ = [("n_anchor", C.c_int32), ("anchor_off", C.c_void_p), ("read_idx", C.c_void_p), ("seq_off", C.c_void_p),
                ("seq", C.c_void_p), ("n_slices", C.c_int64)]


class Pass2ArraysC(C.Structure):
    _fields_ = [("n_kept", C.c_int32), ("anchor_idx", C.c_void_p), ("first0", C.c_void_p), ("sets_per_anchor", C.c_int32),
                ("n_sets", C.c_int32), ("set_read0", C.c_void_p), ("n_alignments", C.c_int32), ("read_off", C.c_void_p),
                ("reads", C.c_void_p), ("ref_off", C.c_void_p), ("refs", C.c_void_p), ("max_cols", C.c_int32), ("al_dup", C.c_void_p)]


class IndelPackArraysC(C.Structure):
    _fields_ = [("n_reads", C.c_int32), ("ev_off", C.c_void_p), ("ev_pos", C.c_void_p), ("ev_len", C.c_void_p), ("n_events", C.c_int64),
                ("ins_off", C.c_void_p), ("ins_bases", C.c_void_p), ("n_ins_bases", C.c_int64), ("tail_off", C.c_void_p),
                ("tail_bases", C.c_void_p), ("n_tail_bases", C.c_int64), ("read_ps", C.c_void_p), ("read_hap", C.c_void_p),
                ("read_flag", C.c_void_p)]


class IndelReadsC(C.Structure):
    _fields_ = [("n_reads", C.c_int32), ("slot_off", C.c_void_p), ("rd_start", C.c_void_p), ("rd_end", C.c_void_p), ("ev_off", C.c_void_p),
                ("ev_pos", C.c_void_p), ("ev_len", C.c_void_p), ("ins_off", C.c_void_p), ("ins_bases", C.c_void_p), ("tail_off", C.c_void_p),
                ("tail_bases", C.c_void_p), ("read_ps", C.c_void_p), ("read_hap", C.c_void_p), ("read_flag", C.c_void_p)]


class WireArraysC(C.Structure):
    _fields_ = [("n_reads", C.c_int32), ("rd_start", C.c_void_p), ("rd_end", C.c_void_p), ("slot_off", C.c_void_p),
                ("codes_len", C.c_int64), ("n_blocks", C.c_int64), ("blk_off", C.c_void_p), ("blk_read", C.c_void_p), ("events", C.c_void_p),
                ("n_events", C.c_int64), ("blk_ev", C.c_void_p), ("ev_bytes", C.c_void_p), ("n_ev_bytes", C.c_int64)]


_lib = None


class NanoCallerHipError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NanoCallerHipError(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
        # torch first: it brings its own HIP runtime, and the library must bind to THAT one -- loaded the other way round the process holds
        # two runtimes and nc_ctx_create fails on the first stream (seen when a host-only entry point was the first call of a process)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
        L.nc_abi_version.restype = C.c_int
        if L.nc_abi_version() != ABI_VERSION:
            raise NanoCallerHipError("%s has ABI version %d, this binding needs %d: rebuild it" % (LIB_PATH, L.nc_abi_version(), ABI_VERSION))
        L.nc_device_count.argtypes = [C.POINTER(C.c_int)]
        L.nc_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
        L.nc_ctx_destroy.argtypes = [vp]
        L.nc_ctx_set_stream.argtypes = [vp, vp]
        L.nc_ctx_sync.argtypes = [vp]
        L.nc_last_error.argtypes = [vp]
        L.nc_last_error.restype = C.c_char_p
        L.nc_malloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
        L.nc_free.argtypes = [vp, vp]
        L.nc_memcpy_h2d.argtypes = [vp, vp, vp, C.c_size_t]
        L.nc_memcpy_d2h.argtypes = [vp, vp, vp, C.c_size_t]
        L.nc_last_kernel_ms.argtypes = [vp, C.c_int, C.POINTER(C.c_float)]
        L.nc_timing_sums.argtypes = [vp, vp, vp]
        L.nc_enable_timing.argtypes = [vp, C.c_int]
        L.nc_set_cnn_precision.argtypes = [vp, C.c_int]
        L.nc_pack_plan.argtypes = [i32, vp, vp, vp, i32, i32, i32, C.POINTER(i64), C.POINTER(i32), C.POINTER(i32),
                                   C.POINTER(i64)]
        L.nc_pack_fill.argtypes = [i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, i64, vp, vp, i64]
        L.nc_snp_scan.argtypes = [vp, C.POINTER(ReadPackC), vp, i32, i32, i32, i32, C.POINTER(ScanParamsC), i32, vp, vp,
                                  C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
        L.nc_snp_scan_begin.argtypes = [vp, C.POINTER(ReadPackC), vp, i32, i32, i32, i32, C.POINTER(ScanParamsC), i32, vp, vp]
        L.nc_snp_scan_end.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
        L.nc_snp_scan_fetch.argtypes = [vp, vp, vp, vp, vp, vp]
        L.nc_snp_scan_fetch_async.argtypes = [vp, vp, vp, vp, vp, vp, vp]
        L.nc_snp_featurize.argtypes = [vp, C.POINTER(ReadPackC), vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp]
        L.nc_snp_scale.argtypes = [vp, vp, vp, dbl, i32, vp, vp]
        L.nc_snp_chunk_depth_async.argtypes = [vp, vp, vp]
        L.nc_load_weights.argtypes = [vp, i32, vp, C.c_size_t]
        L.nc_snp_forward.argtypes = [vp, i32, i64, vp, vp, vp, i32, vp, vp]
        L.nc_snp_forward_drain.argtypes = [vp, i32, i64, vp, vp, vp, i32, vp, vp, vp, vp, vp]
        L.nc_indel_forward.argtypes = [vp, i32, i64, vp, vp]
        L.nc_indel_tensor.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, i32, vp, vp]
        L.nc_indel_scan.argtypes = [vp, C.POINTER(ReadPackC), C.POINTER(IndelEventsC), vp, i32, i32, C.POINTER(IndelScanParamsC), vp]
        L.nc_indel_scan_batch.argtypes = [vp, C.POINTER(ReadPackC), C.POINTER(IndelEventsC), vp, i32, vp, vp, C.POINTER(IndelScanParamsC), vp, vp]
        L.nc_bam_open.argtypes = [C.c_char_p, C.POINTER(vp)]
        L.nc_bam_close.argtypes = [vp]
        L.nc_bam_set_threads.argtypes = [vp, i32]
        L.nc_bam_n_refs.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
        L.nc_bam_ref.argtypes = [vp, i32, C.POINTER(C.c_char_p), C.POINTER(i32)]
        L.nc_bam_error.argtypes = [vp]
        L.nc_bam_error.restype = C.c_char_p
        L.nc_bam_decode.argtypes = [vp, i32, i32, i32, i32, C.POINTER(vp)]
        L.nc_star_msa_tensor.argtypes = [vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp]
        L.nc_star_msa_tensor_dup.argtypes = L.nc_star_msa_tensor.argtypes + [vp]
        L.nc_consensus_strings.argtypes = [vp, i32, i32, vp, vp, vp]
        L.nc_allele_prediction_batch.argtypes = [i32, C.c_char_p, vp, C.c_char_p, vp, vp, vp, vp]
        L.nc_allele_prediction_device.argtypes = [vp, i32, C.c_char_p, vp, C.c_char_p, vp, vp, vp, vp]
        L.nc_set_tensor_format.argtypes = [vp, C.c_int]
        L.nc_star_msa.argtypes = [i32, C.c_char_p, vp, C.c_char_p, i32, i32, i32, i32, i32, i32, vp, vp, C.POINTER(i32)]
        L.nc_bam_decode_regions.argtypes = [C.c_char_p, i32, i32, i32, i32, i32, C.POINTER(vp)]
        L.nc_decoded_view.argtypes = [vp, C.POINTER(DecodedArraysC)]
        L.nc_decoded_free.argtypes = [vp]
        L.nc_indel_slices.argtypes = [vp, i32, vp, i32, i32, vp, C.POINTER(vp)]
        L.nc_slices_view.argtypes = [vp, C.POINTER(SlicesArraysC)]
        L.nc_slices_free.argtypes = [vp]
        L.nc_nw_cigar.argtypes = [C.c_char_p, i32, C.c_char_p, i32, i32, i32, i32, i32, vp, vp, i32, C.POINTER(i32)]
        L.nc_allele_prediction.argtypes = [C.c_char_p, i32, C.c_char_p, i32, i32, C.POINTER(i32), C.POINTER(i32)]
        L.nc_bgzf_compress.argtypes = [vp, i64, i32, vp, i64, C.POINTER(i64), vp, i64, C.POINTER(i64)]
        L.nc_indel_pass2_sets.argtypes = [vp, vp, i32, vp, C.c_char_p, i64, i32, i32, i32, i32, i32, i32, vp, vp, vp, C.POINTER(vp)]
        L.nc_pass2_view.argtypes = [vp, C.POINTER(Pass2ArraysC)]
        L.nc_pass2_free.argtypes = [vp]
        L.nc_d2h_async.argtypes = [vp, vp, vp, vp, C.c_size_t]
        L.nc_wire_build.argtypes = [i32, vp, vp, vp, vp, vp, vp, i32, i64, C.POINTER(vp)]
        L.nc_wire_build_del.argtypes = [i32, vp, vp, vp, vp, vp, vp, i32, i64, vp, vp, vp, C.POINTER(vp)]
        L.nc_wire_build2.argtypes = [i32, vp, vp, vp, vp, vp, vp, i32, i64, vp, vp, vp, i32, C.POINTER(vp)]
        L.nc_wire_apply_deletions.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp]
        L.nc_wire_ref_unpack.argtypes = [vp, vp, i64, vp]
        L.nc_wire_view.argtypes = [vp, C.POINTER(WireArraysC)]
        L.nc_wire_free.argtypes = [vp]
        L.nc_wire_expand.argtypes = [vp, i32, vp, vp, vp, vp, i32, i64, vp, vp, vp, i64, vp, i64, vp]
        L.nc_wire_expand_del.argtypes = L.nc_wire_expand.argtypes + [vp, vp, vp, vp]
        L.nc_wire_expand2.argtypes = L.nc_wire_expand.argtypes + [vp, vp, vp, vp]
        L.nc_snp_set_mates.argtypes = [vp, i32, vp, vp]
        L.nc_decoded_name_groups.argtypes = [vp, vp, vp, C.POINTER(i64)]
        L.nc_wire_ins_unpack.argtypes = [vp, vp, i64, vp, i32, vp]
        L.nc_indel_events_pack8.argtypes = [i32, vp, vp, vp, vp, vp, vp, i64, vp, vp, i64, vp, vp, vp, C.POINTER(i64), C.POINTER(i64)]
        L.nc_indel_events_expand8.argtypes = [vp, i32, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp]
        L.nc_decoded_check.argtypes = [vp, vp, C.POINTER(i64), C.POINTER(i64)]
        L.nc_indel_pack_build.argtypes = [vp, vp, i32, C.POINTER(vp)]
        L.nc_indel_pack_view.argtypes = [vp, C.POINTER(IndelPackArraysC)]
        L.nc_indel_pack_free.argtypes = [vp]
        L.nc_indel_sites_plan.argtypes = [vp, C.POINTER(ReadPackC), vp, i32, i32, i64, C.POINTER(IndelReadsC), vp, i32, vp, vp,
                                          C.POINTER(IndelScanParamsC), i32, i32, C.POINTER(i32), C.POINTER(i64)]
        L.nc_indel_sites_run.argtypes = [vp, vp]
        L.nc_indel_sites_scoring.argtypes = [vp, i32, i32, i32, i32]
        L.nc_indel_sites_fetch.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.POINTER(i64)]
        L.nc_indel_sites_fetch_alt.argtypes = [vp, vp, i64]
        L.nc_indel_sites_stage_ms.argtypes = [vp, vp, vp]
        L.nc_indel_sites_band_stats.argtypes = [vp, vp]
        L.nc_indel_sites_band.argtypes = [vp, i32, i32]
        L.nc_indel_events_pack.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp, i64, vp, vp, vp, C.POINTER(i64)]
        L.nc_indel_events_expand.argtypes = [vp, i32, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp]
        L.nc_inflate_device.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.nc_inflate_device_phase.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.nc_bgzf_crc_device.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp]
        L.nc_bgzf_members.argtypes = [vp, i64, i64, vp, vp, vp, C.POINTER(i64)]
        L.nc_bgzf_scan.argtypes = [vp, i64, i64, i64, vp, vp, vp, C.POINTER(i64), C.POINTER(i64)]
        L.nc_bam_walk.argtypes = [vp, vp, i64, i32, vp, vp, vp, vp, vp]
        L.nc_bam_meta.argtypes = [vp, vp, i64, vp, vp, vp]
        L.nc_bam_codes.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp, vp]
        L.nc_bam_indel_reads.argtypes = [vp, vp, i32, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.nc_indel_vcf_format.argtypes = [C.c_char_p, i64, vp, vp, i32, vp, i32, vp, vp, vp, vp, C.c_char_p, i64, i32, vp, i64, C.POINTER(i64), vp]
        L.nc_synth_indel_truth.argtypes = [vp, i64, C.c_uint64, dbl, dbl, dbl, dbl, i32, vp, vp, vp]
        L.nc_synth_indel_reads.argtypes = [vp, i64, C.c_uint64, dbl, dbl, dbl, dbl, i32, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.nc_cnn_x_limit.argtypes = [vp, i32, C.POINTER(C.c_float)]
        L.nc_snp_trunk_info.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.nc_cnn_range_watch.argtypes = [vp, vp]
        L.nc_argsort4.argtypes = [vp, i64, vp, C.POINTER(i64), vp, i64]
        L.nc_snp_vcf_format.argtypes = [C.c_char_p, i64, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, i64, C.POINTER(i64)]
        for name in EXPORTS:
            fn = getattr(L, name)
            if name not in ("nc_last_error", "nc_bam_error"):
                fn.restype = C.c_int
        _lib = L
    return _lib


def npp(a):
    """host pointer of a contiguous numpy array (or None)"""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)
