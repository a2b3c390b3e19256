"""ctypes binding of libsbgpu.so (include/sbgpu.h).

There is no fallback: if the shared library is missing or a call fails, an
exception is raised.  Nothing here imports the oracle.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SBGPU_LIB") or os.path.join(HERE, "lib", "libsbgpu.so")  # SBGPU_LIB: A/B builds

SBGPU_OK = 0
SBGPU_EINVAL = -1
SBGPU_EUNSUPPORTED = -6
SBGPU_ERCCL = -7
EM_UNSOLVED = -1
EM_OK, EM_INIT_EMPTY, EM_DENOM_ZERO, EM_MAXITER = 0, 1, 2, 3
STATUS_NAMES = {0: "OK", 1: "INIT_EMPTY", 2: "DENOM_ZERO", 3: "MAXITER"}

# every symbol include/sbgpu.h declares (tests check the .so exports all of them)
SYMBOLS = [
    "sbgpu_version", "sbgpu_build_id", "sbgpu_last_error", "sbgpu_device_count", "sbgpu_init", "sbgpu_finalize",
    "sbgpu_device_info", "sbgpu_synchronize", "sbgpu_plan_create", "sbgpu_plan_destroy", "sbgpu_plan_info",
    "sbgpu_plan_classes", "sbgpu_plan_locus_kinds", "sbgpu_em_run_device", "sbgpu_em_run_device_f32", "sbgpu_em_run_device_bias", "sbgpu_em_run_device_bias_f32", "sbgpu_em_last_kernel_ms",
    "sbgpu_set_timing", "sbgpu_em_last_phase_ms", "sbgpu_last_stage_ms", "sbgpu_pair_mates_host", "sbgpu_pair_mates_device", "sbgpu_matepairs_destroy", "sbgpu_matepairs_info",
    "sbgpu_matepairs_pairs", "sbgpu_matepairs_export", "sbgpu_assign_reads_host", "sbgpu_assign_reads_device",
    "sbgpu_bam_index_host", "sbgpu_bam_decode_host", "sbgpu_bam_decode_device", "sbgpu_bamreads_destroy", "sbgpu_bamreads_info",
    "sbgpu_bamreads_reads", "sbgpu_bamreads_export",
    "sbgpu_comm_unique_id", "sbgpu_comm_init", "sbgpu_comm_info", "sbgpu_comm_rccl_ranks", "sbgpu_comm_destroy",
    "sbgpu_allreduce_sum_f64", "sbgpu_allreduce_sum_i64", "sbgpu_allreduce_sum_f64_host", "sbgpu_allreduce_sum_i64_host",
    "sbgpu_insert_pdf_table", "sbgpu_binweight_device", "sbgpu_binweight_host",
    "sbgpu_exonbin_device", "sbgpu_exonbin_host", "sbgpu_segments_host", "sbgpu_hit_features", "sbgpu_frag_lens_host",
    "sbgpu_bins_create", "sbgpu_bins_create_device", "sbgpu_bins_destroy", "sbgpu_quantify_host", "sbgpu_quantify_device",
    "sbgpu_annotation_pin", "sbgpu_annotation_unpin", "sbgpu_annotation_unpin_matching", "sbgpu_release_idle_memory",
    "sbgpu_bins_export_weights", "sbgpu_collapse_pairs_host", "sbgpu_uniq_destroy", "sbgpu_uniq_info", "sbgpu_uniq_export",
    "sbgpu_collapse_pairs_device", "sbgpu_uniq_dev_destroy", "sbgpu_uniq_dev_info", "sbgpu_uniq_dev_hits", "sbgpu_uniq_dev_export", "sbgpu_bins_info", "sbgpu_bins_grouping", "sbgpu_bins_export",
    "sbgpu_format_value", "sbgpu_format_gtf_transcript", "sbgpu_format_context_row", "sbgpu_format_context_row_seq",
    "sbgpu_binseq_device", "sbgpu_binseq_host", "sbgpu_em_batch", "sbgpu_abundance_device", "sbgpu_tpm_device",
    "sbgpu_quantify_resident", "sbgpu_allreduce_max_i64", "sbgpu_allreduce_max_i64_host", "sbgpu_comm_init_host",
    "sbgpu_front_stream_begin", "sbgpu_front_stream_push", "sbgpu_front_stream_end", "sbgpu_front_stream_info", "sbgpu_front_stream_hits",
    "sbgpu_front_stream_destroy", "sbgpu_em_run_device_split",
]


class SbgpuError(RuntimeError):
    pass


class sbgpu_batch_t(C.Structure):
    _fields_ = [
        ("n_loci", C.c_int64),
        ("row_off", C.c_void_p),
        ("iso_off", C.c_void_p),
        ("f_off", C.c_void_p),
        ("count", C.c_void_p),
        ("F", C.c_void_p),
    ]


class sbgpu_abundance_params_t(C.Structure):
    _fields_ = [
        ("total_mapped_reads", C.c_int32),
        ("effective_len_norm", C.c_int32),
        ("filter_by_expression", C.c_int32),
        ("reserved", C.c_int32),
        ("insert_mean", C.c_double),
        ("min_isoform_frac", C.c_double),
    ]


class sbgpu_insert_t(C.Structure):
    _fields_ = [
        ("mean", C.c_double),
        ("sd", C.c_double),
        ("use_emp", C.c_int32),
        ("start_offset", C.c_int32),
        ("end_offset", C.c_int32),
        ("total_reads", C.c_int32),
        ("emp_hist", C.POINTER(C.c_double)),
        ("read_len", C.c_int32),
        ("long_read", C.c_int32),
    ]


class sbgpu_abundances_t(C.Structure):
    _fields_ = ([(n, C.c_void_p) for n in ("theta", "fpkm", "frac", "tpm", "keep", "status", "iters",
                                           "d_theta", "d_fpkm", "d_frac", "d_tpm", "d_keep", "d_status", "d_iters")] +
                [("total_mapped_reads", C.c_int64), ("total_fpkm", C.c_double), ("n_frag_lens", C.c_int64)])


# the caller's all-reduce of a host buffer (sbgpu_comm_init_host): fn(user, buf, n, is_f64, op) -> 0
HOST_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32)


class sbgpu_annotation_t(C.Structure):
    _fields_ = [
        ("n_loci", C.c_int64),
        ("iso_off", C.c_void_p),
        ("exon_off", C.c_void_p),
        ("exon_left", C.c_void_p),
        ("exon_right", C.c_void_p),
        ("seg_off", C.c_void_p),
        ("seg_left", C.c_void_p),
        ("seg_right", C.c_void_p),
    ]


class sbgpu_pairs_t(C.Structure):
    _fields_ = [(n, C.c_int64 if n == "n_pairs" else C.c_void_p) for n in (
        "n_pairs", "pair_locus", "pair_mass", "left_off", "left_code", "left_left", "left_right", "right_off", "right_code",
        "right_left", "right_right")]


class sbgpu_clusters_t(C.Structure):
    _fields_ = [("n_clusters", C.c_int64), ("ref", C.c_void_p), ("left", C.c_void_p), ("right", C.c_void_p), ("strand", C.c_void_p)]


class sbgpu_reads_t(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("read_id", C.c_void_p), ("block_off", C.c_void_p), ("block_left", C.c_void_p),
                ("block_right", C.c_void_p), ("partner_pos", C.c_void_p), ("flags", C.c_void_p), ("nh", C.c_void_p)]


class sbgpu_bam_opts_t(C.Structure):
    _fields_ = [("min_intron", C.c_int32), ("max_intron", C.c_int32), ("unique_only", C.c_int32), ("library", C.c_int32),
                ("n_ref", C.c_int32)]


BAM_STATUS_NAMES = ["OK", "UNMAPPED", "BAD_REF", "ZERO_OP", "OP", "INTRON_LONG", "INTRON_SHORT", "INDEL", "SHORT", "MULTI", "TRUNCATED"]


class sbgpu_hits_t(C.Structure):
    _fields_ = [
        ("n_hits", C.c_int64),
        ("hit_locus", C.c_void_p),
        ("feat_off", C.c_void_p),
        ("feat_code", C.c_void_p),
        ("feat_left", C.c_void_p),
        ("feat_right", C.c_void_p),
    ]


_lib = None


def build():
    """Compile libsbgpu.so for gfx950 (hipcc cross-compiles without a GPU)."""
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "csrc")])


def load():
    """dlopen libsbgpu.so and declare prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SbgpuError(
            "%s is missing: build it with `make -C strawberry_amd/csrc` (or __graft_entry__.build()). "
            "There is no CPU fallback." % LIB_PATH)
    # torch bundles its own libamdhip64.so.7; it must be the ONE HIP runtime of the
    # process (device pointers and streams are shared with it), so it is loaded first
    # and libsbgpu.so's NEEDED libamdhip64.so.7 then resolves to the same object.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, i64p = C.c_void_p, C.POINTER(C.c_int64)
    L.sbgpu_version.restype = C.c_char_p
    L.sbgpu_last_error.restype = C.c_char_p
    L.sbgpu_build_id.restype = C.c_char_p
    L.sbgpu_device_count.restype = C.c_int
    L.sbgpu_init.argtypes = [C.c_int, C.POINTER(vp)]
    L.sbgpu_finalize.argtypes = [vp]
    L.sbgpu_device_info.argtypes = [vp, i64p]
    L.sbgpu_synchronize.argtypes = [vp, vp]
    L.sbgpu_plan_create.argtypes = [vp, C.c_int64, vp, vp, vp, C.POINTER(vp)]
    L.sbgpu_plan_destroy.argtypes = [vp]
    L.sbgpu_plan_info.argtypes = [vp, i64p]
    L.sbgpu_plan_classes.argtypes = [vp, i64p, C.c_int]
    L.sbgpu_plan_locus_kinds.argtypes = [vp, vp]
    L.sbgpu_em_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.sbgpu_set_timing.argtypes = [vp, C.c_int]
    L.sbgpu_last_stage_ms.argtypes = [vp, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_char_p)]
    L.sbgpu_comm_unique_id.argtypes = [vp]
    L.sbgpu_comm_init.argtypes = [vp, C.c_int, C.c_int, vp, C.POINTER(vp)]
    L.sbgpu_comm_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.sbgpu_comm_rccl_ranks.argtypes = [vp, C.POINTER(C.c_int)]
    L.sbgpu_comm_destroy.argtypes = [vp]
    L.sbgpu_allreduce_sum_f64.argtypes = [vp, vp, C.c_int64, vp]
    L.sbgpu_allreduce_sum_i64.argtypes = [vp, vp, C.c_int64, vp]
    L.sbgpu_allreduce_sum_f64_host.argtypes = [vp, vp, C.c_int64]
    L.sbgpu_allreduce_sum_i64_host.argtypes = [vp, vp, C.c_int64]
    L.sbgpu_em_last_phase_ms.argtypes = [vp, C.POINTER(C.c_float), C.c_int]
    L.sbgpu_em_run_device.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    L.sbgpu_em_run_device_f32.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    L.sbgpu_em_run_device_split.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.sbgpu_em_run_device_bias.argtypes = [vp] * 10
    L.sbgpu_em_run_device_bias_f32.argtypes = [vp] * 10
    L.sbgpu_em_batch.argtypes = [vp, C.POINTER(sbgpu_batch_t), vp, vp, vp]
    L.sbgpu_abundance_device.argtypes = [vp, vp, vp, vp, vp, C.POINTER(sbgpu_abundance_params_t), vp, vp, vp, vp, vp]
    L.sbgpu_tpm_device.argtypes = [vp, C.c_int64, vp, vp, vp, vp, vp]
    L.sbgpu_insert_pdf_table.argtypes = [C.POINTER(sbgpu_insert_t), C.c_int32, vp]
    L.sbgpu_binweight_device.argtypes = [vp, C.c_int64, vp, vp, vp, vp, vp, vp, C.c_int32, C.c_int32, C.c_int32,
                                         C.c_int32, vp, vp]
    L.sbgpu_exonbin_device.argtypes = [vp, C.POINTER(sbgpu_annotation_t), C.POINTER(sbgpu_hits_t), C.c_int32, C.c_int32,
                                       vp, vp, vp]
    L.sbgpu_exonbin_host.argtypes = [vp, C.POINTER(sbgpu_annotation_t), C.POINTER(sbgpu_hits_t), C.c_int32, C.c_int32,
                                     vp, vp]
    L.sbgpu_segments_host.argtypes = [C.c_int64, vp, vp, vp, vp, vp, vp, vp, C.c_int64]
    L.sbgpu_segments_host.restype = C.c_int64
    L.sbgpu_frag_lens_host.argtypes = [C.POINTER(sbgpu_annotation_t), C.POINTER(sbgpu_hits_t), C.c_int32, vp, vp]
    L.sbgpu_frag_lens_host.restype = C.c_int64
    L.sbgpu_hit_features.argtypes = [C.c_int, vp, vp, vp, C.c_int, vp, vp, vp, vp, vp, vp]
    L.sbgpu_bins_create.argtypes = [C.POINTER(sbgpu_annotation_t), C.POINTER(sbgpu_hits_t), vp, C.c_int32, C.c_int32,
                                    vp, vp, C.POINTER(vp)]
    L.sbgpu_bins_create_device.argtypes = [vp, C.POINTER(sbgpu_annotation_t), C.POINTER(sbgpu_hits_t), vp, vp, C.c_int32,
                                           C.c_int32, vp, vp, vp, vp, C.POINTER(vp)]
    L.sbgpu_quantify_host.argtypes = [vp, C.POINTER(sbgpu_annotation_t), C.POINTER(sbgpu_hits_t), vp, C.POINTER(sbgpu_insert_t),
                                      C.c_int32, C.c_int32, vp, vp, vp, vp, C.POINTER(sbgpu_insert_t), C.POINTER(vp)]
    L.sbgpu_annotation_pin.argtypes = [vp, C.POINTER(sbgpu_annotation_t)]
    L.sbgpu_annotation_unpin.argtypes = [vp]
    L.sbgpu_release_idle_memory.argtypes = []
    L.sbgpu_release_idle_memory.restype = C.c_int64
    L.sbgpu_annotation_unpin_matching.argtypes = [vp, C.POINTER(sbgpu_annotation_t), C.POINTER(C.c_int32)]
    L.sbgpu_quantify_device.argtypes = [vp, C.POINTER(sbgpu_annotation_t), C.POINTER(sbgpu_hits_t), vp, vp, C.POINTER(sbgpu_insert_t),
                                        C.c_int32, C.c_int32, vp, vp, vp, C.POINTER(vp)]
    L.sbgpu_quantify_resident.argtypes = [vp, C.POINTER(sbgpu_annotation_t), C.POINTER(sbgpu_hits_t), vp, vp, C.POINTER(sbgpu_insert_t),
                                          C.c_int32, C.c_int32, C.c_int64, C.POINTER(sbgpu_abundance_params_t), vp,
                                          C.POINTER(sbgpu_insert_t), C.POINTER(sbgpu_abundances_t), C.POINTER(vp)]
    L.sbgpu_allreduce_max_i64.argtypes = [vp, vp, C.c_int64, vp]
    L.sbgpu_allreduce_max_i64_host.argtypes = [vp, vp, C.c_int64]
    L.sbgpu_comm_init_host.argtypes = [vp, C.c_int, C.c_int, HOST_ALLREDUCE_FN, vp, C.POINTER(vp)]
    L.sbgpu_front_stream_begin.argtypes = [vp, C.POINTER(sbgpu_clusters_t), C.POINTER(sbgpu_bam_opts_t), C.c_int64, C.POINTER(vp)]
    L.sbgpu_front_stream_push.argtypes = [vp, vp, C.c_int64, vp, C.c_int64]
    L.sbgpu_front_stream_end.argtypes = [vp, C.POINTER(sbgpu_annotation_t), C.POINTER(sbgpu_insert_t), C.c_int32, C.c_int32,
                                         C.POINTER(sbgpu_abundance_params_t), vp, C.POINTER(sbgpu_insert_t), C.POINTER(sbgpu_abundances_t),
                                         C.POINTER(vp)]
    L.sbgpu_front_stream_info.argtypes = [vp, i64p]
    L.sbgpu_front_stream_hits.argtypes = [vp, C.POINTER(sbgpu_hits_t), C.POINTER(vp), C.POINTER(vp)]
    L.sbgpu_front_stream_destroy.argtypes = [vp]
    L.sbgpu_front_stream_destroy.restype = None
    L.sbgpu_bins_export_weights.argtypes = [vp, vp]
    L.sbgpu_collapse_pairs_host.argtypes = [C.c_int64, C.POINTER(sbgpu_pairs_t), C.POINTER(vp)]
    L.sbgpu_collapse_pairs_device.argtypes = [vp, C.c_int64, C.POINTER(sbgpu_pairs_t), vp, vp, C.POINTER(vp)]
    L.sbgpu_pair_mates_host.argtypes = [C.c_int64, C.POINTER(sbgpu_reads_t), vp, C.POINTER(vp)]
    L.sbgpu_pair_mates_device.argtypes = [vp, C.c_int64, C.POINTER(sbgpu_reads_t), vp, vp, C.POINTER(vp)]
    L.sbgpu_matepairs_destroy.argtypes = [vp]
    L.sbgpu_matepairs_destroy.restype = None
    L.sbgpu_matepairs_info.argtypes = [vp, i64p]
    L.sbgpu_matepairs_pairs.argtypes = [vp, C.POINTER(sbgpu_pairs_t), C.POINTER(vp)]
    L.sbgpu_matepairs_export.argtypes = [vp] * 10
    L.sbgpu_bam_index_host.argtypes = [vp, C.c_int64, vp, C.c_int64]
    L.sbgpu_bam_index_host.restype = C.c_int64
    L.sbgpu_bam_decode_host.argtypes = [vp, C.c_int64, vp, C.c_int64, C.POINTER(sbgpu_bam_opts_t), C.POINTER(vp)]
    L.sbgpu_bam_decode_device.argtypes = [vp, vp, C.c_int64, vp, C.c_int64, C.POINTER(sbgpu_bam_opts_t), vp, C.POINTER(vp)]
    L.sbgpu_bamreads_destroy.argtypes = [vp]
    L.sbgpu_bamreads_destroy.restype = None
    L.sbgpu_bamreads_info.argtypes = [vp, i64p]
    L.sbgpu_bamreads_reads.argtypes = [vp, C.POINTER(sbgpu_reads_t), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.sbgpu_bamreads_export.argtypes = [vp] * 16
    L.sbgpu_assign_reads_host.argtypes = [C.POINTER(sbgpu_clusters_t), C.c_int64, vp, vp, vp, vp, vp, vp]
    L.sbgpu_assign_reads_device.argtypes = [vp, C.POINTER(sbgpu_clusters_t), C.c_int64, vp, vp, vp, vp, vp, vp, vp]
    L.sbgpu_uniq_dev_destroy.argtypes = [vp]
    L.sbgpu_uniq_dev_destroy.restype = None
    L.sbgpu_uniq_dev_info.argtypes = [vp, i64p]
    L.sbgpu_uniq_dev_hits.argtypes = [vp, C.POINTER(sbgpu_hits_t), C.POINTER(vp), C.POINTER(vp)]
    L.sbgpu_uniq_dev_export.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    L.sbgpu_uniq_destroy.argtypes = [vp]
    L.sbgpu_uniq_destroy.restype = None
    L.sbgpu_uniq_info.argtypes = [vp, i64p]
    L.sbgpu_uniq_export.argtypes = [vp] * 8
    L.sbgpu_bins_destroy.argtypes = [vp]
    L.sbgpu_bins_destroy.restype = None
    L.sbgpu_bins_info.argtypes = [vp, i64p]
    L.sbgpu_bins_grouping.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_char_p)]
    L.sbgpu_bins_export.argtypes = [vp] * 14
    L.sbgpu_format_value.argtypes = [C.c_double, C.c_char_p]
    L.sbgpu_format_gtf_transcript.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_char, C.c_char_p, C.c_char_p,
                                              C.c_char_p, C.c_char_p, C.c_int, vp, vp, C.c_double, C.c_double,
                                              C.c_double, C.c_int32]
    L.sbgpu_format_context_row.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int32, C.c_char_p, C.c_uint32, C.c_int,
                                           C.POINTER(C.c_char_p), vp, vp, vp, C.c_int, vp, vp, C.c_uint32]
    L.sbgpu_format_context_row_seq.argtypes = L.sbgpu_format_context_row.argtypes + [C.c_double, C.c_double, C.c_uint32]
    L.sbgpu_binseq_device.argtypes = [vp, vp, C.c_int64, C.c_int64, C.c_int64, vp, vp, vp, vp, vp, vp, vp, vp]
    L.sbgpu_binseq_host.argtypes = [vp, vp, C.c_int64, C.c_int64, C.c_int64, vp, vp, vp, vp, vp, vp]
    L.sbgpu_binweight_host.argtypes = [vp, C.c_int64, vp, vp, vp, vp, C.POINTER(sbgpu_insert_t), vp]
    for name in SYMBOLS:
        f = getattr(L, name)
        if f.restype is C.c_int and name not in ("sbgpu_device_count", "sbgpu_plan_classes"):
            pass
    _lib = L
    return L


def check(rc, what):
    if rc != SBGPU_OK:
        raise SbgpuError("%s failed (%d): %s" % (what, rc, load().sbgpu_last_error().decode()))
