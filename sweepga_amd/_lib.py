"""ctypes binding of include/sweepga_gpu.h.  There is no CPU fallback: if the HIP library is
missing or no GPU is usable, every compute call raises."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libsweepga_gpu.so")

SWG_OK = 0
ERR_NAMES = {-1: "SWG_ERR_INVALID", -2: "SWG_ERR_NO_DEVICE", -3: "SWG_ERR_HIP", -4: "SWG_ERR_OOM",
             -5: "SWG_ERR_RANGE", -6: "SWG_ERR_UNSUPPORTED"}
K_INF = 2**64 - 1

# every symbol include/sweepga_gpu.h declares
SYMBOLS = ["swg_abi_version", "swg_create", "swg_destroy", "swg_last_error", "swg_stream", "swg_synchronize",
           "swg_filter", "swg_filter_device", "swg_filter64", "swg_filter_device64", "swg_plane_sweep", "swg_plane_sweep_scaffolds",
           "swg_merge_chains", "swg_union_find_sets", "swg_log", "swg_log_range", "swg_profile_enable",
           "swg_profile_reset", "swg_profile_count", "swg_profile_get", "swg_profile_units", "swg_profile_select",
           "swg_paf_open", "swg_paf_open_buffer", "swg_paf_close", "swg_paf_records", "swg_paf_num_lines",
           "swg_paf_ranks", "swg_paf_num_sequences", "swg_paf_sequence_name", "swg_paf_timing", "swg_paf_text",
           "swg_paf_write", "swg_filter_paf", "swg_paf_last_error",
           "swg_parse_ani_method", "swg_parse_identity_value", "swg_paf_ani_input", "swg_ani_median", "swg_paf_ani_stats",
           "swg_filter_multi", "swg_filter_multi64", "swg_memory_info", "swg_reserve", "swg_warmup",
           "swg_aln_open", "swg_aln_close", "swg_aln_records", "swg_aln_num_sequences", "swg_aln_sequence_name",
           "swg_paf_seq_offsets", "swg_aln_seq_offsets", "swg_paf_record_offsets", "swg_aln_record_offsets",
           "swg_paf_tree_filter", "swg_free", "swg_stream_plan", "swg_paf_identity_is_derived",
           "swg_alnstats_open", "swg_alnstats_open_buffer", "swg_alnstats_close", "swg_alnstats_get", "swg_alnstats_pair",
           "swg_alnstats_report", "swg_alnstats_compare", "swg_alnstats_last_error"]


class SwgError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"{ERR_NAMES.get(code, code)}: {text}")
        self.code = code


class SwgConfig(C.Structure):
    _fields_ = [
        ("min_block_length", C.c_uint64),
        ("mapping_filter_mode", C.c_int32),
        ("mapping_max_per_query", C.c_uint64),
        ("mapping_max_per_target", C.c_uint64),
        ("scaffold_filter_mode", C.c_int32),
        ("scaffold_max_per_query", C.c_uint64),
        ("scaffold_max_per_target", C.c_uint64),
        ("overlap_threshold", C.c_double),
        ("scaffold_gap", C.c_uint64),
        ("min_scaffold_length", C.c_uint64),
        ("scaffold_overlap_threshold", C.c_double),
        ("scaffold_max_deviation", C.c_uint64),
        ("scoring_function", C.c_int32),
        ("min_identity", C.c_double),
        ("min_scaffold_identity", C.c_double),
        ("keep_self", C.c_int32),
        ("scaffolds_only", C.c_int32),
    ]


class SwgRecords(C.Structure):
    _fields_ = [
        ("n", C.c_uint64),
        ("q_id", C.c_void_p),
        ("t_id", C.c_void_p),
        ("q_start", C.c_void_p),
        ("q_end", C.c_void_p),
        ("t_start", C.c_void_p),
        ("t_end", C.c_void_p),
        ("identity", C.c_void_p),
        ("matches", C.c_void_p),
        ("block_len", C.c_void_p),
        ("strand", C.c_void_p),
        ("n_seq", C.c_uint32),
        ("seq_genome_last", C.c_void_p),
        ("n_genome_last", C.c_uint32),
        ("seq_genome_two", C.c_void_p),
        ("n_genome_two", C.c_uint32),
    ]


class SwgStats(C.Structure):
    _fields_ = [
        ("n_in", C.c_uint64),
        ("n_retained", C.c_uint64),
        ("n_swept", C.c_uint64),
        ("n_chains", C.c_uint64),
        ("n_chains_kept", C.c_uint64),
        ("n_out", C.c_uint64),
        ("device_ms", C.c_double),
        ("h2d_ms", C.c_double),
        ("d2h_ms", C.c_double),
    ]


class SwgAniInput(C.Structure):
    _fields_ = [
        ("n", C.c_uint64),
        ("eligible", C.c_void_p),
        ("pair", C.c_void_p),
        ("n_pairs", C.c_uint64),
        ("matches", C.c_void_p),
        ("block_len", C.c_void_p),
        ("total_genome_size", C.c_double),
    ]


_lib = None


def load():
    """Loads libsweepga_gpu.so.  Raises (loudly) if the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension is not built (run `python -m sweepga_amd.build` or "
            "__graft_entry__.build()).  sweepga_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    lib.swg_abi_version.restype = C.c_int
    lib.swg_create.restype = C.c_int
    lib.swg_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    lib.swg_destroy.restype = None
    lib.swg_destroy.argtypes = [C.c_void_p]
    lib.swg_last_error.restype = C.c_char_p
    lib.swg_last_error.argtypes = [C.c_void_p]
    lib.swg_stream.restype = C.c_void_p
    lib.swg_stream.argtypes = [C.c_void_p]
    lib.swg_synchronize.restype = C.c_int
    lib.swg_synchronize.argtypes = [C.c_void_p]
    for name in ("swg_filter", "swg_filter_device", "swg_filter64", "swg_filter_device64"):  # swg_records64 = same layout
        f = getattr(lib, name)
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.POINTER(SwgRecords), C.POINTER(SwgConfig), C.c_void_p, C.c_void_p,
                      C.POINTER(SwgStats)]
    lib.swg_plane_sweep.restype = C.c_int
    lib.swg_plane_sweep.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_double, C.c_int,
                                    C.c_void_p]
    lib.swg_plane_sweep_scaffolds.restype = C.c_int
    lib.swg_plane_sweep_scaffolds.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32,
                                              C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_uint64,
                                              C.c_double, C.c_int, C.c_void_p, C.POINTER(C.c_uint64)]
    lib.swg_merge_chains.restype = C.c_int
    lib.swg_merge_chains.argtypes = [C.c_void_p, C.POINTER(SwgRecords), C.c_uint64, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
    lib.swg_union_find_sets.restype = C.c_int
    lib.swg_union_find_sets.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.POINTER(C.c_uint64)]
    lib.swg_log.restype = C.c_int
    lib.swg_log.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    lib.swg_log_range.restype = C.c_int
    lib.swg_log_range.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p]
    lib.swg_profile_enable.restype = C.c_int
    lib.swg_profile_enable.argtypes = [C.c_void_p, C.c_int]
    lib.swg_profile_reset.restype = C.c_int
    lib.swg_profile_reset.argtypes = [C.c_void_p]
    lib.swg_profile_select.restype = C.c_int
    lib.swg_profile_select.argtypes = [C.c_void_p, C.c_char_p]
    lib.swg_profile_count.restype = C.c_int
    lib.swg_profile_count.argtypes = [C.c_void_p]
    lib.swg_profile_units.restype = C.c_int
    lib.swg_profile_units.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
    lib.swg_profile_get.restype = C.c_int
    lib.swg_profile_get.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_uint64),
                                    C.POINTER(C.c_double)]
    lib.swg_paf_open.restype = C.c_int
    lib.swg_paf_open.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
    lib.swg_paf_open_buffer.restype = C.c_int
    lib.swg_paf_open_buffer.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.POINTER(C.c_void_p)]
    lib.swg_paf_close.restype = None
    lib.swg_paf_close.argtypes = [C.c_void_p]
    lib.swg_paf_records.restype = C.POINTER(SwgRecords)
    lib.swg_paf_records.argtypes = [C.c_void_p]
    lib.swg_paf_num_lines.restype = C.c_uint64
    lib.swg_paf_num_lines.argtypes = [C.c_void_p]
    lib.swg_paf_ranks.restype = C.c_void_p
    lib.swg_paf_ranks.argtypes = [C.c_void_p]
    lib.swg_paf_num_sequences.restype = C.c_uint32
    lib.swg_paf_num_sequences.argtypes = [C.c_void_p]
    lib.swg_paf_sequence_name.restype = C.c_char_p
    lib.swg_paf_sequence_name.argtypes = [C.c_void_p, C.c_uint32]
    lib.swg_paf_timing.restype = None
    lib.swg_paf_timing.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.swg_paf_text.restype = C.c_int
    lib.swg_paf_text.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    lib.swg_paf_write.restype = C.c_int
    lib.swg_paf_write.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
    lib.swg_filter_paf.restype = C.c_int
    lib.swg_filter_paf.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.POINTER(SwgConfig), C.c_int,
                                   C.POINTER(SwgStats), C.POINTER(C.c_double)]
    lib.swg_paf_last_error.restype = C.c_char_p
    lib.swg_paf_last_error.argtypes = []
    lib.swg_parse_ani_method.restype = C.c_int
    lib.swg_parse_ani_method.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int)]
    lib.swg_parse_identity_value.restype = C.c_int
    lib.swg_parse_identity_value.argtypes = [C.c_char_p, C.c_double, C.POINTER(C.c_double)]
    lib.swg_paf_ani_input.restype = C.c_int
    lib.swg_paf_ani_input.argtypes = [C.c_void_p, C.c_int, C.POINTER(SwgAniInput)]
    lib.swg_ani_median.restype = C.c_int
    lib.swg_ani_median.argtypes = [C.c_void_p, C.POINTER(SwgAniInput), C.c_void_p, C.c_int, C.c_double, C.c_int,
                                   C.POINTER(C.c_double)]
    lib.swg_filter_multi.restype = C.c_int
    lib.swg_filter_multi.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(SwgRecords), C.POINTER(SwgConfig), C.c_void_p,
                                     C.c_void_p, C.POINTER(SwgStats)]
    lib.swg_filter_multi64.restype = C.c_int
    lib.swg_filter_multi64.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(SwgRecords), C.POINTER(SwgConfig), C.c_void_p,
                                     C.c_void_p, C.POINTER(SwgStats)]
    lib.swg_memory_info.restype = C.c_int
    lib.swg_memory_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.swg_reserve.restype = C.c_int
    lib.swg_reserve.argtypes = [C.c_void_p, C.c_uint64]
    lib.swg_warmup.restype = C.c_int
    lib.swg_warmup.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_int]
    lib.swg_paf_ani_stats.restype = C.c_int
    lib.swg_paf_ani_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_double)]
    lib.swg_aln_open.restype = C.c_int
    lib.swg_aln_open.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    lib.swg_aln_close.restype = None
    lib.swg_aln_close.argtypes = [C.c_void_p]
    lib.swg_aln_records.restype = C.POINTER(SwgRecords)
    lib.swg_aln_records.argtypes = [C.c_void_p]
    for name in ("swg_paf_seq_offsets", "swg_aln_seq_offsets"):
        f = getattr(lib, name)
        f.restype = C.POINTER(C.c_uint64)
        f.argtypes = [C.c_void_p]
    for name in ("swg_paf_record_offsets", "swg_aln_record_offsets"):
        f = getattr(lib, name)
        f.restype = C.POINTER(C.c_uint64)
        f.argtypes = [C.c_void_p, C.c_int]
    lib.swg_aln_num_sequences.restype = C.c_uint32
    lib.swg_aln_num_sequences.argtypes = [C.c_void_p]
    lib.swg_aln_sequence_name.restype = C.c_char_p
    lib.swg_aln_sequence_name.argtypes = [C.c_void_p, C.c_uint32]
    lib.swg_paf_tree_filter.restype = C.c_int
    lib.swg_paf_tree_filter.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_double, C.POINTER(C.c_void_p),
                                        C.POINTER(C.c_uint64)]
    lib.swg_free.restype = None
    lib.swg_free.argtypes = [C.c_void_p]
    _lib = lib
    return lib


class Context:
    """One swg_ctx (one GPU).  Not thread-safe, like the C object it wraps."""

    def __init__(self, device=0):
        self.lib = load()
        h = C.c_void_p()
        rc = self.lib.swg_create(int(device), C.byref(h))
        if rc != SWG_OK:
            raise SwgError(rc, (self.lib.swg_last_error(None) or b"").decode())
        self.handle = h
        self.device = device

    def check(self, rc):
        if rc != SWG_OK:
            raise SwgError(rc, (self.lib.swg_last_error(self.handle) or b"").decode())

    def synchronize(self):
        self.check(self.lib.swg_synchronize(self.handle))

    @property
    def stream(self):
        return self.lib.swg_stream(self.handle)

    def profile(self, on=True):
        self.check(self.lib.swg_profile_enable(self.handle, int(bool(on))))

    def profile_reset(self):
        self.check(self.lib.swg_profile_reset(self.handle))

    def profile_select(self, kernel_name=None):
        """HIP events only around launches of this kernel (None: around every launch again)."""
        self.check(self.lib.swg_profile_select(self.handle, kernel_name.encode() if kernel_name else None))

    def profile_table(self):
        """{kernel name: (launches, total_ms)} accumulated since the last reset."""
        n = self.lib.swg_profile_count(self.handle)
        if n < 0:
            self.check(n)
        out = {}
        for i in range(n):
            name = C.c_char_p()
            launches = C.c_uint64()
            ms = C.c_double()
            self.check(self.lib.swg_profile_get(self.handle, i, C.byref(name), C.byref(launches), C.byref(ms)))
            out[name.value.decode()] = (launches.value, ms.value)
        return out

    def profile_units(self):
        """{kernel name: elements worked on, summed over its launches} for kernels that run on sub-problems of the call
        (the radix sort passes); kernels that always run over the whole record set are absent."""
        n = self.lib.swg_profile_count(self.handle)
        if n < 0:
            self.check(n)
        out = {}
        for i in range(n):
            name = C.c_char_p()
            launches = C.c_uint64()
            ms = C.c_double()
            units = C.c_uint64()
            self.check(self.lib.swg_profile_get(self.handle, i, C.byref(name), C.byref(launches), C.byref(ms)))
            self.check(self.lib.swg_profile_units(self.handle, i, C.byref(units)))
            if units.value:
                out[name.value.decode()] = units.value
        return out

    def memory_info(self):
        """(arena capacity, high-water mark of the last call) in bytes."""
        cap, peak = C.c_uint64(), C.c_uint64()
        self.check(self.lib.swg_memory_info(self.handle, C.byref(cap), C.byref(peak)))
        return cap.value, peak.value

    def reserve(self, arena_bytes):
        self.check(self.lib.swg_reserve(self.handle, int(arena_bytes)))

    def warmup(self, n_records_hint=0, n_seq_hint=0, with_scaffold=True):
        """swg_warmup: device memory for about n_records_hint records and the library's code objects, ahead of the first call
        (meant to run on a thread of its own while the input is read)."""
        self.check(self.lib.swg_warmup(self.handle, C.c_uint64(int(n_records_hint)), C.c_uint32(int(n_seq_hint)),
                                       C.c_int(1 if with_scaffold else 0)))

    def close(self):
        if getattr(self, "handle", None):
            self.lib.swg_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}


def default_context(device=0):
    ctx = _default_ctx.get(device)
    if ctx is None:
        ctx = _default_ctx[device] = Context(device)
    return ctx
