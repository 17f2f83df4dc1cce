"""alnstats (src/bin/alnstats.rs) seen from Python: AlignmentStats of a PAF, print_stats / compare_stats text.
Host code of libsweepga_gpu.so (no GPU needed)."""
import ctypes as C
import os

from ._lib import SWG_OK, SwgError, load


class SwgAlnstatsSummary(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("total_mappings", "total_bases", "total_matches", "self_mappings", "inter_chromosomal",
                                           "inter_genome", "chr_pair_count", "genome_pairs", "above_95_pct")] + \
               [("avg_identity", C.c_double), ("avg_coverage", C.c_double)]


class AlnStats:
    """parse_paf (:103-164) over a file (plain / .gz / .bgz) or over text in memory."""

    def __init__(self, path=None, text=None, threads=0):
        self.lib = lib = load()
        lib.swg_alnstats_open.restype = C.c_int
        lib.swg_alnstats_open.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
        lib.swg_alnstats_open_buffer.restype = C.c_int
        lib.swg_alnstats_open_buffer.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.POINTER(C.c_void_p)]
        lib.swg_alnstats_close.restype = None
        lib.swg_alnstats_close.argtypes = [C.c_void_p]
        lib.swg_alnstats_get.restype = C.POINTER(SwgAlnstatsSummary)
        lib.swg_alnstats_get.argtypes = [C.c_void_p]
        lib.swg_alnstats_pair.restype = C.c_int
        lib.swg_alnstats_pair.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_double),
                                          C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        lib.swg_alnstats_report.restype = C.c_int
        lib.swg_alnstats_report.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        lib.swg_alnstats_compare.restype = C.c_int
        lib.swg_alnstats_compare.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        lib.swg_alnstats_last_error.restype = C.c_char_p
        lib.swg_free.restype = None
        lib.swg_free.argtypes = [C.c_void_p]
        h = C.c_void_p()
        if text is not None:
            data = text if isinstance(text, bytes) else text.encode("utf-8", errors="surrogateescape")
            rc = lib.swg_alnstats_open_buffer(data, len(data), threads, C.byref(h))
        else:
            rc = lib.swg_alnstats_open(os.fsencode(path), threads, C.byref(h))
        if rc != SWG_OK:
            raise SwgError(rc, (lib.swg_alnstats_last_error() or b"").decode(errors="replace"))
        self.handle = h
        self.summary = lib.swg_alnstats_get(h).contents

    @property
    def pairs(self):
        """[(query genome, target genome, coverage %, bases, matches)] in order of first appearance."""
        out = []
        for i in range(int(self.summary.genome_pairs)):
            q, t, c, b, m = C.c_char_p(), C.c_char_p(), C.c_double(), C.c_uint64(), C.c_uint64()
            self.lib.swg_alnstats_pair(self.handle, i, C.byref(q), C.byref(t), C.byref(c), C.byref(b), C.byref(m))
            out.append((q.value.decode(errors="surrogateescape"), t.value.decode(errors="surrogateescape"), c.value, b.value, m.value))
        return out

    def _text(self, rc, p, n):
        if rc != SWG_OK:
            raise SwgError(rc, (self.lib.swg_alnstats_last_error() or b"").decode(errors="replace"))
        s = C.string_at(p.value, n.value)
        self.lib.swg_free(p)
        return s

    def report(self, label, detailed=False):
        """print_stats (:166-228) as bytes."""
        p, n = C.c_void_p(), C.c_uint64()
        return self._text(self.lib.swg_alnstats_report(self.handle, os.fsencode(label), int(detailed), C.byref(p), C.byref(n)), p, n)

    def compare(self, other, file1, file2):
        """compare_stats (:230-284) as bytes."""
        p, n = C.c_void_p(), C.c_uint64()
        return self._text(self.lib.swg_alnstats_compare(self.handle, other.handle, os.fsencode(file1), os.fsencode(file2), C.byref(p),
                                                        C.byref(n)), p, n)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.swg_alnstats_close(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass
