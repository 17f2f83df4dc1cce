""".1aln front end seen from Python: the record derivation of extract_1aln_metadata (src/unified_filter.rs:21-154).

The .1aln DECODER is the reference's un-vendored fastga-rs / onecode dependency and stays with the host that owns it;
`AlnRecords` takes alignments as that reader returns them (full sequence headers, u64 coordinates, matches, strand) and
hands back the SoA columns swg_filter() takes -- unified_filter::filter_file's .1aln branch (:310-317) is then
`PafFilter(cfg).filter_columns(AlnRecords(...).packed())`, and write_1aln_filtered (:158-190) keeps the alignments whose
status is non-zero.  All derivation happens in libsweepga_gpu.so (host code, no GPU needed)."""
import ctypes as C

import numpy as np

from ._lib import SWG_OK, SwgError, load


class SwgAlnInput(C.Structure):
    _fields_ = [
        ("n", C.c_uint64),
        ("query_name", C.POINTER(C.c_char_p)),
        ("target_name", C.POINTER(C.c_char_p)),
        ("query_start", C.c_void_p),
        ("query_end", C.c_void_p),
        ("target_start", C.c_void_p),
        ("target_end", C.c_void_p),
        ("matches", C.c_void_p),
        ("strand", C.c_char_p),
    ]


class AlnRecords:
    """swg_aln_open over decoded alignments: lists of names, u64 arrays, strand characters."""

    def __init__(self, query_names, target_names, query_start, query_end, target_start, target_end, matches, strand):
        self.lib = load()
        n = len(query_names)

        def enc(s):
            return s if isinstance(s, bytes) else s.encode("utf-8", errors="surrogateescape")

        self._q = (C.c_char_p * max(n, 1))(*[enc(s) for s in query_names])
        self._t = (C.c_char_p * max(n, 1))(*[enc(s) for s in target_names])
        self._cols = [np.ascontiguousarray(np.asarray(a, dtype=np.uint64)) for a in
                      (query_start, query_end, target_start, target_end, matches)]
        self._strand = bytes(ord(c) if isinstance(c, str) else int(c) for c in strand) or b"\0"
        inp = SwgAlnInput(n, self._q, self._t, *[a.ctypes.data for a in self._cols], self._strand)
        h = C.c_void_p()
        rc = self.lib.swg_aln_open(C.byref(inp), C.byref(h))
        if rc != SWG_OK:
            raise SwgError(rc, (self.lib.swg_paf_last_error() or b"").decode())
        self.handle = h
        self.records = self.lib.swg_aln_records(h).contents
        self.n = int(self.records.n)

    def _view(self, addr, dtype, n):
        if n == 0 or not addr:
            return np.zeros(0, dtype=dtype)
        buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(addr)
        return np.frombuffer(buf, dtype=dtype, count=n).copy()

    def column(self, name):
        dtype = {"identity": np.float64, "strand": np.uint8}.get(name, np.uint32)
        return self._view(getattr(self.records, name), dtype, self.n)

    @property
    def names(self):
        k = self.lib.swg_aln_num_sequences(self.handle)
        return [self.lib.swg_aln_sequence_name(self.handle, i).decode("utf-8", errors="surrogateescape") for i in range(k)]

    @property
    def seq_offsets(self):
        """[n_seq] what rebasing took off each sequence's coordinates, or None (no value reached 2^32)."""
        ptr = self.lib.swg_aln_seq_offsets(self.handle)
        if not ptr:
            return None
        return self._view(C.addressof(ptr.contents), np.uint64, int(self.records.n_seq))

    def packed(self):
        """The columns as a PackedRecords (what PafFilter.filter_columns takes); copies, so it outlives the handle."""
        from .filter import PackedRecords
        r = self.records
        cols = {k: self.column(k) for k in ("q_id", "t_id", "q_start", "q_end", "t_start", "t_end", "identity", "matches",
                                             "block_len", "strand")}
        g_last = self._view(r.seq_genome_last, np.uint32, int(r.n_seq))
        g_two = self._view(r.seq_genome_two, np.uint32, int(r.n_seq))
        return PackedRecords(self.n, cols, int(r.n_seq), g_last, int(r.n_genome_last), g_two, int(r.n_genome_two), None)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.swg_aln_close(self.handle)
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
