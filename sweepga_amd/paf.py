"""Native PAF ingest/egress (sweepga_amd/csrc/host/paf_io.cpp) seen from Python.

`PafFile` is the host-side half of PafFilter::filter_paf (src/paf_filter.rs:278-289): it owns the mapped
text, the SoA columns swg_filter() takes and the line table the writer reuses.  Parsing and writing run on
host threads in C++; nothing here needs a GPU."""
import ctypes as C
import os

import numpy as np

from ._lib import SWG_OK, SwgError, load


def _err(lib, rc):
    raise SwgError(rc, (lib.swg_paf_last_error() or b"").decode())


class PafFile:
    """open_paf_input + extract_metadata (src/paf.rs:10-30, src/paf_filter.rs:292-376)."""

    def __init__(self, path=None, text=None, threads=0):
        self.lib = load()
        h = C.c_void_p()
        if text is not None:
            if isinstance(text, str):
                text = text.encode("utf-8", errors="surrogateescape")
            rc = self.lib.swg_paf_open_buffer(text, len(text), int(threads), C.byref(h))
        else:
            rc = self.lib.swg_paf_open(os.fsencode(path), int(threads), C.byref(h))
        if rc != SWG_OK:
            _err(self.lib, rc)
        self.handle = h
        self.threads = int(threads)
        self.records = self.lib.swg_paf_records(h).contents  # SwgRecords, pointers owned by the handle
        self.n = int(self.records.n)
        self.n_lines = int(self.lib.swg_paf_num_lines(h))

    def _view(self, addr, dtype, n):
        if n == 0 or not addr:
            return np.zeros(0, dtype=dtype)
        buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(addr)
        a = np.frombuffer(buf, dtype=dtype, count=n)
        a.flags.writeable = False
        return a

    def column(self, name):
        """Read-only numpy view of one record column (valid while the PafFile is open).  When `is_rebased` (a value of the
        file reached 2^32) the four coordinate columns are RELATIVE to `seq_offsets[sequence id]`; `absolute(name)` gives
        RecordMeta's own u64 values."""
        dtype = {"identity": np.float64, "strand": np.uint8}.get(name, np.uint32)
        return self._view(getattr(self.records, name), dtype, self.n)

    @property
    def identity_is_derived(self):
        """True when every record's identity is matches / max(block_len, 1) (no dv:f: tag had the last word): a caller may
        pass identity = NULL to swg_filter and the device evaluates it (swg_paf_identity_is_derived)."""
        self.lib.swg_paf_identity_is_derived.restype = C.c_int
        self.lib.swg_paf_identity_is_derived.argtypes = [C.c_void_p]
        return bool(self.lib.swg_paf_identity_is_derived(self.handle))

    @property
    def is_rebased(self):
        """True when the coordinate columns are relative to `seq_offsets` or `record_offsets` (results of the filter are
        unaffected)."""
        return self.seq_offsets is not None or self.record_offsets(0) is not None

    def record_offsets(self, axis):
        """[n] what was taken off every record's query (axis 0) / target (axis 1) coordinates when the file has a sequence that is
        touched over 2^32 bases or more (rebased per sweep segment), else None (swg_paf_record_offsets)."""
        ptr = self.lib.swg_paf_record_offsets(self.handle, int(axis))
        if not ptr:
            return None
        return self._view(C.addressof(ptr.contents), np.uint64, self.n)

    def absolute(self, name):
        """q_start / q_end / t_start / t_end as u64 in the file's own coordinates (a copy)."""
        if name not in ("q_start", "q_end", "t_start", "t_end"):
            raise ValueError("absolute() is for the four coordinate columns")
        v = self.column(name).astype(np.uint64)
        per_record = self.record_offsets(0 if name[0] == "q" else 1)
        if per_record is not None:
            return v + per_record
        off = self.seq_offsets
        if off is None:
            return v
        ids = self.column("q_id" if name[0] == "q" else "t_id")
        return v + off[ids]

    @property
    def ranks(self):
        return self._view(self.lib.swg_paf_ranks(self.handle), np.uint64, self.n)

    @property
    def names(self):
        k = self.lib.swg_paf_num_sequences(self.handle)
        return [self.lib.swg_paf_sequence_name(self.handle, i).decode("utf-8", errors="surrogateescape")
                for i in range(k)]

    @property
    def seq_genome_last(self):
        return self._view(self.records.seq_genome_last, np.uint32, max(int(self.records.n_seq), 1))

    @property
    def seq_genome_two(self):
        return self._view(self.records.seq_genome_two, np.uint32, max(int(self.records.n_seq), 1))

    @property
    def seq_offsets(self):
        """[n_seq] what rebasing took off each sequence's coordinates, or None (no value of the file reached 2^32)."""
        ptr = self.lib.swg_paf_seq_offsets(self.handle)
        if not ptr:
            return None
        return self._view(C.addressof(ptr.contents), np.uint64, int(self.records.n_seq))

    @property
    def timing_ms(self):
        a, b = C.c_double(), C.c_double()
        self.lib.swg_paf_timing(self.handle, C.byref(a), C.byref(b))
        return {"load": a.value, "parse": b.value}

    def write(self, out_path, status, chain=None, threads=None):
        """write_filtered_output (src/paf_filter.rs:1689-1726) -> number of records written."""
        status = np.ascontiguousarray(status, dtype=np.uint8)
        if status.size < self.n:
            raise ValueError("status has fewer entries than records")
        if chain is not None:
            chain = np.ascontiguousarray(chain, dtype=np.uint32)
            if chain.size < self.n:
                raise ValueError("chain has fewer entries than records")
        kept = C.c_uint64()
        rc = self.lib.swg_paf_write(self.handle, os.fsencode(out_path), status.ctypes.data if status.size else None,
                                    chain.ctypes.data if chain is not None and chain.size else None,
                                    self.threads if threads is None else int(threads), C.byref(kept))
        if rc != SWG_OK:
            _err(self.lib, rc)
        return kept.value

    def close(self):
        if getattr(self, "handle", None):
            self.records = None
            self.lib.swg_paf_close(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
