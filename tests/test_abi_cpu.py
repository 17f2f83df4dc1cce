"""CPU-side checks of the product: the C-ABI library builds/loads and exports every symbol the
header declares; the host mirror's pure-host logic (interning, PAF parsing, config mapping)."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from sweepga_amd import build, _lib
    build.build()  # hipcc cross-compiles gfx950 without a GPU
    return _lib.load()


def test_exports_every_declared_symbol(lib):
    from sweepga_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "sweepga_gpu.h")).read()
    declared = set(re.findall(r"\b(swg_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for s in declared:
        assert getattr(lib, s) is not None
    assert lib.swg_abi_version() == 1


def test_struct_layouts_match_header(lib, tmp_path):
    """ctypes mirrors of swg_config / swg_records / swg_stats / swg_ani_input have the C compiler's sizes; the header
    is plain C (compiled with gcc, not g++)."""
    import ctypes as C
    import subprocess
    from sweepga_amd import _lib
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "sweepga_gpu.h"\nint main(void){printf("%zu %zu %zu %zu\\n",'
                   'sizeof(swg_config),sizeof(swg_records),sizeof(swg_stats),sizeof(swg_ani_input));printf("%zu\\n",sizeof(swg_aln_input));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    from sweepga_amd.aln import SwgAlnInput
    assert sizes == [C.sizeof(_lib.SwgConfig), C.sizeof(_lib.SwgRecords), C.sizeof(_lib.SwgStats), C.sizeof(_lib.SwgAniInput),
                     C.sizeof(SwgAlnInput)]


def test_no_gpu_is_a_loud_error(lib):
    """No CPU fallback: without a device, context creation fails with SWG_ERR_NO_DEVICE."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from sweepga_amd import Context, SwgError
    with pytest.raises(SwgError) as e:
        Context(0)
    assert e.value.code == -2


def test_sequence_index_and_prefixes():
    from sweepga_amd import SequenceIndex
    idx = SequenceIndex()
    assert [idx.get_or_insert(x) for x in ("a#1#chr1", "b#1#chr1", "a#1#chr1", "a#1#chr2", "plain", "x#y#z#w")] == [0, 1, 0, 2, 3, 4]
    assert SequenceIndex.prefix_last("SGDref#1#chrI") == "SGDref#1#"      # paf_filter.rs:1022-1030
    assert SequenceIndex.prefix_last("x#y#z#w") == "x#y#z#"
    assert SequenceIndex.prefix_last("plain") == "plain"
    assert SequenceIndex.prefix_two("SGDref#1#chrI") == "SGDref#1#"       # plane_sweep_scaffold.rs:13-22
    assert SequenceIndex.prefix_two("x#y#z#w") == "x#y#"
    assert SequenceIndex.prefix_two("r#1") == "r#1#"
    assert SequenceIndex.prefix_two("plain") == "plain"
    last, n_last, two, n_two = idx.genome_tables()
    assert list(last[:5]) == [0, 1, 0, 2, 3] and n_last == 4
    assert list(two[:5]) == [0, 1, 0, 2, 3] and n_two == 4


def test_host_paf_parser_matches_oracle_parser(tmp_path):
    """The host mirror's extract_metadata against the oracle's restatement of paf_filter.rs:292-376."""
    from sweepga_amd import FilterConfig, PafFilter
    from tests import orc
    text = ("q1\t1000\t10\t200\t+\tt1\t2000\t30\t220\t150\t190\t60\tNM:i:3\tcg:Z:100=5X50=2I3D\n"
            "q1\t1000\t10\t200\t-\tt1\t2000\t30\t220\t150\t190\t60\tdv:f:0.05\n"
            "short\tline\n"
            "q2\t1000\tx\t+7\t*\tt2\t2000\t1\t2\t\t0\t60\tcg:Z:10M\tdv:f:abc\n"
            "q3\t1\t0\t5\t+\tt3\t1\t0\t5\t5\t5\t0\tcg:Z:5=\tdv:f:0.5\r\n")
    p = tmp_path / "x.paf"
    p.write_text(text, newline="")
    mine = PafFilter(FilterConfig()).extract_metadata(str(p))
    ref = orc.parse_paf_text(text)
    assert len(mine) == len(ref) == 4
    for i, m in enumerate(mine):
        assert (m.rank, m.query_name, m.target_name) == (int(ref.rank[i]), ref.qname[i], ref.tname[i])
        assert (m.query_start, m.query_end, m.target_start, m.target_end) == (int(ref.qs[i]), int(ref.qe[i]), int(ref.ts[i]), int(ref.te[i]))
        assert (m.block_length, m.matches, ord(m.strand)) == (int(ref.block_length[i]), int(ref.matches[i]), int(ref.strand[i]))
        assert m.identity == ref.identity[i]
    assert [m.rank for m in mine] == [0, 1, 3, 4]
    assert mine[0].matches == 150 and mine[0].identity == 150 / 190      # cg:Z: '=' total overrides column 10
    assert mine[1].identity == 1.0 - 0.05                                # dv:f:
    assert mine[2].query_start == 0 and mine[2].query_end == 7 and mine[2].strand == "-" and mine[2].block_length == 0
    assert mine[3].identity == 0.5                                       # last tag wins


def test_pack_records_keeps_wide_coordinates():
    """RecordMeta is u64 (src/paf_filter.rs:58-62): values >= 2^32 go to the library in swg_records64 (rebased there)."""
    from sweepga_amd import RecordMeta, pack_records
    m = RecordMeta(0, "a", "b", 0, 2**32, 0, 10, 10, 1.0, 10, 10, "+")
    p = pack_records([m])
    assert p.wide and p.cols["q_end"].dtype == np.uint64 and int(p.cols["q_end"][0]) == 2**32
    assert p.cols["matches"].dtype == np.uint64
    m = RecordMeta(0, "a", "b", 0, 2**32 - 1, 0, 10, 10, 1.0, 10, 10, "+")
    p = pack_records([m])
    assert not p.wide and p.cols["q_end"].dtype == np.uint32


def test_device_log_restatement_on_host():
    """sweepga_amd/csrc/swg_log.h (the arithmetic the kernels run) == host libm log, checked on the CPU
    for every integer length < 2^24 and a strided sample up to 2^40."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "native", "log_check")
    src = exe + ".cpp"
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
        subprocess.check_call(["g++", "-O2", "-fopenmp", "-mfma", "-o", exe, src])
    assert subprocess.run([exe, "1", str(1 << 24)], capture_output=True, text=True).stdout.strip() == "0"
    assert subprocess.run([exe, str(1 << 24), str(1 << 22), "262147"], capture_output=True, text=True).stdout.strip() == "0"
