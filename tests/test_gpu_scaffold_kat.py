"""swg_filter against the hand-derived scaffold-stage answers of tests/kat_scaffold.py (status and chain number of every
line, and the annotated output text through swg_filter_paf), and the reference's own test inputs replayed through
sweepga-gpu next to the oracle's command line (byte-identical stdout, the reference's asserted line counts)."""
import os
import subprocess

import numpy as np
import pytest

from tests import kat_scaffold as K

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "sweepga-ref")
CODE = {K.D: 0, K.S: 1, K.R: 2}


def _cfg(sw, kw):
    return sw.FilterConfig(**{k: (getattr(sw.FilterMode, v) if isinstance(v, str) else v) for k, v in kw.items()})


@pytest.mark.parametrize("case", K.CASES, ids=[c["name"] for c in K.CASES])
def test_gpu_matches_hand_derivation(case, tmp_path):
    import sweepga_amd as sw
    inp, out = tmp_path / "i.paf", tmp_path / "o.paf"
    inp.write_text(K.paf_text(case))
    f = sw.PafFilter(_cfg(sw, case["cfg"]))
    meta = f.extract_metadata(inp)
    st, ch = f.filter_columns(sw.pack_records(meta))
    assert [int(x) for x in st] == [CODE[s] for s, _ in case["expect"]]
    assert [int(x) for x in ch] == [c for _, c in case["expect"]]
    f.filter_paf(inp, out)
    assert out.read_text() == K.expected_output(case)


@pytest.mark.parametrize("case", K.REPLAY, ids=[c["name"] for c in K.REPLAY])
def test_reference_test_inputs_replayed_on_gpu(case, tmp_path):
    from sweepga_amd import build
    p = tmp_path / "i.paf"
    p.write_text(case["paf"])
    got = subprocess.run([build.CLI, str(p), "--quiet", *case["flags"]], capture_output=True, text=True)
    assert got.returncode == 0, got.stderr
    want = subprocess.run([REF, str(p), *case["flags"]], capture_output=True, text=True)
    assert want.returncode == 0, want.stderr
    assert got.stdout == want.stdout
    lines = [ln for ln in got.stdout.splitlines() if ln and not ln.startswith("[")]
    if case["count"] is not None:
        assert len(lines) == case["count"]
    if case.get("must_contain"):
        assert any(case["must_contain"] in ln for ln in lines)
