"""The oracle against the hand-derived scaffold-stage answers of tests/kat_scaffold.py (CPU only), and the reference's
own test inputs of tests/test_chain_monotonicity.rs / tests/test_centromere_plane_sweep.rs replayed through the oracle's
command line."""
import os
import subprocess

import numpy as np
import pytest

from tests import kat_scaffold as K
from tests import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "sweepga-ref")
CODE = {K.D: orc.DROPPED, K.S: orc.SCAFFOLD, K.R: orc.RESCUED}
MODE = {"OneToOne": orc.ONE_TO_ONE, "OneToMany": orc.ONE_TO_MANY, "ManyToMany": orc.MANY_TO_MANY}


def orc_config(kw):
    return orc.Config(**{k: (MODE[v] if isinstance(v, str) else v) for k, v in kw.items()})


@pytest.mark.parametrize("case", K.CASES, ids=[c["name"] for c in K.CASES])
def test_oracle_matches_hand_derivation(case, tmp_path):
    rec = orc.parse_paf_text(K.paf_text(case))
    st, ch = orc.apply_filters(orc_config(case["cfg"]), rec)
    assert [int(x) for x in st] == [CODE[s] for s, _ in case["expect"]]
    assert [int(x) for x in ch] == [c for _, c in case["expect"]]
    # and through filter_paf: the annotated output text
    inp, out = tmp_path / "i.paf", tmp_path / "o.paf"
    inp.write_text(K.paf_text(case))
    orc.filter_paf(orc_config(case["cfg"]), str(inp), str(out))
    assert out.read_text() == K.expected_output(case)


def _run(binary, paf_path, flags):
    r = subprocess.run([binary, str(paf_path), *flags], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return r.stdout


@pytest.mark.parametrize("case", K.REPLAY, ids=[c["name"] for c in K.REPLAY])
def test_reference_test_inputs_replayed(case, tmp_path):
    """The asserts of the reference's own tests hold for the oracle's command line (output on stdout, as in those tests)."""
    p = tmp_path / "i.paf"
    p.write_text(case["paf"])
    lines = [ln for ln in _run(REF, p, case["flags"]).splitlines() if ln and not ln.startswith("[")]
    if case["count"] is not None:
        assert len(lines) == case["count"]
    if case.get("must_contain"):
        assert any(case["must_contain"] in ln for ln in lines)


def test_larger_jump_keeps_a_superset():
    """tests/test_chain_monotonicity.rs header: with identity filters that every line passes, a larger --scaffold-jump
    keeps a superset of the lines a smaller one keeps."""
    import tempfile
    for text in (K._collinear(), K._fragmented()):
        prev = None
        for gap in (1_000, 2_000, 5_000, 10_000, 30_000, 100_000, 500_000):
            with tempfile.TemporaryDirectory() as d:
                p = os.path.join(d, "i.paf")
                with open(p, "w") as f:
                    f.write(text)
                out = _run(REF, p, ["--scaffold-jump", str(gap), "--min-aln-identity", "0.90", "--scaffold-mass", "0"])
            kept = {ln.split("\tch:Z:")[0].split("\tst:Z:")[0] for ln in out.splitlines() if ln}
            if prev is not None:
                assert prev <= kept
            prev = kept
