"""ANI pre-pass on the GPU (csrc/swg_ani.hip + host parse) against the oracle's calculate_ani_stats
(main.rs:334-688): bit-identical medians for every method, and byte-identical CLI output when the thresholds
are `aniN` presets."""
import os
import subprocess

import numpy as np
import pytest

from sweepga_amd import AniMethod, AniMethodKind, NSort, SwgError, calculate_ani_stats
from tests import gen, orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

METHODS = [
    (AniMethod(AniMethodKind.All), (orc.ANI_ALL, 50.0, orc.NSORT_IDENTITY)),
    (AniMethod(AniMethodKind.Orthogonal), (orc.ANI_ORTHOGONAL, 50.0, orc.NSORT_IDENTITY)),
    (AniMethod(AniMethodKind.NPercentile, 50.0, NSort.Identity), (orc.ANI_NPERCENTILE, 50.0, orc.NSORT_IDENTITY)),
    (AniMethod(AniMethodKind.NPercentile, 100.0, NSort.Identity), (orc.ANI_NPERCENTILE, 100.0, orc.NSORT_IDENTITY)),
    (AniMethod(AniMethodKind.NPercentile, 7.5, NSort.Length), (orc.ANI_NPERCENTILE, 7.5, orc.NSORT_LENGTH)),
    (AniMethod(AniMethodKind.NPercentile, 20.0, NSort.Score), (orc.ANI_NPERCENTILE, 20.0, orc.NSORT_SCORE)),
    (AniMethod(AniMethodKind.NPercentile, 0.001, NSort.Identity), (orc.ANI_NPERCENTILE, 0.001, orc.NSORT_IDENTITY)),
]


def paf_text(rng, n, n_genomes, chr_len):
    rec = gen.random_records(rng, n, n_genomes=n_genomes, chrs_per_genome=3, span=chr_len - 10_000)
    text = gen.records_to_paf(rng, rec)  # a third of the lines carry dv:f: (non-integral matches in the ANI pass)
    # realistic sequence lengths in columns 2 / 7 so that the N-percentile cut falls inside the data
    out = []
    for ln in text.split("\n"):
        f = ln.split("\t")
        if len(f) >= 11:
            f[1] = f[6] = str(chr_len)
        out.append("\t".join(f))
    return "\n".join(out)


@pytest.mark.parametrize("seed,n,chr_len", [(1, 400, 40_000), (2, 6000, 300_000), (3, 30_000, 2_000_000)])
def test_calculate_ani_stats_bit_exact(tmp_path, seed, n, chr_len):
    rng = np.random.default_rng(900 + seed)
    p = tmp_path / "in.paf"
    p.write_text(paf_text(rng, n, int(rng.integers(2, 6)), chr_len))
    for threads in (1, 5):
        for method, (k, pct, so) in METHODS:
            want = orc.calculate_ani_stats(p, k, pct, so)
            got = calculate_ani_stats(p, method, threads=threads)
            assert np.float64(got).tobytes() == np.float64(want).tobytes(), (method, got, want)
            assert 0.0 < got <= 1.0


def test_single_pair_and_no_pairs(tmp_path):
    rng = np.random.default_rng(4)
    p = tmp_path / "one.paf"
    p.write_text(paf_text(rng, 3000, 2, 500_000))   # two genomes -> exactly one unordered pair
    for method, (k, pct, so) in METHODS:
        assert calculate_ani_stats(p, method) == orc.calculate_ani_stats(p, k, pct, so)
    q = tmp_path / "self.paf"
    q.write_text(paf_text(rng, 500, 1, 500_000))    # one genome: nothing takes part
    for method, _ in METHODS:
        assert calculate_ani_stats(q, method) == 0.0


def test_errors(tmp_path):
    row = lambda m, b, tag="": "\t".join(["A#1#c", "1000", "0", "100", "+", "B#1#c", "1000", "0", "100", m, b, "60"] + ([tag] if tag else []))
    nan = tmp_path / "nan.paf"
    nan.write_text(row("nan", "100") + "\n" + row("5", "100") + "\n")
    with pytest.raises(SwgError, match="NaN"):   # the reference panics in partial_cmp().unwrap()
        calculate_ani_stats(nan, AniMethod(AniMethodKind.NPercentile, 50.0, NSort.Identity))
    frac = tmp_path / "frac.paf"
    frac.write_text(row("5", "10.5") + "\n" + row("5", "100") + "\n")
    with pytest.raises(SwgError, match="block length"):
        calculate_ani_stats(frac, AniMethod(AniMethodKind.NPercentile, 50.0, NSort.Length))
    # file-order methods add the values as they are
    assert calculate_ani_stats(frac, AniMethod(AniMethodKind.All)) == orc.calculate_ani_stats(frac, orc.ANI_ALL) == 10.0 / 110.5


ANI_FLAGS = [
    ["--min-aln-identity", "ani50-5", "--scaffold-jump", "0"],
    ["--min-aln-identity", "ani", "--ani-method", "all", "--num-mappings", "1:1"],
    ["--min-scaffold-identity", "ani50+1", "--ani-method", "orthogonal", "--scaffold-jump", "20k", "--scaffold-mass", "2k"],
    ["--min-aln-identity", "ANI25-10", "--min-scaffold-identity", "ani50-2", "--ani-method", "n30-score", "--scaffold-jump", "20k",
     "--scaffold-mass", "2k", "--scaffold-dist", "10k"],
    ["--min-aln-identity", "ani50-3", "--ani-method", "nonsense"],   # unknown method -> n50-identity (main.rs:3578)
]


def test_cli_with_ani_thresholds_byte_identical(tmp_path):
    from sweepga_amd import build
    ref = os.path.join(ROOT, "oracle", "sweepga-ref")
    rng = np.random.default_rng(31)
    paf = tmp_path / "in.paf"
    paf.write_text(paf_text(rng, 20_000, 4, 1_000_000))
    for k, flags in enumerate(ANI_FLAGS):
        o1, o2 = tmp_path / f"gpu{k}.paf", tmp_path / f"ref{k}.paf"
        r = subprocess.run([build.CLI, str(paf), "--output-file", str(o1), *flags], capture_output=True, text=True)
        assert r.returncode == 0 and "ANI pre-pass" in r.stderr, r.stderr
        subprocess.check_call([ref, str(paf), "--output-file", str(o2), *flags])
        assert o1.read_bytes() == o2.read_bytes(), flags
        assert 0 < os.path.getsize(o1) < os.path.getsize(paf)
