"""Hand-derived known answers for the scaffold stage of PafFilter::apply_filters (src/paf_filter.rs:436-747).

Every expected value below was worked out by hand from the Rust source (line numbers cited per case), NOT produced by
running the oracle or the GPU: the CPU test (tests/test_scaffold_kat_cpu.py) holds the oracle to them and the GPU test
(tests/test_gpu_scaffold_kat.py) holds swg_filter to them, so a misreading shared by the oracle and the kernels would
have to be shared by this derivation as well.

A case = dict(name, cfg = keyword arguments of the config (orc.Config / sweepga_amd.FilterConfig share the names used
here), lines = PAF records in INPUT order as (qname, qs, qe, strand, tname, ts, te, matches, block), expect = one
(status, chain number) per line with status in {"dropped", "scaffold", "rescued"}).
Shared settings: mapping filter many:many (paf_filter.rs:1004-1014 -> k = inf on both axes: every mapping with
start < end survives, plane_sweep_exact.rs:219-228), min_identity 0, min_scaffold_identity 0.
"""

D, S, R = "dropped", "scaffold", "rescued"


def _c(name, cfg, rows, why):
    return dict(name=name, cfg=cfg, lines=[r[0] for r in rows], expect=[r[1] for r in rows], why=why)


CASES = []

# ------------------------------------------------------------------------------------------------------------------
# 1. '-' strand chaining gaps (paf_filter.rs:813-833).  G = scaffold_gap = 1000, so overlaps up to G/5 = 200 count as
#    distance and larger ones reject (:802-809, :824-833).  One (q, t, '-') group, sorted by q_start (:777):
#      A q[0,1000)     t[9000,10000)
#      B q[1100,2100)  t[7900,8900)   A->B: q_gap 100; '-': t_start[A] 9000 >= t_end[B] 8900 -> r_gap 100; d = 20,000.
#                                     (the '+' formula would see t_end[A] - t_start[B] = 2100 > 200 and reject)
#      C q[2200,3200)  t[7000,8050)   B->C: q_gap 100; t_start[B] 7900 < t_end[C] 8050 -> overlap 150 <= 200 -> r_gap 150;
#                                     d = 32,500.   A->C: q_start[C] 2200 > q_end[A] + G = 2000 -> out of window (:794).
#      D q[3300,4300)  t[6500,7300)   C->D: t_start[C] 7000 < t_end[D] 7300 -> overlap 300 > 200 -> r_gap = G+1: reject.
#                                     B->D: 3300 > 2100 + 1000: out of window.  D starts a new chain.
#      E q[4400,5400)  t[4500,5500)   D->E: q_gap 100; r_gap = t_start[D] 6500 - t_end[E] 5500 = 1000 = G: allowed (<=, :836).
#      F q[5500,6500)  t[2499,3499)   E->F: r_gap = 4500 - 3499 = 1001 > G: reject.  F alone.
#    Chains in get_sets order (smallest sorted position first): {A,B,C} span 3200, {D,E} span 2100, {F} span 1000.
#    min_scaffold_length 1500 removes {F} (:449-455); scaffold filter many:many keeps both others
#    (plane_sweep_scaffold.rs:204-251 with usize::MAX limits); ids follow that order (:517-521): chain_1, chain_2.
#    scaffold_max_deviation 0: nothing is rescued (:680, :740), F is dropped.  Input order is shuffled.
_cfg1 = dict(scaffold_gap=1000, min_scaffold_length=1500, scaffold_max_deviation=0)
_q, _t = "g1#1#chrA", "g2#1#chrB"
CASES.append(_c("minus_strand_gaps", _cfg1, [
    ((_q, 2200, 3200, "-", _t, 7000, 8050, 950, 1000), (S, 1)),   # C
    ((_q, 0, 1000, "-", _t, 9000, 10000, 950, 1000), (S, 1)),     # A
    ((_q, 5500, 6500, "-", _t, 2499, 3499, 950, 1000), (D, 0)),   # F
    ((_q, 1100, 2100, "-", _t, 7900, 8900, 950, 1000), (S, 1)),   # B
    ((_q, 4400, 5400, "-", _t, 4500, 5500, 950, 1000), (S, 2)),   # E
    ((_q, 3300, 4300, "-", _t, 6500, 7300, 950, 1000), (S, 2)),   # D
], "paf_filter.rs:794-836 ('-' branch :824-833), :449-455, :517-521"))

# ------------------------------------------------------------------------------------------------------------------
# 2. Inversion capture (paf_filter.rs:535-597).  G = 1000, min_scaffold_length 5000, no rescue.
#    P: '+' q[10000,20000) t[30000,40000): the only chain long enough -> chain_1, diagonal_offset = 20000 (:557).
#    A '-' mapping of the same (q, t) that is not an anchor yet is looked at iff NOT (q_end < 10000 - 1000 or
#    q_start > 20000 + 1000) (:574-580) and joins chain_1 iff floor(|t_c - q_c - 20000| / sqrt 2) <= 1000 (:585-592), with
#    integer centres (s + e) / 2.  |dev| = 1415 -> 1000.56 -> 1000: in; |dev| = 1416 -> 1001.26 -> 1001: out.
#      M1 q[12000,12100) c 12050, t[33415,33515) c 33465: dev +1415 -> in
#      M2 q[14000,14100) c 14050, t[35416,35516) c 35466: dev +1416 -> out
#      M3 q[16000,16100) c 16050, t[34585,34685) c 34635: dev -1415 -> in
#      M4 q[8900,9000):  q_end 9000 is not < 9000 -> looked at; t[28900,29000): dev 0 -> in
#      M5 q[8899,8999):  q_end 8999 < 9000 -> skipped; t[28899,28999)
#      M6 q[21000,21100): q_start 21000 is not > 21000 -> looked at; t[41000,41100): dev 0 -> in
#      M7 q[21001,21101): q_start 21001 > 21000 -> skipped; t[41001,41101)
#    The '-' mappings form their own chains ({M5,M4} and {M6,M7} overlap by 99 <= 200 and chain; the rest are > G apart),
#    all of span <= 101 < 5000: removed by the length filter, so none is a pre-sweep scaffold member (:470-476) and the
#    captured ones become anchors = status scaffold (:664-672).  The others: max_deviation 0 -> not kept.
_cfg2 = dict(scaffold_gap=1000, min_scaffold_length=5000, scaffold_max_deviation=0)
CASES.append(_c("inversion_capture_window_and_sqrt2", _cfg2, [
    ((_q, 12000, 12100, "-", _t, 33415, 33515, 95, 100), (S, 1)),      # M1
    ((_q, 10000, 20000, "+", _t, 30000, 40000, 9500, 10000), (S, 1)),  # P
    ((_q, 14000, 14100, "-", _t, 35416, 35516, 95, 100), (D, 0)),      # M2
    ((_q, 16000, 16100, "-", _t, 34585, 34685, 95, 100), (S, 1)),      # M3
    ((_q, 8900, 9000, "-", _t, 28900, 29000, 95, 100), (S, 1)),        # M4
    ((_q, 8899, 8999, "-", _t, 28899, 28999, 95, 100), (D, 0)),        # M5
    ((_q, 21000, 21100, "-", _t, 41000, 41100, 95, 100), (S, 1)),      # M6
    ((_q, 21001, 21101, "-", _t, 41001, 41101, 95, 100), (D, 0)),      # M7
], "paf_filter.rs:557, :574-580, :585-592, :664-672"))

# ------------------------------------------------------------------------------------------------------------------
# 3. Members of a chain the scaffold sweep removed are never rescued (paf_filter.rs:601-604, :675-678).
#    G = 1000, min_scaffold_length 2000, scaffold filter 1:1 (scaffold_overlap_threshold 0.5), max_deviation 5000.
#      X1 '+' q[0,3000)     t[0,3000)      2970/3000
#      Y1 '+' q[500,2900)   t[3500,5900)   2160/2400
#      Z  '+' q[3100,3300)  t[5000,5200)   190/200
#    Chaining (sorted X1, Y1, Z): X1->Y1 query overlap 2500 > 200: reject.  X1->Z: q_gap 100, r_gap 5000-3000 = 2000 > G:
#    reject.  Y1->Z: q_gap 200, target overlap 5900-5000 = 900 > 200: reject.  Three singleton chains.
#    Length filter: X (3000) and Y (2400) stay, Z (200) goes.  pre_sweep_scaffold_members = {X1, Y1} (:470-476).
#    Scaffold sweep on the pair, 1:1: score = weighted_identity * ln(query span) (plane_sweep_exact.rs:76-86):
#    X 0.99 * ln 3000 = 7.93, Y 0.90 * ln 2400 = 7.00.  Y's query interval [500,2900) lies inside X's, so X is the top
#    at every point of Y: Y is never top -> removed; X kept -> chain_1.  filtered_scaffold_members = {Y1}.
#    Rescue (:680-737), anchor X1 with centres (1500,1500):
#      Y1 centres (1700,4700): distance floor(sqrt(200^2+3200^2)) = 3206 <= 5000 -- but Y1 is a filtered scaffold member:
#         `continue` (:675-678) -> dropped.
#      Z centres (3200,5100): |dq| 1700 <= 5000, floor(sqrt(1700^2+3600^2)) = floor(3981.2) = 3981 <= 5000 -> rescued, chain_1.
_cfg3 = dict(scaffold_gap=1000, min_scaffold_length=2000, scaffold_filter_mode="OneToOne", scaffold_overlap_threshold=0.5,
             scaffold_max_deviation=5000)
CASES.append(_c("filtered_scaffold_members_not_rescued", _cfg3, [
    ((_q, 500, 2900, "+", _t, 3500, 5900, 2160, 2400), (D, 0)),   # Y1
    ((_q, 3100, 3300, "+", _t, 5000, 5200, 190, 200), (R, 1)),    # Z
    ((_q, 0, 3000, "+", _t, 0, 3000, 2970, 3000), (S, 1)),        # X1
], "paf_filter.rs:470-476, :601-604, :675-678, :695-737"))

# ------------------------------------------------------------------------------------------------------------------
# 4. Rescue distance: exactly D, D + 1, the truncation of the square root, the |dq| shortcut and integer centres
#    (paf_filter.rs:681-737).  G = 1000, min_scaffold_length 2000, scaffold_max_deviation D = 1000.
#    Anchor X1 '+' q[0,3000) t[0,3000), centres (1500,1500), chain_1.  The candidates lie inside X1's query range, so
#    X1 -> candidate has a query overlap > 200 and never chains with it; the candidates chain at most with each other into
#    chains spanning < 2000, which the length filter removes (they are no scaffold members).
#      R1 q[2050,2150) c 2100, t[2250,2350) c 2300: (600,800)  -> sqrt(1,000,000) = 1000      <= D: rescued
#      R2 q[2060,2140) c 2100, t[2252,2352) c 2302: (600,802)  -> sqrt(1,003,204) = 1001.6 -> 1001 > D: dropped
#      R3 q[1731,1831) c 1781, t[2410,2510) c 2460: (281,960)  -> sqrt(1,000,561) = 1000.28 -> 1000 <= D: rescued
#      R4 q[2451,2551) c 2501, t[1450,1550) c 1500: |dq| = 1001 > D -> skipped before the distance (:699-701): dropped
#      R5 q[2450,2551) c (2450+2551)/2 = 2500 (integer division, :686), t[1450,1550) c 1500: (1000,0) -> 1000 <= D: rescued
#         (with a rounded-up centre 2501 it would be R4's case)
_cfg4 = dict(scaffold_gap=1000, min_scaffold_length=2000, scaffold_max_deviation=1000)
CASES.append(_c("rescue_distance_boundaries", _cfg4, [
    ((_q, 2050, 2150, "+", _t, 2250, 2350, 95, 100), (R, 1)),     # R1
    ((_q, 2060, 2140, "+", _t, 2252, 2352, 76, 80), (D, 0)),      # R2
    ((_q, 0, 3000, "+", _t, 0, 3000, 2970, 3000), (S, 1)),        # X1
    ((_q, 1731, 1831, "+", _t, 2410, 2510, 95, 100), (R, 1)),     # R3
    ((_q, 2451, 2551, "+", _t, 1450, 1550, 95, 100), (D, 0)),     # R4
    ((_q, 2450, 2551, "+", _t, 1450, 1550, 96, 101), (R, 1)),     # R5
], "paf_filter.rs:686-687, :695-701, :706-718"))

# ------------------------------------------------------------------------------------------------------------------
# 5a. Chain numbering over two genome pairs x two chromosome pairs (paf_filter.rs:517-521 over
#     plane_sweep_scaffold.rs:204-251; metadata order from paf_filter.rs:1037-1046, 1105-1111; groups :761-770; chains of
#     a group by sorted position :777, union_find.rs:52-63).  G = 1000, min_scaffold_length 5000, every line its own chain.
#       l0 A#1#c1 q[50000,60000) -> B#1#c1      l1 A#1#c1 -> C#1#c1      l2 A#1#c2 -> B#1#c2
#       l3 A#1#c1 q[0,10000)     -> B#1#c1      l4 A#1#c2 -> C#1#c1
#     Mapping stage regroups by genome pair (prefix up to the last '#') in first-appearance order: (A,B) = [l0,l2,l3],
#     (A,C) = [l1,l4] -> metadata order l0 l2 l3 l1 l4.  (q,t,strand) groups in that order: (c1,Bc1) = {l0,l3},
#     (c2,Bc2) = {l2}, (c1,Cc1) = {l1}, (c2,Cc1) = {l4}.  Inside (c1,Bc1) the chains come in q_start order: l3 first.
#     all_chains = [l3, l0, l2, l1, l4]; the many:many scaffold sweep keeps all and returns them genome pair by genome
#     pair, chromosome pair by chromosome pair = the same order -> chain_1..5 in that order.
_cfg5 = dict(scaffold_gap=1000, min_scaffold_length=5000, scaffold_max_deviation=0)
CASES.append(_c("chain_numbering_genome_and_chromosome_pairs", _cfg5, [
    (("A#1#c1", 50000, 60000, "+", "B#1#c1", 50000, 60000, 9500, 10000), (S, 2)),   # l0
    (("A#1#c1", 0, 10000, "+", "C#1#c1", 0, 10000, 9500, 10000), (S, 4)),           # l1
    (("A#1#c2", 0, 10000, "+", "B#1#c2", 0, 10000, 9500, 10000), (S, 3)),           # l2
    (("A#1#c1", 0, 10000, "+", "B#1#c1", 0, 10000, 9500, 10000), (S, 1)),           # l3
    (("A#1#c2", 0, 10000, "+", "C#1#c1", 20000, 30000, 9500, 10000), (S, 5)),       # l4
], "paf_filter.rs:1037-1046, :761-777, :517-521; plane_sweep_scaffold.rs:204-251"))

# 5b. The two prefix rules disagree on four-part names: the mapping stage groups by the prefix up to the LAST '#'
#     (paf_filter.rs:1022-1030: "A#1#x#" != "A#1#y#"), the scaffold sweep by the first two parts
#     (plane_sweep_scaffold.rs:13-22: both "A#1#").
#       l0 A#1#x#c1 -> B#1#c1    l1 A#1#x#c1 -> C#1#c1    l2 A#1#y#c1 -> B#1#c1
#     Mapping-stage groups in first appearance: (A#1#x#,B#1#) = [l0], (A#1#x#,C#1#) = [l1], (A#1#y#,B#1#) = [l2]:
#     all_chains = [l0, l1, l2].  Scaffold sweep: genome pairs (A#1#,B#1#) = {chain 0, chain 2}, (A#1#,C#1#) = {chain 1}
#     -> kept order [0, 2, 1] -> l0 chain_1, l2 chain_2, l1 chain_3.
CASES.append(_c("chain_numbering_two_prefix_rules", _cfg5, [
    (("A#1#x#c1", 0, 10000, "+", "B#1#c1", 0, 10000, 9500, 10000), (S, 1)),         # l0
    (("A#1#x#c1", 0, 10000, "+", "C#1#c1", 0, 10000, 9500, 10000), (S, 3)),         # l1
    (("A#1#y#c1", 0, 10000, "+", "B#1#c1", 0, 10000, 9500, 10000), (S, 2)),         # l2
], "paf_filter.rs:1022-1030 vs plane_sweep_scaffold.rs:13-22, :116-130"))


def paf_text(case):
    return "".join("\t".join([q, "1000000", str(qs), str(qe), st, t, "1000000", str(ts), str(te), str(m), str(b), "60"]) + "\n"
                   for (q, qs, qe, st, t, ts, te, m, b) in case["lines"])


def expected_output(case):
    """write_filtered_output (paf_filter.rs:1689-1726): kept lines in input order + ch:Z: + st:Z:."""
    out = []
    for line, (st, ch) in zip(paf_text(case).splitlines(), case["expect"]):
        if st == D:
            continue
        out.append(line + (f"\tch:Z:chain_{ch}" if ch else "") + f"\tst:Z:{st}\n")
    return "".join(out)


# ---- inputs of the reference's binary-invoking tests, replayed through both command lines -------------------------------
# (tests/test_chain_monotonicity.rs:19-90, 230-262; tests/test_centromere_plane_sweep.rs:29-33, 92-96).  `count` = the
# number of output lines the reference test asserts (None where it only asserts presence).
def _collinear():
    rows = [(0, 1000), (2000, 3000), (8000, 9000), (20000, 21000), (50000, 51000)]
    return "".join(f"query\t100000\t{a}\t{b}\t+\ttarget\t100000\t{a}\t{b}\t950\t1000\t60\tNM:i:50\tcg:Z:950=50X\n" for a, b in rows)


def _mixed():
    hi = [(0, 1000), (2000, 3000), (5000, 6000), (8000, 9000), (11000, 12000)]
    lo = [(50000, 51000), (80000, 81000), (120000, 121000), (160000, 161000), (195000, 196000)]
    t = "".join(f"query\t200000\t{a}\t{b}\t+\ttarget\t200000\t{a}\t{b}\t980\t1000\t60\tNM:i:20\tcg:Z:980=20X\n" for a, b in hi)
    return t + "".join(f"query\t200000\t{a}\t{b}\t+\ttarget\t200000\t{a}\t{b}\t900\t1000\t60\tNM:i:100\tcg:Z:900=100X\n" for a, b in lo)


def _fragmented():
    out = []
    for i in range(20):
        qs = i * 3000
        m = 950 + (i % 3) * 10
        out.append(f"query\t100000\t{qs}\t{qs + 1000}\t+\ttarget\t100000\t{qs}\t{qs + 1000}\t{m}\t1000\t60\tNM:i:{1000 - m}\tcg:Z:{m}={1000 - m}X\n")
    return "".join(out)


_CENTRO3 = "".join(f"query\t200000000\t{q}\t{q + 1000000}\t-\ttarget\t200000000\t{t}\t{t + 1000000}\t760000\t1000000\t60\tNM:i:240000\tcg:Z:760000=240000X\n"
                   for q, t in [(129000000, 132000000), (130000000, 133000000), (131000000, 134000000)])
_CENTRO_A = ("query\t250000000\t129142789\t132986703\t+\ttarget\t250000000\t129142789\t132986703\t2938926\t3843914\t60\tNM:i:904988\tcg:Z:2938926=904988X\n"
             "query\t250000000\t129213003\t137240549\t-\ttarget\t250000000\t131937578\t139967018\t6372479\t8027546\t60\tNM:i:1655067\tcg:Z:6372479=1655067X\n")
_CENTRO_B = ("query\t100000000\t10000000\t11000000\t+\ttarget\t100000000\t10000000\t11000000\t950000\t1000000\t60\tNM:i:50000\tcg:Z:950000=50000X\n"
             "query\t100000000\t10000000\t12000000\t-\ttarget\t100000000\t20000000\t22000000\t1900000\t2000000\t60\tNM:i:100000\tcg:Z:1900000=100000X\n")

REPLAY = []
for _g in (2_000, 10_000, 30_000, 100_000):   # test_chain_monotonicity.rs:127-166
    REPLAY.append(dict(name=f"collinear_j{_g}", paf=_collinear(),
                       flags=["--scaffold-jump", str(_g), "--min-aln-identity", "0.90", "--scaffold-mass", "0"], count=5))
for _g, _y, _n in ((10_000, "0.95", 5), (100_000, "0.95", 0), (10_000, "0.85", 10), (100_000, "0.85", 10)):   # :170-212
    REPLAY.append(dict(name=f"mixed_j{_g}_Y{_y}", paf=_mixed(),
                       flags=["--scaffold-jump", str(_g), "--min-scaffold-identity", _y, "--scaffold-mass", "0"], count=_n))
for _g in (5_000, 50_000, 500_000):   # :216-257
    REPLAY.append(dict(name=f"fragmented_j{_g}", paf=_fragmented(),
                       flags=["--scaffold-jump", str(_g), "--min-aln-identity", "0.90", "--scaffold-mass", "0"], count=20))
REPLAY.append(dict(name="centromere_Y0.80", paf=_CENTRO3,   # :262-302: the reference passes the threshold as --min-aln-identity
                   flags=["--min-aln-identity", "0.80", "--scaffold-jump", "10000", "--scaffold-mass", "0"], count=0))
REPLAY.append(dict(name="centromere_Y0.75", paf=_CENTRO3,
                   flags=["--min-aln-identity", "0.75", "--scaffold-jump", "10000", "--scaffold-mass", "0"], count=3))
REPLAY.append(dict(name="reverse_scaffold_8mb", paf=_CENTRO_A,    # test_centromere_plane_sweep.rs:20-86
                   flags=["--min-aln-identity", "0", "--scaffold-jump", "100000"], count=None, must_contain="\t-\t"))
REPLAY.append(dict(name="reverse_vs_forward", paf=_CENTRO_B,      # test_centromere_plane_sweep.rs:88-133
                   flags=["--min-aln-identity", "0", "--scaffold-jump", "100000"], count=None, must_contain="\t-\t"))
