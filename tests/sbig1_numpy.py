"""BASELINE.json configs[2] (S-big1) from numpy's PCG64 -- the same bytes on any machine with this numpy --
so that the oracle's answer for the FULL-SIZE record set can be computed once on a CPU box (tools/make_sbig1_golden.py,
~25 min per scaffold flag set) and committed as a fingerprint (tests/golden/sbig1_full_size.json), and the GPU test then
checks 10^7 records against it in seconds.  One pair hgA#1#chr1 -> hgB#1#chr1, 248,956,422 bp, 70 % syntenic
(t = q + N(0, 50 kb)), 30 % repeats, lengths lognormal(median 2 kb, sigma 1.2) in [100, 500 k], identity 0.70 + 0.30
Beta(5, 1.5), 10 % '-' strand."""
import hashlib

import numpy as np

CHR_LEN = 248_956_422
NAMES = ["hgA#1#chr1", "hgB#1#chr1"]


def gen(n, seed=1234, chr_len=CHR_LEN):
    rng = np.random.Generator(np.random.PCG64(seed))
    ln = np.exp(7.6009 + 1.2 * rng.standard_normal(n)).clip(100, 500_000).astype(np.int64)
    ln = np.minimum(ln, max(chr_len // 2, 100))
    room = (chr_len - ln).astype(np.float64)
    qs = (rng.random(n) * room).astype(np.int64)
    syn = rng.random(n) < 0.7
    ts_syn = np.minimum(np.maximum(qs + 50_000.0 * rng.standard_normal(n), 0.0), room).astype(np.int64)
    ts_rep = (rng.random(n) * room).astype(np.int64)
    ts = np.where(syn, ts_syn, ts_rep)
    ident = 0.70 + 0.30 * rng.beta(5.0, 1.5, n)
    matches = np.floor(ident * ln).astype(np.int64)
    identity = matches / ln
    strand = (rng.random(n) < 0.1).astype(np.uint8)
    u32 = lambda a: np.ascontiguousarray(a.astype(np.uint32))
    return dict(q_id=np.zeros(n, np.uint32), t_id=np.ones(n, np.uint32), q_start=u32(qs), q_end=u32(qs + ln), t_start=u32(ts),
                t_end=u32(ts + ln), identity=np.ascontiguousarray(identity), matches=u32(matches), block_len=u32(ln), strand=strand)


def fingerprint(status, chain):
    """What is committed instead of 5 x 10^7 bytes: hashes of the two result columns + a few counts."""
    status = np.ascontiguousarray(status, dtype=np.uint8)
    chain = np.ascontiguousarray(chain, dtype=np.uint32)
    return dict(n=int(len(status)), kept=int((status != 0).sum()), scaffold=int((status == 1).sum()), rescued=int((status == 2).sum()),
                max_chain=int(chain.max()) if len(chain) else 0, chain_sum=int(chain.astype(np.uint64).sum()),
                status_sha256=hashlib.sha256(status.tobytes()).hexdigest(), chain_sha256=hashlib.sha256(chain.tobytes()).hexdigest())
