"""ANI pre-pass for `aniN` identity thresholds (src/main.rs:296-688, src/cli.rs:76-130) over the C ABI.

Host threads parse the ANI view of the PAF, the GPU sorts / cuts / sums (csrc/swg_ani.hip); the ORTHOGONAL
method runs its fixed 1:1 filter through swg_filter.  No CPU evaluation here."""
import ctypes as C
import enum
from typing import NamedTuple, Optional

from ._lib import SWG_OK, SwgError, default_context, load
from .paf import PafFile


class AniMethodKind(enum.IntEnum):  # src/main.rs:174-178
    All = 0
    Orthogonal = 1
    NPercentile = 2


class NSort(enum.IntEnum):  # src/main.rs:181-187
    Length = 0
    Identity = 1
    Score = 2


class AniMethod(NamedTuple):
    kind: AniMethodKind
    percentile: float = 50.0
    sort: NSort = NSort.Identity


DEFAULT_ANI_METHOD = AniMethod(AniMethodKind.NPercentile, 50.0, NSort.Identity)  # main.rs:3578 (fallback for unknown strings)


def parse_ani_method(s: str) -> Optional[AniMethod]:
    """src/main.rs:296-330"""
    k, p, so = C.c_int(), C.c_double(50.0), C.c_int(1)
    if not load().swg_parse_ani_method(s.encode(), C.byref(k), C.byref(p), C.byref(so)):
        return None
    return AniMethod(AniMethodKind(k.value), p.value, NSort(so.value))


def parse_identity_value(value: str, ani_percentile: Optional[float] = None) -> float:
    """src/cli.rs:76-130; raises ValueError where the reference returns Err."""
    lib = load()
    o = C.c_double()
    rc = lib.swg_parse_identity_value(value.encode(), -1.0 if ani_percentile is None else float(ani_percentile), C.byref(o))
    if rc != SWG_OK:
        raise ValueError((lib.swg_paf_last_error() or b"").decode())
    return o.value


def calculate_ani_stats(paf, method: AniMethod = DEFAULT_ANI_METHOD, ctx=None, threads=0) -> float:
    """src/main.rs:334-688: median per-genome-pair ANI of a PAF (path or open PafFile)."""
    ctx = ctx or default_context()
    own = not isinstance(paf, PafFile)
    pf = PafFile(paf, threads=threads) if own else paf
    try:
        o = C.c_double()
        rc = ctx.lib.swg_paf_ani_stats(ctx.handle, pf.handle, int(method.kind), float(method.percentile), int(method.sort),
                                       int(threads), C.byref(o))
        if rc != SWG_OK:
            raise SwgError(rc, (ctx.lib.swg_paf_last_error() or b"").decode())
        return o.value
    finally:
        if own:
            pf.close()
