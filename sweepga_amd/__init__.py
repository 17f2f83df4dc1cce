"""sweepga_amd: MI355X-native plane-sweep / scaffold filter (drop-in for sweepga's filter path).

The product is libsweepga_gpu.so (hand-written HIP for gfx950, C ABI in include/sweepga_gpu.h);
this package is the thin host-side mirror of the reference's filter interface over that ABI.
"""
from ._lib import Context, SwgError, K_INF, default_context, load  # noqa: F401
from .filter import (ChainStatus, FilterConfig, FilterMode, PafFilter, PlaneSweepMapping, RecordMeta,  # noqa: F401
                     ScoringFunction, SequenceIndex, pack_records, stream_plan, plane_sweep_both, plane_sweep_query,
                     plane_sweep_target, USIZE_MAX, UnionFind, merge_mappings_into_chains, plane_sweep_scaffolds)
from .paf import PafFile  # noqa: F401
from .aln import AlnRecords  # noqa: F401
from .alnstats import AlnStats  # noqa: F401
from .ani import (AniMethod, AniMethodKind, NSort, calculate_ani_stats, parse_ani_method,  # noqa: F401
                  parse_identity_value)

__version__ = "0.1.0"
