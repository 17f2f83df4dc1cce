"""Builds libsweepga_gpu.so (hand-written HIP for gfx950) in-tree with hipcc."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsweepga_gpu.so")
SOURCES = ["swg_context.hip", "swg_sort.hip", "swg_sweep.hip", "swg_filter.hip", "swg_chain.hip", "swg_chain_table.hip", "swg_scaffold_sweep.hip", "swg_scaffold.hip", "swg_pair.hip", "swg_segsort.hip",
           "swg_union_find.hip", "swg_ani.hip", "swg_shard.hip", "swg_stream.hip",
           os.path.join("host", "paf_io.cpp"), os.path.join("host", "tree_filter.cpp"),
           os.path.join("host", "alnstats.cpp")]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (looked at $HIPCC, /opt/rocm/bin/hipcc, PATH)")


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "sweepga_gpu.h")]
    deps += [os.path.join(CSRC, "host", f) for f in os.listdir(os.path.join(CSRC, "host"))]
    deps = [d for d in deps if not os.path.isdir(d)]
    if os.path.isdir(OBJ_DIR):   # objects of an interrupted build that were never linked
        deps += [os.path.join(OBJ_DIR, f) for f in os.listdir(OBJ_DIR) if f.endswith(".o")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.isfile(d))


CLI_SRC = os.path.join(CSRC, "host", "sweepga_gpu_cli.cpp")
CLI = os.path.join(HERE, "bin", "sweepga-gpu")


def build_cli(force=False, verbose=False):
    """The C++ host (reference-compatible command line) linked against libsweepga_gpu.so."""
    if not force and os.path.exists(CLI) and os.path.getmtime(CLI) >= max(os.path.getmtime(CLI_SRC), os.path.getmtime(LIB)):
        return CLI
    os.makedirs(os.path.dirname(CLI), exist_ok=True)
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-o", CLI, CLI_SRC, "-L", HERE, "-lsweepga_gpu",
           "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return CLI


STATS_SRC = os.path.join(CSRC, "host", "alnstats_cli.cpp")
STATS = os.path.join(HERE, "bin", "alnstats")


def build_alnstats(force=False, verbose=False):
    """The reference's second binary (src/bin/alnstats.rs) over the library's host code."""
    if not force and os.path.exists(STATS) and os.path.getmtime(STATS) >= max(os.path.getmtime(STATS_SRC), os.path.getmtime(LIB)):
        return STATS
    os.makedirs(os.path.dirname(STATS), exist_ok=True)
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-o", STATS, STATS_SRC, "-L", HERE, "-lsweepga_gpu",
           "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return STATS


SYNTH_SRC = os.path.join(CSRC, "host", "paf_synth.cpp")
SYNTH = os.path.join(HERE, "bin", "paf-synth")


def build_synth(force=False, verbose=False):
    """Synthetic PAF generator used by bench.py's end-to-end leg."""
    if not force and os.path.exists(SYNTH) and os.path.getmtime(SYNTH) >= os.path.getmtime(SYNTH_SRC):
        return SYNTH
    os.makedirs(os.path.dirname(SYNTH), exist_ok=True)
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-o", SYNTH, SYNTH_SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return SYNTH


SORT_BENCH_SRC = os.path.join(HERE, "..", "tests", "native", "sort_bench.cpp")
SORT_BENCH = os.path.join(HERE, "bin", "sort_bench")


def build_sort_bench(force=False, verbose=False):
    """tests/native/sort_bench.cpp: the library's radix sorts on their own, timed and verified element by element (a GPU box
    tool; tests/test_gpu_sort_native.py runs it)."""
    if not force and os.path.exists(SORT_BENCH) and os.path.getmtime(SORT_BENCH) >= max(os.path.getmtime(SORT_BENCH_SRC), os.path.getmtime(LIB)):
        return SORT_BENCH
    os.makedirs(os.path.dirname(SORT_BENCH), exist_ok=True)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O2", "-std=c++17", SORT_BENCH_SRC, "-o", SORT_BENCH, "-L", HERE, "-lsweepga_gpu",
           "-Wl,-rpath,$ORIGIN/.."]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return SORT_BENCH


OBJ_DIR = os.path.join(CSRC, "build")


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs += [os.path.join(CSRC, "host", f) for f in os.listdir(os.path.join(CSRC, "host")) if f.endswith(".h")]
    hs.append(os.path.join(HERE, "..", "include", "sweepga_gpu.h"))
    return hs


# SWG_DEFINES (environment): extra compiler flags of an experiment build, e.g. "-DSWG_SEG_TIMING" (part of the stamp: switching
# it recompiles every object)
COMPILE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + os.environ.get("SWG_DEFINES", "").split()
STAMP = os.path.join(OBJ_DIR, "compile.stamp")


def _stamp_text(hipcc):
    """What the cached objects were compiled with: compiler path, its version and the flags.  A different stamp invalidates
    every object (the mtime test alone would keep objects of another compiler or other flags)."""
    try:
        ver = subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout.strip()
    except OSError:
        ver = "?"
    return "\n".join([hipcc, ver, " ".join(COMPILE_FLAGS)]) + "\n"


def build_lib(force=False, verbose=False):
    """One object per translation unit (cached under csrc/build/, recompiled when the source or any header changed, or
    when the compiler / flags differ from the stamp), compiled side by side, then linked into libsweepga_gpu.so.  The
    link runs whenever the library is missing or OLDER than any object: an interrupted build (objects written, link not
    reached) is finished by the next call instead of leaving a stale library behind fresh objects."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()
    stamp = _stamp_text(hipcc)
    try:
        same_stamp = open(STAMP).read() == stamp
    except OSError:
        same_stamp = False
    if not same_stamp:
        force_objs = True
        if os.path.exists(STAMP):
            os.remove(STAMP)   # rewritten only after every object has been compiled with the new settings
    else:
        force_objs = force
    newest_header = max(os.path.getmtime(h) for h in _headers())
    jobs, objs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ_DIR, os.path.basename(s) + ".o")
        objs.append(obj)
        if force_objs or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), newest_header):
            jobs.append([hipcc] + COMPILE_FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as ex:
            list(ex.map(run, jobs))
    if not same_stamp:
        with open(STAMP, "w") as f:
            f.write(stamp)
    if jobs or not os.path.exists(LIB) or max(os.path.getmtime(o) for o in objs) > os.path.getmtime(LIB):
        run([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB + ".tmp"] + objs + ["-lz", "-lpthread"])
        os.replace(LIB + ".tmp", LIB)   # never a half-written library
    return LIB


def build(force=False, verbose=False):
    if force or stale():
        build_lib(force=force, verbose=verbose)
    build_cli(force=force, verbose=verbose)
    build_alnstats(force=force, verbose=verbose)
    build_synth(force=force, verbose=verbose)
    build_sort_bench(force=force, verbose=verbose)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
