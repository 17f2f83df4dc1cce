"""`roofline.traffic` divides a launch label's PMC bytes by the HIP-event time of the SAME launches: the label the library
gives a launch (SWG_LAUNCH / SWG_LAUNCH_N in csrc/*.hip) must be what tools/pmc_traffic.py makes of the kernel's rocprof name.
This reads every launch site of the library's translation units and checks it (no GPU needed)."""
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import pmc_traffic  # noqa: E402

LAUNCH = re.compile(r'SWG_LAUNCH(?:_N)?\(ctx,\s*"([a-z0-9_]+)",(?:\s*[^,]+,)?\s*([A-Za-z0-9_]+)(<[^<>]*(?:<[^<>]*>)?[^<>]*>)?<<<')


def launches(path):
    text = open(path).read()
    return [(m.group(1), m.group(2), m.group(3) or "") for m in LAUNCH.finditer(text)]


def rocprof_name(kernel, targs):
    # template arguments as rocprofv3 prints them: constants folded, `true` / `false` spelled out
    args = targs.strip("<>")
    args = re.sub(r"\(int\)\s*", "", args)
    args = args.replace("PAIR_S_MAX", "1024").replace("PAIR_M_MAX", "4096").replace("PAIR_XL_MAX", "262144").replace("OUT_NT", "1024")
    return f"void swg_scaf::(anonymous namespace)::{kernel}<{args}>(args)" if args else f"swg_scaf::(anonymous namespace)::{kernel}(args)"


def test_every_launch_label_is_what_pmc_traffic_derives_from_the_kernel_name():
    seen = 0
    for path in sorted(glob.glob(os.path.join(ROOT, "sweepga_amd", "csrc", "*.hip"))):
        for label, kernel, targs in launches(path):
            derived = pmc_traffic.short_name(rocprof_name(kernel, targs))
            assert derived == label, (os.path.basename(path), label, kernel, targs, derived)
            seen += 1
    assert seen >= 200   # (the regular expression still finds the launch sites)
