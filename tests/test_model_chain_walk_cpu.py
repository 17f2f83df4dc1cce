"""tools/model_chain_walk.py (the executable model of chain_walk_kernel's 64-elements-per-step logic) against the reference's
sequential greedy (src/paf_filter.rs:784-851): a slice of its random cases in the CPU suite."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_batched_walk_model_equals_sequential_greedy():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "model_chain_walk.py"), "2500"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "equal to the sequential greedy" in r.stdout and "whole-window passes" in r.stdout
    # all three work-list rules (round 3's, and round 4's two shorter lists) are modelled and agree
    for rule in ("count", "low", "pair"):
        assert any(line.startswith(rule) and "equal to the sequential greedy" in line for line in r.stdout.splitlines()), r.stdout
