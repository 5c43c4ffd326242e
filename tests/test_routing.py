"""CPU: the routing of the library — which kernel family serves which (Dz, K, pass) — is DATA the tests hold the code to: ROUTING.md is
generated from the library's own router (mimo_plan_shape, host-only) and this test regenerates it; the BASELINE configurations and
the reference's own example shapes are pinned to their families by name."""
import os
import subprocess
import sys

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_routing_md_matches_the_library():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "routing_table.py"), "--check"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr


def test_baseline_and_example_shapes_keep_their_families():
    import routing_table as rt
    from mimo_amd import _lib
    lib = _lib.load()
    pins = [
        # (Dz, K, gibbs) -> family                      BASELINE.json configs / reference examples
        ((2, 4, False), "small"), ((2, 4, True), "small"),            # C1: examples/gmm/toy (2-D, K = 4)
        ((16, 64, False), "fused"),                                   # C2
        ((8, 256, True), "rowwave"),                                  # C3: label kernel + slot-table statistics
        ((12, 64, False), "fused"),                                   # C4: joint rows of 8 + 4 columns
        ((32, 128, False), "two-stage"),                              # C5 shape
        ((2, 50, False), "narrow"), ((2, 50, True), "narrow"),        # examples/ilr/evaluate_sine.py:35 (50 experts over dx = dy = 1)
        ((2, 100, False), "narrow"),                                  # examples/ilr/evaluate_sinc.py:35
        ((20, 16, False), "mid"), ((32, 32, False), "mid"), ((28, 8, False), "mid"), ((12, 96, False), "mid"), ((24, 48, False), "mid"),
        ((16, 8, False), "narrow"), ((28, 4, False), "narrow"),
        ((8, 32, False), "rowwave-vi"), ((32, 128, True), "rowwave"), ((20, 96, True), "rowwave"),
        ((20, 16, True), "mid"), ((28, 48, True), "mid"), ((16, 32, True), "rowwave"), ((12, 112, False), "mid"), ((20, 96, False), "mid"),
    ]
    for (D, K, gibbs), kind in pins:
        assert rt.route(lib, D, K, gibbs)[0] == kind, (D, K, gibbs, rt.route(lib, D, K, gibbs))
    # the C3 sweep reads Z twice (label kernel + statistics) with the histogram counted in the label kernel; at small N the
    # round-2 statistics kernel runs
    kind, launches, zpasses, desc = rt.route(lib, 8, 256, True)
    assert launches == 2 and zpasses == 2 and "gibbs_rowwave_kernel" in desc and "label_stats_slots_kernel" in desc
    assert "label_stats_kernel" in rt.route(lib, 8, 256, True, N=100_000)[3] and "slots" not in rt.route(lib, 8, 256, True, N=100_000)[3]
    # one-pass label statistics from Dz = 17, streamed label kernel where Theta does not fit LDS
    assert "gibbs_stream_kernel" in rt.route(lib, 32, 128, True)[3] and "label_stats_sorted_kernel" in rt.route(lib, 32, 128, True)[3]
    # structured blocks: reduced maps on the same families
    assert rt.route(lib, 16, 64, False, structure=2)[0] == "narrow" and rt.route(lib, 32, 64, False, structure=1)[0] == "fused"
