"""CPU: static check of the gfx950 ISA hipcc emits for the kernels (no GPU needed: hipcc cross-compiles).

Every s_barrier must be reached with the wave's own LDS loads / stores drained (s_waitcnt lgkmcnt(0) on every
path).  hipcc omits that wait where LLVM's memory model deems LDS traffic totally ordered across waves; on gfx950
with two workgroups per CU it is not, and the fused kernels read stale z-tile rows (DESIGN.md section 4,
wg_sync in mimo_kernels.hip).  tools/check_barrier_waits.py is the dataflow; this test keeps it green."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


SOURCES = [("mimo_kernels.hip", 100), ("mimo_small.hip", 60), ("mimo_rowwave.hip", 30), ("mimo_wide.hip", 10), ("mimo_narrow.hip", 100), ("mimo_narrow_table.hip", 100), ("mimo_narrow_grouped.hip", 50), ("mimo_narrow_big.hip", 30), ("mimo_mid.hip", 30), ("mimo_predict.hip", 5)]


@pytest.fixture(scope="module")
def assemblies(tmp_path_factory):
    """hipcc -S of every kernel file, all started together (the tile kernels alone take minutes): {source: (Popen, asm path)}."""
    out = tmp_path_factory.mktemp("isa")
    procs = {}
    for source, _ in SOURCES:
        asm = str(out / (source + ".s"))
        cmd = [HIPCC, "--offload-arch=gfx950", "--cuda-device-only", "-O3", "-std=c++17", "-fno-honor-nans",
               "-Wno-unused-function", "-S", "-o", asm, os.path.join(ROOT, "mimo_amd", "csrc", source)]
        procs[source] = (subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True), asm)
    yield procs
    for p, _ in procs.values():
        if p.poll() is None:
            p.kill()


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("source,at_least", SOURCES)
def test_every_barrier_is_reached_with_lds_drained(assemblies, source, at_least):
    proc, asm = assemblies[source]
    _, err = proc.communicate()
    assert proc.returncode == 0, err[-2000:]
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_barrier_waits as cbw
    kernels = cbw.kernels_of(asm)
    assert len(kernels) >= at_least, "kernel symbols not found in the assembly"
    bad = {k: cbw.check(L) for k, L in kernels.items()}
    bad = {k: v for k, v in bad.items() if v}
    assert not bad, f"{len(bad)} kernels reach an s_barrier with LDS operations pending, e.g. {list(bad.items())[:3]}"
    # the checker itself: a store right before a barrier is caught, a drained one is not
    assert cbw.check(["k:", "\tds_write_b64 v1, v[2:3]", "\ts_barrier", "\ts_endpgm"]) == [2]
    assert cbw.check(["k:", "\tds_write_b64 v1, v[2:3]", "\ts_waitcnt lgkmcnt(0)", "\ts_barrier", "\ts_endpgm"]) == []
    assert cbw.check(["k:", "\tds_write_b64 v1, v[2:3]", "\ts_cbranch_scc1 .LBB0_2", ".LBB0_1:", "\ts_waitcnt lgkmcnt(0)",
                      ".LBB0_2:", "\ts_barrier", "\ts_endpgm"]) == [6]
