"""Register budget of the gfx950 step kernels (compile-time facts: no GPU needed).  The step kernels run at four waves per
SIMD (1024-thread workgroups, one per CU, or two 512-thread ones): 128 VGPRs is the ceiling, and a spill of either kind
or any scratch use is a performance regression that no parity test would notice (27 scalar spills in the
PEER variant of the resident kernel)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not installed")
def test_step_kernels_fit_their_register_budget_without_spills():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py")], capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr
    rows = {}
    for ln in out.stdout.splitlines()[1:]:
        f = ln.split()
        name, (sgpr, vgpr, sspill, vspill, scratch, occ) = " ".join(f[:-6]), (int(v) for v in f[-6:])
        rows[name] = dict(sgpr=sgpr, vgpr=vgpr, sspill=sspill, vspill=vspill, scratch=scratch, occ=occ)
    step = {k: v for k, v in rows.items() if "fused_step_kernel" in k or "persistent_steps_kernel" in k}
    # three instantiations each: plain / force-only / peer, and plain / predicted / peer
    assert sum("fused_step_kernel" in k for k in step) == 3 and sum("persistent_steps_kernel" in k for k in step) == 3, rows
    for name, r in rows.items():
        assert r["sspill"] == 0 and r["vspill"] == 0 and r["scratch"] == 0, (name, r)
    # Register counts pinned per variant: twice in round 2 a change OUTSIDE the step loop moved the allocation of the whole
    # kernel and cost 4 % that no test saw.  A deliberate change of the kernels updates this table - together with a fresh
    # A/B measurement (tools/ab.py) of what it did to the step time.
    pinned = {"fused_step_kernel<false, 0, false>": (118, 80), "fused_step_kernel<true, 0, false>": (100, 43),
              "fused_step_kernel<false, 0, true>": (116, 86), "persistent_steps_kernel<false, false>": (110, 97),
              "persistent_steps_kernel<true, false>": (110, 99), "persistent_steps_kernel<false, true>": (122, 105)}
    for name, r in step.items():
        key = name.split("saa::")[-1]
        assert key in pinned, name
        assert (r["vgpr"], r["sgpr"]) == pinned[key], (name, r, pinned[key])
    for name, r in step.items():
        assert r["vgpr"] <= 128 and r["occ"] >= 4, (name, r)
        # scalar registers: 112 is the allocation band the grid sizing of the resident kernel assumes (saa_kernels.hip:
        # persistent_max_blocks, MI355X_MICROARCH.md "Residency")
        assert r["sgpr"] <= 112, (name, r)



@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not installed")
def test_predictor_kernels_keep_their_staging_in_registers():
    """The predictor's GEMM runs two workgroups of four waves per CU (256 registers per lane) and the H = 50 recurrence
    kernel keeps a gate row's weights in registers.  Three times during their writing an innocent-looking change left an
    array in scratch memory (an array of HIP float4 structs, an array inside a struct, a lambda called twice and therefore
    not inlined): one global load at a time, each waited for - parity intact, 2-4x slower."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py"), "--file=saa_predictor.hip"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    rows = {}
    for ln in out.stdout.splitlines()[1:]:
        f = ln.split()
        rows[" ".join(f[:-6])] = dict(zip(("sgpr", "vgpr", "sspill", "vspill", "scratch", "occ"), (int(v) for v in f[-6:])))
    gemm = {k: v for k, v in rows.items() if "gemm_nt_kernel" in k}
    lstm = {k: v for k, v in rows.items() if "lstm_recurrence_kernel" in k}
    assert len(gemm) == 2 and len(lstm) == 2, rows
    for name, r in rows.items():
        assert r["vspill"] == 0 and r["scratch"] == 0, (name, r)
    for name, r in gemm.items():
        assert r["vgpr"] <= 256 and r["occ"] == 2 and r["sspill"] == 0, (name, r)
    # (the recurrence kernel's thirteen argument pointers do not all fit the scalar file next to its loop state: some live
    # in lanes of a vector register between their uses - register to register, no memory traffic)
    for name, r in lstm.items():
        assert r["vgpr"] <= 256, (name, r)
