"""Register budget of the compiled product kernels (no GPU needed: hipcc cross-compiles).

A kernel family that spilled 30-140 vector registers (eight columns per tile on complex vectors) once came out of the
compiler computing wrong sums after an unrelated change; such variants are not built any more, and this test keeps it that
way: no kernel of the product path may spill more than a handful of vector registers."""
import re
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "cdmft-lanc-ed_amd" / "csrc"
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.parametrize("src", ["hxv_tiled.hip", "hxv_jobs.hip"])
def test_no_kernel_spills_heavily(tmp_path, src):
    if not Path(HIPCC).exists():
        pytest.skip("hipcc not available")
    out = tmp_path / (src + ".s")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", str(out), str(CSRC / src)],
                          stderr=subprocess.DEVNULL)
    txt = out.read_text()
    worst = []
    n = 0
    for m in re.finditer(r"\.name:\s+(\S+)", txt):
        blk = txt[max(0, m.start() - 1500): m.start() + 1500]
        vs = re.search(r"\.vgpr_spill_count:\s+(\d+)", blk)
        if not vs:
            continue
        n += 1
        if int(vs.group(1)) > 8:
            worst.append((m.group(1), int(vs.group(1))))
    assert n > 0
    assert not worst, worst
