"""Static checks on the gfx950 assembly hipcc emits for the product kernels (no GPU needed: hipcc cross-compiles).

Why this exists.  Round 2 lost a kernel family to "wrong sums, once an abort" (fuzz seed 8).  The cause, found in round 3
in the ISA of the failing build (profiles/r03_rootcause_pass_up8_lz_exec0.txt): under a 64-VGPR budget with > 100 spilled
registers the register allocator split the live range of threadIdx.x and hipcc (ROCm 7.2.0, AMD clang 22) placed the copy

    v_mov_b32 v56, v8          ; v8 = threadIdx.x

in the exit block of a divergent loop ABOVE the `s_or_b64 exec, exec, s[..]` that re-enables the lanes.  A divergent loop
is left with EXEC == 0, so the copy wrote no lane, and v56 -- the thread index for the rest of the kernel -- held whatever
the previous wave had left in that register.  `exec0_findings` looks for exactly that window: vector instructions
between the fall-through of `s_andn2_b64 exec, exec, m ; s_cbranch_execnz loop` and the next write of EXEC.  Lane
operations (v_writelane / v_readlane / v_readfirstlane: how SGPR spills travel) ignore EXEC and are legitimate there.
"""
from __future__ import annotations

import re
import subprocess
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "cdmft-lanc-ed_amd" / "csrc"
HIPCC = "/opt/rocm/bin/hipcc"
KERNEL_SOURCES = ["hxv_tiled.hip", "hxv_jobs.hip", "hxv_kernels.hip", "hxv_lanczos.hip", "hxv_eigh.hip"]
LANE_OPS = ("v_writelane", "v_readlane", "v_readfirstlane")


def compile_to_asm(src: Path, out: Path) -> str:
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", str(out), str(src)],
                          stderr=subprocess.DEVNULL)
    return out.read_text()


def compile_all(tmp: Path) -> dict:
    """{source name: assembly text} for every kernel source of the product library (compiled in parallel)."""
    with ThreadPoolExecutor(max_workers=4) as ex:
        futs = {s: ex.submit(compile_to_asm, CSRC / s, tmp / (s + ".s")) for s in KERNEL_SOURCES}
        return {s: f.result() for s, f in futs.items()}


def kernel_bodies(txt: str):
    """(mangled name, body) of every function in an assembly file."""
    for m in re.finditer(r"^(_Z\S+|[A-Za-z_]\w*):[^\n]*\n(.*?)^\.Lfunc_end\d+:", txt, re.S | re.M):
        yield m.group(1), m.group(2)


def exec0_findings(body: str):
    """Vector instructions that run with EXEC == 0: after the fall-through of a divergent loop's back edge, before EXEC is written."""
    lines = [re.sub(r";.*", "", l).strip() for l in body.splitlines()]
    lines = [l for l in lines if l and (not l.startswith(".") or l.startswith(".LBB"))]
    bad = []
    for i, l in enumerate(lines):
        if not (l.startswith("s_cbranch_execnz") and i > 0 and re.match(r"s_andn2_b64 exec, exec,", lines[i - 1])):
            continue
        for j in range(i + 1, len(lines)):
            x = lines[j]
            if re.match(r"s_\w+ exec\b", x) or "saveexec" in x:
                break  # EXEC restored
            if x.startswith(("s_branch", "s_cbranch", "s_endpgm", "s_setpc")):
                break  # leaves the block: not followed further (conservative towards silence, the bug class sits right at the exit)
            op = x.split()[0]
            if (op.startswith("v_") and not op.startswith(LANE_OPS)) or op.startswith(("ds_", "global_", "scratch_", "buffer_", "flat_")):
                bad.append(x)
    return bad


def kernel_metadata(txt: str) -> dict:
    """{mangled kernel name: {vgpr_spill_count, sgpr_spill_count, private_segment_fixed_size, group_segment_fixed_size, ...}} from .amdgpu_metadata."""
    out = {}
    meta = txt[txt.find("amdhsa.kernels:"):]
    for blk in meta.split("  - .agpr_count:")[1:]:
        nm = re.search(r"\.name:\s+(\S+)", blk)
        if not nm:
            continue
        d = {}
        for k in ("sgpr_count", "sgpr_spill_count", "vgpr_count", "vgpr_spill_count", "private_segment_fixed_size", "kernarg_segment_size",
                  "group_segment_fixed_size"):
            mm = re.search(r"\." + k + r":\s+(\d+)", blk)
            if mm:
                d[k] = int(mm.group(1))
        out[nm.group(1)] = d
    return out


def demangle(name: str) -> str:
    try:
        return re.sub(r"\(.*", "", subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip())
    except OSError:
        return name


if __name__ == "__main__":  # python tests/isa_lint.py [file.s ...]: lint assembly files (or the library's sources when none is given)
    import sys
    import tempfile

    if len(sys.argv) > 1:
        texts = {p: Path(p).read_text() for p in sys.argv[1:]}
    else:
        texts = compile_all(Path(tempfile.mkdtemp()))
    total = 0
    for src, txt in texts.items():
        n = 0
        for name, body in kernel_bodies(txt):
            n += 1
            for x in exec0_findings(body):
                print(f"{src}: {demangle(name)}: EXEC==0 window: {x}")
                total += 1
        print(f"{src}: {n} kernels")
    print("suspicious instructions:", total)
    sys.exit(1 if total else 0)
