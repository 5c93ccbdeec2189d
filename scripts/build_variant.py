"""A/B builds: libhxv.so with extra -D flags on ONE source (default hxv_tiled.hip) -> gpurun_ab/libhxv_<tag>.so; run with HXV_LIB=<that file>.
usage: build_variant.py <tag> "-DHXV_NT_LOADS=1 ..." [source.hip]"""
import subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge
ge.build_engine()
tag, flags = sys.argv[1], sys.argv[2].split()
src = ge.CSRC / (sys.argv[3] if len(sys.argv) > 3 else "hxv_tiled.hip")
out = ROOT / "gpurun_ab"
out.mkdir(exist_ok=True)
obj = out / f"{src.name}.{tag}.o"
subprocess.check_call([ge.HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + flags + ["-c", "-o", str(obj), str(src)], stderr=subprocess.DEVNULL)
objs = [str(o) for o in (ge.LIB.parent / "obj").glob("*.o") if o.name != src.name + ".o"] + [str(obj)]
lib = out / f"libhxv_{tag}.so"
subprocess.check_call([ge.HIPCC, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", str(lib)] + objs + ["-ldl"])
print(lib)
