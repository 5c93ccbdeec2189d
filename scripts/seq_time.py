"""What does pass A cost when it follows pass B?  (profiles/r03_ab_base_C3.log: alone 1.79 ms, inside the product 2.09 ms.)
Times every launch with events in several sequences: B A B A ..., A A A ..., B <idle> A, B <copy> A."""
import os, sys
os.environ.setdefault("HXV_EXPERIMENTS", "1")
sys.path.insert(0, "cdmft-lanc-ed_amd")
import torch, hxv
from hxv import models

m, (nup, ndw) = models.hm_2dsquare(Nbath=3), (8, 8)
sec = hxv.HxvSector.from_model(m, nup, ndw)
n = sec.fullElems
v = torch.randn(n, dtype=torch.complex128, device="cuda")
hv = torch.empty(sec.localElems, dtype=torch.complex128, device="cuda")
big = torch.empty(1 << 28, dtype=torch.float64, device="cuda")  # 2 GiB
big2 = torch.empty_like(big)
sec.apply_device(v, hv)
torch.cuda.synchronize()
st = torch.cuda.current_stream()

def run(seq, reps=6):
    evs = []
    for _ in range(reps):
        for what in seq:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if what in "AB":
                sec.set_option("passes", 1 if what == "A" else 2)
                e0.record(); sec.apply_device(v, hv); e1.record()
            elif what == "F":
                sec.set_option("passes", 3)
                e0.record(); sec.apply_device(v, hv); e1.record()
            elif what == "c":
                e0.record(); big2.copy_(big); e1.record()
            elif what == "s":
                e0.record(); torch.cuda._sleep(int(2.4e6 * 3)); e1.record()  # ~3 ms idle
            evs.append((what, e0, e1))
    torch.cuda.synchronize()
    acc = {}
    for i, (what, e0, e1) in enumerate(evs):
        if i < len(seq):
            continue
        acc.setdefault((i % len(seq), what), []).append(e0.elapsed_time(e1))
    return "  ".join(f"{w}{k}:{min(t):.3f}/{sum(t)/len(t):.3f}" for (k, w), t in sorted(acc.items()))

# the handle's stream must be torch's current stream for the events to see the kernels
for seq in ["BA", "AA", "BB", "BsA", "BcA", "AsB", "F", "BAA", "BBA"]:
    print(f"{seq:5s} (min/avg ms per slot)  {run(seq)}", flush=True)
sec.close()
