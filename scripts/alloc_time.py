import time, torch
torch.cuda.init(); torch.cuda.synchronize()
for gb in (20, 80, 150):
    t0 = time.perf_counter(); x = torch.empty(gb * (1 << 30), dtype=torch.uint8, device="cuda"); torch.cuda.synchronize(); t1 = time.perf_counter()
    x.zero_(); torch.cuda.synchronize(); t2 = time.perf_counter()
    del x; torch.cuda.empty_cache(); torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"{gb} GB: malloc {1e3*(t1-t0):.1f} ms, memset {1e3*(t2-t1):.1f} ms, free {1e3*(t3-t2):.1f} ms", flush=True)
