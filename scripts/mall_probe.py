"""Does the 256 MiB Infinity Cache absorb a write->read hand-off between two kernels?  Times (a) dst.copy_(src) repeated on the
same buffers and (b) a producer/consumer pair (y = x*2 ; z = y+1) for buffer sizes from 8 MiB to 2 GiB: effective GB/s vs size."""
import torch, time
torch.cuda.init()
def bench(fn, nrep=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(nrep): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / nrep
for mb in (8, 16, 32, 64, 96, 128, 192, 256, 512, 1024, 2048):
    n = mb * 1024 * 1024 // 8
    x = torch.randn(n, dtype=torch.float64, device="cuda"); y = torch.empty_like(x); z = torch.empty_like(x)
    t_copy = bench(lambda: y.copy_(x))
    def pc():
        torch.mul(x, 2.0, out=y); torch.add(y, 1.0, out=z)
    t_pc = bench(pc)
    t_rd = bench(lambda: torch.sum(x))
    print(f"{mb:5d} MiB  copy {2*mb/1024/t_copy*1000:7.0f} GiB/s   producer->consumer (4 x size moved) {4*mb/1024/t_pc*1000:7.0f} GiB/s   sum(read only) {mb/1024/t_rd*1000:7.0f} GiB/s", flush=True)
    del x, y, z
