import os, sys, time
sys.path.insert(0, "/root/repo/multimodal-sentiment-analysis_amd")
import torch, torch.distributed as dist
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1, device_id=torch.device("cuda:0"))
from mmsa_amd import parallel
dev = torch.device("cuda:0")
table = torch.randn(30522, 256, device=dev)
ids = torch.randint(0, 30522, (4096,), device=dev)
ex = parallel.SparseRowExchange(4096, 256, 30522, dev)
def timed(name, fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); s.record()
    for _ in range(n): fn()
    e.record(); t1 = time.perf_counter(); torch.cuda.synchronize()
    print(f"{name:10s} host {1e6*(t1-t0)/n:8.1f} us   gpu {1e3*s.elapsed_time(e)/n:8.1f} us", flush=True)
timed("pack", lambda: ex.pack(table, ids))
hs = []
timed("start", lambda: hs.append(ex.start()))
timed("finish", lambda: ex.finish(table, ex.start()))
timed("mask", lambda: parallel.first_occurrence_mask(ids))
timed("zero", lambda: table.zero_())
timed("index_add", lambda: table.index_add_(0, ex.ids_all[0], ex.rows_all[0]))
timed("index_sel", lambda: torch.index_select(table, 0, ids, out=ex.rows))
dist.destroy_process_group()
