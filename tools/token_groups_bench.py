import sys, os
sys.path.insert(0, "multimodal-sentiment-analysis_amd")
import torch
from mmsa_amd import ops, data
b = data.make_batch(0, 32)
ids = torch.from_numpy(b["input_ids"]).cuda()
groups = torch.empty(5 * ids.numel(), dtype=torch.int32, device="cuda")
table = torch.zeros(2 * 30522, dtype=torch.int32, device="cuda")
for _ in range(5): ops.token_groups(ids, 30522, 0, groups, table)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(50): ops.token_groups(ids, 30522, 0, groups, table)
e.record(); torch.cuda.synchronize()
print("token_groups B=32: %.1f us" % (s.elapsed_time(e) / 50 * 1e3), "table zero:", int(table.abs().sum()))
