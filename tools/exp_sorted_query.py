#!/usr/bin/env python3
"""Experiment (not part of the product): how fast is k_query when the batch is ordered for locality?
  random : the bench batch as is
  sorted : queries sorted by MPHF group
  xcd    : 8 interleaved streams (stream x = groups with g % 8 == x, sorted by group), 256-query chunks dealt round-robin, so
           that block b (observed to land on XCD b % 8) only touches groups of its own residue class
"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import blight_amd
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else "C4"
k, m, n, b, n_kmers, nq = bench.WORKLOADS[wl]
dev = torch.device("cuda", 0)
ascii_, offs, total = bench.synth_unitigs(torch, dev, n_kmers, k, 4)
torch.cuda.synchronize()  # the build runs on the library stream: inputs must be complete
ix = blight_amd.kmer_Set_Light(k, m, n, min(n, 6), 1, b)
ix.construct_index_from_device(ascii_.data_ptr(), offs.data_ptr(), offs.numel() - 1, total)
q, is_pos = bench.synth_queries(torch, ix, dev, ascii_, offs, total, k, nq, 44, 1)
del ascii_
stream = torch.cuda.current_stream().cuda_stream
ids = torch.empty(nq, device=dev, dtype=torch.int64)

def timeit(qq, reps=10):
    n_ = qq.shape[0]
    assert ids.numel() >= n_
    for _ in range(2):
        ix.query_kmers_device(qq.data_ptr(), n_, ids.data_ptr(), stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ix.query_kmers_device(qq.data_ptr(), n_, ids.data_ptr(), stream)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

res = {"workload": wl, "nq": nq}
res["random_ms"] = timeit(q)
g = torch.empty(nq, device=dev, dtype=torch.int32)
ix.kmer_groups_device(q.data_ptr(), nq, g.data_ptr(), stream)
torch.cuda.synchronize()
t0 = time.time(); order = torch.argsort(g.long()); torch.cuda.synchronize(); res["torch_argsort_ms"] = (time.time() - t0) * 1e3
qs = q[order].contiguous()
res["sorted_ms"] = timeit(qs)
# xcd-interleaved
gl = g.long()
key = (gl % 8) * (1 << 32) + gl // 8
order2 = torch.argsort(key)
qx = q[order2].reshape(-1)
cnt = torch.bincount((gl % 8), minlength=8).tolist()
chunks = max((c + 255) // 256 for c in cnt)
A = torch.empty((chunks, 8, 256), device=dev, dtype=torch.int64)
o = 0
for x in range(8):
    seg = qx[o:o + cnt[x]]
    pad = chunks * 256 - cnt[x]
    if pad:
        seg = torch.cat([seg, seg[-1:].expand(pad)])
    A[:, x, :] = seg.reshape(chunks, 256)
    o += cnt[x]
A = A.reshape(-1, 1).contiguous()
ids = torch.empty(A.shape[0], device=dev, dtype=torch.int64)  # streams are unequal: size the output for the padded batch
res["xcd_ms"] = timeit(A)
res["xcd_padded_nq"] = A.shape[0]
print(json.dumps(res))
