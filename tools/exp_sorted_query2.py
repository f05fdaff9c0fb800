#!/usr/bin/env python3
"""Experiment: k_query time vs sort granularity (queries ordered by the top t bits of their minimizer)."""
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
torch.cuda.synchronize()
ix = blight_amd.kmer_Set_Light(k, m, n, min(n, 6), 1, b)
ix.construct_index_from_device(ascii_.data_ptr(), offs.data_ptr(), offs.numel() - 1, total)
q, is_pos = bench.synth_queries(torch, ix, dev, ascii_, offs, total, k, nq, 44, 1)
del ascii_
stream = torch.cuda.current_stream().cuda_stream
ids = torch.empty(nq, device=dev, dtype=torch.int64)

def timeit(qq, reps=10):
    n_ = qq.shape[0]
    for _ in range(2):
        ix.query_kmers_device(qq.data_ptr(), n_, ids.data_ptr(), stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ix.query_kmers_device(qq.data_ptr(), n_, ids.data_ptr(), stream)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

res = {"workload": wl, "nq": nq, "random_ms": timeit(q)}
g = torch.empty(nq, device=dev, dtype=torch.int32)
ix.kmer_groups_device(q.data_ptr(), nq, g.data_ptr(), stream)   # group = minimizer >> (2m-1-n): n bits
torch.cuda.synchronize()
gl = g.long()
for t in (4, 6, 8, 10, 12, n):
    key = gl >> (n - t)
    order = torch.argsort(key)          # not stable: within a bin the order stays random, as a one-pass binning would leave it
    res["top%d_bits_ms" % t] = timeit(q[order].contiguous())
    del order
print(json.dumps(res))
