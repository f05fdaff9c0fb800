#!/usr/bin/env python3
"""Measurements for DESIGN.md: (1) k=63 workload (C5-like, smaller), (2) file_query-style sequence self-query throughput
through the host-buffer API (PCIe inclusive)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import blight_amd, bench
dev = torch.device("cuda", 0)
out = {}
# (2) sequence self-query on C3-scale unitigs: query_sequences over the first 2e5 unitigs (host buffers)
k, m, n, b = 31, 9, 7, 6
ascii_, offs, total = bench.synth_unitigs(torch, dev, 230_000_000, k, 3)
torch.cuda.synchronize()
ix = blight_amd.kmer_Set_Light(k, m, n, 6, 1, b)
ix.construct_index_from_device(ascii_.data_ptr(), offs.data_ptr(), offs.numel() - 1, total)
nseq = 400_000
o = offs[:nseq + 1].cpu().numpy().astype(np.uint64)
raw = bytes(ascii_[: int(o[-1])].cpu().numpy())
import ctypes as C
lib = blight_amd.load_library()
nk = lib.blmx_sequence_kmer_count(ix._h, o.ctypes.data_as(C.POINTER(C.c_uint64)), nseq)
ids = np.empty(nk, np.int64)
for rep in range(3):
    t = time.time()
    rc = lib.blmx_query_sequences(ix._h, raw, o.ctypes.data_as(C.POINTER(C.c_uint64)), nseq, ids.ctypes.data_as(C.POINTER(C.c_int64)))
    dt = time.time() - t
assert rc == 0 and (ids >= 0).all()
out["seq_self_query"] = {"kmers": int(nk), "seconds": dt, "kmers_per_s": nk / dt, "note": "host buffers in/out (PCIe inclusive), C3 index"}
# host-buffer k-mer batch (PCIe inclusive)
q = np.ascontiguousarray(np.random.default_rng(1).integers(0, 1 << 62, 20_000_000, dtype=np.uint64))
for rep in range(3):
    t = time.time(); r = ix.query_kmer_hash(q); dt = time.time() - t
out["host_buffer_batch"] = {"queries": int(q.size), "seconds": dt, "queries_per_s": q.size / dt}
del ix, ascii_
torch.cuda.empty_cache()
print(json.dumps(out), flush=True)
