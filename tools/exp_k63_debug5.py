import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, blight_amd, bench
k, m, n, b, nk = 63, 9, 14, 8, 100_000_000
dev = torch.device("cuda", 0)
ascii_, offs, total = bench.synth_unitigs(torch, dev, nk, k, 4)
torch.cuda.synchronize()
ix = blight_amd.kmer_Set_Light(k, m, n, 6, 1, b)
ix.construct_index_from_device(ascii_.data_ptr(), offs.data_ptr(), offs.numel() - 1, total)
s = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev); g.manual_seed(7)
npos = 50_000_000
p = torch.randint(0, total, (npos,), device=dev, dtype=torch.int64, generator=g)
u = torch.searchsorted(offs, p, right=True) - 1
print("u range", int(u.min()), int(u.max()), "nseq", offs.numel() - 1, flush=True)
p2 = torch.minimum(p, offs[u + 1] - k)
print("p2 min", int(p2.min()), "clamped", int((p2 != p).sum()), "p2<offs[u]", int((p2 < offs[u]).sum()), flush=True)
pos_k = torch.empty((npos, 2), device=dev, dtype=torch.int64)
err = torch.zeros(1, device=dev, dtype=torch.int32)
ix.kmers_at_device(ascii_.data_ptr(), p2.data_ptr(), npos, pos_k.data_ptr(), err.data_ptr(), s)
ids = torch.empty(npos, device=dev, dtype=torch.int64)
ix.set_query_binning(-1)
ix.query_kmers_device(pos_k.data_ptr(), npos, ids.data_ptr(), s)
torch.cuda.synchronize()
f = (ids >= 0)
print("found", int(f.sum()), "err", int(err.item()), flush=True)
for c in range(0, npos, 1 << 22):
    print(c, int(f[c:c + (1 << 22)].sum()), flush=True)
