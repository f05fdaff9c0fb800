import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, blight_amd, bench
k, m, n, b, nk = [int(x) for x in sys.argv[1:6]]
seed = int(sys.argv[6]) if len(sys.argv) > 6 else 5
NQ = int(sys.argv[7]) if len(sys.argv) > 7 else 2_000_000
dev = torch.device("cuda", 0)
def P(*a): print(*a, flush=True)
ascii_, offs, total = bench.synth_unitigs(torch, dev, nk, k, seed)
torch.cuda.synchronize(); P("gen ok", total)
kb = 16 if k > 32 else 8
ix = blight_amd.kmer_Set_Light(k, m, n, 6, 1, b)
ix.construct_index_from_device(ascii_.data_ptr(), offs.data_ptr(), offs.numel() - 1, total)
torch.cuda.synchronize(); P("build ok", {x: ix.stats()[x] for x in ("number_kmer", "largest_bucket", "largest_mphf", "max_levels_used", "fallback_keys", "total_nt", "positions_total_size")})
nq = NQ
q, is_pos = bench.synth_queries(torch, ix, dev, ascii_, offs, total, k, nq, 7, kb // 8)
P("queries ok")
s = torch.cuda.current_stream().cuda_stream
outs = []
for pol in (-1, -1, 1, -1):
    ids = torch.full((nq,), -7, device=dev, dtype=torch.int64)
    ix.set_query_binning(pol)
    ix.query_kmers_device(q.data_ptr(), nq, ids.data_ptr(), s)
    torch.cuda.synchronize()
    P("pol", pol, "found", int(((ids >= 0) & is_pos).sum().item()), "untouched", int((ids == -7).sum().item()), "neg found", int(((ids >= 0) & ~is_pos).sum().item()))
    outs.append(ids)
P("direct==direct", bool(torch.equal(outs[0], outs[1])), "direct==binned", bool(torch.equal(outs[0], outs[2])))
