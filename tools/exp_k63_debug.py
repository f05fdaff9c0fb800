import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, blight_amd, bench
dev = torch.device("cuda", 0)
res = []
for (k, m, n, b, nk) in [(63, 9, 14, 8, 50_000_000), (63, 9, 7, 8, 50_000_000), (63, 9, 14, 6, 50_000_000), (63, 9, 14, 8, 5_000_000), (47, 9, 14, 8, 50_000_000), (31, 9, 14, 8, 50_000_000)]:
    ascii_, offs, total = bench.synth_unitigs(torch, dev, nk, k, 5)
    torch.cuda.synchronize()
    kb = 16 if k > 32 else 8
    ix = blight_amd.kmer_Set_Light(k, m, n, 6, 1, b)
    ix.construct_index_from_device(ascii_.data_ptr(), offs.data_ptr(), offs.numel() - 1, total)
    nq = 4_000_000
    q, is_pos = bench.synth_queries(torch, ix, dev, ascii_, offs, total, k, nq, 7, kb // 8)
    ids = torch.empty(nq, device=dev, dtype=torch.int64)
    r = {"cfg": [k, m, n, b, nk], "st": {x: ix.stats()[x] for x in ("number_kmer", "largest_bucket", "largest_mphf", "max_levels_used", "fallback_keys")}}
    for pol in (-1, 1):
        ix.set_query_binning(pol)
        ix.query_kmers_device(q.data_ptr(), nq, ids.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        r["found_pol%d" % pol] = int(((ids >= 0) & is_pos).sum().item())
    r["npos"] = int(is_pos.sum().item())
    res.append(r)
    print(json.dumps(r), flush=True)
    ix.close(); del ascii_, q
