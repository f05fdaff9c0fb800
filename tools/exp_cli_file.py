#!/usr/bin/env python3
"""Write a C3-scale synthetic unitig FASTA and run the reference-style CLI (tools/bench_blight) on it: end-to-end
construct_index(file) + file_query(file) wall times as a user of the reference would see them."""
import os, sys, subprocess, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, bench
dev = torch.device("cuda", 0)
nk = int(sys.argv[1]) if len(sys.argv) > 1 else 230_000_000
k = 31
ascii_, offs, total = bench.synth_unitigs(torch, dev, nk, k, 3)
a = ascii_[:total].cpu().numpy()
o = offs.cpu().numpy()
del ascii_
path = "/tmp/c3_unitigs.fa"
t = time.time()
nseq = len(o) - 1
# record i = ">\n" + seq + "\n": build the output buffer with vectorised copies
out = np.empty(total + 3 * nseq, dtype=np.uint8)
starts = o[:-1] + 3 * np.arange(nseq)
out[starts] = ord(">")
out[starts + 1] = ord("\n")
out[(o[1:] + 3 * np.arange(nseq) + 2)] = ord("\n")
mask = np.ones(out.size, dtype=bool)
mask[starts] = False; mask[starts + 1] = False; mask[o[1:] + 3 * np.arange(nseq) + 2] = False
out[mask] = a
out.tofile(path)
print("wrote %s: %.1f MB, %d records in %.1f s" % (path, out.size / 1e6, nseq, time.time() - t), flush=True)
torch.cuda.empty_cache()
t = time.time()
r = subprocess.run([os.path.join(ROOT, "tools", "bench_blight"), "-g", path, "-k", "31", "-m", "9", "-n", "7", "-s", "6", "-b", "6"], capture_output=True, text=True)
print(r.stdout[-900:], r.stderr[-300:], "cli wall %.2f s" % (time.time() - t), flush=True)
os.remove(path)
