import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, blight_amd, bench
k, nk = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda", 0)
for seed in (4, 5):
    ascii_, offs, total = bench.synth_unitigs(torch, dev, nk, k, seed)
    torch.cuda.synchronize()
    lens = offs[1:] - offs[:-1]
    print("seed", seed, "nseq", lens.numel(), "total", total, "min len", int(lens.min()), "max len", int(lens.max()), "kmers", int((lens - (k - 1)).sum()),
          "offs0", int(offs[0]), "bad chars", int(((ascii_[:total] != 65) & (ascii_[:total] != 67) & (ascii_[:total] != 71) & (ascii_[:total] != 84)).sum()), flush=True)
    bad = (lens < k).nonzero().flatten()
    print("  short count", bad.numel(), bad[:5].tolist(), [int(lens[i]) for i in bad[:5].tolist()], flush=True)
    del ascii_, offs
