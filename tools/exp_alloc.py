import time, torch
dev = torch.device("cuda", 0)
torch.cuda.synchronize()
for gb in (1, 4, 20, 20, 4):
    t = time.time(); x = torch.empty(int(gb * 1e9), dtype=torch.uint8, device=dev); torch.cuda.synchronize(); t1 = time.time() - t
    t = time.time(); x.zero_(); torch.cuda.synchronize(); t2 = time.time() - t
    t = time.time(); x.zero_(); torch.cuda.synchronize(); t3 = time.time() - t
    t = time.time(); del x; torch.cuda.empty_cache(); torch.cuda.synchronize(); t4 = time.time() - t
    print("%2d GB: alloc %.1f ms, first zero %.1f ms, second zero %.1f ms, free %.1f ms" % (gb, t1 * 1e3, t2 * 1e3, t3 * 1e3, t4 * 1e3), flush=True)
