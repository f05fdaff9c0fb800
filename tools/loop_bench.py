#!/usr/bin/env python3
"""loop_bench.py -- the C4-size rehearsal of the sharded exchange on ONE GPU (what profiles/r0N_loopback_rehearsal_c4.txt holds): the
index as `world` whole-input shards, one thread per rank, the library's loopback transport (flows = device copies on the same card);
every rank asks 1e8 / world fresh k-mers per step (three batches rotate); ms per step = the slowest rank.  No scaling number -- the
wire is a device copy -- but everything above the transport is the code that runs over RCCL: partition per rank, count exchange,
piece pipeline, the owners' kernels over all senders' slots, IDs back, gathers.  --ship 0/1: minimizers stay home / travel.

  python tools/loop_bench.py [--worlds 2,4,8] [--ship -1] [--pieces 0] [--steps 12] [--workload C4]"""
import argparse
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("BLIGHT_SCRATCH_CACHE_GB", "48")
os.environ.setdefault("BLIGHT_LOOPBACK_TIMEOUT", "120")


def run_ranks(world, fn):
    res, err = [None] * world, [None] * world

    def wrap(r):
        try:
            res[r] = fn(r)
        except BaseException as e:  # noqa: BLE001
            err[r] = e
    th = [threading.Thread(target=wrap, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    for e in err:
        if e is not None:
            raise e
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--worlds", default="2,4,8")
    ap.add_argument("--ship", default="-1", help="comma list of ship_minimizers settings to run (-1 = the library's default for the world)")
    ap.add_argument("--pieces", type=int, default=0)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--workload", default="C4")
    ap.add_argument("--queries", type=int, default=0)
    args = ap.parse_args()
    import torch
    import bench
    import blight_amd as B
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    k, m, n, b, n_kmers, nq = bench.WORKLOADS[args.workload]
    nq = args.queries or nq
    kw = 1 if k <= 32 else 2
    ascii_, offs, total = bench.synth_unitigs(torch, dev, n_kmers, k, 4)
    single = B.kmer_Set_Light(k, m, n, min(n, 6), 1, b, device=0)
    single.construct_index_from_device(ascii_.data_ptr(), offs.data_ptr(), offs.numel() - 1, total)
    time.sleep(1.0)  # (the filter's background passes)
    stream = torch.cuda.current_stream().cuda_stream
    # single index: the same 1e8 queries per step
    qs = [bench.synth_queries(torch, single, dev, ascii_, offs, total, k, nq, 44 + 1000 * j, kw, 0.5)[0] for j in range(3)]
    want = [torch.empty(nq, dtype=torch.int64, device=dev) for _ in range(3)]
    for j in range(3):
        single.query_kmers_device(qs[j].data_ptr(), nq, want[j].data_ptr(), stream)
    torch.cuda.synchronize()
    ms = []
    for i in range(8):
        t0 = time.perf_counter()
        single.query_kmers_device(qs[i % 3].data_ptr(), nq, want[i % 3].data_ptr(), stream)
        torch.cuda.synchronize()
        ms.append((time.perf_counter() - t0) * 1e3)
    print("single index, %d queries per step: ms per step %s" % (nq, " ".join("%.2f" % x for x in ms)), flush=True)
    for world in [int(x) for x in args.worlds.split(",")]:
        shards = []
        for r in range(world):
            a = B.kmer_Set_Light(k, m, n, min(n, 6), 1, b, device=0)
            a.set_shard(r, world)
            a.construct_index_from_device(ascii_.data_ptr(), offs.data_ptr(), offs.numel() - 1, total)
            shards.append(a)
        torch.cuda.synchronize()
        time.sleep(1.0)  # (the filters' background passes)
        per = nq // world
        for ship in [int(x) for x in args.ship.split(",")]:
            uid = B.comm_loopback_id()
            comms = [None] * world

            def mk(r):
                comms[r] = B.Comm(uid, r, world, device=0)
                comms[r].set_tuning("ship_minimizers", ship)
                if args.pieces:
                    comms[r].set_tuning("pieces", args.pieces)
            run_ranks(world, mk)
            streams = [torch.cuda.Stream(device=dev) for _ in range(world)]
            outs = [[torch.empty(per, dtype=torch.int64, device=dev) for _ in range(3)] for _ in range(world)]
            torch.cuda.synchronize()
            times = [[0.0] * args.steps for _ in range(world)]
            bar = threading.Barrier(world)

            def work(r):
                torch.cuda.set_device(0)
                for i in range(args.steps):
                    j = i % 3
                    bar.wait()
                    t0 = time.perf_counter()
                    comms[r].query_kmers_sharded_device(shards[r], qs[j][r * per:(r + 1) * per].data_ptr(), per, outs[r][j].data_ptr(), streams[r].cuda_stream)
                    streams[r].synchronize()
                    bar.wait()  # a step ends when its slowest rank has its answers
                    times[r][i] = (time.perf_counter() - t0) * 1e3
            run_ranks(world, work)
            equal = all(bool(torch.equal(outs[r][j], want[j][r * per:(r + 1) * per])) for r in range(world) for j in range(3))
            step_ms = [max(times[r][i] for r in range(world)) for i in range(args.steps)]
            print("loopback world %d, ship_minimizers %d, %d queries per rank and step: ms per step %s ; ids equal to the single index: %s" % (
                world, ship, per, " ".join("%.2f" % x for x in step_ms), equal), flush=True)
            for c in comms:
                c.close()
            del outs
        for a in shards:
            a.close()
        del shards
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
