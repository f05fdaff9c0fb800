"""The reference's CLI harness (bench_blight.cpp) rebuilt on the C++ facade include/blight.h: construct + self-query of
the bundled lambda file must print the reference's recap (Good 48,462 / Erroneous 0)."""
import os
import subprocess

import pytest

import golden_util as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_bench_blight_cli_self_query():
    exe = os.path.join(ROOT, "tools", "bench_blight")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tools"), "bench_blight"])
    out = subprocess.run([exe, "-g", G.LAMBDA, "-k", "31", "-m", "7", "-n", "5", "-s", "3", "-t", "1", "-b", "6"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "Kmer in graph: 48,462" in out.stdout
    assert "Super Kmer in graph: 3,708" in out.stdout
    assert "Largest MPHF: 3,036" in out.stdout and "Largest Bucket: 1,337" in out.stdout
    assert "Good kmer: 48,462" in out.stdout and "Erroneous kmers: 0" in out.stdout
    assert "Query performed: 48,462" in out.stdout
    # ctor errors surface as the reference's std::invalid_argument text
    bad = subprocess.run([exe, "-g", G.LAMBDA, "-k", "31", "-m", "10", "-n", "5", "-s", "3"], capture_output=True, text=True, timeout=60)
    assert bad.returncode != 0 and "minimizer_length must be odd" in bad.stderr


def test_bench_blight_cli_dump_then_load(tmp_path):
    """-d writes the index image, a second process -l loads it and answers the same file identically."""
    exe = os.path.join(ROOT, "tools", "bench_blight")
    img = str(tmp_path / "lambda.blmx")
    common = ["-k", "31", "-m", "7", "-n", "5", "-s", "3", "-t", "1", "-b", "6"]
    a = subprocess.run([exe, "-g", G.LAMBDA, "-d", img] + common, capture_output=True, text=True, timeout=300)
    assert a.returncode == 0, a.stderr
    b = subprocess.run([exe, "-q", G.LAMBDA, "-l", img] + common, capture_output=True, text=True, timeout=300)
    assert b.returncode == 0, b.stderr
    assert "Kmer in graph" not in b.stdout  # nothing was constructed
    assert "Good kmer: 48,462" in b.stdout and "Erroneous kmers: 0" in b.stdout
    wrong = subprocess.run([exe, "-q", G.LAMBDA, "-l", img, "-k", "31", "-m", "7", "-n", "6", "-s", "3", "-b", "6"],
                           capture_output=True, text=True, timeout=60)
    assert wrong.returncode != 0 and "differ" in wrong.stderr


def test_per_read_calls_from_openmp_threads(tmp_path):
    """tools/bench_reads: one kmer_Set_Light::query_sequence_hash call per read from 8 OpenMP threads through the C++ facade --
    the calling pattern of the reference's applications (Colored_De_Bruijn_graph_snippet.cpp:122-157) -- against the oracle:
    found / absent counts, an order-independent checksum over (read, k-mer, ID), and the atomic number_query counter."""
    import json

    import numpy as np
    from oracle import oracle as O
    exe = os.path.join(ROOT, "tools", "bench_reads")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tools"), "bench_reads"])
    c = G.Case("lambda_k31_m7_n5_b6")
    rng = np.random.default_rng(11)
    genome = c.seqs[3]
    reads = []
    for i in range(4000):
        if i % 3 == 2:
            reads.append("".join("ACGT"[j] for j in rng.integers(0, 4, 150)))
        else:
            a = int(rng.integers(0, len(genome) - 150))
            reads.append(genome[a:a + int(rng.integers(31, 150))])
    reads.append("ACGT")  # shorter than k: a call that returns nothing
    rf = tmp_path / "reads.txt"
    rf.write_text("\n".join(reads) + "\n")
    orc = O.Oracle(c.k, c.m, c.n, c.b, c.seqs)
    found = absent = checksum = kmers = 0
    for i, r in enumerate(reads):
        ids = orc.query_sequence(r) if len(r) >= c.k else np.zeros(0, np.int64)
        kmers += ids.size
        found += int((ids >= 0).sum())
        absent += int((ids < 0).sum())
        for j, v in enumerate(ids.tolist()):
            checksum = (checksum + (i * 1000003 + j + 1) * ((v + 2) % (1 << 64))) % (1 << 64)
    for threads in ("8", "1"):
        out = subprocess.run([exe, "-g", G.LAMBDA, "-q", str(rf), "-k", "31", "-m", "7", "-n", "5", "-s", "3", "-b", "6", "-t", threads],
                             capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr
        res = json.loads(out.stdout.strip().splitlines()[-1])
        assert (res["reads"], res["kmers"], res["found"], res["absent"], res["checksum"]) == (len(reads), kmers, found, absent, checksum)
        assert res["number_query"] == kmers + max(0, len(reads[0]) - 31 + 1)  # + the warm-up call


def test_scalar_and_per_read_calls_from_openmp_threads():
    """kmer_Set_Light::query_kmer_hash(kmer_t) -- ONE k-mer per call (reference blight.cpp:545-550) -- from 8 OpenMP threads through the
    C++ facade (tools/libbench_calls.so attaches it to a handle of this process): every answer equals the oracle's, also while a
    second team of threads issues one query_sequence_hash call per read on the same index, so that k-mer and read requests share
    launches of the small-call combiner."""
    import ctypes
    import threading

    import numpy as np
    import blight_amd as B
    from oracle import oracle as O
    libpath = os.path.join(ROOT, "tools", "libbench_calls.so")
    if not os.path.exists(libpath):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tools"), "libbench_calls.so"])
    lib = ctypes.CDLL(libpath)
    u64p, i64p = ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_int64)
    lib.blight_bench_kmer_calls.argtypes = [ctypes.c_void_p, u64p, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, i64p]
    lib.blight_bench_read_calls.argtypes = [ctypes.c_void_p, ctypes.c_char_p, u64p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 5
    c = G.Case("lambda_k31_m7_n5_b6")
    orc = O.Oracle(c.k, c.m, c.n, c.b, c.seqs)
    ix = B.kmer_Set_Light(c.k, c.m, c.n, 3, 1, c.b)
    ix.construct_index_from_sequences(c.seqs)
    ix.set_tuning("absent_filter", 1)
    rng = np.random.default_rng(5)
    pos = orc.canonical_kmers(c.seqs[3])[::7][:6000]
    neg = np.array([min(x, O.p_rc(x, c.k, 8)) for x in (int(v) & ((1 << 62) - 1) for v in rng.integers(0, 1 << 62, 6000))], dtype=np.uint64)
    q = np.ascontiguousarray(np.concatenate([pos, neg])[rng.permutation(len(pos) + len(neg))])
    want = orc.query_kmers(q)
    assert (want >= 0).sum() >= len(pos)

    def scalar(threads, out):
        got = np.empty(len(q), dtype=np.int64)
        secs, found, cs = ctypes.c_double(), ctypes.c_uint64(), ctypes.c_uint64()
        rc = lib.blight_bench_kmer_calls(ix._h, q.ctypes.data_as(u64p), len(q), threads, ctypes.byref(secs), ctypes.byref(found), ctypes.byref(cs),
                                         got.ctypes.data_as(i64p))
        out.append((rc, got, found.value))

    for threads in (8, 1):
        res = []
        scalar(threads, res)
        rc, got, found = res[0]
        assert rc == 0 and np.array_equal(got, want) and found == int((want >= 0).sum()), threads
    # both kinds of call at once
    genome = c.seqs[3]
    reads = [genome[a:a + 150] for a in rng.integers(0, len(genome) - 150, 3000).tolist()] + ["".join("ACGT"[j] for j in rng.integers(0, 4, 150)) for _ in range(1000)]
    chars = "".join(reads).encode()
    offs = (np.arange(len(reads) + 1, dtype=np.uint64) * 150)
    rwant = np.concatenate([orc.query_sequence(r) for r in reads])
    res, rres = [], []

    def per_read():
        secs = ctypes.c_double()
        r = [ctypes.c_uint64() for _ in range(4)]
        rc = lib.blight_bench_read_calls(ix._h, chars, offs.ctypes.data_as(u64p), len(reads), 4, 1, ctypes.byref(secs), *[ctypes.byref(x) for x in r])
        rres.append((rc, r[0].value, r[1].value, r[2].value))

    th = [threading.Thread(target=scalar, args=(4, res)), threading.Thread(target=per_read)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert res[0][0] == 0 and np.array_equal(res[0][1], want)
    assert rres[0] == (0, rwant.size, int((rwant >= 0).sum()), int((rwant < 0).sum()))
    ix.close()


def test_shard_query_cpp_world1(tmp_path):
    """tools/shard_query: the multi-GPU path from C++ only (kmer_Set_Light::communicator of include/blight.h: construct_index from
    this rank's part of the file over RCCL, query_sequences_hash through the native exchange), here with a world of one rank.
    The reads are the indexed file itself plus foreign reads: counts and the (read, position, ID) checksum equal the oracle's."""
    import json

    import numpy as np
    from oracle import oracle as O
    exe = os.path.join(ROOT, "tools", "shard_query")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tools"), "shard_query"])
    c = G.Case("lambda_k31_m7_n5_b6")
    rng = np.random.default_rng(3)
    reads = list(c.seqs) + ["".join("ACGT"[j] for j in rng.integers(0, 4, 120)) for _ in range(500)] + ["ACGT"]
    rf = tmp_path / "reads.txt"
    rf.write_text("\n".join(reads) + "\n")
    orc = O.Oracle(c.k, c.m, c.n, c.b, c.seqs)
    kmers = found = absent = checksum = 0
    for i, r in enumerate(reads):
        ids = orc.query_sequence(r) if len(r) >= c.k else np.zeros(0, np.int64)
        kmers += ids.size
        found += int((ids >= 0).sum())
        absent += int((ids < 0).sum())
        for j, v in enumerate(ids.tolist()):
            checksum = (checksum + (i * 1000003 + j + 1) * ((v + 2) % (1 << 64))) % (1 << 64)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([exe, "-g", G.LAMBDA, "-q", str(rf), "-k", "31", "-m", "7", "-n", "5", "-b", "6", "-r", "0", "-w", "1",
                          "-i", str(tmp_path / "comm.id")], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert (res["number_kmer"], res["reads"], res["kmers"], res["found"], res["absent"], res["checksum"]) == (48462, len(reads), kmers, found, absent, checksum)
    assert found >= 48462


@pytest.mark.parametrize("world", [2, 3])
def test_shard_query_cpp_loopback_ranks(tmp_path, world):
    """tools/shard_query --transport loopback --world N: the whole multi-GPU path from C++ (construct_index from each rank's part of
    the file, halo exchange, query_sequences_hash through the native exchange) with N ranks as N threads of one process on the box's one
    GPU.  The ranks' counts and (read, position, ID) checksums add up to the oracle's over all reads."""
    import json

    import numpy as np
    from oracle import oracle as O
    exe = os.path.join(ROOT, "tools", "shard_query")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tools"), "shard_query"])
    c = G.Case("lambda_k31_m7_n5_b6")
    rng = np.random.default_rng(5 + world)
    reads = list(c.seqs) + ["".join("ACGT"[j] for j in rng.integers(0, 4, 120)) for _ in range(700)] + ["ACGT"]
    rf = tmp_path / "reads.txt"
    rf.write_text("\n".join(reads) + "\n")
    orc = O.Oracle(c.k, c.m, c.n, c.b, c.seqs)
    kmers = found = absent = checksum = 0
    for i, r in enumerate(reads):
        ids = orc.query_sequence(r) if len(r) >= c.k else np.zeros(0, np.int64)
        kmers += ids.size
        found += int((ids >= 0).sum())
        absent += int((ids < 0).sum())
        for j, v in enumerate(ids.tolist()):
            checksum = (checksum + (i * 1000003 + j + 1) * ((v + 2) % (1 << 64))) % (1 << 64)
    env = dict(os.environ, BLIGHT_LOOPBACK_TIMEOUT="90")
    out = subprocess.run([exe, "-g", G.LAMBDA, "-q", str(rf), "-k", "31", "-m", "7", "-n", "5", "-b", "6", "--transport", "loopback", "--world", str(world)],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    res = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(res) == world and sorted(r["rank"] for r in res) == list(range(world))
    assert all(r["transport"] == "loopback" and r["number_kmer"] == 48462 for r in res)
    assert sum(r["reads"] for r in res) == len(reads)
    assert (sum(r["kmers"] for r in res), sum(r["found"] for r in res), sum(r["absent"] for r in res)) == (kmers, found, absent)
    assert sum(r["checksum"] for r in res) % (1 << 64) == checksum


def test_bench_watchdog_ends_a_run_that_does_not_finish():
    """RCCL has no timeout: an N > 1 run of bench.py whose collective never completes must not hang until somebody kills it.  The
    watchdog (BLIGHT_BENCH_WATCHDOG_S) ends every rank by itself and rank 0 leaves one JSON line with a null value and the reason.
    Driven here through the world-1 rehearsal of the N > 1 code path with a limit far below the time the workload takes."""
    import json
    import sys
    env = dict(os.environ, BLIGHT_BENCH_WATCHDOG_S="2")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rehearse-sharded", "--workload", "C3", "--cpu-seconds", "0", "--reads", "0",
                        "--no-fasta-leg", "--no-host-legs", "--steps", "200000", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 4, (r.returncode, r.stderr[-400:])
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["value"] is None and "watchdog" in line["error"]
    assert "giving up" in r.stderr
