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
