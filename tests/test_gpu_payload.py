"""Payload applications: coloured / abundance de Bruijn graph on top of the k-mer IDs, reference
Colored_De_Bruijn_graph_snippet.cpp:113-157 and Abundance_De_Bruijn_graph_snippet.cpp:117-157.  Expected values come from
the CPU oracle's query_sequence_hash on every line that starts with A/C/G/T, accumulated with numpy."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _rand_seq(rng, n):
    return "".join("ACGT"[i] for i in rng.integers(0, 4, n))


def _make_inputs(tmp_path, rng, k):
    graph = [_rand_seq(rng, k - 1 + int(rng.geometric(1 / 50))) for _ in range(600)]
    gpath = tmp_path / "graph.fa"
    gpath.write_text("".join(">u%d\n%s\n" % (i, s) for i, s in enumerate(graph)))
    files = []
    for c in range(3):
        lines = []
        for i in rng.choice(len(graph), size=250, replace=False):
            s = graph[int(i)]
            if rng.random() < 0.3:  # a sub-read, sometimes reverse-complemented
                a = int(rng.integers(0, max(1, len(s) - k)))
                s = s[a:a + k + int(rng.integers(0, 40))]
                if rng.random() < 0.5:
                    s = s[::-1].translate(str.maketrans("ACGT", "TGCA"))
            lines.append(">r\n" + s)
        lines.append(">alien\n" + _rand_seq(rng, 90))      # not in the graph
        lines.append(">short\n" + "ACGT" * 3)              # shorter than k
        lines.append("NNNN" + graph[0])                    # does not start with A/C/G/T: skipped entirely
        if c == 0:
            lines += [graph[1]] * 300                      # abundance wraps at 256
        p = tmp_path / ("color%d.fa" % c)
        p.write_text("\n".join(lines) + "\n")
        files.append(str(p))
    return graph, str(gpath), files


def _expected(orc, files, nk, k):
    colors = np.zeros((nk, len(files)), dtype=bool)
    counts = np.zeros(nk, dtype=np.int64)
    found = []
    for c, f in enumerate(files):
        n_found = 0
        for line in open(f).read().split("\n"):
            if line[:1] in ("A", "C", "G", "T") and len(line) >= k:
                ids = orc.query_sequence(line)
                ids = ids[ids >= 0]
                n_found += ids.size
                colors[ids, c] = True
                np.add.at(counts, ids, 1)
        found.append(n_found)
    return colors, (counts % 256).astype(np.uint8), found


def test_colors_and_abundance_against_oracle(tmp_path):
    import blight_amd as B
    from oracle import oracle as O
    k, m, n, b = 31, 9, 7, 6
    rng = np.random.default_rng(17)
    graph, gpath, files = _make_inputs(tmp_path, rng, k)
    ix = B.kmer_Set_Light(k, m, n, 3, 1, b)
    ix.construct_index(gpath)
    orc = O.Oracle(k, m, n, b, graph)
    want_colors, want_counts, want_found = _expected(orc, files, ix.number_kmer, k)
    col = B.Payload(ix, "colors", ncolors=len(files))
    ab = B.Payload(ix, "abundance")
    for c, f in enumerate(files):
        found, absent = col.add_file(f, c)
        assert found == want_found[c] and absent > 0
        ab.add_file(f)
    assert np.array_equal(col.export(), want_colors)
    got = ab.export()
    assert np.array_equal(got, want_counts)
    assert want_counts.max() >= 44 and (np.bincount(want_counts)[1:] > 0).sum() > 3  # wrapped and non-trivial
    with pytest.raises(ValueError):
        col.add_file(files[0], 7)  # colour out of range
    # IDs that are already on the device
    import torch
    ids = torch.from_numpy(orc.query_sequence(graph[5])).cuda()
    ab2 = B.Payload(ix, "abundance")
    ab2.add_ids_device(ids.data_ptr(), ids.numel(), 0, torch.cuda.current_stream().cuda_stream)
    ab2.add_ids_device(ids.data_ptr(), ids.numel(), 0, torch.cuda.current_stream().cuda_stream)
    e = ab2.export()
    assert int(e.sum()) == 2 * ids.numel() and set(np.unique(e)) <= {0, 2}


def test_payload_dbg_cli(tmp_path):
    """tools/payload_dbg = the two reference snippets: outcolor / outabundance line for line."""
    from oracle import oracle as O
    k, m, n, b = 31, 9, 7, 6
    rng = np.random.default_rng(23)
    graph, gpath, files = _make_inputs(tmp_path, rng, k)
    fof = tmp_path / "fof.txt"
    fof.write_text("\n".join(files) + "\n")
    qpath = tmp_path / "query.fa"
    qlines = [graph[i] for i in (3, 40, 41)] + [_rand_seq(rng, 60), ">header"]
    qpath.write_text("\n".join(qlines) + "\n")
    exe = os.path.join(ROOT, "tools", "payload_dbg")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tools"), "payload_dbg"])
    orc = O.Oracle(k, m, n, b, graph)
    want_colors, want_counts, _ = _expected(orc, files, int(orc.stats["number_kmer"]), k)
    common = ["-g", gpath, "-o", str(fof), "-q", str(qpath), "-k", str(k), "-m", str(m), "-n", str(n), "-s", "3", "-b", str(b)]
    for mode, outname in (([], "outcolor"), (["-a"], "outabundance")):
        r = subprocess.run([exe] + common + mode, cwd=tmp_path, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        got = (tmp_path / outname).read_text()
        want = ""
        for line in qlines:
            if line[:1] not in ("A", "C", "G", "T"):
                continue
            for i in orc.query_sequence(line):
                if i >= 0:
                    want += ("".join("%d\t" % c for c in np.nonzero(want_colors[i])[0]) if not mode else "%d\t" % want_counts[i])
                want += "\n"
        assert got == want
