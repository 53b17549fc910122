"""The quality gate and the whole ``fastqdedup`` command line on the GPU path, against
the reference's known answers and an oracle-side restatement of deduplicate_cluster. GPU only."""
import gzip
import logging
import math
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def F():
    import fastqdedup_amd
    return fastqdedup_amd


def test_average_error_rate_known_answers(F):
    """reference tests/test__fastq.py:6-25"""
    assert F.average_error_rate(chr(10) + chr(30), phred_offset=0) == 0.0505
    assert F.average_error_rate(chr(43) + chr(63)) == 0.0505
    for i in list(range(33)) + [127]:
        with pytest.raises(ValueError, match="outside of valid phred range"):
            F.average_error_rate(chr(i))
    with pytest.raises(ValueError, match="phred_scores must be ASCII encoded"):
        F.average_error_rate(chr(128))
    assert math.isnan(F.average_error_rate(""))


def test_quality_means_bit_exact(F, oracle):
    """Batch kernel vs the oracle (itself equal to the reference's _fastq on 20 k strings)."""
    rng = random.Random(9)
    strs = ["".join(chr(rng.randint(33, 126)) for _ in range(rng.choice([0, 1, 7, 50, 150, 301])))
            for _ in range(3000)] + [chr(c) for c in range(33, 127)]
    raw = np.frombuffer("".join(strs).encode(), dtype=np.uint8)
    off = np.concatenate([[0], np.cumsum([len(s) for s in strs])]).astype(np.uint64)
    ctx = F.Context(0)
    for thr in (0.001, 0.05):
        flags, means, nd = ctx.quality_filter(raw, off, threshold=thr, want_means=True)
        want = np.array([oracle.average_error_rate(s) for s in strs])
        assert np.array_equal(means, want, equal_nan=True)
        assert flags.tolist() == [0 if m > thr else 1 for m in want]
        assert nd == int((flags == 0).sum())
    fixed = ["".join(chr(rng.randint(40, 75)) for _ in range(32)) for _ in range(5000)]
    flags, means, _ = ctx.quality_filter(np.frombuffer("".join(fixed).encode(), dtype=np.uint8), None, 32,
                                         threshold=0.001, want_means=True)
    assert means.tolist() == [oracle.average_error_rate(s) for s in fixed]
    with pytest.raises(ValueError):
        ctx.quality_filter(np.frombuffer(b"III II", dtype=np.uint8), None, 3)


def test_quality_kernel_matches_reference_fixture(F):
    """fqd_quality_filter against tests/golden/fastq_error_rates.json: values the reference's own
    _fastq.average_error_rate (_fastqmodule.c:38-76) gave, compared bit for bit."""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fastq_error_rates.json")) as fh:
        fx = json.load(fh)
    ctx = F.Context(0)
    rows = [{"phred": ch, "offset": 33, "value": v} for ch, v in fx["single"].items()] + fx["strings"]
    for off in sorted({r["offset"] for r in rows}):
        sel = [r for r in rows if r["offset"] == off]
        strs = [r["phred"] for r in sel]
        raw = np.frombuffer("".join(strs).encode("latin-1") or b"\0", dtype=np.uint8)[: sum(map(len, strs))]
        offs = np.concatenate([[0], np.cumsum([len(s) for s in strs])]).astype(np.uint64)
        _, means, _ = ctx.quality_filter(raw if raw.size else np.zeros(1, np.uint8), offs, phred_offset=off,
                                         want_means=True)
        for r, m in zip(sel, means.tolist()):
            assert (math.isnan(m) if r["value"] == "nan" else float(m).hex() == r["value"]), r
    for r in fx["errors"]:
        with pytest.raises(ValueError) as err:
            F.average_error_rate(r["phred"], phred_offset=r["offset"])
        assert str(err.value) == r["message"]


def _fq(path, recs):
    data = "".join(f"@{n}\n{s}\n+\n{q}\n" for n, s, q in recs).encode()
    (gzip.open if path.endswith(".gz") else open)(path, "wb").write(data)


def _names(path):
    lines = (gzip.open if path.endswith(".gz") else open)(path, "rb").read().decode().split("\n")
    return [l[1:].split()[0] for l in lines[0::4] if l]


def _oracle_cli(oracle, files_recs, slices, d, thr, method, edit):
    """deduplicate_cluster (__init__.py:209-288) restated on the oracle: returns kept record numbers."""
    n = min(len(r) for r in files_recs)
    keys, passed = [], []
    for i in range(n):
        q = "".join(files_recs[f][i][2][slices[f] if slices else slice(None)] for f in range(len(files_recs)))
        keys.append("".join(files_recs[f][i][1][slices[f] if slices else slice(None)]
                            for f in range(len(files_recs))))
        passed.append(0 if (thr < 1.0 and oracle.average_error_rate(q) > thr) else 1)
    enc = [k.encode() for k in keys]
    raw = np.frombuffer(b"".join(enc) or b"\0", dtype=np.uint8)
    off = np.concatenate([[0], np.cumsum([len(e) for e in enc])]).astype(np.uint64)
    out = oracle.dedup(raw, off, np.array(passed, dtype=np.uint32), max_distance=d, use_edit_distance=edit,
                       method=method)
    return out["kept_read_ids"].tolist(), n - sum(passed), sum(passed), out["n_clusters"]


def test_pass2_rule_through_the_cli(F, tmp_path, caplog):
    """SURVEY.md 8a-S4, measured with the reference: output = r1, r6; the three INFO lines."""
    recs = [("r1", "AAAAAAAA", "!" * 8), ("r2", "AAAAAAAA", "I" * 8), ("r3", "AAAAAAAC", "I" * 8),
            ("r4", "AAAAAAAA", "I" * 8), ("r5", "GGGGGGGG", "!" * 8), ("r6", "TTTTTTTT", "I" * 8),
            ("r7", "TTTTTTTN", "I" * 8)]
    src, dst = str(tmp_path / "in.fastq"), str(tmp_path / "out.fastq")
    _fq(src, recs)
    with caplog.at_level(logging.INFO, logger="fastqdedup"):
        F.deduplicate_cluster([src], [dst], None)
    assert _names(dst) == ["r1", "r6"]
    text = caplog.text
    assert "2 records out of 7 records had an error rate higher than 0.001 and were discarded." in text
    assert "Processed 5 sequences." in text
    assert "Found 2 distinct reads in 2 clusters." in text


def test_debug_log_prints_the_references_trie_table(F, oracle, tmp_path, caplog):
    """`-v`: the layer table of the reference's DEBUG log (__init__.py:133-157, :260-264), i.e. the
    census of the trie built from the reads that passed the filter -- here from the device table,
    compared with the oracle's real trie. Reads with an 'R' make the alphabet grow lazily."""
    from fastqdedup_amd.cli import trie_stats
    rng = random.Random(8)
    mols = ["".join(rng.choice("ACGT") for _ in range(24)) for _ in range(150)]
    recs = []
    for i in range(1500):
        s = "".join(rng.choice("ACGTNR") if rng.random() < 0.01 else ch for ch in rng.choice(mols))
        recs.append((f"r{i}", s, "".join(chr(rng.choice([73, 73, 73, 40])) for _ in range(24))))
    src, dst = str(tmp_path / "in.fastq"), str(tmp_path / "out.fastq")
    _fq(src, recs)
    with caplog.at_level(logging.DEBUG, logger="fastqdedup"):
        logging.getLogger("fastqdedup").setLevel(logging.DEBUG)
        try:
            F.deduplicate_cluster([src], [dst], None, max_average_error_rate=0.05)
        finally:
            logging.getLogger("fastqdedup").setLevel(logging.NOTSET)
    trie = oracle.Trie("ACGTN")
    for _, s, q in recs:
        if not oracle.average_error_rate(q) > 0.05:
            trie.add_sequence(s)
    assert "Calculated stats." in caplog.text
    assert trie_stats(trie) in caplog.text


@pytest.mark.parametrize("args", [
    dict(spec="16,16", d=1, thr=0.001, method="directional", edit=False, gz=True),
    dict(spec=None, d=2, thr=1.0, method="adjacency", edit=False, gz=False),
    dict(spec="4:20,::2", d=1, thr=0.01, method="highest_count", edit=True, gz=True),
])
def test_cli_end_to_end_matches_oracle(F, oracle, tmp_path, args):
    rng = random.Random(3)
    mols = [("".join(rng.choice("ACGT") for _ in range(60)), "".join(rng.choice("ACGT") for _ in range(60)))
            for _ in range(400)]
    r1, r2 = [], []
    for i in range(3000):
        a, b = rng.choice(mols)
        a = "".join(rng.choice("ACGTN") if rng.random() < 0.004 else ch for ch in a)
        b = "".join(rng.choice("ACGTN") if rng.random() < 0.004 else ch for ch in b)
        qual = lambda: "".join(chr(rng.choice([73, 73, 73, 60, 45, 35])) for _ in range(60))
        r1.append((f"frag{i}/1 desc", a, qual()))
        r2.append((f"frag{i}/2", b, qual()))
    ext = ".fastq.gz" if args["gz"] else ".fastq"
    f1, f2 = str(tmp_path / ("a" + ext)), str(tmp_path / ("b" + ext))
    _fq(f1, r1)
    _fq(f2, r2 + [("extra/2", "ACGT", "IIII")])          # the longer file is cut by zip()
    o1, o2 = str(tmp_path / ("o1" + ext)), str(tmp_path / ("o2" + ext))
    argv = [f1, f2, "-o", o1, "-o", o2, "-d", str(args["d"]), "-c", args["method"], "-q"]
    if args["spec"]:
        argv += ["-l", args["spec"]]
    argv += ["-E"] if args["thr"] >= 1.0 else ["-e", str(args["thr"])]
    if args["edit"]:
        argv.append("--edit")
    F.main(argv)
    slices = F.length_string_to_slices(args["spec"]) if args["spec"] else None
    kept, _, _, _ = _oracle_cli(oracle, [r1, r2], slices, args["d"], args["thr"], args["method"], args["edit"])
    assert _names(o1) == [r1[i][0].split()[0] for i in kept]
    assert _names(o2) == [r2[i][0] for i in kept]


def test_cli_streams_the_files_in_chunks(F, oracle, tmp_path, monkeypatch):
    """The files are read chunk by chunk in both passes (FQD_FASTQ_CHUNK_RECORDS; default 2 M records):
    chunks of 257 records must give what one chunk gives, on files of different lengths, ragged keys,
    quality-failed first holders spread over the chunks."""
    rng = random.Random(5)
    mols = ["".join(rng.choice("ACGT") for _ in range(rng.randint(30, 36))) for _ in range(300)]
    r1 = []
    for i in range(2500):
        s = "".join(rng.choice("ACGTN") if rng.random() < 0.01 else ch for ch in rng.choice(mols))
        r1.append((f"q{i}", s, "".join(chr(rng.choice([73, 73, 35])) for _ in range(len(s)))))
    f1 = str(tmp_path / "a.fastq.gz")
    _fq(f1, r1)
    outs = {}
    for chunk in ("257", "100000"):
        monkeypatch.setenv("FQD_FASTQ_CHUNK_RECORDS", chunk)
        o = str(tmp_path / f"o{chunk}.fastq")
        F.main([f1, "-o", o, "-d", "1", "-e", "0.2", "-q"])
        outs[chunk] = _names(o)
    assert outs["257"] == outs["100000"]
    kept, _, _, _ = _oracle_cli(oracle, [r1], None, 1, 0.2, "directional", False)
    assert outs["257"] == [r1[i][0] for i in kept]


def test_cli_detects_unsynced_files(F, tmp_path):
    from fastqdedup_amd.fastq import FastqFormatError
    f1, f2 = str(tmp_path / "a.fastq"), str(tmp_path / "b.fastq")
    _fq(f1, [("x/1", "ACGT", "IIII"), ("y/1", "ACGT", "IIII")])
    _fq(f2, [("x/2", "ACGT", "IIII"), ("z/2", "ACGT", "IIII")])
    with pytest.raises(FastqFormatError, match="not mates"):
        F.deduplicate_cluster([f1, f2], [str(tmp_path / "o1"), str(tmp_path / "o2")], None)
    with pytest.raises(ValueError, match="Amount of output files"):
        F.deduplicate_cluster([f1, f2], [str(tmp_path / "o1")], None)
